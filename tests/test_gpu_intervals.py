"""Every interval the half-width scans make CONTAINS the reference's key -- checked pair by pair on the device's own numbers.

The round-4 scans (zebra_amd/csrc/zh_approx.hip: scan_mfma_kernel, scan_approx_kernel, scan_approx128_kernel, sweep128h_kernel) do not compute
Metric::distance for the members of a visited leaf (/root/reference/src/database/index/lsh.rs:310-323 with src/distance.rs:23,41,106); they
compute an INTERVAL per (stored row, query) pair and score exactly only what the intervals cannot rule out.  That is the reference's result only
if no interval ever excludes the reference's key.  The end-to-end suites would notice a violated interval only when it happens to change a
top-k; here zh_debug_scan_pairs hands back, for one batch, every scored pair's interval (and, with zh_debug_keep_raw, the scan's raw
{x^ . h^ / sigma_x, |x|^2}) and the test asserts lo <= oracle key <= hi for ALL of them, for every kernel, dimension and key order, on generic,
scaled (2^+-40), subnormal-heavy, integer-valued and near-duplicate rows, and a zero query.

It also MEASURES what zh_approx_bound assumes of the matrix cores: |acc - exact| <= ops * 2u * sum |x^_i h^_i| for the sum of
v_mfma_f32_16x16x32_f16 accumulators ("any order, at most 2u per operation": the ISA documents neither).  The exact sum of the products of the
rounded operands is an integer below 2^53, i.e. exact in float64; the observed maximum is printed and asserted against the assumed constant."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


def set_s128h_variant(monkeypatch, variant):
    """fused: sweep128h_lean_kernel<CH, KINDA> + sweep128h_boundary_kernel<KINDA> with intervals, bounds and the queries' lists inside the sweep
    (round 6; chosen by the library for long leaves, forced here: ZH_S128H_FUSED=1; top_k <= 64); lean: the same kernels writing raw pairs for
    select_tau_kernel (ZH_S128H_FUSED=0); r5: sweep128h_kernel (ZH_S128H_KERNEL=r5); dma: sweep128h_dma_kernel (ZH_S128H_DMA=1)"""
    env = {"fused": {"ZH_S128H_FUSED": "1"}, "lean": {"ZH_S128H_FUSED": "0"}, "r5": {"ZH_S128H_KERNEL": "r5"}, "dma": {"ZH_S128H_DMA": "1"},
           "lean_halves": {"ZH_S128H_FUSED": "0", "ZH_S128H_BYTES": "0"}}[variant]  # (lean reads the copy of BYTES when every row is one of integers 0 .. 255)
    for var in ("ZH_S128H_FUSED", "ZH_S128H_KERNEL", "ZH_S128H_DMA", "ZH_S128H_BYTES"):
        if var in env:
            monkeypatch.setenv(var, env[var])
        else:
            monkeypatch.delenv(var, raising=False)

U = 2.0 ** -24


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def unsortable(s):
    """inverse of the kernels' f32_sortable (bits ^ (sign ? ~0 : 0x80000000)) -> float64 values"""
    s = np.asarray(s, np.uint32)
    u = np.where(s & np.uint32(0x80000000), s ^ np.uint32(0x80000000), ~s)
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def run_batch(torch, ix, Q, k, metric):
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    B = Q.shape[0]
    ids = torch.empty((B, k), dtype=torch.int64, device=dev)
    keys = torch.empty((B, k), dtype=torch.int64, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    ix.search_batch_device(q.data_ptr(), B, k, metric, ids.data_ptr(), keys.data_ptr(), counts.data_ptr())
    torch.cuda.synchronize()
    return ids.cpu().numpy().view(np.uint64), keys.cpu().numpy().view(np.uint64), counts.cpu().numpy()


def oracle_values(X, Q, pairs, kinda):
    """the reference's key of every pair in the scale its interval is made in (zebra_hip.h, zh_debug_pair): float64"""
    val = np.empty(pairs.shape[0], np.float64)
    order = np.argsort(pairs["query"], kind="stable")
    qs = pairs["query"][order]
    starts = np.flatnonzero(np.r_[True, qs[1:] != qs[:-1]])
    ends = np.r_[starts[1:], qs.shape[0]]
    for s, e in zip(starts, ends):
        b = int(qs[s])
        idx = order[s:e]
        rows, inv = np.unique(pairs["row"][idx], return_inverse=True)
        if kinda == 0:
            kk = zo.distance_batch(zo.L2SQ, 0, X[rows], Q[b])
        else:
            kk = zo.distance_batch(zo.COSINE, zo.PARITY if kinda == 2 else zo.CORRECTED, X[rows], Q[b])
        v = zo.key_to_float(kk)[inv]
        if kinda == 2:  # the literal key 1 - distance compares as the bits of the f64: positives ascending, then negatives by magnitude
            v = np.where(v > 0, v, 2.0 - v)
        val[idx] = v
    return val


def f16_copy(V, table_ex=None):
    """the kernels' fp16 copy of the rows of V (qhalf_kernel / row_half_kernel / row_half128_kernel): per row (or for the table) a power-of-two scale
    2^(14 - ex) with max |v| in [2^(ex-1), 2^ex), round to nearest even, subnormal halves written as zeros -> (integer-valued float64 halves, scales)"""
    V = np.asarray(V, np.float32)
    m = np.abs(V).max(axis=1)
    if table_ex is None:
        _, ex = np.frexp(m)
        ex = np.where(m > 0, ex, 14)
    else:
        ex = np.full(V.shape[0], table_ex)
    sigma = np.ldexp(np.float32(1), (14 - ex).astype(np.int32)).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        h = (V * sigma[:, None]).astype(np.float16)
    h = np.where(np.abs(h.astype(np.float32)) < 6.103515625e-05, np.float16(0), h)
    return h.astype(np.float64), sigma.astype(np.float64)


def check_containment(X, Q, info, pairs, qmeta, what):
    kinda = 0 if info["metric"] != 0 else (2 if info["cosine_mode"] == 0 else 1)
    assert info["pairs"] == pairs.shape[0] and (pairs["visit"] != 0xFFFFFFFF).all(), what  # every key slot belongs to a visit
    conv = (pairs["flags"] & 1) != 0
    assert conv.any(), what
    p = pairs[conv]
    val = oracle_values(X, Q, p, kinda)
    unsure = (p["lo"] == 0) & (p["hi"] == 0xFFFFFFFF)
    lo, hi = unsortable(p["lo"]), unsortable(p["hi"])
    ok = unsure | ((lo <= val) & (val <= hi))
    if not ok.all():
        i = int(np.flatnonzero(~ok)[0])
        raise AssertionError(f"{what}: interval [{lo[i]!r}, {hi[i]!r}] does not contain the reference's {val[i]!r} "
                             f"(row {p['row'][i]}, query {p['query'][i]}; {int((~ok).sum())} of {ok.size} pairs)")
    # how tight, for the record: the half-width relative to the value where both are finite
    fin = ~unsure & np.isfinite(lo) & np.isfinite(hi) & (np.abs(val) > 0)
    rel = float(np.median((hi[fin] - lo[fin]) / 2 / np.abs(val[fin]))) if fin.any() else float("nan")
    return {"pairs": int(ok.size), "unsure": int(unsure.sum()), "median_rel_halfwidth": rel}


def check_rounding_model(X, Q, info, pairs, qmeta, d, what, table_ex=None):
    """|raw_s - exact| against what zh_approx_bound assumes of the scan's sum (header of this file) -> the observed maximum in units of u"""
    assert info["raw_kept"] == 1
    kind = info["approx_scan"]
    p = pairs[(pairs["flags"] & 1) != 0]
    qh, qs = f16_copy(Q)
    if kind == 1:    # VALU: f32 rows x fp16 queries, v_fma_mix_f32 chains + butterfly: <= 33 roundings of u on the longest path
        xr, xs = X.astype(np.float64), np.ones(X.shape[0])
        assumed = 33.0
    elif kind == 2:  # matrix cores: fp16 rows (per-row scale) x fp16 queries: ops = 33 d / 128 + 2 operations of <= 2u
        xr, xs = f16_copy(X)
        assumed = 2.0 * (33.0 * (d // 128) + 2.0)
    else:            # sweep128h: fp16 rows under ONE table scale
        xr, xs = f16_copy(X, table_ex)
        assumed = 2.0 * (33.0 + 2.0 + 40.0)
    worst = 0.0
    for b in np.unique(p["query"]):
        if not np.isfinite(qmeta[b]).all():
            continue
        sel = p[p["query"] == b]
        rows = sel["row"]
        ex_sum = xr[rows] @ qh[b]                    # exact: integers (or f32 x integer) summed in float64
        ab_sum = np.abs(xr[rows]) @ np.abs(qh[b])
        got = sel["raw_s"].astype(np.float64) * xs[rows]   # raw_s = acc / sigma_x (a power of two: exact)
        good = np.isfinite(got) & np.isfinite(ex_sum) & (ab_sum > 0) & np.isfinite(sel["raw_a2"])
        if kind != 1:
            good &= np.abs(xr[rows]).max(axis=1) > 0
        if good.any():
            err = np.abs(got[good] - ex_sum[good]) / ab_sum[good] / U
            worst = max(worst, float(err.max()))
    assert worst <= assumed, f"{what}: the scan's sum is {worst:.2f} u * sum|x h| from the exact one; zh_approx_bound assumes <= {assumed:.0f} u"
    return worst, assumed


def special_rows(X, d, rng):
    """the cases VERDICT r4 #2 names, planted in a generic table (the forest is built over the result, so they sit in ordinary leaves)"""
    X = X.copy()
    n = X.shape[0]
    X[100:160] *= np.float32(2.0 ** 40)                      # huge rows (|x|^2 ~ 2^90: still finite in f32)
    X[200:260] *= np.float32(2.0 ** -40)                     # tiny rows
    X[300:360] = np.round(X[300:360] * 4.0)                  # integer-valued rows: ties, exact products
    sub = X[400:460]                                        # subnormal-heavy in fp16: one large element, the rest 2^-13..2^-16 of it
    sub *= np.float32(2.0 ** -14)
    sub[:, 7] = np.float32(3.0)
    X[400:460] = sub
    base = X[:60]
    X[500:560] = base                                       # exact duplicates
    X[600:660] = np.nextafter(base, np.float32(np.inf))     # one ulp away in every coordinate
    X[700:760] = base * np.float32(1.0 + 2.0 ** -12)        # near-duplicates below the fp16 resolution
    X[800:810] = 0.0                                        # zero rows
    X[810:820] *= np.float32(1e18)                          # |x|^2 overflows f32
    return X


def special_queries(Q, X):
    Q = Q.copy()
    Q[0] = X[0]                         # a stored row itself (and its duplicates / near-duplicates)
    Q[1] = 0.0                          # the zero query
    Q[2] = X[110] * np.float32(1.5)     # next to the huge rows
    Q[3] = X[210] * np.float32(0.75)    # next to the tiny rows
    Q[4] = X[310]                       # an integer row
    Q[5] = X[410]                       # a subnormal-heavy row
    Q[6] = -X[3]                        # the antipode: negative cosines (the literal key's second half of the order)
    Q[7] = Q[7] * np.float32(2.0 ** 40)
    Q[8] = Q[8] * np.float32(2.0 ** -40)
    Q[9] = np.round(Q[9] * 3.0)
    return Q


def metrics(za):
    return [("l2sq", za.L2SquaredDistance()), ("l2", za.L2Distance()), ("cos-literal", za.CosineDistance(parity=True)),
            ("cos-corrected", za.CosineDistance(parity=False))]


def build(za, X, d, M, T, mode):
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X)
    ix.set_forest(f.arrays())   # (the oracle's forest: the data holds rows whose planes overflow)
    ix.set_sweep_mode(mode)
    ix.set_hash_mode("dense")
    ix.debug_keep_raw(True)
    return ix, f


@pytest.mark.parametrize("rows", ["copy", "f32"])
@pytest.mark.parametrize("d", [256, 384, 512, 768, 1024])
def test_matrix_core_scan_intervals_contain_the_key(za, torch, d, rows, monkeypatch, record_property):
    """scan_mfma_kernel<d> (zh_set_sweep_mode 4): fp16 rows AND fp16 queries, the sum formed inside v_mfma_f32_16x16x32_f16.  rows = f32
    (ZH_ROW_HALF_META_ONLY=1; what an index does whose fp16 copy does not fit): scan_mfma_kernel<d, true> converts the f32 rows itself, with the
    per-row scales of the copy -- same operand bits, same intervals"""
    if rows == "f32":
        monkeypatch.setenv("ZH_ROW_HALF_META_ONLY", "1")
    else:
        monkeypatch.delenv("ZH_ROW_HALF_META_ONLY", raising=False)
    rng = np.random.default_rng(d)
    n, M, T, k, B = 5000, 400, 8, 10, 40
    X = special_rows(zo.synth_rows(n, d), d, rng)
    Q = special_queries(zo.synth_queries(B, d, n), X)
    ix, f = build(za, X, d, M, T, "approx")
    for name, m in metrics(za):
        ids, keys, counts = run_batch(torch, ix, Q, k, m)
        info, pairs, qmeta = ix.debug_scan_pairs()
        assert info["approx_scan"] == 2 and info["queries"] == B and info["top_k"] == k, info
        copy_bytes = ix.stats()["row_copy_bytes"]
        assert (copy_bytes < n * d) == (rows == "f32"), (rows, copy_bytes)  # (the copy itself is 2 d bytes per row; scales and norms alone 8)
        r = check_containment(X, Q, info, pairs, qmeta, f"scan_mfma_kernel<{d}> {name}")
        w, a = check_rounding_model(X, Q, info, pairs, qmeta, d, f"scan_mfma_kernel<{d}> {name}")
        print(f"scan_mfma_kernel<{d}> {name}: {r['pairs']} pairs, {r['unsure']} uncertain, median half-width {r['median_rel_halfwidth']:.2e} of the value; "
              f"max |acc - exact| = {w:.2f} u * sum|x^ h^| (assumed <= {a:.0f} u), rho = {info['row_rho']:.3e}")
        record_property(f"mfma_rounding_u_d{d}_{name}", w)
        # ... and the answer is the oracle's (the end-to-end claim, on the same batch)
        om, omode = {"l2sq": (zo.L2SQ, 0), "l2": (zo.L2, 0), "cos-literal": (zo.COSINE, zo.PARITY), "cos-corrected": (zo.COSINE, zo.CORRECTED)}[name]
        oi, ok, oc = f.search_batch(Q, k, om, omode)
        assert (counts == oc).all()
        for b in range(B):
            c = int(oc[b])
            assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (name, b)
    ix.close()


@pytest.mark.parametrize("d", [128, 384, 768])
def test_valu_scan_intervals_contain_the_key(za, torch, d):
    """scan_approx_kernel<d, G, KINDA> / scan_approx128_kernel (zh_set_sweep_mode 5): f32 rows, fp16 queries, v_fma_mix_f32"""
    rng = np.random.default_rng(1000 + d)
    n, M, T, k, B = 5000, 400, 8, 10, 40
    X = special_rows(zo.synth_rows(n, d), d, rng)
    Q = special_queries(zo.synth_queries(B, d, n), X)
    ix, f = build(za, X, d, M, T, "approx-valu")
    for name, m in metrics(za):
        run_batch(torch, ix, Q, k, m)
        info, pairs, qmeta = ix.debug_scan_pairs()
        assert info["approx_scan"] == 1, info
        r = check_containment(X, Q, info, pairs, qmeta, f"scan_approx_kernel<{d}> {name}")
        w, a = check_rounding_model(X, Q, info, pairs, qmeta, d, f"scan_approx_kernel<{d}> {name}")
        print(f"scan_approx_kernel<{d}> {name}: {r['pairs']} pairs, {r['unsure']} uncertain, median half-width {r['median_rel_halfwidth']:.2e}; "
              f"max |s - exact| = {w:.2f} u * sum|x h^| (assumed <= {a:.0f} u)")
    ix.close()


@pytest.mark.parametrize("variant", ["lean", "lean_halves", "r5", "dma"])  # (the fused sweep writes no raw pairs: zh_debug_keep_raw keeps it off)
@pytest.mark.parametrize("kind", [0, 1])
def test_leaf_major_half_width_intervals_contain_the_key(za, torch, monkeypatch, kind, variant):
    """sweep128h_kernel (d = 128, zh_set_sweep_mode 6): a row-major fp16 copy under ONE table scale (rows it does not serve: NaNs -> the exact path);
    dma: sweep128h_dma_kernel (ZH_S128H_DMA=1), the same tiles through LDS-DMA"""
    set_s128h_variant(monkeypatch, variant)
    d, n, M, T, k, B = 128, 8000, 600, 6, 10, 48
    rng = np.random.default_rng(77 + kind)
    X = zo.synth_rows(n, d, kind=kind)
    if kind == 0:
        X = special_rows(X, d, rng)
        X[100:160] *= np.float32(2.0 ** -36)   # (huge rows at 2^4 only: ONE scale must serve the table, rows far below it become NaNs)
        X[810:820] = 0.0
    Q = zo.synth_queries(B, d, n, kind=kind)
    if kind == 0:
        Q = special_queries(Q, X)
    else:
        Q[0] = X[0]
        Q[1] = 0.0
    ix, f = build(za, X, d, M, T, "leaf-half")
    m_all = np.abs(X[np.isfinite(X).all(axis=1)]).max()
    _, table_ex = np.frexp(np.float32(m_all))
    for name, m in metrics(za):
        run_batch(torch, ix, Q, k, m)
        info, pairs, qmeta = ix.debug_scan_pairs()
        assert info["approx_scan"] == 3, info
        assert ix.stats()["approx_byte_rows"] == (1 if kind == 1 and variant == "lean" else 0)  # (kind 1: SIFT-style rows -> sweep128b_lean_kernel)
        r = check_containment(X, Q, info, pairs, qmeta, f"sweep128h_kernel kind {kind} {name}")
        w, a = check_rounding_model(X, Q, info, pairs, qmeta, d, f"sweep128h_kernel kind {kind} {name}", table_ex=int(table_ex))
        print(f"sweep128h_kernel kind {kind} {name}: {r['pairs']} pairs, {r['unsure']} uncertain, median half-width {r['median_rel_halfwidth']:.2e}; "
              f"max |acc - exact| = {w:.2f} u * sum|x^ h^| (assumed <= {a:.0f} u), rho = {info['row_rho']:.3e}")
    ix.close()


def test_windows_and_contexts_are_reachable(za, torch):
    """the same read-back through a pipelined context holding a WINDOW of two batches (what bench.py times)"""
    d, n, M, T, k, B = 768, 6000, 500, 15, 100, 32
    X = zo.synth_rows(n, d)
    ix, f = build(za, X, d, M, T, "approx")
    dev = torch.device("cuda", 0)
    Qs = [zo.synth_queries(B, d, n, b0=j * B) for j in range(2)]
    qd = [torch.from_numpy(q).to(dev) for q in Qs]
    res = [dict(ids=torch.empty((B, k), dtype=torch.int64, device=dev), keys=torch.empty((B, k), dtype=torch.int64, device=dev),
                counts=torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(2)]
    ctx = ix.search_context()
    m = za.L2Distance()
    ctx.begin_window([q.data_ptr() for q in qd], B, k, m)
    ctx.finish_window([r["ids"].data_ptr() for r in res], [r["keys"].data_ptr() for r in res], [r["counts"].data_ptr() for r in res])
    ctx.wait()
    torch.cuda.synchronize()
    info, pairs, qmeta = ix.debug_scan_pairs(ctx)
    assert info["approx_scan"] == 2 and info["queries"] == 2 * B
    r = check_containment(X, np.concatenate(Qs), info, pairs, qmeta, "window of two")
    assert r["pairs"] > 2 * B * T * 50
    ctx.close()
    ix.close()
