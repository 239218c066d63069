"""The table scan with HALF-WIDTH queries (zh_set_sweep_mode 4 / the default wherever it applies; zebra_amd/csrc/zh_approx.hip):
the scan reads fp16 copies of the queries and gives every (row, query) pair an INTERVAL that contains the reference's key; the
intervals pick the candidates, and only the rows they cannot rule out are scored with the reference's arithmetic
(Metric::distance, /root/reference/src/distance.rs:19-49,103-114; the leaf's `take` nearest, lsh.rs:317-323; the final top_k,
lsh.rs:557-564).  Ids, keys and counts must equal the oracle's bit for bit -- including where the intervals cannot decide:
ties at the cut, rows within the bound of the threshold, norms that overflow, zero vectors, NaNs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)


def set_s128h_variant(monkeypatch, variant):
    """fused: sweep128h_lean_kernel<CH, KINDA> + sweep128h_boundary_kernel<KINDA> with intervals, bounds and the queries' lists inside the sweep
    (round 6; chosen by the library for long leaves, forced here: ZH_S128H_FUSED=1; top_k <= 64); lean: the same kernels writing raw pairs for
    select_tau_kernel (ZH_S128H_FUSED=0); r5: sweep128h_kernel (ZH_S128H_KERNEL=r5); dma: sweep128h_dma_kernel (ZH_S128H_DMA=1).
    fused / lean read the 128-byte copy of a table of integers 0 .. 255 (sweep128b_lean_kernel, sweep128h_boundary_kernel<.., true>) when every
    stored row qualifies; fused_halves / lean_halves keep them on the copy of halves whatever the rows are (ZH_S128H_BYTES=0)"""
    env = {"fused": {"ZH_S128H_FUSED": "1"}, "lean": {"ZH_S128H_FUSED": "0"}, "r5": {"ZH_S128H_KERNEL": "r5"}, "dma": {"ZH_S128H_DMA": "1"},
           "fused_halves": {"ZH_S128H_FUSED": "1", "ZH_S128H_BYTES": "0"}, "lean_halves": {"ZH_S128H_FUSED": "0", "ZH_S128H_BYTES": "0"}}[variant]
    for var in ("ZH_S128H_FUSED", "ZH_S128H_KERNEL", "ZH_S128H_DMA", "ZH_S128H_BYTES"):
        if var in env:
            monkeypatch.setenv(var, env[var])
        else:
            monkeypatch.delenv(var, raising=False)


@pytest.fixture(scope="module")
def za():
    import zebra_amd
    return zebra_amd


def all_metrics(za):
    return [(za.L2SquaredDistance(), zo.L2SQ, 0), (za.L2Distance(), zo.L2, 0),
            (za.CosineDistance(parity=True), zo.COSINE, zo.PARITY), (za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)]


def check(ix, f, Q, k, m, om, omode, what=""):
    ids, keys, counts = ix.search_batch(Q, k, m)
    oi, ok, oc = f.search_batch(Q, k, om, omode)
    assert (counts == oc).all(), (what, om, omode)
    for b in range(Q.shape[0]):
        c = int(oc[b])
        assert (ids[b, :c] == oi[b, :c]).all() and (keys[b, :c] == ok[b, :c]).all(), (what, om, omode, b)
    return ix.stats()


CASES = [
    # n, d, M, T, k, batch, kind, approx expected
    (20000, 384, 256, 15, 10, 64, 0, True),    # one leaf per tree; four 16-lane groups, three loads per pair
    (12000, 768, 512, 8, 100, 48, 0, True),    # the cfg3 / cfg4 shape in small: two 32-lane groups
    (9000, 128, 300, 10, 10, 96, 1, True),     # SIFT-style integer rows: the fp16 copy is exact
    (9000, 256, 100, 64, 10, 12, 0, True),     # 64 trees: 4 rows per wave
    (6000, 512, 200, 5, 10, 33, 0, True),
    (5000, 1024, 128, 6, 20, 17, 0, True),
    (3001, 512, 3002, 5, 10, 700, 0, True),    # ONE leaf per tree, visited by every query: more pairs than a wave's list holds
    (7000, 768, 24, 6, 10, 9, 0, True),        # leaves ~ top_k: backup visits that take fewer than top_k rows (the exact path)
    (4000, 1536, 64, 3, 10, 7, 0, False),      # a dimension the half-width scan does not cover: the f32 scan
    (5000, 384, 5, 15, 10, 8, 0, False),       # reference defaults: thousands of visits per pair -- not this path's regime
]


def scan_code(mode, d, T=8):
    """zh_stats_t::approx_scan of a half-width batch: 2 where the matrix-core kernel runs (scan_mfma_kernel: up to 16 trees; mode 5 keeps the
    VALU kernel)"""
    return 2 if mode == "approx" and d in (256, 384, 512, 768, 1024) and T <= 16 else 1


@pytest.mark.parametrize("mode", ["approx", "approx-valu"])
@pytest.mark.parametrize("n,d,M,T,k,B,kind,expect", CASES)
def test_half_width_scan_equals_oracle(za, n, d, M, T, k, B, kind, expect, mode):
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode(mode)
    ix.set_hash_mode("dense")
    for m, om, omode in all_metrics(za):
        st = check(ix, f, Q, k, m, om, omode)
        assert st["approx_scan"] == (scan_code(mode, d, T) if expect else 0), (om, omode, st)
        assert st["approx_fallbacks_accum"] == 0, st
        if expect:
            assert st["approx_survivors"] >= min(k, 1) and st["approx_list_entries"] > 0
    ix.close()


@pytest.mark.parametrize("n,d,M,T,k,B,kind,expect", [c for c in CASES if c[1] in (256, 384, 512, 768, 1024) and c[7]])
def test_matrix_core_scan_on_f32_rows_equals_oracle(za, monkeypatch, n, d, M, T, k, B, kind, expect):
    """no room for the fp16 copy of the table (64M x 768 on one GPU; here: ZH_ROW_HALF_META_ONLY=1): the same matrix-core scan, the wave converting
    its 16 f32 rows itself (scan_mfma_kernel<d, true>) -- per-row scales and norms are all the index keeps (8 bytes per row)"""
    monkeypatch.setenv("ZH_ROW_HALF_META_ONLY", "1")
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode("approx")
    ix.set_hash_mode("dense")
    for m, om, omode in all_metrics(za):
        st = check(ix, f, Q, k, m, om, omode)
        assert st["approx_scan"] == scan_code("approx", d, T), (om, omode, st)
        assert st["approx_fallbacks_accum"] == 0, st
        if st["approx_scan"] == 2:
            assert 0 < st["row_copy_bytes"] < n * d, st  # (no copy: 2 d bytes per row would be more)
    # rows appended afterwards: scales and norms follow
    X2 = zo.synth_rows(n + 500, d, kind=kind)[n:] * np.float32(8.0)
    ix.append(X2)
    f2 = zo.Forest.build(np.concatenate([X, X2]), M, T)
    ix.set_forest(f2.arrays())
    check(ix, f2, Q, k, za.L2Distance(), zo.L2, 0, "appended")
    ix.close()


@pytest.mark.parametrize("mode", ["approx", "approx-valu"])
def test_exact_visits_are_exercised(za, mode):
    """leaves shorter than top_k send the walk to backup subtrees with n < top_k (lsh.rs:340-345): those visits must hand over
    exactly their `take` nearest rows -- the exact path of the half-width scan"""
    n, d, M, T, k, B = 6000, 768, 120, 8, 32, 24
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode(mode)
    ix.set_hash_mode("dense")
    seen = 0
    for m, om, omode in all_metrics(za):
        st = check(ix, f, Q, k, m, om, omode)
        assert st["approx_scan"] == scan_code(mode, d) and st["approx_fallbacks_accum"] == 0
        seen += st["approx_exact_visits"]
    assert seen > 0
    ix.close()


def _adversarial_rows(n, d, rng):
    """rows built to sit on the cut: exact duplicates, rows one ulp apart, integer-valued rows (ties in every metric), zero rows,
    rows with huge / tiny norms, a NaN row and an infinite row"""
    X = zo.synth_rows(n, d)
    base = X[:64].copy()
    X[100:164] = base                                   # duplicates of rows 0..63: equal keys, ids decide
    X[200:264] = np.nextafter(base, np.float32(np.inf))  # one ulp away in every coordinate
    X[300:364] = np.nextafter(base, np.float32(-np.inf))
    X[400:700] = np.round(X[400:700] * 2.0)             # small integers: many exact ties
    X[700:710] = 0.0                                    # zero rows (simsimd's zero-norm cases)
    X[710:720] *= np.float32(1e18)                      # |x|^2 overflows f32
    X[720:730] *= np.float32(1e-20)                     # squares underflow
    X[730, 3] = np.nan
    X[731, 5] = np.inf
    X[732] = -X[0]                                      # the antipode: negative cosine (parity order)
    return X


@pytest.mark.parametrize("mode", ["approx", "approx-valu"])
def test_adversarial_rows_and_queries(za, mode):
    rng = np.random.default_rng(7)
    n, d, M, T, k, B = 4000, 768, 1024, 6, 50, 40
    X = _adversarial_rows(n, d, rng)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X)
    ix.set_forest(f.arrays())        # (the oracle's forest: a NaN sample row makes a NaN plane, whose payload bits need not agree)
    ix.set_sweep_mode(mode)          # (the matrix-core kernel rounds the ROWS too: the huge / tiny / non-finite rows get scales out of range)
    Q = zo.synth_queries(B, d, n)
    Q[0] = X[0]                      # a stored row itself: distance 0, its duplicates tie
    Q[1] = X[410]                    # an integer row
    Q[2] = 0.0                       # the zero query
    Q[3] = X[0] * np.float32(1e18)   # overflowing query norm: every L2 key is +inf, ids decide
    Q[4] = X[5] * np.float32(1e-20)
    Q[6] = -X[1]                     # every near row has a negative cosine
    Q[7] = np.round(Q[7] * 3.0)
    # (no NaN / inf QUERY: every key would be a NaN, whose sign bit differs between the host's and the GPU's arithmetic)
    for m, om, omode in all_metrics(za):
        st = check(ix, f, Q, k, m, om, omode, "adversarial")
        assert st["approx_scan"] == scan_code(mode, d)
    ix.close()


@pytest.mark.parametrize("mode", ["approx", "approx-valu"])
def test_lists_that_run_over_are_redone_on_the_device(za, monkeypatch, mode):
    """a candidate list, the survivors, the table of exact visits or their key scratch running over raises a flag on the device
    and the f32 scan + select + final enqueued behind redo the batch in stream order: same results, counted in the stats"""
    n, d, M, T, k, B = 8000, 768, 120, 8, 32, 24
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode(mode)
    code = scan_code(mode, d, T)
    ix.set_hash_mode("dense")
    m, om, omode = za.L2Distance(), zo.L2, 0
    st = check(ix, f, Q, k, m, om, omode)
    assert st["approx_scan"] == code and st["approx_fallbacks_accum"] == 0 and st["approx_exact_visits"] > 0
    seen = 0
    for caps, bit in (("64,0,0", 1), ("0,1,0", 4), ("0,0,8", 8)):
        monkeypatch.setenv("ZH_APX_CAPS", caps)
        ix.stats(reset=True)
        st = check(ix, f, Q, k, m, om, omode, caps)
        assert st["approx_scan"] == code and st["approx_fallbacks_accum"] == 1 and (st["approx_last_overflow"] & bit), (caps, st)
        seen += 1
    # the 8192-slot lists an index gets after an overflow (final_interval_kernel's larger variant), all four keys
    monkeypatch.setenv("ZH_APX_CAPS", "8192,0,0")
    for mm, omm, omo in all_metrics(za):
        ix.stats(reset=True)
        st = check(ix, f, Q, k, mm, omm, omo, "8192 slots")
        assert st["approx_scan"] == code and st["approx_fallbacks_accum"] == 0
    monkeypatch.delenv("ZH_APX_CAPS")
    ix.stats(reset=True)
    st = check(ix, f, Q, k, m, om, omode)  # mode 4 keeps trying after strikes
    assert st["approx_scan"] == code and st["approx_fallbacks_accum"] == 0
    assert seen == 3
    ix.close()


LEAF_CASES = [
    # n, M, T, k, batch, kind
    (20000, 2000, 6, 10, 64, 1),     # SIFT-style integer rows: the fp16 copy is exact
    (9000, 300, 10, 10, 96, 1),
    (12000, 1500, 5, 100, 40, 0),    # float rows: rounded under the table's common scale
    (3001, 3002, 5, 10, 300, 0),     # ONE leaf per tree, visited by every query: many groups on the same rows
    (7000, 24, 6, 10, 9, 0),         # leaves ~ top_k: tiles that span many groups, backup visits (the exact path)
    (5000, 700, 16, 100, 33, 2),
]


@pytest.mark.parametrize("variant", ["fused", "lean", "r5", "dma", "fused_halves", "lean_halves"])
@pytest.mark.parametrize("n,M,T,k,B,kind", LEAF_CASES)
def test_leaf_major_half_width_sweep_equals_oracle(za, monkeypatch, n, M, T, k, B, kind, variant):
    """d = 128, leaf by leaf at half width on the matrix cores (sweep128h_kernel, zh_set_sweep_mode 6): an fp16 copy of the rows under one
    scale, the queries of up to four groups as the A operand, 16 stored rows as B; same intervals, same stages behind it.  dma: the variant
    whose row tiles travel straight into LDS (sweep128h_dma_kernel, ZH_S128H_DMA=1: two tiles in flight per wave, explicit waits)"""
    d = 128
    set_s128h_variant(monkeypatch, variant)
    X = zo.synth_rows(n, d, kind=kind)
    Q = zo.synth_queries(B, d, n, kind=kind)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode("leaf-half")
    ix.set_hash_mode("dense")
    for m, om, omode in all_metrics(za):
        st = check(ix, f, Q, k, m, om, omode)
        assert st["approx_scan"] == 3 and st["table_scan"] == 0, (om, omode, st)
        assert st["approx_fallbacks_accum"] == 0, st
        assert st["approx_fused"] == (1 if variant.startswith("fused") and k <= 64 else 0), st
        assert st["approx_byte_rows"] == (1 if kind == 1 and variant in ("fused", "lean") else 0), st
    ix.close()


@pytest.mark.parametrize("variant", ["fused", "lean"])
def test_byte_rows_copy_follows_the_stored_rows(za, monkeypatch, variant):
    """a table of integers 0 .. 255 is swept from an exact copy of 128 bytes per row (row_byte128_kernel): appended rows of bytes join it; ONE
    appended element that is anything else (a fraction, a negative, 256, NaN) and the copy is re-made in halves; ZH_S128H_BYTES=0 and the round-5
    kernels get the copy of halves from the same rows and a later batch the bytes again; replaced rows (clear + refill) are checked afresh"""
    set_s128h_variant(monkeypatch, variant)
    n, d, M, T, k, B = 9000, 128, 1200, 5, 10, 48
    X = zo.synth_rows(2 * n, d, kind=1)
    X[5] = 0.0; X[6] = 255.0; X[7, ::2] = 0.0; X[8] = X[9]
    assert X.min() >= 0 and X.max() <= 255 and np.all(X == np.round(X))
    Q = zo.synth_queries(B, d, n, kind=1)
    Q[0] = X[6]; Q[1] = 0.0; Q[2] = X[8]; Q[3] = Q[3] + np.float32(0.37)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X[:n])
    f = zo.Forest.build(X[:n], M, T)
    ix.set_forest(f.arrays())
    ix.set_sweep_mode("leaf-half")
    fused = 1 if variant == "fused" else 0
    for mm, omm, omo in all_metrics(za):
        st = check(ix, f, Q, k, mm, omm, omo, "bytes")
        assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 1 and st["approx_fused"] == fused, st
        assert omm == zo.COSINE or st["approx_fallbacks_accum"] == 0, st  # (the zero query's cosine intervals are all "nothing certain": its list may run over)
    bytes_copy = st["row_copy_bytes"]
    assert bytes_copy < n * 256, st
    m, om = za.L2Distance(), zo.L2
    monkeypatch.setenv("ZH_S128H_BYTES", "0")
    st = check(ix, f, Q, k, m, om, 0, "halves on request")
    assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 0 and st["row_copy_bytes"] >= n * 256, st
    monkeypatch.delenv("ZH_S128H_BYTES")
    monkeypatch.setenv("ZH_S128H_KERNEL", "r5")
    monkeypatch.delenv("ZH_S128H_FUSED")
    st = check(ix, f, Q, k, m, om, 0, "the round-5 kernel reads halves")
    assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 0, st
    set_s128h_variant(monkeypatch, variant)
    st = check(ix, f, Q, k, m, om, 0, "bytes again")
    assert st["approx_byte_rows"] == 1, st
    monkeypatch.setenv("ZH_S128H_PRETEST", "0")   # the fused epilogue without its pre-test (every chunk takes fused_slot): the same answers
    for mm, omm, omo in all_metrics(za):
        st = check(ix, f, Q, k, mm, omm, omo, "bytes, no pre-test")
        assert st["approx_byte_rows"] == 1, st
    monkeypatch.delenv("ZH_S128H_PRETEST")
    ix.append(X[n:n + n // 2])          # more rows of bytes: added to the copy
    f2 = zo.Forest.build(X[:n + n // 2], M, T)
    ix.set_forest(f2.arrays())
    st = check(ix, f2, Q, k, m, om, 0, "appended bytes")
    assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 1, st
    for bad in (np.float32(0.5), np.float32(-1.0), np.float32(256.0), np.float32(np.nan)):
        Y = X[n + n // 2:].copy()
        Y[17, 101] = bad
        iy = za.LSHIndex(d, za.LSHIndexOptions(M, T))
        iy.append(X[:n + n // 2])
        iy.set_forest(f2.arrays())
        iy.set_sweep_mode("leaf-half")
        st = check(iy, f2, Q, k, m, om, 0, "bytes before the append")
        assert st["approx_byte_rows"] == 1, st
        iy.append(Y)
        Z = np.concatenate([X[:n + n // 2], Y])
        f3 = zo.Forest.build(Z, M, T)
        iy.set_forest(f3.arrays())
        st = check(iy, f3, Q, k, m, om, 0, "an appended element %r" % bad)
        assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 0, (bad, st)
        iy.close()
    ix.clear()                          # replaced rows are checked afresh
    W = X[:n] * np.float32(0.5)
    ix.append(W)
    fw = zo.Forest.build(W, M, T)
    ix.set_forest(fw.arrays())
    st = check(ix, fw, Q * np.float32(0.5), k, m, om, 0, "replaced by rows of halves")
    assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 0, st
    ix.clear()
    ix.append(X[:n])
    ix.set_forest(f.arrays())
    st = check(ix, f, Q, k, m, om, 0, "replaced by rows of bytes")
    assert st["approx_scan"] == 3 and st["approx_byte_rows"] == 1, st
    ix.close()


@pytest.mark.parametrize("variant", ["fused", "lean"])
def test_leaf_major_half_width_adversarial_rows_appends_and_overflow(za, monkeypatch, variant):
    """rows that do not survive the table's common scale (tiny, huge, non-finite: stored as NaNs -> the exact path), ties, appended rows on a
    larger scale (the copy is re-made), and lists that run over (redone by the f32 leaf-major sweep on the device)"""
    set_s128h_variant(monkeypatch, variant)
    rng = np.random.default_rng(11)
    n, d, M, T, k, B = 6000, 128, 800, 6, 20, 48
    X = zo.synth_rows(2 * n, d, kind=0)
    X[100:164] = X[:64]
    X[200:264] = np.nextafter(X[:64], np.float32(np.inf))
    X[400:700] = np.round(X[400:700] * 2.0)
    X[700:710] = 0.0
    X[710:720] *= np.float32(1e-12)
    X[730, 3] = np.nan
    X[731, 5] = np.inf
    X[n:] *= np.float32(64.0)           # the appended half: a larger scale
    m, om, omode = za.L2Distance(), zo.L2, 0
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.append(X[:n])
    f = zo.Forest.build(X[:n], M, T)
    ix.set_forest(f.arrays())
    ix.set_sweep_mode("leaf-half")
    Q = zo.synth_queries(B, d, n)
    Q[0] = X[0]; Q[1] = X[410]; Q[2] = 0.0
    for mm, omm, omo in all_metrics(za):
        st = check(ix, f, Q, k, mm, omm, omo, "adversarial")
        assert st["approx_scan"] == 3
    ix.append(X[n:])
    f2 = zo.Forest.build(X, M, T)
    ix.set_forest(f2.arrays())
    Q2 = np.concatenate([Q[:B // 2], zo.synth_queries(B // 2, d, n) * np.float32(64.0)])
    st = check(ix, f2, Q2, k, m, om, omode, "appended")
    assert st["approx_scan"] == 3
    monkeypatch.setenv("ZH_APX_CAPS", "64,0,0")
    ix.stats(reset=True)
    st = check(ix, f2, Q2, k, m, om, omode, "lists run over")
    assert st["approx_scan"] == 3 and st["approx_fallbacks_accum"] == 1
    monkeypatch.delenv("ZH_APX_CAPS")
    ix.close()


def test_fused_sweep_is_the_librarys_choice_for_long_leaves_and_steps_back_after_an_overflow(za, monkeypatch):
    """leaves of thousands of rows: the library picks the FUSED half-width sweep by itself (no switch set); lists that run over send the index to the
    unfused sweep (select_tau / select_emit: exact per-visit bounds, shorter lists) instead of the f32 sweep"""
    for var in ("ZH_S128H_FUSED", "ZH_S128H_KERNEL", "ZH_S128H_DMA"):
        monkeypatch.delenv(var, raising=False)
    n, d, M, T, k, B = 24000, 128, 5000, 4, 10, 40
    X = zo.synth_rows(n, d, kind=1)
    Q = zo.synth_queries(B, d, n, kind=1)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    ix.set_sweep_mode("leaf-half")
    ix.set_hash_mode("dense")
    m, om = za.L2Distance(), zo.L2
    st = check(ix, f, Q, k, m, om, 0, "library's choice")
    assert st["approx_scan"] == 3 and st["approx_fused"] == 1 and st["approx_fallbacks_accum"] == 0, st
    monkeypatch.setenv("ZH_APX_CAPS", "16,0,0")
    st = check(ix, f, Q, k, m, om, 0, "lists run over")          # redone by the f32 sweep on the device: still the oracle's answer
    assert st["approx_fused"] == 1 and st["approx_fallbacks_accum"] == 1, st
    monkeypatch.delenv("ZH_APX_CAPS")
    st = check(ix, f, Q, k, m, om, 0, "after the overflow")
    assert st["approx_scan"] == 3 and st["approx_fused"] == 0 and st["approx_fallbacks_accum"] == 1, st
    ix.close()


def test_matrix_core_scan_follows_the_stored_rows(za):
    """the matrix-core scan keeps an fp16 copy of the stored rows, a scale per row and the largest relative rounding error of any row
    (row_half_kernel): all three must follow
    rows that are appended (only the new rows are measured) and rows that are replaced (clear + refill to the same count)"""
    n, d, M, T, k, B = 6000, 512, 256, 6, 10, 32
    X = zo.synth_rows(2 * n, d)
    X[n:] *= np.float32(2.0 ** 12)  # the appended half on another scale
    m, om, omode = za.L2Distance(), zo.L2, 0
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X[:n])
    ix.set_sweep_mode("approx")
    ix.set_hash_mode("dense")
    Q = zo.synth_queries(B, d, n)
    assert ix.stats()["row_copy_bytes"] == 0
    st = check(ix, zo.Forest.build(X[:n], M, T), Q, k, m, om, omode, "first half")
    assert st["approx_scan"] == 2 and st["row_copy_bytes"] >= n * (2 * d + 8)
    ix.append(X[n:])
    ix.build()
    Q2 = np.concatenate([Q[:B // 2], zo.synth_queries(B // 2, d, n) * np.float32(2.0 ** 12)])
    st = check(ix, zo.Forest.build(X, M, T), Q2, k, m, om, omode, "both halves")
    assert st["approx_scan"] == 2
    assert ix.stats()["row_copy_bytes"] >= 2 * n * (2 * d + 8)
    ix.clear()
    assert ix.stats()["row_copy_bytes"] == 0
    Y = zo.synth_rows(2 * n, d, seed=5) * np.float32(2.0 ** -30)
    ix.add(Y)
    st = check(ix, zo.Forest.build(Y, M, T), zo.synth_queries(B, d, 2 * n, seed_rows=5) * np.float32(2.0 ** -30), k, m, om, omode, "refilled")
    assert st["approx_scan"] == 2
    ix.close()


def test_windows_through_contexts_and_default_mode(za):
    """the pipelined window calls (results consumed in stream order) and the default sweep mode: a forest with long leaves that
    has served a few batches takes the half-width scan by itself"""
    import torch
    n, d, M, T, k, B = 30000, 768, 2048, 15, 10, 64
    X = zo.synth_rows(n, d)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    m, om, omode = za.CosineDistance(parity=True), zo.COSINE, zo.PARITY
    Qs = [zo.synth_queries(B, d, n, b0=i * B) for i in range(3)]
    for _ in range(5):
        ix.search_batch(Qs[0], k, m)
    st = check(ix, f, Qs[0], k, m, om, omode, "default mode")
    assert st["table_scan"] == 1 and st["approx_scan"] == 2, st  # (15 trees: the matrix-core kernel)
    dev = torch.device("cuda", 0)
    dq = [torch.from_numpy(q).to(dev) for q in Qs]
    out = [dict(ids=torch.empty((B, k), dtype=torch.int64, device=dev), keys=torch.empty((B, k), dtype=torch.int64, device=dev),
                counts=torch.empty(B, dtype=torch.int32, device=dev)) for _ in Qs]
    ix.set_sweep_mode("approx")  # (which sweep the cost model picks for this small window is not the point here)
    ctx = ix.search_context()
    s = torch.cuda.Stream(device=dev)
    ctx.begin_window([q.data_ptr() for q in dq], B, k, m, s.cuda_stream)
    ctx.finish_window([o["ids"].data_ptr() for o in out], [o["keys"].data_ptr() for o in out], [o["counts"].data_ptr() for o in out],
                      ix.sweep_stream())
    s.synchronize()  # stream order is enough: nothing is redone from the host
    host = [(o["ids"].cpu().numpy().view(np.uint64), o["keys"].cpu().numpy().view(np.uint64), o["counts"].cpu().numpy().view(np.uint32)) for o in out]
    ctx.wait()
    assert ix.stats()["approx_scan"] == 2 and ix.stats()["window_batches"] == 3
    for q, (ids, keys, counts) in zip(Qs, host):
        oi, ok, oc = f.search_batch(q, k, om, omode)
        assert (counts == oc).all() and (ids == oi).all() and (keys == ok).all()
    ctx.close()
    ix.close()


def test_fused_sweep_through_pipelined_contexts_windows_and_a_shard_group(za, monkeypatch):
    """what bench.py's cfg5 loop does, in small: two contexts in flight, each a WINDOW of several batches, the d = 128 sweep FUSED (lists, bounds and
    counters are per context: nothing of one window may leak into the other), directly and through a one-rank shard group (the exchange's merge behind it)"""
    import torch
    monkeypatch.setenv("ZH_S128H_FUSED", "1")
    n, d, M, T, k, B, W = 16000, 128, 1500, 6, 10, 40, 3
    X = zo.synth_rows(n, d, kind=1)
    f = zo.Forest.build(X, M, T)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), id_base=7)
    ix.add(X)
    ix.set_sweep_mode("leaf-half")
    ix.set_hash_mode("dense")
    g = za.ShardGroup(ix, za.shard_unique_id(), 1, 0)
    dev = torch.device("cuda", 0)
    m, om = za.L2Distance(), zo.L2
    for make in (lambda: ix.search_context(), lambda: g.search_context()):
        ctxs = [make(), make()]
        sharded = isinstance(ctxs[0], za.ShardContext)
        rounds = 4
        Qh = [[zo.synth_queries(B, d, n, b0=(r * W + j) * B, kind=1) for j in range(W)] for r in range(rounds)]
        Qs = [[torch.from_numpy(q).to(dev) for q in row] for row in Qh]
        outs = [[[torch.zeros((B, k), dtype=torch.int64, device=dev), torch.zeros((B, k), dtype=torch.int64, device=dev),
                  torch.zeros(B, dtype=torch.int32, device=dev)] for _ in range(W)] for _ in range(rounds)]

        def begin(c, r):
            args = ([q.data_ptr() for q in Qs[r]], B, k, m)
            c.begin_window(*args) if sharded else c.begin_window(*args, None)

        def finish(c, r):
            ptrs = [[o[i].data_ptr() for o in outs[r]] for i in range(3)]
            c.finish_window(*ptrs) if sharded else c.finish_window(*ptrs, None)

        begin(ctxs[0], 0)
        finish(ctxs[0], 0)
        for r in range(1, rounds):        # window r is begun and finished while window r - 1 is still on the GPU
            begin(ctxs[r & 1], r)
            finish(ctxs[r & 1], r)
            ctxs[(r - 1) & 1].wait()
        ctxs[(rounds - 1) & 1].wait()
        torch.cuda.synchronize()
        st = ix.stats()
        assert st["approx_scan"] == 3 and st["approx_fused"] == 1 and st["approx_fallbacks_accum"] == 0, st
        for r in range(rounds):
            for j in range(W):
                oi, ok, oc = f.search_batch(Qh[r][j], k, om, 0)
                assert (outs[r][j][2].cpu().numpy().view(np.uint32) == oc).all(), (sharded, r, j)
                assert (outs[r][j][0].cpu().numpy().view(np.uint64) == oi + np.uint64(7)).all(), (sharded, r, j)
                assert (outs[r][j][1].cpu().numpy().view(np.uint64) == ok).all(), (sharded, r, j)
        for c in ctxs:
            c.close()
    g.close()
    ix.close()
