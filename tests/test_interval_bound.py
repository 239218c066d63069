"""The half-width scans' INTERVALS, restated on the CPU (zebra_amd/csrc/zh_approx.hip: qhalf_kernel, row_half_kernel / row_half128_kernel,
approx_interval, zh_approx_bound): every (row, query) pair's interval must contain the reference's key -- the canonical f32 sums of
Metric::distance (/root/reference/src/distance.rs:19-49,103-114) as the oracle computes them -- whatever order the products are summed in.

This is a property test of the bound's FORMULAS, not of the kernels (those are compared with the oracle bit for bit in tests/test_gpu_approx.py):
the fp16 copies are made as the kernels make them (power-of-two scale, round to nearest even, subnormal halves written as zero, the rounding
error measured), the sum of products is formed in f32 in several orders (in sequence, pairwise, a random order, per 32-element block as an MFMA
step, with every partial sum rounded DOWN instead of to nearest), and the interval arithmetic is carried out in f32 as on the device."""
import numpy as np
import pytest

from oracle import zebra_oracle as zo

F = np.float32
U = 2.0 ** -24


def f16_no_subnormal(v):
    h = v.astype(np.float16)  # round to nearest even
    h[np.abs(h.astype(F)) < F(2.0 ** -14)] = 0
    return h


def scale_of(m):
    ex = 14
    if m > 0:
        ex = int(np.frexp(F(m))[1])
    return F(2.0) ** F(14 - ex), F(2.0) ** F(ex - 14)


def f32_sum(v, order):
    """the f32 sum of v in one of several orders (every partial sum rounded to f32)"""
    v = v.astype(F)
    if order == "seq":
        s = F(0)
        for t in v:
            s = F(s + t)
        return s
    if order == "pairwise":
        w = v.copy()
        while w.size > 1:
            if w.size & 1:
                w = np.concatenate([w, np.zeros(1, F)])
            w = (w[0::2] + w[1::2]).astype(F)
        return w[0]
    if order == "lanes":  # the VALU kernels: a lane's share in sequence (32-lane groups), then the butterfly
        w = np.concatenate([v, np.zeros((-v.size) % 32, F)]).reshape(-1, 32)
        col = np.zeros(32, F)
        for r in w:
            col = (col + r).astype(F)
        return f32_sum(col, "pairwise")
    if order == "blocks32":  # an MFMA step's 32 products in sequence, the steps into four accumulators, as scan_mfma_kernel
        acc = [F(0)] * 4
        for i in range(0, v.size, 32):
            s = acc[(i // 32) & 3]
            for t in v[i:i + 32]:
                s = F(s + t)
            acc[(i // 32) & 3] = s
        return F(F(acc[0] + acc[1]) + F(acc[2] + acc[3]))
    if order == "down":  # every partial sum rounded toward zero: at most 2 u per operation
        s = F(0)
        for t in v:
            e = np.float64(s) + np.float64(t)
            r = F(e)
            if abs(np.float64(r)) > abs(e):
                r = np.nextafter(r, F(0))
            s = r
        return s
    rng = np.random.default_rng(int(order))
    return f32_sum(v[rng.permutation(v.size)], "seq")


def qhalf(q):
    """qhalf_kernel: {halves, 1 / sigma, f32 |q|^2, upper estimate of |q|, upper estimate of |q - h / sigma|}"""
    sigma, inv = scale_of(np.max(np.abs(q)))
    h = f16_no_subnormal((q * sigma).astype(F))
    df = (q - h.astype(F) * inv).astype(F)
    d2 = f32_sum((df * df).astype(F), "pairwise")
    s2 = f32_sum((q * q).astype(F), "pairwise")
    return h, inv, s2, F(np.sqrt(s2)) * F(1.0 + 1e-5), F(np.sqrt(d2)) * F(1.001)


def row_half(x, sigma=None):
    """row_half_kernel (sigma None: the row's own scale) / row_half128_kernel (the table's): {halves, 1 / sigma, rho of this row}"""
    if sigma is None:
        sigma, inv = scale_of(np.max(np.abs(x)))
    else:
        inv = F(1.0) / sigma
    v = (x * sigma).astype(F)
    h = f16_no_subnormal(v)
    df = (v - h.astype(F)).astype(F)
    d2 = f32_sum((df * df).astype(F), "pairwise")
    s2 = f32_sum((v * v).astype(F), "pairwise")
    rho = F(np.sqrt(d2)) * F(1.001) / (F(np.sqrt(s2)) * F(0.9999)) if s2 > 0 else F(0)
    return h, inv, rho


def bound(metric_cos, d, kind):
    c0 = (d + 255) // 256 + 8.0
    ops = (2 * 33.0 + 2.0 + 40.0) if kind == 2 else ((33.0 * (d // 128) + 2.0) if kind else 0.0)
    return F(1.01 * (2.0 * c0 + 80.0 + 2.0 * ops) * U) if metric_cos else F(1.01 * (c0 + 100.0 + ops) * U)


def interval_l2(s, a2, inv_q, b2, nq, dq, Kc, rho, rho_n):
    sh = F(s * inv_q)
    nx = F(F(np.sqrt(a2)) * F(1.0 + 1e-5)) * F(1.0 + 2.0 * rho_n)
    nn = F(nx + nq)
    V = F(F(a2 + b2) - F(2.0) * sh)
    E = F(Kc * nn * nn + F(2.02) * nx * (dq + rho * (nq + dq)) + F(2.2) * rho_n * nx * nx)
    return np.float64(F(V - E)), np.float64(F(V + E))


def interval_cos(s, a2, inv_q, b2, dq, Kc, rho, rho_n):
    sh = F(s * inv_q)
    nx, nq = F(np.sqrt(a2)), F(np.sqrt(b2))
    r = F(1.0) - F(sh / F(nx * nq))
    r = r if r > 0 else F(0)
    dqr = F(dq / nq)
    e = F(Kc + F(1.01) * (dqr + rho * (F(1.0) + dqr) + rho_n))
    return np.float64(F(r - e)), np.float64(F(r + e))


def rows_and_queries(seed, d, n, big=30):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(F)
    Q = rng.standard_normal((n, d)).astype(F)
    Q[: n // 4] = X[: n // 4] + (rng.standard_normal((n // 4, d)) * 1e-3).astype(F)   # near neighbours: the distance is all cancellation
    Q[n // 4] = X[n // 4]                                                                # the row itself
    Q[n // 4 + 1:n // 4 + 4] = np.round(X[n // 4 + 1:n // 4 + 4] * 8) / 8                    # queries fp16 holds exactly beside rows it does not: only rho covers
    X[n // 2:n // 2 + 4] = np.round(X[n // 2:n // 2 + 4] * 20)                           # integer rows
    Q[n // 2:n // 2 + 4] = np.round(Q[n // 2:n // 2 + 4] * 20)
    X[-3] *= F(2.0 ** big); Q[-3] *= F(2.0 ** big)
    X[-2] *= F(2.0 ** -big); Q[-2] *= F(2.0 ** -big)
    X[-1, ::7] *= F(2.0 ** -20)                                                          # elements far below the row's largest: subnormal halves
    return X, Q


# kind 0's bound is for the VALU kernels' own tree (<= 33 roundings to nearest on the longest chain); the matrix-core kinds' for ANY order within an
# MFMA step at 2 u per operation -- they are also given chains far longer than the hardware's (a 768-term sum in sequence, rounded down)
ORDERS = {0: ["lanes", "pairwise"], 1: ["blocks32", "pairwise", "seq", "down", "7", "8"], 2: ["blocks32", "pairwise", "seq", "down", "7", "8"]}


@pytest.mark.parametrize("d,kind", [(768, 0), (768, 1), (384, 1), (128, 2)])
def test_intervals_contain_the_reference_key(d, kind):
    """kind 0: f32 rows x fp16 queries (the VALU kernels); 1: both operands fp16, per-row scales, rho measured (scan_mfma_kernel); 2: one scale for the
    table, |x|^2 from the rounded row as well (sweep128h_kernel)"""
    n = 24
    # (one scale for a whole table: rows 2^10 below its largest element are not served by it -- the kernel scores them exactly; here: magnitudes 2^+-3)
    X, Q = rows_and_queries(1000 + d + kind, d, n, big=3 if kind == 2 else 30)
    table_sigma = scale_of(np.max(np.abs(X[np.isfinite(X).all(axis=1)])))[0] if kind == 2 else None
    usable, prep = [], []
    for x in X:
        if kind == 0:
            prep.append((None, F(1), F(0)))
            usable.append(True)
            continue
        h, inv, rho = row_half(x, table_sigma)
        ok = bool(rho <= F(9.765625e-4)) if kind == 2 else True   # (row_half128_kernel stores NaNs past 2^-10: the exact path)
        prep.append((h, inv, rho))
        usable.append(ok)
    rho_all = max([p[2] for p, ok in zip(prep, usable) if ok] + [F(0)]) if kind else F(0)
    rho_n = rho_all if kind == 2 else F(0)
    checked = 0
    for i, x in enumerate(X):
        if not usable[i]:
            continue
        xh, inv_x, _ = prep[i]
        for q in (Q[i], Q[(i + 5) % n]):
            h, inv_q, b2, nq, dq = qhalf(q)
            ab, a2c, b2c, l2c = zo.distance_sums(x, q)
            for order in ORDERS[kind]:
                if kind == 0:
                    s = f32_sum((x * h.astype(F)).astype(F), order)   # (fma_mix: one rounding per term; the products here are rounded too: more error, not less)
                    a2 = f32_sum((x * x).astype(F), "pairwise")
                else:
                    s = F(f32_sum((xh.astype(F) * h.astype(F)).astype(F), order) * inv_x)
                    a2 = f32_sum((x * x).astype(F), "pairwise") if kind == 1 else F(f32_sum((xh.astype(F) ** 2).astype(F), order) * inv_x * inv_x)
                if not (np.isfinite(s) and np.isfinite(a2) and np.isfinite(b2)):
                    continue  # (the device hands such a pair the interval (-inf, +inf))
                lo, hi = interval_l2(s, a2, inv_q, b2, nq, dq, bound(False, d, kind), rho_all, rho_n)
                if np.isfinite(lo) and np.isfinite(hi) and F(a2 + b2) < F(1e37):
                    assert lo <= np.float64(l2c) <= hi, ("l2", d, kind, i, order, lo, float(l2c), hi)
                    checked += 1
                if a2 > 0 and b2 > 0:
                    key = np.array([zo.distance(zo.COSINE, zo.CORRECTED, x, q)], np.uint64).view(np.float64)[0]
                    lo, hi = interval_cos(s, a2, inv_q, b2, dq, bound(True, d, kind), rho_all, rho_n)
                    if np.isfinite(lo) and np.isfinite(hi) and F(np.sqrt(a2)) > F(1e-12) and F(np.sqrt(b2)) > F(1e-12):
                        assert lo <= key <= hi, ("cos", d, kind, i, order, lo, key, hi)
                        checked += 1
    assert checked > 150
