"""The `port-fast` CPU baseline (oracle/zebra_cpu_fast.cpp: free summation order, nth_element, no re-score) computes
the same thing as the bit-exact oracle up to f32 summation order: keys within 1e-5 relative, the same neighbours except
where two candidates' keys are closer than that."""
import numpy as np
import pytest

from oracle import zebra_cpu_fast as zf
from oracle import zebra_oracle as zo


@pytest.mark.parametrize("n,d,M,T,k,kind", [(6000, 96, 128, 6, 10, 0), (5000, 384, 5, 4, 10, 0), (8192, 128, 512, 5, 10, 1),
                                            (4000, 768, 300, 3, 100, 2)])
def test_fast_port_matches_the_oracle_within_summation_order(n, d, M, T, k, kind):
    X = zo.synth_rows(n, d, kind=kind)
    f = zo.Forest.build(X, M, T)
    ff = zf.FastForest(X, f.arrays())
    Q = zo.synth_queries(24, d, n, kind=kind)
    for om, fm, mode in ((zo.L2SQ, zf.L2SQ, 0), (zo.L2, zf.L2, 0), (zo.COSINE, zf.COSINE, zo.PARITY), (zo.COSINE, zf.COSINE, zo.CORRECTED)):
        oi, ok, oc = f.search_batch(Q, k, om, mode)
        fi, fk, fc, rows = ff.search_batch(Q, k, fm, mode, nthreads=2)
        assert (oc == fc).all() and rows > 0
        for b in range(Q.shape[0]):
            c = int(oc[b])
            a, g = zo.key_to_float(ok[b, :c]), zo.key_to_float(fk[b, :c])
            assert np.allclose(a, g, rtol=1e-5, atol=1e-6)
            if kind == 1 and om != zo.COSINE:  # integer-valued rows: every summation order is exact
                assert (oi[b, :c] == fi[b, :c]).all() and (ok[b, :c] == fk[b, :c]).all()
            else:
                assert len(set(oi[b, :c].tolist()) & set(fi[b, :c].tolist())) >= c - 2
    assert zf.isa() in ("avx2", "avx512")
