"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
CPU: the oracle still produces them; GPU: the HIP path reproduces them bit for bit."""
import glob
import os

import numpy as np
import pytest

from oracle import zebra_oracle as zo

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
MET = [("l2sq", zo.L2SQ, 0), ("l2", zo.L2, 0), ("cos_parity", zo.COSINE, zo.PARITY), ("cos_corrected", zo.COSINE, zo.CORRECTED)]


def test_fixtures_exist():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    n, d, M, T, k, B, kind, seed = g["params"].tolist()
    X, Q = g["X"], g["Q"]
    assert (zo.synth_rows(n, d, kind=kind).view(np.uint32) == X.view(np.uint32)).all()
    f = zo.Forest.build(X, M, T, seed=seed)
    assert zo.canonical_forest(f.arrays(), d) == g["forest_hash"].tolist()
    for name, om, omode in MET:
        ids, keys, counts = f.search_batch(Q, k, om, omode)
        assert (counts == g[f"{name}_counts"]).all() and (ids == g[f"{name}_ids"]).all() and (keys == g[f"{name}_keys"]).all()
        assert (zo.distance_batch(om, omode, X[:64], Q[0]) == g[f"{name}_rowkeys"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_hip_reproduces_golden(path):
    import zebra_amd as za
    g = np.load(path)
    n, d, M, T, k, B, kind, seed = g["params"].tolist()
    X, Q = g["X"], g["Q"]
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=seed)
    ix.add(X)
    assert zo.canonical_forest(ix.get_forest(), d) == g["forest_hash"].tolist()
    mets = {"l2sq": za.L2SquaredDistance(), "l2": za.L2Distance(), "cos_parity": za.CosineDistance(True),
            "cos_corrected": za.CosineDistance(False)}
    for name, m in mets.items():
        ids, keys, counts = ix.search_batch(Q, k, m)
        gc = g[f"{name}_counts"]
        assert (counts == gc).all()
        for b in range(B):
            assert (ids[b, :gc[b]] == g[f"{name}_ids"][b, :gc[b]]).all() and (keys[b, :gc[b]] == g[f"{name}_keys"][b, :gc[b]]).all()
        assert (m.distance_batch(X[:64], Q[0]) == g[f"{name}_rowkeys"]).all()
