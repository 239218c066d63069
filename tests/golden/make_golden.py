#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference (Rust, /root/reference) has no tests, fixtures or golden vectors and cannot be run here
(parity unpinned), so these fixtures are produced by the build's own CPU oracle
(oracle/zebra_oracle.c) at small sizes.  They pin the oracle against accidental change and give the HIP path
a second, file-based target besides the live oracle.  Inputs are stored, not only their seeds, so a fixture
stays meaningful even if the synthetic generator changes.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import zebra_oracle as zo  # noqa: E402

CASES = {
    # name: n, d, M, T, k, B, kind, seed
    "defaults_d16": (600, 16, 5, 15, 10, 6, 0, 1),     # reference defaults max_node_size 5 / num_trees 15
    "oneleaf_d64": (2000, 64, 128, 6, 10, 8, 0, 2),
    "k100_d128_sift": (1500, 128, 256, 4, 100, 5, 1, 3),
    "odd_d30": (700, 30, 20, 3, 7, 5, 0, 4),
}


def main():
    for name, (n, d, M, T, k, B, kind, seed) in CASES.items():
        X = zo.synth_rows(n, d, kind=kind)
        Q = zo.synth_queries(B, d, n, kind=kind)
        f = zo.Forest.build(X, M, T, seed=seed)
        out = dict(X=X, Q=Q, params=np.array([n, d, M, T, k, B, kind, seed], np.int64),
                   forest_hash=np.array(zo.canonical_forest(f.arrays(), d)))
        for mname, om, omode in (("l2sq", zo.L2SQ, 0), ("l2", zo.L2, 0), ("cos_parity", zo.COSINE, zo.PARITY),
                                 ("cos_corrected", zo.COSINE, zo.CORRECTED)):
            ids, keys, counts = f.search_batch(Q, k, om, omode)
            out[f"{mname}_ids"], out[f"{mname}_keys"], out[f"{mname}_counts"] = ids, keys, counts
            out[f"{mname}_rowkeys"] = zo.distance_batch(om, omode, X[:64], Q[0])
        signs, dots = f.hash_signs(Q[0])
        out["signs_q0"], out["dots_q0"] = signs, dots
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "ok")


if __name__ == "__main__":
    main()
