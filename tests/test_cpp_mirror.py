"""include/zebra.hpp: compiles everywhere (CPU check); on the GPU box the C++ parity program runs against
expected neighbours computed by the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
LIBDIR = os.path.join(ROOT, "zebra_amd", "lib")


def _compile(out):
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lzebra_hip", f"-Wl,-rpath,{LIBDIR}", "-o", out])


def test_cpp_mirror_compiles_and_links(tmp_path):
    assert os.path.exists(os.path.join(LIBDIR, "libzebra_hip.so"))
    _compile(str(tmp_path / "t"))


@pytest.mark.gpu
def test_cpp_mirror_parity(tmp_path):
    from oracle import zebra_oracle as zo
    n, d, M, T, k, B, seed = 3000, 64, 48, 7, 10, 21, 4242
    X = zo.synth_rows(n, d)
    Q = zo.synth_queries(B, d, n)
    f = zo.Forest.build(X, M, T, seed=seed)
    fx = tmp_path / "fixture.bin"
    with open(fx, "wb") as fh:
        np.array([n, d, M, T, k, B], np.uint32).tofile(fh)
        np.array([seed], np.uint64).tofile(fh)
        X.tofile(fh)
        Q.tofile(fh)
        for om, omode in ((zo.L2SQ, 0), (zo.COSINE, zo.PARITY)):
            ids, keys, counts = f.search_batch(Q, k, om, omode)
            counts.tofile(fh), ids.tofile(fh), keys.tofile(fh)
        np.array([zo.distance(zo.L2SQ, 0, X[0], Q[0]), zo.distance(zo.COSINE, zo.PARITY, X[0], Q[0]),
                  zo.distance(zo.COSINE, zo.CORRECTED, X[0], Q[0]), zo.distance(zo.L2, 0, X[0], Q[0])], np.uint64).tofile(fh)
    exe = str(tmp_path / "t")
    _compile(exe)
    r = subprocess.run([exe, str(fx)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout


EXAMPLE = os.path.join(ROOT, "examples", "search_example.c")


def _compile_c(out):
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), EXAMPLE,
                           "-L", LIBDIR, "-lzebra_hip", f"-Wl,-rpath,{LIBDIR}", "-o", out])


def test_c_example_compiles_as_c99(tmp_path):
    _compile_c(str(tmp_path / "ex"))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    exe = str(tmp_path / "ex")
    _compile_c(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "example ok" in r.stdout, r.stdout + r.stderr
