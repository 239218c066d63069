"""The reference's calling pattern: LSHIndex::search with ONE query from many worker threads at once
(/root/reference/src/database/core.rs:299-303 -- rayon over the query batch; lsh.rs:544-549 is the public one-query call).
zh_search_batch combines callers that arrive while a batch is on the GPU into ONE internal batch (the thread that finds the
engine idle leads a round): 64 threads issuing single-query calls must reach >= 20x the throughput of the same calls issued one
after another, and every answer must equal the oracle's bit for bit.  The callers are std::threads of a small C++ program
against the C ABI (tests/cpp/test_concurrent.cpp) -- Python threads would measure the interpreter lock; a second test drives
the same path from Python threads for the answers alone."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

from oracle import zebra_oracle as zo  # noqa: E402  (the checker)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_concurrent.cpp")
LIBDIR = os.path.join(ROOT, "zebra_amd", "lib")


def _compile(out):
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lzebra_hip", f"-Wl,-rpath,{LIBDIR}", "-o", out])


def test_concurrent_harness_compiles(tmp_path):
    _compile(str(tmp_path / "tc"))


@pytest.mark.gpu
def test_64_threads_of_single_query_calls_are_combined(tmp_path):
    import zebra_amd as za
    # the cfg2 shape: 1M x 384, cosine top-10 (the reference's literal key), max_node_size 1024, 15 trees
    n, d, M, T, k, NT, PER = 1_000_000, 384, 1024, 15, 10, 64, 48
    Q = zo.synth_queries(NT * PER, d, n)
    Q.tofile(tmp_path / "q.bin")
    exe = str(tmp_path / "tc")
    _compile(exe)
    r = subprocess.run([exe, str(tmp_path / "q.bin"), str(tmp_path / "r.bin")] + [str(x) for x in (n, d, M, T, k, NT, PER)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    w = r.stdout.split()
    serial, threaded, calls = float(w[1]), float(w[3]), int(w[9])
    NQ = NT * PER
    with open(tmp_path / "r.bin", "rb") as fh:
        counts = np.fromfile(fh, np.uint32, NQ)
        ids = np.fromfile(fh, np.uint64, NQ * k).reshape(NQ, k)
        keys = np.fromfile(fh, np.uint64, NQ * k).reshape(NQ, k)
    # the same index in this process (same options, seeds and synthetic rows -> the same forest), its forest handed to the oracle
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T), seed=0x5EB2A003, reserve_rows=n)
    ix.append_synthetic(n, seed=0x5EB2A001, first_row=0, kind=0)
    ix.build()
    f = zo.Forest.from_arrays(zo.synth_rows(n, d), M, ix.get_forest())
    ix.close()
    oi, ok, oc = f.search_batch(Q, k, zo.COSINE, zo.PARITY)
    assert (counts == oc).all()
    sel = np.arange(k)[None, :] < oc[:, None]
    assert (ids[sel] == oi[sel]).all() and (keys[sel] == ok[sel]).all()
    assert calls >= NQ // 2, r.stdout
    assert threaded >= 20.0 * serial, r.stdout


@pytest.mark.gpu
def test_python_threads_get_their_own_answers():
    import zebra_amd as za
    from zebra_amd import _ffi
    n, d, M, T, NT, PER = 30000, 128, 256, 6, 16, 12
    X = zo.synth_rows(n, d)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M, T))
    ix.add(X)
    f = zo.Forest.from_arrays(X, M, ix.get_forest())
    Q = zo.synth_queries(NT * PER, d, n)
    lib, h = _ffi.lib(), ix._h
    # two kinds of caller at once (different top_k / metric): a round only combines what can share a batch
    kinds = [(10, za.L2SquaredDistance(), zo.L2SQ, 0), (7, za.CosineDistance(parity=False), zo.COSINE, zo.CORRECTED)]
    out = {}
    for t in range(NT):
        k = kinds[t % 2][0]
        out[t] = (np.zeros((PER, k), np.uint64), np.zeros((PER, k), np.uint64), np.zeros(PER, np.uint32))
    errs = []

    def worker(t):
        k, m = kinds[t % 2][0], kinds[t % 2][1]
        ids, keys, counts = out[t]
        for j in range(PER):
            rc = lib.zh_search_batch(h, Q[t * PER + j].ctypes.data_as(C.c_void_p), 1, k, m.metric, m.mode, ids[j].ctypes.data_as(C.c_void_p),
                                     keys[j].ctypes.data_as(C.c_void_p), counts[j:j + 1].ctypes.data_as(C.c_void_p))
            if rc:
                errs.append(rc)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs
    for t in range(NT):
        k, _, om, omode = kinds[t % 2]
        oi, ok, oc = f.search_batch(Q[t * PER:(t + 1) * PER], k, om, omode)
        ids, keys, counts = out[t]
        assert (counts == oc).all()
        sel = np.arange(k)[None, :] < oc[:, None]
        assert (ids[sel] == oi[sel]).all() and (keys[sel] == ok[sel]).all(), t
    ix.close()
