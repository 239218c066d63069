"""Two REAL ranks on two devices (skipped where fewer than two are visible): zh_shard_group_create = ncclCommInitRank with
n_ranks = 2, zh_shard_search_batch_device on a two-shard index -- local search per rank, ONE ncclAllGather of the packed top-k
+ status word over xGMI, the merge kernel on every rank -- against the oracle's two-shard merge (the reference only
claims shardability, /root/reference/README.md:31; the parity definition at S > 1 is top-k of the union of the shards'
candidates).  Then a local failure injected on rank 1 ONLY (ZH_SHARD_INJECT in that process): rank 1 gets its own code, rank 0
gets ZH_EPEER, nobody hangs, and the next batch on the same group answers normally.

Every rank is a fresh child process (this file run as a script) that touches only its own device: no fork after GPU
initialisation, never more than two processes on the card."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, M, T, K, B = 40000, 128, 512, 6, 10, 48


def _worker(rank, world, outdir):
    sys.path.insert(0, ROOT)
    import zebra_amd as za
    from zebra_amd import sharding
    from oracle import zebra_oracle as zo  # (synthetic rows only: the shard's content; the checking happens in the parent)
    first, n = sharding.shard_rows(N, world, rank)
    X = zo.synth_rows(n, D, row0=first)
    Mr = sharding.per_shard_max_node_size(M * world, world, K)
    ix = za.LSHIndex(D, za.LSHIndexOptions(Mr, T), seed=zo.SEED_INDEX + rank, device=rank, id_base=first)
    ix.add(X)
    np.savez(os.path.join(outdir, "forest%d.npz" % rank), **ix.get_forest())
    uid_path = os.path.join(outdir, "uid.bin")
    if rank == 0:
        uid = za.shard_unique_id()
        with open(uid_path + ".tmp", "wb") as fh:
            fh.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no unique id after 120 s" % rank)
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    g = za.ShardGroup(ix, uid, world, rank)
    res = {"ranks": g.ranks(), "rank": g.rank()}
    Q = zo.synth_queries(B, D, N)
    m = za.L2SquaredDistance()
    ids, keys, counts = g.search_batch(Q, K, m)
    np.savez(os.path.join(outdir, "batch0_r%d.npz" % rank), ids=ids, keys=keys, counts=counts)
    # batch 2: the local search of rank 1 fails (injected in THIS process only); every rank still joins the all-gather
    if rank == 1:
        os.environ["ZH_SHARD_INJECT"] = str(-2)  # ZH_ENOMEM: not the collective-halving code
    try:
        g.search_batch(Q, K, m)
        res["second"] = 0
    except za.ZhError as e:
        res["second"] = e.code
    os.environ.pop("ZH_SHARD_INJECT", None)
    ids, keys, counts = g.search_batch(Q, K, m)
    np.savez(os.path.join(outdir, "batch2_r%d.npz" % rank), ids=ids, keys=keys, counts=counts)
    with open(os.path.join(outdir, "res%d.json" % rank), "w") as fh:
        json.dump(res, fh)
    g.close()
    ix.close()


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
    sys.exit(0)

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_two_ranks_two_devices(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (found %d)" % torch.cuda.device_count())
    from oracle import zebra_oracle as zo
    from zebra_amd import _ffi, sharding
    world = 2
    env = dict(os.environ)
    env.pop("ZH_SHARD_INJECT", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(tmp_path)], env=env, cwd=ROOT)
             for r in range(world)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=500))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    assert rcs == [0, 0], rcs
    res = [json.load(open(tmp_path / ("res%d.json" % r))) for r in range(world)]
    assert [r["ranks"] for r in res] == [2, 2] and [r["rank"] for r in res] == [0, 1]
    # the failing rank sees its own code, the other one ZH_EPEER: the same verdict everywhere, nobody hung
    assert res[1]["second"] == -2 and res[0]["second"] == _ffi.ZH_EPEER, res
    # the oracle's two-shard merge
    Q = zo.synth_queries(B, D, N)
    per = []
    for r in range(world):
        first, n = sharding.shard_rows(N, world, r)
        X = zo.synth_rows(n, D, row0=first)
        Mr = sharding.per_shard_max_node_size(M * world, world, K)
        f = zo.Forest.build(X, Mr, T, seed=zo.SEED_INDEX + r)
        got_forest = dict(np.load(tmp_path / ("forest%d.npz" % r)))
        assert zo.canonical_forest(got_forest, D) == zo.canonical_forest(f.arrays(), D), r
        ids, keys, counts = f.search_batch(Q, K, zo.L2SQ)
        ids = np.where(np.arange(K)[None, :] < counts[:, None], ids + np.uint64(first), np.uint64(2**64 - 1))
        keys = np.where(np.arange(K)[None, :] < counts[:, None], keys, np.uint64(2**64 - 1))
        per.append((ids, keys, counts))
    want = zo.merge_topk(np.stack([p[0] for p in per]), np.stack([p[1] for p in per]), np.stack([p[2] for p in per]), K)
    for batch in ("batch0", "batch2"):
        for r in range(world):  # every rank holds the merged answer
            got = np.load(tmp_path / ("%s_r%d.npz" % (batch, r)))
            assert (got["counts"] == want[2]).all(), (batch, r)
            for b in range(B):
                c = int(want[2][b])
                assert (got["ids"][b, :c] == want[0][b, :c]).all() and (got["keys"][b, :c] == want[1][b, :c]).all(), (batch, r, b)
