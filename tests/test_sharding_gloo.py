"""The N > 1 path's HOST logic on CPU, two processes over the gloo backend: row ranges and id_base (zebra_amd/sharding.py),
the packed [ids | keys | counts] buffer of zh_packed_result_words words per rank and its rank-major layout after an
all-gather (what zh_shard_search_finish hands to the merge kernel), the control-plane calls bench.py makes
(broadcast of the 128-byte unique id, barrier, max-over-ranks of a float64), and that merging the gathered per-shard
top-k equals the S-shard oracle run in one process (SURVEY s8e 'parity definition').  No GPU here: each rank's shard
engine is the oracle standing in for the local search, and gloo's all_gather stands in for the ncclAllGather that
libzebra_hip.so issues itself on the GPU (tests/test_gpu_shard.py exercises that one with a one-rank communicator)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import zebra_oracle as zo
from zebra_amd import sharding

N, D, M, T, K, B = 6001, 32, 64, 5, 10, 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_search(rank, world):
    first, n = sharding.shard_rows(N, world, rank)
    X = zo.synth_rows(n, D, row0=first)
    f = zo.Forest.build(X, sharding.per_shard_max_node_size(M * world, world, K), T, seed=zo.SEED_INDEX + rank)
    Q = zo.synth_queries(B, D, N)
    ids, keys, counts = f.search_batch(Q, K, zo.L2SQ)
    ids = np.where(np.arange(K)[None, :] < counts[:, None], ids + np.uint64(first), np.uint64(2**64 - 1))
    keys = np.where(np.arange(K)[None, :] < counts[:, None], keys, np.uint64(2**64 - 1))
    return ids, keys, counts


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, keys, counts = _shard_search(rank, world)
    t_ids = torch.from_numpy(ids.view(np.int64).copy())
    t_keys = torch.from_numpy(keys.view(np.int64).copy())
    t_counts = torch.from_numpy(counts.view(np.int32).copy())
    g_ids = torch.empty((world, B, K), dtype=torch.int64)
    g_keys = torch.empty((world, B, K), dtype=torch.int64)
    g_counts = torch.empty((world, B), dtype=torch.int32)
    dist.all_gather_into_tensor(g_ids.view(-1), t_ids.view(-1))
    dist.all_gather_into_tensor(g_keys.view(-1), t_keys.view(-1))
    dist.all_gather_into_tensor(g_counts.view(-1), t_counts.view(-1))
    # the packed form the library exchanges: one buffer per rank, ONE all-gather (zh_packed_result_words is pure
    # arithmetic: callable without a GPU)
    from zebra_amd import _ffi
    W = int(_ffi.lib().zh_packed_result_words(B, K))
    assert W == 2 * B * K + (B + 1) // 2
    # control plane, as bench.py: rank 0's unique id (here: any 128 bytes) reaches every rank
    uid = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    assert uid[0] == bytes(range(128))
    packed = torch.zeros(W, dtype=torch.int64)
    p_ids, p_keys, p_counts = sharding.packed_views(torch, packed, B, K)
    p_ids.copy_(t_ids), p_keys.copy_(t_keys), p_counts.copy_(t_counts)
    g_packed = torch.empty((world, W), dtype=torch.int64)
    dist.all_gather_into_tensor(g_packed.view(-1), packed.view(-1))
    for r in range(world):
        a, b_, c = sharding.packed_views(torch, g_packed[r], B, K)
        assert torch.equal(a, g_ids[r]) and torch.equal(b_, g_keys[r]) and torch.equal(c, g_counts[r])
    # the exchange as libzebra_hip.so issues it: W + 1 words per rank, the last one the rank's STATUS word.  Rank 1 plays a rank
    # whose local search failed with ZH_ELIMIT: an empty slot (counts 0) + its code; every rank derives the same verdict from
    # the gathered words (zh_shard_verdict is pure arithmetic: the real function, no GPU)
    import ctypes
    SW = int(_ffi.lib().zh_shard_exchange_words(B, K))
    assert SW == W + 1
    for failing, code in ((None, 0), (1, _ffi.ZH_ELIMIT), (0, _ffi.ZH_ENOMEM)):
        slot = torch.zeros(SW, dtype=torch.int64)
        my_code = code if rank == failing else 0
        if my_code == 0:
            slot[:W] = packed
        word = int(_ffi.lib().zh_shard_status_word(my_code, 1000 * (rank + 1)))
        slot[W] = word - (1 << 64) if word >= (1 << 63) else word
        g_slots = torch.empty((world, SW), dtype=torch.int64)
        dist.all_gather_into_tensor(g_slots.view(-1), slot.view(-1))
        words = (ctypes.c_uint64 * world)(*[int(g_slots[r, W].item()) & ((1 << 64) - 1) for r in range(world)])
        first, vmax, all_el = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
        verdict = _ffi.lib().zh_shard_verdict(words, world, rank, ctypes.byref(first), ctypes.byref(vmax), ctypes.byref(all_el))
        assert vmax.value == 1000 * world  # the largest visits-per-query any rank reported: sizes the next chunk everywhere
        if failing is None:
            assert verdict == 0 and first.value == world and all_el.value == 0
        else:
            assert first.value == failing and all_el.value == (1 if code == _ffi.ZH_ELIMIT else 0)
            assert verdict == (code if rank == failing else _ffi.ZH_EPEER)  # own code on the failing rank, ZH_EPEER elsewhere
            _, _, c = sharding.packed_views(torch, g_slots[failing][:W], B, K)
            assert int(c.sum()) == 0  # nothing for the merge to read from the failed rank
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    np.savez(os.path.join(out, f"r{rank}.npz"), ids=g_ids.numpy().view(np.uint64), keys=g_keys.numpy().view(np.uint64),
             counts=g_counts.numpy().view(np.uint32), tmax=t.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_allgather_merge(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # every rank holds the same gathered tensors, rank-major
    for k in ("ids", "keys", "counts"):
        assert (got[0][k] == got[1][k]).all()
    assert got[0]["tmax"][0] == 2.0 and got[1]["tmax"][0] == 2.0
    ref = [_shard_search(r, world) for r in range(world)]
    for r in range(world):
        assert (got[0]["ids"][r] == ref[r][0]).all() and (got[0]["keys"][r] == ref[r][1]).all()
        assert (got[0]["counts"][r] == ref[r][2]).all()
    merged = zo.merge_topk(got[0]["ids"], got[0]["keys"], got[0]["counts"], K)
    # parity definition at S > 1: top-k of the union of the shards' candidate lists
    Q = zo.synth_queries(B, D, N)
    Xall = zo.synth_rows(N, D)
    for b in range(B):
        pool = sorted((int(ref[r][1][b, i]), int(ref[r][0][b, i])) for r in range(world) for i in range(ref[r][2][b]))[:K]
        assert [(int(merged[1][b, i]), int(merged[0][b, i])) for i in range(merged[2][b])] == pool
        # global ids really are rows of the unsharded set
        for i in range(merged[2][b]):
            gid = int(merged[0][b, i])
            assert zo.distance(zo.L2SQ, 0, Xall[gid], Q[b]) == int(merged[1][b, i])


def test_shard_arithmetic():
    for total, world in ((100_000_000, 8), (10, 3), (7, 8), (1_000_000_000, 8)):
        spans = [sharding.shard_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == total
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert a + n == b
    assert sharding.per_shard_max_node_size(32768, 8, 10) == 4096  # cfg4 per-shard 4096 / 15
    assert sharding.per_shard_max_node_size(32768, 1, 10) == 32768
    assert sharding.per_shard_max_node_size(64, 8, 100) == 202
