"""Host-side logic of the multi-GPU path: one process per GPU, stored rows sharded by
contiguous row range, one independent LSHIndex per shard (exactly how a user of the reference would
shard, README.md:31), queries replicated, per-rank top-k exchanged with ONE all-gather (a packed
[ids | keys | counts] buffer per rank) and merged by (key, id).  torch.distributed is plumbing only: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo"
in the CPU tests.  The merge itself is zh_merge_topk_device (a kernel) on the GPU path."""


def shard_rows(total_rows, world, rank):
    """rank -> (first global row, number of local rows); the last rank takes the remainder"""
    per = total_rows // world
    first = rank * per
    n = per if rank < world - 1 else total_rows - first
    return first, n


def per_shard_max_node_size(total_budget, world, top_k):
    """A query scores ~0.68 * max_node_size rows per tree and per shard.  Keeping
    max_node_size_shard = budget / world keeps the rows scored per query -- the work of a batch --
    constant as shards are added, so adding GPUs raises QPS instead of only raising recall.
    Never below 2*top_k + 2, which keeps leaves comfortably >= top_k (one leaf per tree)."""
    return max(total_budget // world, 2 * top_k + 2)


def all_gather_topk(dist, ids, keys, counts, g_ids, g_keys, g_counts):
    """ids/keys [B,k] int64, counts [B] int32 of this rank -> [S,B,k] / [S,B] of all ranks, rank-major.
    Three small collectives (B*k*16 bytes + B*4 per rank: latency-bound on xGMI, not bandwidth-bound)."""
    dist.all_gather_into_tensor(g_ids.view(-1), ids.view(-1))
    dist.all_gather_into_tensor(g_keys.view(-1), keys.view(-1))
    dist.all_gather_into_tensor(g_counts.view(-1), counts.view(-1))


def packed_views(torch, packed, b, k):
    """packed: int64 tensor of packed_result_words(b, k) words -> (ids [b,k] i64, keys [b,k] i64, counts [b] i32) views,
    the three sections a rank's search writes into so that ONE all-gather of `packed` carries everything"""
    ids = packed[: b * k].view(b, k)
    keys = packed[b * k: 2 * b * k].view(b, k)
    counts = packed[2 * b * k:].view(torch.int32)[:b]
    return ids, keys, counts


def all_gather_packed(dist, packed, g_packed):
    """packed [W] int64 of this rank -> g_packed [S, W] of all ranks, rank-major: the batch's one exchange step"""
    dist.all_gather_into_tensor(g_packed.view(-1), packed.view(-1))
