"""Row-range arithmetic of the multi-GPU path: one process per GPU, stored rows sharded by contiguous row
range, one independent LSHIndex per shard (exactly how a user of the reference would shard, README.md:31), queries
replicated.  The exchange itself -- ONE ncclAllGather of every rank's packed [ids | keys | counts] buffer, then the
merge kernel -- lives inside libzebra_hip.so (zh_shard_group_* / zh_shard_search_*, zebra_amd.ShardGroup); nothing here
talks to a collective library."""


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


def packed_views(torch, packed, b, k):
    """packed: int64 tensor of packed_result_words(b, k) words -> (ids [b,k] i64, keys [b,k] i64, counts [b] i32) views,
    the three sections a rank's search writes into so that ONE all-gather of `packed` carries everything"""
    ids = packed[: b * k].view(b, k)
    keys = packed[b * k: 2 * b * k].view(b, k)
    counts = packed[2 * b * k:].view(torch.int32)[:b]
    return ids, keys, counts
