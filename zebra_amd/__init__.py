"""zebra_amd -- MI355X (gfx950) implementation of Zebra's LSH bucket-scan + distance hot path.

Host-side mirror of the reference crate's interface for this path, over the C ABI of
include/zebra_hip.h:

    reference (Rust)                                    here
    Embedding<N>           src/lib.rs:15-46             numpy float32 rows of length N
    DistanceUnit = u64     src/distance.rs:13           numpy uint64 keys
    the 13 metric structs  src/distance.rs:15-190       CosineDistance, L2SquaredDistance, L2Distance, ChebyshevDistance, ...
    LSHIndexOptions<N>     src/database/index/lsh.rs:122-139   LSHIndexOptions
    LSHIndex<N>            src/database/index/lsh.rs:144-565   LSHIndex
    Database<N,Met,Mod>    src/database/core.rs:45-313  Database (insert_records / query_vectors only)
"""
from ._ffi import COSINE, COSINE_CORRECTED, COSINE_PARITY, L2, L2SQ, MAX_TOPK, ZhError  # noqa: F401
from .index import (BrayCurtisDistance, CanberraDistance, ChebyshevDistance, CosineDistance, Database,  # noqa: F401
                    HammingDistance, L2Distance, L2SquaredDistance, L3Distance, L4Distance, LSHIndex, LSHIndexOptions,
                    ManhattanDistance, MinkowskiDistance, PNormDistance, ShardContext, ShardGroup, merge_topk_device, merge_topk_packed_device,
                    packed_result_words, shard_unique_id, synth_queries_device, trim_device_memory)
