#!/usr/bin/env python3
"""bench.py -- throughput of the LSH bucket-scan + distance hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N >= 1: for N > 1 this process starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (the same ranks, started by torchrun)

One "step" = one batch of B synthetic queries through the library: hash -> walk -> sweep -> select -> final, plus,
for N > 1, the RCCL all-gather of every rank's packed top-k and the merge kernel (zh_shard_search_*: the exchange is
inside libzebra_hip.so, which links librccl; torch.distributed is used only to launch, to hand rank 0's 128-byte
unique id to the other ranks and for the timing barriers -- on the CPU/gloo side).  Queries and stored vectors are
resident in HBM before the timed region starts; the timed span ends with every batch's results in (pinned) HOST memory.

Workloads (BASELINE.json `configs`; index options from BASELINE.md s3):
  N = 1  -> cfg3: 10M x 768 f32, L2 top-100, batch 1024, max_node_size 4096, num_trees 15.  The metric's own config
            (cfg4, 100M x 768) is 307 GB and does not fit one 288 GB GPU, so the largest single-GPU config is the bench
            line.  The same run then measures, untimed relative to `value`, the other configurations on this GPU
            (`other_configs`: cfg2, one cfg4 shard, one cfg5 shard, the reference-default-options regime, and the N = 1
            point of the multi-GPU series) -- each with its own roofline block.
  N > 1  -> scale64m: 64M x 768 cosine top-10, batch 1024 (the metric's shape at the largest N that fits ONE GPU with
            headroom), strong scaling: rows sharded N ways (global ids), one forest per shard, queries
            replicated, per-shard max_node_size = 32768 / N (4096 at N = 8, BASELINE.md's cfg4 value; leaves stay far
            above top_k at every N, so every point is in the one-leaf-per-tree regime and the rows scored per query --
            the work of a batch -- is the same at every N).  Its N = 1 point is `other_configs.scale64m_n1` of the
            N = 1 run.  `--workload cfg4` runs the metric's own 100M x 768 config the same way (N >= 2);
            `--workload cfg3` the round-1 series (10M rows, 4096 / N).
  --emulate-ranks N on ONE GPU: rank 0's shard of an N-rank run (rows / N, max_node_size / N) with the exchange step
            on a one-rank RCCL communicator in the loop: the per-rank work of a series point ("per-rank, emulated").
Batches are software-pipelined two deep: the small latency-bound kernels of batch i+1 run beside the HBM-bound sweep of
batch i.  --no-pipeline times the blocking call.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md "Indexed rows: gather into LDS", row "2,048 rows shared by every workgroup (the XCD's L2)": 66-73 GB/s per
# CU = 16.8-18.8 TB/s chip-wide for whole rows gathered from L2 -- the operand path that binds the table scan (one query per
# (row, query) pair from L2).  The upper end is the peak, so that frac never flatters.
GATHER_CEILING_GBS = [6410.0, 6560.0]  # random 512-byte .. 3-KiB rows gathered from HBM into registers, two boxes (profiles/micro/r03_gather512.csv)
L2_GATHER_PEAK_GBS = 18800.0
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate -- the `peak` of an L2-bound kernel (the 16.8-18.8 TB/s of
                       # the guide's LDS-gather table is a LOWER bound of one gather shape, not the L2's rate)
L2_GATHER_RANGE_GBS = [16800.0, 18800.0]
# what the bare gather of the scan's access shape reaches from an L2-sized table on this chip with nothing else going on
# (profiles/micro/gather512.hip, r03_gather512.csv: 6-MB table 22.7 TB/s, 3-MB table 31.3 TB/s): the MEASURED ceiling, above the guide's figure
L2_GATHER_MEASURED_GBS = [22700.0, 31300.0]
SEED_ROWS, SEED_Q, SEED_INDEX = 0x5EB2A001, 0x5EB2A002, 0x5EB2A003

WORKLOADS = {
    # name: rows (total), dim, metric, k, batch, max_node_size (total budget over the shards), trees, kind
    "cfg1": dict(rows=10_000, dim=384, metric="cosine", k=10, batch=1, M=5, T=15, kind=0,
                 desc="10k x 384-d f32 vectors, cosine top-10, single query, reference default options (max_node_size 5)"),
    "cfg2": dict(rows=1_000_000, dim=384, metric="cosine", k=10, batch=256, M=1024, T=15, kind=0, window=4,
                 desc="1M x 384-d cosine top-10, query batch=256, LSH index on 1 MI355X"),
    "cfg3": dict(rows=10_000_000, dim=768, metric="l2", k=100, batch=1024, M=4096, T=15, kind=0,
                 desc="10M x 768-d L2 top-100, query batch=1024, 1 MI355X (HBM-bound candidate sweep)"),
    "cfg4": dict(rows=100_000_000, dim=768, metric="cosine", k=10, batch=1024, M=32768, T=15, kind=0,
                 desc="100M x 768-d cosine top-10, batch=1024, vectors sharded across GPUs + RCCL top-k merge"),
    # (window 4, round 6: the fused half-width sweep is HBM-bound on the rows it loads, and a leaf's rows are loaded once per group of <= 4 queries
    # of the WINDOW: 0.69 row loads per scored row at four batches against 0.83 at two -- 370 k against 315 k QPS on one shard, for twice the latency)
    "cfg5": dict(rows=1_000_000_000, dim=128, metric="l2", k=10, batch=4096, M=65536, T=15, kind=1, window=4,
                 desc="1B x 128-d SIFT-style L2 top-10, batch=4096, 8 GPUs"),
    "scale64m": dict(rows=64_000_000, dim=768, metric="cosine", k=10, batch=1024, M=32768, T=15, kind=0,
                     desc="64M x 768-d cosine top-10, batch=1024 (the metric's shape at the largest N that fits one GPU), "
                          "rows sharded across GPUs + RCCL top-k merge"),
    # (its own host loop: every batch its own internal batch, two begun ahead -- the batch's time is hash + walk, a chain of dependent
    # latencies that only other batches' chains can overlap; measured r03: window 2 / one ahead 11.9 ms, window 1 / two ahead 10.5 ms)
    "refdefault": dict(rows=1_000_000, dim=384, metric="l2sq", k=10, batch=256, M=5, T=15, kind=0, window=1, lookahead_depth=2,
                       desc="1M x 384-d L2^2 top-10, batch=256, the reference's DEFAULT options (max_node_size 5, lsh.rs:131-138): "
                            "the wandering walk of DefaultTextDatabase"),
    "tiny": dict(rows=200_000, dim=768, metric="l2", k=100, batch=256, M=1024, T=15, kind=0,
                 desc="200k x 768-d L2 top-100 (debug)"),
}
# what the N = 1 run measures besides `value`: (key, workload, shards it is one of, steps[, window])
# cfg3_window_of_one: the bench line's workload with every batch its own internal batch -- the latency / throughput trade of --window
# (steps: a multiple of the window and enough windows that the pipeline's fill and drain -- one window's light kernels -- do not show)
OTHER_CONFIGS = [("cfg3_window_of_one", "cfg3", 1, 12, 1), ("cfg4_one_of_8_shards", "cfg4", 8, 20), ("cfg2", "cfg2", 1, 120),
                 ("cfg5_one_of_8_shards", "cfg5", 8, 24),
                 # the same shard with EIGHT API batches per window instead of four (a leaf's rows are loaded once per <= 4 queries of the WINDOW that visit it:
                 # 0.51 row loads per scored row instead of 0.69) -- throughput for latency, the caller's choice (zh_search_ctx_begin_window)
                 ("cfg5_shard_window_of_eight", "cfg5", 8, 32, 8),
                 ("reference_default_options", "refdefault", 1, 24), ("scale64m_n1", "scale64m", 1, 8),
                 ("cfg1_single_query", "cfg1", 1, 60),  # (configs[0]: the reference's own CPU-runnable case; what matters is latency_ms.p50_blocking_single_batch)
                 # north_star's ">= 70 % of HBM on the candidate distance sweep" on a driver-run line: the bench line's workload through the f32
                 # LEAF-MAJOR sweep (zh_set_sweep_mode(1): sweep_kernel<768>, HBM-bound, SURVEY s8(d)'s bytes are what it moves) beside the faster
                 # L2-bound matrix-core scan that is the default ("deliberate deviations")
                 ("cfg3_leaf_major_f32", "cfg3", 1, 6)]
OTHER_SWEEP_MODE = {"cfg3_leaf_major_f32": "leaf"}


def rank_env(base, rank, world, port):
    """environment of rank `rank` of a `world`-rank run on this node (what torch.distributed.run would set)"""
    e = dict(base)
    e.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
              "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL across processes)
    return e


def count_devices():
    """GPUs visible to a rank, counted in a short-lived CHILD interpreter: whatever the count costs (on ROCm builds without amdsmi
    torch falls back to hipGetDeviceCount, which initialises HIP), the launcher itself never touches a GPU -- the ranks are started
    from a process that has not initialised the runtime."""
    import subprocess
    try:
        p = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
        return int(p.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001 -- no torch / no answer: the ranks will say what is wrong
        return 0


def launch_ranks(n, argv, timeout_s=None, poll_s=0.2, count=count_devices, script=None):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (one rank per GPU; this parent never touches
    a GPU and never execs), relay rank 0's stdout, SUPERVISE the ranks: on the first non-zero exit (or after `timeout_s`) the
    survivors are terminated, then killed, and that exit code (124 for the timeout) is returned -- a rank that dies at start-up
    cannot leave the others waiting in a rendezvous or a collective.  The ranks meet through a FILE store (ZH_BENCH_RDZV_FILE:
    no port is picked here, so none can be taken between a bind and its reuse); MASTER_PORT is only there for code that reads it.
    Fewer than N devices: an error line, rc 3."""
    import subprocess
    import tempfile
    import threading
    found = count()
    if found < n:
        print(json.dumps({"error": "needs %d devices, found %d" % (n, found), "n_gpus": n}))
        return 3
    timeout_s = timeout_s or float(os.environ.get("ZH_BENCH_LAUNCH_TIMEOUT_S", "3300"))
    rdzv = tempfile.mkdtemp(prefix="zh_bench_rdzv_")
    base = dict(os.environ)
    base["ZH_BENCH_RDZV_FILE"] = os.path.join(rdzv, "store")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=rank_env(base, r, n, 29400),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out = []
    rd = threading.Thread(target=lambda: out.extend(procs[0].stdout), daemon=True)  # (a full pipe must not block rank 0)
    rd.start()
    t0, why, code = time.monotonic(), None, 0
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            why, code = "ranks failed (rank, exit code): %s" % bad, abs(bad[0][1])
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() - t0 > timeout_s:
            why, code = "timeout after %.0f s" % timeout_s, 124
            break
        time.sleep(poll_s)
    if why:  # end the survivors: a rank waiting for a dead peer would sit out its own (long) timeouts
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10:
            time.sleep(poll_s)
        for p in procs:
            if p.poll() is None:
                p.kill()
    rcs = [p.wait() for p in procs]
    rd.join(timeout=10)
    try:
        import shutil
        shutil.rmtree(rdzv, ignore_errors=True)
    except Exception:  # noqa: BLE001
        pass
    text = "".join(out)
    sys.stdout.write(text)
    sys.stdout.flush()
    if why:
        print("bench.py: %s" % why, file=sys.stderr)
        print(json.dumps({"error": why, "n_gpus": n}))  # the LAST line says the run failed, whatever rank 0 printed before
        return code or 1
    return 0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="override: " + ",".join(WORKLOADS))
    ap.add_argument("--rows", type=int, default=None, help="override total rows (debug)")
    ap.add_argument("--max-node-size", type=int, default=None, help="override max_node_size (debug)")
    ap.add_argument("--batch", type=int, default=None, help="override the query batch (debug)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of EACH CPU baseline leg (0 = skip)")
    ap.add_argument("--recall-queries", type=int, default=256)
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-main-recall", action="store_true", help="debug: no recall run (torch brute force) for the bench line's own workload only")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: only the bench line's own workload")
    ap.add_argument("--only-other", default=None, help="debug: comma list of other_configs keys to run (in THIS order)")
    ap.add_argument("--sleep-before-other", type=float, default=0.0, help="debug: idle seconds before each of the other configurations (the order effect, DESIGN.md s9)")
    ap.add_argument("--data", choices=["iid", "clustered", "clustered-shuffled"], default="iid",
                    help="iid: BASELINE.md's ~N(0,1) rows (the bench line); clustered: 128-row clusters, where recall@k is informative; "
                         "clustered-shuffled: the same clusters with their rows scattered over the table (a row's cluster from a hash of its id)")
    ap.add_argument("--no-pipeline", action="store_true", help="one blocking search call per step")
    ap.add_argument("--in-flight", type=int, default=2, help="windows in flight when pipelined (contexts); the look-ahead host loop uses one more")
    ap.add_argument("--lookahead", choices=["auto", "on", "off"], default="auto",
                    help="host loop: begin window w+1 before finishing window w (its hash runs beside w's walk).  Pays when hash + walk "
                         "outweigh the sweep (reference-default options: -8 %% per batch), costs elsewhere; auto decides from the warm-up's stage times")
    ap.add_argument("--lookahead-depth", type=int, default=None,
                    help="windows BEGUN ahead of the one being finished when the look-ahead loop runs (their hash and walk are enqueued on their "
                         "own streams and run beside the current window's walk / sweep)")
    ap.add_argument("--window", type=int, default=None,
                    help="batches handled as ONE internal batch (zh_search_begin_window): rows shared across the window's queries")
    ap.add_argument("--sweep-mode", choices=["auto", "leaf", "scan", "approx", "approx-valu", "leaf-half"], default="auto",
                    help="zh_set_sweep_mode: leaf by leaf, table scan, or chosen per batch by the library (default)")
    ap.add_argument("--corrected-key", action="store_true",
                    help="debug / A-B: cosine workloads timed with the CORRECTED key (cosine distance) instead of the reference's literal one "
                         "(distance.rs:23-25: similarity): same cost per pair, far shorter candidate lists on long leaves")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="on ONE GPU: run rank 0's shard of an N-rank job, exchange on a one-rank RCCL communicator")
    ap.add_argument("--debug-normal-priority-sweeps", action="store_true", help="A/B: sweeps on a normal-priority stream (N = 1)")
    ap.add_argument("--pmc-summary", default=None, help="profiles/*_pmc_hbm_bytes.json to take roofline.traffic from")
    ap.add_argument("--serial-windows", action="store_true",
                    help="profiling: the pipelined calls and the same --window, but ONE window on the GPU at a time (drained before the next "
                         "begins): kernels do not overlap, so per-kernel PMC counters are the kernel's own, at the timed run's window size")
    ap.add_argument("--profile-run", action="store_true",
                    help="profiling: nothing but full windows of the bench workload (warm-up rounded up to whole windows, no R_unique "
                         "pass, no host-buffer leg, no recall): every launch of the sweep kernel in the process is comparable")
    return ap.parse_args()


def make_metric(za, name, parity=True):
    return {"cosine": za.CosineDistance(parity=parity), "l2": za.L2Distance(), "l2sq": za.L2SquaredDistance()}[name]


def exact_topk(torch, X, q, k, metric, chunk=1 << 20):
    """brute-force top-k of the true metric (cosine distance / L2) with torch matmul, for recall"""
    n = X.shape[0]
    best_v = torch.full((q.shape[0], k), float("inf"), device=q.device)
    best_i = torch.zeros((q.shape[0], k), dtype=torch.int64, device=q.device)
    qn = (q * q).sum(1)
    for s in range(0, n, chunk):
        xs = X[s:s + chunk]
        dots = q @ xs.T
        xn = (xs * xs).sum(1)
        if metric == "cosine":
            dist = 1.0 - dots / torch.sqrt(qn[:, None] * xn[None, :]).clamp_min(1e-30)
        else:
            dist = qn[:, None] + xn[None, :] - 2.0 * dots
        v, i = torch.topk(dist, min(k, xs.shape[0]), dim=1, largest=False)
        cv = torch.cat([best_v, v], 1)
        ci = torch.cat([best_i, i + s], 1)
        o = torch.topk(cv, k, dim=1, largest=False)
        best_v, best_i = o.values, torch.gather(ci, 1, o.indices)
    return best_i


class Env:
    """process-wide state: torch, the library mirror, rank / world, the control-plane process group"""

    def __init__(self, args):
        import torch
        import zebra_amd as za
        from zebra_amd import sharding
        self.torch, self.za, self.sharding, self.args = torch, za, sharding, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            print(json.dumps({"error": "--gpus %d but WORLD_SIZE is %d" % (args.gpus, self.world), "n_gpus": args.gpus}))
            sys.exit(2)
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # control plane only (unique id, barriers, max-over-ranks of the elapsed time): gloo on the CPU.  The data
            # path's one collective is the ncclAllGather inside libzebra_hip.so.
            rdzv = os.environ.get("ZH_BENCH_RDZV_FILE")  # set by launch_ranks: a file store, no port to race for
            if rdzv:
                dist.init_process_group("gloo", init_method="file://" + rdzv, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def unique_id(self):
        """rank 0's RCCL unique id, on every rank"""
        uid = [self.za.shard_unique_id() if self.rank == 0 else None]
        if self.dist:
            self.dist.broadcast_object_list(uid, src=0)
        return uid[0]


def run_workload(env, name, S, rank, steps, warmup, exchange, recall=False, M_override=None, rows_override=None,
                 batch_override=None, kind_override=None, window_override=None, sweep_override=None):
    """Build rank `rank`'s shard of an S-way sharding of workload `name` on this GPU, time `steps` batches, and return
    the result fields.  exchange: a ShardGroup is created (world ranks when S == world > 1, else ONE rank) and every
    batch goes through zh_shard_search_* (local search + all-gather + merge)."""
    torch, za, sharding, args, dev = env.torch, env.za, env.sharding, env.args, env.dev
    wl = dict(WORKLOADS[name])
    if rows_override:
        wl["rows"] = rows_override
    if M_override:
        wl["M"] = M_override
    if batch_override:
        wl["batch"] = batch_override
    if kind_override is not None:
        wl["kind"] = kind_override
    first_row, rows_local = sharding.shard_rows(wl["rows"], S, rank)
    M_shard = sharding.per_shard_max_node_size(wl["M"], S, wl["k"]) if S > 1 else wl["M"]
    d, T, k, B = wl["dim"], wl["T"], wl["k"], wl["batch"]
    metric = make_metric(za, wl["metric"], parity=not args.corrected_key)
    n_total = wl["rows"]

    # ---- setup (untimed): synthetic rows on the device, GPU forest build ------------------------------------------
    t_setup = time.perf_counter()
    ix = za.LSHIndex(d, za.LSHIndexOptions(M_shard, T), seed=SEED_INDEX + rank, device=env.local_rank, id_base=first_row,
                     reserve_rows=rows_local)
    ix.set_sweep_mode(sweep_override or args.sweep_mode)
    ix.append_synthetic(rows_local, seed=SEED_ROWS, first_row=first_row, kind=wl["kind"])
    t_fill = time.perf_counter() - t_setup
    ix.build()
    t_build = time.perf_counter() - t_setup - t_fill
    group = None
    if exchange:
        real = env.world > 1 and S == env.world
        uid = env.unique_id() if real else za.shard_unique_id()
        group = za.ShardGroup(ix, uid, env.world if real else 1, env.rank if real else 0)

    pipelined = not args.no_pipeline
    NS = max(2, args.in_flight)
    WIN = max(1, window_override or args.window or wl.get("window", 2)) if pipelined else 1  # (default 2)
    if args.profile_run and pipelined:
        warmup = (warmup + WIN - 1) // WIN * WIN  # whole windows only: every sweep launch of the process is a full-window launch
    n_batches = steps + warmup
    queries = []
    for i in range(n_batches):
        q = torch.empty((B, d), dtype=torch.float32, device=dev)
        za.synth_queries_device(env.local_rank, q.data_ptr(), n_total, B, d, b0=i * B, seed_rows=SEED_ROWS, seed_q=SEED_Q,
                                kind=wl["kind"])
        queries.append(q)

    def make_results():
        r = dict(ids=torch.empty((B, k), dtype=torch.int64, device=dev), keys=torch.empty((B, k), dtype=torch.int64, device=dev),
                 counts=torch.empty(B, dtype=torch.int32, device=dev))
        r.update(h_ids=torch.empty((B, k), dtype=torch.int64).pin_memory(), h_keys=torch.empty((B, k), dtype=torch.int64).pin_memory(),
                 h_counts=torch.empty(B, dtype=torch.int32).pin_memory())
        return r

    def to_host(r, stream):
        """the span ends with the merged top-k ON THE HOST: async copies behind the last kernel"""
        with torch.cuda.stream(stream):
            r["h_ids"].copy_(r["ids"], non_blocking=True)
            r["h_keys"].copy_(r["keys"], non_blocking=True)
            r["h_counts"].copy_(r["counts"], non_blocking=True)

    r0 = make_results()
    cur = torch.cuda.current_stream()
    LA = max(1, args.lookahead_depth or wl.get("lookahead_depth", 1))  # extra slots of the look-ahead loop (default 1)
    if group is None:
        heavy = ix.sweep_stream()
        if args.debug_normal_priority_sweeps:
            _hs = torch.cuda.Stream(device=dev, priority=0)
            heavy = _hs.cuda_stream
        slots = [dict(ctx=ix.search_context(), stream=torch.cuda.Stream(device=dev, priority=-1), res=[make_results() for _ in range(WIN)])
                 for _ in range(NS + LA)] if pipelined else []

        def begin(sl, i, nw):
            sl["nw"] = nw
            sl["ctx"].begin_window([queries[i + j].data_ptr() for j in range(nw)], B, k, metric, sl["stream"].cuda_stream)

        def finish(sl):
            rs = sl["res"][:sl["nw"]]
            sl["ctx"].finish_window([r["ids"].data_ptr() for r in rs], [r["keys"].data_ptr() for r in rs],
                                    [r["counts"].data_ptr() for r in rs], heavy)
            sl["held"] = True
            if "t_begin" in sl:  # behind the window's last kernel, on its own stream: the results are complete on the DEVICE here
                sl["ev_r"].record(sl["stream"])

        def retire(sl):
            """zh_search_wait FIRST, then the D2H copies: the header says a batch's outputs are complete only when wait has returned
            (a prefiltered batch whose lists ran over is redone from the host inside it) -- copies queued right behind finish could
            carry the discarded results to the host.  Called when the slot is needed again (two windows later) and at the end."""
            if not sl.get("held"):
                return
            sl["ctx"].wait()
            for r in sl["res"][:sl["nw"]]:
                to_host(r, sl["stream"])
            sl["held"] = False
            if "t_begin" in sl:
                _mark_end(sl)

        def drain(sl):
            retire(sl)
            sl["stream"].synchronize()

        def blocking(i):
            ix.search_batch_device(queries[i].data_ptr(), B, k, metric, r0["ids"].data_ptr(), r0["keys"].data_ptr(),
                                   r0["counts"].data_ptr(), cur.cuda_stream)
            to_host(r0, cur)
    else:
        # The merged results complete on a stream the LIBRARY owns (zh_shard_ctx_stream).  The D2H copies go behind it with
        # hipMemcpyAsync on the raw handle -- not through a torch stream wrapper: torch's pinned-memory allocator records
        # events on every stream a pinned block was used on when the block is freed, i.e. after the library destroyed it.
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        slots = [dict(ctx=group.search_context(), res=[make_results() for _ in range(WIN)]) for _ in range(NS + LA if pipelined else 0)]

        def begin(sl, i, nw):
            sl["nw"] = nw
            sl["ctx"].begin_window([queries[i + j].data_ptr() for j in range(nw)], B, k, metric)

        def retire(sl):
            pass  # (the sharded contexts never take a path that is redone from the host: results are final in stream order)

        def finish(sl):
            rs = sl["res"][:sl["nw"]]
            sl["ctx"].finish_window([r["ids"].data_ptr() for r in rs], [r["keys"].data_ptr() for r in rs],
                                    [r["counts"].data_ptr() for r in rs])
            xs = sl["ctx"].stream()
            for r in rs:
                for h, dsrc in ((r["h_ids"], r["ids"]), (r["h_keys"], r["keys"]), (r["h_counts"], r["counts"])):
                    rc = hip.hipMemcpyAsync(h.data_ptr(), dsrc.data_ptr(), h.numel() * h.element_size(), 2, xs)  # hipMemcpyDeviceToHost
                    assert rc == 0, rc

        def drain(sl):
            sl["ctx"].wait()
            assert hip.hipStreamSynchronize(sl["ctx"].stream()) == 0

        def blocking(i):
            group.search_batch_device(queries[i].data_ptr(), B, k, metric, r0["ids"].data_ptr(), r0["keys"].data_ptr(),
                                      r0["counts"].data_ptr())
            to_host(r0, cur)

    look = {"ahead": args.lookahead == "on"}
    # Per-window latency inside the pipelined run: two events on the window's own (light, high-priority) stream -- one recorded
    # right before begin() is called (the stream is idle then: the slot's previous window completed long ago, so the event's
    # timestamp is the submission time), one behind the D2H copies of the window's last batch.  Only the single-GPU loop: the
    # sharded loop's results complete on a stream the library owns.
    lat_store = {"on": False, "ms": [], "ready_ms": []}
    lat = None

    def _mark_begin(sl):
        if "ev_b" not in sl:
            sl["ev_b"], sl["ev_e"], sl["ev_r"] = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        sl["ev_b"].record(sl["stream"])
        sl["t_begin"] = True  # (the end event goes behind the D2H copies, in retire)

    def _mark_end(sl):
        sl["ev_e"].record(sl["stream"])

    def _window_latency(sl):
        sl["ev_e"].synchronize()
        lat_store["ready_ms"].append(sl["ev_b"].elapsed_time(sl["ev_r"]))
        return sl["ev_b"].elapsed_time(sl["ev_e"])

    def run(first, n):
        nonlocal lat
        lat = lat_store["ms"] if (lat_store["on"] and group is None and not look["ahead"]) else None
        if not pipelined:
            for i in range(first, first + n):
                blocking(i)
            torch.cuda.synchronize()
            return
        # Window w + 1 is BEGUN (hash + walk enqueued on its own stream; begin first retires window w + 1 - NS of that slot,
        # long done) before window w is FINISHED: finish blocks the host until w's counting pass has run, and with the next
        # window's hash already queued the GPU runs it beside w's walk (the MFMA-bound hash and the latency-bound walk of the
        # reference-default regime do not compete) and beside the sweep of w - 1; finish then leaves w's sweep queued.
        wins = []
        i = first
        while i < first + n:
            nw = min(WIN, first + n - i)
            wins.append((i, nw))
            i += nw
        if args.serial_windows:  # profiling: one window on the GPU at a time
            for w in range(len(wins)):
                begin(slots[0], *wins[w])
                finish(slots[0])
                drain(slots[0])
        elif not look["ahead"]:  # begin + finish of a window back to back, NS windows in flight
            for w in range(len(wins)):
                sl = slots[w % NS]
                retire(sl)  # the window this slot held (two windows ago: long done): wait, then its results to the host
                if lat is not None:
                    if "t_begin" in sl:
                        lat.append(_window_latency(sl))
                    _mark_begin(sl)
                elif "t_begin" in sl:
                    del sl["t_begin"]
                begin(sl, *wins[w])
                finish(sl)
            if lat is not None:
                for sl in slots[:NS]:
                    drain(sl)
                    if "t_begin" in sl:
                        lat.append(_window_latency(sl))
                        del sl["t_begin"]
        else:                  # LA more slots: LA windows begun ahead, one being finished, NS - 1 sweeping
            for w in range(min(LA, len(wins))):
                retire(slots[w % (NS + LA)])
                begin(slots[w % (NS + LA)], *wins[w])
            for w in range(len(wins)):
                if w + LA < len(wins):
                    retire(slots[(w + LA) % (NS + LA)])
                    begin(slots[(w + LA) % (NS + LA)], *wins[w + LA])
                finish(slots[w % (NS + LA)])
        for sl in slots:
            drain(sl)

    torch.cuda.synchronize()
    for sl in slots:  # every slot allocates its scratch once, whatever --warmup is (not counted as warmup)
        begin(sl, 0, min(WIN, n_batches))
        finish(sl)
        drain(sl)
    if not pipelined:
        blocking(0)
    ix.set_profiling(1)  # hipEvents around every stage, on the stream the kernels run on
    ix.stats(reset=True)
    if warmup:
        run(0, warmup)
    if args.lookahead == "auto" and pipelined and warmup:
        sw = ix.stats()
        look["ahead"] = sw["ms_hash"] + sw["ms_walk"] > 1.5 * sw["ms_sweep"]  # same decision on every rank would need a collective:
        if env.dist and exchange:                                                 # rank 0 decides
            flag = [look["ahead"]]
            env.dist.broadcast_object_list(flag, src=0)
            look["ahead"] = flag[0]
    ix.stats(reset=True)
    torch.cuda.synchronize()
    env.barrier()
    torch.cuda.synchronize()
    lat_store["on"] = pipelined and not args.serial_windows
    t0 = time.perf_counter()
    run(warmup, steps)
    lat_store["on"] = False
    torch.cuda.synchronize()
    env.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if env.dist and exchange:
        t = torch.tensor([elapsed], dtype=torch.float64)
        env.dist.all_reduce(t, op=env.dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ix.stats()
    last = slots[((steps + WIN - 1) // WIN - 1) % (NS + LA if look["ahead"] else NS)]["res"][(steps - 1) % WIN] if pipelined else r0
    last_host = (last["h_ids"].numpy().copy(), last["h_counts"].numpy().copy())

    # ---- untimed: R_unique of the timed batches (per internal batch = per window: a row shared by two batches of a window
    # counts once, so `achieved` never exceeds what the kernel really moved) ---------------------------------------
    ix.set_profiling(2)
    uniq = tot = 0
    for i in range(0 if args.profile_run else max(1, min(steps // WIN, 4))):
        ix.stats(reset=True)
        if pipelined:
            begin(slots[0], warmup + i * WIN, min(WIN, steps - i * WIN))
            finish(slots[0])
            drain(slots[0])
        else:
            blocking(warmup + i)
            torch.cuda.synchronize()
        s2 = ix.stats()
        uniq += s2["rows_unique"]
        tot += s2["rows_scored"]
    uniq_frac = uniq / max(tot, 1) if tot else 1.0
    ix.set_profiling(0)

    # one batch's sweep is issued as several launches of the same kernel (~12 GB each): per-launch figures
    n_launch = max(st["sweep_launches_accum"], 1)
    rows_per_launch = st["sweep_rows_accum"] / n_launch
    sweep_ms = st["ms_sweep"] / n_launch
    n_timed = max(st["timed_batches"], 1) * (WIN if steps >= WIN else 1)  # the library counts internal batches = windows
    launches_per_batch = n_launch / n_timed
    # algorithmic bytes of one sweep launch (DESIGN.md "Kernels"): every distinct stored row crosses HBM once (4*d bytes),
    # every scored row reads a 4-byte leaf id and writes an 8-byte key, plus the query batch
    # SURVEY s8(d)'s numerator, the same for both sweep kinds: every row of every DISTINCT LEAF the launch's visits touch crosses
    # HBM once (4*d bytes), every scored (row, query) pair reads a 4-byte leaf id and writes an 8-byte key, plus the queries
    bytes_alg = 4.0 * d * rows_per_launch * uniq_frac + 12.0 * rows_per_launch + 4.0 * d * B / launches_per_batch
    bytes_nosharing = (4.0 * d + 12.0) * rows_per_launch
    scan = st["scan_batches_accum"] > 0
    half = st.get("approx_batches_accum", 0) > 0  # the scan read fp16 copies of the queries (zh_approx.hip)
    kind = 1 if wl["metric"] == "cosine" else 0
    kname = ("scan_sweep_kernel<%d, %d>" if scan else "sweep_kernel<%d, %d, ...>") % (d, kind)  # <D, KIND (0 = L2, 1 = cosine), ...>
    if not scan and d == 128:
        kname = "sweep128_kernel<%d, ...>" % kind  # the half-wave kernel of the 512-byte rows
    leaf_half = half and st.get("approx_scan", 0) == 3  # d = 128, leaf by leaf from the fp16 copy of the rows (sweep128h_kernel)
    fused = leaf_half and st.get("approx_fused", 0) == 1
    byte_rows = leaf_half and st.get("approx_byte_rows", 0) == 1  # every stored element an integer 0..255 (cfg5's SIFT-style rows): the EXACT 128-byte copy
    if leaf_half:
        # round 6: sweep128h_lean_kernel (+ sweep128h_boundary_kernel for the chunks that cross a leaf group).  FUSED (long leaves, top_k <= 64): the
        # intervals, the bounds and the queries' candidate lists are made inside the sweep -- no 8-byte result per scored row is written (or read
        # back by a select pass), so the algorithmic bytes per scored row are the row (2 d, once per distinct leaf group) and its 4-byte leaf id
        per_row = 4.0 if fused else 12.0
        kinda = 0 if wl["metric"] != "cosine" else 2  # (approx_interval's kind: 0 the L2 family, 2 the reference's literal cosine key)
        kname = (("sweep128b_lean_kernel<64, %d>" if byte_rows else "sweep128h_lean_kernel<16, %d>") % kinda) if fused else "sweep128%s_lean_kernel<4, -1>" % ("b" if byte_rows else "h")
        row_b = 1.0 * d if byte_rows else 2.0 * d  # what the copy holds per stored row = what a launch must move per distinct row (traffic: PMC)
        bytes_alg = row_b * rows_per_launch * uniq_frac + per_row * rows_per_launch + 2.0 * d * B / launches_per_batch
        bytes_nosharing = (row_b + per_row) * rows_per_launch
    elif half:  # <D, groups per wave, 0 = L2 family / 1 = cosine distance / 2 = the reference's literal cosine key>
        kname = "scan_approx_kernel<%d, %d, %d>" % (d, 2 if d >= 512 else 4, 0 if wl["metric"] != "cosine" else 2)
        if st.get("approx_scan", 0) == 2:
            # the same scan on the matrix cores: <D, false> from the index's fp16 copy of the stored rows; <D, true> (no room for the copy, 2 d bytes per
            # row): the scan converts the f32 rows itself
            kname = "scan_mfma_kernel<%d, %s>" % (d, "true" if st.get("row_copy_bytes", 0) < rows_local * d else "false")
    s8d_GBps = bytes_alg / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    common = {"kernel": kname, "launch_ms": sweep_ms, "rows_per_launch": rows_per_launch, "unique_row_fraction": uniq_frac,
              "rows_loaded_per_launch": st["swept_rows_accum"] / n_launch, "launches_per_batch": launches_per_batch,
              "window_batches": WIN}
    if st.get("prefiltered"):
        # The batch was hashed from row scores and PREFILTERED (zh_set_sweep_mode): no sweep ran.  `ms_sweep` is prefilter_kernel -- one
        # wave per (query, tree) replays the pair's visits on the score table -- and one launch serves the whole internal batch.
        # Algorithmic bytes per scored row: one 16-byte record {row id, |r|^2/2, |r|} of its leaf slot + 4 (its score); per visit: 8 (the log
        # entry {leaf offset, take | length << 16}).  The score is a 4-byte read from a random 1-KiB row of the table: the kernel moves a
        # 64-byte sector for it (sector_GBps).
        n_b = max(st["timed_batches"], 1)
        pf_ms = st["ms_sweep"] / n_b
        rows_b, visits_b = st["sweep_rows_accum"] / n_b, st["visits"]
        bytes_alg = 20.0 * rows_b + 8.0 * visits_b
        bytes_sector = (64.0 + 16.0) * rows_b + 8.0 * visits_b
        g = bytes_alg / (pf_ms * 1e-3) / 1e9 if pf_ms else 0.0
        roof = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_launch": bytes_alg, "kernel": "prefilter_kernel<%d>" % (2 if wl["metric"] == "cosine" else 0), "launch_ms": pf_ms,
                "rows_per_launch": rows_b, "visits_per_launch": visits_b, "launches_per_batch": 1.0 / (WIN if steps >= WIN else 1), "window_batches": WIN,
                "sector_GBps": bytes_sector / (pf_ms * 1e-3) / 1e9 if pf_ms else 0.0,
                "exact_rows_per_launch": st["prefilter_exact_rows"], "exact_visits_per_launch": st["prefilter_exact_visits"],
                "prefilter_fallbacks": st["prefilter_fallbacks_accum"],
                "note": "no sweep in this regime: candidates are judged on the hash's row scores (64-byte sector gathers: latency-bound, not "
                        "byte-bound); the batch's time is the hash (row-score GEMM + sign gather) and the walk -- stage_ms_per_batch"}
    elif not scan:
        # Leaf-major sweep: SURVEY s8(d)'s bytes ARE what the kernel moves through HBM (PMC traffic 0.99-1.01x): HBM roofline.
        roof = {"bound": "hbm", "achieved": s8d_GBps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": s8d_GBps / HBM_PEAK_GBS,
                "traffic": None, "bytes_per_launch": bytes_alg,
                "sweep_mode": ("leaf by leaf at HALF WIDTH, FUSED: %s x fp16 queries on v_mfma_f32_16x16x32_f16; the intervals, "
                               "a running top-k bound per visit and wave, and the queries' candidate lists are made inside the sweep -- no per-pair "
                               "result is written, no select pass reads one back (zh_approx.hip, round 6)"
                               % ("an EXACT 128-byte copy of the rows (every element an integer 0..255, checked when the copy is made; bytes -> halves in "
                                  "registers)" if byte_rows else "fp16 rows (one table scale)")) if fused else
                              ("leaf by leaf at half width: raw pairs to the key scratch, select_tau / select_emit behind" if leaf_half else
                               "leaf by leaf, f32 rows, canonical keys"),
                "achieved_no_sharing_GBps": bytes_nosharing / (sweep_ms * 1e-3) / 1e9 if sweep_ms else 0.0,
                # what a bare random whole-row gather into registers reaches on this chip (no arithmetic, same access shape, 64-GB
                # table; profiles/micro/gather512.hip -> profiles/micro/r03_gather512.csv): the ceiling of a leaf-major sweep
                "measured_gather_ceiling_GBps": GATHER_CEILING_GBS,
                "frac_of_measured_gather_ceiling": s8d_GBps / GATHER_CEILING_GBS[0]}
        roof.update(common)
    else:
        # Table scan: the stored rows cross HBM ONCE per window (address order) and every (row, query) pair fetches its query
        # (4*d bytes) from L2.  The binding resource is the L2 -> CU operand path, not HBM, so that is the roofline quoted:
        #   achieved = (pairs + stored rows of a launch) * 4*d bytes / launch time   [what the CUs pull through L1 misses]
        #   peak     = the guide's measured ceiling for whole rows gathered from L2 (MI355X_MICROARCH.md, 'Indexed rows', L2 row)
        # beside it: the HBM side of the same launch (hbm_frac: bytes that cross HBM BY DESIGN / launch time / 8 TB/s; `traffic`
        # is the PMC figure for the same launch) and, without a frac, SURVEY s8(d)'s numerator over the launch time -- the figure
        # comparable with the leaf-major sweep and round 1, which exceeds the HBM peak because one HBM read of a row serves every
        # tree that wants it: those bytes are not bytes this kernel moves.
        stored = st["swept_rows_accum"] / n_launch
        qb = 2.0 * d if half else 4.0 * d  # bytes of one query as the scan reads it: fp16 copy (round 4) or f32
        mfma = half and st.get("approx_scan", 0) == 2  # ... on the matrix cores, from the index's fp16 copy of the stored rows
        rb = 2.0 * d + 8.0 if mfma else 4.0 * d        # bytes of one stored row as the scan reads it (fp16 copy + {|x|^2, 1 / scale})
        if mfma and st.get("row_copy_bytes", 0) < rows_local * d:
            rb = 4.0 * d + 8.0                         # ... the f32 rows themselves where the index keeps no copy (scan_mfma_kernel<D, true>)
        l2_bytes = qb * rows_per_launch + rb * stored
        by_design = stored * (rb + 8.0 * T) + 8.0 * rows_per_launch + qb * B / launches_per_batch
        l2_GBps = l2_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms else 0.0
        roof = {"bound": "l2", "achieved": l2_GBps, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": l2_GBps / L2_PEAK_GBS,
                "traffic": None, "bytes_per_launch": l2_bytes,
                "peak_source": "MI355X_MICROARCH.md 'L2 (per XCD)': ~34.5 TB/s aggregate",
                "guide_l2_gather_GBps": L2_GATHER_RANGE_GBS,  # 'Indexed rows: gather into LDS', rows served from the XCD's L2 (a lower bound of one shape)
                "frac_of_guide_l2_gather": l2_GBps / L2_GATHER_PEAK_GBS,
                "hbm_bytes_by_design_per_launch": by_design,
                "hbm_frac": by_design / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if sweep_ms else 0.0,
                "hbm_peak_GBps": HBM_PEAK_GBS,
                "measured_l2_gather_ceiling_GBps": L2_GATHER_MEASURED_GBS,
                "frac_of_measured_l2_gather": [l2_GBps / L2_GATHER_MEASURED_GBS[0], l2_GBps / L2_GATHER_MEASURED_GBS[1]],
                "measured_l2_gather_note": "the bare register gather of the same shape from a 6-MB / 3-MB table with nothing else running "
                                           "(profiles/micro/gather512.hip): the measured ceiling of this access shape, between the guide's gather figure and its L2 peak",
                "query_bytes_per_pair": qb, "row_bytes_per_stored_row": rb,
                "s8d_equivalent_GBps": s8d_GBps, "s8d_bytes_per_launch": bytes_alg,
                "s8d_note": "SURVEY s8(d) numerator / launch time: comparable with the leaf-major sweep, NOT a roofline fraction (a stored row "
                            "is read once for every tree that wants it)",
                "sweep_mode": ("table scan with HALF-WIDTH operands on the matrix cores: the index's fp16 copy of the stored rows (2*d bytes per row, "
                               "in the MFMA operand's order) streamed once per batch window, 16 rows x 16 (row, query) pairs per v_mfma_f32_16x16x32_f16 "
                               "tile, each pair's query halves (2*d bytes) from L2 -> an interval per pair; the intervals pick the candidates and only "
                               "the survivors get the reference's key from the f32 rows (zh_approx.hip)") if mfma else
                              ("table scan with HALF-WIDTH queries: stored rows (f32) streamed once per batch window, each scored against the fp16 "
                               "copy of every query that visits one of its leaves (2*d bytes per pair from L2) -> an interval per pair; the intervals "
                               "pick the candidates and only the survivors get the reference's key (zh_approx.hip)") if half else
                              ("table scan: stored rows streamed once per batch window, each scored against every query that visits one of "
                               "its leaves (queries from L2)")}
        roof.update(common)

    out = {
        "qps": B * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps,
        "config": {"workload": f"{name}: {wl['desc']}", "rows_total": n_total, "rows_per_gpu": rows_local, "dim": d,
                   "metric": wl["metric"] + ("(parity key)" if wl["metric"] == "cosine" else ""), "top_k": k, "batch": B,
                   "max_node_size": M_shard, "num_trees": T,
                   "parallelism": (f"rows sharded x{S}, queries replicated, one RCCL all-gather of the packed top-k + merge per batch"
                                   if S > 1 else "1 GPU")},
        "roofline": roof,
        "stage_ms_per_batch": {s_: st["ms_" + s_] / n_timed for s_ in ("hash", "walk", "sweep", "select", "final")},
        "host_loop": ("look-ahead (windows w+1..w+%d begun before w is finished)" % LA) if look["ahead"] else "begin + finish back to back",
        "visits_per_batch": st["visits"] / max(st["window_batches"], 1), "rows_scored_per_batch": st["rows_scored"] / max(st["window_batches"], 1),
        "window_batches": WIN,
        "setup_s": {"fill": t_fill, "build": t_build},
    }
    if st.get("approx_batches_accum"):
        # the table scan read half-width (fp16) copies of the queries; what the intervals left for the exact passes (last internal batch)
        out["half_width_scan"] = {"batches": st["approx_batches_accum"], "redone_by_the_f32_scan": st["approx_fallbacks_accum"],
                                  "last_overflow_bits": st["approx_last_overflow"], "list_entries_per_query": st["approx_list_entries"] / max(st["batch"], 1),
                                  "survivors_scored_exactly_per_query": st["approx_survivors"] / max(st["batch"], 1),
                                  "visits_ranked_exactly": st["approx_exact_visits"],
                                  # device memory the index holds for the fp16 copy of its stored rows (0: the VALU kernel on the f32 rows)
                                  "fp16_row_copy_bytes": st.get("row_copy_bytes", 0),
                                  # the matrix-core scan: a tile column is a DISTINCT query of a wave's pairs -- columns per pair of the last batch (a 1-in-64
                                  # sample of the waves; 1.0 = nothing shared: every pair fetches its own copy of its query)
                                  "columns_per_pair": (st["approx_columns"] / st["approx_column_pairs"]) if st.get("approx_column_pairs") else None,
                                  # the order the scan keeps its view of the rows in (0 id order, 2 / 3 sorted by the leaves of that many trees; measured, best kept)
                                  "scan_order_keys": st.get("scan_order_keys"), "scan_order_share": st.get("scan_order_share_permille", 0) / 1000.0}
    if lat_store["ms"]:
        ls = sorted(lat_store["ms"])
        lr = sorted(lat_store["ready_ms"])
        out["latency_ms"] = {"p50_window_submit_to_results": lr[len(lr) // 2], "max_window_submit_to_results": lr[-1],
                             "p50_window_submit_to_host_incl_slot_reuse": ls[len(ls) // 2], "max_window_submit_to_host_incl_slot_reuse": ls[-1],
                             "windows": len(ls), "batches_per_window": WIN, "windows_in_flight": NS,
                             "note": "hipEvent spans on the window's own stream from an event recorded before begin() (stream idle = submission time).  "
                                     "submit_to_results: to an event right behind the window's LAST kernel (its top-k complete in device memory; the D2H "
                                     "copy of a window is ~1 MB more) -- the window's latency.  submit_to_host_incl_slot_reuse (ADVICE r4: what round 4 "
                                     "called submit_to_host): to an event behind the D2H copies, which this loop only queues when the slot is REUSED, "
                                     "`windows_in_flight` windows later, after zh_search_wait -- it includes that delay and is not the window's latency.  "
                                     "No batch of a window completes before the whole window"}
    if group is not None:
        out["ranks_seen"] = group.ranks()  # ncclCommCount of the communicator the exchange ran on
        # every rank must have run the same host loop on the same shapes, or the collectives were not the same sequence:
        # (ranks RCCL connected, world, window, windows in flight, look-ahead, batch, top_k; the sweep kind is a rank's own business)
        mine = [out["ranks_seen"], env.world if (env.world > 1 and S == env.world) else 1, WIN, NS, int(look["ahead"]), B, k]
        out["preflight"] = {"ranks_seen": mine[0], "expected_ranks": mine[1], "ok": mine[0] == mine[1]}
        if env.dist and S == env.world:
            allv = [None] * env.world
            env.dist.all_gather_object(allv, mine)
            same = all(v == allv[0] for v in allv)
            out["preflight"].update({"all_ranks_same_loop": same, "per_rank": allv if not same else None})
            out["preflight"]["ok"] = out["preflight"]["ok"] and same
            if not same and env.rank == 0:
                print("bench.py preflight: the ranks did NOT run the same loop [ranks_seen, world, window, in_flight, lookahead, B, k]: %s"
                      % allv, file=sys.stderr)
        if not out["preflight"]["ok"] and env.rank == 0:
            print("bench.py preflight FAILED: %s" % out["preflight"], file=sys.stderr)

    if recall and args.recall_queries > 0:
        nq = min(args.recall_queries, B)
        q = queries[-1][:nq]
        brute = recall != "planted"  # "planted": only whether the planted neighbour is returned (no torch brute force in this process yet: DESIGN.md s9, the order effect)

        def search_ids(m):
            if group is None:
                ix.search_batch_device(queries[-1].data_ptr(), B, k, m, r0["ids"].data_ptr(), r0["keys"].data_ptr(),
                                       r0["counts"].data_ptr(), cur.cuda_stream)
            else:
                group.search_batch_device(queries[-1].data_ptr(), B, k, m, r0["ids"].data_ptr(), r0["keys"].data_ptr(),
                                          r0["counts"].data_ptr())
            torch.cuda.synchronize()
            return r0["ids"][:nq].clone()

        rec_metric = metric if wl["metric"] != "cosine" else make_metric(za, "cosine", parity=False)
        got = search_ids(rec_metric)
        got_parity = search_ids(metric) if wl["metric"] == "cosine" and brute else None
        # exact neighbours over the whole (sharded) set: local exact top-k, gathered, re-ranked by true distance
        Xt = _wrap_rows(torch, ix, rows_local, d, dev) if brute else None
        true_local = exact_topk(torch, Xt, q, k, "cosine" if wl["metric"] == "cosine" else "l2") + first_row if brute else None
        if not brute:
            true_ids = None
        elif env.dist and S == env.world:
            all_true = [torch.empty_like(true_local, device="cpu") for _ in range(S)]
            env.dist.all_gather(all_true, true_local.cpu())
            cand = torch.cat(all_true, 1).to(dev)
            dl = _true_dist(torch, Xt, q, cand - first_row, rows_local, wl["metric"]).cpu()
            env.dist.all_reduce(dl, op=env.dist.ReduceOp.MIN)
            true_ids = torch.gather(cand, 1, torch.topk(dl.to(dev), k, dim=1, largest=False).indices)
        else:
            true_ids = true_local

        # the planted neighbour (query = stored row + 0.3 * noise): is it among the returned ids?
        pl = np.array([_planted_row(SEED_Q, (n_batches - 1) * B + b, n_total) for b in range(nq)], dtype=np.int64)
        here = (pl >= first_row) & (pl < first_row + rows_local) if group is None or group.ranks() < S else np.ones(nq, bool)

        def rec(g):
            # one shard of S: only the queries planted next to one of THIS shard's rows have their neighbourhood here (for the others
            # this shard's true top-k are random far rows, which no index finds): recall over those queries
            gt, tt = g.cpu().numpy(), true_ids.cpu().numpy()
            idx = [b for b in range(nq) if here[b]]
            return sum(len(set(gt[b].tolist()) & set(tt[b].tolist())) for b in idx) / (len(idx) * k) if idx else None

        out["planted_neighbour_hit_rate"] = float((got.cpu().numpy()[here] == pl[here, None]).any(1).mean()) if here.any() else None
        out[f"recall_at_{k}"] = rec(got) if brute else None
        out[f"recall_at_{k}_reference_key"] = (rec(got_parity) if got_parity is not None else out[f"recall_at_{k}"]) if brute else None
        if S > 1 and (group is None or group.ranks() < S):
            out["recall_note"] = ("one shard of %d: recall is against this shard's rows only, over the %d of %d sampled queries whose planted "
                                  "neighbour lives in this shard" % (S, int(here.sum()), nq))

    # sanity of the last timed batch as it arrived on the host
    assert (last_host[1] <= k).all() and (last_host[1] > 0).all(), "empty results in the last timed batch"
    # PCIe-inclusive rate through the host-buffer entry point: reported beside, never as, `value`
    if group is None and S == 1 and not args.profile_run:
        # zh_search_batch (HOST pointers, blocking: what the shim's search_batch / Database::query_vectors reaches, core.rs:290-313) with a batch of
        # four API batches' worth of host-resident queries (4096 at cfg3): the library cuts it into windows over two contexts, copies beside the
        # kernels (zh_api.hip, search_host_windows).  PCIe inclusive, pageable memory on both sides.
        nb_h = max(1, min(4, n_batches))
        qh = np.concatenate([queries[(warmup + i) % n_batches].cpu().numpy() for i in range(nb_h)])
        ix.search_batch(qh[:B], k, metric)  # (one API batch first: the windows want the visits per pair of an earlier batch)
        ix.search_batch(qh, k, metric)      # (warm: the second context's scratch)
        hw0 = ix.stats()["host_window_calls_accum"]
        th = time.perf_counter()
        for _ in range(3):
            ix.search_batch(qh, k, metric)
        out["host_buffers_qps"] = 3 * qh.shape[0] / (time.perf_counter() - th)
        out["host_buffers"] = {"queries_per_call": int(qh.shape[0]), "calls": 3, "calls_run_as_windows": ix.stats()["host_window_calls_accum"] - hw0,
                               "note": "blocking zh_search_batch calls back to back, queries and results in pageable host memory"}
        # ... and with SIXTEEN API batches per call: a blocking call pays its pipeline's fill and drain once, so the rate depends on how much it is handed
        qh16 = np.concatenate([queries[(warmup + i) % n_batches].cpu().numpy() for i in range(16)])
        ix.search_batch(qh16, k, metric)
        th = time.perf_counter()
        ix.search_batch(qh16, k, metric)
        out["host_buffers"]["qps_at_16_batches_per_call"] = qh16.shape[0] / (time.perf_counter() - th)
        out["host_buffers"]["queries_per_large_call"] = int(qh16.shape[0])
        # what a latency-bound caller sees: ONE batch through the blocking device-pointer call, results on the host
        lb = []
        for i in range(5):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            blocking(warmup + i % max(steps, 1))
            torch.cuda.synchronize()
            lb.append((time.perf_counter() - tb) * 1e3)
        out.setdefault("latency_ms", {})["p50_blocking_single_batch"] = sorted(lb)[len(lb) // 2]
    return out, ix, group, wl, M_shard


def cpu_baselines(env, ix, wl, M_shard, seconds):
    """The reference CPU path beside the GPU number, two legs on this box's host cores, on a bounded sample of the SAME
    workload: the FULL stored set (rows copied back from the GPU: bit-identical to what the GPU searched) and its forest
    (built by the HIP path, checked equal to the oracle's build in tests/), as many queries as fit `seconds`.
      port-bitexact: the oracle itself (the checker: 256 scalar accumulators + butterfly per row, qsort per leaf, a
                     second scoring pass) -- the algorithm with the GPU's summation order
      port-fast:     the same algorithm written for speed (oracle/zebra_cpu_fast.cpp: SIMD dot / L2 in free summation
                     order, nth_element, no re-score), the closer stand-in for the Rust crate + simsimd
    Vectors and trees are held in RAM (no fjall, no bincode), which favours the CPU relative to the real reference."""
    from oracle import zebra_cpu_fast as zf
    from oracle import zebra_oracle as zo
    import psutil
    d, T, k, n = wl["dim"], wl["T"], wl["k"], len(ix)
    own = None
    if n * d * 4 * 1.5 > psutil.virtual_memory().available:  # a small host: a reduced stored set, stated in `sample`
        n = int(min(n, max(8 * M_shard, 262144)))
        own = env.za.LSHIndex(d, env.za.LSHIndexOptions(M_shard, T), seed=SEED_INDEX, device=env.local_rank)
        own.append_synthetic(n, seed=SEED_ROWS, first_row=0, kind=wl["kind"])
        own.build()
        ix = own
    X = np.empty((n, d), np.float32)
    step = 1 << 20
    for s in range(0, n, step):
        X[s:s + step] = ix.read_rows(s, min(step, n - s))
    g = ix.get_forest()
    if own is not None:
        own.close()
    full = "the full" if own is None else "a REDUCED set of"
    n_plant = wl["rows"] if own is None else n
    cores = zo.num_threads()
    om = {"cosine": zo.COSINE, "l2": zo.L2, "l2sq": zo.L2SQ}[wl["metric"]]
    legs = []

    def leg(kind, search, note):
        bq, done, b0, rows = max(cores, 16), 0, 1 << 20, 0
        search(zo.synth_queries(cores, d, n_plant, b0=b0 - cores, kind=wl["kind"]))  # warm (threads, page faults)
        t0 = time.perf_counter()
        while True:
            r = search(zo.synth_queries(bq, d, n_plant, b0=b0, kind=wl["kind"]))
            rows += r
            done += bq
            b0 += bq
            el = time.perf_counter() - t0
            if el >= seconds or done >= 16384:
                break
            bq = min(bq * 2, 4096)
        legs.append({"value": done / el, "unit": "queries/s", "cores": cores, "kind": "port", "variant": kind,
                     "sample": f"{done} queries against {full} {n} x {d} stored rows (same metric, k={k}, max_node_size={M_shard}, "
                               f"num_trees={T}; {rows / max(done, 1):.0f} rows scored per query), {el:.1f} s on {cores} OpenMP threads; {note}"})

    f = zo.Forest.from_arrays(X, M_shard, g)
    leg("port-bitexact", lambda Q: f.search_batch(Q, k, om, zo.PARITY, nthreads=cores, stats=True)[3].rows_scored,
        "the oracle: GPU summation order emulated, qsort per leaf, second scoring pass")
    ff = zf.FastForest(X, g)
    leg("port-fast", lambda Q: ff.search_batch(Q, k, om, zo.PARITY, nthreads=cores)[3],
        f"free summation order ({zf.isa()} build), nth_element, no second scoring pass")
    return legs


def kernel_sources_sha():
    """content hash of the kernel / host-pipeline sources (zebra_amd/csrc/*.hip, *.h, *.cpp): a PMC summary records the hash it
    was collected at, bench.py the hash it runs at -- equal means the counters describe THESE kernels (.git does not travel
    to the GPU box, a content hash does)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "zebra_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(args, name, S, roof, prof_override=None):
    """roofline.traffic: HBM-side (fabric) bytes per sweep launch from the rocprofv3 PMC passes of THIS command -- separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section), collected with
    --serial-windows at the SAME --window as the timed run (kernels must not overlap for per-kernel counters) and summarised
    under profiles/ by profiles/summarize.py.  bench.py cannot collect counters on itself: it quotes the newest committed
    summary, and only when that summary was taken at the same launch granularity AND window; the commit it was collected at
    travels with it, next to the last commit that touched the kernel sources."""
    prof_name = {("cfg3", 1): "cfg3", ("cfg2", 1): "cfg2", ("cfg4", 8): "cfg4shard", ("cfg5", 8): "cfg5shard", ("refdefault", 1): "refdefault",
                 ("scale64m", 1): "scale64m"}.get((name, S))
    prof_name = prof_override or prof_name
    cands = [args.pmc_summary] if args.pmc_summary else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % prof_name)))
    if not cands or not prof_name or args.rows:
        return
    try:
        pm = json.load(open(cands[-1]))
        kn = roof["kernel"].split(" ...")[0].rstrip(",")
        ent = [v for k_, v in pm.items() if k_.startswith(kn)][0]
        meta = pm.get("_meta", {})
        traffic = ent["hbm_bytes_per_launch"]
        prof = (meta.get("bench_line_under_kernel_trace") or {}).get("roofline") or {}
        if "hbm_bytes_per_launch" not in ent:
            return
        key = "rows_loaded_per_launch" if roof["bound"] == "l2" else "rows_per_launch"
        if prof.get(key) and abs(prof[key] / roof[key] - 1) > 0.10:
            return  # collected at another launch granularity
        if roof["bound"] == "l2" and prof.get("window_batches", 1) != roof["window_batches"]:
            return  # collected at another window (other pairs per launch)
        roof["traffic"] = traffic
        if roof["bound"] == "l2":
            roof["traffic_over_hbm_by_design"] = traffic / roof["hbm_bytes_by_design_per_launch"]
        else:
            roof["traffic_over_algorithmic"] = traffic / roof["bytes_per_launch"]
        if ent.get("l2_hit_rate") is not None:
            roof["l2_hit_rate"] = ent["l2_hit_rate"]
            roof["l2_request_bytes_per_launch"] = ent.get("l2_request_bytes_per_launch")
            if roof["bound"] == "l2" and ent.get("l2_request_bytes_per_launch") and roof.get("launch_ms"):
                # ALL of the kernel's L2 requests (TCC_REQ x 128 B: operands + row -> leaf entries, visit records, stores), over this run's launch
                # time, against the same peak: how close the scan is to what the L2s answer, where `frac` counts the algorithmic operand bytes only
                g = ent["l2_request_bytes_per_launch"] / (roof["launch_ms"] * 1e-3) / 1e9
                roof["l2_requests_GBps"] = g
                roof["l2_requests_frac_of_peak"] = g / roof["peak"]
        roof["traffic_source"] = {"file": os.path.relpath(cands[-1], ROOT), "collected_at_commit": meta.get("commit"),
                                  "collected_at_kernel_sources_sha": meta.get("kernel_sources_sha"),
                                  "kernel_sources_sha_now": kernel_sources_sha(),
                                  "counters_describe_these_kernels": meta.get("kernel_sources_sha") == kernel_sources_sha(),
                                  "note": "committed PMC summary of this command (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with "
                                          "--serial-windows: same window, one window on the GPU at a time; not collected by this run).  FETCH_SIZE "
                                          "counts what L2 fetches from the fabric: for the table scan that includes queries evicted from L2 and "
                                          "re-fetched from the Infinity Cache, which are not HBM reads"}
    except Exception:
        return


LINE_LIMIT = 6000  # bytes; the driver keeps an 8-KB tail of stdout and parses its LAST line (r04's 25.9-KB line did not parse)
PREFILTER_DTYPE = {
    "scan_mfma": "f16 (stored rows AND queries, v_mfma_f32_16x16x32_f16, f32 accumulate) -> an interval per pair; every returned key is the f32 canonical one",
    "sweep128h": "f16 (stored rows AND queries, v_mfma_f32_16x16x32_f16, f32 accumulate) -> an interval per pair; every returned key is the f32 canonical one",
    "sweep128b": "u8 stored rows (an exact copy: every element an integer 0..255) widened to f16 in registers x f16 queries (v_mfma_f32_16x16x32_f16, f32 accumulate) -> an interval per pair; every returned key is the f32 canonical one",
    "scan_approx": "f16 queries x f32 stored rows (v_fma_mix_f32) -> an interval per pair; every returned key is the f32 canonical one",
}


def _clean(o, sig=6):
    """JSON-safe copy: non-finite floats -> None (no NaN / Infinity tokens), floats to `sig` significant digits, numpy scalars -> Python"""
    import math
    if isinstance(o, dict):
        return {str(k): _clean(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v, sig) for v in o]
    if isinstance(o, (bool, type(None), str)):
        return o
    if isinstance(o, (int, np.integer)):
        return int(o)
    if isinstance(o, (float, np.floating)):
        f = float(o)
        if not math.isfinite(f):
            return None
        return float("%.*g" % (sig, f))
    return str(o)


def _pick(dct, keys):
    return {k: dct[k] for k in keys if dct and k in dct and dct[k] is not None}


ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "launch_ms", "launches_per_batch", "window_batches",
             "rows_per_launch", "rows_loaded_per_launch", "unique_row_fraction", "hbm_frac", "hbm_bytes_by_design_per_launch", "traffic_over_hbm_by_design",
             "traffic_over_algorithmic", "l2_hit_rate", "l2_request_bytes_per_launch", "l2_requests_GBps", "l2_requests_frac_of_peak",
             "frac_of_guide_l2_gather", "guide_l2_gather_GBps", "frac_of_measured_l2_gather", "measured_l2_gather_ceiling_GBps",
             "frac_of_measured_gather_ceiling", "query_bytes_per_pair", "row_bytes_per_stored_row", "l2_requests_per_pair", "sector_GBps", "visits_per_launch")


def prefilter_dtype_of(roof):
    kn = (roof or {}).get("kernel", "")
    for k, v in PREFILTER_DTYPE.items():
        if kn.startswith(k):
            return v
    return None


def compact_line(full, limit=LINE_LIMIT):
    """The ONE line the driver parses: the contract's keys + `roofline` + `cpu_baseline` + a few top-level figures, under `limit` bytes,
    strictly JSON (no NaN / Infinity).  Everything else (other_configs' full blocks, notes, recall tables) goes to bench_detail.json and to an
    EARLIER stdout line.  Optional parts are dropped, least important first, should the line still run over."""
    roof = _pick(full.get("roofline") or {}, ROOF_KEYS)
    roof.setdefault("traffic", None)
    ts = (full.get("roofline") or {}).get("traffic_source") or {}
    if ts:
        roof["traffic_file"] = ts.get("file")
        roof["counters_describe_these_kernels"] = ts.get("counters_describe_these_kernels")
    cpu = full.get("cpu_baseline")
    if cpu:
        cpu = dict(cpu)
        cpu["sample"] = str(cpu.get("sample", ""))[:400]
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "prefilter_dtype", "data")}
    cfg = dict(full.get("config") or {})
    cfg["workload"] = str(cfg.get("workload", ""))[:200]
    cfg["parallelism"] = str(cfg.get("parallelism", ""))[:120]
    line["config"] = cfg
    line["roofline"] = roof
    line["cpu_baseline"] = cpu
    # the same workload by an index created AFTER other work churned device memory in this process (the recall run of the bench line's own
    # rows, which follows torch's brute-force GEMMs and every other configuration): the "order effect" of DESIGN.md s9, beside `value`
    line["value_index_created_late_in_process"] = ((full.get("other_configs") or {}).get("recall_iid") or {}).get("queries_per_s_this_gpu")
    line["host_buffers_qps"] = full.get("host_buffers_qps")
    line["host_buffers_queries_per_call"] = (full.get("host_buffers") or {}).get("queries_per_call")
    line["host_buffers_qps_at_16_batches_per_call"] = (full.get("host_buffers") or {}).get("qps_at_16_batches_per_call")
    optional = []  # (key, value), most important first
    optional.append(("stage_ms_per_batch", full.get("stage_ms_per_batch")))
    lat = full.get("latency_ms") or {}
    optional.append(("latency_ms", _pick(lat, ("p50_window_submit_to_results", "p50_window_submit_to_host_incl_slot_reuse", "p50_blocking_single_batch", "batches_per_window", "windows_in_flight")) or None))
    k = (full.get("config") or {}).get("top_k")
    rc = full.get("recall") or {}
    rsum = {"clustered": _pick(rc.get("informative") or {}, (f"recall_at_{k}", "planted_neighbour_hit_rate", "queries_per_s")),
            "iid": _pick(rc.get("bench_data_iid") or {}, (f"recall_at_{k}", "planted_neighbour_hit_rate"))}
    if rc.get("at_this_point"):
        rsum["at_this_point"] = _pick(rc["at_this_point"], ("max_node_size_per_shard", f"recall_at_{k}", f"recall_at_{k}_reference_key", "planted_neighbour_hit_rate"))
    optional.append(("recall", rsum))
    oc = {}
    for key, v in (full.get("other_configs") or {}).items():
        r = v.get("roofline") or {}
        e = {"qps": v.get("queries_per_s_this_gpu"), "ms_per_batch": v.get("ms_per_batch"), "kernel": r.get("kernel"), "bound": r.get("bound"),
             "frac": r.get("frac"), "launch_ms": r.get("launch_ms"), "traffic_ratio": r.get("traffic_over_algorithmic", r.get("traffic_over_hbm_by_design"))}
        if v.get("host_buffers_qps") is not None:
            e["host_buffers_qps"] = v["host_buffers_qps"]
            if (v.get("host_buffers") or {}).get("qps_at_16_batches_per_call") is not None:
                e["host_qps_16_batches_per_call"] = v["host_buffers"]["qps_at_16_batches_per_call"]
        if key == "cfg1_single_query":  # (the single-query configuration: what counts is one blocking call, results on the host)
            e["p50_blocking_single_query_ms"] = (v.get("latency_ms") or {}).get("p50_blocking_single_batch")
        r10 = v.get("recall_at_10")
        if r10:
            e["recall_at_10_clustered_corrected_key"] = (r10.get("clustered_rows") or {}).get("corrected_key")
            e["recall_at_10_reference_key"] = (r10.get("iid_rows") or {}).get("reference_key")
        if "recall_at_100" in v:
            e["recall_at_100"] = v["recall_at_100"]
        cpp = (v.get("half_width_scan") or {}).get("columns_per_pair")
        if cpp is not None:
            e["columns_per_pair"] = cpp
        oc[key] = {a: b for a, b in e.items() if b is not None}
    optional.append(("other_configs", oc or None))
    hw = full.get("half_width_scan")
    optional.append(("half_width_scan", _pick(hw or {}, ("redone_by_the_f32_scan", "list_entries_per_query", "survivors_scored_exactly_per_query",
                                                          "visits_ranked_exactly", "fp16_row_copy_bytes", "columns_per_pair", "scan_order_keys")) or None))
    for key in ("preflight", "ranks_seen", "emulated", "pipelined_batches_in_flight", "timed_span", "detail"):
        optional.append((key, full.get(key)))
    for key, v in optional:
        if v is not None:
            line[key] = v
    line = _clean(line)
    dump = lambda o: json.dumps(o, allow_nan=False, separators=(",", ":"))  # noqa: E731
    text = dump(line)
    for key, _ in reversed(optional):  # still too long: drop optional parts, least important first
        if len(text) <= limit:
            break
        if line.pop(key, None) is not None:
            line["dropped_for_length"] = line.get("dropped_for_length", []) + [key]
            text = dump(line)
    if len(text) > limit:  # (cannot happen with the fields above; never print an unparsable line)
        line = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                          "vs_baseline", "dtype", "data")}
        line["error"] = "bench line over %d bytes after dropping every optional part" % limit
        text = dump(line)
    return text


def write_detail(full):
    """everything the run measured, as bench_detail.json at the repo root and under gpurun_out/ (which travels back from the GPU box)"""
    doc = json.dumps(_clean(full, sig=9), allow_nan=False)
    where = []
    for dname in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(dname, exist_ok=True)
            with open(os.path.join(dname, "bench_detail.json"), "w") as f:
                f.write(doc + "\n")
            where.append(os.path.relpath(os.path.join(dname, "bench_detail.json"), ROOT))
        except OSError:
            pass
    return doc, where


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher: be one
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    env = Env(args)
    torch = env.torch
    emu = args.emulate_ranks if env.world == 1 else 0
    S = emu or env.world
    name = args.workload or ("cfg3" if S == 1 else "scale64m")
    kind = {"clustered": 2, "clustered-shuffled": 3}.get(args.data) if WORKLOADS[name]["kind"] == 0 else None
    will_run_others = S == 1 and env.world == 1 and not args.no_other_configs and name == "cfg3" and not args.rows
    # The order of things ("order effect"): an index whose device buffers are allocated after torch's brute-force recall ran in this
    # process scans up to a third slower (the matrix-core scan; not clocks, not the TLB, not the L2 hit rate: profiles/r05_order_effect*.txt).  So the
    # TIMED runs come first -- the bench line's own workload, then the other configurations, each with only the planted-neighbour check -- and the
    # brute-force recall runs (which rebuild their small / medium indexes) after all of them.  --no-other-configs: the recall in place, as before.
    main_recall = False if (args.no_recall or args.no_main_recall) else ("planted" if will_run_others else True)
    res, ix, group, wl, M_shard = run_workload(env, name, S, env.rank if not emu else 0, args.steps, args.warmup, exchange=S > 1,
                                               recall=main_recall, M_override=args.max_node_size,
                                               rows_override=args.rows, batch_override=args.batch, kind_override=kind)
    if args.data == "iid":  # (the committed counters are those of the iid command)
        pmc_traffic(args, name, S, res["roofline"])
    cpu = None
    if env.rank == 0 and S == 1 and args.cpu_seconds > 0:
        cpu = cpu_baselines(env, ix, wl, M_shard, args.cpu_seconds)
    if group is not None:
        group.close()
    ix.close()
    del ix
    torch.cuda.empty_cache()

    other = None
    if will_run_others:
        other = {}
        only = args.only_other.split(",") if args.only_other else None
        todo = OTHER_CONFIGS if not only else sorted((c for c in OTHER_CONFIGS if c[0] in only), key=lambda c: only.index(c[0]))
        later = []  # the cosine configurations' recall runs, after every timed run
        for key, wname, shards, steps, *win in todo:
            if args.sleep_before_other > 0:
                torch.cuda.synchronize()
                time.sleep(args.sleep_before_other)
            is_cos = WORKLOADS[wname]["metric"] == "cosine" and not args.no_recall
            in_place = is_cos and wname == "scale64m"  # (64M rows are not built twice: the last timed configuration, its scan the VALU kernel)
            r, ix2, g2, _, _ = run_workload(env, wname, shards, 0, steps, 2, exchange=False, recall=(True if in_place else ("planted" if is_cos else False)),
                                            window_override=win[0] if win else None, sweep_override=OTHER_SWEEP_MODE.get(key))
            ix2.close()
            del ix2
            torch.cuda.empty_cache()
            if is_cos:
                later.append((key, wname, shards, win, r if in_place else None))
            pmc_traffic(args, wname, shards, r["roofline"], prof_override="cfg3leaf" if key == "cfg3_leaf_major_f32" else None)
            other[key] = {"queries_per_s_this_gpu": r["qps"], "ms_per_batch": r["ms_per_step"], "steps": r["steps"],
                          "config": r["config"], "roofline": {kk: r["roofline"][kk] for kk in
                                                              ("bound", "kernel", "achieved", "peak", "frac", "launch_ms", "bytes_per_launch", "unique_row_fraction", "launches_per_batch",
                                                               "rows_per_launch", "rows_loaded_per_launch", "window_batches", "sweep_mode", "hbm_bytes_by_design_per_launch",
                                                               "hbm_frac", "s8d_equivalent_GBps", "measured_gather_ceiling_GBps", "frac_of_measured_gather_ceiling",
                                                               "measured_l2_gather_ceiling_GBps", "frac_of_measured_l2_gather", "query_bytes_per_pair",
                                                               "traffic", "traffic_over_hbm_by_design", "traffic_over_algorithmic", "traffic_source", "l2_hit_rate", "l2_request_bytes_per_launch",
                                                               "l2_requests_GBps", "l2_requests_frac_of_peak", "row_bytes_per_stored_row",
                                                               "visits_per_launch", "sector_GBps", "exact_rows_per_launch", "exact_visits_per_launch", "prefilter_fallbacks", "note")
                                                              if kk in r["roofline"]},
                          "stage_ms_per_batch": r["stage_ms_per_batch"], "host_loop": r["host_loop"], "visits_per_batch": r["visits_per_batch"],
                          "rows_scored_per_batch": r["rows_scored_per_batch"], "latency_ms": r.get("latency_ms"),
                          "host_buffers_qps": r.get("host_buffers_qps"), "host_buffers": r.get("host_buffers"),
                          "half_width_scan": r.get("half_width_scan"), "planted_neighbour_hit_rate": r.get("planted_neighbour_hit_rate")}
        # ---- the recall runs (torch brute force) ----
        for key, wname, shards, win, r_iid in later:
            # BASELINE.json's metric is "queries/sec + recall@10" on the 768-d cosine top-10 shape: recall against GPU brute force under
            # BOTH keys -- the corrected one (cosine distance) and the reference's literal one (distance.rs:23-25 sorts by similarity:
            # ~0 by construction) -- on the bench's iid rows (where only the planted neighbour can be found) and on
            # clustered rows (128 consecutive rows share a centre: informative), same shape, same index options
            kq = WORKLOADS[wname]["k"]
            runs = {}
            for tag, kd in (("iid", None), ("clustered", 2)):
                if tag == "iid" and r_iid is not None:
                    runs[tag] = r_iid
                    continue
                rr, ix3, _, _, _ = run_workload(env, wname, shards, 0, 2, 1, exchange=False, recall=True, kind_override=kd,
                                                window_override=win[0] if win else None)
                ix3.close()
                del ix3
                torch.cuda.empty_cache()
                runs[tag] = rr
            r, rcl = runs["iid"], runs["clustered"]
            other[key]["recall_at_10"] = {
                "top_k": kq,
                "iid_rows": {"corrected_key": r.get(f"recall_at_{kq}"), "reference_key": r.get(f"recall_at_{kq}_reference_key"),
                             "planted_neighbour_hit_rate": r.get("planted_neighbour_hit_rate")},
                "clustered_rows": {"corrected_key": rcl.get(f"recall_at_{kq}"), "reference_key": rcl.get(f"recall_at_{kq}_reference_key"),
                                   "planted_neighbour_hit_rate": rcl.get("planted_neighbour_hit_rate"),
                                   "queries_per_s_this_gpu": rcl["qps"]},
                "against": "GPU brute force over this GPU's stored rows" + ((" (one shard of %d: over the sampled queries planted in this "
                                                                              "shard -- %s)" % (shards, rcl.get("recall_note"))) if shards > 1 else ""),
                "note": "iid rows in 768-d: every non-planted neighbour is a random row, recall@k ~ rows scanned / rows for ANY index; "
                        "the literal key returns the LEAST similar candidates, so its recall is ~0 by construction"}
        if not only or "recall_clustered" in only:
            # clustered rows (128 consecutive rows share a centre), where recall@k against brute force is informative; then the same clusters with
            # their rows scattered over the table (rows inserted in arbitrary order: what the scan's measured row order is for); then the bench
            # line's own iid rows (whose recall run was put off until here)
            for okey, kd, what, st_ in (("recall_clustered", 2, "cfg3 shape, clustered rows", 12),
                                        ("recall_clustered_shuffled", 3, "cfg3 shape, clustered rows scattered over the table", 12),
                                        ("recall_iid", None, "cfg3 (the bench line's rows)", 2)):
                if okey == "recall_iid" and main_recall != "planted":
                    continue
                r, ix2, _, _, _ = run_workload(env, "cfg3", 1, 0, st_, 2, exchange=False, recall=True, kind_override=kd)
                ix2.close()
                del ix2
                torch.cuda.empty_cache()
                other[okey] = {"workload": what, "queries_per_s_this_gpu": r["qps"], "steps": st_,
                               "recall_at_100": r.get(f"recall_at_{wl['k']}"), "planted_neighbour_hit_rate": r.get("planted_neighbour_hit_rate"),
                               "half_width_scan": r.get("half_width_scan"), "roofline": {kk: r["roofline"].get(kk) for kk in ("kernel", "launch_ms", "frac")}}
            ri = other.get("recall_iid")
            if ri:  # the bench line's own recall figures
                res[f"recall_at_{wl['k']}"] = ri["recall_at_100"]
                res[f"recall_at_{wl['k']}_reference_key"] = ri["recall_at_100"]

    if env.rank == 0:
        k = wl["k"]
        out = {
            "metric": "queries/sec", "value": res["qps"], "unit": "queries/s", "n_gpus": env.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            # the arithmetic type of the path's RESULTS: every returned key is the canonical f32 sum (bit-equal to the oracle).  The scan that
            # PICKS the candidates may multiply fp16 copies on the matrix cores: `prefilter_dtype` says so at top level.
            "dtype": "f32",
            "data": {"iid": "synthetic", "clustered": "synthetic (clustered: 128-row clusters)",
                     "clustered-shuffled": "synthetic (clustered, ~150 rows per cluster scattered over the table)"}[args.data],
            "pipelined_batches_in_flight": max(2, args.in_flight) if not args.no_pipeline else 1,
            "timed_span": "queries resident in HBM -> merged top-k in pinned host memory (D2H inside the span)",
            "config": res["config"],
            "latency_ms": res.get("latency_ms"),
            "half_width_scan": res.get("half_width_scan"),
            "recall": None,  # filled below
            f"recall_at_{k}": res.get(f"recall_at_{k}"), f"recall_at_{k}_reference_key": res.get(f"recall_at_{k}_reference_key"),
            "planted_neighbour_hit_rate": res.get("planted_neighbour_hit_rate"),
            "roofline": res["roofline"], "cpu_baseline": cpu[1] if cpu else None, "cpu_baseline_bitexact": cpu[0] if cpu else None,
            "host_buffers_qps": res.get("host_buffers_qps"), "host_buffers": res.get("host_buffers"), "stage_ms_per_batch": res["stage_ms_per_batch"], "host_loop": res["host_loop"],
            "stage_ms_note": "hipEvent spans on each batch's own stream; with batches in flight they overlap other batches' sweeps",
            "setup_s": res["setup_s"],
        }
        if "ranks_seen" in res:
            out["ranks_seen"] = res["ranks_seen"]
        if emu:
            out["emulated"] = f"rank 0 of {emu}, per-rank work only; exchange on a one-rank RCCL communicator"
            out["n_gpus"] = 1
        if S > 1:
            out["series_note"] = ("N > 1 default series = scale64m (64M x 768 cosine top-10, strong scaling); its N = 1 point is "
                                  "other_configs.scale64m_n1 of the N = 1 run (whose `value` is cfg3, the largest single-GPU BASELINE config)")
        if "preflight" in res:
            out["preflight"] = res["preflight"]
        if other is not None:
            out["other_configs"] = other
        # BASELINE's metric is "queries/sec + recall@k": the recall that says something first.  On the bench line's iid ~N(0,1)
        # rows in 768-d every non-planted neighbour is indistinguishable from a random row, so recall@k against brute force is
        # ~(rows scanned / rows) for ANY index -- the planted neighbour (query = stored row + 0.3 noise) is the informative part
        # there; on clustered rows (same shape, same speed) recall@k itself is informative.
        rc = (other or {}).get("recall_clustered") or {}
        cos_literal = wl["metric"] == "cosine"
        out["recall"] = {
            "informative": {"data": "clustered rows (128 consecutive rows share a centre), cfg3 shape", f"recall_at_{k}": rc.get("recall_at_100") if k == 100 else None,
                            "planted_neighbour_hit_rate": rc.get("planted_neighbour_hit_rate"), "queries_per_s": rc.get("queries_per_s_this_gpu")}
            if rc else None,
            "bench_data_iid": {f"recall_at_{k}": res.get(f"recall_at_{k}"), "planted_neighbour_hit_rate": res.get("planted_neighbour_hit_rate"),
                               "note": "iid Gaussian rows: recall@k vs brute force is uninformative by construction (see comment in bench.py); "
                                       "the planted neighbour is what an index can find"},
            "cosine_key_note": ("this workload's metric is cosine with the reference's LITERAL key (distance.rs:23-25 sorts by similarity, SURVEY F4): "
                                f"recall_at_{k}_reference_key is ~0 by construction; recall_at_{k} is measured with the corrected key, same cost")
            if cos_literal else ("the bench line's own workload is L2; the metric's cosine top-10 shape is measured under both keys in "
                                 "other_configs.cfg4_one_of_8_shards.recall_at_10 and other_configs.scale64m_n1.recall_at_10 (corrected key: cosine "
                                 "distance; reference key: distance.rs:23-25 sorts by SIMILARITY, recall ~0 by construction, SURVEY F4)"),
        }
        if S > 1 or emu:
            out["recall"]["at_this_point"] = {"max_node_size_per_shard": M_shard, f"recall_at_{k}": res.get(f"recall_at_{k}"),
                                              f"recall_at_{k}_reference_key": res.get(f"recall_at_{k}_reference_key"),
                                              "planted_neighbour_hit_rate": res.get("planted_neighbour_hit_rate"), "data": args.data,
                                              "note": res.get("recall_note")}
    # RCCL writes a version banner through C stdio: every rank flushes it before the last barrier, so that rank 0's JSON
    # line is the LAST line of the job's stdout
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if env.dist:
        env.dist.barrier()
        env.dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if env.rank == 0:
        out["prefilter_dtype"] = prefilter_dtype_of(out["roofline"])
        doc, where = write_detail(out)
        out["detail"] = ", ".join(where) + "; also the stdout line before this one"
        # the full record first (one line, any length), the COMPACT record last: the driver parses the last line of stdout
        print(json.dumps({"bench_detail": json.loads(doc)}, allow_nan=False), flush=True)
        print(compact_line(out), flush=True)


def _planted_row(seed_q, b, n_rows):
    """row the synthetic query b was planted next to (same formula as the device generator)"""
    M64 = (1 << 64) - 1

    def sm(z):
        z = (z + 0x9E3779B97F4A7C15) & M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)
    return sm(seed_q ^ ((b * 0xA24BAED4963EE407) & M64)) % n_rows


def _wrap_rows(torch, ix, n, d, dev):
    """torch view of the library-owned stored rows (device pointer -> tensor, no copy)"""
    class _Ext:
        pass
    e = _Ext()
    e.__cuda_array_interface__ = {"shape": (n, d), "typestr": "<f4", "data": (ix.rows_device_ptr(), False), "version": 3,
                                  "strides": None}
    return torch.as_tensor(e, device=dev)


def _true_dist(torch, X, q, local_ids, n_local, metric):
    """true distance of candidate ids that live on this shard (inf elsewhere)"""
    valid = (local_ids >= 0) & (local_ids < n_local)
    idx = local_ids.clamp(0, n_local - 1)
    rows = X[idx.reshape(-1)].reshape(idx.shape[0], idx.shape[1], -1)
    if metric == "cosine":
        dots = (rows * q[:, None, :]).sum(2)
        dv = 1.0 - dots / torch.sqrt((rows * rows).sum(2) * (q * q).sum(1)[:, None]).clamp_min(1e-30)
    else:
        dv = ((rows - q[:, None, :]) ** 2).sum(2)
    return torch.where(valid, dv, torch.full_like(dv, float("inf")))


if __name__ == "__main__":
    main()
