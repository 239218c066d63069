#!/usr/bin/env python3
"""bench.py -- throughput of the LSH bucket-scan + distance hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one batch of B synthetic queries through zh_search_batch_device (hash -> walk -> sweep ->
select -> final), plus, for N > 1, the RCCL all-gather of every rank's top-k and the merge kernel.
Queries and stored vectors are resident in HBM before the timed region starts.

Workloads (BASELINE.json `configs`; index options from BASELINE.md s3):
  N = 1  -> cfg3: 10M x 768 f32, L2 top-100, batch 1024, max_node_size 4096, num_trees 15.  The metric's own
            config (cfg4, 100M x 768) is 307 GB and does not fit one 288 GB GPU, so the largest single-GPU
            config is used, as the bench contract prescribes.
  N > 1  -> the SAME config (the other configs are parity-test cases, not bench lines), strong scaling: the 10M
            rows sharded N ways (global ids), one forest per shard, queries replicated, per-shard
            max_node_size = 4096 / N so that the rows scored per query -- the work of a batch -- is the same at
            every N; one all-gather of every rank's top-k + merge kernel per batch.
            `--workload cfg4` runs the metric's own 100M x 768 cosine config the same way (needs N >= 2;
            per-shard max_node_size 32768 / N = 4096 at N = 8 as in BASELINE.md).
Batches are software-pipelined two deep (zh_search_begin / finish on two contexts and streams): the small
latency-bound kernels of batch i+1 run beside the HBM-bound sweep of batch i.  --no-pipeline times the blocking call.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEED_ROWS, SEED_Q, SEED_INDEX = 0x5EB2A001, 0x5EB2A002, 0x5EB2A003

WORKLOADS = {
    # name: rows (total), dim, metric, k, batch, max_node_size (total budget), trees, kind
    "cfg1": dict(rows=10_000, dim=384, metric="cosine", k=10, batch=1, M=5, T=15, kind=0,
                 desc="10k x 384-d f32 vectors, cosine top-10, single query, reference default options (max_node_size 5)"),
    "cfg2": dict(rows=1_000_000, dim=384, metric="cosine", k=10, batch=256, M=1024, T=15, kind=0,
                 desc="1M x 384-d cosine top-10, query batch=256, LSH index on 1 MI355X"),
    "cfg3": dict(rows=10_000_000, dim=768, metric="l2", k=100, batch=1024, M=4096, T=15, kind=0,
                 desc="10M x 768-d L2 top-100, query batch=1024, 1 MI355X (HBM-bound candidate sweep)"),
    "cfg4": dict(rows=100_000_000, dim=768, metric="cosine", k=10, batch=1024, M=32768, T=15, kind=0,
                 desc="100M x 768-d cosine top-10, batch=1024, vectors sharded across GPUs + RCCL top-k merge"),
    "cfg5": dict(rows=1_000_000_000, dim=128, metric="l2", k=10, batch=4096, M=65536, T=15, kind=1,
                 desc="1B x 128-d SIFT-style L2 top-10, batch=4096, 8 GPUs"),
    "tiny": dict(rows=200_000, dim=768, metric="l2", k=100, batch=256, M=1024, T=15, kind=0,
                 desc="200k x 768-d L2 top-100 (debug)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="override: " + ",".join(WORKLOADS))
    ap.add_argument("--rows", type=int, default=None, help="override total rows (debug)")
    ap.add_argument("--max-node-size", type=int, default=None, help="override max_node_size (debug)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--recall-queries", type=int, default=64)
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--data", choices=["iid", "clustered"], default="iid",
                    help="iid: BASELINE.md's ~N(0,1) rows (the bench line); clustered: 128-row clusters, where recall@k is informative")
    ap.add_argument("--no-pipeline", action="store_true", help="one blocking zh_search_batch_device per step")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="batches in flight when pipelined; default 2 (one sweeping, one in its light phases)")
    ap.add_argument("--debug-normal-priority-sweeps", action="store_true", help="A/B: sweeps on a normal-priority torch stream")
    ap.add_argument("--debug-single-device", action="store_true",
                    help="debug: all ranks on cuda:0, exchange over gloo through host copies (RCCL needs one device per rank)")
    ap.add_argument("--debug-exchange-one-rank", action="store_true",
                    help="debug: on ONE GPU, run the N > 1 exchange step (RCCL all-gather of the packed top-k + merge kernel) "
                         "with a world of one rank inside the pipelined loop")
    ap.add_argument("--pmc-summary", default=None, help="profiles/*_pmc_hbm_bytes.json to take roofline.traffic from")
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


def cpu_baseline(wl, M_shard, seconds, za, torch, device):
    """The oracle (a CPU *port* of the reference algorithm: the Rust crate cannot be built here) timed on
    this box's host cores, on a bounded sample: the same dim / metric / k / max_node_size / num_trees, a
    smaller stored set (so it fits host RAM and builds quickly) and as many 16-query batches as fit the time
    budget.  In the one-leaf-per-tree regime the rows scored per query (~0.68 * M * T) do not depend on the
    number of stored rows, so the per-query cost is representative."""
    from oracle import zebra_oracle as zo
    d, T, k = wl["dim"], wl["T"], wl["k"]
    n_cpu = int(min(wl["rows_local"], max(8 * M_shard, 262144)))
    X = zo.synth_rows(n_cpu, d, kind=wl["kind"])
    # forest of the sample: built by the HIP path (bit-identical to the oracle's build, tests/test_gpu_parity.py)
    ix = za.LSHIndex(d, za.LSHIndexOptions(M_shard, T), seed=SEED_INDEX, device=device)
    ix.append(X)
    ix.build()
    f = zo.Forest.from_arrays(X, M_shard, ix.get_forest())
    ix.close()
    om = {"cosine": zo.COSINE, "l2": zo.L2, "l2sq": zo.L2SQ}[wl["metric"]]
    cores = zo.num_threads()
    bq, done, t0, b0 = 16 * max(1, cores // 8), 0, time.perf_counter(), 0
    f.search_batch(zo.synth_queries(cores, d, n_cpu, kind=wl["kind"]), k, om, zo.PARITY, nthreads=cores)  # warm
    t0 = time.perf_counter()
    rows = 0
    while True:
        Q = zo.synth_queries(bq, d, n_cpu, b0=b0, kind=wl["kind"])
        _, _, _, st = f.search_batch(Q, k, om, zo.PARITY, nthreads=cores, stats=True)
        rows += st.rows_scored
        done += bq
        b0 += bq
        el = time.perf_counter() - t0
        if el >= seconds or done >= 4096:
            break
    return {"value": done / el, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{done} queries against {n_cpu} x {d} stored rows (same metric, k={k}, max_node_size={M_shard}, "
                      f"num_trees={T}; {rows / done:.0f} rows scored per query), {el:.1f} s on {cores} OpenMP threads"}


def main():
    args = parse()
    import torch
    import zebra_amd as za
    from zebra_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with that many ranks" % args.gpus, file=sys.stderr)
            sys.exit(2)
    if args.debug_single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    EX = world > 1 or args.debug_exchange_one_rank  # the exchange step (all-gather + merge) is part of a batch
    if EX:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            for key, val in (("MASTER_PORT", "29642"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(key, val)
        if args.debug_single_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    name = args.workload or "cfg3"
    wl = dict(WORKLOADS[name])
    if args.rows:
        wl["rows"] = args.rows
    if args.max_node_size:
        wl["M"] = args.max_node_size
    if args.data == "clustered" and wl["kind"] == 0:
        wl["kind"] = 2
    S = world
    first_row, rows_local = sharding.shard_rows(wl["rows"], S, rank)
    wl["rows_local"] = rows_local
    M_shard = sharding.per_shard_max_node_size(wl["M"], S, wl["k"]) if S > 1 else wl["M"]
    d, T, k, B = wl["dim"], wl["T"], wl["k"], wl["batch"]
    metric = make_metric(za, wl["metric"], parity=True)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- setup (untimed): synthetic rows on the device, GPU forest build -------------------------------
    t_setup = time.perf_counter()
    ix = za.LSHIndex(d, za.LSHIndexOptions(M_shard, T), seed=SEED_INDEX + rank, device=local_rank,
                     id_base=first_row, reserve_rows=rows_local)
    ix.append_synthetic(rows_local, seed=SEED_ROWS, first_row=first_row, kind=wl["kind"])
    t_fill = time.perf_counter() - t_setup
    ix.build()
    t_build = time.perf_counter() - t_setup - t_fill
    n_total = wl["rows"]

    n_batches = args.steps + args.warmup
    queries = []
    for i in range(n_batches):
        q = torch.empty((B, d), dtype=torch.float32, device=dev)
        za.synth_queries_device(local_rank, q.data_ptr(), n_total, B, d, b0=i * B, seed_rows=SEED_ROWS, seed_q=SEED_Q,
                                kind=wl["kind"])
        queries.append(q)
    # one result buffer per in-flight batch: [ids B*k | keys B*k | counts B] packed so that the N > 1 exchange is ONE
    # all-gather (B*k*16 + B*4 bytes per rank: latency-bound), merged by zh_merge_topk_packed_device on every rank
    W = za.packed_result_words(B, k)

    def make_exchange():
        ex = dict(packed=torch.empty(W, dtype=torch.int64, device=dev))
        ex["ids"], ex["keys"], ex["counts"] = sharding.packed_views(torch, ex["packed"], B, k)
        if EX:
            ex.update(g_packed=torch.empty((S, W), dtype=torch.int64, device=dev), m_ids=torch.empty((B, k), dtype=torch.int64, device=dev),
                      m_keys=torch.empty((B, k), dtype=torch.int64, device=dev), m_counts=torch.empty(B, dtype=torch.int32, device=dev))
        return ex

    def exchange(ex, stream_ptr):
        """the one exchange step of the path: every rank's packed top-k to every rank, then the merge kernel"""
        if args.debug_single_device:  # gloo: through the host
            torch.cuda.synchronize()
            hp, hg = ex["packed"].cpu(), torch.empty(ex["g_packed"].shape, dtype=torch.int64)
            sharding.all_gather_packed(dist, hp, hg)
            ex["g_packed"].copy_(hg)
            torch.cuda.synchronize()
        else:
            sharding.all_gather_packed(dist, ex["packed"], ex["g_packed"])
        za.merge_topk_packed_device(local_rank, S, B, k, ex["g_packed"].data_ptr(), ex["m_ids"].data_ptr(), ex["m_keys"].data_ptr(),
                                    ex["m_counts"].data_ptr(), stream_ptr)

    ex0 = make_exchange()
    ids, keys, counts = ex0["ids"], ex0["keys"], ex0["counts"]

    def step(q):
        ix.search_batch_device(q.data_ptr(), B, k, metric, ids.data_ptr(), keys.data_ptr(), counts.data_ptr(), stream)
        if EX:
            exchange(ex0, stream)

    def barrier():
        if EX:
            dist.barrier()

    # several batches in flight: slot = step mod NS, each slot has its own context, stream and result buffers
    pipelined = not args.no_pipeline
    if pipelined:
        # every batch's sweep back to back on the index's lowest-priority stream; the light work of each slot on a
        # high-priority stream; torch's and RCCL's own streams are normal priority: three hardware-queue pools,
        # so neither the light kernels nor the collectives are queued behind sweep launches
        heavy = ix.sweep_stream()
        if args.debug_normal_priority_sweeps:
            _hs = torch.cuda.Stream(device=dev, priority=0)
            heavy = _hs.cuda_stream
        slots = []
        NS = max(2, args.in_flight) if args.in_flight else 2
        for _ in range(NS):
            sl = dict(ctx=ix.search_context(), stream=torch.cuda.Stream(device=dev, priority=-1))
            if EX:  # the exchange of a slot's batch runs on its own stream: the slot's NEXT batch must not queue behind it
                sl.update(xstream=torch.cuda.Stream(device=dev), ev_final=torch.cuda.Event(), ev_xdone=torch.cuda.Event(), xused=False)
            sl.update(make_exchange())
            slots.append(sl)

        def p_begin(i):
            sl = slots[i % NS]
            sl["ctx"].begin(queries[i].data_ptr(), B, k, metric, sl["stream"].cuda_stream)

        def p_finish(i):
            sl = slots[i % NS]
            if EX and sl["xused"]:  # the packed result buffer is free again once the slot's previous exchange has read it
                sl["stream"].wait_event(sl["ev_xdone"])  # (long done: that batch finished NS batches ago)
            sl["ctx"].finish(sl["ids"].data_ptr(), sl["keys"].data_ptr(), sl["counts"].data_ptr(), heavy)
            if EX:
                sl["ev_final"].record(sl["stream"])
                with torch.cuda.stream(sl["xstream"]):
                    sl["xstream"].wait_event(sl["ev_final"])
                    exchange(sl, sl["xstream"].cuda_stream)
                    sl["ev_xdone"].record(sl["xstream"])
                sl["xused"] = True

        def run(first, n):
            # begin + finish of batch i back to back on slot i % NS: begin first retires batch i-NS of that slot (long
            # done), finish blocks the host only until batch i's own counting pass has run -- beside the sweep of
            # batch i-1, which is still on the GPU -- and leaves batch i's sweep queued behind it
            for i in range(first, first + n):
                p_begin(i)
                p_finish(i)
            for sl in slots:
                sl["ctx"].wait()
                sl["stream"].synchronize()
                if EX:
                    sl["xstream"].synchronize()
    else:
        def run(first, n):
            for i in range(first, first + n):
                step(queries[i])
            torch.cuda.synchronize()

    torch.cuda.synchronize()
    if pipelined:  # every slot allocates its scratch once, whatever --warmup is (not counted as warmup)
        for sl in slots:
            sl["ctx"].begin(queries[0].data_ptr(), B, k, metric, sl["stream"].cuda_stream)
            sl["ctx"].finish(sl["ids"].data_ptr(), sl["keys"].data_ptr(), sl["counts"].data_ptr(), heavy)
            sl["ctx"].wait()
    if args.warmup:
        run(0, args.warmup)
    ix.set_profiling(1)  # hipEvents around every stage, on the stream the kernels run on
    ix.stats(reset=True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if EX:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.debug_single_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ix.stats()

    # ---- untimed: R_unique of the timed batches, recall, sanity of the last result ---------------------
    ix.set_profiling(2)
    uniq = tot = 0
    for i in range(min(args.steps, 4)):
        ix.stats(reset=True)
        ix.search_batch_device(queries[args.warmup + i].data_ptr(), B, k, metric, ids.data_ptr(), keys.data_ptr(),
                               counts.data_ptr(), stream)
        s2 = ix.stats()
        uniq += s2["rows_unique"]
        tot += s2["rows_scored"]
    uniq_frac = uniq / max(tot, 1)
    ix.set_profiling(0)

    # one batch's sweep is issued as several launches of the same kernel (~12 GB each): per-launch figures
    n_launch = max(st["sweep_launches_accum"], 1)
    rows_per_launch = st["sweep_rows_accum"] / n_launch
    sweep_ms = st["ms_sweep"] / n_launch
    launches_per_batch = n_launch / max(st["timed_batches"], 1)
    # algorithmic bytes of one sweep launch (DESIGN.md "Kernels"): every distinct stored row crosses HBM once
    # (4*d bytes), every scored row reads a 4-byte leaf id and writes an 8-byte key, plus the query batch
    bytes_alg = 4.0 * d * rows_per_launch * uniq_frac + 12.0 * rows_per_launch + 4.0 * d * B / launches_per_batch
    bytes_nosharing = (4.0 * d + 12.0) * rows_per_launch
    achieved = bytes_alg / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0

    # roofline.traffic: HBM bytes per sweep launch from the rocprofv3 PMC passes of THIS command (separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled on gfx950), summarised under profiles/ by
    # profiles/summarize.py; bench.py cannot collect counters on itself, so it quotes the committed summary.
    traffic, traffic_src = None, None
    import glob
    cands = [args.pmc_summary] if args.pmc_summary else sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_bytes.json")))
    kname = "sweep_kernel<%d, %d," % (d, 1 if wl["metric"] == "cosine" else 0)  # <D, KIND (0 = L2, 1 = cosine), ...>
    if cands and name == "cfg3" and S == 1 and not args.rows:
        try:
            pm = json.load(open(cands[-1]))
            traffic = [v for k_, v in pm.items() if k_.startswith(kname)][0]["hbm_bytes_per_launch"]
            if abs(traffic / bytes_alg - 1) > 0.5:
                traffic = None  # the committed summary is from a different launch granularity
            traffic_src = os.path.relpath(cands[-1], ROOT)
        except Exception:
            traffic = None

    recall = recall_parity = planted = None
    if not args.no_recall and args.recall_queries > 0:
        nq = min(args.recall_queries, B)
        q = queries[-1][:nq]
        got_parity = None
        rec_metric = metric if wl["metric"] != "cosine" else make_metric(za, "cosine", parity=False)
        def search_ids(m):
            ix.search_batch_device(queries[-1].data_ptr(), B, k, m, ids.data_ptr(), keys.data_ptr(), counts.data_ptr(), stream)
            out = ids.clone()
            if EX:
                exchange(ex0, stream)
                torch.cuda.synchronize()
                out = ex0["m_ids"].clone()
            return out[:nq]

        got = search_ids(rec_metric)
        if wl["metric"] == "cosine":
            got_parity = search_ids(metric)
        # exact neighbours over the whole (sharded) set: local exact top-k, gathered, merged by distance
        Xt = _wrap_rows(torch, ix, rows_local, d, dev)
        true_local = exact_topk(torch, Xt, q, k, wl["metric"]) + first_row
        if S > 1:
            # re-rank the union of the shards' exact top-k by true distance
            xdev = "cpu" if args.debug_single_device else dev
            all_true = [torch.empty_like(true_local, device=xdev) for _ in range(S)]
            dist.all_gather(all_true, true_local.to(xdev))
            cand = torch.cat(all_true, 1).to(dev)
            dl = _true_dist(torch, Xt, q, cand - first_row, rows_local, wl["metric"]).to(xdev)
            dist.all_reduce(dl, op=dist.ReduceOp.MIN)
            dl = dl.to(dev)
            true_ids = torch.gather(cand, 1, torch.topk(dl, k, dim=1, largest=False).indices)
        else:
            true_ids = true_local

        def rec(g):
            hit = 0
            gt, tt = g.cpu().numpy(), true_ids.cpu().numpy()
            for b in range(nq):
                hit += len(set(gt[b].tolist()) & set(tt[b].tolist()))
            return hit / (nq * k)

        # the planted neighbour (query = stored row + 0.3 * noise): is it among the returned ids?
        pl = np.array([_planted_row(SEED_Q, (n_batches - 1) * B + b, n_total) for b in range(nq)], dtype=np.int64)
        planted = float((got.cpu().numpy() == pl[:, None]).any(1).mean())
        recall = rec(got)
        recall_parity = rec(got_parity) if got_parity is not None else recall

    # PCIe-inclusive rate through the host-buffer entry point (zh_search_batch): reported beside, never as, `value`
    host_qps = None
    if S == 1:
        qh = [queries[args.warmup + i % max(args.steps, 1)].cpu().numpy() for i in range(3)]
        ix.search_batch(qh[0], k, metric)
        th = time.perf_counter()
        for q_ in qh:
            ix.search_batch(q_, k, metric)
        host_qps = 3 * B / (time.perf_counter() - th)

    cpu = None
    if rank == 0 and S == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(wl, M_shard, args.cpu_seconds, za, torch, local_rank)

    if rank == 0:
        qps = B * args.steps / elapsed
        out = {
            "metric": "queries/sec", "value": qps, "unit": "queries/s", "n_gpus": S, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if args.data == "iid" else "synthetic (clustered: 128-row clusters)",
            "pipelined_batches_in_flight": (max(2, args.in_flight) if args.in_flight else 2) if pipelined else 1,
            "config": {"workload": f"{name}: {wl['desc']}", "rows_total": n_total, "rows_per_gpu": rows_local,
                       "dim": d, "metric": wl["metric"] + ("(parity key)" if wl["metric"] == "cosine" else ""),
                       "top_k": k, "batch": B, "max_node_size": M_shard, "num_trees": T,
                       "parallelism": f"rows sharded x{S}, queries replicated, all-gather top-k merge" if S > 1 else "1 GPU"},
            f"recall_at_{k}": recall, f"recall_at_{k}_reference_key": recall_parity, "planted_neighbour_hit_rate": planted,
            "roofline": {"bound": "hbm", "kernel": "sweep_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": sweep_ms, "rows_per_launch": rows_per_launch, "unique_row_fraction": uniq_frac,
                         "rows_loaded_per_launch": st["swept_rows_accum"] / n_launch, "launches_per_batch": launches_per_batch,
                         "bytes_per_launch": bytes_alg, "achieved_no_sharing_GBps": bytes_nosharing / (sweep_ms * 1e-3) / 1e9 if sweep_ms else 0.0},
            "cpu_baseline": cpu, "host_buffers_qps": host_qps,
            "stage_ms_per_batch": {s_: st["ms_" + s_] / max(st["timed_batches"], 1) for s_ in ("hash", "walk", "sweep", "select", "final")},
            "stage_ms_note": "hipEvent spans on each batch's own stream; with batches in flight they overlap other batches' sweeps",
            "setup_s": {"fill": t_fill, "build": t_build},
        }
    # RCCL writes a version banner through C stdio: every rank flushes it before the last barrier, so that rank 0's JSON
    # line is the LAST line of the job's stdout
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if EX:
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)


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
