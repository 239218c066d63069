#!/usr/bin/env python3
"""Turns rocprofv3 outputs into the small summaries committed under profiles/.

  on the GPU box (called by profiles/collect.sh, per configuration):
      python3 profiles/summarize.py --box gpurun_out/prof_r02/cfg3 gpurun_out/prof_r02/cfg3.summary.json
  in the container, after gpurun merged gpurun_out/ back:
      python profiles/summarize.py r02 <commit>
  -> profiles/r02_<config>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the (pipelined) command
     profiles/r02_<config>_pmc.json           per kernel: FETCH_SIZE / WRITE_SIZE (and SQ / MFMA counters where collected),
                                              HBM bytes per launch, the command, and the commit they were collected at
Units/corrections: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a
16-B-per-lane coalesced streaming read (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def kname(s):
    return s.split("(")[0].replace("void ", "")


def agg_counters(d):
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return {}
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = kname(r["Kernel_Name"])
        a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        a[k]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, cs in a.items():
        n_counters = max(1, len([c for c in cs if c != "_ns"]))
        out[k] = {c: sum(v) / len(v) for c, v in cs.items() if c != "_ns"}
        out[k]["launches"] = len(cs["_ns"]) // n_counters
        out[k]["avg_ns_under_pmc"] = sum(cs["_ns"]) / len(cs["_ns"])
    return out


def launch_classes(cfg_dir):
    """the sweep / scan kernels' dispatches of the --kernel-trace run, split by grid size: a batch's last launch covers fewer stored rows
    than the full ones, and one mixed average hides a slow launch"""
    fs = glob.glob(os.path.join(cfg_dir, "stats", "**", "*kernel_trace.csv"), recursive=True)
    if not fs:
        return {}
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = kname(r["Kernel_Name"])
        if not any(t in k for t in ("scan_approx", "scan_mfma", "scan_sweep", "sweep_kernel", "sweep128", "prefilter_kernel", "walk_blocked")):
            continue
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        a[k][str(grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, gs in a.items():
        out[k] = {g: {"launches": len(v), "avg_ms": sum(v) / len(v) / 1e6, "min_ms": min(v) / 1e6, "max_ms": max(v) / 1e6} for g, v in sorted(gs.items(), key=lambda x: -len(x[1]))}
    return out


def box(cfg_dir, out_json):
    fetch, write = agg_counters(os.path.join(cfg_dir, "fetch")), agg_counters(os.path.join(cfg_dir, "write"))
    sq, mfma = agg_counters(os.path.join(cfg_dir, "sq")), agg_counters(os.path.join(cfg_dir, "mfma"))
    l2 = agg_counters(os.path.join(cfg_dir, "l2"))
    out = {}
    for k in sorted(set(fetch) | set(write) | set(sq) | set(mfma) | set(l2)):
        f, w = fetch.get(k), write.get(k)
        e = {"launches": (f or w or sq.get(k) or mfma.get(k) or l2.get(k))["launches"]}
        if f:
            e.update(FETCH_SIZE_KiB_raw=f["FETCH_SIZE"], read_bytes_corrected_x2=2 * f["FETCH_SIZE"] * 1024, avg_ns_under_pmc=f["avg_ns_under_pmc"])
        if w:
            e.update(WRITE_SIZE_KiB=w["WRITE_SIZE"], write_bytes=w["WRITE_SIZE"] * 1024)
        if f or w:
            e["hbm_bytes_per_launch"] = (2 * f["FETCH_SIZE"] * 1024 if f else 0) + (w["WRITE_SIZE"] * 1024 if w else 0)
        for src in (sq.get(k), mfma.get(k), l2.get(k)):
            if src:
                e.update({c: v for c, v in src.items() if c not in ("launches",)})
        if l2.get(k) and "TCC_REQ_sum" in l2[k]:  # L2 requests are 128-byte lines
            e["l2_request_bytes_per_launch"] = l2[k]["TCC_REQ_sum"] * 128.0
            hm = l2[k].get("TCC_HIT_sum", 0) + l2[k].get("TCC_MISS_sum", 0)
            e["l2_hit_rate"] = l2[k].get("TCC_HIT_sum", 0) / hm if hm else None
        out[k] = e
    for k, cl in launch_classes(cfg_dir).items():
        out.setdefault(k, {})["launches_by_grid_size"] = cl
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in out.items():
        if "hbm_bytes_per_launch" in v:
            print(f"{k[:60]:60s} {v['launches']:5d} launches {v['hbm_bytes_per_launch'] / 1e9:10.3f} GB/launch {v.get('avg_ns_under_pmc', 0) / 1e6:9.3f} ms")


def container(tag, commit):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(os.path.dirname(here), "gpurun_out", f"prof_{tag}")
    # only the configurations of THIS collection (collect.sh lists them): gpurun_out/ keeps the directories of earlier collections,
    # and re-labelling their summaries with this commit would claim counters for kernels that were not measured at it
    lst = os.path.join(root, "configs.txt")
    only = set(open(lst).read().split()) if os.path.exists(lst) else None
    for cmdf in sorted(glob.glob(os.path.join(root, "*.cmd"))):
        cfg = os.path.basename(cmdf)[:-4]
        if only is not None and cfg not in only:
            continue
        st = glob.glob(os.path.join(root, cfg, "stats", "**", "*kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(here, f"{tag}_{cfg}_kernel_stats.csv"))
        sj = os.path.join(root, f"{cfg}.summary.json")
        pm = json.load(open(sj)) if os.path.exists(sj) else {}
        if cfg == "hashbig":  # only the MFMA pass
            pm = agg_counters(os.path.join(root, cfg, "mfma"))
        log = os.path.join(root, f"{cfg}.stats.log")
        line = None
        if os.path.exists(log):
            for ln in open(log):
                if ln.startswith('{"metric"'):
                    line = json.loads(ln)
        shaf = os.path.join(root, "kernel_sources.sha")
        pmcf = os.path.join(root, f"{cfg}.pmccmd")
        pm["_meta"] = {"commit": commit, "command": "python3 " + open(cmdf).read().strip(),
                       "kernel_sources_sha": open(shaf).read().strip() if os.path.exists(shaf) else None,
                       "pmc_passes": ("python3 " + open(pmcf).read().strip() + "  (one rocprofv3 --pmc pass per counter group)") if os.path.exists(pmcf)
                                     else "the same command + --no-pipeline, one rocprofv3 --pmc pass per counter group",
                       "bench_line_under_kernel_trace": line and {k: line[k] for k in ("value", "ms_per_step", "roofline", "stage_ms_per_batch", "config") if k in line}}
        json.dump(pm, open(os.path.join(here, f"{tag}_{cfg}_pmc.json"), "w"), indent=1)
        print("wrote", cfg)


if __name__ == "__main__":
    if sys.argv[1] == "--sha":  # content hash of the kernel sources (the same function bench.py reports)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        print(bench.kernel_sources_sha())
    elif sys.argv[1] == "--box":
        box(sys.argv[2], sys.argv[3])
    else:
        container(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
