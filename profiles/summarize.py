#!/usr/bin/env python3
"""Turn the rocprofv3 outputs that gpurun merged into gpurun_out/ into the small summaries committed here.

  python profiles/summarize.py r01 gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write

Commands that produced the inputs (on the MI355X box, one per pass, as MI355X_MICROARCH.md prescribes):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-recall
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-recall
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-recall
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

tag, d_stats, d_fetch, d_write = sys.argv[1:5]
d_mfma = sys.argv[5] if len(sys.argv) > 5 else None
here = os.path.dirname(os.path.abspath(__file__))
stats = glob.glob(os.path.join(d_stats, "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))


def agg(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        a[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(
            (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return {k: {"launches": len(v), "avg_value_KiB": sum(x[0] for x in v) / len(v),
                "avg_ns_under_pmc": sum(x[1] for x in v) / len(v)} for k, v in a.items()}


fetch, write = agg(d_fetch), agg(d_write)
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k), write.get(k)
    out[k] = {"launches": (f or w)["launches"],
              "FETCH_SIZE_KiB_raw": f and f["avg_value_KiB"], "WRITE_SIZE_KiB": w and w["avg_value_KiB"],
              "read_bytes_corrected_x2": f and 2 * f["avg_value_KiB"] * 1024, "write_bytes": w and w["avg_value_KiB"] * 1024,
              "hbm_bytes_per_launch": (2 * f["avg_value_KiB"] * 1024 if f else 0) + (w["avg_value_KiB"] * 1024 if w else 0),
              "avg_ns_under_pmc": (f or w)["avg_ns_under_pmc"]}
json.dump(out, open(os.path.join(here, f"{tag}_pmc_hbm_bytes.json"), "w"), indent=1)

if d_mfma:  # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace ...
    f = glob.glob(os.path.join(d_mfma, "**", "*counter_collection.csv"), recursive=True)[0]
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        a[k]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    m = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in a.items()}
    json.dump(m, open(os.path.join(here, f"{tag}_pmc_mfma.json"), "w"), indent=1)
    for k, v in m.items():
        if "hash_dense" in k or "sweep" in k:
            print("MFMA pass:", k, {c: round(x, 1) for c, x in v.items()})
for k, v in out.items():
    print(f"{k:45s} {v['launches']:4d} launches  {v['hbm_bytes_per_launch'] / 1e9:10.3f} GB/launch  {v['avg_ns_under_pmc'] / 1e6:9.3f} ms")
