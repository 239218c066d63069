#!/bin/bash
# kernel-trace times of the interval stages (select_tau / select_emit / final_*) of one workload:  gpurun -- bash profiles/final_stage_times.sh "<bench args>"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fst -- python3 bench.py $1 --cpu-seconds 0 --no-recall --no-other-configs --profile-run > gpurun_out/fst.log 2>&1
tail -1 gpurun_out/fst.log | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/fst/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(t in n for t in ("final_", "select_tau", "select_emit", "exact_")):
        print("  %-62s calls %4s avg_ms %8.3f" % (n[:62], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
rm -rf gpurun_out/fst
