#!/bin/bash
# A/B on one box: ZH_GROUP 2 vs 4 builds (gpurun_ab/libzebra_hip_g{2,4}.so) x window 1 / 2 / 4, per workload.
#   gpurun -- bash profiles/ab_window.sh "cfg3" "cfg5 --emulate-ranks 8" ...
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  frac", round(r["frac"],3), " uniq", round(r["unique_row_fraction"],3), " loaded/scored", round(r["rows_loaded_per_launch"]/r["rows_per_launch"],3), " launch_ms", round(r["launch_ms"],3), {k: round(v,2) for k,v in j["stage_ms_per_batch"].items()})'
for wl in "$@"; do
for g in g2 g4; do
  cp gpurun_ab/libzebra_hip_$g.so zebra_amd/lib/libzebra_hip.so
  for w in 1 2 4; do
    echo -n "$wl $g window $w: "
    python bench.py --workload $wl --window $w --steps 24 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs 2>/dev/null | python -c "$fmt"
  done
done; done
cp gpurun_ab/libzebra_hip_g4.so zebra_amd/lib/libzebra_hip.so
