#!/bin/bash
# A/B of run-time knobs on ONE box:   gpurun -- bash profiles/ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
# (each further argument is one environment setting, "-" = none), every setting run twice, alternating.
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  frac", round(r["frac"],3), " uniq", round(r["unique_row_fraction"],3), " loaded/scored", round(r["rows_loaded_per_launch"]/r["rows_per_launch"],3), " launch_ms", round(r["launch_ms"],3), {k: round(v,2) for k,v in j["stage_ms_per_batch"].items()})'
args="$1"; shift
for rep in 1 2; do
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo -n "[$args] {$e}: "
  env $e python bench.py $args --cpu-seconds 0 --no-recall --no-other-configs 2>gpurun_out/ab_env.err | python -c "$fmt" || tail -3 gpurun_out/ab_env.err
done; done
