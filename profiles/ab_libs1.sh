#!/bin/bash
# as ab_libs.sh, every variant ONCE per pass (passes: $PASSES, default 1) -- timing experiments with many variants
#   gpurun -- bash profiles/ab_libs1.sh "<bench args>" name1 name2 ...
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  launch_ms", round(r["launch_ms"],3), r.get("kernel"), {k: round(v,2) for k,v in j["stage_ms_per_batch"].items()})'
args="$1"; shift
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
for rep in $(seq 1 ${PASSES:-1}); do
for v in "$@"; do
  cp gpurun_ab/lib_$v.so zebra_amd/lib/libzebra_hip.so
  echo -n "[$args] $v: "
  timeout -k 10 300 python bench.py $args --cpu-seconds 0 --no-recall --no-other-configs 2>gpurun_out/ab_libs.err | python -c "$fmt" || tail -3 gpurun_out/ab_libs.err
done; done
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
