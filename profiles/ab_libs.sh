#!/bin/bash
# A/B of library BUILDS on one box: gpurun_ab/lib_<name>.so (git-ignored; built with `make EXTRA=-D...`), each run twice, alternating:
#   gpurun -- bash profiles/ab_libs.sh "<bench args>" name1 name2 ...
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  launch_ms", round(r["launch_ms"],3), {k: round(v,2) for k,v in j["stage_ms_per_batch"].items()})'
args="$1"; shift
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp gpurun_ab/lib_$v.so zebra_amd/lib/libzebra_hip.so
  echo -n "[$args] $v: "
  python bench.py $args --cpu-seconds 0 --no-recall --no-other-configs 2>gpurun_out/ab_libs.err | python -c "$fmt" || tail -3 gpurun_out/ab_libs.err
done; done
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
