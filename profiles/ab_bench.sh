#!/bin/bash
# A/B on one box: put two builds of the library at gpurun_ab/libzebra_hip_{old,new}.so (git-ignored), then
#   gpurun -- bash profiles/ab_bench.sh
for w in cfg2 cfg3; do
for v in old new old new; do
  cp gpurun_ab/libzebra_hip_$v.so zebra_amd/lib/libzebra_hip.so
  echo -n "$w $v: "
  python bench.py --workload $w --steps 30 --cpu-seconds 0 --no-recall 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],3), {k: round(v,3) for k,v in j['stage_ms_per_batch'].items()})"
done; done
cp gpurun_ab/libzebra_hip_new.so zebra_amd/lib/libzebra_hip.so
