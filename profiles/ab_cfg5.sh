#!/bin/bash
# cfg5, one of 8 shards (125M x 128): the leaf-major sweep at half width (sweep128h_kernel, the default) against the f32 sweep (ZH_NO_LEAF_HALF=1)
#   gpurun -- bash profiles/ab_cfg5.sh   -> gpurun_out/ab_cfg5.txt
: > gpurun_out/ab_cfg5.txt
for v in half f32; do
  if [ $v = f32 ]; then export ZH_NO_LEAF_HALF=1; else unset ZH_NO_LEAF_HALF; fi
  timeout -k 10 400 python bench.py --workload cfg5 --emulate-ranks 8 --steps 8 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs > gpurun_out/ab_cfg5_$v.json 2> gpurun_out/ab_cfg5_$v.err
  python3 - >> gpurun_out/ab_cfg5.txt <<PY
import json
j=json.loads([l for l in open('gpurun_out/ab_cfg5_$v.json') if l.startswith('{')][-1])
h=j.get('half_width_scan') or {}
print('%-5s %8d QPS  %7.3f ms/batch  %6.3f ms/launch  %s  list entries %6.0f  exact rows %6.0f per query  stages %s' % ('$v', j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['kernel'], h.get('list_entries_per_query',0), h.get('survivors_scored_exactly_per_query',0), {k_: round(v_,2) for k_,v_ in j['stage_ms_per_batch'].items()}))
PY
done
cat gpurun_out/ab_cfg5.txt
