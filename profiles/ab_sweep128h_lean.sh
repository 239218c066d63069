#!/bin/bash
# cfg5, one of 8 shards (125M x 128): sweep128h_lean_kernel + sweep128h_boundary_kernel (round 6, the default) against the round-5 kernel
# (ZH_S128H_KERNEL=r5), alternating on one box
#   gpurun -- bash profiles/ab_sweep128h_lean.sh   -> gpurun_out/ab_sweep128h_lean.txt
out=gpurun_out/ab_sweep128h_lean.txt
: > $out
for v in lean r5 lean r5; do
  if [ $v = r5 ]; then export ZH_S128H_KERNEL=r5; else unset ZH_S128H_KERNEL; fi
  timeout -k 10 400 python bench.py --workload cfg5 --emulate-ranks 8 --steps 8 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs > gpurun_out/ab_s128l_$v.json 2> gpurun_out/ab_s128l_$v.err || { echo "$v FAILED" >> $out; tail -5 gpurun_out/ab_s128l_$v.err >> $out; exit 1; }
  python3 - >> $out <<PY
import json
j=json.loads([l for l in open('gpurun_out/ab_s128l_$v.json') if l.startswith('{')][-1])
h=j.get('half_width_scan') or {}
r=j['roofline']
print('%-5s %8d QPS  %7.3f ms/batch  %6.3f ms/launch  frac %.3f  %s  list entries %6.0f  exact rows %6.0f per query  stages %s' % ('$v', j['value'], j['ms_per_step'], r['launch_ms'], r['frac'], r['kernel'], h.get('list_entries_per_query',0), h.get('survivors_scored_exactly_per_query',0), {k_: round(v_,2) for k_,v_ in j['stage_ms_per_batch'].items()}))
PY
done
cat $out
