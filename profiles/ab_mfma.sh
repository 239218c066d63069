#!/bin/bash
# The half-width table scan on the matrix cores (scan_mfma_kernel: modes 0 and 4) against the VALU kernel on the f32 rows (mode 5), same box:
#   gpurun -- bash profiles/ab_mfma.sh      -> gpurun_out/ab_mfma.txt
: > gpurun_out/ab_mfma.txt
for v in approx approx-valu; do
  for w in cfg3 cfg2 cfg4; do
    extra=""; [ $w = cfg4 ] && extra="--emulate-ranks 8"
    timeout -k 10 200 python bench.py --workload $w $extra --sweep-mode $v --steps 12 --warmup 3 --cpu-seconds 0 --no-recall --no-other-configs > gpurun_out/ab_${v}_$w.json 2>/dev/null
    python3 - >> gpurun_out/ab_mfma.txt <<PY
import json
j=json.loads([l for l in open('gpurun_out/ab_${v}_$w.json') if l.startswith('{')][-1])
h=j.get('half_width_scan') or {}
print('%-12s %-5s %8d QPS  %6.3f ms/batch  %6.3f ms/launch  list entries %6.0f  exact rows %6.0f per query' % ('$v','$w',j['value'], j['ms_per_step'], j['roofline']['launch_ms'], h.get('list_entries_per_query',0), h.get('survivors_scored_exactly_per_query',0)))
PY
  done
done
cat gpurun_out/ab_mfma.txt
