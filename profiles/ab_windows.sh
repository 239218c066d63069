: > gpurun_out/ab3.txt
run() { # label, args...
  l=$1; shift
  timeout -k 10 200 python bench.py "$@" --steps 12 --warmup 3 --cpu-seconds 0 --no-recall --no-other-configs > gpurun_out/ab3_$l.json 2>/dev/null
  python3 - >> gpurun_out/ab3.txt <<PY
import json
j=json.loads([l for l in open('gpurun_out/ab3_$l.json') if l.startswith('{')][-1])
h=j.get('half_width_scan') or {}; lat=j.get('latency_ms') or {}
print('%-22s %8d QPS  %6.3f ms/batch  %6.3f ms/launch  %s  p50 %s' % ('$l',j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['kernel'], lat.get('p50_window_submit_to_host')))
PY
}
run cfg3_auto
run cfg3_w1_auto --window 1
run cfg3_w1_approx --window 1 --sweep-mode approx
run cfg3_w1_scan --window 1 --sweep-mode scan
run cfg3_w3 --window 3
run cfg3_w4 --window 4
run cfg5_auto --workload cfg5 --emulate-ranks 8
run scale64m --workload scale64m
run scale64m_e2 --workload scale64m --emulate-ranks 2
cat gpurun_out/ab3.txt
