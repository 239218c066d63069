#!/bin/bash
# round 5: the scan's row order (leaf in tree 0, leaf in tree 1, leaf in tree 2, id) against id order (ZH_NO_ROW_ORDER=1), by data; stops at a GPU fault
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; h=j.get("half_width_scan") or {}; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  launch_ms", round(r["launch_ms"],3), "columns/pairs", h.get("columns_per_pair"), "order", h.get("scan_order_keys"), h.get("scan_order_share"), "host", round(j.get("host_buffers_qps") or 0), "setup", {k: round(v,1) for k,v in j.get("setup_s",{}).items()})'
common="--steps 20 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs"
set -o pipefail
chk() { if grep -q "Memory access fault" gpurun_out/ab.err; then echo "GPU memory access fault: stopping"; tail -3 gpurun_out/ab.err; exit 99; fi; }
run() { echo -n "$1: "; shift; "$@" 2>gpurun_out/ab.err | python3 -c "$fmt" || tail -3 gpurun_out/ab.err; chk; }
for data in clustered-shuffled clustered iid; do
  run "cfg3 $data, measured choice" python bench.py $common --data $data
  for o in 0 2 3; do run "cfg3 $data, ZH_ROW_ORDER=$o" env ZH_ROW_ORDER=$o python bench.py $common --data $data; done
done
