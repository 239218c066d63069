#!/bin/bash
# round 5: the matrix-core scan with a tile column per DISTINCT query and the scan's rows in tree-0 leaf order (ZH_NO_ROW_ORDER=1: in id order) on one
# box -- iid, clustered and clustered-shuffled rows -- then the split final stage on the literal cosine key (cfg4 shard, 64M rows).  Stops at a GPU fault.
#   gpurun -- bash profiles/ab_dedupe.sh
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; h=j.get("half_width_scan") or {}; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  launch_ms", round(r["launch_ms"],3), {k: round(v,2) for k,v in j["stage_ms_per_batch"].items()}, "exact rows/query", round(h.get("survivors_scored_exactly_per_query",0)), "columns/pairs", h.get("columns_per_pair"), "host", round(j.get("host_buffers_qps") or 0))'
common="--steps 20 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs"
set -o pipefail
chk() { if grep -q "Memory access fault" gpurun_out/ab.err; then echo "GPU memory access fault: stopping"; tail -3 gpurun_out/ab.err; exit 99; fi; }
run() { echo -n "$1: "; shift; "$@" 2>gpurun_out/ab.err | python3 -c "$fmt" || tail -3 gpurun_out/ab.err; chk; }
for data in iid clustered clustered-shuffled; do
  run "cfg3 $data, rows in tree-0 leaf order" python bench.py $common --data $data
  run "cfg3 $data, rows in id order" env ZH_NO_ROW_ORDER=1 python bench.py $common --data $data
done
run "cfg3 iid again" python bench.py $common
run "cfg4 shard (literal cosine key)" python bench.py --workload cfg4 --emulate-ranks 8 --steps 12 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs
run "scale64m N=1 (literal cosine key)" python bench.py --workload scale64m --steps 4 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs
