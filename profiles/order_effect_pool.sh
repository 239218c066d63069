#!/bin/bash
# The order effect (VERDICT r5 #6b) with and without the library's block cache (ZH_POOL, zh_api.hip): the SECOND / THIRD index a
# process creates -- after the first one's buffers were freed -- and the scan's launch time over it.
#   gpurun -- bash profiles/order_effect_pool.sh   -> gpurun_out/order_effect_pool.txt
fmt='import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-2])["bench_detail"]
print("    main cfg3", round(j["value"]), "qps  launch_ms", round(j["roofline"]["launch_ms"],3))
for k,v in j["other_configs"].items():
    r=v.get("roofline") or {}
    print("   ", k, round(v["queries_per_s_this_gpu"]), "qps  launch_ms", round(r.get("launch_ms",0),3), r.get("kernel"))'
out=gpurun_out/order_effect_pool.txt
: > $out
for pool in 1 0 1 0; do
  for order in cfg2,cfg4_one_of_8_shards cfg4_one_of_8_shards,cfg2; do
    echo "== ZH_POOL=$pool  main cfg3, then $order (no recall runs)" | tee -a $out
    ZH_POOL=$pool bash profiles/run_checked.sh gpurun_out/order_pool timeout -k 10 300 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-recall --only-other $order || { echo FAILED | tee -a $out; tail -3 gpurun_out/order_pool.err | tee -a $out; exit 1; }
    python3 -c "$fmt" < gpurun_out/order_pool.out 2>&1 | tee -a $out
  done
done
