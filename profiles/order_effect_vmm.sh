#!/bin/bash
# The order effect (VERDICT r5 #6b): the SECOND / THIRD index a process creates -- its buffers allocated after earlier ones
# were freed -- scans 15-40 % slower.  Is it the physical backing hipMalloc hands out after the pool has been churned?  Same orders with ZH_VMM=1:
# every buffer of 2 MiB and more backed by 256-MiB physical allocations mapped into a reserved range (zh_dev_alloc, zh_api.hip).
#   gpurun -- bash profiles/order_effect_vmm.sh   -> gpurun_out/order_effect_vmm.txt
fmt='import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-2])["bench_detail"]
print("    main cfg3", round(j["value"]), "qps  launch_ms", round(j["roofline"]["launch_ms"],3))
for k,v in j["other_configs"].items():
    r=v.get("roofline") or {}
    print("   ", k, round(v["queries_per_s_this_gpu"]), "qps  launch_ms", round(r.get("launch_ms",0),3), r.get("kernel"))'
out=gpurun_out/order_effect_vmm.txt
: > $out
for vmm in 1 0 1; do
  for order in cfg2,cfg4_one_of_8_shards cfg4_one_of_8_shards,cfg2; do
    echo "== ZH_VMM=$vmm  main cfg3, then $order (no recall runs)" | tee -a $out
    ZH_VMM=$vmm timeout -k 10 400 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-recall --only-other $order 2>gpurun_out/order.err | python3 -c "$fmt" 2>&1 | tee -a $out || tail -3 gpurun_out/order.err | tee -a $out
  done
done
