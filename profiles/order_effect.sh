#!/bin/bash
# DESIGN.md s9 / VERDICT r4 #8: one cfg4 shard measured right after cfg2's recall runs (torch brute force: large f32 GEMMs) took 5.0 ms per scan launch
# instead of 3.4-4.0.  Is it WHERE the next index's buffers land (allocator state) or WHEN it runs (clocks / power after the GEMMs)?  Same process,
# same order, with and without idle seconds between the configurations; the device's clocks / power sampled beside it.
#   gpurun -- bash profiles/order_effect.sh
fmt='import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-2])["bench_detail"]
for k,v in j["other_configs"].items():
    r=v.get("roofline") or {}
    print("   ", k, round(v["queries_per_s_this_gpu"]), "qps  launch_ms", round(r.get("launch_ms",0),3), r.get("kernel"))'
( while true; do rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2 | head -2 | tr "\n" " "; echo; sleep 1; done ) > gpurun_out/order_effect_smi.txt 2>&1 &
SMI=$!

for sl in 0 20; do
  echo "== cfg2 (with its recall runs) then a cfg4 shard, $sl s idle before each:"
  python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-main-recall --only-other cfg2,cfg4_one_of_8_shards --sleep-before-other $sl 2>gpurun_out/order.err | python3 -c "$fmt" || tail -3 gpurun_out/order.err
done
echo "== the cfg4 shard first (no torch GEMMs before it), then cfg2:"
python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-main-recall --only-other cfg4_one_of_8_shards,cfg2 2>gpurun_out/order.err | python3 -c "$fmt" || tail -3 gpurun_out/order.err
echo "== cfg2 then the cfg4 shard, no recall runs at all:"
python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-recall --only-other cfg2,cfg4_one_of_8_shards 2>gpurun_out/order.err | python3 -c "$fmt" || tail -3 gpurun_out/order.err
kill $SMI
wc -l gpurun_out/order_effect_smi.txt
echo "== one window on the GPU at a time (--serial-windows): is the SCAN slower, or what runs beside it?  cfg2 (recall) then cfg4, then cfg4 then cfg2:"
python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-main-recall --serial-windows --only-other cfg2,cfg4_one_of_8_shards 2>gpurun_out/order.err | python3 -c "$fmt" || tail -3 gpurun_out/order.err
python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-main-recall --serial-windows --only-other cfg4_one_of_8_shards,cfg2 2>gpurun_out/order.err | python3 -c "$fmt" || tail -3 gpurun_out/order.err
