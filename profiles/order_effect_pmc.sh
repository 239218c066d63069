#!/bin/bash
# the cfg4 shard's matrix-core scan is a third slower when its index is created after cfg2's recall runs (torch brute force) in the same
# process, with or without idle time in between (profiles/r05_order_effect.txt: not clocks).  Address translation?  UTCL1 (the vector L1's TLB) and L2
# counters of scan_mfma_kernel<768> by grid size, both orders; one window on the GPU at a time (--serial-windows) so that counters are per kernel.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
sum='import csv,sys,glob,collections
d=sys.argv[1]
f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]
        if "scan_mfma_kernel<768" not in k: continue
        acc[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(acc):
    print("   grid", g, {c: (len(v), round(sum(v)/len(v))) for c,v in acc[g].items()})'
for order in "cfg2,cfg4_one_of_8_shards" "cfg4_one_of_8_shards,cfg2"; do
  for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo "$order$pmc" | md5sum | cut -c1-8)
    echo "== order $order, counters $pmc"
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d gpurun_out/oe_$tag -- python3 bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-main-recall --serial-windows --only-other $order > gpurun_out/oe_$tag.log 2>&1
    python3 -c "$sum" gpurun_out/oe_$tag
    python3 - <<PY
import json
try:
    j=json.loads(open("gpurun_out/oe_$tag.log").read().strip().splitlines()[-2])["bench_detail"]
    for k,v in j["other_configs"].items():
        r=v.get("roofline") or {}
        print("   ", k, round(v["queries_per_s_this_gpu"]), "qps  launch_ms", round(r.get("launch_ms",0),3))
except Exception as e:
    print("   (no bench line:", e, ")")
PY
    rm -rf gpurun_out/oe_$tag
  done
done
