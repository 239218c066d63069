#!/bin/bash
# cfg3's scan launch by how the library's buffers are backed: hipMalloc, or the virtual-memory API with the range aligned to / made of 4 KiB, 2 MiB, 1 GiB
#   gpurun -- bash profiles/vmm_align.sh  -> gpurun_out/vmm_align.txt
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  launch_ms", round(r["launch_ms"],3))'
out=gpurun_out/vmm_align.txt; : > $out
for rep in 1 2; do
for cfg in "ZH_VMM=0" "ZH_VMM=1 ZH_VMM_ALIGN_MB=0 ZH_VMM_CHUNK_MB=256" "ZH_VMM=1 ZH_VMM_ALIGN_MB=2 ZH_VMM_CHUNK_MB=256" "ZH_VMM=1 ZH_VMM_ALIGN_MB=1024 ZH_VMM_CHUNK_MB=1024" "ZH_VMM=1 ZH_VMM_ALIGN_MB=2 ZH_VMM_CHUNK_MB=2"; do
  echo -n "{$cfg}: " | tee -a $out
  env $cfg timeout -k 10 300 python bench.py --steps 12 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs 2>gpurun_out/vmm_align.err | python3 -c "$fmt" 2>&1 | tail -1 | tee -a $out
done; done
