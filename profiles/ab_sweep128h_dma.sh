#!/bin/bash
# d = 128 at half width, leaf by leaf (one of 8 cfg5 shards): the register-staged kernel (one tile in flight per wave; default) against the LDS-DMA
# kernel (ZH_S128H_DMA=1: two tiles in flight per wave), same box, alternating:   gpurun -- bash profiles/ab_sweep128h_dma.sh
fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch  launch_ms", round(r["launch_ms"],3), "frac of 8 TB/s", round(r["frac"],3))'
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export ZH_S128H_DMA=1; else unset ZH_S128H_DMA; fi
  echo -n "ZH_S128H_DMA=$v: "
  python bench.py --workload cfg5 --emulate-ranks 8 --steps 8 --cpu-seconds 0 --no-recall --no-other-configs 2>/dev/null | python -c "$fmt"
done
