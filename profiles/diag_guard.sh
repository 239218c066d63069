#!/bin/bash
# round 5: a run of the cfg3 bench with ZH_MFMA_DEDUPE=0 ended in a GPU memory access fault once.  Diagnostic build (-DZH_SCAN_GUARD: every index the
# matrix-core scan derives is checked, violations are reported instead of dereferenced), both column modes, the row order off; stops at the first fault.
fmt='import sys,json; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]; h=j.get("half_width_scan") or {}; print(round(j["value"]), "qps  launch_ms", round(r["launch_ms"],3), "columns/pairs", h.get("columns_per_pair"), "host", round(j.get("host_buffers_qps") or 0))'
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
cp gpurun_ab/lib_guard.so zebra_amd/lib/libzebra_hip.so
common="--steps 8 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs"
rc=0
for v in 1 0; do
  echo "== guard build, ZH_MFMA_DEDUPE=$v"
  ZH_NO_ROW_ORDER=1 ZH_MFMA_DEDUPE=$v bash profiles/run_checked.sh gpurun_out/diag_$v python bench.py $common || { rc=1; break; }
  python3 -c "$fmt" gpurun_out/diag_$v.out; grep -i "guard" gpurun_out/diag_$v.err | sort | uniq -c
done
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
exit $rc
