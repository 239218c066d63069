#!/bin/bash
# The matrix-core scan under -DZH_SCAN_GUARD at FULL sizes (round 6; ADVICE r5 / VERDICT r5 #6a): every index scan_mfma_kernel derives -- global
# (query ids, key slots) and LDS (pair list, column table, regrouped records, columns) -- is checked before it is used; a violation is reported by
# zh_search_wait ("scan guard tripped, bits ...") instead of dereferenced.  The guard library is built here, by profiles/build_variant.sh, from the
# tree's sources:          profiles/build_variant.sh guard "-DZH_SCAN_GUARD"     (in the container, before gpurun)
#   gpurun -- bash profiles/diag_guard.sh      -> gpurun_out/diag_guard.txt ; stops at the first GPU memory access fault (run_checked.sh)
fmt='import sys,json; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]; h=j.get("half_width_scan") or {}; print(round(j["value"]), "qps  launch_ms", round(r["launch_ms"],3), r.get("kernel"), "columns/pairs", h.get("columns_per_pair"), "redone", h.get("redone_by_the_f32_scan"))'
out=gpurun_out/diag_guard.txt
: > $out
[ -f gpurun_ab/lib_guard.so ] || { echo "gpurun_ab/lib_guard.so missing: run profiles/build_variant.sh guard \"-DZH_SCAN_GUARD\" first" | tee -a $out; exit 2; }
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
cp gpurun_ab/lib_guard.so zebra_amd/lib/libzebra_hip.so
common="--steps 8 --warmup 4 --cpu-seconds 0 --no-recall --no-other-configs"
rc=0
i=0
for args in "" "--data clustered" "--data clustered-shuffled" "--workload cfg4 --emulate-ranks 8" "--workload cfg2 --steps 40" "--workload scale64m --steps 4 --warmup 2"; do
  i=$((i+1))
  echo "== guard build: bench.py $args" | tee -a $out
  bash profiles/run_checked.sh gpurun_out/diag_$i python bench.py $common $args >> $out 2>&1 || { rc=1; echo "FAILED" | tee -a $out; break; }
  python3 -c "$fmt" gpurun_out/diag_$i.out | tee -a $out
  echo "guard reports: $(grep -ci 'scan guard tripped' gpurun_out/diag_$i.err)" | tee -a $out
  grep -i "guard" gpurun_out/diag_$i.err | sort | uniq -c | head -5 | tee -a $out
done
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
exit $rc
