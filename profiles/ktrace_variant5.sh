#!/bin/bash
# kernel-trace stats of library variants on one cfg5 shard: usage ktrace_variant5.sh <variant> ...   -> gpurun_out/kt5_<variant>.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
for v in "$@"; do
cp gpurun_ab/lib_$v.so zebra_amd/lib/libzebra_hip.so
rm -rf gpurun_out/kt5_$v
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt5_$v -- python3 bench.py --workload cfg5 --emulate-ranks 8 --steps 4 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs > gpurun_out/kt5_$v.log 2>&1
f=$(find gpurun_out/kt5_$v -name '*kernel_stats.csv' | head -1)
echo "== $v" | tee gpurun_out/kt5_$v.txt
python3 - "$f" <<'PY' | tee -a gpurun_out/kt5_$v.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:7]:
    print("%-70s calls %5s avg_ms %8.3f total_ms %9.2f" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
rm -rf gpurun_out/kt5_$v
done
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
