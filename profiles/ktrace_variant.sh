#!/bin/bash
# kernel-trace stats of one lib variant on cfg3: usage ktrace.sh <variant>
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
v=$1
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
cp gpurun_ab/lib_$v.so zebra_amd/lib/libzebra_hip.so
rm -rf gpurun_out/kt_$v
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$v -- python3 bench.py --workload cfg3 --steps 6 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs --profile-run > gpurun_out/kt_$v.log 2>&1
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
f=$(find gpurun_out/kt_$v -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]:
    print("%-60s calls %5s avg_ms %8.3f total_ms %9.2f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
