#!/bin/bash
# kernel-trace stats (every kernel) of the tree's library on one cfg5 shard -> gpurun_out/kt5_full.txt   [args: extra bench flags]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/kt5_full
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt5_full -- python3 bench.py --workload cfg5 --emulate-ranks 8 --steps 8 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs --profile-run $@ > gpurun_out/kt5_full.log 2>&1
f=$(find gpurun_out/kt5_full -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/kt5_full.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:40]:
    print("%-90s calls %5s avg_ms %8.3f total_ms %9.2f" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY
tail -1 gpurun_out/kt5_full.log | cut -c1-600
rm -rf gpurun_out/kt5_full
