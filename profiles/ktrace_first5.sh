#!/bin/bash
# cfg5 shard, library variants (also ones whose results are INVALID: the batch then falls back): the byte / half lean sweep launches of the FIRST windows
#   gpurun -- bash profiles/ktrace_first5.sh <variant> ...     ("tree" = the tree's own library)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
for v in "$@"; do
[ "$v" = tree ] || cp gpurun_ab/lib_$v.so zebra_amd/lib/libzebra_hip.so
rm -rf gpurun_out/ktf5_$v
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktf5_$v -- python3 bench.py --workload cfg5 --emulate-ranks 8 --steps 4 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs > gpurun_out/ktf5_$v.log 2>&1
python3 - "$v" <<'PY' | tee -a gpurun_out/ktf5.txt
import csv,glob,sys
v=sys.argv[1]
f=glob.glob(f'gpurun_out/ktf5_{v}/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'sweep128' in r['Kernel_Name'] and 'lean' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
full=sorted(x for x in d[:16] if x > 0.6*max(d[:16]))
print(v, rows[0]['Kernel_Name'][:40], 'launches (ms):', ' '.join('%.3f'%x for x in d[:12]), '| median of the full ones %.3f' % full[len(full)//2])
PY
rm -rf gpurun_out/ktf5_$v
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
done
