#!/bin/bash
# timing experiments whose results are INVALID (the batch then falls back to the f32 scan): the scan launches of the FIRST TWO windows only, per variant
#   gpurun -- bash profiles/ktrace_first.sh <variant> ...
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp zebra_amd/lib/libzebra_hip.so gpurun_ab/_keep.so
for v in "$@"; do
cp gpurun_ab/lib_$v.so zebra_amd/lib/libzebra_hip.so
rm -rf gpurun_out/ktf_$v
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktf_$v -- python3 bench.py --workload cfg3 --steps 6 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs --profile-run > gpurun_out/ktf_$v.log 2>&1
python3 - "$v" <<'PY'
import csv,glob,sys
v=sys.argv[1]
f=glob.glob(f'gpurun_out/ktf_{v}/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'scan_mfma' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows]
print(v, 'scan_mfma launches (ms):', ' '.join('%.3f'%x for x in d[:12]))
PY
rm -rf gpurun_out/ktf_$v
done
cp gpurun_ab/_keep.so zebra_amd/lib/libzebra_hip.so
