#!/bin/bash
# issue-side counters of the byte-row sweep on one cfg5 shard (one rocprofv3 --pmc pass per group, kernel trace only beside it):
#   gpurun -- bash profiles/pmc_cfg5_sq.sh   -> gpurun_out/r06_cfg5shard_sq.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
args="bench.py --workload cfg5 --emulate-ranks 8 --steps 6 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs --profile-run --serial-windows"
out=gpurun_out/pmc5sq; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq -- python3 $args > $out/sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/mix -- python3 $args > $out/mix.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/insts -- python3 $args > $out/insts.log 2>&1
python3 - <<'PY' | tee gpurun_out/r06_cfg5shard_sq.txt
import csv, glob, collections
print("sweep128b_lean_kernel<64, 0>, full launches of one cfg5 shard (201.6M rows each), averages per launch:")
for grp in ("sq", "mix", "insts"):
    fs = glob.glob(f"gpurun_out/pmc5sq/{grp}/*/*counter_collection.csv")
    if not fs:
        print(grp, "no counter file (a counter of the group is not served on this build)")
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "sweep128b_lean" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 3000000:
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for c, d in acc.items():
        v = [sum(x) for x in d.values()]
        print("  %-28s %.4g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $out
