#!/bin/bash
# Instruction mix and wait counters of the table scan (cfg3, --serial-windows: one window on the GPU at a time), one rocprofv3
# --pmc pass per group:   gpurun -- bash profiles/pmc_scan_mix.sh      -> gpurun_out/pmc_scan_mix.txt
# ARGS="<bench.py workload args>" (default: cfg3, 6 steps); KERNEL=<substring of the kernel name> (default: the scan the bench line runs, scan_mfma_kernel since round 4; ARGS="--sweep-mode approx-valu" KERNEL=scan_approx_kernel for the VALU kernel, ZH_NO_APPROX=1
# KERNEL=scan_sweep_kernel for the f32 scan)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export KERNEL=${KERNEL:-scan_mfma_kernel}
OUT=gpurun_out/pmc_mix; rm -rf $OUT; mkdir -p $OUT
args="bench.py ${ARGS:---steps 6 --warmup 2} --cpu-seconds 0 --no-recall --no-other-configs --profile-run --serial-windows"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $args > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY' > gpurun_out/pmc_scan_mix.txt
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_mix/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["KERNEL"] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_ns_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(os.environ["KERNEL"] + ", per launch (mean over launches; --serial-windows, window 2):")
for k in sorted(acc):
    if not k.startswith("_ns_"):
        v = acc[k]
        print("  %-34s %16.0f   (%d launches, %.3f ms under the counters)" % (k, sum(v) / len(v), len(v), sum(acc["_ns_" + k]) / len(v) / 1e6))
PY
cat gpurun_out/pmc_scan_mix.txt
