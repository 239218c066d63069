#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (one pass per counter group, the program directly after
# `--`, as MI355X_MICROARCH.md prescribes):   gpurun -- bash profiles/collect.sh r02 <commit> [configs...]
# then, back in the container:                python profiles/summarize.py r02 <commit>
# Every configuration is its own command, so that a kernel name in a stats file belongs to ONE workload.
TAG=${1:-r02}; COMMIT=${2:-unknown}; shift 2
CFGS=${@:-"cfg3 cfg2 cfg4shard cfg5shard refdefault scale64m hashbig"}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$COMMIT" > $OUT/commit.txt
common="--cpu-seconds 0 --no-recall --no-other-configs"
for c in $CFGS; do
  case $c in
    cfg3)       args="bench.py --steps 5 --warmup 2 $common" ;;
    cfg2)       args="bench.py --workload cfg2 --steps 20 --warmup 3 $common" ;;
    cfg4shard)  args="bench.py --workload cfg4 --emulate-ranks 8 --steps 5 --warmup 2 $common" ;;
    cfg5shard)  args="bench.py --workload cfg5 --emulate-ranks 8 --steps 5 --warmup 2 $common" ;;
    refdefault) args="bench.py --workload refdefault --steps 5 --warmup 2 $common" ;;
    scale64m)   args="bench.py --workload scale64m --steps 3 --warmup 1 $common" ;;
    hashbig)    args="profiles/hash_dense_microbench.py" ;;
    *) echo "unknown config $c"; continue ;;
  esac
  echo "== $c: $args"
  echo "$args" > $OUT/$c.cmd
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/stats -- python3 $args > $OUT/$c.stats.log 2>&1
  tail -1 $OUT/$c.stats.log | cut -c1-300
  [ "$c" = hashbig ] && { rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$c/mfma -- python3 $args > $OUT/$c.mfma.log 2>&1; continue; }
  # counters per kernel need kernels that do not overlap: the blocking call
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$c/fetch -- python3 $args --no-pipeline > $OUT/$c.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$c/write -- python3 $args --no-pipeline > $OUT/$c.write.log 2>&1
  if [ "$c" = refdefault ]; then
    rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$c/sq -- python3 $args --no-pipeline > $OUT/$c.sq.log 2>&1
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$c/mfma -- python3 $args --no-pipeline > $OUT/$c.mfma.log 2>&1
  fi
  # keep what travels back small: the per-dispatch traces are summarised on the box
  python3 profiles/summarize.py --box $OUT/$c $OUT/$c.summary.json > $OUT/$c.summary.log 2>&1
  find $OUT/$c -name '*kernel_trace.csv' -delete; find $OUT/$c -name '*counter_collection.csv' -delete; find $OUT/$c -name '*agent_info.csv' -delete
done
ls -la $OUT
