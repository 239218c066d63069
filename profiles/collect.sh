#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (one pass per counter group, the program directly after
# `--`, as MI355X_MICROARCH.md prescribes):   gpurun -- bash profiles/collect.sh r02 <commit> [configs...]
# then, back in the container:                python profiles/summarize.py r02 <commit>
# Every configuration is its own command, so that a kernel name in a stats file belongs to ONE workload.
TAG=${1:-r06}; COMMIT=${2:-unknown}; shift 2
CFGS=${@:-"cfg3 cfg3leaf cfg2 cfg4shard cfg5shard refdefault scale64m hashbig"}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
echo "$COMMIT" > $OUT/commit.txt
echo "$CFGS" > $OUT/configs.txt   # summarize.py writes summaries for these only
python3 profiles/summarize.py --sha > $OUT/kernel_sources.sha
# --profile-run: nothing but full windows of the workload (every sweep launch of the process is comparable with bench.py's
# launch_ms); the PMC passes add --serial-windows: the SAME window as the timed run, one window on the GPU at a time (per-kernel
# counters need kernels that do not overlap)
common="--cpu-seconds 0 --no-recall --no-other-configs --profile-run"
for c in $CFGS; do
  case $c in
    cfg3)       args="bench.py --steps 6 --warmup 2 $common" ;;
    cfg3leaf)   args="bench.py --sweep-mode leaf --steps 4 --warmup 2 $common" ;;             # the f32 leaf-major sweep of the bench line's workload (HBM-bound)
    cfg3clu)    args="bench.py --data clustered --steps 6 --warmup 2 $common" ;;            # rows inserted cluster by cluster
    cfg3shuf)   args="bench.py --data clustered-shuffled --steps 6 --warmup 2 $common" ;;   # the same clusters, ids scattered
    cfg2)       args="bench.py --workload cfg2 --steps 20 --warmup 3 $common" ;;
    cfg4shard)  args="bench.py --workload cfg4 --emulate-ranks 8 --steps 6 --warmup 2 $common" ;;
    cfg5shard)  args="bench.py --workload cfg5 --emulate-ranks 8 --steps 6 --warmup 2 $common" ;;
    refdefault) args="bench.py --workload refdefault --steps 6 --warmup 2 $common" ;;
    scale64m)   args="bench.py --workload scale64m --steps 4 --warmup 2 $common" ;;
    hashbig)    args="profiles/hash_dense_microbench.py" ;;
    *) echo "unknown config $c"; continue ;;
  esac
  echo "== $c: $args"
  echo "$args" > $OUT/$c.cmd
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/stats -- python3 $args > $OUT/$c.stats.log 2>&1
  tail -1 $OUT/$c.stats.log | cut -c1-300
  [ "$c" = hashbig ] && { rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$c/mfma -- python3 $args > $OUT/$c.mfma.log 2>&1; continue; }
  # counters per kernel need kernels that do not overlap: one window at a time, same window size
  echo "$args --serial-windows" > $OUT/$c.pmccmd
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$c/fetch -- python3 $args --serial-windows > $OUT/$c.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$c/write -- python3 $args --serial-windows > $OUT/$c.write.log 2>&1
  if [ "$c" != refdefault ]; then  # the sweep's L2 side: requests, hit rate
    rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/$c/l2 -- python3 $args --serial-windows > $OUT/$c.l2.log 2>&1
  fi
  if [ "$c" = refdefault ]; then
    rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$c/sq -- python3 $args --serial-windows > $OUT/$c.sq.log 2>&1
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$c/mfma -- python3 $args --serial-windows > $OUT/$c.mfma.log 2>&1
  fi
  # keep what travels back small: the per-dispatch traces are summarised on the box
  python3 profiles/summarize.py --box $OUT/$c $OUT/$c.summary.json > $OUT/$c.summary.log 2>&1
  find $OUT/$c -name '*kernel_trace.csv' -delete; find $OUT/$c -name '*counter_collection.csv' -delete; find $OUT/$c -name '*agent_info.csv' -delete
done
ls -la $OUT
