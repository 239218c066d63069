#!/bin/bash
# the multi-GPU series' per-rank work, emulated on ONE GPU (rank 0's shard of an N-rank job, the exchange on a one-rank communicator):
#   gpurun -- bash profiles/series_emulated.sh   -> gpurun_out/series_per_rank_emulated.txt     (N = 1 is the bench line's other_configs.scale64m_n1)
out=gpurun_out/series_per_rank_emulated.txt
echo "# bench.py --workload scale64m --emulate-ranks N --steps 8 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs: rank 0's shard of the 64M x 768 cosine top-10 series (literal key), per-rank work emulated on ONE MI355X (the exchange on a one-rank communicator); a PREDICTION of per-rank work, not a scaling measurement" > $out
for n in 2 4 8; do
  timeout -k 10 500 python bench.py --workload scale64m --emulate-ranks $n --steps 8 --warmup 2 --cpu-seconds 0 --no-recall --no-other-configs 2>gpurun_out/series.err | tail -1 > gpurun_out/series_$n.json || { echo "N=$n FAILED" >> $out; tail -3 gpurun_out/series.err >> $out; continue; }
  python3 - >> $out <<PY
import json
j=json.load(open('gpurun_out/series_$n.json')); r=j['roofline']; h=j.get('half_width_scan') or {}
print('scale64m, one of $n shards: %d qps %.3f ms/batch %s launch_ms %.3f  list entries %d, exact rows %d per query' % (j['value'], j['ms_per_step'], r['kernel'], r['launch_ms'], h.get('list_entries_per_query',0), h.get('survivors_scored_exactly_per_query',0)))
PY
done
cat $out
