fmt='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["value"]), "qps", round(j["ms_per_step"],3), "ms/batch", j["host_loop"][:40], {k: round(v,2) for k,v in j["stage_ms_per_batch"].items()}, (j.get("latency_ms") or {}).get("p50_window_submit_to_host"))'
for rep in 1 2; do
for a in "--window 2 --lookahead on --lookahead-depth 1" "--window 2 --lookahead on --lookahead-depth 2" "--window 4 --lookahead on --lookahead-depth 1" "--window 4 --lookahead on --lookahead-depth 2" "--window 4 --lookahead off --in-flight 3" "--window 8 --lookahead on --lookahead-depth 1"; do
  echo -n "[$a]: "
  python3 bench.py --workload refdefault --steps 16 --warmup 8 --cpu-seconds 0 --no-recall --no-other-configs $a 2>gpurun_out/ab_ref.err | python3 -c "$fmt" || tail -3 gpurun_out/ab_ref.err
done; done
