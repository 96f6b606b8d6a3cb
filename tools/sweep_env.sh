#!/bin/bash
# Runs bench.py (no CPU baseline; BENCH_ARGS adds flags) once per argument, each argument a list of environment
# assignments ("" = defaults), and prints the headline and the stage timers of each run.
# Usage on the GPU box: bash tools/sweep_env.sh "" "PRB_SEED_FUSED=0" "PRB_SEED_ROW_SHIFT=-1"
mkdir -p gpurun_out
n=0
for e in "$@"; do
  n=$((n + 1))
  env $e timeout -k 10 400 python bench.py --cpu-queries 0 $BENCH_ARGS > gpurun_out/sweep_$n.json 2> gpurun_out/sweep_$n.err || { tail -5 gpurun_out/sweep_$n.err; exit 1; }
  python - "$n" "$e" <<'PY'
import json, sys
r = json.loads(open(f"gpurun_out/sweep_{sys.argv[1]}.json").read().strip().splitlines()[-1])
s = r["stage_ms_per_step"]
print("[%s]" % sys.argv[2], "q/s %.3f" % r["value"], "ms/step %.0f" % r["ms_per_step"],
      {k: round(s[k]) for k in ("ungapped", "sort", "filter", "gapped_front", "gapped", "gapped_t1", "gapped_t2", "gapped_t3", "gapped_slow", "raccess")}, flush=True)
PY
done
