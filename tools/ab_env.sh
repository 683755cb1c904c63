#!/bin/bash
# A/B of one environment switch on one box:  tools/ab_env.sh VAR "bench args" [reps]   (VAR=0 against VAR=1)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
VAR="$1"; ARGS="$2"; REPS="${3:-3}"
for rep in $(seq 1 $REPS); do
  for val in 0 1; do
    env SPHX_BENCH_SPECIES_LINE=0 $VAR=$val python3 "$ROOT/bench.py" --no-cpu $ARGS > "$ROOT/gpurun_out/ab_tmp.log" 2>/dev/null
    echo "$VAR=$val [$ARGS] $(grep -o '"ms_per_step": [0-9.]*' "$ROOT/gpurun_out/ab_tmp.log") $(grep -o '"ms_search": [0-9.]*' "$ROOT/gpurun_out/ab_tmp.log") $(grep -o '"ms_prep": [0-9.]*' "$ROOT/gpurun_out/ab_tmp.log")"
  done
done
