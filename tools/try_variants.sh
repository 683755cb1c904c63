#!/bin/bash
# bench.py once per library build under variants_tmp/ (experiments; run on the GPU box):  tools/try_variants.sh [bench args]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for so in "" $ROOT/variants_tmp/*.so; do
  name=$(basename "${so:-default}")
  SPHX_LIB="$so" python3 "$ROOT/bench.py" --no-cpu "$@" > "$ROOT/gpurun_out/var_$name.log" 2>&1
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' "$ROOT/gpurun_out/var_$name.log") $(grep -o '"ms_search": [0-9.]*' "$ROOT/gpurun_out/var_$name.log") $(grep -o '"fallback_queries_last_step": [0-9]*' "$ROOT/gpurun_out/var_$name.log")"
done
