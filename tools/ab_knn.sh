#!/bin/bash
# A/B of library builds under variants_tmp/ on one box: headline, cube / loop forms, the Sedov blast under hydro_update's
# sums (escapers: outlier-level variants of the list-mode kernel), 150 steps of the expanding cube (levels 4-5)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for so in "" $ROOT/variants_tmp/*.so; do
  name=$(basename "${so:-default}")
  for rep in 1 2; do
  for wl in "--workload polytrope" "--workload uniform_cube --forms loop" "--workload sedov --dt cfl --steps 50"; do
    SPHX_BENCH_SPECIES_LINE=0 SPHX_LIB="$so" python3 "$ROOT/bench.py" --no-cpu $wl > "$ROOT/gpurun_out/ab_tmp.log" 2>/dev/null
    echo "$name [$wl] $(grep -o '"ms_per_step": [0-9.]*' "$ROOT/gpurun_out/ab_tmp.log") $(grep -o '"ms_search": [0-9.]*' "$ROOT/gpurun_out/ab_tmp.log")"
  done
  done
  SPHX_LIB="$so" python3 "$ROOT/tools/long_run.py" uniform_cube loop 153 2>&1 | sed "s/^/$name cube-long /"
done
