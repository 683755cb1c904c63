#!/bin/bash
# A/B of the pass geometry on one box: default library vs variants_tmp/*.so, each with 2 / 3 / 4 persistent workgroups per CU
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export SPHX_BENCH_SPECIES_LINE=0
for so in "" $ROOT/variants_tmp/*.so; do
  for wgs in 2 3 4; do
    name=$(basename "${so:-default}")
    SPHX_LIB="$so" SPHX_BLOB_WGS=$wgs python3 "$ROOT/bench.py" --no-cpu --device-warmup 1 "$@" > "$ROOT/gpurun_out/ab_${name}_$wgs.log" 2>&1
    python3 - "$ROOT/gpurun_out/ab_${name}_$wgs.log" "$name wgs=$wgs" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); p=d["per_pass_ms"]
    print(sys.argv[2], "ms/step %.4f search %.3f prep %.3f dens %.3f pi %.3f visc %.3f"%(d["ms_per_step"],p["ms_search"],p["ms_prep"],p["ms_density"],p["ms_pi"],p["ms_visc"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-400:])
PY
  done
done
