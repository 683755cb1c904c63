#!/bin/bash
# Per-kernel hardware counters of `bench.py` (run on the GPU box through gpurun):
#   [KREGEX='blob_.*'] tools/pmc_profile.sh <out-prefix> [bench.py args...]
# One rocprofv3 --pmc pass per counter group (no trace domains beside --kernel-trace), CSV output
# under gpurun_out/pmc_<prefix>/, summarised per kernel and per launch into
# gpurun_out/<prefix>_pmc_per_launch.json by tools/pmc_summary.py.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PREFIX="$1"; shift
OUT="$ROOT/gpurun_out/pmc_$PREFIX"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SPHX_BENCH_SPECIES_LINE=0      # (bench.py's extra with-species run would mix its launches into the statistics)
GROUPS_=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE"
)
g=0
for grp in "${GROUPS_[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace ${KREGEX:+--kernel-include-regex "$KREGEX"} --output-format csv -d "$OUT/g$g" -o pmc -- \
      python3 "$ROOT/bench.py" --no-cpu --device-warmup 0 --steps 3 --warmup 1 "$@" > "$OUT/g$g.log" 2>&1 || echo "group $g failed (see $OUT/g$g.log)"
  g=$((g+1))
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$ROOT/gpurun_out/${PREFIX}_pmc_per_launch.json"
echo "wrote gpurun_out/${PREFIX}_pmc_per_launch.json"
