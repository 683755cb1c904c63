#!/bin/bash
# Per-kernel time of `bench.py` under rocprofv3 --kernel-trace --stats (run on the GPU box through gpurun):
#   tools/kstats.sh <out-prefix> [bench.py args...]   ->  gpurun_out/<prefix>_kernel_stats.csv
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PREFIX="$1"; shift
OUT="$ROOT/gpurun_out/ks_$PREFIX"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SPHX_BENCH_SPECIES_LINE=0      # (bench.py's extra with-species run would mix its launches into the statistics)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ks -- \
    python3 "$ROOT/bench.py" --no-cpu --device-warmup 0 --steps 20 --warmup 3 "$@" > "$OUT/run.log" 2>&1
grep '^{' "$OUT/run.log" | tail -1 > "$ROOT/gpurun_out/${PREFIX}_bench.json"
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/gpurun_out/${PREFIX}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s calls %5s avg_us %9.1f tot_ms %8.2f %5s%%"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6,r["Percentage"]))
PY
