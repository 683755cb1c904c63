#!/bin/bash
# Kernel times + hardware counters of one bench workload, from the current build (run on the GPU box through gpurun):
#   tools/profile_workload.sh <tag> [bench.py args...]
# -> gpurun_out/r03_<tag>_{bench.json,kernel_stats.csv,pmc_per_launch.json} and profiles/latest_<tag>_profile.json
#    (copy the gpurun_out files into profiles/ to commit them; tag = bench.py's workload_tag)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="$1"; shift
"$ROOT/tools/kstats.sh" "r03_$TAG" "$@" > "$ROOT/gpurun_out/r03_${TAG}_kstats.txt" 2>&1
"$ROOT/tools/pmc_profile.sh" "r03_$TAG" "$@" >> "$ROOT/gpurun_out/r03_${TAG}_kstats.txt" 2>&1
cd "$ROOT" && python3 tools/make_search_profile.py "r03_$TAG" "${NPART:-1000000}" 40 "$TAG" > "gpurun_out/r03_${TAG}_profile.txt" 2>&1
cp "profiles/latest_${TAG}_profile.json" "gpurun_out/" 2>/dev/null
[ "$TAG" = polytrope ] && cp profiles/latest_search_profile.json gpurun_out/ 2>/dev/null
head -12 "gpurun_out/r03_${TAG}_kstats.txt"; tail -25 "gpurun_out/r03_${TAG}_profile.txt"
