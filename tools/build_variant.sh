#!/bin/bash
# A variant build of libsphx.so for A/B timing (tools/try_variants.sh picks up variants_tmp/*.so through SPHX_LIB):
#   tools/build_variant.sh <name> [-DMACRO=value ...]     ->  variants_tmp/<name>.so
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
OBJ="$ROOT/variants_tmp/obj_$NAME"
mkdir -p "$OBJ"
pids=()
for src in "$ROOT"/sph-code_amd/csrc/*.hip; do
  ff=""; [ "$(basename "$src")" = "sphx_knn_group.hip" ] && ff="-mllvm -amdgpu-sched-strategy=max-memory-clause"    # (as sph-code_amd/build.py FILE_FLAGS)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $ff "$@" -c -o "$OBJ/$(basename "$src" .hip).o" "$src" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || { echo "compile failed"; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/variants_tmp/$NAME.so" "$OBJ"/*.o && echo "built variants_tmp/$NAME.so"
