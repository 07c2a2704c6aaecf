#!/bin/bash
# Build libgpnative.so for gfx950 (hipcc cross-compiles without a GPU).
# Usage: gptorch_amd/csrc/build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/../../build/obj"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
pids=()
for f in "$HERE"/*.hip; do
  o="$OBJ/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/gpn_common.h" -nt "$o" ] || [ "$HERE/kernel_fn.h" -nt "$o" ] || [ "$HERE/../../include/gpnative.h" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgpnative.so" "$OBJ"/*.o
echo "built $OUT/libgpnative.so"
# the RCCL adapter of the gpn_dist_comm callback table (a separate library: libgpnative itself links no
# communication runtime)
if [ ! -f "$OUT/libgpnative_rccl.so" ] || [ "$HERE/rccl_adapter.cpp" -nt "$OUT/libgpnative_rccl.so" ] || [ "$HERE/../../include/gpnative.h" -nt "$OUT/libgpnative_rccl.so" ]; then
  $HIPCC -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$HERE/rccl_adapter.cpp" -o "$OUT/libgpnative_rccl.so" -L/opt/rocm/lib -lrccl -lamdhip64
  echo "built $OUT/libgpnative_rccl.so"
fi
