#!/bin/bash
# Build libgpnative.so for gfx950 (hipcc cross-compiles without a GPU).
# Usage: gptorch_amd/csrc/build.sh [extra hipcc flags]
# Incremental by modification time (an object is rebuilt when its source, a shared header or THIS script is newer);
# GPN_FORCE_REBUILD=1 rebuilds everything (what a fresh checkout does anyway: build/ is not tracked).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/../../build/obj"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
ARCH="${GPN_OFFLOAD_ARCH:-gfx950}"     # the one place the target is named: --offload-arch AND what gpn_arch() reports
FLAGS="--offload-arch=$ARCH -DGPN_ARCH=$ARCH -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
# an object is stale when its source, ANY header of this directory, the public header or this script is newer
stale() {
  local f="$1" o="$2" h
  [ -n "${GPN_FORCE_REBUILD:-}" ] && return 0
  [ ! -f "$o" ] && return 0
  [ "$f" -nt "$o" ] && return 0
  for h in "$HERE"/*.h "$HERE/../../include/gpnative.h" "$HERE/build.sh"; do [ "$h" -nt "$o" ] && return 0; done
  return 1
}
pids=()
for f in "$HERE"/*.hip; do
  o="$OBJ/$(basename "${f%.hip}").o"
  if stale "$f" "$o"; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
# the tools' build: the six sources that carry A/B switches once more with -DGPN_DEBUG_SWITCHES (per-thread variant
# setters, masked streams, the instrumented leaf); everything else is shared with the product library, which exports
# none of it
DBG="$OBJ/dbg"
mkdir -p "$DBG"
for f in "$HERE"/gemm_f64.hip "$HERE"/potrf.hip "$HERE"/profile.hip "$HERE"/leaf16.hip "$HERE"/colpanel.hip "$HERE"/refine.hip "$HERE"/ppotrf.hip; do
  o="$DBG/$(basename "${f%.hip}").o"
  if stale "$f" "$o"; then
    $HIPCC $FLAGS -DGPN_DEBUG_SWITCHES -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=$ARCH -shared -fPIC -o "$OUT/libgpnative.so" "$OBJ"/*.o
echo "built $OUT/libgpnative.so"
shared=()
for o in "$OBJ"/*.o; do
  case "$(basename "$o")" in gemm_f64.o|potrf.o|profile.o|leaf16.o|colpanel.o|refine.o|ppotrf.o) ;; *) shared+=("$o") ;; esac
done
$HIPCC --offload-arch=$ARCH -shared -fPIC -o "$OUT/libgpnative_dbg.so" "${shared[@]}" "$DBG"/*.o
echo "built $OUT/libgpnative_dbg.so"
# the RCCL adapter of the gpn_dist_comm callback table (a separate library: libgpnative itself links no
# communication runtime)
if [ ! -f "$OUT/libgpnative_rccl.so" ] || [ "$HERE/rccl_adapter.cpp" -nt "$OUT/libgpnative_rccl.so" ] || [ "$HERE/../../include/gpnative.h" -nt "$OUT/libgpnative_rccl.so" ]; then
  $HIPCC -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$HERE/rccl_adapter.cpp" -o "$OUT/libgpnative_rccl.so" -L/opt/rocm/lib -lrccl -lamdhip64
  echo "built $OUT/libgpnative_rccl.so"
fi
