#!/bin/bash
# Builds libdcq.so (gfx950 only) next to the package: deepcompressor_amd/libdcq.so
# -ffp-contract=off: the reference rounds after every multiply and add; a fused multiply-add changes results.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$HERE/../libdcq.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I"$ROOT/include" -I"$HERE"
       -Wall -Wno-unused-function -Wno-unused-variable ${DCQ_EXTRA_FLAGS:-})
OBJS=()
PIDS=()
mkdir -p "$HERE/build"
for src in "$HERE"/*.hip; do
  obj="$HERE/build/$(basename "${src%.hip}").o"
  if [[ ! -f "$obj" || "$src" -nt "$obj" || "$HERE/dcq_common.cuh" -nt "$obj" || "$ROOT/include/dcq.h" -nt "$obj" ]]; then
    rm -f "$obj"
    "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$obj" &
    PIDS+=($!)
  fi
  OBJS+=("$obj")
done
for pid in "${PIDS[@]:-}"; do
  [[ -n "$pid" ]] && wait "$pid"
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
