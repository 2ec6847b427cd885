#!/usr/bin/env bash
# Builds votenet_amd/lib/libvotenet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# -ffp-contract=off: distance / interpolation expressions must be evaluated un-fused, left to
# right, exactly as the reference writes them (bit-exact FPS picks and ball-query decisions).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
pids=()
for src in "$HERE"/*.hip; do
  obj="$HERE/obj/$(basename "${src%.hip}").o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/common.h" -nt "$obj" ] || [ "$HERE/mlp_types.h" -nt "$obj" ] || [ "$HERE/../../include/votenet_hip.h" -nt "$obj" ]; then
    extra=""
    # fps.hip: no NaN can occur (distances of finite points); dropping NaN canonicalisation shortens the
    # serial per-round instruction chain.  Infinities (empty bucket boxes) are still honoured.
    [ "$(basename "$src")" = "fps.hip" ] && extra="-fno-honor-nans"
    $HIPCC $FLAGS $extra -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$HERE"/obj/*.o -o "$OUT/libvotenet_hip.so"
echo "built $OUT/libvotenet_hip.so"
