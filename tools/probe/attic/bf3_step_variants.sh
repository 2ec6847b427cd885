#!/usr/bin/env bash
# probe: the train step with mlp_fast.o rebuilt under extra -D flags, one argument per variant ("" = the committed build).  GPU box only.
set -u
cd "$(dirname "$0")/../.."
C=votenet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for rep in 1 2; do
for extra in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $extra -c $C/mlp_fast.hip -o $C/obj/mlp_fast.o 2>/dev/null || { echo "build failed for $extra"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/obj/*.o -o votenet_amd/lib/libvotenet_hip.so
  echo "== [$extra]"
  timeout 200 python tools/probe/ab_bf3_mask.py 63 2>&1 | grep "^mask"
done
done
