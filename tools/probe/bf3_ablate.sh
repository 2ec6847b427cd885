#!/usr/bin/env bash
# probe: rebuild mlp_fast.o with extra -D flags per argument (e.g. "-DBF3_ABL=3" -- 1 no MFMAs, 2 no epilogue, 4 no global loads,
# 8 no staging: results wrong by construction, only the time is read -- or "-DBF3_SETS=4 -DBF3_VPM=5") and time the BF3 GEMM shapes.
# GPU box only.
set -u
cd "$(dirname "$0")/../.."
C=votenet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for extra in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $extra -c $C/mlp_fast.hip -o $C/obj/mlp_fast.o 2>/dev/null || { echo "build failed for $extra"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/obj/*.o -o votenet_amd/lib/libvotenet_hip.so
  echo "== $extra"
  timeout 200 python tools/probe/bf3_gemm.py ${NSHAPES:-5} 2>&1 | grep "^("
done
