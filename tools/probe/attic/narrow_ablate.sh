#!/usr/bin/env bash
# probe: the narrow forward GEMM of sa1 (SRC 3) with phases removed (-DBF3_ABL bits: 1 no MFMAs, 2 no epilogue, 4 no global loads after the
# prologue, 8 no staging: wrong results by construction, only the time is read).  GPU box only; rebuilds the in-tree library THERE.
set -u
cd "$(dirname "$0")/../.."
C=votenet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for abl in 0 1 2 3 8 9 10 11 12 15; do
  /opt/rocm/bin/hipcc $FLAGS -DBF3_ABL=$abl -c $C/mlp_fast.hip -o /tmp/mlp_fast_abl.o 2>/dev/null || { echo "build failed for $abl"; continue; }
  objs=$(ls $C/obj/*.o | grep -v mlp_fast.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mlp_fast_abl.o -o tools/probe/lib/libvotenet_abl.so
  echo "== BF3_ABL=$abl  $(VARIANT=abl timeout 200 python tools/probe/bench_narrow.py 2>&1 | grep -E "fwd narrow|fwd materialised" | tr '\n' ' ')"
done
