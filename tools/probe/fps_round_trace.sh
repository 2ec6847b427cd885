#!/usr/bin/env bash
# Builds the instrumented (-DFPS_TRACE) and the ablated (-DFPS_ABLATE=1..3) variants of fps.hip into tools/probe/lib/ (they
# travel to the GPU box with the snapshot).  On the GPU box: python tools/probe/fps_round_trace.py > profiles/rNN_fps_round_trace.txt
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
mkdir -p "$ROOT/tools/probe/lib"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-honor-nans -Wno-unused-function -shared"
S="$ROOT/votenet_amd/csrc/fps.hip $ROOT/votenet_amd/csrc/common.hip"
/opt/rocm/bin/hipcc $F -DFPS_TRACE $S -o "$ROOT/tools/probe/lib/libfps_trace.so" &
for a in 1 2 3; do /opt/rocm/bin/hipcc $F -DFPS_ABLATE=$a $S -o "$ROOT/tools/probe/lib/libfps_ablate$a.so" & done
wait
echo built
