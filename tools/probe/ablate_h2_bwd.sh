#!/usr/bin/env bash
# Ceiling of moving the GRADIENT GEMMs (bf16 x 3: the input-gradient kernels of mlp_fast.hip, every weight gradient of mlp_wgrad_fast.hip) to
# two pieces: a library whose bf16 x 3 kernels issue three of their six products per slab (results wrong by construction, only the time
# is read), against the built one, alternating on one box.   bash tools/probe/ablate_h2_bwd.sh   (build here, run on the GPU box)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"; ROOT="$HERE/../.."; C="$ROOT/votenet_amd/csrc"
if [ "${1:-}" = build ]; then
  mkdir -p "$HERE/lib" /tmp/variant_halfmm
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include"
  /opt/rocm/bin/hipcc $F -DBF3_ABL=32 -c "$C/mlp_fast.hip" -o /tmp/variant_halfmm/a.o &
  /opt/rocm/bin/hipcc $F -DWG_ABL_HALF -c "$C/mlp_wgrad_fast.hip" -o /tmp/variant_halfmm/b.o &
  wait
  objs=$(ls "$C"/obj/*.o | grep -v "/mlp_fast.o" | grep -v "/mlp_wgrad_fast.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_halfmm/a.o /tmp/variant_halfmm/b.o -o "$HERE/lib/libvotenet_halfmm.so"
  echo "built $HERE/lib/libvotenet_halfmm.so"; exit 0
fi
for r in 1 2 3; do for v in "" halfmm; do VARIANT=$v python3 "$HERE/variant_step.py" 2>&1 | tail -1; done; done
