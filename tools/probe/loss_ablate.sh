#!/usr/bin/env bash
# probe: where the loss kernel's time goes (-DLOSS_ABL bits: 1 no proposal terms, 2 no seed terms, 4 no dual term).  GPU box only.
set -u
cd "$(dirname "$0")/../.."
C=votenet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
mkdir -p tools/probe/lib
for abl in ${ABLS:-0 1 2 4 3 7}; do
  /opt/rocm/bin/hipcc $FLAGS -DLOSS_ABL=$abl ${EXTRA:-} -c $C/loss.hip -o /tmp/loss_abl.o || { echo "build failed"; exit 1; }
  objs=$(ls $C/obj/*.o | grep -v "/loss.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/loss_abl.o -o tools/probe/lib/libvotenet_loss$abl.so
  echo "LOSS_ABL=$abl $(VARIANT=loss$abl python tools/probe/loss_time.py 2>&1 | tail -1)"
done
