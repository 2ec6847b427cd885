#!/usr/bin/env bash
# probe: the sa1 sampling kernel's waves at s_setprio 1..3 inside the pipelined train step (-DFPS_PRIO): the kernel's in-step time
# and the step.  GPU box only; builds variant libraries under tools/probe/lib.
set -u
cd "$(dirname "$0")/../.."
C=votenet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-honor-nans -Wno-unused-function"
mkdir -p tools/probe/lib
for pr in ${PRIOS:-1 3}; do
  /opt/rocm/bin/hipcc $FLAGS -DFPS_PRIO=$pr -c $C/fps.hip -o /tmp/fps_prio.o || { echo "build failed"; exit 1; }
  objs=$(ls $C/obj/*.o | grep -v "/fps.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fps_prio.o -o tools/probe/lib/libvotenet_prio$pr.so
done
for rep in 1 2 3; do
  python tools/probe/variant_step.py 2>&1 | tail -2
  for pr in ${PRIOS:-1 3}; do VARIANT=prio$pr python tools/probe/variant_step.py 2>&1 | tail -2; done
done
