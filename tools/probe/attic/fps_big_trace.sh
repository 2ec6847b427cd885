#!/bin/bash
# kernel-trace of the config-5 FPS (4 x 80000 -> 2048) for library variants: bash tools/probe/fps_big_trace.sh VARIANT...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fps_big; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fps_big.py <<'P'
import os, sys
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R)
import torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import synth, tf_sampling
x = torch.from_numpy(synth.room_batch(4, 80000, 7)).to("cuda:0")
for _ in range(5):
    tf_sampling.farthest_point_sample(2048, x)
torch.cuda.synchronize()
P
for v in "$@"; do
  VARIANT=$v rocprofv3 --kernel-trace -d $O/$v -o s -- python3 /tmp/fps_big.py > $O/$v.log 2>&1
  db=$(find $O/$v -name "*.db" | head -1)
  echo "== $v"; python3 $R/tools/rocpd_stats.py $db 12 | cut -c1-60,100-150 | tail -8
  rm -rf $O/$v
done
