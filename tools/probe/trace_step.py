"""A dozen train steps with module-level toggles from the environment, for rocprofv3 --kernel-trace (scratch):
   TOGGLES="pointnet2.ASSEMBLE_FIRST=False mlp.COEF_TAIL=True" rocprofv3 --kernel-trace ... -- python3 tools/probe/trace_step.py"""
import os, sys, importlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)  # as bench.py, before torch is imported
import torch
from votenet_amd import loss as VL, model as VM, synth
for t in os.environ.get("TOGGLES", "").split():
    name, val = t.split("=")
    modname, attr = name.rsplit(".", 1)
    setattr(importlib.import_module("votenet_amd." + modname), attr, eval(val))
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
fwd = os.environ.get("WORKLOAD") == "fwd"
import time
for i in range(16 if not fwd else 40):
    if i == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if fwd:
        net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]])
    else:
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
torch.cuda.synchronize()
print("ms per call over the last calls: %.3f" % ((time.perf_counter() - t0) / ((16 if not fwd else 40) - 10) * 1e3))
torch.cuda.synchronize()
