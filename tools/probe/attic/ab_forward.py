"""Same-box A/B of the forward pass (batch statistics, geometry of the next batches prefetched) for a module-level toggle:
   python tools/probe/ab_forward.py pointnet2.ASSEMBLE_FIRST False True"""
import os, sys, time, importlib, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth
modname, attr = sys.argv[1].rsplit(".", 1)
mod = importlib.import_module("votenet_amd." + modname)
vals = [eval(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]])
for rep in range(3):
    for v in vals:
        setattr(mod, attr, v)
        run(9); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("%s = %r: %.3f ms per forward" % (sys.argv[1], v, dt / 60 * 1e3), flush=True)
