"""Same-box A/B of the train step and the forward pass for an integer debug hook of the library:
   python tools/probe/ab_hook.py votenet_debug_xcd_pair 0 1"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import _lib as L, loss as VL, model as VM, synth
hook = getattr(L.lib(), sys.argv[1]); vals = [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def trn(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def fwd(k):
    for i in range(k):
        net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]])
def t(fn, k):
    fn(8); torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter(); fn(k); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k * 1e3; gc.enable(); return dt
for rep in range(3):
    for v in vals:
        hook(v)
        print("%s(%d): train step %.3f ms   forward %.3f ms" % (sys.argv[1], v, t(trn, 40), t(fwd, 60)), flush=True)
