"""Which call sites allocate (torch.empty / zeros / full) during one train step, and how often."""
import os, sys, collections, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import synth, loss as VL
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
for i in range(8):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
torch.cuda.synchronize()
sites = collections.Counter()
orig = {n: getattr(torch, n) for n in ("empty", "zeros", "full", "empty_like", "zeros_like")}
def wrap(name):
    f = orig[name]
    def g(*a, **k):
        fr = traceback.extract_stack(limit=3)[0]
        sites[(name, os.path.basename(fr.filename), fr.lineno, fr.name)] += 1
        return f(*a, **k)
    return g
for n in orig: setattr(torch, n, wrap(n))
net.train_step(xs[2], gt=gts[2], next_x=xs[0])
for n in orig: setattr(torch, n, orig[n])
torch.cuda.synchronize()
print("allocations in one step:", sum(sites.values()))
for (name, f, ln, fn), c in sites.most_common(40):
    print("%3d  torch.%-10s %s:%d %s" % (c, name, f, ln, fn))
