"""Which tensor-library ops a train step still dispatches (each one is a launch): prints the dict.  GPU box only."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.ops = {}
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = str(func); self.ops[n] = self.ops.get(n, 0) + 1
        return func(*args, **(kwargs or {}))
net = VM.VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000)]
for i in range(4):
    net.train_step(xs[i % 2], gt=gts[i % 2], next_x=[xs[(i + 1) % 2]])
torch.cuda.synchronize()
with Count() as c:
    net.train_step(xs[0], gt=gts[0], next_x=[xs[1]])
torch.cuda.synchronize()
for k, v in sorted(c.ops.items(), key=lambda kv: -kv[1]):
    print("%4d  %s" % (v, k))
