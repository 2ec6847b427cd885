"""cProfile of the host side of one train step (enqueue only): where the host's ~4 ms per step go."""
import os, sys, time, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import synth, loss as VL
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
for i in range(6):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
pr = cProfile.Profile()
for i in range(6, 26):
    torch.cuda.synchronize()
    pr.enable()
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:12000])
