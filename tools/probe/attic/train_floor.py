"""Train step with its geometry already on the device and no geometry chain beside it, against the pipelined step the bench times."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
VM.GEOMETRY_GRAPHS = False  # the geometry of a batch is computed once, launch by launch, and handed to every step that uses it
net = VM.VoteNetHotPath(dev, seed=0)
def piped(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
piped(8); torch.cuda.synchronize(); gc.disable()
t0 = time.perf_counter(); piped(40); torch.cuda.synchronize(); print("pipelined: %.3f ms per step" % ((time.perf_counter() - t0) / 40 * 1e3))
net._prefetched.clear()
for x in xs:
    net.prefetch_geometry(x)
torch.cuda.synchronize()
saved = dict(net._prefetched)
def floor(k):
    for i in range(k):
        x = xs[i % 3]
        net._prefetched[id(x)] = saved[id(x)]
        net.train_step(x, gt=gts[i % 3])
floor(8); torch.cuda.synchronize()
t0 = time.perf_counter(); floor(40); torch.cuda.synchronize(); print("geometry given, no chain beside the step: %.3f ms per step" % ((time.perf_counter() - t0) / 40 * 1e3))
