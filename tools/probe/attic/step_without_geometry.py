"""What the concurrent geometry chain costs the step: the pipelined train step (weight-gradient stream on) with the geometry of its batch
GIVEN (computed once, outside the loop) against the usual loop that computes the next batch's geometry underneath every step."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
seeds = (1000, 500000, 900000)
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in seeds]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in seeds]
VM.GEOMETRY_GRAPHS = False  # the geometry of a batch is computed once, launch by launch, and handed to every step that uses it
net = VM.VoteNetHotPath(dev, seed=0)
def usual(k):
    VM.GEOMETRY_GRAPHS = True  # as the bench runs it: the chain of the next batch replayed as one HIP graph
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
usual(6); torch.cuda.synchronize()
VM.GEOMETRY_GRAPHS = False
net.__dict__.setdefault("_prefetched", {}).clear()
for x in xs:
    net.prefetch_geometry(x)
torch.cuda.synchronize()
saved = dict(net._prefetched)
def given(k):
    VM.GEOMETRY_GRAPHS = False
    for i in range(k):
        x = xs[i % 3]
        net._prefetched[id(x)] = saved[id(x)]
        net.train_step(x, gt=gts[i % 3])
for rep in range(3):
    for name, f in (("geometry of the next batch underneath", usual), ("geometry given", given)):
        f(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); f(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("%-40s %.3f ms per step" % (name, dt / 40 * 1e3), flush=True)
