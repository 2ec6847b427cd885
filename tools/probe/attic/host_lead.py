"""Where in the train step does the GPU catch up with the host?  After every module's forward / backward call the host time and a HIP
event: lead = (GPU time the event fires) - (host time it was enqueued), in a free-running loop.  A lead near zero = the GPU ran dry there."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
from votenet_amd import hostpin; hostpin.pin(0)  # as bench.py: the host threads on eight cores of the GPU's NUMA node
import torch
from votenet_amd import loss as VL, model as VM, pointnet2 as P, synth
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
marks = []
on = [False]
def wrap(obj, meth, label):
    f = getattr(obj, meth)
    def g(*a, **k):
        r = f(*a, **k)
        if on[0]:
            e = torch.cuda.Event(enable_timing=True); e.record()
            marks.append((label, time.perf_counter(), e))
        return r
    setattr(obj, meth, g)
for name in ("sa1", "sa2", "sa3", "sa4", "fp1", "fp2", "proposal"):
    wrap(getattr(net, name), "forward", name + ".fwd")
    wrap(getattr(net, name), "backward", name + ".bwd")
wrap(net, "vote", "vote.fwd")
wrap(net, "update_moving_averages", "ema")
wrap(net, "backward", "backward(all)")
wrap(net, "train_step", "step end")
for i in range(8):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
e0 = torch.cuda.Event(enable_timing=True); e0.record(); t0 = time.perf_counter()
on[0] = True
N = 30
for i in range(8, 8 + N):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
torch.cuda.synchronize()
per = len(marks) // N
print("%d marks per step; GPU %.3f ms per step" % (per, e0.elapsed_time(marks[-1][2]) / N))
acc = {}
for s in range(10, N):  # steady state
    base_h = marks[s * per - 1][1]; base_g = e0.elapsed_time(marks[s * per - 1][2])
    for j in range(per):
        lab, th, ev = marks[s * per + j]
        tg = e0.elapsed_time(ev)
        a = acc.setdefault((j, lab), [0.0, 0.0, 0.0])
        a[0] += tg - (th - t0) * 1e3; a[1] += (th - base_h) * 1e3; a[2] += tg - base_g
n = N - 10
print("%-16s %10s %14s %14s" % ("after", "lead ms", "host ms in step", "GPU ms in step"))
for (j, lab), a in sorted(acc.items()):
    print("%-16s %10.2f %14.2f %14.2f" % (lab, a[0] / n, a[1] / n, a[2] / n))
