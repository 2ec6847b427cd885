"""probe: per-step times of the pipelined train step (HIP events at the step boundaries on the main stream) -- which steps are the slow ones?"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
NB = int(os.environ.get("NB", "3"))
seeds = [1000 + 7919 * i for i in range(NB)]
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in seeds]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in seeds]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k, evs=None):
    for i in range(k):
        net.train_step(xs[i % NB], gt=gts[i % NB], next_x=[xs[(i + 1) % NB]])
        if evs is not None:
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
run(10); torch.cuda.synchronize()
if os.environ.get("NOGC"): gc.collect(); gc.disable()
evs = []
t0 = time.perf_counter(); run(80, evs); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 80 * 1e3
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
s = sorted(d)
print("mean %.3f (wall %.3f)  median %.3f  p10 %.3f  p90 %.3f  max %.3f" % (sum(d) / len(d), wall, s[len(s) // 2], s[len(s) // 10], s[9 * len(s) // 10], s[-1]))
print("steps > median + 0.2 ms:", [(i, round(v, 2)) for i, v in enumerate(d) if v > s[len(s) // 2] + 0.2])
print("by batch index (mean):", [round(sum(d[i] for i in range(len(d)) if (i + 1) % NB == b) / max(1, len([i for i in range(len(d)) if (i + 1) % NB == b])), 3) for b in range(NB)])
