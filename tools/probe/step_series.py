"""The pipelined train step's boundaries, one HIP event per step: the series of step times (where are the long ones?).
python tools/probe/step_series.py [steps]"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
EARLY = os.environ.get("GC_EARLY") == "1"  # collect garbage and create the events BEFORE the warm-up steps: no idle gap in front of the timed steps
if EARLY:
    gc.collect(); gc.disable()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
for i in range(12):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
torch.cuda.synchronize()
if not EARLY:
    gc.collect(); gc.disable()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
marks[0].record()
for i in range(N):
    t0 = time.perf_counter()
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
    host.append((time.perf_counter() - t0) * 1e3)
    marks[i + 1].record()
torch.cuda.synchronize()
ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(N)]
srt = sorted(ms)
print("steps %d: mean %.3f  median %.3f  min %.3f  p90 %.3f  max %.3f" % (N, sum(ms) / N, srt[N // 2], srt[0], srt[int(N * 0.9)], srt[-1]))
print("by batch (i %% 3): " + "  ".join("%d: mean %.3f" % (k, sum(ms[k::3]) / len(ms[k::3])) for k in range(3)))
print("series (ms):", " ".join("%.2f" % v for v in ms))
print("host enqueue (ms):", " ".join("%.2f" % v for v in host[:60]))
