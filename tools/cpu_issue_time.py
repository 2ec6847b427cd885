"""Host time to ENQUEUE one train step (no synchronisation inside): if it approaches the GPU time of a step, the step is
launch-bound on the host and faster kernels will not show."""
import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import importlib.util
_s = importlib.util.spec_from_file_location('hp', os.path.join(R, 'votenet_amd', 'hostpin.py')); hostpin = importlib.util.module_from_spec(_s); _s.loader.exec_module(hostpin)
if not os.environ.get('NO_PIN'): hostpin.pin(0)  # as bench.py: the host threads on eight cores of the GPU's NUMA node, before torch is imported
import torch
from votenet_amd import synth, loss as VL
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
for i in range(6):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
import gc; gc.collect(); gc.disable()
host, gpu = [], []
for i in range(6, 36):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); gpu.append(t2 - t0)
host.sort(); gpu.sort()
print("host enqueue time per step: median %.2f ms (min %.2f); step from an idle GPU to done: median %.2f ms" % (host[15] * 1e3, host[0] * 1e3, gpu[15] * 1e3))

# how far ahead of the GPU does the host run in a free-running loop?
torch.cuda.synchronize()
evs, ts = [], []
e0 = torch.cuda.Event(enable_timing=True); e0.record(); t0 = time.perf_counter()
for i in range(36, 76):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
lead = [e0.elapsed_time(e) - t * 1e3 for e, t in zip(evs, ts)]
print("free-running: host finishes enqueuing step k this many ms before the GPU finishes it: after 5 steps %.1f, 20 steps %.1f, 40 steps %.1f"
      % (lead[4], lead[19], lead[39]))
print("GPU ms per step over the last 30: %.3f ; host ms per step: %.3f" % ((e0.elapsed_time(evs[39]) - e0.elapsed_time(evs[9])) / 30, (ts[39] - ts[9]) / 30 * 1e3))
