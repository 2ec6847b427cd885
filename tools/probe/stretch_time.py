"""(1) GPU wall time of the static small-kernel stretch of a train step (fp1 forward ... fp1 backward: between the end of sa4's forward and
the start of sa4's backward) in the steady-state pipelined loop, against the sum of its kernels (profiles/r03_serial_last_step.txt: 83
launches, 1.18 ms serial of which 0.3 ms weight gradients that run on their own stream).  (2) Do the branches of a two-stream capture run
concurrently when the graph is replayed on this ROCm?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util
_s = importlib.util.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hp = importlib.util.module_from_spec(_s); _s.loader.exec_module(hp)
if not os.environ.get("NO_PIN"): hp.pin(0)
import torch
from votenet_amd import synth, loss as VL
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
marks = []
f0, b0 = net.sa4.forward, net.sa4.backward
def fwd(*a, **k):
    r = f0(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("a", e, time.perf_counter()))
    return r
def bwd(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("b", e, time.perf_counter()))
    return b0(*a, **k)
net.sa4.forward, net.sa4.backward = fwd, bwd
for i in range(8):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
torch.cuda.synchronize(); marks.clear()
import gc; gc.collect(); gc.disable()
e0 = torch.cuda.Event(enable_timing=True); e0.record()
N = 30
for i in range(8, 8 + N):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
e1 = torch.cuda.Event(enable_timing=True); e1.record()
torch.cuda.synchronize()
A = [m for m in marks if m[0] == "a"]; B = [m for m in marks if m[0] == "b"]
gpu = sorted(a[1].elapsed_time(b[1]) for a, b in zip(A, B))
host = sorted((b[2] - a[2]) * 1e3 for a, b in zip(A, B))
print("step %.3f ms; stretch (end of sa4 forward -> start of sa4 backward): GPU wall median %.3f ms (min %.3f max %.3f); host enqueue of it median %.3f ms"
      % (e0.elapsed_time(e1) / N, gpu[N // 2], gpu[0], gpu[-1], host[N // 2]))

# (2) two-stream capture
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(64, 1 << 14, device=dev); b = torch.randn(64, 1 << 14, device=dev)
def work(t):
    for _ in range(40):
        t.mul_(1.0001)   # a small-grid kernel: 1M elements -> leaves most CUs idle
def two(streamed):
    if streamed:
        ev = torch.cuda.Event(); ev.record(); s2.wait_event(ev)
        work(a)
        with torch.cuda.stream(s2):
            work(b)
        ev2 = torch.cuda.Event(); ev2.record(s2); torch.cuda.current_stream().wait_event(ev2)
    else:
        work(a); work(b)
def timeit(fn, n=20):
    torch.cuda.synchronize(); t = torch.cuda.Event(enable_timing=True); u = torch.cuda.Event(enable_timing=True)
    t.record()
    for _ in range(n): fn()
    u.record(); torch.cuda.synchronize(); return t.elapsed_time(u) / n
with torch.cuda.stream(s1):
    two(True); two(False)
    print("eager: one stream %.3f ms, two streams %.3f ms" % (timeit(lambda: two(False)), timeit(lambda: two(True))))
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s1): two(False)
    with torch.cuda.graph(g2, stream=s1): two(True)
    print("graph replay: serial capture %.3f ms, two-stream capture %.3f ms" % (timeit(g1.replay), timeit(g2.replay)))
