"""Can the step be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed?  Scratch probe, GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import synth, loss as VL
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
net = VoteNetHotPath(dev, seed=0)


def timeit(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3


print("eager forward %.3f ms" % timeit(lambda: net.forward(x)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        net.forward(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = net.forward(x)
    ref = net.forward(x)
    g.replay()
    torch.cuda.synchronize()
    print("captured; replay matches eager:", bool(torch.equal(out["proposals_output"], ref["proposals_output"])))
    print("graph forward %.3f ms" % timeit(lambda: g.replay()))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400])
