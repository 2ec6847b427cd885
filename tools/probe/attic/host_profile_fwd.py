"""Host side of the forward pass: enqueue time per call (no sync inside) against GPU time, and the cProfile top (scratch, GPU box)."""
import os, sys, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k, j=0):
    for i in range(j, j + k):
        net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]])
run(9); torch.cuda.synchronize()
import gc; gc.collect(); gc.disable()
host = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(1, i); host.append(time.perf_counter() - t0)
host.sort(); print("host enqueue per forward: median %.3f ms" % (host[15] * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); print("free-running: %.3f ms per forward" % ((time.perf_counter() - t0) / 60 * 1e3))
pr = cProfile.Profile(); pr.enable(); run(30); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
