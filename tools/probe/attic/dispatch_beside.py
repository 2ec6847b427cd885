"""Does a long-running kernel on another stream delay the DISPATCH of kernels on this one?  A chain of dependent medium kernels on the
main stream, alone and beside one long kernel (sa1 FPS, 8 workgroups for 1.7 ms) per ~1.9 ms on a side stream (scratch, GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import synth, tf_sampling
dev = torch.device("cuda:0")
x0 = torch.from_numpy(synth.room_batch(8, 20480, 1)).to(dev)
side = torch.cuda.Stream(device=dev)
for nel, label in ((1 << 16, "tiny (64K floats)"), (1 << 22, "medium (4M floats, ~10 us)"), (1 << 24, "16M floats (~35 us)")):
    a = torch.zeros(nel, device=dev)
    def chain(n):
        for _ in range(n):
            a.add_(1.0)
    for beside in (False, True):
        chain(50); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps, n = 6, 100
        e0.record()
        for r in range(reps):
            if beside:
                with torch.cuda.stream(side):
                    tf_sampling.farthest_point_sample(2048, x0)
            chain(n)
        e1.record(); torch.cuda.synchronize()
        print("%-28s %s: %.2f us per kernel" % (label, "beside FPS" if beside else "alone     ", e0.elapsed_time(e1) / (reps * n) * 1e3), flush=True)
