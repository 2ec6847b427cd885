"""Persistent-workgroup caps of the fast GEMM inside the pipelined forward pass and the train step (geometry chains running
beside the GEMMs: a workgroup that shares its CU with an FPS workgroup lags, and a launch ends with its slowest workgroup)."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import _lib as L, loss as VL, model as VM, synth
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def fwd(k):
    for i in range(k):
        net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]])
def trn(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def t(fn, k):
    fn(8); torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter(); fn(k); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k * 1e3; gc.enable(); return dt
for rep in range(2):
    for c22, c41 in ((1024, 2048), (2048, 4096), (4096, 8192), (8192, 16384), (768, 1536)):
        L.lib().votenet_debug_fast_workgroups(c22, c41)
        print("caps %5d / %5d: forward %.3f ms   train step %.3f ms" % (c22, c41, t(fwd, 60), t(trn, 30)), flush=True)
