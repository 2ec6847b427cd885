"""Same-box A/B of the train step and the forward pass: GEMMs on bf16 x 3 images (votenet_debug_fast_bf3 1) vs fp32 MFMA (0)."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth, _lib as L
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def fwd(k):
    for i in range(k):
        net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3]])
for rep in range(3):
    for v in (0, 1):
        L.lib().votenet_debug_fast_bf3(v)
        run(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        fwd(6); torch.cuda.synchronize()
        t0 = time.perf_counter(); fwd(40); torch.cuda.synchronize(); df = time.perf_counter() - t0
        print("bf3 = %d: train %.3f ms per step, forward %.3f ms" % (v, dt / 40 * 1e3, df / 40 * 1e3), flush=True)
