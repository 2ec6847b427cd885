"""Same-box A/B of the train step: Gram matrices on bf16 x 3 split operands (votenet_debug_gram_bf3 1) vs the fp32 MFMA kernel (0)."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth, _lib as L, mlp as M
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for rows, c in [(524288, 128), (1048576, 64)]:
    z = torch.randn(rows, c, device=dev); ss = torch.stack([torch.ones(c), torch.zeros(c)]).to(dev)
    for v in (0, 1):
        L.lib().votenet_debug_gram_bf3(v)
        print("gram %d x %d alone, bf3 = %d: %.4f ms" % (rows, c, v, t(lambda: M.gram(z, ss, True))), flush=True)
for rep in range(3):
    for v in (0, 1):
        L.lib().votenet_debug_gram_bf3(v)
        run(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("gram bf3 = %d: train %.3f ms per step" % (v, dt / 40 * 1e3), flush=True)
