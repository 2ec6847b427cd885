"""Which stream bounds the backward pass?  Train step with some weight-gradient-stream kernels skipped (WRONG gradients: timing only).
GPU box only."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth, mlp as M
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def timed(tag):
    run(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
    print("%-40s %.3f ms per step" % (tag, dt / 40 * 1e3), flush=True)
timed("as is")
orig = {k: getattr(M, k) for k in ("gram", "pool_wgrad", "assembled_wgrad_bn", "narrow_wgrad_bn", "wgrad_dense_bn", "wgrad_dense")}
_g = {}
def fake_gram(xz, ss, relu):
    c = xz.shape[1]
    if c not in _g: _g[c] = torch.zeros((c + 1, c), dtype=torch.float32, device=xz.device)
    return _g[c]
M.gram = fake_gram; timed("gram skipped")
M.pool_wgrad = lambda *a, **k: None; timed("+ pool_wgrad skipped")
M.assembled_wgrad_bn = lambda *a, **k: None; M.narrow_wgrad_bn = lambda *a, **k: None; timed("+ assembled / narrow wgrad skipped")
M.wgrad_dense_bn = lambda *a, **k: None; M.wgrad_dense = lambda *a, **k: None; timed("+ dense wgrads skipped (whole stream ~empty)")
for k, v in orig.items(): setattr(M, k, v)
timed("as is again")
