"""What is the step sensitive to?  The train step with single kernel families SKIPPED (wrong numbers, right timing): how much of a
family's alone-time the step gets back when the family disappears -- on the weight-gradient stream and on the main chain."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth, mlp as M, pointnet2 as P
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
real = {k: getattr(M, k) for k in ("gram", "pool_wgrad", "assembled_wgrad_bn", "narrow_wgrad_bn", "pool_dgrad", "dgrad_bn_half", "group_linear_backward_decomposed", "narrow_dgrad_bn_reduce")}
_g = {}
def fake_gram(xz, ss, relu, half=None):
    c = xz.shape[1]
    if c not in _g: _g[c] = torch.zeros(c + 1, c, device=dev)
    return _g[c]
_buf = {}
def buf(shape):
    if shape not in _buf: _buf[shape] = torch.zeros(shape, device=dev)
    return _buf[shape]
def fake_dgrad_half(z, coef, relu, wT, da, half):
    return buf((z.shape[0], wT.shape[1]))
def fake_pool_dgrad(xz, *a, **k):
    out = buf(tuple(xz.shape))
    if k.get("below") is not None:
        cin = xz.shape[1]
        return out, buf((5 * cin,))
    return out
cfgs = [("nothing skipped", {}),
        ("gram skipped (wgrad stream, 0.33 ms alone)", {"gram": fake_gram}),
        ("gram + sparse gather skipped (0.69)", {"gram": fake_gram, "pool_wgrad": lambda *a, **k: None}),
        ("all four big wgrad families skipped (1.0)", {"gram": fake_gram, "pool_wgrad": lambda *a, **k: None, "assembled_wgrad_bn": lambda *a, **k: None, "narrow_wgrad_bn": lambda *a, **k: None}),
        ("assembled plain dgrad skipped (main, ~0.25)", {"dgrad_bn_half": fake_dgrad_half}),
        ("Gram dgrad + scatter skipped (main, ~0.63)", {"pool_dgrad": fake_pool_dgrad})]
for rep in range(2):
    for name, patch in cfgs:
        for k, v in real.items(): setattr(M, k, v)
        for k, v in patch.items(): setattr(M, k, v)
        net.drop_graphs()
        run(8); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("%-52s %.3f ms per step" % (name, dt / 40 * 1e3), flush=True)
