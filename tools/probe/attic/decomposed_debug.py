import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth, pointnet2 as P
dev = torch.device("cuda:0")
b, n = 2, 8192
x = torch.from_numpy(synth.room_batch(b, n, 7)).to(dev)
net = VM.VoteNetHotPath(dev, seed=5, npoints=(1024, 512, 256, 128))
cot = net.make_cotangents(b, seed=1)
res = {}
for flag in (False, True, False):
    P.ASSEMBLED_DECOMPOSED = flag
    net.store.grad.zero_()
    net.store.refresh_transposes()
    tape = []
    out = net.forward(x, tape)
    net.backward(tape, cot)
    torch.cuda.synchronize()
    res.setdefault(flag, []).append({k: net.store.g(k).clone() for k in net.store.views})
a, c = res[False]
d = res[True][0]
worst = []
for k in a:
    sc = float(a[k].abs().max()) + 1e-30
    worst.append((float((a[k] - d[k]).abs().max()) / sc, float((a[k] - c[k]).abs().max()) / sc, k))
for w in sorted(worst, reverse=True)[:25]:
    print("%.3e (run-to-run %.1e)  %s" % w)
