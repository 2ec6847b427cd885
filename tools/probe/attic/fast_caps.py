"""Alone-time of the three large fast-kernel GEMMs of an SA level on the piece layout (assembled forward with statistics, forward with
pooling, assembled input gradient with the reduce of the layer below) against the cap on persistent workgroups: every workgroup ends with
fp64 atomics on the same 2 c addresses."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth, mlp as M, _lib as L, pointnet2 as P
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
tape = []
net.forward(x, tape)
torch.cuda.synchronize()
hook = L.lib().votenet_debug_fast_workgroups
hook.restype = None
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, rec in zip(("sa2", "sa3"), tape[1:3]):
    r0, r1, r2 = rec["recs"]; half = r0["half"]
    L0, L1, L2 = r0["layer"], r1["layer"], r2["layer"]
    geo, Pt, wx = r0["geo"], r0["P"], r0["wx"]
    bn0 = M.FrozenBN(torch.stack([r0["scale"], r0["shift"]]).contiguous())
    bn1 = M.FrozenBN(torch.stack([r1["scale"], r1["shift"]]).contiguous())
    z1 = r1["z"]
    rows, c = z1.shape
    coef = torch.randn(5 * c, device=dev) * 0.1
    coef[3 * c:4 * c], coef[4 * c:] = r1["scale"], r1["shift"]
    da = torch.randn(rows, c, device=dev)
    below = (r0["scale"], r0["shift"], r0["mean"], r0["var"], True)
    out = []
    for cap in (1024, 768, 512, 384, 256):
        hook(cap, 2 * cap)
        a = timeit(lambda: M.assembled_linear(geo, Pt, wx, L1.p("W"), L1.p("b"), bn0, True, half=half))
        b = timeit(lambda: M.linear_dense_pool(z1, L2.p("W"), 64, L2.p("b"), None, None, True, keep_z=False, in_bn=bn1, half=half, gamma=L2.p("gamma")))
        d = timeit(lambda: M.assembled_dgrad_bn_reduce(z1, coef, True, L1.wT(), da, geo, Pt, wx, below, half=half))
        out.append("cap %4d: fwd+bn %.0f  fwd+pool %.0f  dgrad+reduce %.0f" % (cap, a, b, d))
    hook(1024, 2048)
    print("%s (%d rows):\n  %s" % (name, rows, "\n  ".join(out)), flush=True)
