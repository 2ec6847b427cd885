"""Alone-time of votenet_pool_wgrad_sparse (+ finish) at the SA levels' shapes on a room batch, against the number of workgroups."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth, mlp as M, _lib as L
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
tape = []
net.forward(x, tape)
torch.cuda.synchronize()
hook = L.lib().votenet_debug_sparse_workgroups
hook.restype = None
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, rec in zip(("sa1", "sa2", "sa3", "sa4"), tape[:4]):
    r = rec["recs"][-1]; half = r["half"]; Lr = r["layer"]
    xz = r["x"]; cin = xz.shape[1]; cout = Lr.cout
    G = rec["argmax"].shape[0]
    gout = torch.randn(G, cout, device=dev); coef = torch.randn(5 * cout, device=dev) * 0.1
    coef[3 * cout:4 * cout], coef[4 * cout:] = r["scale"], r["shift"]
    dw = torch.zeros(cin, cout, device=dev); gram = torch.zeros(cin + 1, cin, device=dev)
    out = []
    for wgs in (96, 192, 384, 768, 1536, 3072):
        hook(wgs)
        t = timeit(lambda: M.pool_wgrad(xz, r["in_scale"], r["in_shift"], r["in_relu"], gram, Lr.p("W"), Lr.p("b"), coef, True, gout, rec["argmax"],
                                        rec["zsel"], 64, dw, half=half))
        out.append("%d: %.0f" % (wgs, t))
    hook(384)
    print("%s (%d pieces of %d centres, %d -> %d): sparse + finish, us by workgroups  %s" % (name, half.nh, G, cin, cout, "  ".join(out)), flush=True)
