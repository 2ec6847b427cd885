"""Alone-time of votenet_pool_dgrad_scatter (the dense GEMM excluded) at the SA levels' shapes on a room batch, against workgroups."""
import os, sys, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth, mlp as M, _lib as L
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
tape = []
net.forward(x, tape)
torch.cuda.synchronize()
hook = L.lib().votenet_debug_scatter_workgroups
hook.restype = None
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, rec in zip(("sa1", "sa2", "sa3", "sa4"), tape[:4]):
    r = rec["recs"][-1]; below = rec["recs"][-2]; half = r["half"]; Lr = r["layer"]
    xz = r["x"]; rows, cin = xz.shape; cout = Lr.cout
    G = rec["argmax"].shape[0]
    gout = torch.randn(G, cout, device=dev); coef = torch.randn(5 * cout, device=dev) * 0.1
    coef[3 * cout:4 * cout], coef[4 * cout:] = r["scale"], r["shift"]
    da = torch.randn(rows, cin, device=dev)
    wT = Lr.wT()
    sums = torch.zeros(2 * cin, dtype=torch.float64, device=dev)
    def run():
        L.check(L.lib().votenet_pool_dgrad_scatter_half(half.nh, half.G, cin, cout, L.ptr(gout), L.ptr(rec["argmax"]), L.ptr(rec["zsel"]), L.ptr(coef), 1,
                L.ptr(wT), L.ptr(da), L.ptr(half.hc), L.ptr(half.wh), L.ptr(xz), L.ptr(below["scale"]), L.ptr(below["shift"]), L.ptr(below["mean"]),
                L.ptr(below["var"]), 1e-5, 1, L.ptr(sums), None, L.stream_ptr()))
    out = []
    for wgs in (32, 64, 128, 256, 512, 1024):
        hook(wgs)
        out.append("%d: %.0f" % (wgs, timeit(run)))
    hook(0)
    print("%s (%d pieces of %d centres, %d -> %d, %d rows): scatter us by workgroups  %s" % (name, half.nh, G, cin, cout, rows, "  ".join(out)), flush=True)
