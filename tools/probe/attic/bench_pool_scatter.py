"""pool_dgrad (dense Gram-form GEMM + arg-max scatter with the BatchNorm reduce of the layer below) alone on the GPU (scratch)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import mlp as M
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
for groups, cin, cout in ((8192, 128, 256), (4096, 128, 256), (16384, 64, 128), (2048, 128, 128)):
    k, rows = 64, groups * 64
    g = torch.Generator().manual_seed(1)
    xz = torch.randn(rows, cin, generator=g).to(dev)
    w = (torch.randn(cin, cout, generator=g) * 0.1).to(dev); wT = w.t().contiguous()
    coef = torch.randn(5 * cout, generator=g).to(dev); gout = torch.randn(groups, cout, generator=g).to(dev)
    argmax = torch.randint(0, 64, (groups, cout), generator=g, dtype=torch.int32).to(dev); zsel = torch.randn(groups, cout, generator=g).to(dev)
    sc = torch.ones(cin, device=dev); sh = torch.zeros(cin, device=dev); mu = torch.zeros(cin, device=dev); var = torch.ones(cin, device=dev)
    da = torch.randn(rows, cin, device=dev)
    import ctypes
    from votenet_amd import _lib as L
    sums = torch.zeros(2 * cin, dtype=torch.float64, device=dev)
    def run():
        L.check(L.lib().votenet_pool_dgrad_scatter(groups, k, cin, cout, L.ptr(gout), L.ptr(argmax), L.ptr(zsel), L.ptr(coef), 1, L.ptr(wT), L.ptr(da),
                                                   L.ptr(xz), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(var), 1e-5, 1, L.ptr(sums), L.stream_ptr()))
    ms = gpu_ms(run, it=20)
    print("scatter %6d groups %3d -> %3d: %.4f ms  %.2f TB/s" % (groups, cin, cout, ms, 3.0 * rows * cin * 4 / ms / 1e9))
