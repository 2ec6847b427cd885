"""Does walking a tensor BACK TO FRONT in the kernel after the one that streamed it front to back find its tail in the Infinity Cache
(256 MB; sa2's activations are 268 MB each)?  pool_dgrad = dense GEMM (reads x, writes da, front to back) then the arg-max scatter pass
(read-modify-write of da, reads x for the reduce): the pair with the scatter forward and reversed (votenet_debug_scatter_reverse)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import mlp as M, _lib as L
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
hook = L.lib().votenet_debug_scatter_reverse
hook.restype = None
for name, groups, cin, cout in (("sa2", 8192, 128, 256), ("sa1", 16384, 64, 128), ("sa3", 4096, 128, 256)):
    k = 64
    rows = groups * k
    g = torch.Generator().manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    xz = rnd(rows, cin)
    aff = torch.stack([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2]).to(dev).contiguous()
    w, b = rnd(cin, cout) * 0.1, rnd(cout) * 0.1
    wT = w.t().contiguous()
    coef = rnd(5 * cout)
    gout = rnd(groups, cout)
    arg = torch.randint(0, k, (groups, cout), generator=g, dtype=torch.int32).to(dev)
    zsel = rnd(groups, cout)
    below = (aff[0], aff[1], rnd(cin) * 0.1, (torch.rand(cin, generator=g) + 0.5).to(dev), True)
    mm = M.pool_dgrad_prepare(w, b, coef, 1)
    def pair():
        return M.pool_dgrad(xz, aff[0], aff[1], True, w, b, wT, coef, True, gout, arg, zsel, k, below=below, mm=mm)
    res = {}
    for rev in (0, 1, 0, 1):
        hook(rev)
        res.setdefault(rev, []).append(gpu_ms(pair, it=10))
    hook(0)
    print("%s (%d x %d, %.0f MB per tensor): dense + scatter forward %.4f / %.4f ms, scatter reversed %.4f / %.4f ms"
          % (name, rows, cin, rows * cin * 4 / 1e6, res[0][0], res[0][1], res[1][0], res[1][1]))
