"""dgrad_bn + bn_backward_reduce (two launches) against dgrad_bn_reduce (fused epilogue) at the SA-layer shapes; VARIANT=name picks
tools/probe/lib/libvotenet_NAME.so.  Also checks the fused sums against the separate pass."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import _lib as L
v = os.environ.get("VARIANT")
if v:
    L._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % v)
from votenet_amd import mlp as M
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
print("variant", v or "in-tree")
for rows, c, cout in ((524288, 128, 128), (1048576, 64, 64), (262144, 128, 128), (131072, 128, 128), (262144, 256, 128), (65536, 256, 256), (8192, 256, 256)):
    g = torch.Generator().manual_seed(3)
    z = torch.randn(rows, c, generator=g).to(dev); da = torch.randn(rows, c, generator=g).to(dev)
    coef = torch.randn(5 * c, generator=g).to(dev)
    wT = (torch.randn(c, cout, generator=g) * 0.1).to(dev)
    zp = torch.randn(rows, cout, generator=g).to(dev)
    sc = torch.randn(cout, generator=g).to(dev); sh = torch.randn(cout, generator=g).to(dev)
    mu = torch.randn(cout, generator=g).to(dev); var = (torch.rand(cout, generator=g) + 0.5).to(dev)
    def two():
        o = M.dgrad_bn(z, coef, True, wT, da=da)
        return o, M.bn_backward_reduce(zp, sc, sh, mu, var, True, o)
    def one():
        return M.dgrad_bn(z, coef, True, wT, da=da, below=(zp, sc, sh, mu, var, True))
    o2, s2 = two(); o1, s1 = one()
    assert torch.equal(o1, o2)
    err = ((s1 - s2).abs() / (s2.abs() + 1.0)).max().item()
    t_gemm = gpu_ms(lambda: M.dgrad_bn(z, coef, True, wT, da=da), it=10)
    t2 = gpu_ms(two, it=10); t1 = gpu_ms(one, it=10)
    print("%8d x %3d -> %3d: gemm %.3f  gemm+reduce %.3f  fused %.3f ms  (sums rel err %.1e)" % (rows, c, cout, t_gemm, t2, t1, err))
