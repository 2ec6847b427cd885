"""The narrow-first-layer GEMMs of sa1 alone on the GPU (1048576 rows, 6 -> 64 -> 64) next to the materialised forms they replace.
VARIANT=name picks tools/probe/lib/libvotenet_NAME.so."""
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
rows, k0, c0, c1 = 1048576, 6, 64, 64
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
u8 = rnd(rows, 8); u8[:, k0:] = 0
w0, b0, w1 = rnd(k0, c0) * 0.5, rnd(c0) * 0.1, rnd(c0, c1) * 0.2
z0 = M.narrow_z0(u8, w0, b0)
sc, sh, mu, var = torch.rand(c0, generator=g).to(dev) + 0.5, rnd(c0), rnd(c0), torch.rand(c0, generator=g).to(dev) + 0.5
bn = M.FrozenBN(torch.stack([sc, sh]))
z1, da1, coef1 = rnd(rows, c1), rnd(rows, c1), rnd(5 * c1)
wT = w1.t().contiguous()
w1 = w1.contiguous()
imgs = M.SplitImages([w1, wT])  # the bf16 x 3 images: the GEMMs then take the split-operand kernels the step runs
imgs.refresh()
print("split images registered:", imgs.nseg)
dw = torch.zeros(c0, c1, device=dev)
t = {}
t["fwd materialised (z0 read)"] = gpu_ms(lambda: M.linear_dense(z0, w1, None, sc, sh, True), it=10)
t["fwd narrow"] = gpu_ms(lambda: M.narrow_linear(u8, w0, b0, w1, None, bn), it=10)
t["dgrad+reduce materialised"] = gpu_ms(lambda: M.dgrad_bn(z1, coef1, True, wT, da=da1, below=(z0, sc, sh, mu, var, True)), it=10)
t["dgrad+reduce narrow"] = gpu_ms(lambda: M.narrow_dgrad_bn_reduce(z1, coef1, True, wT, da1, u8, w0, b0, (sc, sh, mu, var, True)), it=10)
t["dgrad plain (EPI 1)"] = gpu_ms(lambda: M.dgrad_bn(z1, coef1, True, wT, da=da1), it=10)
t["wgrad materialised"] = gpu_ms(lambda: M.wgrad_dense_bn(z0, z1, coef1, True, dw, da=da1, in_scale=sc, in_shift=sh, in_relu=True), it=10)
t["wgrad narrow"] = gpu_ms(lambda: M.narrow_wgrad_bn(u8, w0, b0, sc, sh, True, z1, coef1, True, da1, dw), it=10)
for k, x in t.items():
    print("%-32s %.4f ms  %.1f TF/s" % (k, x, 2.0 * rows * c0 * c1 / x / 1e9))
