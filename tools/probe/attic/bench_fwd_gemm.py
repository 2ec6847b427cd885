"""Forward GEMMs of the SA layers alone on the GPU: fwd+bn (EPI 0) and fwd+pool (EPI 2) at the step's shapes.  VARIANT=name picks
tools/probe/lib/libvotenet_NAME.so."""
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
cap22, cap41 = int(os.environ.get("CAP22", "0")), int(os.environ.get("CAP41", "0"))
if cap22 or cap41:
    L.lib().votenet_debug_fast_workgroups(cap22, cap41)
print("variant", v or "in-tree", "caps", cap22, cap41)
tot = 0.0
for rows, c, cout, pool in ((1048576, 64, 64, 0), (1048576, 64, 128, 64), (524288, 128, 128, 0), (524288, 128, 256, 64), (262144, 128, 128, 0),
                            (262144, 128, 256, 64), (131072, 128, 128, 0), (131072, 128, 256, 64), (8192, 256, 256, 0), (8192, 512, 256, 0)):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, c, generator=g).to(dev); w = (torch.randn(c, cout, generator=g) * 0.1).to(dev)
    sc = torch.rand(c, generator=g).to(dev) + 0.5; sh = torch.randn(c, generator=g).to(dev)
    if pool:
        fn = lambda: M.linear_dense_pool(x, w, pool, None, sc, sh, True, keep_z=False)
    else:
        fn = lambda: M.linear_dense(x, w, None, sc, sh, True)
    t = gpu_ms(fn, it=10); tot += t
    print("%8d x %3d -> %3d %s: %.4f ms  %.1f TF/s" % (rows, c, cout, "pool" if pool else "bn  ", t, 2.0 * rows * c * cout / t / 1e9))
print("total %.3f ms" % tot)
