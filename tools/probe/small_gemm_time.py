"""The small GEMMs of the step's static stretch (fp1 / fp2 / voting / proposal mlp2: 2048-8192 rows) alone on the GPU, on a library VARIANT."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
out = []
for rows, cin, cout in ((4096, 512, 256), (4096, 256, 256), (8192, 512, 256), (8192, 256, 256), (8192, 320, 256), (8192, 256, 320), (8192, 256, 128), (2048, 128, 128)):
    x = torch.randn(rows, cin, generator=g).to(dev)
    w = (torch.randn(cin, cout, generator=g) * 0.1).to(dev)
    sc, sh = torch.rand(cin, generator=g).to(dev) + 0.5, torch.randn(cin, generator=g).to(dev) * 0.1
    img = M.SplitImages([w]); img.refresh()
    t_stats = timeit(lambda: M.linear_dense(x, w, None, sc, sh, True, want_stats=True))
    t_plain = timeit(lambda: M.linear_dense(x, w, None, sc, sh, True, want_stats=False))
    out.append("%5d x %3d -> %3d: fwd+stats %5.1f us  plain %5.1f us" % (rows, cin, cout, t_stats, t_plain))
    img.close()
print("variant %s (includes ~4 us of launch + one zero fill of the statistics per call)\n  " % (os.environ.get("VARIANT") or "(built)") + "\n  ".join(out))
