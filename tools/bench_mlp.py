"""Per-layer timing of the MLP GEMM kernels at VoteNet shapes (scratch tool, GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
def timeit(fn, it=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
B = 8
print("%-28s %9s %8s %8s" % ("layer", "ms", "TFLOP/s", "GB/s"))
tot = 0
for name, n, m, K, c, widths in [("sa1", 20480, 2048, 64, 3, [64, 64, 128]), ("sa2", 2048, 1024, 64, 128, [128, 128, 256]),
                                 ("sa3", 1024, 512, 64, 256, [128, 128, 256]), ("sa4", 512, 256, 64, 256, [128, 128, 256]),
                                 ("prop", 1024, 256, 64, 256, [128, 128, 128])]:
    xyz = torch.rand(B, n, 3, device=dev); new_xyz = torch.rand(B, m, 3, device=dev)
    feat = torch.randn(B, n, c, device=dev); idx = torch.randint(0, n, (B, m, K), device=dev, dtype=torch.int32)
    rows = B * m * K
    cin = 3 + c
    w = torch.randn(cin, widths[0], device=dev)
    t = timeit(lambda: M.linear_gather(xyz, new_xyz, feat, idx, w))
    fl = 2.0 * rows * cin * widths[0]; by = rows * widths[0] * 4
    print("%-28s %9.3f %8.1f %8.0f" % ("%s L0 gather %dx%d->%d" % (name, rows, cin, widths[0]), t, fl / t / 1e9, by / t / 1e6)); tot += t
    z, _ = M.linear_gather(xyz, new_xyz, feat, idx, w)
    dz = torch.randn_like(z)
    dw = torch.zeros_like(w)
    t = timeit(lambda: M.wgrad_gather(xyz, new_xyz, feat, idx, dz, dw))
    print("%-28s %9.3f %8.1f" % ("   wgrad", t, fl / t / 1e9))
    for i in (1, 2):
        ci, co = widths[i - 1], widths[i]
        w = torch.randn(ci, co, device=dev); sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
        t = timeit(lambda: M.linear_dense(z, w, None, sc, sh, True))
        fl = 2.0 * rows * ci * co; by = rows * (ci + co) * 4
        print("%-28s %9.3f %8.1f %8.0f" % ("%s L%d dense %dx%d->%d" % (name, i, rows, ci, co), t, fl / t / 1e9, by / t / 1e6)); tot += t
        z2, _ = M.linear_dense(z, w, None, sc, sh, True)
        dz2 = torch.randn_like(z2); dw = torch.zeros_like(w)
        t = timeit(lambda: M.wgrad_dense(z, dz2, dw, sc, sh, True))
        print("%-28s %9.3f %8.1f" % ("   wgrad", t, fl / t / 1e9))
        wt = w.t().contiguous()
        t = timeit(lambda: M.linear_dense(dz2, wt, want_stats=False))
        print("%-28s %9.3f %8.1f" % ("   dgrad", t, fl / t / 1e9))
        z = z2
print("forward total %.3f ms" % tot)
