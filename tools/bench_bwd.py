"""Fused vs unfused BatchNorm-backward GEMMs at VoteNet shapes (scratch tool, GPU box only)."""
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
print("%-34s %8s %8s %8s %8s %8s | %8s %8s %8s" % ("layer rows x cin -> c (k)", "apply", "dgrad", "dg_bn", "dg_bn+r", "reduce", "wgrad", "wg_bn", ""))
for name, rows, cin, c, k in [("sa1 L2", 8 * 2048 * 64, 64, 128, 64), ("sa1 L1", 8 * 2048 * 64, 64, 64, 0),
                              ("sa2 L2", 8 * 1024 * 32, 128, 256, 32), ("sa2 L1", 8 * 1024 * 32, 128, 128, 0),
                              ("sa3 L2", 8 * 512 * 16, 128, 256, 16), ("sa3 L1", 8 * 512 * 16, 128, 128, 0),
                              ("fp1 L0", 8 * 512, 512, 256, 0), ("fp2 L0", 8 * 1024, 512, 256, 0)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, c, device=dev) * 0.1
    zprev = torch.randn(rows, cin, device=dev)
    ps = [torch.ones(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(cin, device=dev), torch.ones(cin, device=dev)]
    z, stats = M.linear_dense(x, w)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    sc, sh, mu, var = M.bn_finalize(rows, stats, gamma, beta)
    if k:
        _, argmax = M.bn_relu_max(z, k, sc, sh, True, want_argmax=True)
        up = torch.randn(rows // k, c, device=dev); src = dict(gout=up, argmax=argmax, k=k)
    else:
        argmax = None; up = torch.randn(rows, c, device=dev); src = dict(da=up)
    sums = M.bn_backward_reduce(z, sc, sh, mu, var, True, up, argmax, k)
    coef = M.bn_backward_coef(rows, sc, sh, mu, var, gamma, sums, None, None)
    dz = M.bn_backward_apply(z, coef, True, up, argmax, k)
    wT = w.t().contiguous(); dw = torch.zeros_like(w)
    da_ref, _ = M.linear_dense(dz, wT, want_stats=False)
    t_apply = timeit(lambda: M.bn_backward_apply(z, coef, True, up, argmax, k))
    t_dg = timeit(lambda: M.linear_dense(dz, wT, want_stats=False))
    t_dgbn = timeit(lambda: M.dgrad_bn(z, coef, True, wT, **src))
    t_dgbnr = float("nan")  # (the fused-reduction epilogue was measured slower than the separate pass and removed)
    t_red = timeit(lambda: M.bn_backward_reduce(zprev, *ps, True, da_ref))
    t_wg = timeit(lambda: M.wgrad_dense(x, dz, dw))
    t_wgbn = timeit(lambda: M.wgrad_dense_bn(x, z, coef, True, dw, **src))
    print("%-34s %8.3f %8.3f %8.3f %8.3f %8.3f | %8.3f %8.3f" % ("%s %dx%d->%d (%d)" % (name, rows, cin, c, k), t_apply, t_dg, t_dgbn,
                                                                  t_dgbnr, t_red, t_wg, t_wgbn))
