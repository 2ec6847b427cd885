"""Backward of a pooled SA layer, Gram form (pool_bwd.hip) against the direct form, kernel by kernel, alone on the GPU."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
from votenet_amd import mlp as M, _lib as L
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
for name, groups, cin, cout in (("sa2", 8192, 128, 256), ("sa3", 4096, 128, 256), ("sa1", 16384, 64, 128), ("prop", 2048, 128, 128)):
    k, rows = 64, groups * 64
    g = torch.Generator().manual_seed(1)
    xz = torch.randn(rows, cin, generator=g).to(dev)
    aff = torch.stack([torch.randn(cin, generator=g) * 0.3 + 1, torch.randn(cin, generator=g) * 0.2]).to(dev).contiguous()
    w = (torch.randn(cin, cout, generator=g) * 0.15).to(dev); b = torch.zeros(cout, device=dev)
    gamma = torch.ones(cout, device=dev); beta = torch.zeros(cout, device=dev)
    z, st, pool = M.linear_dense_pool(xz, w, k, b, aff[0], aff[1], True)
    sc, sh, mean, var = M.bn_finalize(rows, st, gamma, beta)
    out, arg, zsel = M.bn_pool_finalize(pool, sc, sh, True, want_argmax=True, want_zsel=True)
    gout = torch.randn(groups, cout, generator=g).to(dev)
    sums = M.bn_backward_reduce_pool(gout, zsel, sc, sh, mean, var, True)
    coef = M.bn_backward_coef(rows, sc, sh, mean, var, gamma, sums, torch.zeros(cout, device=dev), torch.zeros(cout, device=dev))
    wT = w.t().contiguous(); dw = torch.zeros(cin, cout, device=dev)
    mm = torch.empty((cin + 1, cin), device=dev); da = torch.empty(rows, cin, device=dev); G = M.gram(xz, aff, True)
    lib = L.lib(); P = L.ptr; S = L.stream_ptr
    t = {}
    t["direct dgrad"] = timeit(lambda: M.dgrad_bn(z, coef, True, wT, gout=gout, argmax=arg, k=k), it=20)
    t["direct wgrad"] = timeit(lambda: M.wgrad_dense_bn(xz, z, coef, True, dw, gout=gout, argmax=arg, k=k, in_scale=aff[0], in_shift=aff[1], in_relu=True), it=20)
    t["prepare"] = timeit(lambda: lib.votenet_pool_dgrad_prepare(cin, cout, P(w), P(b), P(coef), P(mm), P(mm[cin]), S()), it=20)
    t["dense dgrad"] = timeit(lambda: M.linear_dense(xz, mm[:cin], mm[cin], aff[0], aff[1], True, want_stats=False), it=20)
    t["scatter"] = timeit(lambda: lib.votenet_pool_dgrad_scatter(groups, k, cin, cout, P(gout), P(arg), P(zsel), P(coef), 1, P(wT), P(da), None, None, None, None, None, 1e-5, 0, None, S()), it=20)
    t["gram"] = timeit(lambda: M.gram(xz, aff, True), it=20)
    t["sparse wgrad + finish"] = timeit(lambda: M.pool_wgrad(xz, aff[0], aff[1], True, G, w, b, coef, True, gout, arg, zsel, k, dw), it=20)
    print(name, "  ".join("%s %.3f" % kv for kv in t.items()))
