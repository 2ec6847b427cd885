"""Representative MLP GEMM launches in isolation for PMC collection (MFMA busy cycles, HBM bytes).  GPU box only.
sa1 L2 (1 048 576 x 64 -> 128: HBM-bound, 21 flop/B) and sa2 L2 (262 144 x 128 -> 256: MFMA-bound, 43 flop/B):
forward (folded BN+ReLU on the input, statistics epilogue), fused BatchNorm-backward dgrad and wgrad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
for rows, ci, co, k in [(8 * 2048 * 64, 64, 128, 64), (8 * 1024 * 32, 128, 256, 32)]:
    x = torch.randn(rows, ci, device=dev); w = torch.randn(ci, co, device=dev) * 0.1
    sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
    gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    wT = w.t().contiguous()
    img = M.SplitImages([w, wT]); img.refresh()   # the fused GEMMs below then run on bf16 x 3 images (mlp_fast.hip, BF3) ...
    from votenet_amd import _lib as L_
    L_.lib().votenet_debug_fast_bf3(0)             # ... after one pass on the fp32 MFMA kernels, for the same counters side by side
    for _ in range(3):
        z, stats = M.linear_dense(x, w, None, sc, sh, True)
        M.linear_dense_pool(x, w, 64, None, sc, sh, True, keep_z=False)
    L_.lib().votenet_debug_fast_bf3(1)
    for _ in range(3):
        z, stats = M.linear_dense(x, w, None, sc, sh, True)
        M.linear_dense_pool(x, w, 64, None, sc, sh, True, keep_z=False)
    s, h, mu, var = M.bn_finalize(rows, stats, gamma, beta)
    _, argmax = M.bn_relu_max(z, k, s, h, True, want_argmax=True)
    gout = torch.randn(rows // k, co, device=dev)
    sums = M.bn_backward_reduce(z, s, h, mu, var, True, gout, argmax, k)
    coef = M.bn_backward_coef(rows, s, h, mu, var, gamma, sums, None, None)
    dw = torch.zeros_like(w)
    for _ in range(3):
        M.dgrad_bn(z, coef, True, wT, gout=gout, argmax=argmax, k=k)
        M.wgrad_dense_bn(x, z, coef, True, dw, gout=gout, argmax=argmax, k=k, in_scale=sc, in_shift=sh)
torch.cuda.synchronize()
