"""Forward GEMM variants at the MFMA-bound pooled shapes, alone on the GPU (scratch probe)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import mlp as M
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
for rows, ci, co in ((524288, 128, 256), (1048576, 64, 128), (262144, 128, 256), (524288, 128, 128)):
    x = torch.randn(rows, ci, device=dev); w = torch.randn(ci, co, device=dev) * 0.1
    sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
    fl = 2.0 * rows * ci * co
    t = {}
    t["plain, z stored, no stats"] = gpu_ms(lambda: M.linear_dense(x, w, None, sc, sh, True, want_stats=False), it=10)
    t["z stored + stats (EPI 0)"] = gpu_ms(lambda: M.linear_dense(x, w, None, sc, sh, True, want_stats=True), it=10)
    if M.linear_pool_supported(rows, ci, co, 64):
        t["pool epilogue, z not stored (EPI 2)"] = gpu_ms(lambda: M.linear_dense_pool(x, w, 64, None, sc, sh, True, keep_z=False), it=10)
        t["pool epilogue + z stored"] = gpu_ms(lambda: M.linear_dense_pool(x, w, 64, None, sc, sh, True, keep_z=True), it=10)
    print("%d x %d -> %d (%.1f GFLOP)" % (rows, ci, co, fl / 1e9))
    for k, v in t.items():
        print("   %-38s %.4f ms  %.1f TFLOP/s" % (k, v, fl / v / 1e9))
