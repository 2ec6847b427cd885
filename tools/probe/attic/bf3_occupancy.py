"""BF3 pooled forward GEMM at one and two workgroups per CU (unused dynamic LDS lowers the occupancy): do the two waves of a SIMD
overlap at all?  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from votenet_amd import mlp as M, _lib as L
dev = torch.device("cuda:0")
def timeit(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for rows, cin, cout in [(524288, 128, 256), (1048576, 64, 128)]:
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.1
    sc = torch.ones(cin, device=dev); sh = torch.zeros(cin, device=dev)
    img = M.SplitImages([w]); img.refresh()
    for caps in ((1024, 2048), (512, 1024), (256, 512)):
        L.lib().votenet_debug_fast_workgroups(*caps)
        for dyn in (0, 60000):
            L.lib().votenet_debug_fast_dyn_lds(dyn)
            t = timeit(lambda: M.linear_dense_pool(x, w, 64, None, sc, sh, True, keep_z=False))
            print("%s  persistent workgroups <= %4d  dynamic LDS %5d (%d workgroup(s) per CU): %.4f ms" % ((rows, cin, cout), caps[0], dyn, 1 if dyn else 2, t), flush=True)
    img.close()
L.lib().votenet_debug_fast_dyn_lds(0); L.lib().votenet_debug_fast_workgroups(1024, 2048)
