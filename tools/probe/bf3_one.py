"""One BF3 GEMM shape, a few launches in both modes (for PMC passes).  usage: bf3_one.py rows cin cout pool"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from votenet_amd import mlp as M, _lib as L
dev = torch.device("cuda:0")
rows, cin, cout, pool = [int(v) for v in sys.argv[1:5]]
x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.1
sc = torch.ones(cin, device=dev); sh = torch.zeros(cin, device=dev)
img = M.SplitImages([w]); img.refresh()
for mode in (0, 1):
    L.lib().votenet_debug_fast_bf3(mode)
    for _ in range(4):
        if pool: M.linear_dense_pool(x, w, pool, None, sc, sh, True, keep_z=False)
        else: M.linear_dense(x, w, None, sc, sh, True)
torch.cuda.synchronize()
