"""One GEMM layer in isolation for PMC collection (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
rows, ci, co = 524288, 256, 128
x = torch.randn(rows, ci, device=dev); w = torch.randn(ci, co, device=dev)
sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
for _ in range(3):
    M.linear_dense(x, w, want_stats=False)          # dgrad-like
    M.linear_dense(x, w, None, sc, sh, True)        # forward-like
torch.cuda.synchronize()
