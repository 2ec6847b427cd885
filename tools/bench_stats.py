import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import mlp as M
from tools.bench_mlp_util import timeit
dev = torch.device("cuda:0")
for rows, ci, co in [(1 << 20, 64, 64), (1 << 20, 64, 128), (1 << 18, 128, 128), (1 << 18, 128, 256)]:
    x = torch.randn(rows, ci, device=dev); w = torch.randn(ci, co, device=dev)
    sc = torch.ones(ci, device=dev); sh = torch.zeros(ci, device=dev)
    r = []
    for aff in (False, True):
        for st in (False, True):
            r.append(timeit(lambda: M.linear_dense(x, w, None, sc if aff else None, sh if aff else None, True, want_stats=st)))
    print("%8d x %3d -> %3d   plain %.3f  stats %.3f  affine %.3f  affine+stats %.3f" % (rows, ci, co, *r))
