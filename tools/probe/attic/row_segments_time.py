"""probe: votenet_row_segments at the shapes of the train step's static stretch, alone on the GPU."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
rows = 8192
xyz, feats = torch.randn(rows, 3, device=dev), torch.randn(rows, 256, device=dev)
xp = torch.empty(rows, 264, device=dev)
print("vote input  [xyz | feats | 0] -> (8192, 264): %.1f us" % t(lambda: M.row_segments(rows, [(xp[:, :3], xyz, None), (xp[:, 3:259], feats, None), (xp[:, 259:], None, None)])))
off = torch.randn(rows, 264, device=dev)
v_xyz, v_p = torch.empty(rows, 3, device=dev), torch.empty(rows, 256, device=dev)
x = xp[:, :259]
print("votes = x + off, split             : %.1f us" % t(lambda: M.row_segments(rows, [(v_xyz, x[:, :3], off[:, :3]), (v_p, x[:, 3:], off[:, 3:259])])))
d = torch.empty(rows, 256, device=dev)
print("d_seeds = d_votes[:,3:] + d_in[:,3:]: %.1f us" % t(lambda: M.row_segments(rows, [(d, x[:, 3:], off[:, 3:259])])))
a, b = torch.randn(4096, 256, device=dev), torch.randn(4096, 256, device=dev)
o = torch.empty(4096, 256, device=dev)
print("add_rows 4096 x 256                : %.1f us" % t(lambda: M.row_segments(4096, [(o, a, b)])))
g = torch.randn(256, 264, device=dev); gw = torch.randn(256, 259, device=dev)
print("dW += scratch[:, :259] (256 rows)  : %.1f us" % t(lambda: M.row_segments(256, [(gw, gw, g[:, :259])])))
