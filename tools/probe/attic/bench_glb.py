"""group_linear_backward alone at the step's shapes: atomics (default) against the gather-sum over the inverse index
(deterministic mode's kernel), and the atomics form with z aliased to da (half the HBM reads: read- or atomic-bound?)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import mlp as M, model as VM, synth
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 7)).to(dev)
xyz = x
for name, c in (("sa1", 64), ("sa2", 128), ("sa3", 128), ("sa4", 128)):
    mod = getattr(net, name)
    g = mod.geometry(xyz)
    fps_idx, new_xyz, idx, cnt = g[:4]
    b, m, k = idx.shape
    n = xyz.shape[1]
    rows = b * m * k
    z = torch.randn(rows, c, device=dev); da = torch.randn(rows, c, device=dev); coef = torch.randn(5 * c, device=dev)
    dw = torch.zeros(3, c, device=dev)
    t0 = gpu_ms(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, z, da, coef, True, dw), it=10)
    t1 = gpu_ms(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, da, da, coef, True, dw), it=10)
    prev = M.set_deterministic(True)
    M._inverse_of(idx, n)
    t2 = gpu_ms(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, z, da, coef, True, dw), it=10)
    M.set_deterministic(prev)
    print("%s n %d m %d k %d c %d rows %d: atomics %.3f ms   (z aliased to da %.3f)   gather-sum %.3f ms   mean cnt %.1f  rows/point %.1f"
          % (name, n, m, k, c, rows, t0, t1, t2, cnt.float().mean().item(), rows / (b * n)))
    xyz = new_xyz
