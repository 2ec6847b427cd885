"""group_linear_backward alone: real (z, da) against z aliased to da (half the HBM reads) -> is the pass read- or atomic-bound?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, R + "/tools"]
import torch, numpy as np
from votenet_amd import mlp as M, tf_grouping, tf_sampling
from bench_legs import gpu_ms
from votenet_amd import synth
dev = torch.device("cuda:0")
xyz_all = torch.from_numpy(synth.room_batch(8, 20480, 7)[..., :3].copy()).to(dev)
for (n, m, k, r, c) in ((20480, 2048, 64, 0.2, 64), (2048, 1024, 64, 0.4, 128)):
    xyz = xyz_all[:, :n].contiguous()
    fi = tf_sampling.farthest_point_sample(m, xyz); new_xyz = tf_sampling.gather_point(xyz, fi)
    idx, cnt = tf_grouping.query_ball_point(r, k, xyz, new_xyz)
    rows = 8 * m * k
    z = torch.randn(rows, c, device=dev); da = torch.randn(rows, c, device=dev); coef = torch.randn(5 * c, device=dev)
    dw = torch.zeros(3, c, device=dev)
    t0 = gpu_ms(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, z, da, coef, True, dw), it=10)
    t1 = gpu_ms(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, da, da, coef, True, dw), it=10)
    S = torch.zeros(8, n, c, device=dev)
    t2 = gpu_ms(lambda: torch.zeros(8, n, c, device=dev), it=10)
    print("n %d m %d k %d c %d: (z,da) %.3f ms   z aliased to da %.3f ms   zero fill of S %.3f ms   mean cnt %.1f" % (n, m, k, c, t0, t1, t2, cnt.float().mean().item()))
