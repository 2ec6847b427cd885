"""First SA layer assembled in the next GEMM's loader (assemble.hip) against votenet_group_linear + the ordinary GEMM, at the step's
sa2 / sa3 / sa4 shapes: values (tolerance) and time."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import mlp as M, model as VM, synth
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 7)).to(dev)
xyz = x
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
for name, cf, c0, c1 in (("sa1", 3, 64, 64), ("sa2", 128, 128, 128), ("sa3", 256, 128, 128), ("sa4", 256, 128, 128)):
    mod = getattr(net, name)
    fps_idx, new_xyz, idx, cnt = mod.geometry(xyz)[:4]
    b, m, k = idx.shape
    n = xyz.shape[1]
    rows = b * m * k
    if name != "sa1":
        feat = rnd(b, n, cf)
        W0, b0, W1 = rnd(3 + cf, c0) * 0.2, rnd(c0) * 0.1, rnd(c0, c1) * 0.2
        gamma, beta = rnd(c0) * 0.2 + 1.0, rnd(c0) * 0.1
        # materialised path
        def mat():
            P, _ = M.linear_dense(feat.reshape(b * n, cf), W0[3:].contiguous(), None, want_stats=False)
            z0, st = M.group_linear(xyz, new_xyz, idx, P, W0[:3].contiguous(), b0)
            bn = M.PendingBN(st, gamma, beta, rows)
            z1, st1 = M.linear_dense(z0, W1, None, None, None, True, in_bn=bn)
            return z1, st1, bn
        geo, cntv, mom = M.assemble_rows(xyz, new_xyz, idx, pts_cnt=cnt)
        wx = W0[:3].contiguous()
        def asm():
            P, _ = M.linear_dense(feat.reshape(b * n, cf), W0[3:].contiguous(), b0, want_stats=False)
            st = M.assemble_stats(P, cntv, wx, mom)
            bn = M.PendingBN(st, gamma, beta, rows)
            z1, st1 = M.assembled_linear(geo, P, wx, W1, None, bn)
            return z1, st1, bn
        z1a, s1a, bna = mat(); z1b, s1b, bnb = asm()
        rel = lambda a, b_: float((a.double() - b_.double()).abs().max() / b_.double().abs().max())
        print("%s rows %d c0 %d: z1 rel %.2e  stats1 rel %.2e  bn0 mean rel %.2e var rel %.2e" % (name, rows, c0, rel(z1b, z1a), rel(s1b, s1a), rel(bnb.mean, bna.mean), rel(bnb.var, bna.var)))
        ta, tb = gpu_ms(mat, it=10), gpu_ms(asm, it=10)
        tg = gpu_ms(lambda: M.assemble_rows(xyz, new_xyz, idx, pts_cnt=cnt), it=10)
        print("   materialised (P GEMM + group_linear + GEMM) %.3f ms   assembled (P GEMM + stats + GEMM) %.3f ms   [geometry-time rows kernel %.3f ms]" % (ta, tb, tg))
    xyz = new_xyz
