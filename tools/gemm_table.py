"""Per-launch table of the MFMA GEMMs of one train step: shape, flops, in-step duration and TFLOP/s, then the same launches
alone on the GPU (scratch tool, GPU box only).  python tools/gemm_table.py [--alone]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, mlp as M, model as VM, synth
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
for i in range(8):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
torch.cuda.synchronize()
M.PROFILE_SHAPES = True
rows = {}
NS = 3
for s in range(NS):
    M.PROFILE_EVENTS = []
    i = 8 + s
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
    torch.cuda.synchronize()
    ev = [M.resolve_event(e) for e in M.PROFILE_EVENTS]
    M.PROFILE_EVENTS = None
    for j, (e0, e1, kind, fl, shape) in enumerate(ev):
        rows.setdefault(j, [kind, fl, shape, 0.0])[3] += e0.elapsed_time(e1) / NS
tot_ms = sum(r[3] for r in rows.values()); tot_fl = sum(r[1] for r in rows.values())
print("%3s %-14s %-34s %8s %8s %7s" % ("#", "kind", "shape (rows, cin, cout, note)", "GFLOP", "ms", "TF/s"))
for j, (kind, fl, shape, ms) in rows.items():
    print("%3d %-14s %-34s %8.2f %8.4f %7.1f" % (j, kind, str(shape), fl / 1e9, ms, fl / ms / 1e9))
print("sum: %.1f GFLOP, %.3f ms summed -> %.1f TF/s over the summed time" % (tot_fl / 1e9, tot_ms, tot_fl / tot_ms / 1e9))
agg = {}
for kind, fl, shape, ms in rows.values():
    k = (shape[3] if shape else kind)
    a = agg.setdefault(k, [0.0, 0.0, 0]); a[0] += fl; a[1] += ms; a[2] += 1
for k, (fl, ms, c) in sorted(agg.items(), key=lambda t: -t[1][1]):
    print("  %-14s %3d launches %8.1f GFLOP %8.3f ms %7.1f TF/s" % (k, c, fl / 1e9, ms, fl / ms / 1e9))
