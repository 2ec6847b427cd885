"""Time the first-layer backward of an SA level (GroupPointGrad at the layer output width) in its forms on a room batch: atomics on the
full layout, the gather over the inverse index (deterministic mode), atomics on the half-group layout."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth, mlp as M, pointnet2 as P
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
tape = []
net.forward(x, tape)
torch.cuda.synchronize()
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, rec in zip(("sa2", "sa3", "sa4"), tape[1:4]):
    r0 = rec["recs"][0]; half = r0["half"]
    xyz, new_xyz, idx, cnt = r0["xyz"], r0["new_xyz"], r0["idx"], rec["pts_cnt"]
    b, n = xyz.shape[:2]; c = r0["P"].shape[1]
    rows = idx.numel()
    coef = torch.randn(5 * c, device=dev) * 0.1
    dw = torch.zeros(3, c, device=dev)
    da_h = torch.randn(half.rows, c, device=dev)
    da_f = torch.randn(rows, c, device=dev)
    z0 = M.assemble_z0(M.assemble_rows(xyz, new_xyz, idx, pts_cnt=cnt)[0], r0["P"], r0["wx"])
    M.arena_begin(dev)
    t_half = float("nan")  # (the row-major pass with one atomic per real row on the compact rows: removed; 110 / 112 / 80 us at sa2-4 with 32-row pieces)
    if getattr(half, "order", None) is None:
        M.half_sort_rows(half, b * n)
    t_sort = timeit(lambda: M.half_sort_rows(half, b * n))
    t_sorted = timeit(lambda: M.group_linear_backward_half(half, b, n, r0["P"], r0["wx"], da_h, coef, True, dw))
    t_full = timeit(lambda: M.group_linear_backward_assembled(xyz, new_xyz, idx, cnt, r0["P"], r0["wx"], da_f, coef, True, dw))
    t_fullz = timeit(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, z0, da_f, coef, True, dw))
    prev = M.set_deterministic(True)
    M._inverse_of(idx, n)
    t_gather = timeit(lambda: M.group_linear_backward(xyz, new_xyz, idx, cnt, z0, da_f, coef, True, dw))
    M.set_deterministic(prev)
    M.arena_end()
    print("%s: rows %d (half %d)  atomics full/assembled %.1f us  full/stored z %.1f us  gather (stored z, inverse index) %.1f us  atomics half %.1f us  sorted half %.1f us (+ %.1f us bucketing with the geometry)"
          % (name, rows, half.rows, t_full, t_fullz, t_gather, t_half, t_sorted, t_sort), flush=True)
