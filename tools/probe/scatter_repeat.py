"""probe: inside real train steps (default mode, launch by launch, weight gradients on their own stream), every pooled input-gradient
call (mlp.pool_dgrad: dense GEMM + arg-max scatter) is issued TWICE on the same inputs; its rows are stored, not accumulated, so the two
results must be bit-equal.  Counts the calls / rows that differ.   python tools/probe/scatter_repeat.py [steps]
(run several at once -- tools/probe/rep3.sh -- to share the GPU between processes)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
import votenet_amd
from votenet_amd import loss as VL, model as VM, synth, mlp as M
dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", "8")), 20480
if os.environ.get("DET") == "1":
    votenet_amd.set_deterministic(True)
VM.STRETCH_GRAPH = False
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000)]
net = VM.VoteNetHotPath(dev, seed=0)
net.init_optimizer()
found = []
calls = [0]
orig = M.pool_dgrad
def twice(*a, **k):
    r0 = orig(*a, **k)
    d0 = r0[0] if isinstance(r0, tuple) else r0
    keep = d0.clone()
    r1 = orig(*a, **k)
    d1 = r1[0] if isinstance(r1, tuple) else r1
    calls[0] += 1
    half = k.get("half")
    ne = (keep != d1).any(dim=1)
    if half is not None:  # rows past the pieces in use are never written
        live = torch.arange(ne.numel(), device=ne.device) < 16 * half.nh_dev.view(())
        ne = ne & live
    found.append((a[0].shape, ne.sum()))
    return r1
M.pool_dgrad = twice
bad_calls = bad_rows = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    net.train_step(xs[i % 2], None, 1, gt=gts[i % 2])
    torch.cuda.synchronize()
    for shape, cnt in found:
        c = int(cnt)
        if c:
            bad_calls += 1
            bad_rows += c
            print("step %d: pool_dgrad on %s rows x %d: %d rows differ between two launches on the same inputs" % (i, shape[0], shape[1], c))
    del found[:]
print("%d of %d pool_dgrad calls differed (%d rows)" % (bad_calls, calls[0], bad_rows))
