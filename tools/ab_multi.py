"""Same-box A/B over several (net attribute, module toggle) configurations:  python tools/ab_multi.py   (scratch tool)"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth, mlp as M
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
cfgs = [("default", False, False), ("inline wgrad of sa2/sa1", True, False), ("inline + SPLIT_ADHOC", True, True), ("SPLIT_ADHOC only", False, True)]
for rep in range(3):
    for name, inline, adhoc in cfgs:
        net.inline_wgrad_tail = inline
        M.SPLIT_ADHOC = adhoc
        run(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("%-28s %.3f ms per step" % (name, dt / 40 * 1e3), flush=True)
