"""Same-box A/B of the train step for an attribute of the net object:  python tools/ab_net.py overlap_wgrad True False   (scratch tool)"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth
attr = sys.argv[1]
vals = [eval(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
for rep in range(3):
    for v in vals:
        setattr(net, attr, v)
        run(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("net.%s = %r: %.3f ms per step" % (attr, v, dt / 40 * 1e3), flush=True)
