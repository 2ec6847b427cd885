"""Train step with the geometry of the next batch prefetched one step ahead against two steps ahead (the ring of geometry graphs holds
three): two ahead, the piece counts are on the host long before the step that needs them starts -- no wait at its start."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
from votenet_amd import hostpin; hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000, 1300000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000, 1300000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k, ahead):
    for i in range(k):
        net.train_step(xs[i % 4], gt=gts[i % 4], next_x=[xs[(i + a) % 4] for a in range(1, ahead + 1)])
for rep in range(3):
    for ahead in (1, 2):
        run(8, ahead); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40, ahead); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("lookahead %d: %.3f ms per step" % (ahead, dt / 40 * 1e3), flush=True)
