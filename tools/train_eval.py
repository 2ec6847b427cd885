"""End-to-end check on synthetic rooms (GPU box only): train the hot path with the reference's loss, then run the predict
tower (decode -> 3D NMS) on held-out scenes and report mAP@0.25 / @0.5 with the reference's evaluator logic.
    python tools/train_eval.py [steps] [train_batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
_spec = importlib.util.spec_from_file_location("votenet_hostpin", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "votenet_amd", "hostpin.py"))
hostpin = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(hostpin)  # by path: the package's __init__ would import torch first
hostpin.pin(0)  # as bench.py: the host threads on eight cores of the GPU's NUMA node
import numpy as np, torch
from votenet_amd import evaluator as E, loss as VL, synth
from votenet_amd.model import VoteNetHotPath

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
B, n = 8, 20480
net = VoteNetHotPath(dev, seed=0)
net.init_optimizer(1e-3)
xs = [torch.from_numpy(synth.room_batch(B, n, 5000 + B * i)).to(dev) for i in range(nb)]
gts = [VL.gt_to_device(synth.room_gt(B, n, 5000 + B * i), dev) for i in range(nb)]
val_x = [torch.from_numpy(synth.room_batch(B, n, 90000 + B * i)).to(dev) for i in range(4)]
val_gt = [E.gt_for_eval(synth.room_gt(B, n, 90000 + B * i)) for i in range(4)]


def evaluate():
    res = {}
    for thr in (0.25, 0.5):
        aps = []
        for x, g in zip(val_x, val_gt):
            pred = net.predict(x, 0.25)
            aps.append(E.eval_det(pred, g, thr)[1])
        res[thr] = float(np.nanmean(aps))
    return res


t0 = time.time()
print("step 0: mAP", evaluate())
for i in range(steps):
    net.train_step(xs[i % nb], gt=gts[i % nb], next_x=xs[(i + 1) % nb])  # geometry of the next batch under this step
    if (i + 1) % 100 == 0:
        l = net.last_losses.cpu().numpy()
        print("step %d  cost %.3f  vote %.3f obj %.3f box %.3f sem %.3f  pos %d  (%.1f s)" % (i + 1, l[0], l[1], l[2], l[9], l[8], int(l[10]),
                                                                                              time.time() - t0))
    if (i + 1) % 300 == 0:
        print("step %d: mAP" % (i + 1), evaluate())
