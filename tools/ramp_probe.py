import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from votenet_amd import synth, loss as VL
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
gt = VL.gt_to_device(synth.room_gt(8, 20480, 1000), dev)
net = VoteNetHotPath(dev, seed=0)
ts = []
for i in range(60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    net.train_step(x, gt=gt)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
