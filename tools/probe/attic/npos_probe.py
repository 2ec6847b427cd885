"""probe: positives / negatives of the loss per training step at a test's configuration (a batch without a positive proposal has a NaN
loss by definition, model.py's reduce_mean of an empty set): is a multi-step test sitting next to that edge?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import dp, synth, loss as VL, model as VM
dev = torch.device("cuda", 0)
for rank in (0, 1):
    seeds = dp.scene_seeds(rank, 2, base=300)
    x = torch.from_numpy(synth.room_batch(2, 4096, seeds[0])).to(dev)
    gt = VL.gt_to_device(synth.room_gt(2, 4096, seeds[0]), dev)
    net = VM.VoteNetHotPath(dev, seed=10, npoints=(512, 256, 128, 64))
    net.init_optimizer(1e-3)
    tr = []
    for i in range(6):
        net.train_step(x, gt=gt)
        l = net.last_losses.cpu().tolist()
        tr.append((round(l[0], 4), l[10], l[11]))
    print("rank", rank, tr)
