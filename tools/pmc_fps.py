"""The sa1 FPS launch in isolation, for PMC (HBM traffic) collection.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import synth, tf_sampling as S
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
for _ in range(5):
    S.farthest_point_sample(2048, x)
torch.cuda.synchronize()
