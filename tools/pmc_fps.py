"""The sa1 FPS launch in isolation, for PMC (HBM traffic) collection.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from votenet_amd import synth, tf_grouping as G, tf_sampling as S
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
for _ in range(5):
    S._INDEX_CACHE.clear()
    f = S.farthest_point_sample(2048, x)          # spatial index (5 launches) + sampling kernel
    G.query_ball_point(0.2, 64, x, S.gather_point(x, f))  # over the index the sampling left behind
torch.cuda.synchronize()
