"""sa1 geometry kernels alone (spatial index, FPS, ball query) for a rocprofv3 --kernel-trace --stats run (scratch tool)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import torch
from votenet_amd import synth, tf_grouping as G, tf_sampling as S
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
for _ in range(12):
    S._INDEX_CACHE.clear()
    f = S.farthest_point_sample(2048, x)
    c = S.gather_point(x, f)
    G.query_ball_point(0.2, 64, x, c)
torch.cuda.synchronize()
