import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from votenet_amd import synth, tf_sampling as S, tf_grouping as G
from votenet_amd.model import VoteNetHotPath
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(4, 80000, 5)).to(dev)
for m in (1024, 2048):
    t = timeit(lambda: S.farthest_point_sample(m, x), it=3, warm=1)
    print("FPS 4x80000 -> %d: %.2f ms" % (m, t))
net = VoteNetHotPath(dev)
t = timeit(lambda: net.forward(x), it=3, warm=1)
print("forward 4x80000 (default npoints): %.2f ms" % t)
