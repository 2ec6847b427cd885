"""The inference tower of BASELINE config 3: forward + box decode + 3D NMS (model.py:98-139) on 8 x 20 480-point scenes."""
import sys, os, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path[:0] = [R, R + "/tools"]
from votenet_amd import synth
from votenet_amd.model import VoteNetHotPath
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
i = [0]
def run2(pipe, sync):
    k = i[0]; i[0] += 1
    return net.predict(xs[k % 3], 0.25, next_x=[xs[(k + 1) % 3], xs[(k + 2) % 3]] if pipe else None, sync=sync)
for pipe in (False, True):
    for sync in (True, False):
        print("predict, 8 scenes: geometry prefetch %-5s kept list sized on the host %-5s: %.3f ms per call" % (pipe, sync, timeit(lambda: run2(pipe, sync), it=30, warm=6)))
def fw():
    k = i[0]; i[0] += 1
    return net.forward(xs[k % 3], next_x=[xs[(k + 1) % 3], xs[(k + 2) % 3]])
print("  forward only, prefetch: %.3f ms" % timeit(fw, it=30, warm=6))
