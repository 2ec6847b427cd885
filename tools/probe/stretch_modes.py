"""Steady-state train step under the combinations of model.GEOMETRY_GRAPHS x model.STRETCH_GRAPH (do two graphs in flight on two streams
get in each other's way?), host threads pinned as bench.py does."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util
_s = importlib.util.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hp = importlib.util.module_from_spec(_s); _s.loader.exec_module(hp)
pinned = None if os.environ.get('NO_PIN') else hp.pin(0)
import torch, gc
from votenet_amd import synth, loss as VL, model as VM
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
modes = [(True, False, True), (True, True, True)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), "pinned to", pinned)
for rep in range(3):
    for gg, sg, sb in modes:
        VM.GEOMETRY_GRAPHS, VM.STRETCH_GRAPH, VM.STRETCH_SEGMENTS = gg, sg, sb
        run(9); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); t1 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("geometry graph %-5s stretch graph %-5s segments %-5s: %.3f ms per step (host enqueue %.3f)" % (gg, sg, sb, dt / 40 * 1e3, (t1 - t0) / 40 * 1e3), flush=True)
