"""Same-box A/B of the train step over several configurations, alternating (the GPU's clock drifts over a process's life: only
neighbouring lines compare):  python tools/ab_multi.py   (scratch tool; edit `cfgs`)"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth, mlp as M, pointnet2 as P, _lib as L
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def caps(a, b):
    L.lib().votenet_debug_fast_workgroups(a, b)
cfgs = [("PREFETCH_AFTER 2", lambda: setattr(VM, "PREFETCH_AFTER", 2)),
        ("PREFETCH_AFTER 3", lambda: setattr(VM, "PREFETCH_AFTER", 3)),
        ("PREFETCH_AFTER 4", lambda: setattr(VM, "PREFETCH_AFTER", 4)),
        ("PREFETCH_AFTER 1", lambda: setattr(VM, "PREFETCH_AFTER", 1))]
for rep in range(3):
    for name, setup in cfgs:
        setup(); net.drop_graphs()
        run(8); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("%-28s %.3f ms per step" % (name, dt / 40 * 1e3), flush=True)
