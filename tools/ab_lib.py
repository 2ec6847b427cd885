"""Same-box A/B of the train step for a debug hook of the library:  python tools/ab_lib.py votenet_debug_wgrad_bf3 0 1   (scratch tool)"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)  # as bench.py, before torch is imported
import torch
from votenet_amd import loss as VL, model as VM, synth, _lib as L
hook = getattr(L.lib(), sys.argv[1])
hook.restype = None
vals = [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def run(k):
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
for rep in range(3):
    for v in vals:
        hook(v); net.drop_graphs()  # captured graphs keep the kernels of the other setting
        run(6); torch.cuda.synchronize(); gc.collect(); gc.disable()
        t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
        print("%s(%d): %.3f ms per step" % (sys.argv[1], v, dt / 40 * 1e3), flush=True)
