"""The train step with every kernel on ONE stream and its geometry given (no prefetch chain, no weight-gradient stream): a
--kernel-trace of this run gives each kernel's duration ALONE on the GPU in step context -- the table of what the step is made of
when nothing overlaps (tools/rocpd_stats.py on the .db).  Prints the wall time of that serial step beside the pipelined one.
GPU box only:   rocprofv3 --kernel-trace -d out -o s -- python3 tools/serial_step.py"""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)  # as bench.py, before torch is imported
import torch
from votenet_amd import loss as VL, model as VM, synth
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):  # a library variant under tools/probe/lib (same-box A/B of kernel alone-times)
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
for h in os.environ.get("HOOKS", "").split():  # e.g. HOOKS="votenet_debug_bn_reduce_passes=16"
    name, val = h.split("=")
    getattr(L_.lib(), name)(*[int(v) for v in val.split(",")])
dev = torch.device("cuda:0")
seeds = (1000, 500000, 900000)
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in seeds]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in seeds]
VM.GEOMETRY_GRAPHS = False  # the geometry of a batch is computed once, launch by launch, and handed to every step that uses it
net = VM.VoteNetHotPath(dev, seed=0)
net.overlap_wgrad = False
for x in xs:
    net.prefetch_geometry(x)
torch.cuda.synchronize()
saved = dict(net._prefetched)
K = int(os.environ.get("STEPS", "12"))
def serial(k):
    for i in range(k):
        x = xs[i % 3]
        net._prefetched[id(x)] = saved[id(x)]
        net.train_step(x, gt=gts[i % 3])
serial(4); torch.cuda.synchronize(); gc.disable()
t0 = time.perf_counter(); serial(K); torch.cuda.synchronize()
print("serial step (one stream, geometry given): %.3f ms per step over %d steps" % ((time.perf_counter() - t0) / K * 1e3, K))
