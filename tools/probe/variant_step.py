"""The pipelined train step on a library VARIANT (tools/probe/lib/libvotenet_$VARIANT.so; empty = the built one): ms per step.
Run the variants alternately from one shell loop: process-to-process spread is ~1 %, box-to-box more."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import loss as VL, model as VM, synth
import importlib
for t in os.environ.get("TOGGLES", "").split():  # e.g. TOGGLES="pointnet2.ASSEMBLED_DECOMPOSED=False mlp.NARROW_MASK=False"
    name, val = t.split("=")
    modname, attr = name.rsplit(".", 1)
    setattr(importlib.import_module("votenet_amd." + modname), attr, eval(val))
for h in os.environ.get("HOOKS", "").split():  # e.g. HOOKS="votenet_debug_fps_lds_floor=131072"
    name, val = h.split("=")
    getattr(L_.lib(), name)(*[int(v) for v in val.split(",")])
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
main = torch.cuda.Stream(priority=int(os.environ["MAIN_PRIORITY"])) if os.environ.get("MAIN_PRIORITY") else None
def run(k):
    if main is not None:
        with torch.cuda.stream(main):
            for i in range(k):
                net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
        return
    for i in range(k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
run(10); torch.cuda.synchronize(); gc.collect(); gc.disable()
from votenet_amd import tf_sampling
res = []
for rep in range(3):
    t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e3)
tf_sampling.PROFILE_EVENTS = []
run(3); torch.cuda.synchronize()
ev = [e0.elapsed_time(e1) for (e0, e1, b_, n_, m_) in tf_sampling.PROFILE_EVENTS if n_ == 20480]
tf_sampling.PROFILE_EVENTS = None
print("sa1 FPS inside the step: %s ms" % " ".join("%.3f" % v for v in ev))
print("variant %-10s %-60s ms per step: %s" % (os.environ.get("VARIANT") or "(built)", os.environ.get("TOGGLES", "") + " " + os.environ.get("HOOKS", "") + (" main stream priority " + os.environ["MAIN_PRIORITY"] if os.environ.get("MAIN_PRIORITY") else ""), " ".join("%.3f" % v for v in res)))
