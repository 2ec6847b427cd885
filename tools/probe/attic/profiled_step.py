"""What the ONE instrumented step of a short bench run (HIP events around the sampling / ball-query launches: the geometry chain of the next
batch is enqueued launch by launch) costs itself and its follower: per-step GPU and host times around it."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth, tf_sampling, tf_grouping
dev = torch.device("cuda:0")
xs = [torch.from_numpy(synth.room_batch(8, 20480, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, s), dev) for s in (1000, 500000, 900000)]
net = VM.VoteNetHotPath(dev, seed=0)
def step(i):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
for i in range(12): step(i)
torch.cuda.synchronize(); gc.collect(); gc.disable()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
host, mallocs = [], []
def ndev():
    st = torch.cuda.memory_stats()
    return st.get("num_device_alloc", 0), st.get("num_device_free", 0), st.get("num_alloc_retries", 0)
marks[0].record()
for k in range(60):
    i = 12 + k
    if k == 9: tf_sampling.PROFILE_EVENTS, tf_grouping.PROFILE_EVENTS = [], []
    if k == 10: tf_sampling.PROFILE_EVENTS = tf_grouping.PROFILE_EVENTS = None
    a0 = ndev(); t0 = time.perf_counter(); step(i); host.append((time.perf_counter() - t0) * 1e3); a1 = ndev(); mallocs.append("%d/%d" % (a1[0] - a0[0], a1[1] - a0[1]))
    marks[k + 1].record()
torch.cuda.synchronize()
print("GPU ms :", " ".join("%.2f" % marks[k].elapsed_time(marks[k + 1]) for k in range(60)))
print("host ms:", " ".join("%.2f" % h for h in host))
print("hipMalloc/hipFree calls of the allocator per step:", " ".join(mallocs), "| reserved MB", torch.cuda.memory_reserved() >> 20)
