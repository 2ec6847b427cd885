"""config 5's sampling on the split kernel: whole round / without the exchange between the workgroups (wrong indices, timing only),
and the spatial index alone -- where does a round's time go?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from votenet_amd import _lib as L_, synth, tf_sampling
dev = torch.device("cuda:0")
lib = L_.lib()
def timeit(fn, it=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
b, n, m = 4, 80000, 2048
x = torch.from_numpy(synth.room_batch(b, n, 7)).to(dev)
for mode, name in ((0, "L2-resident kernel"), (1, "split 4 workgroups x 12 waves"), (2, "  ... no exchange between workgroups"),
                   (3, "split 12 workgroups x 4 waves"), (4, "  ... no exchange between workgroups"),
                   (5, "split 6 workgroups x 8 waves"), (6, "  ... no exchange between workgroups")):
    lib.votenet_debug_fps_split(mode)
    t = timeit(lambda: tf_sampling.farthest_point_sample(m, x))
    t1 = timeit(lambda: tf_sampling.farthest_point_sample(2, x))
    print("%-42s %.3f ms; with m = 2 (index build + one round): %.3f ms -> %.3f us per round" % (name, t, t1, (t - t1) * 1e3 / (m - 2)))
lib.votenet_debug_fps_split(0)
