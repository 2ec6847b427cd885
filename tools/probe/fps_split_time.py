"""config 5's sampling (4 x 80000 -> 2048; also 1 / 8 scenes, 30 000 and 98 304 points): the scene over four workgroups
(fps_bucket_split_kernel) against the L2-resident kernel -- same indices?  ms per call, us per round, fraction of the HBM model
(tf_sampling_g.cu:130-147: 16 B per point and round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from votenet_amd import _lib as L_, synth, tf_sampling
dev = torch.device("cuda:0")
lib = L_.lib()
lib.votenet_debug_fps_split_timeouts.restype = __import__("ctypes").c_uint
def timeit(fn, it=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
SP = int(os.environ.get("SPLIT", "1"))
for (b, n, m) in [(4, 80000, 2048), (1, 80000, 2048), (8, 80000, 2048), (4, 30000, 1024), (2, 98304, 512), (9, 40000, 256)]:
    x = torch.from_numpy(synth.room_batch(b, n, 7)).to(dev)
    res = {}
    for on in (0, int(os.environ.get("SPLIT", "1"))):
        lib.votenet_debug_fps_split(on)
        idx = tf_sampling.farthest_point_sample(m, x)
        torch.cuda.synchronize()
        t = timeit(lambda: tf_sampling.farthest_point_sample(m, x))
        res[on] = (idx.clone(), t)
    same = torch.equal(res[0][0], res[SP][0])
    model = b * n * 16.0 * (m - 1)
    print("%d x %6d -> %4d: L2-resident %.3f ms (%.3f us per round, %.3f of 8 TB/s)   split %.3f ms (%.3f us per round, %.3f)   indices equal: %s   timeouts %d" % (
        b, n, m, res[0][1], res[0][1] * 1e3 / (m - 1), model / (res[0][1] * 1e-3) / 8e12, res[SP][1], res[SP][1] * 1e3 / (m - 1),
        model / (res[SP][1] * 1e-3) / 8e12, same, lib.votenet_debug_fps_split_timeouts()))
lib.votenet_debug_fps_split(0)
