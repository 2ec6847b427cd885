"""A/B of the sa1 farthest-point sampling kernels alone on the GPU (scratch tool): lock-step fps_bucket_kernel vs the
director / worker fps_async_kernel, room scenes and the uniform cube, 8 x 20480 -> 2048."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import _lib, synth, tf_sampling as S
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
L = _lib.lib()
L.votenet_fps_debug_async.restype = None
B, n, m = 8, int(os.environ.get("N", 20480)), int(os.environ.get("M", 2048))
alg = B * (m - 1) * n * 16 + B * n * 12 + B * m * 4
for kind, x in (("room", synth.room_batch(B, n, 1000)), ("uniform", synth.uniform_batch(B, n, 1000))):
    x = torch.from_numpy(x).to(dev)
    res = {}
    for mode in (0, 1):
        L.votenet_fps_debug_async(mode)
        ms = gpu_ms(lambda: S.farthest_point_sample(m, x), it=10, warm=3)
        res[mode] = (ms, S.farthest_point_sample(m, x))
        print("%-8s %-10s %.4f ms  %.3f us/round  effective %.0f GB/s  frac %.3f" % (kind, "async" if mode else "lock-step", ms, ms * 1e3 / (m - 1),
                                                                                  alg / ms / 1e6, alg / ms / 1e6 / 8000))
    import ctypes
    st = (ctypes.c_ulonglong * 8)()
    L.votenet_fps_async_stats(st)
    print("   async scene 0: rounds %d, pulls %d, spin iterations %d, cycles waiting %d of %d (s_memtime ticks); arg-max sections %d, publish + box tests %d" % tuple(st[:7]))
    print("   same indices:", bool(torch.equal(res[0][1], res[1][1])))
L.votenet_fps_debug_async(1)
