"""Not a test (not collected): the REFERENCE's own device kernels (oracle/_ref/libref_*_gpu.so, see oracle/ref_gpu.py) timed on this
GPU beside the product's kernels for the same calls -- the sa1 level of the headline workload (8 x 20480 room scenes -> 2048 centres,
r = 0.2, K = 64) and one config-5 scene.  Inputs resident, wall clock around a synchronised call, best of 5.

    gpurun -- python tests/time_reference_kernels.py > gpurun_out/reference_kernels_time.txt
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from oracle import ref_gpu as R  # noqa: E402
from votenet_amd import synth, tf_grouping, tf_sampling  # noqa: E402


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


def main():
    assert R.available()
    dev = torch.device("cuda:0")
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    S, G = R._lib("sampling"), R._lib("grouping")
    print("device: %s" % torch.cuda.get_device_name(0))
    print("%-58s %12s %12s %8s" % ("call", "reference ms", "product ms", "ratio"))
    for name, xyz, m in (("8 x 20480 room scenes -> 2048 (sa1)", synth.room_batch(8, 20480, 1000), 2048),
                         ("1 x 80000 scan scene -> 2048 (config 5)", synth.room_batch(1, 80000, 5000, size=(8.0, 3.0, 8.0), nbox=(15, 25)), 2048)):
        x = torch.from_numpy(xyz).to(dev)
        b, n, _ = x.shape
        temp = torch.empty((32, n), dtype=torch.float32, device=dev)
        out = torch.zeros((b, m), dtype=torch.int32, device=dev)
        t_ref = best(lambda: S.ref_gpu_farthest_point_sample(b, n, m, p(x), p(temp), p(out)))
        tf_sampling.farthest_point_sample(m, x)
        t_own = best(lambda: tf_sampling.farthest_point_sample(m, x))
        own = tf_sampling.farthest_point_sample(m, x)
        assert torch.equal(own, out)
        print("%-58s %12.3f %12.3f %8.1f" % ("farthest_point_sample " + name, t_ref, t_own, t_ref / t_own))
        if b == 8:
            centres = tf_sampling.gather_point(x, own).contiguous()
            idx = torch.zeros((b, m, 64), dtype=torch.int32, device=dev)
            cnt = torch.zeros((b, m), dtype=torch.int32, device=dev)
            t_ref = best(lambda: G.ref_gpu_query_ball_point(b, n, m, ctypes.c_float(0.2), 64, p(x), p(centres), p(idx), p(cnt)))
            tf_grouping.query_ball_point(0.2, 64, x, centres)
            t_own = best(lambda: tf_grouping.query_ball_point(0.2, 64, x, centres))
            oi, oc = tf_grouping.query_ball_point(0.2, 64, x, centres)
            assert torch.equal(oi, idx) and torch.equal(oc, cnt)
            print("%-58s %12.3f %12.3f %8.1f" % ("query_ball_point r=0.2 K=64, same clouds", t_ref, t_own, t_ref / t_own))
            g = torch.zeros((b, m, 64, 3), dtype=torch.float32, device=dev)
            t_ref = best(lambda: G.ref_gpu_group_point(b, n, 3, m, 64, p(x), p(idx), p(g)))
            t_own = best(lambda: tf_grouping.group_point(x, oi))
            assert torch.equal(tf_grouping.group_point(x, oi), g)
            print("%-58s %12.3f %12.3f %8.1f" % ("group_point xyz, same indices", t_ref, t_own, t_ref / t_own))
    print("reference = tf_sampling_g.cu / tf_grouping_g.cu as they are, hipcc -O2 -ffp-contract=off for gfx950, launch shapes of their own")
    print("launchers (32 x 512 threads for the sampling, b x 256 for the grouping); results identical to the product's (asserted above).")


if __name__ == "__main__":
    main()
