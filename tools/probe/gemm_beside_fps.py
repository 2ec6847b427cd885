"""One forward GEMM (524288 x 128 -> 128, fwd+bn) alone and beside the sa1 FPS kernel (8 workgroups holding 8 CUs for 1.7 ms), for
several persistent-workgroup caps: 1024 workgroups are 4 per CU on 256 CUs but 4.13 per CU on the 248 that are left."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import _lib as L, mlp as M, synth, tf_sampling
dev = torch.device("cuda:0")
x0 = torch.from_numpy(synth.room_batch(8, 20480, 1)).to(dev)
side = torch.cuda.Stream(device=dev)
rows, c, co = 524288, 128, 128
x = torch.randn(rows, c, device=dev); w = torch.randn(c, co, device=dev) * 0.1
sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
def gemms(n):
    for _ in range(n):
        M.linear_dense(x, w, None, sc, sh, True)
for cap in (1024, 992, 744, 1240, 1488, 2048):
    L.lib().votenet_debug_fast_workgroups(cap, 2 * cap)
    for beside in (False, True):
        gemms(3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if beside:
            with torch.cuda.stream(side):
                for _ in range(3):
                    tf_sampling.farthest_point_sample(2048, x0)
            torch.cuda._sleep(200000)  # let the FPS kernel take its CUs first
        e0.record(); gemms(6); e1.record(); torch.cuda.synchronize()
        print("cap %4d %s: %.4f ms per GEMM" % (cap, "beside FPS" if beside else "alone     ", e0.elapsed_time(e1) / 6), flush=True)
