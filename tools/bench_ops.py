"""Per-op timing of the geometry kernels at VoteNet layer shapes (scratch tool, GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from votenet_amd import tf_sampling as S, tf_grouping as G, tf_interpolate as I

dev = torch.device("cuda:0")
def timeit(fn, it=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

B = 8
rng = np.random.default_rng(0)
xyz = torch.from_numpy(rng.random((B, 20480, 3), dtype=np.float32) * 5).to(dev)
for (n, m, r, K) in [(20480, 2048, 0.2, 64), (2048, 1024, 0.4, 64), (1024, 512, 0.8, 64), (512, 256, 1.2, 64)]:
    x = xyz[:, :n].contiguous()
    t = timeit(lambda: S.farthest_point_sample(m, x))
    alg = B * (m - 1) * n * 16 + B * n * 12 + B * m * 4
    print("fps   n=%5d m=%4d : %8.3f ms  effective %8.1f GB/s" % (n, m, t, alg / t / 1e6))
    fi = S.farthest_point_sample(m, x); nx = S.gather_point(x, fi)
    t = timeit(lambda: G.query_ball_point(r, K, x, nx))
    alg = B * m * n * 12 + B * m * (K + 1) * 4
    idx, cnt = G.query_ball_point(r, K, x, nx)
    print("ballq n=%5d m=%4d : %8.3f ms  effective %8.1f GB/s  mean cnt %.1f" % (n, m, t, alg / t / 1e6, cnt.float().mean().item()))
    for c in (3, 128):
        pts = torch.randn(B, n, c, device=dev)
        t = timeit(lambda: G.group_point(pts, idx))
        by = B * m * K * 4 + B * m * K * c * 4 + B * n * c * 4
        print("group n=%5d m=%4d c=%3d: %8.3f ms  %8.1f GB/s" % (n, m, c, t, by / t / 1e6))
x1 = xyz[:, :1024].contiguous(); x2 = xyz[:, :512].contiguous()
t = timeit(lambda: I.three_nn(x1, x2)); print("three_nn 1024x512: %.3f ms" % t)

# ---- predict tail: 3D IoU matrix + NMS at config-3 size (8 scenes x 256 proposals)
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
from votenet_amd import tf_nms3d as NM
c = cases.nms_random(b=8, n=256, seed=33, room=6.0)
bb, sc, ob = (torch.from_numpy(c[k]).to(dev) for k in ("bboxes", "scores", "objectiveness"))
t = timeit(lambda: NM.iou3d_matrix(bb)); print("iou3d_matrix 8x256x256: %.3f ms (%.0f pair IoUs/us)" % (t, 8 * 256 * 256 / t / 1e3))
t = timeit(lambda: NM.NMS3D(bb, sc, ob, 0.25)); print("NMS3D 8x256 thr 0.25 (incl. count readback): %.3f ms, kept %d" % (t, len(NM.NMS3D(bb, sc, ob, 0.25))))
