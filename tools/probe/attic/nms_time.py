"""NMS3D alone on the GPU at the predict tower's size (8 x 256 boxes) and at larger sets: ms per call.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from votenet_amd import tf_nms3d
dev = torch.device("cuda:0")
def timeit(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
def boxes(b, n, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((b, n, 1, 3)) * 6
    s = rng.random((b, n, 1, 3)) * 1.5 + 0.3
    ang = rng.random((b, n)) * np.pi
    corners = np.array([[dx, dy, dz] for dz in (-1, 1) for dx, dy in ((-1, -1), (1, -1), (1, 1), (-1, 1))], dtype=np.float64) * 0.5
    p = corners[None, None] * s
    ca, sa = np.cos(ang)[..., None], np.sin(ang)[..., None]
    x = p[..., 0] * ca - p[..., 1] * sa; y = p[..., 0] * sa + p[..., 1] * ca
    bb = np.stack([x, y, p[..., 2]], -1) + c
    return bb.astype(np.float32), rng.random((b, n)).astype(np.float32), rng.normal(size=(b, n, 2)).astype(np.float32)
for b, n in [(8, 256), (8, 512), (2, 1024)]:
    bb, sc, ob = boxes(b, n, 1)
    B, S, O = (torch.from_numpy(a).to(dev) for a in (bb, sc, ob))
    out = tf_nms3d.NMS3D(B, S, O, 0.25)
    print("%d x %d boxes: %.4f ms per call, %d kept" % (b, n, timeit(lambda: tf_nms3d.NMS3D(B, S, O, 0.25)), int(out.shape[0])), flush=True)
