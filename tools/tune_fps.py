"""Sweep brute-force FPS configurations (scratch tuning tool, GPU box only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from votenet_amd import tf_sampling as S, _lib
L = _lib.lib()
dev = torch.device("cuda:0")
def timeit(fn, it=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
rng = np.random.default_rng(0)
for n, m in [(512, 256), (1024, 512), (1024, 256), (2048, 1024), (4096, 1024)]:
    x = torch.from_numpy(rng.random((8, n, 3), dtype=np.float32) * 5).to(dev)
    L.votenet_fps_debug_config(0, 0)
    ref = S.farthest_point_sample(m, x)
    res = []
    for nw in (1, 2, 4, 8, 16):
        for p in (1, 2, 4, 8, 16):
            if 64 * nw * p < n or 64 * nw * p > 4 * n or (nw, p) in [(1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (4, 1), (8, 16), (16, 8), (16, 16)]:
                continue
            L.votenet_fps_debug_config(nw, p)
            out = S.farthest_point_sample(m, x)
            ok = bool((out == ref).all())
            res.append((timeit(lambda: S.farthest_point_sample(m, x)), nw, p, ok))
    L.votenet_fps_debug_config(0, 0)
    print("n=%d m=%d: " % (n, m) + "  ".join("(%d,%d)=%.3f%s" % (nw, p, t, "" if ok else "!") for t, nw, p, ok in sorted(res)))
