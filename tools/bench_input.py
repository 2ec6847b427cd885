"""Input pipeline (votenet_subsample_augment + votenet_augment_boxes) at the BASELINE shape: 8 scenes x 50 000 raw depth
points -> 20 480, against the numpy restatement of the reference's per-scene code on one host core."""
import sys, time, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
import numpy as np
from votenet_amd import input_pipeline as IP
from oracle import oracle_input as OI
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
b, n_raw, n_out = 8, 50000, 20480
rng = np.random.default_rng(0)
aug = IP.draw_augmentation(b, np.random.RandomState(0))
off = np.arange(b + 1) * n_raw
for dt, cols in ((np.float32, 3), (np.float64, 6)):
    rawn = rng.normal(size=(b * n_raw, cols)).astype(dt)
    raw = torch.from_numpy(rawn).to(dev)
    ch = torch.from_numpy(IP.draw_choice(np.random.RandomState(1), [n_raw] * b, n_out)).to(dev)
    row = 3 * rawn.itemsize
    alg = b * n_out * (row + 12)
    for name, c in (("device draw", None), ("host choice", ch)):
        extra = b * n_out * 4 if c is not None else 0
        ms = timeit(lambda: IP.subsample_augment(raw, off, n_out, aug, c, seed=3), it=50)
        print("points %s cols=%d %-11s: %.4f ms  %.0f GB/s algorithmic (%.1f MB)" % (dt.__name__, cols, name, ms, (alg + extra) / ms / 1e6, (alg + extra) / 1e6))
    t = time.perf_counter()
    for s in range(b):
        OI.augment_points(rawn[s * n_raw:(s + 1) * n_raw], np.random.RandomState(s).choice(n_raw, n_out, replace=False), aug.flip_x[s], aug.flip_z[s], aug.angle[s], aug.scale[s], literal=True)
    print("  numpy, one core (choice + augmentation as the reference writes them): %.2f ms / batch" % ((time.perf_counter() - t) * 1e3))
cnt = rng.integers(3, 12, b)
pk = lambda a: IP.pack_ragged(a, dev)
dc, boff = pk([rng.normal(size=(c, 3)) for c in cnt]); ds, _ = pk([np.abs(rng.normal(size=(c, 3))) + .3 for c in cnt])
dh, _ = pk([rng.uniform(-3, 3, c) for c in cnt]); dk, _ = pk([rng.integers(0, 10, c).astype(np.int32) for c in cnt])
print("boxes: %.4f ms / batch" % timeit(lambda: IP.augment_boxes(dc, ds, dh, dk, boff, aug), it=50))
