"""The un-indexed ball query at the sizes below sa1 (sa3: 1024 -> 512, sa4: 512 -> 256, proposal: 1024 votes -> 256): four waves x eight
groups against sixteen waves with the cloud as one super-chunk (votenet_debug_ball_query_small)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import _lib as L, synth, tf_grouping as G, tf_sampling as S
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
hook = L.lib().votenet_debug_ball_query_small
hook.restype = None
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for n, m, r in ((2048, 1024, 0.4), (1024, 512, 0.8), (512, 256, 1.2), (1024, 256, 0.3)):
    c = S.gather_point(x, S.farthest_point_sample(n, x)).contiguous()
    q = S.gather_point(c, S.farthest_point_sample(m, c)).contiguous()
    row = []
    for form in (4, 16, 0):
        hook(form)
        row.append(timeit(lambda: G.query_ball_point(r, 64, c, q)))
    hook(0)
    print("n %5d m %4d r %.1f: four waves %.1f us, sixteen x eight %.1f us, one super-chunk %.1f us (incl. ~8 us of launch + allocation)" % (n, m, r, *row), flush=True)
