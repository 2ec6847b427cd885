"""ThreeInterpolateGrad at the two FP shapes of the step: scatter-add with atomics against the gather-sum over the taps' inverse index."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import mlp as M, synth, tf_interpolate as TI, tf_sampling as S
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
l2 = S.gather_point(x, S.farthest_point_sample(1024, x)).contiguous()
l3 = S.gather_point(l2, S.farthest_point_sample(512, l2)).contiguous()
l4 = S.gather_point(l3, S.farthest_point_sample(256, l3)).contiguous()
for name, x1, x2 in (("fp2 (1024 <- 512)", l2, l3), ("fp1 (512 <- 256)", l3, l4)):
    dist, idx = TI.three_nn(x1, x2)
    w = TI.three_nn_weights(dist)
    b, n, m, c = 8, x1.shape[1], x2.shape[1], 256
    wide = torch.randn(b, n, 512, device=dev)
    g = wide[:, :, :c]
    t_at = timeit(lambda: TI.three_interpolate_grad_raw(m, idx, w, g))
    inv = M.inverse_index(idx, m)
    gc = g.contiguous()
    t_ga = timeit(lambda: M.csr_gather_sum(gc.view(b * n, c), inv, b * m, weight=w, div=3))
    ref = TI.three_interpolate_grad_raw(m, idx, w, g)
    got = M.csr_gather_sum(gc.view(b * n, c), inv, b * m, weight=w, div=3).view(b, m, c)
    print("%s: atomics %.1f us, gather over the inverse index %.1f us (contiguous input; both incl. ~6 us launch + allocation), max diff %.2e"
          % (name, t_at, t_ga, (ref - got).abs().max().item()), flush=True)
