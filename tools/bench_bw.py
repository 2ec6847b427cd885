"""Achievable HBM bandwidth on this box with plain torch kernels (calibration for the roofline discussion)."""
import torch
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
n = 1 << 27  # 512 MB fp32
x = torch.randn(n, device=dev); y = torch.empty_like(x)
t = timeit(lambda: y.fill_(1.0), it=20); print("fill   512 MB: %.3f ms  %.2f TB/s (write)" % (t, n * 4 / t / 1e9))
t = timeit(lambda: y.copy_(x), it=20); print("copy   512 MB: %.3f ms  %.2f TB/s (read+write)" % (t, 2 * n * 4 / t / 1e9))
t = timeit(lambda: x.sum(), it=20); print("sum    512 MB: %.3f ms  %.2f TB/s (read)" % (t, n * 4 / t / 1e9))
t = timeit(lambda: torch.add(x, y, out=y), it=20); print("add  2r+1w    : %.3f ms  %.2f TB/s" % (t, 3 * n * 4 / t / 1e9))
