"""Where the time of a static-stretch GEMM goes: the forward GEMM + statistics of 2048-8192 rows x cin -> 256 at cin = 32 ... 512,
alone on the GPU, 50 launches captured into one HIP graph and replayed (each waits for the one before it, as in the stretch's chain).  A launch is
T(cin) = T_fixed + (cin / 16) * t_slab: the fit gives the fixed part (dispatch, per-channel tables, first loads, statistics atomics,
coefficient tail) and the dependent chain's cost per 16-deep slab; beside them the time the same multiply-adds take at the rate the big layers
reach (100 TFLOP/s).   python tools/probe/stretch_chain.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import numpy as np, torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):  # a library variant under tools/probe/lib (tools/probe/build_variant.sh)
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, it=50, rounds=5):
    """us per launch of `it` calls captured into ONE HIP graph and replayed (the host is out of the picture, as in the step's stretch: the
    launch-by-launch loop is bound by ~12 us of host work per call); the minimum of a few replays."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            for _ in range(it):
                fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / it * 1e3)
    return best


print("forward GEMM + BatchNorm statistics, fp16 x 2 operands, cout = 256; us per launch in a dependent chain of launches")
print("%6s | %s | %8s %8s | %s" % ("rows", "  ".join("cin=%-4d" % c for c in (32, 64, 128, 256, 512)), "T_fixed", "t_slab", "cin=512: chain / math at 100 TFLOP/s"))
for rows in (2048, 4096, 8192, 16384):
    ts = []
    for cin in (32, 64, 128, 256, 512):
        x = torch.randn(rows, cin, generator=g).to(dev)
        w = (torch.randn(cin, 256, generator=g) * 0.1).to(dev)
        sc, sh = torch.rand(cin, generator=g).to(dev) + 0.5, torch.randn(cin, generator=g).to(dev) * 0.1
        img = M.SplitImages([w], pieces=2)
        img.refresh()
        st = torch.zeros(512, dtype=torch.float64, device=dev)
        ts.append(timeit(lambda: M.linear_dense(x, w, None, sc, sh, True, want_stats=True)))
        img.close()
    nk = np.array([2, 4, 8, 16, 32], dtype=np.float64)
    slope, icpt = np.polyfit(nk, np.array(ts), 1)
    math = 2.0 * rows * 512 * 256 / 100e12 * 1e6
    print("%6d | %s | %8.1f %8.2f | %.1f us of chain, %.1f us of math" % (rows, "  ".join("%7.1f " % t for t in ts), icpt, slope, 32 * slope, math))
print("(each launch includes one zero fill of its statistics buffer -- a second graph node, ~2-3 us -- and the gap between dependent graph nodes)")
