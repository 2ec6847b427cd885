"""bf16 x 3 split-operand GEMM (votenet_debug_fast_bf3) against the fp32 MFMA kernel: error vs float64 and time per launch at the
forward shapes of the train step (scratch tool, GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from votenet_amd import mlp as M, _lib as L
dev = torch.device("cuda:0")
lib = L.lib()
def timeit(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
torch.manual_seed(0)
shapes = [(1048576, 64, 64, 0), (1048576, 64, 128, 64), (524288, 128, 128, 0), (524288, 128, 256, 64), (262144, 128, 256, 64),
          (131072, 128, 128, 0), (131072, 128, 256, 64), (8192, 256, 256, 0), (8192, 512, 256, 0), (4096, 256, 128, 0), (2048, 128, 128, 0)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
print("%-28s %9s %9s %8s %8s | %10s %10s" % ("rows,cin,cout,pool", "fp32 ms", "bf3 ms", "fp32 TF", "bf3 TF", "err fp32", "err bf3"))
for rows, cin, cout, pool in shapes:
    x = torch.randn(rows, cin, device=dev) * 2 + 0.3
    w = torch.randn(cin, cout, device=dev) * 0.1
    b = torch.randn(cout, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.2
    img = M.SplitImages([w]); img.refresh()
    nref = min(rows, 8192)
    ref = (torch.relu(x[:nref].double() * sc.double() + sh.double()) @ w.double() + b.double())
    def run():
        if pool: return M.linear_dense_pool(x, w, pool, b, sc, sh, True, keep_z=True)
        return M.linear_dense(x, w, b, sc, sh, True)
    out = {}
    for mode in (0, 1):
        lib.votenet_debug_fast_bf3(mode)
        r = run()
        z, st = r[0], r[1]
        err = float((z[:nref].double() - ref).abs().max() / ref.abs().max())
        zf = z.double()
        st_ref = torch.cat([zf.sum(0), (zf * zf).sum(0)])
        serr = float(((st - st_ref).abs() / st_ref.abs().clamp_min(1.0)).max())
        extra = ""
        if pool:
            zmax, zmin, amax, amin = r[2]
            g = z.view(rows // pool, pool, cout)
            ok = bool((g.max(1).values == zmax).all() and (g.min(1).values == zmin).all() and (g.argmax(1).int() == amax).all()) if False else \
                 bool((g.max(1).values == zmax).all() and (g.min(1).values == zmin).all())
            extra = " pool_ok=%s" % ok
        t = timeit(lambda: (M.linear_dense_pool(x, w, pool, b, sc, sh, True, keep_z=False) if pool else run()))
        out[mode] = (t, err, serr, extra)
    fl = 2.0 * rows * cin * cout
    print("%-28s %9.4f %9.4f %8.1f %8.1f | %10.2e %10.2e  stats %.1e %.1e%s" % ((rows, cin, cout, pool), out[0][0], out[1][0], fl / out[0][0] / 1e9,
          fl / out[1][0] / 1e9, out[0][1], out[1][1], out[0][2], out[1][2], out[1][3]), flush=True)
lib.votenet_debug_fast_bf3(0)
