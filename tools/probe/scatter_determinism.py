"""probe: is the arg-max scatter of the pooled input gradient (votenet_pool_dgrad_scatter, full layout K = 64 / piece layout) bit-reproducible
run after run?  (the deterministic mode promises it; bench.py --gpus 2's self-check found it was not after the round-5 rewrite)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import mlp as M
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
for (G, cin, cout) in [(2048, 128, 256), (8192, 128, 256), (4096, 128, 128), (16384, 64, 128)]:
    k = 64
    rows = G * k
    xz = rnd(rows, cin)
    w, b = rnd(cin, cout) * 0.1, rnd(cout) * 0.1
    wT = w.t().contiguous()
    coef = rnd(5 * cout) * 0.3
    gout = rnd(G, cout)
    zsel = rnd(G, cout)
    argmax = torch.randint(0, 64, (G, cout), generator=g, dtype=torch.int32).to(dev)
    sc, sh = torch.rand(cin, generator=g).to(dev) + 0.5, rnd(cin) * 0.1
    below = (sc, sh, rnd(cin) * 0.1, torch.rand(cin, generator=g).to(dev) + 0.5, True)
    ref = None
    bad = 0
    for it in range(40):
        da, sums = M.pool_dgrad(xz, sc, sh, True, w, b, wT, coef, True, gout, argmax, zsel, k, below=below)
        torch.cuda.synchronize()
        if ref is None:
            ref = (da.clone(), sums.clone())
        else:
            if not torch.equal(da, ref[0]):
                bad += 1
                d = (da != ref[0])
                print("  iteration %d: %d elements of da differ, max |diff| %g" % (it, int(d.sum()), float((da - ref[0]).abs().max())))
    print("G %d cin %d cout %d: %d of 39 repeats differ; sums equal: %s" % (G, cin, cout, bad, torch.equal(sums, ref[1])))

# ---- the piece layout (the timed path): da must be bit-reproducible too (rows are stored, nothing is accumulated with atomics); here with a
# second stream of GEMMs beside it in the same process
print("piece layout, a second stream busy beside it:")
side = torch.cuda.Stream()
A_, B_ = torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)
for (G, cin, cout) in [(2048, 128, 256), (4096, 128, 256), (4096, 64, 128)]:
    cnt = torch.randint(1, 65, (1, G), generator=g, dtype=torch.int32).to(dev)
    half = M.half_groups(cnt)
    half.resolve()
    rows = half.rows
    xz = rnd(rows, cin)
    w, b = rnd(cin, cout) * 0.1, rnd(cout) * 0.1
    wT = w.t().contiguous()
    coef = rnd(5 * cout) * 0.3
    gout, zsel = rnd(G, cout), rnd(G, cout)
    # an arg-max slot inside the ball's real neighbours
    argmax = (torch.rand(G, cout, generator=g).to(dev) * cnt.view(G, 1).float()).int().clamp_(0, 63)
    sc, sh = torch.rand(cin, generator=g).to(dev) + 0.5, rnd(cin) * 0.1
    below = (sc, sh, rnd(cin) * 0.1, torch.rand(cin, generator=g).to(dev) + 0.5, True)
    ref, bad = None, 0
    for it in range(60):
        with torch.cuda.stream(side):
            for _ in range(3):
                C_ = A_ @ B_
        da, sums = M.pool_dgrad(xz, sc, sh, True, w, b, wT, coef, True, gout, argmax, zsel, 64, below=below, half=half)
        torch.cuda.synchronize()
        if ref is None:
            ref = da.clone()
        elif not torch.equal(da, ref):
            bad += 1
    print("G %d (%d pieces) cin %d cout %d: %d of 59 repeats differ" % (G, half.nh, cin, cout, bad))
