"""The four backward GEMM families that hold the train step down, launched in isolation for counter collection (GPU box only):
assembled input gradient with the BatchNorm-backward reduce of the layer below in its epilogue, assembled weight gradient,
the same two of sa1's narrow layer, and the Gram matrix of a pooled layer -- at sa2's shape (524 288 x 128 x 128, real sa2 geometry of
room scenes) and sa1's (1 048 576 x 64 x 64).

    python tools/pmc_mlp_bwd.py alone     every kernel REP times back to back on one stream
    python tools/pmc_mlp_bwd.py beside    the input-gradient chain on the main stream with the weight-gradient kernels on a second
                                          stream (what the train step does): read the durations from a --kernel-trace run

tools/pmc_mlp_bwd.sh wraps the rocprofv3 passes and writes profiles/r03_pmc_mlp_bwd.txt."""
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import _lib as L_
if os.environ.get("VARIANT"):  # tools/probe/build_variant.sh NAME ...: the library with one source rebuilt under other flags
    L_._LIB_PATH = os.path.join(R, "tools", "probe", "lib", "libvotenet_%s.so" % os.environ["VARIANT"])
from votenet_amd import mlp as M, model as VM, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
REP = int(os.environ.get("REP", "4"))
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
x = torch.from_numpy(synth.room_batch(8, 20480, 7)).to(dev)
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
pos = lambda *s: (torch.rand(*s, generator=g) + 0.5).to(dev)

# ---- sa1 (narrow) and sa2 (assembled) geometry of real room scenes
geom1 = net.sa1.geometry(x, points=None)
new1, idx1 = geom1[1], geom1[2]
u8, mom = geom1[4:6] if len(geom1) >= 6 else M.narrow_rows(x, new1, None, idx1)
geom2 = net.sa2.geometry(new1)
new2, idx2, cnt2 = geom2[1], geom2[2], geom2[3]
geo, cntv, mom2 = geom2[4:7] if len(geom2) >= 7 else M.assemble_rows(new1, new2, idx2, pts_cnt=cnt2)

jobs = {}
# sa2: 524288 x 128 -> 128
rows, c = geo.shape[0], 128
b, n = new1.shape[:2]
P = rnd(b * n, c) * 0.5
wx = rnd(3, c) * 0.5
w1 = rnd(c, c) * 0.1
wT = w1.t().contiguous()
img = M.SplitImages([w1, wT])
img.refresh()
z1, da1, coef1 = rnd(rows, c), rnd(rows, c), rnd(5 * c)
bn0 = (pos(c), rnd(c), rnd(c), pos(c), True)
dw = torch.zeros(c, c, device=dev)
aff = torch.stack([bn0[0], bn0[1]]).contiguous()
jobs["sa2 dgrad_bn_reduce assembled"] = lambda: M.assembled_dgrad_bn_reduce(z1, coef1, True, wT, da1, geo, P, wx, bn0)
jobs["sa2 wgrad_bn assembled"] = lambda: M.assembled_wgrad_bn(geo, P, wx, bn0[0], bn0[1], True, z1, coef1, True, da1, dw)
jobs["sa2 gram"] = lambda: M.gram(z1, aff, True)
jobs["sa2 fwd-type dense (Gram-form dgrad)"] = lambda: M.linear_dense(z1, w1, None, bn0[0], bn0[1], True, want_stats=False)
jobs["sa2 dgrad_bn plain (EPI 1)"] = lambda: M.dgrad_bn(z1, coef1, True, wT, da=da1)
# the forward families at the same shapes, for the same counters side by side
w2 = rnd(c, 2 * c) * 0.1
img3 = M.SplitImages([w2])
img3.refresh()
bnp = M.FrozenBN(torch.stack([bn0[0], bn0[1]]).contiguous())
jobs["sa2 fwd+pool (128 -> 256)"] = lambda: M.linear_dense_pool(z1, w2, 64, None, bn0[0], bn0[1], True, keep_z=False)
jobs["sa2 fwd+bn assembled"] = lambda: M.assembled_linear(geo, P, wx, w1, None, bnp)
# sa1: 1048576 x 64 -> 64
rows1, c1 = u8.shape[0], 64
w0, b0 = rnd(6, c1) * 0.5, rnd(c1) * 0.1
w1n = rnd(c1, c1) * 0.2
wTn = w1n.t().contiguous()
img2 = M.SplitImages([w1n, wTn])
img2.refresh()
z1n, da1n, coef1n = rnd(rows1, c1), rnd(rows1, c1), rnd(5 * c1)
bn0n = (pos(c1), rnd(c1), rnd(c1), pos(c1), True)
dwn = torch.zeros(c1, c1, device=dev)
affn = torch.stack([bn0n[0], bn0n[1]]).contiguous()
jobs["sa1 dgrad_bn_reduce narrow"] = lambda: M.narrow_dgrad_bn_reduce(z1n, coef1n, True, wTn, da1n, u8, w0, b0, bn0n)
jobs["sa1 wgrad_bn narrow"] = lambda: M.narrow_wgrad_bn(u8, w0, b0, bn0n[0], bn0n[1], True, z1n, coef1n, True, da1n, dwn)
jobs["sa1 gram"] = lambda: M.gram(z1n, affn, True)

torch.cuda.synchronize()
if mode == "alone":
    for name, fn in jobs.items():
        for _ in range(REP):
            fn()
        torch.cuda.synchronize()
elif mode == "beside":
    side = torch.cuda.Stream()
    pairs = [("sa2 dgrad_bn_reduce assembled", "sa2 wgrad_bn assembled"), ("sa2 fwd-type dense (Gram-form dgrad)", "sa2 gram"),
             ("sa1 dgrad_bn_reduce narrow", "sa1 wgrad_bn narrow")]
    for main_k, side_k in pairs:
        for _ in range(REP):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                jobs[side_k]()
            jobs[main_k]()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
elif mode == "time":
    from bench_legs import gpu_ms
    for name, fn in jobs.items():
        t = gpu_ms(fn, it=10)
        rr, cc = (rows1, c1) if name.startswith("sa1") else (rows, c)
        co = 2 * cc if "256" in name else cc
        print("%-40s %.4f ms  %6.1f TF/s" % (name, t, 2.0 * rr * cc * co / t / 1e9))
    side = torch.cuda.Stream()
    for main_k, side_k in [("sa2 dgrad_bn_reduce assembled", "sa2 wgrad_bn assembled"), ("sa2 fwd-type dense (Gram-form dgrad)", "sa2 gram"),
                           ("sa1 dgrad_bn_reduce narrow", "sa1 wgrad_bn narrow")]:
        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                jobs[side_k]()
            jobs[main_k]()
            torch.cuda.current_stream().wait_stream(side)
        print("%-40s + %-28s together %.4f ms" % (main_k, side_k, gpu_ms(both, it=10)))
torch.cuda.synchronize()
