"""Is the fp16 x 2 path (split2: v_cvt_pk_f16_f32 + v_fma_mix_f32 with op_sel; v_mfma_f32_32x32x16_f16; the packed fma of the statistics) exact
beside ANOTHER kernel's MFMA wavefronts?  Round 5 found packed-f32 instructions with op_sel[1] = 1 returning wrong low halves in that situation
(profiles/r05_pk_opsel_hazard.txt); the instructions new in round 6 get the same treatment: the weight image and a forward GEMM (statistics and
pooled epilogue included) are repeated N times while a second stream runs MFMA GEMMs on the same GPU, every repetition compared BIT FOR BIT with
the first one computed alone.  Run three of these at once (tools/probe/h2_repeat3.sh).   python tools/probe/h2_repeat.py [N]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import numpy as np, torch
from votenet_amd import mlp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "1")))
rows, cin, cout = 65536, 128, 256
x = (torch.randn(rows, cin, generator=g) * 2 + 0.3).to(dev)
w = (torch.randn(cin, cout, generator=g) * 0.12).to(dev)
b = torch.randn(cout, generator=g).to(dev)
sc = (torch.rand(cin, generator=g) + 0.5).to(dev)
sh = (torch.randn(cin, generator=g) * 0.2).to(dev)
img = mlp.SplitImages([w], pieces=2)
img.refresh()
torch.cuda.synchronize()
img0 = img.buf.clone()
hi = (w.cpu().numpy() * np.float32(256.0)).astype(np.float16)
z0, st0, pl0 = mlp.linear_dense_pool(x, w, 64, b, sc, sh, True, keep_z=True)
z0, st0, pl0 = z0.clone(), st0.clone(), [p.clone() for p in pl0]
zz0, sst0 = mlp.linear_dense(x, w, b, sc, sh, True)
zz0, sst0 = zz0.clone(), sst0.clone()
torch.cuda.synchronize()
# the neighbour: MFMA GEMMs on a second stream (bf16 matmul through the library torch ships: any MFMA wavefronts will do)
side = torch.cuda.Stream()
A = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
bad = {"image": 0, "z": 0, "stats": 0, "pool": 0, "z_plain": 0, "stats_plain": 0}
for i in range(N):
    with torch.cuda.stream(side):
        for _ in range(3):
            A2 = A @ A
    img.refresh()
    z, st, pl = mlp.linear_dense_pool(x, w, 64, b, sc, sh, True, keep_z=True)
    zz, sst = mlp.linear_dense(x, w, b, sc, sh, True)
    torch.cuda.synchronize()
    bad["image"] += int(not torch.equal(img.buf, img0))
    bad["z"] += int(not torch.equal(z, z0))
    bad["pool"] += int(any(not torch.equal(p, q) for p, q in zip(pl, pl0)))
    bad["z_plain"] += int(not torch.equal(zz, zz0))
    # (the statistics are fp64 atomics of per-workgroup fp32 partials: order-dependent in the last bits by design -- compared to 1e-12)
    bad["stats"] += int(not torch.allclose(st, st0, rtol=1e-11, atol=0))
    bad["stats_plain"] += int(not torch.allclose(sst, sst0, rtol=1e-11, atol=0))
print("h2_repeat: %d repetitions beside an MFMA stream: mismatches %s" % (N, bad))
sys.exit(1 if any(bad.values()) else 0)
