"""GPU: split-K of the fused GEMMs of few row tiles (csrc/mlp_fast.hip, FastArgs::sk_ws; mlp._SplitK) -- the shapes of the model's
static stretch (feature propagation, voting, proposal head: utils.py:286-293, model.py:53-57,89-93): forward with BatchNorm
statistics, plain forward, the BatchNorm-backward input gradient and the one that reduces the layer below.  Split results against the
unsplit kernel (same products, another association of the fp32 sums), against float64, and bit for bit against themselves run after
run (the parts are added in a fixed order whichever workgroup arrives last); both GEMM forms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture()
def sk(hiplib):
    from votenet_amd import mlp as M
    prev = M.SPLIT_K
    yield hiplib
    M.SPLIT_K = prev
    hiplib.votenet_debug_split_k(640, 4, 4, 400)


def would_split(lib, rows, cin, cout):
    return lib.votenet_mlp_split_k_floats(rows, cin, cout) > 0


SHAPES = [(4096, 512, 256), (8192, 512, 256), (8192, 256, 256), (8192, 256, 320), (8192, 320, 256), (2048, 128, 128), (4096, 256, 512),
          (8192, 256, 512), (2048, 128, 64), (1024, 512, 128)]


@pytest.mark.parametrize("rows,cin,cout", SHAPES)
def test_forward_split_vs_unsplit_vs_float64(sk, dev, gemm_form, rows, cin, cout):
    from votenet_amd import mlp as M
    rng = np.random.default_rng(rows + cin + cout)
    x = T((rng.normal(size=(rows, cin)) * 2 + 0.3).astype(np.float32), dev)
    w = T((rng.normal(size=(cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32), dev)
    b = T(rng.normal(size=cout).astype(np.float32), dev)
    sc = T((rng.random(cin) + 0.5).astype(np.float32), dev)
    sh = T((rng.normal(size=cin) * 0.2).astype(np.float32), dev)
    img = M.SplitImages([w])
    img.refresh()
    M.SPLIT_K = False
    z0, st0 = M.linear_dense(x, w, b, sc, sh, True)
    zp0, _ = M.linear_dense(x, w, b, sc, sh, True, want_stats=False)
    M.SPLIT_K = True
    z1, st1 = M.linear_dense(x, w, b, sc, sh, True)      # (registers the tickets on first use)
    assert would_split(sk, rows, cin, cout), "the plan leaves this shape alone: nothing tested"
    z1, st1 = M.linear_dense(x, w, b, sc, sh, True)
    zp1, _ = M.linear_dense(x, w, b, sc, sh, True, want_stats=False)
    a = torch.relu(x.double() * sc.double() + sh.double())
    ref = a @ w.double() + b.double()
    bound = float((a.abs() @ w.double().abs()).max())
    e0, e1 = float((z0.double() - ref).abs().max()) / bound, float((z1.double() - ref).abs().max()) / bound
    assert e1 <= 2e-6 and e1 <= 2.0 * e0 + 2e-7, (e0, e1)
    assert float((z1 - z0).abs().max()) / bound < 1e-6 and float((zp1 - zp0).abs().max()) / bound < 1e-6
    assert torch.equal(z1, zp1)  # the statistics epilogue does not touch what is stored
    # the statistics are those of the complete tiles
    zf = z1.double()
    st_ref = torch.cat([zf.sum(0), (zf * zf).sum(0)])
    assert float(((st1 - st_ref).abs() / st_ref.abs().clamp_min(1.0)).max()) < 1e-5
    # bit-reproducible: fixed order of the parts
    for _ in range(3):
        z2, st2 = M.linear_dense(x, w, b, sc, sh, True)
        assert torch.equal(z2, z1)
    img.close()


@pytest.mark.parametrize("parts,min_slabs", [(2, 2), (4, 2), (3, 2)])
def test_every_part_count_the_hook_allows(sk, dev, gemm_form, parts, min_slabs):
    from votenet_amd import mlp as M
    rows, cin, cout = 2048, 384, 128  # 24 slabs: 2, 3 and 4 parts divide them into an even number
    rng = np.random.default_rng(parts)
    x = T(rng.normal(size=(rows, cin)).astype(np.float32), dev)
    w = T((rng.normal(size=(cin, cout)) * 0.1).astype(np.float32), dev)
    img = M.SplitImages([w])
    img.refresh()
    M.SPLIT_K = False
    z0, _ = M.linear_dense(x, w)
    M.SPLIT_K = True
    M.linear_dense(x, w)
    sk.votenet_debug_split_k(100000, parts, min_slabs, 400)
    n = sk.votenet_mlp_split_k_floats(rows, cin, cout)
    assert n == (rows // 128) * (cout // 64) * parts * 128 * 64, n
    z1, st1 = M.linear_dense(x, w)
    bound = float((x.double().abs() @ w.double().abs()).max())
    assert float((z1 - z0).abs().max()) / bound < 1e-6
    assert float((z1.double() - x.double() @ w.double()).abs().max()) / bound < 2e-6
    img.close()


@pytest.mark.parametrize("rows,c,cprev", [(8192, 256, 512), (8192, 256, 256), (4096, 256, 512), (8192, 320, 256), (2048, 128, 128)])
def test_input_gradients_split_vs_unsplit_vs_float64(sk, dev, gemm_form, rows, c, cprev):
    """votenet_mlp_dgrad_bn (SRC 1: dz = A g + B + C z with the ReLU mask rebuilt in the loader) and votenet_mlp_dgrad_bn_reduce (its
    epilogue reads z of the layer below and reduces that layer's BatchNorm backward; coefficient tail by the last workgroup)."""
    from votenet_amd import mlp as M
    rng = np.random.default_rng(rows + c + cprev)
    z = T(rng.normal(size=(rows, c)).astype(np.float32), dev)
    da = T(rng.normal(size=(rows, c)).astype(np.float32), dev)
    coef = T(np.concatenate([rng.random(c) + 0.5, rng.normal(size=c) * 0.1, rng.normal(size=c) * 0.1, rng.random(c) + 0.5,
                             rng.normal(size=c) * 0.2]).astype(np.float32), dev)
    wT = T((rng.normal(size=(c, cprev)) * 0.1).astype(np.float32), dev)
    zb = T(rng.normal(size=(rows, cprev)).astype(np.float32), dev)
    bsc, bsh, bmu = (T(rng.normal(size=cprev).astype(np.float32), dev) for _ in range(3))
    bvar = T((rng.random(cprev) + 0.5).astype(np.float32), dev)
    img = M.SplitImages([wT])
    img.refresh()
    M.SPLIT_K = False
    d0 = M.dgrad_bn(z, coef, True, wT, da=da)
    r0, s0 = M.dgrad_bn(z, coef, True, wT, da=da, below=(zb, bsc, bsh, bmu, bvar, True))
    M.SPLIT_K = True
    M.dgrad_bn(z, coef, True, wT, da=da)
    assert would_split(sk, rows, c, cprev)
    d1 = M.dgrad_bn(z, coef, True, wT, da=da)
    r1, s1 = M.dgrad_bn(z, coef, True, wT, da=da, below=(zb, bsc, bsh, bmu, bvar, True))
    A, Bc, C, S, H = (coef[i * c:(i + 1) * c].double() for i in range(5))
    g = torch.where(z.double() * S + H > 0, da.double(), torch.zeros_like(da, dtype=torch.float64))
    dz = A * g + Bc + C * z.double()
    ref = dz @ wT.double()
    bound = float((dz.abs() @ wT.double().abs()).max())
    e0, e1 = float((d0.double() - ref).abs().max()) / bound, float((d1.double() - ref).abs().max()) / bound
    assert e1 <= 2e-6 and e1 <= 2.0 * e0 + 2e-7, (e0, e1)
    assert torch.equal(r1, d1) and torch.equal(r0, d0)
    gm = d1.double() * (zb * bsc + bsh > 0).double()
    zhat = (zb.double() - bmu.double()) / torch.sqrt(bvar.double() + M.BN_EPS)
    s_exact = torch.cat([gm.sum(0), (gm * zhat).sum(0)])
    scale = torch.cat([gm.abs().sum(0), (gm * zhat).abs().sum(0)])
    assert float(((s1 - s_exact).abs() / (scale + 1e-30)).max()) < 1e-5
    assert float(((s0 - s_exact).abs() / (scale + 1e-30)).max()) < 1e-4  # (d0's own mask may differ from d1's on entries at the ReLU edge)
    for _ in range(3):
        assert torch.equal(M.dgrad_bn(z, coef, True, wT, da=da), d1)
    img.close()


def test_unarmed_and_large_launches_never_split(sk, dev):
    from votenet_amd import mlp as M
    assert sk.votenet_mlp_split_k_floats(524288, 128, 128) == 0      # many row tiles: left alone
    assert sk.votenet_mlp_split_k_floats(1000, 256, 256) == 0        # rows % 128 != 0: the generic kernel
    x = torch.randn(2048, 256, device=dev)
    w = torch.randn(256, 128, device=dev) * 0.1
    M.SPLIT_K = True
    a, _ = M.linear_dense(x, w)
    M.SPLIT_K = False
    b, _ = M.linear_dense(x, w)
    # straight through the C ABI without arming: the unsplit kernel, bit for bit
    import ctypes
    from votenet_amd import _lib as L
    z = torch.empty(2048, 128, device=dev)
    d = M._desc_dense(x, None, None, True, None)
    L.check(sk.votenet_mlp_linear(ctypes.byref(d), 2048, 256, 128, L.ptr(w), None, L.ptr(z), None, L.stream_ptr()))
    assert torch.equal(z, b)
    assert float((a - b).abs().max()) < 1e-4


def test_stretch_with_and_without_split_k_agree(sk, dev):
    """The whole forward pass (fp1 / fp2 / voting / proposal head run split) against the unsplit pass."""
    from votenet_amd import mlp as M
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = torch.from_numpy(synth.room_batch(2, 4096, 21)).to(dev)
    net = VM.VoteNetHotPath(dev, seed=3, npoints=(512, 256, 128, 64))
    M.SPLIT_K = False
    a = net.forward(x)
    a = {k: v.clone() for k, v in a.items()}
    M.SPLIT_K = True
    b = net.forward(x)
    for k in ("seeds_points", "votes_xyz"):
        scale = max(1.0, float(a[k].abs().max()))
        assert float((a[k] - b[k]).abs().max()) <= 1e-4 * scale, k
