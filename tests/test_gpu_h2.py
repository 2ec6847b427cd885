"""GPU: the forward GEMMs on fp16 x 2 split operands (mlp_fast.hip, H2; mlp_types.h: split2) -- x = hi + lo with hi = rne16(x),
lo = rne16(x - hi), a product = three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi), fp32 accumulate -- against float64, the CPU
oracle, the bf16 x 3 form and the fp32 MFMA kernel, at the tolerances tests/test_gpu_bf3.py holds the bf16 x 3 form to."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def split2_numpy(w):
    """The kernel's split restated: round to fp16 (nearest even), subtract (exact in fp32), round again."""
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


@pytest.fixture()
def h2(hiplib):
    yield hiplib
    hiplib.votenet_debug_fast_bf3(1)
    hiplib.votenet_debug_fast_h2(1)


def test_the_two_piece_image_has_the_kernels_lds_order_and_22_bits(h2, dev):
    from votenet_amd import mlp
    rng = np.random.default_rng(0)
    cin, cout = 64, 128
    w = (rng.normal(size=(cin, cout)) * np.exp(rng.uniform(-9, 3, size=(cin, cout)))).astype(np.float32)
    w[0, 0], w[1, 1], w[2, 2], w[3, 3] = 0.0, -0.0, np.float32(2.0 ** -34), np.float32(200.0)
    w[4, 4], w[5, 5], w[6, 6], w[7, 7] = np.float32(3e-6 / 256), np.float32(-5e-5 / 256), np.float32(7e-8 / 256), np.float32(6.1e-5 / 256)  # hi itself a subnormal
    w[3, 3] = np.float32(200.0)               # the image holds 2^8 w (exact): |w| < 255
    ws = w * np.float32(256.0)
    hi, lo = split2_numpy(ws)
    rec = (hi.astype(np.float64) + lo.astype(np.float64)) / 256.0
    ok = np.abs(w) >= 2.0 ** -10  # hi AND lo normal numbers (of 2^8 w): 22 significant bits
    assert (np.abs(rec - w)[ok] <= np.abs(w)[ok] * 2.0 ** -22).all()
    assert (np.abs(rec - w) <= np.maximum(np.abs(w) * 2.0 ** -22, 2.0 ** -33)).all()  # below: the subnormal spacing of the lo piece / 2^8
    wt = T(w, dev)
    img = mlp.SplitImages([wt], pieces=2)
    img.refresh()
    torch.cuda.synchronize()
    got = N(img.buf).view(np.uint16)[:cin * cout * 2].reshape(cin // 16, 2, 2, cout, 8)  # [slab][piece][k-half][column][8 fp16]
    for p, piece in enumerate((hi, lo)):
        exp = piece.view(np.uint16).reshape(cin // 16, 2, 8, cout).transpose(0, 1, 3, 2)
        assert (got[:, p] == exp).all(), "piece %d of the image differs" % p
    img.close()


@pytest.mark.parametrize("rows,cin,cout,pool", [(4096, 64, 64, 0), (8192, 64, 128, 64), (4096, 128, 128, 0), (8192, 128, 256, 64),
                                                (2048, 256, 256, 0), (1024, 512, 256, 0), (640, 256, 320, 0), (256, 128, 128, 0)])
def test_h2_gemm_vs_oracle_float64_bf16x3_and_the_fp32_mfma_kernel(h2, dev, O, rows, cin, cout, pool):
    from votenet_amd import mlp
    rng = np.random.default_rng(rows + cin + cout)
    x = (rng.normal(size=(rows, cin)) * 2 + 0.3).astype(np.float32)
    w = (rng.normal(size=(cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    sc = (rng.random(cin) + 0.5).astype(np.float32)
    sh = (rng.normal(size=cin) * 0.2).astype(np.float32)
    xt, wt, bt, sct, sht = (T(a, dev) for a in (x, w, b, sc, sh))

    def run():
        if pool:
            z, st, pl = mlp.linear_dense_pool(xt, wt, pool, bt, sct, sht, True, keep_z=True)
            return N(z), N(st), [N(p) for p in pl]
        z, st = mlp.linear_dense(xt, wt, bt, sct, sht, True)
        return N(z), N(st), None
    img = mlp.SplitImages([wt], pieces=2)
    img.refresh()
    z2, st2, pl2 = run()
    h2.votenet_debug_fast_h2(0)       # the two-piece image ignored: the fp32 MFMA kernel
    z0, st0, _ = run()
    h2.votenet_debug_fast_h2(1)
    img.close()
    img3 = mlp.SplitImages([wt], pieces=3)
    img3.refresh()
    z3, _, _ = run()
    img3.close()
    a = np.maximum(x * sc + sh, 0.0).astype(np.float32)
    oz = O.linear(a, w, b)
    ref = a.astype(np.float64) @ w.astype(np.float64) + b
    bound = max(1.0, float((np.abs(a) @ np.abs(w)).max()))
    assert np.abs(z2 - oz).max() <= 1e-5 * bound
    e2, e3, e0 = (np.abs(z - ref).max() / bound for z in (z2, z3, z0))
    assert e2 <= 2e-6 and e2 <= 2.0 * e0 + 2e-7, "fp16 x 2: %.3g of the product magnitude, bf16 x 3: %.3g, fp32 MFMA: %.3g" % (e2, e3, e0)
    assert not np.array_equal(z2, z0) and not np.array_equal(z2, z3)  # three different kernels did run
    sref = np.concatenate([ref.sum(0), (ref * ref).sum(0)])
    assert np.allclose(st2, sref, rtol=1e-5, atol=1e-3 * bound)
    if pool:
        g = z2.reshape(rows // pool, pool, cout)
        assert (pl2[0] == g.max(1)).all() and (pl2[1] == g.min(1)).all()
        assert (pl2[2] == g.argmax(1)).all() and (pl2[3] == g.argmin(1)).all()


def test_small_and_large_magnitudes_inside_fp16s_range(h2, dev):
    """Operands far from unit scale, where the lo piece is a subnormal or hi is near fp16's maximum: the error stays at the fp32 kernel's
    level relative to the product magnitude."""
    from votenet_amd import mlp
    rng = np.random.default_rng(3)
    rows, cin, cout = 4096, 128, 128
    for xs, ws in ((1e-2, 1.0), (1.0, 1e-3), (300.0, 0.05), (3e-2, 1e-2), (500.0, 1.0), (1.0, 50.0)):
        x = (rng.normal(size=(rows, cin)) * xs).astype(np.float32)
        w = (rng.normal(size=(cin, cout)) * ws).astype(np.float32)
        xt, wt = T(x, dev), T(w, dev)
        img = mlp.SplitImages([wt], pieces=2)
        img.refresh()
        z2 = N(mlp.linear_dense(xt, wt)[0])
        img.close()
        z0 = N(mlp.linear_dense(xt, wt)[0])
        ref = x.astype(np.float64) @ w.astype(np.float64)
        bound = float((np.abs(x) @ np.abs(w)).max())
        e2, e0 = np.abs(z2 - ref).max() / bound, np.abs(z0 - ref).max() / bound
        assert e2 <= 2e-6 and e2 <= 3.0 * e0 + 3e-7, (xs, ws, e2, e0)


def test_a_backward_entry_point_ignores_a_two_piece_image(h2, dev):
    """Gradients do not fit fp16's range: the BatchNorm-backward input-gradient GEMM has no two-piece kernel; handed a matrix whose image
    is fp16 x 2 it must multiply by the matrix itself (fp32 MFMA kernel) -- never read the image as three bf16 pieces."""
    from votenet_amd import mlp
    rng = np.random.default_rng(4)
    rows, c = 4096, 128
    z = T(rng.normal(size=(rows, c)).astype(np.float32), dev)
    da = T((rng.normal(size=(rows, c)) * 1e-6).astype(np.float32), dev)   # gradient-sized
    coef = T(np.concatenate([np.ones(c), np.zeros(c), np.zeros(c), np.ones(c), np.zeros(c)]).astype(np.float32), dev)
    wT = T((rng.normal(size=(c, c)) * 0.1).astype(np.float32), dev)
    ref = mlp.dgrad_bn(z, coef, True, wT, da=da)
    img = mlp.SplitImages([wT], pieces=2)
    img.refresh()
    got = mlp.dgrad_bn(z, coef, True, wT, da=da)
    img.close()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("cin,cout", [(128, 256), (128, 128), (64, 128)])
@pytest.mark.parametrize("cscale", [1.0, 1e-6, 1e-12])
def test_the_gram_form_dgrad_matrix_as_a_scaled_two_piece_image(h2, dev, cin, cout, cscale):
    """votenet_pool_dgrad_prepare_h2: W diag(C) W^T is gradient-sized (C is a BatchNorm-backward coefficient): its fp16 x 2 image is
    scaled by powers of two taken from |W[j,:]|, |W[k,:]| and max|C| inside the launch that forms it; the GEMM scales its staged input
    channels and its output columns back.  Against float64 and the fp32 MFMA kernel, at coefficient magnitudes from 1 down to 1e-12,
    with weight rows of very different norms."""
    from votenet_amd import mlp
    g = torch.Generator().manual_seed(cin + cout)
    groups, k = 512, 64
    rows = groups * k
    w = torch.randn(cin, cout, generator=g) * 0.15
    w[::7] *= 30.0     # rows of very different norms
    w[3::11] *= 1e-3
    w[5] = 0.0         # a dead input channel
    w = w.to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    coef = (torch.randn(5 * cout, generator=g) * cscale).to(dev)
    xz = torch.randn(rows, cin, generator=g).to(dev)
    aff = torch.stack([torch.randn(cin, generator=g) * 0.3 + 1, torch.randn(cin, generator=g) * 0.2]).to(dev).contiguous()
    assert mlp.ADHOC_H2 and mlp.FORWARD_H2
    mm = mlp.pool_dgrad_prepare(w, b, coef, rows)
    assert getattr(mm, "_h2", None) is not None and mm._img.numel() == cin * cin * 4
    mlp.ADHOC_H2 = False
    try:
        mm0 = mlp.pool_dgrad_prepare(w, b, coef, rows)
    finally:
        mlp.ADHOC_H2 = True
    assert getattr(mm0, "_img", None) is None and torch.equal(mm0, mm)  # the fp32 matrix itself is the same
    # the scale vectors are powers of two, the scaled matrix fits fp16 with room to spare
    sv = N(mm._h2)
    assert (np.frexp(sv)[0] == 0.5).all()
    scaled = N(mm[:cin]).astype(np.float64) * (sv[0][:, None] / 16.0) ** -1 / (sv[1][None, :] * 16.0)
    assert np.abs(scaled).max() <= 2.0 ** 13
    from votenet_amd import _lib as L

    def dense(m):
        if getattr(m, "_h2", None) is not None:
            L.check(L.lib().votenet_register_split_weights_scaled(L.ptr(m), cin, cin, L.ptr(m._img), L.ptr(m._h2[0]), L.ptr(m._h2[1])))
        try:
            return mlp.linear_dense(xz, m[:cin], m[cin], aff[0], aff[1], True, want_stats=False)[0]
        finally:
            L.lib().votenet_register_split_weights(L.ptr(m), cin, cin, None)
    d2, d0 = dense(mm), dense(mm0)
    a = torch.relu(xz.double() * aff[0].double() + aff[1].double())
    exact = a @ mm[:cin].double() + mm[cin].double()
    bound = float((a.abs() @ mm[:cin].double().abs()).max())
    e2, e0 = float((d2.double() - exact).abs().max()) / bound, float((d0.double() - exact).abs().max()) / bound
    assert not torch.equal(d2, d0)
    assert e2 <= 2e-6 and e2 <= 3.0 * e0 + 3e-7, (e2, e0)
    # and the whole pooled-layer input gradient through mlp.pool_dgrad
    gout = (torch.randn(groups, cout, generator=g) * cscale).to(dev)
    arg = torch.randint(0, k, (groups, cout), generator=g, dtype=torch.int32).to(dev)
    zsel = torch.randn(groups, cout, generator=g).to(dev)
    wT = w.t().contiguous()
    da2 = mlp.pool_dgrad(xz, aff[0], aff[1], True, w, b, wT, coef, True, gout, arg, zsel, k, mm=mm)
    da0 = mlp.pool_dgrad(xz, aff[0], aff[1], True, w, b, wT, coef, True, gout, arg, zsel, k, mm=mm0)
    d = float((da2 - da0).abs().max() / da0.abs().max())
    assert 0.0 < d < 5e-6, d
