"""GPU: the fused GEMMs on bf16 x 3 split operands (mlp_fast.hip, BF3) -- fp32 operands split EXACTLY into three bf16 pieces (each rounded to nearest even),
six v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulate -- against the CPU oracle (oracle/oracle_mlp.c), float64, and the
fp32 MFMA kernels they replace.  Tolerance: the one tests/test_gpu_mlp.py holds the fp32 kernels to (1e-5 of the magnitude of
the accumulated products); the split's own error (the three dropped cross terms) is < 2^-23 of a product."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def bf16_rne(x):
    """float32 -> the float32 value of its bfloat16 rounding (nearest even), as v_cvt_pk_bf16_f32 does for finite inputs."""
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3_numpy(w):
    """The kernel's split restated: round to bf16, subtract (exact), twice."""
    hi = bf16_rne(w)
    r1 = w - hi
    mid = bf16_rne(r1)
    r2 = r1 - mid
    lo = bf16_rne(r2)
    return hi, mid, lo


@pytest.fixture()
def bf3(hiplib):
    """Restore the default (images are used) whatever the test switched."""
    yield hiplib
    hiplib.votenet_debug_fast_bf3(1)


def test_the_split_is_exact_and_the_image_has_the_kernels_lds_order(bf3, dev):
    from votenet_amd import mlp
    rng = np.random.default_rng(0)
    cin, cout = 64, 128
    w = (rng.normal(size=(cin, cout)) * np.exp(rng.uniform(-20, 20, size=(cin, cout)))).astype(np.float32)  # every exponent range
    w[0, 0], w[1, 1], w[2, 2] = 0.0, -0.0, np.float32(2.0 ** -120)
    hi, mid, lo = split3_numpy(w)
    assert ((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)) == w.astype(np.float64)).all()
    assert ((lo.view(np.uint32) & 0xFFFF) == 0).all()  # a bfloat16 value
    assert (np.abs(mid) <= np.abs(w) * 2.0 ** -8).all() and (np.abs(lo) <= np.abs(w) * 2.0 ** -16).all()
    wt = T(w, dev)
    img = mlp.SplitImages([wt])
    img.refresh()
    torch.cuda.synchronize()
    got = N(img.buf).view(np.uint16)[:cin * cout * 3].reshape(cin // 16, 3, 2, cout, 8)  # [slab][piece][k-half][column][8 bf16]
    for p, piece in enumerate((hi, mid, lo)):
        exp = (piece.view(np.uint32) >> 16).astype(np.uint16).reshape(cin // 16, 2, 8, cout).transpose(0, 1, 3, 2)
        assert (got[:, p] == exp).all(), "piece %d of the image differs" % p
    img.close()


@pytest.mark.parametrize("rows,cin,cout,pool", [(4096, 64, 64, 0), (8192, 64, 128, 64), (4096, 128, 128, 0), (8192, 128, 256, 64),
                                                (2048, 256, 256, 0), (1024, 512, 256, 0), (640, 256, 320, 0), (256, 128, 128, 0)])
def test_bf3_gemm_vs_oracle_float64_and_the_fp32_mfma_kernel(bf3, dev, O, rows, cin, cout, pool):
    from votenet_amd import mlp
    rng = np.random.default_rng(rows + cin + cout)
    x = (rng.normal(size=(rows, cin)) * 2 + 0.3).astype(np.float32)
    w = (rng.normal(size=(cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    sc = (rng.random(cin) + 0.5).astype(np.float32)
    sh = (rng.normal(size=cin) * 0.2).astype(np.float32)
    xt, wt, bt, sct, sht = (T(a, dev) for a in (x, w, b, sc, sh))
    img = mlp.SplitImages([wt])
    img.refresh()

    def run():
        if pool:
            z, st, pl = mlp.linear_dense_pool(xt, wt, pool, bt, sct, sht, True, keep_z=True)
            return N(z), N(st), [N(p) for p in pl]
        z, st = mlp.linear_dense(xt, wt, bt, sct, sht, True)
        return N(z), N(st), None
    bf3.votenet_debug_fast_bf3(1)
    z3, st3, pl3 = run()
    bf3.votenet_debug_fast_bf3(0)
    z0, st0, pl0 = run()
    a = np.maximum(x * sc + sh, 0.0).astype(np.float32)   # the folded BatchNorm + ReLU of the loader, in fp32 as the kernel does
    oz = O.linear(a, w, b)
    ref = a.astype(np.float64) @ w.astype(np.float64) + b
    bound = max(1.0, float((np.abs(a) @ np.abs(w)).max()))
    assert np.abs(z3 - oz).max() <= 1e-5 * bound          # the oracle, at the fp32 kernels' tolerance
    e3, e0 = np.abs(z3 - ref).max() / bound, np.abs(z0 - ref).max() / bound
    assert e3 <= 2e-6 and e3 <= 2.0 * e0 + 2e-7, "bf16 x 3: %.3g of the product magnitude, fp32 MFMA: %.3g" % (e3, e0)
    assert not np.array_equal(z3, z0)                     # the two modes did run different kernels
    sref = np.concatenate([ref.sum(0), (ref * ref).sum(0)])
    assert np.allclose(st3, sref, rtol=1e-5, atol=1e-3 * bound)
    if pool:
        g = z3.reshape(rows // pool, pool, cout)
        assert (pl3[0] == g.max(1)).all() and (pl3[1] == g.min(1)).all()
        assert (pl3[2] == g.argmax(1)).all() and (pl3[3] == g.argmin(1)).all()
    img.close()


def test_a_matrix_without_an_image_runs_the_fp32_kernel_whatever_the_switch(bf3, dev):
    from votenet_amd import mlp
    rng = np.random.default_rng(5)
    x, w = T(rng.normal(size=(4096, 128)).astype(np.float32), dev), T((rng.normal(size=(128, 128)) * 0.1).astype(np.float32), dev)
    bf3.votenet_debug_fast_bf3(1)
    za, _ = mlp.linear_dense(x, w)
    bf3.votenet_debug_fast_bf3(0)
    zb, _ = mlp.linear_dense(x, w)
    assert torch.equal(za, zb)
    img = mlp.SplitImages([w])
    img.refresh()
    bf3.votenet_debug_fast_bf3(1)
    zc, _ = mlp.linear_dense(x, w)
    assert not torch.equal(zc, za) and float((zc - za).abs().max()) < 1e-4
    img.close()  # the registration is withdrawn: the fp32 kernel again
    zd, _ = mlp.linear_dense(x, w)
    assert torch.equal(zd, za)


def test_dgrad_with_the_batchnorm_backward_folded_in_on_split_operands(bf3, dev):
    """votenet_mlp_dgrad_bn (loader SRC 1: dz = A g + B + C z with the ReLU mask) on an image of W^T against float64."""
    from votenet_amd import mlp
    rng = np.random.default_rng(9)
    rows, c, cprev = 4096, 128, 128
    z = T(rng.normal(size=(rows, c)).astype(np.float32), dev)
    da = T(rng.normal(size=(rows, c)).astype(np.float32), dev)
    coef = T(np.concatenate([rng.random(c) + 0.5, rng.normal(size=c) * 0.1, rng.normal(size=c) * 0.1, rng.random(c) + 0.5,
                             rng.normal(size=c) * 0.2]).astype(np.float32), dev)
    wT = T((rng.normal(size=(c, cprev)) * 0.1).astype(np.float32), dev)
    img = mlp.SplitImages([wT])
    img.refresh()
    outs = []
    for mode in (1, 0):
        bf3.votenet_debug_fast_bf3(mode)
        outs.append(mlp.dgrad_bn(z, coef, True, wT, da=da))
    A, Bc, C, S, H = (coef[i * c:(i + 1) * c].double() for i in range(5))
    g = torch.where(z.double() * S + H > 0, da.double(), torch.zeros_like(da, dtype=torch.float64))
    ref = (A * g + Bc + C * z.double()) @ wT.double()
    bound = float(((A * g + Bc + C * z.double()).abs() @ wT.double().abs()).max())
    e3, e0 = float((outs[0].double() - ref).abs().max()) / bound, float((outs[1].double() - ref).abs().max()) / bound
    assert e3 <= 2e-6 and e3 <= 2.0 * e0 + 2e-7
    img.close()


def test_forward_rebuilds_the_images_after_the_weights_changed_by_hand(bf3, dev):
    from votenet_amd import model as VM, synth
    x = torch.from_numpy(synth.room_batch(2, 4096, 3)).to(dev)
    net = VM.VoteNetHotPath(dev, seed=1, npoints=(512, 256, 128, 64))
    bf3.votenet_debug_fast_bf3(1)
    key = "proposals_output"
    a = net.forward(x)[key]
    name = [n for n, v in net.store.views.items() if n.startswith("sa2") and v.dim() == 2 and v.shape == (128, 128)][0]
    net.store[name].add_(torch.randn_like(net.store[name]) * 0.1)  # by hand, behind the store's back (a scaling would vanish in the BatchNorm)
    b3 = net.forward(x)[key]
    bf3.votenet_debug_fast_bf3(0)
    b0 = net.forward(x)[key]
    assert not torch.allclose(a, b0, rtol=1e-3, atol=1e-3)          # the change is visible ...
    scale = float(b0.abs().max())
    assert float((b3 - b0).abs().max()) <= 2e-4 * max(1.0, scale)  # ... and the BF3 pass saw the same weights (a stale image would not)


# The backward pass on split operands is held to float64 autograd by tests/test_gpu_backward.py::test_full_backward_vs_autograd, which
# runs once with the images in use and once on the fp32 MFMA kernels.  (Two whole train steps that differ in rounding are NOT
# comparable element by element: the proposal module samples and groups the *predicted* votes, so a last-bit difference upstream
# moves a neighbour set and with it a few per cent of the gradient -- measured 2.5e-2 between the two GEMM forms, the same
# size as between two runs of one form with atomics.)


@pytest.mark.parametrize("rows,c,relu", [(4096, 128, True), (40000, 128, True), (4111, 64, True), (131072, 64, False), (130, 128, True)])
def test_gram_matrix_on_split_operands_vs_float64_and_the_fp32_kernel(bf3, dev, rows, c, relu):
    """votenet_mlp_gram (pool_bwd.hip): a^T a of the activated input of a pooled layer, the contraction over the rows -- the BF3 kernel
    stages 8 consecutive rows of a channel per fragment.  Ragged row counts (padding rows must add nothing), both widths."""
    from votenet_amd import mlp
    rng = np.random.default_rng(rows + c)
    z = T((rng.normal(size=(rows, c)) * 1.5 + 0.2).astype(np.float32), dev)
    ss = T(np.stack([rng.random(c) + 0.5, rng.normal(size=c) * 0.3]).astype(np.float32), dev)
    a = z.double() * ss[0].double() + ss[1].double()
    if relu:
        a = torch.relu(a)
    ref = a.t() @ a
    bound = float((a.abs().t() @ a.abs()).max())
    errs, gs = [], []
    for mode in (1, 3, 0):  # fp16 x 2 pieces (the default), bf16 x 3 pieces, the fp32 MFMA kernel
        bf3.votenet_debug_gram_bf3(mode)
        g = mlp.gram(z, ss, relu)[:c]
        gs.append(g)
        errs.append(float((g.double() - ref).abs().max()) / bound)
    bf3.votenet_debug_gram_bf3(1)
    assert not torch.equal(gs[0], gs[1]) and not torch.equal(gs[1], gs[2])  # three kernels did run
    for e in errs[:2]:
        assert e <= 3e-6 and e <= 2.0 * errs[2] + 5e-7, "fp16 x 2: %.3g of the accumulated magnitude, bf16 x 3: %.3g, fp32 MFMA: %.3g" % tuple(errs)


@pytest.mark.parametrize("cin,cout", [(128, 256), (128, 128), (64, 128)])
def test_the_gram_form_dgrad_matrix_gets_its_image_from_the_launch_that_forms_it(bf3, dev, cin, cout):
    """votenet_pool_dgrad_prepare_split: the image written beside W diag(C) W^T is bit for bit the image votenet_split_weights makes
    of that matrix, and pool_dgrad on it equals pool_dgrad on the fp32 kernel to the split's accuracy (and differs: it really ran)."""
    from votenet_amd import mlp
    g = torch.Generator().manual_seed(cin + cout)
    groups, k = 1024, 64
    rows = groups * k
    w = (torch.randn(cin, cout, generator=g) * 0.15).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    coef = torch.randn(5 * cout, generator=g).to(dev)
    assert rows >= mlp.SPLIT_ADHOC_ROWS
    prev, mlp.SPLIT_ADHOC = mlp.SPLIT_ADHOC, True  # off by default (measured slower inside the step): the path stays tested
    prev_h2, mlp.ADHOC_H2 = mlp.ADHOC_H2, False    # (the default since round 6 is the fp16 x 2 image: tests/test_gpu_h2.py)
    try:
        mm = mlp.pool_dgrad_prepare(w, b, coef, rows)
    finally:
        mlp.SPLIT_ADHOC = prev
    assert getattr(mm, "_img", None) is not None
    ref = mlp.SplitImages([mm[:cin]])
    ref.refresh()
    torch.cuda.synchronize()
    assert torch.equal(mm._img.view(torch.int32), ref.buf.view(torch.int32)[:mm._img.numel() // 4])
    ref.close()
    mm0 = mlp.pool_dgrad_prepare(w, b, coef, 1)  # below SPLIT_ADHOC_ROWS: no image
    assert getattr(mm0, "_img", None) is None and torch.equal(mm0, mm)
    xz = torch.randn(rows, cin, generator=g).to(dev)
    aff = torch.stack([torch.randn(cin, generator=g) * 0.3 + 1, torch.randn(cin, generator=g) * 0.2]).to(dev).contiguous()
    gout = torch.randn(groups, cout, generator=g).to(dev)
    arg = torch.randint(0, k, (groups, cout), generator=g, dtype=torch.int32).to(dev)
    zsel = torch.randn(groups, cout, generator=g).to(dev)
    wT = w.t().contiguous()
    da3 = mlp.pool_dgrad(xz, aff[0], aff[1], True, w, b, wT, coef, True, gout, arg, zsel, k, mm=mm)
    da1 = mlp.pool_dgrad(xz, aff[0], aff[1], True, w, b, wT, coef, True, gout, arg, zsel, k, mm=mm0)
    mlp.ADHOC_H2 = prev_h2
    d = float((da3 - da1).abs().max() / da1.abs().max())
    assert 0.0 < d < 2e-6, d
    a = torch.relu(xz.double() * aff[0].double() + aff[1].double())
    exact = a @ mm[:cin].double() + mm[cin].double()
    dense3, _ = _dense_only(mlp, xz, mm, cin, aff)
    e3 = float((dense3.double() - exact).abs().max() / exact.abs().max())
    assert e3 < 1e-5, e3


def _dense_only(mlp, xz, mm, cin, aff):
    from votenet_amd import _lib as L
    L.check(L.lib().votenet_register_split_weights(L.ptr(mm), cin, cin, L.ptr(mm._img)))
    try:
        return mlp.linear_dense(xz, mm[:cin], mm[cin], aff[0], aff[1], True, want_stats=False)
    finally:
        L.lib().votenet_register_split_weights(L.ptr(mm), cin, cin, None)


def test_a_module_driven_directly_after_an_optimizer_step_multiplies_by_the_current_weights(bf3, dev):
    """The optimizer updates the flat bucket through a raw pointer.  forward() refreshes the bf16 x 3 images at its start; a module
    called on its own afterwards (SAModule.forward, backbone) must not multiply by the images of the previous generation: the
    parameter accessor rebuilds them on first use (ParamStore.generation)."""
    from votenet_amd import loss as VL, mlp as M, model as VM, synth
    net = VM.VoteNetHotPath(dev, seed=1, npoints=(512, 256, 128, 64))
    x = torch.from_numpy(synth.room_batch(2, 4096, 9)).to(dev)
    gt = VL.gt_to_device(synth.room_gt(2, 4096, 9), dev)
    net.init_optimizer(lr=0.05)  # a large step: stale images would be far off
    net.train_step(x, gt=gt)
    gen = net.store.generation
    assert net.store._split_gen != gen  # the images are those of the previous generation now
    M.arena_begin(dev)
    try:
        _, direct, _ = net.sa1.forward(x, x)  # no refresh_split() by the caller
    finally:
        M.arena_end()
    assert net.store._split_gen == gen
    bf3.votenet_debug_fast_bf3(0)  # the same call on the fp32 kernels reads the weights themselves
    M.arena_begin(dev)
    try:
        _, ref, _ = net.sa1.forward(x, x)
    finally:
        M.arena_end()
    bf3.votenet_debug_fast_bf3(1)
    assert float((direct - ref).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("rows,cin,cout", [(4096, 128, 128), (40000, 128, 128), (4111, 64, 64), (65536, 64, 128), (8192, 256, 128), (130, 128, 64),
                                           (100000, 128, 256)])
def test_weight_gradient_on_split_operands_vs_float64_and_the_fp32_kernel(bf3, dev, rows, cin, cout):
    """mlp_wgrad_fast.hip, BF3: dW = act(x)^T dz with the contraction over the rows -- row-major bf16 images in LDS, fragments through
    ds_read_b64_tr_b16.  Plain and BatchNorm-backward right operands, ragged row counts (padding rows must add nothing), every tile
    shape (64- and 128-wide blocks on either side); error against float64 as for the other split-operand GEMMs, and it really ran."""
    from votenet_amd import mlp
    rng = np.random.default_rng(rows + cin + cout)
    x = T((rng.normal(size=(rows, cin)) * 1.2).astype(np.float32), dev)
    sc, sh = T((rng.random(cin) + 0.5).astype(np.float32), dev), T((rng.normal(size=cin) * 0.3).astype(np.float32), dev)
    dz = T(rng.normal(size=(rows, cout)).astype(np.float32), dev)
    a = torch.relu(x.double() * sc.double() + sh.double())
    ref = a.t() @ dz.double()
    bound = float((a.abs().t() @ dz.double().abs()).max())
    errs, outs = [], []
    for mode in (1, 0):
        bf3.votenet_debug_wgrad_bf3(mode)
        dw = torch.zeros(cin, cout, device=dev)
        mlp.wgrad_dense(x, dz, dw, sc, sh, True)
        errs.append(float((dw.double() - ref).abs().max()) / bound)
        outs.append(dw)
    bf3.votenet_debug_wgrad_bf3(1)
    assert errs[0] <= 3e-6 and errs[0] <= 2.0 * errs[1] + 5e-7, "bf16 x 3: %.3g of the accumulated magnitude, fp32 MFMA: %.3g" % tuple(errs)
    assert not torch.equal(outs[0], outs[1])
    # the BatchNorm-backward right operand rebuilt in the loader: dz = A g + B + C z with g = da masked by the ReLU of z
    z = T(rng.normal(size=(rows, cout)).astype(np.float32), dev)
    da = T(rng.normal(size=(rows, cout)).astype(np.float32), dev)
    coef = T(np.concatenate([rng.normal(size=cout), rng.normal(size=cout) * 0.1, rng.normal(size=cout) * 0.2, rng.random(cout) + 0.5,
                             rng.normal(size=cout) * 0.3]).astype(np.float32), dev)
    A, B, C, S, H = [coef[i * cout:(i + 1) * cout].double() for i in range(5)]
    g = torch.where(z.double() * S + H > 0, da.double(), torch.zeros_like(da, dtype=torch.float64))
    dzb = A * g + B + C * z.double()
    ref2 = a.t() @ dzb
    bound2 = float((a.abs().t() @ dzb.abs()).max())
    dw2 = torch.zeros(cin, cout, device=dev)
    mlp.wgrad_dense_bn(x, z, coef, True, dw2, da=da, in_scale=sc, in_shift=sh, in_relu=True)
    assert float((dw2.double() - ref2).abs().max()) / bound2 <= 4e-6
