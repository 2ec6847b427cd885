"""GPU: grouped-point MLP kernels (fused gather + fp32 MFMA GEMM + BN statistics, BN/ReLU folded
into the next load, max over K) against the CPU oracle (oracle/oracle_mlp.c).
Tolerance: 1e-5 relative to the magnitude of the accumulated products (fp32 round-off class)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def close(got, exp, scale=None, tol=1e-5):
    s = np.abs(exp).max() if scale is None else scale
    return np.abs(got - exp).max() <= tol * max(s, 1.0)


@pytest.mark.parametrize("rows,cin,cout", [(128, 16, 64), (1000, 64, 64), (4096, 64, 128), (777, 128, 256), (300, 259, 256),
                                           (512, 256, 259), (130, 128, 79), (64, 6, 64), (5, 3, 7), (2048, 512, 256)])
def test_linear_dense_vs_oracle(hiplib, dev, O, rows, cin, cout):
    from votenet_amd import mlp
    rng = np.random.default_rng(rows + cin)
    x = rng.normal(size=(rows, cin)).astype(np.float32)
    w = (rng.normal(size=(cin, cout)) * np.sqrt(2.0 / cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    z, stats = mlp.linear_dense(T(x, dev), T(w, dev), T(b, dev))
    oz = O.linear(x, w, b)
    bound = (np.abs(x) @ np.abs(w)).max()
    assert close(N(z), oz, bound)
    st = N(stats)
    assert np.allclose(st[:cout], oz.astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)
    assert np.allclose(st[cout:], (oz.astype(np.float64) ** 2).sum(0), rtol=1e-5, atol=1e-3)


def test_linear_dense_with_folded_bn_relu(hiplib, dev, O):
    from votenet_amd import mlp
    rng = np.random.default_rng(3)
    rows, cin, cout = 1500, 64, 128
    zprev = rng.normal(size=(rows, cin)).astype(np.float32) * 3 + 1
    gamma, beta = rng.normal(size=cin).astype(np.float32), rng.normal(size=cin).astype(np.float32)
    w = (rng.normal(size=(cin, cout)) * 0.2).astype(np.float32)
    mean, var = O.bn_stats(zprev)
    a = O.bn_relu(zprev, mean, var, gamma, beta)
    oz = O.linear(a, w, None)
    # device: statistics of zprev come from a first linear with identity weights
    zp, st = mlp.linear_dense(T(zprev, dev), torch.eye(cin, device=dev))
    scale, shift, dmean, dvar = mlp.bn_finalize(rows, st, T(gamma, dev), T(beta, dev))
    assert np.allclose(N(dmean), mean, rtol=1e-5, atol=1e-6) and np.allclose(N(dvar), var, rtol=1e-5, atol=1e-6)
    z, _ = mlp.linear_dense(T(zprev, dev), T(w, dev), None, scale, shift, True)
    assert close(N(z), oz, (np.abs(a) @ np.abs(w)).max())
    assert close(N(mlp.bn_relu(T(zprev, dev), scale, shift)), a)


@pytest.mark.parametrize("b,n,m,k,c,cout", [(2, 300, 20, 16, 0, 64), (2, 300, 20, 16, 3, 64), (1, 500, 33, 64, 128, 128),
                                            (2, 256, 16, 64, 256, 128), (1, 100, 7, 8, 5, 79)])
def test_linear_gather_vs_oracle(hiplib, dev, O, b, n, m, k, c, cout):
    """First SA layer: sample_and_group concat [dxyz, feat] (utils.py:50-57) folded into the GEMM load."""
    from votenet_amd import mlp
    rng = np.random.default_rng(n + c)
    xyz = rng.random((b, n, 3), dtype=np.float32)
    new_xyz = rng.random((b, m, 3), dtype=np.float32)
    feat = rng.normal(size=(b, n, c)).astype(np.float32) if c else None
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    w = (rng.normal(size=(3 + c, cout)) * 0.3).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    x = O.group_concat(xyz, new_xyz, feat, idx).reshape(b * m * k, 3 + c)
    oz = O.linear(x, w, bias)
    z, st = mlp.linear_gather(T(xyz, dev), T(new_xyz, dev), T(feat, dev) if c else None, T(idx, dev), T(w, dev), T(bias, dev))
    assert close(N(z), oz, (np.abs(x) @ np.abs(w)).max())
    assert np.allclose(N(st)[:cout], oz.astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("b,n,m,k,c,cout", [(2, 300, 20, 16, 3, 64), (1, 500, 33, 64, 128, 128), (2, 256, 16, 64, 256, 128),
                                            (3, 97, 5, 7, 8, 32), (1, 64, 3, 5, 16, 4)])
def test_group_linear_vs_oracle(hiplib, dev, O, b, n, m, k, c, cout):
    """First SA layer with the linear map applied before the grouping (votenet_group_linear): P = feat W[3:] per point, then
    z = P[idx] + dxyz W[0:3] + bias per grouped row -- against the oracle's convolution over the sample_and_group concat
    (utils.py:50-57,125-127), to fp32 rounding (1e-5 of the term bound), and against the fused GATHER GEMM."""
    from votenet_amd import mlp
    rng = np.random.default_rng(n + c + cout)
    xyz = rng.random((b, n, 3), dtype=np.float32)
    new_xyz = rng.random((b, m, 3), dtype=np.float32)
    feat = rng.normal(size=(b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, k)).astype(np.int32)
    w = (rng.normal(size=(3 + c, cout)) * 0.3).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    x = O.group_concat(xyz, new_xyz, feat, idx).reshape(b * m * k, 3 + c)
    oz = O.linear(x, w, bias)
    wd = T(w, dev)
    P, _ = mlp.linear_dense(T(feat.reshape(b * n, c), dev), wd[3:], want_stats=False)
    z, st = mlp.group_linear(T(xyz, dev), T(new_xyz, dev), T(idx, dev), P, wd[:3], T(bias, dev))
    bound = (np.abs(x) @ np.abs(w)).max()
    assert close(N(z), oz, bound)
    assert np.allclose(N(st)[:cout], oz.astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)
    assert np.allclose(N(st)[cout:], (oz.astype(np.float64) ** 2).sum(0), rtol=1e-5, atol=1e-3)
    zg, _ = mlp.linear_gather(T(xyz, dev), T(new_xyz, dev), T(feat, dev), T(idx, dev), wd, T(bias, dev))
    assert close(N(z), N(zg), bound)
    z2, none = mlp.group_linear(T(xyz, dev), T(new_xyz, dev), T(idx, dev), P, wd[:3], None, want_stats=False)
    assert none is None and close(N(z2) + bias, N(z), bound)


def test_group_linear_argument_errors(hiplib, dev):
    from votenet_amd import mlp, _lib
    xyz = torch.rand(1, 10, 3, device=dev)
    idx = torch.zeros(1, 2, 4, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.InvalidArgumentError):  # cout must be a power of two in [4, 1024]
        mlp.group_linear(xyz, xyz[:, :2].contiguous(), idx, torch.zeros(10, 12, device=dev), torch.zeros(3, 12, device=dev))


@pytest.mark.parametrize("rows,cin,cout", [(1024, 64, 128), (4096, 128, 256), (640, 32, 128)])
def test_linear_pool_epilogue_matches_separate_pass(hiplib, dev, rows, cin, cout):
    """votenet_mlp_linear_pool + votenet_bn_pool_finalize (max-pool started in the GEMM epilogue on raw z) against
    votenet_mlp_linear + votenet_bn_relu_max (utils.py:132), including negative and zero BatchNorm scales."""
    from votenet_amd import mlp, _lib
    g = torch.Generator().manual_seed(rows + cout)
    x = torch.randn(rows, cin, generator=g).to(dev)
    w = (torch.randn(cin, cout, generator=g) * 0.2).to(dev)
    sc_in = (torch.randn(cin, generator=g) * 0.3 + 1).to(dev)
    sh_in = (torch.randn(cin, generator=g) * 0.2).to(dev)
    z_ref, st_ref = mlp.linear_dense(x, w, None, sc_in, sh_in, True)
    z, st, pool = mlp.linear_dense_pool(x, w, 64, None, sc_in, sh_in, True)
    assert torch.equal(z, z_ref)
    assert np.allclose(N(st), N(st_ref), rtol=1e-6, atol=1e-4)
    scale = (torch.randn(cout, generator=g)).to(dev)
    scale[::7] = 0.0  # a zero scale: every row ties
    shift = (torch.randn(cout, generator=g) * 0.5).to(dev)
    for relu in (True, False):
        out_ref, arg_ref = mlp.bn_relu_max(z_ref, 64, scale, shift, relu, want_argmax=True)
        out, arg = mlp.bn_pool_finalize(pool, scale, shift, relu, want_argmax=True)
        assert torch.equal(out, out_ref)
        # the arg-max may differ among rows that tie AFTER BatchNorm(+ReLU); the value at it must be the maximum
        act = z_ref.view(rows // 64, 64, cout) * scale + shift
        if relu:
            act = torch.where(act > 0, act, torch.zeros_like(act))
        picked = act.gather(1, arg.long()[:, None, :])[:, 0, :]
        assert torch.equal(picked, out_ref)
        if not relu:  # without the ReLU plateau only exact float ties can differ (a zero scale ties every row)
            nz = scale != 0
            assert (arg == arg_ref)[:, nz].float().mean() > 0.99
        out3, arg3, zsel = mlp.bn_pool_finalize(pool, scale, shift, relu, want_argmax=True, want_zsel=True)
        assert torch.equal(out3, out) and torch.equal(arg3, arg)
        assert torch.equal(zsel, z_ref.view(rows // 64, 64, cout).gather(1, arg.long()[:, None, :])[:, 0, :])
    z2, _, pool2 = mlp.linear_dense_pool(x, w, 64, None, sc_in, sh_in, True, keep_z=False)
    assert z2 is None and all(torch.equal(a, b) for a, b in zip(pool, pool2))
    assert not mlp.linear_pool_supported(rows, cin, 64, 64) and not mlp.linear_pool_supported(rows, cin, cout, 32)
    with pytest.raises(_lib.InvalidArgumentError):
        mlp.linear_dense_pool(x, w, 32)


def test_sa_mlp_stack_cfg1(hiplib, dev, O):
    """BASELINE config 1 end to end: 2048 pts -> FPS 512 -> ball r=0.2 K=32 -> MLP 64,64,128 (BNReLU) -> max over K."""
    import cases
    from votenet_amd import mlp, tf_grouping, tf_sampling
    xyz = cases.cfg1_cloud()
    rng = np.random.default_rng(1)
    dims = [6, 64, 64, 128]
    ws = [(rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.normal(size=dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    gs = [1 + 0.1 * rng.normal(size=dims[i + 1]).astype(np.float32) for i in range(3)]
    be = [0.1 * rng.normal(size=dims[i + 1]).astype(np.float32) for i in range(3)]
    # oracle
    fidx = O.farthest_point_sample(512, xyz)
    new_xyz = O.gather_point(xyz, fidx)
    idx, _ = O.query_ball_point(0.2, 32, xyz, new_xyz)
    a = O.group_concat(xyz, new_xyz, xyz, idx).reshape(-1, 6)
    for i in range(3):
        zz = O.linear(a, ws[i], bs[i])
        mean, var = O.bn_stats(zz)
        a = O.bn_relu(zz, mean, var, gs[i], be[i])
    exp = O.max_over_k(a, 32)
    # device
    x = T(xyz, dev)
    dfidx = tf_sampling.farthest_point_sample(512, x)
    dnew = tf_sampling.gather_point(x, dfidx)
    didx, _ = tf_grouping.query_ball_point(0.2, 32, x, dnew)
    rows = 512 * 32
    z, st = mlp.linear_gather(x, dnew, x, didx, T(ws[0], dev), T(bs[0], dev))
    sc, sh, _, _ = mlp.bn_finalize(rows, st, T(gs[0], dev), T(be[0], dev))
    for i in (1, 2):
        z, st = mlp.linear_dense(z, T(ws[i], dev), T(bs[i], dev), sc, sh, True)
        sc, sh, _, _ = mlp.bn_finalize(rows, st, T(gs[i], dev), T(be[i], dev))
    out, arg = mlp.bn_relu_max(z, 32, sc, sh, True, want_argmax=True)
    got = N(out)
    assert got.shape == (512, 128)
    assert np.abs(got - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max())
    # argmax points at a row attaining the max
    full = N(mlp.bn_relu(z, sc, sh)).reshape(512, 32, 128)
    assert (np.take_along_axis(full, N(arg)[:, None, :].astype(np.int64), 1)[:, 0, :] == got).all()


@pytest.mark.parametrize("rows,cin,cout", [(4096, 64, 128), (2048, 128, 256), (777, 48, 79), (640, 256, 128)])
def test_consumer_side_bn_finalize(hiplib, dev, rows, cin, cout):
    """struct votenet_bn_raw: a consumer that derives BatchNorm scale / shift from the producer's raw sums in its prologue
    (and records scale | shift | mean | var) against the stand-alone votenet_bn_finalize + the same consumer."""
    from votenet_amd import mlp
    g = torch.Generator().manual_seed(rows + cin)
    x0 = torch.randn(rows, 32, generator=g).to(dev)
    w0 = (torch.randn(32, cin, generator=g) * 0.3).to(dev)
    gamma = (torch.randn(cin, generator=g) * 0.3 + 1).to(dev)
    beta = (torch.randn(cin, generator=g) * 0.2).to(dev)
    w = (torch.randn(cin, cout, generator=g) * 0.2).to(dev)
    zp, st = mlp.linear_dense(x0, w0)
    sc, sh, mean, var = mlp.bn_finalize(rows, st, gamma, beta)
    z_ref, st_ref = mlp.linear_dense(zp, w, None, sc, sh, True)
    pend = mlp.PendingBN(st, gamma, beta, rows)
    z, st2 = mlp.linear_dense(zp, w, None, None, None, True, in_bn=pend)
    assert pend.done and torch.equal(z, z_ref) and torch.equal(st2, st_ref)
    for got, exp in zip(pend.out, (sc, sh, mean, var)):
        assert torch.equal(got, exp)
    # a second consumer of the same BatchNorm reads the recorded vectors
    z3, _ = mlp.linear_dense(zp, w, None, None, None, True, in_bn=pend)
    assert torch.equal(z3, z_ref)
    # the activation pass and the pooled finalize as consumers
    pend = mlp.PendingBN(st, gamma, beta, rows)
    assert torch.equal(mlp.bn_relu(zp, None, None, True, bn=pend), mlp.bn_relu(zp, sc, sh, True))
    assert torch.equal(pend.out[2], mean) and torch.equal(pend.out[3], var)
    if mlp.linear_pool_supported(rows, 32, cin, 64):
        _, stp, pool = mlp.linear_dense_pool(x0, w0, 64)
        pend = mlp.PendingBN(stp, gamma, beta, rows)
        out, arg = mlp.bn_pool_finalize(pool, None, None, True, want_argmax=True, bn=pend)
        scp, shp, _, _ = mlp.bn_finalize(rows, stp, gamma, beta)
        out_ref, arg_ref = mlp.bn_pool_finalize(pool, scp, shp, True, want_argmax=True)
        assert torch.equal(out, out_ref) and torch.equal(arg, arg_ref) and torch.equal(pend.out[0], scp)
    # finalize() on a BatchNorm nobody consumed launches the stand-alone kernel
    pend = mlp.PendingBN(st, gamma, beta, rows)
    s2, h2 = pend.finalize()
    assert torch.equal(s2, sc) and torch.equal(h2, sh)


@pytest.mark.parametrize("form", ["narrow", "assembled", "stored"])
def test_sa_module_cfg1_every_first_layer_form_vs_oracle(hiplib, dev, O, form):
    """BASELINE config 1 through pointnet2.SAModule, against the CPU oracle's materialised grouped convolution: the first layer never
    stored and rebuilt from eight floats per row (leaf module, csrc/narrow.hip), assembled inside its consumers from the per-point
    table (csrc/assemble.hip), and stored (votenet_group_linear)."""
    import cases
    from votenet_amd import pointnet2 as P
    xyz = cases.cfg1_cloud()
    rng = np.random.default_rng(1)
    dims = [6, 64, 64, 128]
    ws = [(rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i])).astype(np.float32) for i in range(3)]
    bs = [rng.normal(size=dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    gs = [1 + 0.1 * rng.normal(size=dims[i + 1]).astype(np.float32) for i in range(3)]
    be = [0.1 * rng.normal(size=dims[i + 1]).astype(np.float32) for i in range(3)]
    fidx = O.farthest_point_sample(512, xyz)
    new_xyz = O.gather_point(xyz, fidx)
    idx, _ = O.query_ball_point(0.2, 32, xyz, new_xyz)
    a = O.group_concat(xyz, new_xyz, xyz, idx).reshape(-1, 6)
    for i in range(3):
        zz = O.linear(a, ws[i], bs[i])
        mean, var = O.bn_stats(zz)
        a = O.bn_relu(zz, mean, var, gs[i], be[i])
    exp = O.max_over_k(a, 32)
    store = P.ParamStore(dev)
    mod = P.SAModule(store, "sa", 512, 0.2, 32, 3, [64, 64, 128], leaf=(form == "narrow"))
    store.materialize(0)
    for i in range(3):
        store["sa/conv%d/W" % i].copy_(T(ws[i], dev))
        store["sa/conv%d/b" % i].copy_(T(bs[i], dev))
        store["sa/conv%d/gamma" % i].copy_(T(gs[i], dev))
        store["sa/conv%d/beta" % i].copy_(T(be[i], dev))
    old = P.ASSEMBLE_FIRST
    P.ASSEMBLE_FIRST = form != "stored"
    try:
        x = T(xyz, dev)
        tape = []
        nx, out, didx = mod.forward(x, x, tape=tape)
    finally:
        P.ASSEMBLE_FIRST = old
    assert tape[0]["recs"][0]["kind"] == {"narrow": "narrow", "assembled": "assembled", "stored": "gather"}[form]
    assert np.array_equal(N(didx), idx) and np.array_equal(N(nx), new_xyz)
    got = N(out)[0]
    assert np.abs(got - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max())
