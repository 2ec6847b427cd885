"""GPU: the PIECE layout of a set-abstraction level (csrc/half.hip and the *_half entries of include/votenet_hip.h) against the full
layout of the same level: the rows a ball repeats (tf_grouping_g.cu:26-29 pads with the first hit) dropped by pieces of 16, the ball's
slot 0 standing for the dropped copies with a weight.  Same values per row, same sums up to their association."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / max(1e-12, float(b.double().abs().max())))


PS, NP, TP = 16, 4, 8  # rows per piece, pieces per ball, pieces per 128-row GEMM tile


def layout_reference(cnt):
    """numpy restatement of votenet_half_groups: a ball keeps pieces 0 .. ceil(pts_cnt / 16) - 1; centres in ascending order, then all-copy
    pieces until the count is a multiple of 8; a ball's slot 0 stands for its dropped pieces."""
    G = cnt.size
    kc = np.clip((np.maximum(cnt, 1) + PS - 1) // PS, 1, NP)
    pos = np.full((G, NP - 1), -1, np.int64)
    hc = [c * NP for c in range(G)]
    for c in range(G):
        for j in range(1, kc[c]):
            pos[c, j - 1] = len(hc) - G
            hc.append(c * NP + j)
    c = 0
    while len(hc) % TP:
        for j in range(1, NP):
            if len(hc) % TP and pos[c, j - 1] < 0:
                pos[c, j - 1] = len(hc) - G
                hc.append(c * NP + j)
        c += 1
    wh = np.ones(len(hc), np.float32)
    wh[:G] = 1 + PS * (NP - 1 - (pos >= 0).sum(1))
    return pos, np.array(hc), wh


@pytest.mark.parametrize("G,kind", [(8, "mixed"), (4096, "mixed"), (64, "full"), (64, "empty"), (24, "one"), (16384, "mixed")])
def test_piece_layout_of_a_level(hiplib, dev, G, kind):
    from votenet_amd import mlp as M
    assert M.PIECE == PS and M.BALL_PIECES == NP
    rng = np.random.RandomState(G)
    cnt = {"mixed": rng.randint(0, 65, G), "full": np.full(G, 64), "empty": np.zeros(G, np.int64),
           "one": np.where(np.arange(G) == 7, 40, 3)}[kind].astype(np.int32)
    half = M.half_groups(torch.from_numpy(cnt).to(dev).view(1, G)).resolve()
    pos, hc, wh = layout_reference(cnt)
    assert half.nh == len(hc) and half.nh % TP == 0
    assert np.array_equal(half.pos.cpu().numpy().reshape(G, NP - 1), pos) and np.array_equal(half.hc.cpu().numpy(), hc)
    assert np.array_equal(half.wh.cpu().numpy(), wh)
    # every slot of the full layout is represented by exactly one compact row, a compact row by as many slots as its weight says
    fi = half.full_index()
    assert torch.equal(torch.bincount(fi, minlength=half.rows).float(), half.row_weights())
    # twice the same layout (a scan, not atomics)
    again = M.half_groups(torch.from_numpy(cnt).to(dev).view(1, G)).resolve()
    assert torch.equal(again.hc, half.hc) and torch.equal(again.pos, half.pos)


def _rows_of(half, dev):
    """compact row -> the full-layout row of the slot it holds."""
    code = half.hc.long()[:, None]
    s = torch.arange(PS, device=dev)[None, :]
    return ((code // NP) * 64 + (code % NP) * PS + s).reshape(-1)


def _totals(t_full, half, dev):
    """Sum of the full-layout rows every compact row stands for."""
    out = torch.zeros(half.rows, t_full.shape[1], dtype=torch.float64, device=dev)
    out.index_add_(0, half.full_index(), t_full.double())
    return out


@pytest.mark.parametrize("b,n,m,cf,c2,radius", [(2, 600, 64, 128, 256, 0.5), (1, 500, 48, 32, 128, 0.42), (2, 400, 32, 64, 256, 0.62)])
def test_stage_kernels_on_compact_rows_match_the_full_layout(hiplib, dev, gemm_form, b, n, m, cf, c2, radius):
    from votenet_amd import mlp as M
    from votenet_amd import tf_grouping, tf_sampling
    k, c0, c1 = 64, 128, 128
    g = torch.Generator().manual_seed(7 * n + cf)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    xyz = (torch.rand(b, n, 3, generator=g) * 2.0).to(dev)
    feat = rnd(b, n, cf)
    new_xyz = tf_sampling.gather_point(xyz, tf_sampling.farthest_point_sample(m, xyz))
    idx, cnt = tf_grouping.query_ball_point(radius, k, xyz, new_xyz)
    assert bool((cnt <= 31).any()), cnt.flatten().tolist()
    G, rows = b * m, b * m * k
    w0, b0, w1, w2, b2 = rnd(3 + cf, c0) * 0.3, rnd(c0) * 0.1, rnd(c0, c1) * 0.2, rnd(c1, c2) * 0.2, rnd(c2) * 0.1
    wx = w0[:3].contiguous()
    w1T, w2T = w1.t().contiguous(), w2.t().contiguous()
    img = M.SplitImages([w1, w1T, w2])
    img.refresh()
    P, _ = M.linear_dense(feat.reshape(b * n, cf), w0[3:].contiguous(), b0, want_stats=False)
    # ---- geometry
    geo, cntv, mom = M.assemble_rows(xyz, new_xyz, idx, pts_cnt=cnt)
    half = M.half_groups(cnt)
    _, cntv_h, mom_h = M.assemble_rows_half(xyz, new_xyz, idx, cnt, half)
    half.resolve()
    full = _rows_of(half, dev)
    saved = 1.0 - half.rows / rows
    assert 0.0 < saved <= 0.75 and half.rows % 128 == 0, cnt.flatten().tolist()
    assert torch.equal(half.geo, geo[full]) and torch.equal(cntv_h, cntv) and relerr(mom_h, mom) < 1e-12
    assert torch.equal(half.full_rows(half.geo), geo)  # a dropped slot is a copy of slot 0: the expansion IS the full layout
    # ---- forward: layer 1 (assembled loader), layer 2 (pool in the epilogue) with ONE BatchNorm per layer for both layouts
    st0 = M.assemble_stats(P, cntv, wx, mom)
    gamma0 = rnd(c0) * 0.2 + 1.0
    bn0 = M.PendingBN(st0, gamma0, rnd(c0) * 0.1, rows)
    z1, st1 = M.assembled_linear(geo, P, wx, w1, None, bn0)
    z1h, st1h = M.assembled_linear(half.geo, P, wx, w1, None, bn0, half=half)
    assert torch.equal(z1h, z1[full])
    assert relerr(st1h[:c1], st1[:c1]) < 1e-6 and relerr(st1h[c1:], st1[c1:]) < 1e-6
    bn1 = M.PendingBN(st1, rnd(c1) * 0.2 + 1.0, rnd(c1) * 0.1, rows)
    bn1.finalize()
    gamma2 = rnd(c2) * 0.3 + 0.2  # some negative gammas: the min side of the pool
    _, st2, pool = M.linear_dense_pool(z1, w2, k, b2, bn1.scale, bn1.shift, True, keep_z=False)
    _, st2h, poolh = M.linear_dense_pool(z1h, w2, k, b2, bn1.scale, bn1.shift, True, keep_z=False, half=half, gamma=gamma2)
    assert relerr(st2h[:c2], st2[:c2]) < 1e-6 and relerr(st2h[c2:], st2[c2:]) < 1e-6
    bn2 = M.PendingBN(st2, gamma2, rnd(c2) * 0.1, rows)
    bn2.finalize()
    out, arg, zsel = M.bn_pool_finalize(pool, bn2.scale, bn2.shift, True, want_argmax=True, want_zsel=True)
    outh, argh, zselh = M.bn_pool_finalize(poolh, bn2.scale, bn2.shift, True, want_argmax=True, want_zsel=True, half=half)
    assert torch.equal(outh, out) and torch.equal(zselh, zsel) and torch.equal(argh, arg)
    assert int((bn2.scale < 0).sum()) > 0
    # ---- backward of the pooled layer in Gram form
    gout, coef2 = rnd(G, c2), rnd(5 * c2) * 0.3
    coef2[3 * c2:4 * c2], coef2[4 * c2:] = bn2.scale, bn2.shift
    aff1 = torch.stack([bn1.scale, bn1.shift]).contiguous()
    Gm, Gh = M.gram(z1, aff1, True), M.gram(z1h, aff1, True, half=half)
    dw2, dw2h = torch.zeros(c1, c2, device=dev), torch.zeros(c1, c2, device=dev)
    M.pool_wgrad(z1, bn1.scale, bn1.shift, True, Gm, w2, b2, coef2, True, gout, arg, zsel, k, dw2)
    M.pool_wgrad(z1h, bn1.scale, bn1.shift, True, Gh, w2, b2, coef2, True, gout, arg, zsel, k, dw2h, half=half)
    assert relerr(Gh[:c1], Gm[:c1]) < 2e-6 and relerr(Gh[c1], Gm[c1]) < 2e-6 and relerr(dw2h, dw2) < 1e-5
    # the two forms of the piece layout's arg-max gather: walking centres (default, round 5) and walking pieces -- the same sums
    assert M.POOL_WGRAD_CENTRES
    M.POOL_WGRAD_CENTRES = False
    try:
        Gp, dw2p = M.gram(z1h, aff1, True, half=half), torch.zeros(c1, c2, device=dev)
        M.pool_wgrad(z1h, bn1.scale, bn1.shift, True, Gp, w2, b2, coef2, True, gout, arg, zsel, k, dw2p, half=half)
    finally:
        M.POOL_WGRAD_CENTRES = True
    assert relerr(dw2p, dw2) < 1e-5 and relerr(dw2p, dw2h) < 2e-6 and relerr(Gp[c1], Gh[c1]) < 2e-6  # (row c1: the weighted column sums)
    below = (bn1.scale, bn1.shift, bn1.mean, bn1.var, True)
    da1, sums1 = M.pool_dgrad(z1, bn1.scale, bn1.shift, True, w2, b2, w2T, coef2, True, gout, arg, zsel, k, below=below)
    da1h, sums1h = M.pool_dgrad(z1h, bn1.scale, bn1.shift, True, w2, b2, w2T, coef2, True, gout, arg, zsel, k, below=below, half=half)
    assert relerr(da1h, _totals(da1, half, dev)) < 2e-6
    zh1 = (z1.double() - bn1.mean.double()) / torch.sqrt(bn1.var.double() + M.BN_EPS)
    scale1 = torch.cat([da1.double().abs().sum(0), (da1.double() * zh1).abs().sum(0)])
    assert float(((sums1h - sums1).abs() / (scale1 + 1e-30)).max()) < 1e-5
    # ---- backward of layer 1 over the assembled layer 0 (da: random per full row; the compact rows carry the totals)
    da1 = rnd(rows, c1)
    da1h = _totals(da1, half, dev).float()
    coef1 = rnd(5 * c1) * 0.3
    coef1[3 * c1:4 * c1], coef1[4 * c1:] = bn1.scale, bn1.shift
    dw1, dw1h = torch.zeros(c0, c1, device=dev), torch.zeros(c0, c1, device=dev)
    M.assembled_wgrad_bn(geo, P, wx, bn0.scale, bn0.shift, True, z1, coef1, True, da1, dw1)
    M.assembled_wgrad_bn(half.geo, P, wx, bn0.scale, bn0.shift, True, z1h, coef1, True, da1h, dw1h, half=half)
    assert relerr(dw1h, dw1) < 1e-5
    bn0.finalize()
    below0 = (bn0.scale, bn0.shift, bn0.mean, bn0.var, True)
    da0, sums0 = M.assembled_dgrad_bn_reduce(z1, coef1, True, w1T, da1, geo, P, wx, below0)
    da0h, sums0h = M.assembled_dgrad_bn_reduce(z1h, coef1, True, w1T, da1h, half.geo, P, wx, below0, half=half)
    assert relerr(da0h, _totals(da0, half, dev)) < 1e-5
    z0 = M.assemble_z0(geo, P, wx)
    scale0 = torch.cat([da0.double().abs().sum(0), (da0.double() * ((z0.double() - bn0.mean.double()) / torch.sqrt(bn0.var.double() + M.BN_EPS))).abs().sum(0)])
    assert float(((sums0h - sums0).abs() / (scale0 + 1e-30)).max()) < 1e-5
    # ---- backward of layer 0: scatter to the points and the xyz rows of dW
    coef0 = rnd(5 * c0) * 0.3
    coef0[3 * c0:4 * c0], coef0[4 * c0:] = bn0.scale, bn0.shift
    dwx, dwxh = torch.zeros(3, c0, device=dev), torch.zeros(3, c0, device=dev)
    da0t = _totals(da0, half, dev).float()
    S, _ = M.group_linear_backward_assembled(xyz, new_xyz, idx, cnt, P, wx, da0, coef0, True, dwx)
    # over the rows bucketed by point: every compact row once, the rows of a point consecutive
    M.half_sort_rows(half, b * n)
    order = half.order.long()
    assert torch.equal(torch.sort(order)[0], torch.arange(half.rows, device=dev))
    prow_sorted = half.geo[order, 3].view(torch.int32)
    assert bool((prow_sorted[1:] >= prow_sorted[:-1]).all())
    Sh = M.group_linear_backward_half(half, b, n, P, wx, da0t, coef0, True, dwxh)
    assert relerr(Sh, S) < 1e-5 and relerr(dwxh, dwx) < 1e-4
    # ---- the same backward of layer 0 DECOMPOSED over the points (round 4): a plain input-gradient GEMM, the masked scatter that
    # reduces the BatchNorm backward itself, a pass over the points -- against the epilogue reduce + the sorted scatter with the
    # coefficient vector those sums give (the chain the model ran before)
    da0p = M.dgrad_bn_half(z1h, coef1, True, w1T, da1h, half)
    assert relerr(da0p, da0h) < 1e-6  # the same GEMM without the epilogue's reduce
    bn0v = (bn0.scale, bn0.shift, bn0.mean, bn0.var)
    dg_ref, db_ref = torch.zeros(c0, device=dev), torch.zeros(c0, device=dev)
    coef_ref = M.bn_backward_coef(rows, *bn0v, gamma0, sums0h, dg_ref, db_ref)
    dwx_ref = torch.zeros(3, c0, device=dev)
    S_ref = M.group_linear_backward_half(half, b, n, P, wx, da0h, coef_ref, True, dwx_ref)
    for tails in (True, False):  # the coefficient vector from the kernel's tail / from the separate launch
        prev, M.COEF_TAIL = M.COEF_TAIL, tails
        try:
            dg, db, dwx_dec = torch.zeros(c0, device=dev), torch.zeros(c0, device=dev), torch.zeros(3, c0, device=dev)
            S_dec, coef_dec = M.group_linear_backward_decomposed(half, b, n, P, wx, da0p, bn0v, True, (rows, gamma0, dg, db), cntv_h, mom_h, dwx_dec)
        finally:
            M.COEF_TAIL = prev
        for q in range(3):  # A, B, C (the last two blocks are the layer's scale / shift)
            assert relerr(coef_dec[q * c0:(q + 1) * c0], coef_ref[q * c0:(q + 1) * c0]) < 2e-5, q
        assert torch.equal(coef_dec[3 * c0:], coef_ref[3 * c0:])
        assert relerr(dg, dg_ref) < 2e-5 and relerr(db, db_ref) < 2e-5
        assert relerr(S_dec, S_ref) < 2e-5 and relerr(dwx_dec, dwx_ref) < 1e-4
    img.close()


def test_model_with_and_without_the_half_group_layout(hiplib, dev):
    """The whole hot path with sa1 (narrow first layer) and sa2 / sa3 / sa4 (assembled first layer) on the piece layout against the full layout: same outputs and losses to fp32
    rounding (the BatchNorm sums are associated differently), the same gradient in the L2 sense (tests/test_gpu_narrow.py: two fp32
    evaluations of a forward pass move ReLU / arg-max decisions), fewer grouped rows."""
    from votenet_amd import loss as VL
    from votenet_amd import model as VM
    from votenet_amd import pointnet2 as P
    from votenet_amd import synth
    b, n = 2, 8192
    x = torch.from_numpy(synth.room_batch(b, n, 9)).to(dev)
    gt = VL.gt_to_device(synth.room_gt(b, n, 9), dev)
    net = VM.VoteNetHotPath(dev, seed=6, npoints=(1024, 512, 256, 128))
    fixed = {}

    def once():
        net.store.grad.zero_()
        net.store.refresh_transposes()
        tape = []
        out = net.forward(x, tape)
        losses, cot = VL.votenet_loss(out, gt)
        cot = fixed.setdefault("cot", cot)
        net.backward(tape, cot)
        torch.cuda.synchronize()
        return tape, out["proposals_output"].clone(), losses.clone(), net.store.grad.clone()
    assert P.HALF_GROUPS  # the default
    tape, o1, l1, g1 = once()
    sas = [t for t in tape if t.get("op") == "sa"]
    halves = [t["recs"][0].get("half") for t in sas]
    assert [h is not None for h in halves] == [True, True, True, True, True]
    for t, h in zip(sas[:4], halves[:4]):
        assert t["recs"][1]["z"].shape[0] == h.rows < t["recs"][0]["rows"]
    # the proposal module groups the votes, which exist only inside the step: its count stays on the device
    assert halves[4].nh_limit is not None and halves[4].true_count() * 16 < sas[4]["recs"][0]["rows"] == halves[4].rows
    r1 = net.predict(x, 0.25, batch_statistics=True)  # inference through the same layout
    P.HALF_GROUPS = False
    try:
        tape, o0, l0, g0 = once()
        assert all(t["recs"][0].get("half") is None for t in tape if t.get("op") == "sa")
        r0 = net.predict(x, 0.25, batch_statistics=True)
    finally:
        P.HALF_GROUPS = True
    assert relerr(o1, o0) < 5e-5 and relerr(l1, l0) < 5e-5
    assert float((g1.double() - g0.double()).norm() / g0.double().norm()) < 1e-2
    assert len(r1) == len(r0)


@pytest.mark.parametrize("b,n,m,c,radius,b0_shift", [(2, 700, 64, 3, 0.5, 0.0), (1, 500, 48, 0, 0.45, 0.0), (2, 700, 64, 3, 0.5, 40.0)])
def test_narrow_stage_kernels_on_compact_rows_match_the_full_layout(hiplib, dev, gemm_form, b, n, m, c, radius, b0_shift):
    """sa1's form (csrc/narrow.hip: z0 rebuilt from eight floats per row) on the compact rows.  b0_shift = 40: a first-layer bias that
    puts |mean| / std of z0 at ~50 (the masked input-gradient tail derives sum g zhat as a DIFFERENCE of two sums of that
    size, (sum_d W0 UG + b0 s1) - mean s1; it does so in double on double totals, so the bias cancels to the last bit of its fp32
    value and what remains is the accuracy of the fp32 per-lane partials of UG and s1 -- the same as the round-3 epilogue's)."""
    from votenet_amd import mlp as M
    from votenet_amd import tf_grouping, tf_sampling
    k, c0, c1 = 64, 64, 64
    g = torch.Generator().manual_seed(11 * n + c)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    xyz = (torch.rand(b, n, 3, generator=g) * 2.0).to(dev)
    feat = xyz.clone() if c == 3 else None
    new_xyz = tf_sampling.gather_point(xyz, tf_sampling.farthest_point_sample(m, xyz))
    idx, cnt = tf_grouping.query_ball_point(radius, k, xyz, new_xyz)
    assert bool((cnt <= 31).any()) and bool((cnt > 31).any()), cnt.flatten().tolist()
    rows, k0 = b * m * k, 3 + c
    w0, b0, w1 = rnd(k0, c0) * 0.5, rnd(c0) * 0.1 + b0_shift, rnd(c0, c1) * 0.2
    w1T = w1.t().contiguous()
    img = M.SplitImages([w1, w1T])
    img.refresh()
    u8, mom = M.narrow_rows(xyz, new_xyz, feat, idx)
    half = M.half_groups(cnt)
    _, mom_h = M.narrow_rows_half(xyz, new_xyz, feat, idx, cnt, half)
    half.resolve()
    full = _rows_of(half, dev)
    assert half.rows < rows and torch.equal(half.u8, u8[full]) and relerr(mom_h, mom) < 1e-12
    st0 = M.narrow_stats(rows, mom_h, w0, b0)
    bn0 = M.PendingBN(st0, rnd(c0) * 0.2 + 1.0, rnd(c0) * 0.1, rows)
    bn0.finalize()
    z1, st1 = M.narrow_linear(u8, w0, b0, w1, None, bn0)
    z1h, st1h = M.narrow_linear(half.u8, w0, b0, w1, None, bn0, half=half)
    assert torch.equal(z1h, z1[full]) and relerr(st1h[:c1], st1[:c1]) < 1e-6 and relerr(st1h[c1:], st1[c1:]) < 1e-6
    da1 = rnd(rows, c1)
    da1h = _totals(da1, half, dev).float()
    coef1 = rnd(5 * c1) * 0.3
    bn1 = M.PendingBN(st1, rnd(c1) * 0.2 + 1.0, rnd(c1) * 0.1, rows)
    bn1.finalize()
    coef1[3 * c1:4 * c1], coef1[4 * c1:] = bn1.scale, bn1.shift
    dw1, dw1h = torch.zeros(c0, c1, device=dev), torch.zeros(c0, c1, device=dev)
    M.narrow_wgrad_bn(u8, w0, b0, bn0.scale, bn0.shift, True, z1, coef1, True, da1, dw1)
    M.narrow_wgrad_bn(half.u8, w0, b0, bn0.scale, bn0.shift, True, z1h, coef1, True, da1h, dw1h, half=half)
    assert relerr(dw1h, dw1) < 1e-5
    below0 = (bn0.scale, bn0.shift, bn0.mean, bn0.var, True)
    sums, ug = M.narrow_dgrad_bn_reduce(z1, coef1, True, w1T, da1, u8, w0, b0, below0)
    sums_h, ug_h = M.narrow_dgrad_bn_reduce(z1h, coef1, True, w1T, da1h, half.u8, w0, b0, below0, half=half)
    # scale of the sums: the absolute sums of their terms
    dz1 = coef1[:c1] * da1 * (z1 * bn1.scale + bn1.shift > 0) + coef1[c1:2 * c1] + coef1[2 * c1:3 * c1] * z1
    da0 = dz1.double() @ w1T.double()
    z0 = M.narrow_z0(u8, w0, b0).double()
    zh0 = (z0 - bn0.mean.double()) / torch.sqrt(bn0.var.double() + M.BN_EPS)
    scale = torch.cat([da0.abs().sum(0), (da0 * zh0).abs().sum(0)])
    assert float(((sums_h - sums).abs() / (scale + 1e-30)).max()) < 1e-5
    uscale = (u8.double().abs().t() @ da0.abs())
    assert float(((ug_h - ug).abs() / (uscale + 1e-30)).max()) < 1e-5
    # ---- round 4: the first layer's ReLU mask recorded by the forward GEMM and read by the input-gradient epilogue (EPI 7) instead of a
    # rebuild of z0 per accumulator element; the second BatchNorm-backward sum derived in the coefficient tail from ug and the first
    z1m, st1m, mask = M.narrow_linear(half.u8, w0, b0, w1, None, bn0, half=half, want_mask=True)
    assert torch.equal(z1m, z1h) and torch.equal(st1m, st1h)
    z0h = M.narrow_z0(half.u8, w0, b0)
    act = (torch.clamp_min(z0h * bn0.scale + bn0.shift, 0.0) > 0)  # the loader's own test: relu(bn0(z0)) > 0
    bits = torch.arange(16, device=dev)
    want = (act.view(half.rows, c0 // 16, 16).long() << bits).sum(-1)
    assert torch.equal(mask.long() & 0xffff, want)
    gamma0 = rnd(c0) * 0.2 + 1.0
    ref_dg, ref_db = torch.zeros(c0, device=dev), torch.zeros(c0, device=dev)
    coef_ref, ug_ref = M.narrow_dgrad_bn_reduce(z1h, coef1, True, w1T, da1h, half.u8, w0, b0, below0, tail=(rows, gamma0, ref_dg, ref_db), half=half)
    dg, db = torch.zeros(c0, device=dev), torch.zeros(c0, device=dev)
    coef_m, ug_m = M.narrow_dgrad_bn_reduce(z1h, coef1, True, w1T, da1h, half.u8, w0, b0, below0, tail=(rows, gamma0, dg, db), half=half, mask=mask)
    assert float(((ug_m - ug_ref).abs() / (uscale + 1e-30)).max()) < 1e-5
    assert torch.equal(coef_m[3 * c0:], coef_ref[3 * c0:])
    # A, B, C follow from the sums: hold them to the sums' own scale (B, C divide differences of large sums)
    for q in range(3):
        assert relerr(coef_m[q * c0:(q + 1) * c0], coef_ref[q * c0:(q + 1) * c0]) < 1e-4, q
    assert relerr(dg, ref_dg) < 1e-4 and relerr(db, ref_db) < 1e-4
    img.close()


@pytest.mark.parametrize("radius", [0.1, 0.25, 0.6])
def test_level_with_the_count_on_the_device_and_the_xyz_gradient(hiplib, dev, radius):
    """An SA level whose geometry is made inside the step (the proposal module, model.py:89): the piece count never reaches the host,
    every kernel stops at the count it reads on the device; and the gradient with respect to the coordinates on the piece layout.
    Against the same level with the count on the host (features / parameter gradients) and on the full layout (all of it)."""
    from votenet_amd import pointnet2 as P
    b, n, m, cin = 2, 1024, 128, 64
    store = P.ParamStore(dev)
    mod = P.SAModule(store, "t", m, radius, 64, cin, [128, 128, 128], mlp2=[128, 64])
    store.materialize(5)
    g = torch.Generator().manual_seed(9)
    xyz = torch.rand(b, n, 3, generator=g).to(dev)
    pts = torch.randn(b, n, cin, generator=g).to(dev)
    gout = None

    def run(ahead, xyz_grad):
        nonlocal gout
        store.grad.zero_()
        tape = []
        geom = mod.geometry(xyz, points=pts, ahead=ahead)
        _, out, _ = mod.forward(xyz, pts, tape=tape, geom=geom)
        if gout is None:
            gout = torch.randn(out.shape, generator=g).to(dev)
        d_feat, d_xyz = mod.backward(tape[0], gout, need_feat_grad=True, need_xyz_grad=xyz_grad)
        P.wgrad_join()
        torch.cuda.synchronize()
        return tape[0], out.clone(), d_feat.clone(), (d_xyz.clone() if d_xyz is not None else None), store.grad.clone()
    assert P.HALF_GROUPS
    rec_h, out_h, df_h, _, g_h = run(True, False)          # count on the host
    rec_d, out_d, df_d, dx_d, g_d = run(False, True)       # count on the device, xyz gradient
    half_h, half_d = rec_h["recs"][0]["half"], rec_d["recs"][0]["half"]
    assert half_h.nh_limit is None and half_d.nh_limit is not None and half_d.true_count() == half_h.nh <= half_d.nh == 4 * half_d.G
    assert relerr(out_d, out_h) < 1e-6 and relerr(df_d, df_h) < 1e-5
    assert float((g_d.double() - g_h.double()).norm() / g_h.double().norm()) < 1e-5
    P.HALF_GROUPS = False
    try:
        rec_f, out_f, df_f, dx_f, g_f = run(False, True)
        assert rec_f["recs"][0].get("half") is None
    finally:
        P.HALF_GROUPS = True
    assert relerr(out_d, out_f) < 5e-5 and relerr(df_d, df_f) < 1e-3 and relerr(dx_d, dx_f) < 1e-3
    assert float((g_d.double() - g_f.double()).norm() / g_f.double().norm()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["pieces", "full"])
def test_pooled_input_gradient_is_bit_reproducible_beside_matrix_kernels(hiplib, dev, layout):
    """Guard for the packed-f32 hazard of round 5 (profiles/r05_pk_opsel_hazard.txt): v_pk_fma_f32 whose low half takes
    src1's HIGH register returns wrong low halves in lanes 48-63 while another kernel's MFMA wavefronts share the compute unit -- the
    arg-max scatter's row loop had that form (one list entry missing in 16 columns of a row, now and then).  Its rows are stored, not
    accumulated: launched again and again on the same inputs BESIDE split-operand GEMMs on a second stream, every result must be
    bit-equal to the first.  A behavioural guard only: with the hazardous operand order this test still passes most of the time (the
    window needs MFMA wavefronts on the scatter's own compute units; in the step it took three processes to open it) -- the proof that
    the form is absent is tools/check_isa_hazards.py (tests/test_abi.py), the proof that it is harmful tools/probe/src/pk_opsel_hazard.hip."""
    from votenet_amd import mlp as M
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    G, cin, cout = 4096, 128, 256
    half = None
    if layout == "pieces":
        cnt = torch.randint(1, 65, (1, G), generator=g, dtype=torch.int32).to(dev)
        half = M.half_groups(cnt)
        half.resolve()
        rows = half.rows
        argmax = (torch.rand(G, cout, generator=g).to(dev) * cnt.view(G, 1).float()).int().clamp_(0, 63)
    else:
        rows = G * 64
        argmax = torch.randint(0, 64, (G, cout), generator=g, dtype=torch.int32).to(dev)
    xz = rnd(rows, cin)
    w, b = rnd(cin, cout) * 0.1, rnd(cout) * 0.1
    wT = w.t().contiguous()
    coef, gout, zsel = rnd(5 * cout) * 0.3, rnd(G, cout), rnd(G, cout)
    sc, sh = torch.rand(cin, generator=g).to(dev) + 0.5, rnd(cin) * 0.1
    below = (sc, sh, rnd(cin) * 0.1, torch.rand(cin, generator=g).to(dev) + 0.5, True)
    # the neighbour: a split-operand (MFMA) GEMM on a second stream, launched around every call
    side = torch.cuda.Stream()
    ax, aw = rnd(65536, 256), (rnd(256, 256) * 0.1).contiguous()
    imgs = M.SplitImages([aw])
    imgs.refresh()
    ref, bad = None, 0
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(2):
                M.linear_dense(ax, aw, None, None, None, False, want_stats=False)
        da, _ = M.pool_dgrad(xz, sc, sh, True, w, b, wT, coef, True, gout, argmax, zsel, 64, below=below, half=half)
        torch.cuda.synchronize()
        live = da if half is None else da[:16 * half.nh]
        if ref is None:
            ref = live.clone()
        elif not torch.equal(live, ref):
            bad += 1
    imgs.close()
    assert bad == 0, "%d of 39 repeats differ from the first launch" % bad
