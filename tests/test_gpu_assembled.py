"""GPU: the first SA layer assembled inside its consumers (csrc/assemble.hip, SRC 4 / EPI 6 of mlp_fast.hip, MODE 3 of
mlp_wgrad_fast.hip, votenet_group_linear_backward_assembled) against the same layer materialised (votenet_assemble_z0 writes z0 with
the kernels' own arithmetic; votenet_group_linear is the stored form the reference's conv over the grouped tensor corresponds to)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / max(1e-12, float(b.double().abs().max())))


@pytest.mark.parametrize("b,n,m,k,cf,c0,c1,radius", [(2, 600, 64, 64, 128, 128, 128, 0.5), (1, 400, 32, 64, 16, 64, 64, 0.3),
                                                     (2, 300, 16, 32, 32, 128, 256, 0.25), (1, 500, 8, 16, 8, 64, 320, 2.0)])
def test_assembled_first_layer_matches_the_materialised_layer(hiplib, dev, gemm_form, b, n, m, k, cf, c0, c1, radius):
    from votenet_amd import mlp as M
    from votenet_amd import tf_grouping, tf_sampling
    g = torch.Generator().manual_seed(3 * n + cf)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    pos = lambda s: torch.rand(s, generator=g).to(dev) + 0.5
    xyz = (torch.rand(b, n, 3, generator=g) * 2.0).to(dev)
    feat = rnd(b, n, cf)
    new_xyz = tf_sampling.gather_point(xyz, tf_sampling.farthest_point_sample(m, xyz))
    idx, cnt = tf_grouping.query_ball_point(radius, k, xyz, new_xyz)
    rows = b * m * k
    assert M.assembled_supported(rows, c0, c1)
    w0, b0, w1 = rnd(3 + cf, c0) * 0.3, rnd(c0) * 0.1, rnd(c0, c1) * 0.2
    wx = w0[:3].contiguous()
    wT = w1.t().contiguous()
    # images for the second layer's forward GEMM (fp16 x 2 when gemm_form == 2, as the model registers its forward matrices) and its
    # input-gradient GEMM (the transposed copy: always bf16 x 3); gemm_form == 0 ignores them
    img_f = M.SplitImages([w1], pieces=2 if gemm_form == 2 else 3)
    img_f.refresh()
    img = M.SplitImages([wT])
    img.refresh()
    P, _ = M.linear_dense(feat.reshape(b * n, cf), w0[3:].contiguous(), b0, want_stats=False)
    # geometry records and the per-point sums (padding slots repeat slot 0: with and without pts_cnt the sums agree)
    geo, cntv, mom = M.assemble_rows(xyz, new_xyz, idx, pts_cnt=cnt)
    geo2, cntv2, _ = M.assemble_rows(xyz, new_xyz, idx)
    bi = torch.arange(b, device=dev)[:, None, None]
    dx = (xyz[bi, idx.long()] - new_xyz[:, :, None, :]).reshape(rows, 3)
    prow = (idx.long() + torch.arange(b, device=dev)[:, None, None] * n).reshape(rows)
    assert torch.equal(geo[:, :3], dx) and torch.equal(geo[:, 3].view(torch.int32).long(), prow) and torch.equal(geo2, geo)
    cref = torch.zeros(b * n, 4, dtype=torch.float64, device=dev)
    cref.index_add_(0, prow, torch.cat([torch.ones(rows, 1, device=dev), dx], 1).double())
    fx = lambda t: torch.cat([t[:, :1].double(), t[:, 1:].double() / 2.0 ** 32], 1)  # count, fixed-point sums -> metres
    assert torch.equal(cntv[:, 0].double(), cref[:, 0]) and relerr(fx(cntv), cref) < 1e-6 and relerr(fx(cntv2), cref) < 1e-6
    assert torch.equal(M.assemble_rows(xyz, new_xyz, idx, pts_cnt=cnt)[1], cntv)  # integer atomics: order independent
    assert relerr(mom[:3], dx.double().sum(0)) < 1e-12
    # z0 and its BatchNorm statistics
    z0 = M.assemble_z0(geo, P, wx)
    z0ref = P[prow].double() + dx.double() @ wx.double()
    assert relerr(z0, z0ref) < 1e-6
    zg, _ = M.group_linear(xyz, new_xyz, idx, (P - b0).contiguous().view(b, n, c0), wx, b0)  # the stored form (bias added last)
    assert relerr(zg, z0ref) < 1e-6
    st = M.assemble_stats(P, cntv, wx, mom)
    assert relerr(st[:c0], z0ref.sum(0)) < 1e-6 and relerr(st[c0:], (z0ref * z0ref).sum(0)) < 1e-6
    gamma0, beta0 = rnd(c0) * 0.2 + 1.0, rnd(c0) * 0.1
    bn0 = M.PendingBN(st, gamma0, beta0, rows)
    # second layer forward: the assembled loader against the ordinary GEMM on the materialised z0 with the same BatchNorm
    z1, st1 = M.assembled_linear(geo, P, wx, w1, None, bn0)
    z1m, st1m = M.linear_dense(z0, w1, None, bn0.scale, bn0.shift, True)
    assert relerr(z1, z1m) < 1e-6 and relerr(st1, st1m) < 1e-6
    a0 = torch.relu(z0.double() * bn0.scale.double() + bn0.shift.double())
    assert relerr(z1, a0 @ w1.double()) < 2e-5
    # backward of the second layer
    da1, coef1 = rnd(rows, c1), rnd(5 * c1)
    dw, dwm = torch.zeros(c0, c1, device=dev), torch.zeros(c0, c1, device=dev)
    M.assembled_wgrad_bn(geo, P, wx, bn0.scale, bn0.shift, True, z1, coef1, True, da1, dw)
    M.wgrad_dense_bn(z0, z1, coef1, True, dwm, da=da1, in_scale=bn0.scale, in_shift=bn0.shift, in_relu=True)
    assert relerr(dw, dwm) < 1e-5
    below = (bn0.scale, bn0.shift, bn0.mean, bn0.var, True)
    da0, sums = M.assembled_dgrad_bn_reduce(z1, coef1, True, wT, da1, geo, P, wx, below)
    split_k, M.SPLIT_K = M.SPLIT_K, False  # bit for bit against the UNSPLIT stored-layer kernel (split-K adds the same products in another order)
    try:
        da0m, sumsm = M.dgrad_bn(z1, coef1, True, wT, da=da1, below=(z0,) + below)
    finally:
        M.SPLIT_K = split_k
    assert torch.equal(da0, da0m)
    scale = torch.cat([da0.double().abs().sum(0), (da0.double() * ((z0.double() - bn0.mean.double()) / torch.sqrt(bn0.var.double() + M.BN_EPS))).abs().sum(0)])
    assert float(((sums - sumsm).abs() / (scale + 1e-30)).max()) < 1e-6
    # backward of the first layer (GroupPointGrad at the layer's output width + the xyz rows of dW)
    coef0 = rnd(5 * c0)
    dwx, dwxm = torch.zeros(3, c0, device=dev), torch.zeros(3, c0, device=dev)
    S, _ = M.group_linear_backward_assembled(xyz, new_xyz, idx, cnt, P, wx, da0, coef0, True, dwx)
    Sm, _ = M.group_linear_backward(xyz, new_xyz, idx, cnt, z0, da0, coef0, True, dwxm)
    assert relerr(S, Sm) < 1e-5 and relerr(dwx, dwxm) < 1e-4
    img.close()
    img_f.close()


def test_model_with_and_without_the_assembled_first_layers(hiplib, dev):
    """The whole hot path with the first layer of sa2 / sa3 / sa4 / the proposal module assembled inside its consumers against the same
    network with those layers stored (votenet_group_linear): same outputs and losses to fp32 rounding, same gradient in the L2 sense
    (two fp32 evaluations of the forward pass move ReLU / arg-max decisions: see tests/test_gpu_narrow.py), less memory."""
    from votenet_amd import loss as VL
    from votenet_amd import model as VM
    from votenet_amd import pointnet2 as P
    from votenet_amd import synth
    b, n = 2, 4096
    x = torch.from_numpy(synth.room_batch(b, n, 9)).to(dev)
    gt = VL.gt_to_device(synth.room_gt(b, n, 9), dev)
    net = VM.VoteNetHotPath(dev, seed=6, npoints=(512, 256, 128, 64))
    fixed = {}

    def once():
        net.store.grad.zero_()
        net.store.refresh_transposes()
        tape = []
        out = net.forward(x, tape)
        losses, cot = VL.votenet_loss(out, gt)
        cot = fixed.setdefault("cot", cot)  # one set of cotangents: the loss graph's discrete decisions are not under test
        net.backward(tape, cot)
        torch.cuda.synchronize()
        return tape, out["proposals_output"].clone(), losses.clone(), net.store.grad.clone()
    assert P.ASSEMBLE_FIRST
    tape, o1, l1, g1 = once()
    kinds = [t["recs"][0]["kind"] for t in tape if t.get("op") == "sa"]
    assert kinds == ["narrow", "assembled", "assembled", "assembled", "assembled"], kinds
    assert all(t["recs"][0]["z"] is None for t in tape if t.get("op") == "sa")
    P.ASSEMBLE_FIRST = False
    try:
        tape, o0, l0, g0 = once()
        assert [t["recs"][0]["kind"] for t in tape if t.get("op") == "sa"] == ["narrow", "gather", "gather", "gather", "gather"]
    finally:
        P.ASSEMBLE_FIRST = True
    assert relerr(o1, o0) < 5e-5 and relerr(l1, l0) < 5e-5
    assert float((g1.double() - g0.double()).norm() / g0.double().norm()) < 1e-2
