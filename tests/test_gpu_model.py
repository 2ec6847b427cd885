"""GPU: the layer stack (votenet_amd.model / pointnet2) against the CPU oracle composed the same way."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def N(t):
    return t.detach().cpu().numpy()


def oracle_chain(O, x, layers, k=None):
    for L in layers:
        z = O.linear(x, N(L.p("W")), N(L.p("b")))
        if L.bn:
            mean, var = O.bn_stats(z)
            x = O.bn_relu(z, mean, var, N(L.p("gamma")), N(L.p("beta")), relu=L.relu)
        else:
            x = z
    return O.max_over_k(x, k) if k else x


def oracle_sa(O, mod, xyz, pts, sample_xyz=None):
    b = xyz.shape[0]
    fidx = O.farthest_point_sample(mod.npoint, sample_xyz if sample_xyz is not None else xyz)
    new_xyz = O.gather_point(xyz, fidx)
    idx, _ = O.query_ball_point(mod.radius, mod.nsample, xyz, new_xyz)
    g = O.group_concat(xyz, new_xyz, pts, idx).reshape(-1, 3 + pts.shape[2])
    out = oracle_chain(O, g, mod.mlp, mod.nsample)
    if mod.mlp2:
        out = oracle_chain(O, out, mod.mlp2)
    return new_xyz, out.reshape(b, mod.npoint, -1)


def oracle_fp(O, mod, x1, x2, p1, p2):
    dist, idx = O.three_nn(x1, x2)
    itp = O.three_interpolate(p2, idx, O.three_nn_weights(dist))
    x = np.concatenate([itp, p1], 2).reshape(-1, itp.shape[2] + p1.shape[2])
    return oracle_chain(O, x, mod.mlp).reshape(x1.shape[0], x1.shape[1], -1)


def test_forward_small_vs_oracle(hiplib, dev, O):
    """The full layer stack on a reduced cloud (2 x 4096 pts, 512/256/128/64 samples), layer by layer."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = synth.room_batch(2, 4096, 77)
    net = VM.VoteNetHotPath(dev, seed=3, npoints=(512, 256, 128, 64))
    # perturb BN affine parameters so they are exercised
    g = torch.Generator().manual_seed(1)
    for name, v in net.store.views.items():
        if name.endswith("gamma"):
            v.copy_((1 + 0.2 * torch.randn(v.shape, generator=g)).to(dev))
        if name.endswith("beta") or name.endswith("/b"):
            v.copy_((0.1 * torch.randn(v.shape, generator=g)).to(dev))
    # the proposal layer samples 256 of the seeds; with 256 seeds here use all of them
    out = net.forward(torch.from_numpy(x).to(dev))

    l1x, l1p = oracle_sa(O, net.sa1, x, x)
    l2x, l2p = oracle_sa(O, net.sa2, l1x, l1p)
    l3x, l3p = oracle_sa(O, net.sa3, l2x, l2p)
    l4x, l4p = oracle_sa(O, net.sa4, l3x, l3p)
    l3p2 = oracle_fp(O, net.fp1, l3x, l4x, l3p, l4p)
    seeds = oracle_fp(O, net.fp2, l2x, l3x, l2p, l3p2)
    assert (N(out["seeds_xyz"]) == l2x).all()

    def relerr(a, b):
        return np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert relerr(N(out["seeds_points"]), seeds) < 1e-4
    xx = np.concatenate([l2x, seeds], 2).reshape(-1, 259)
    votes = (xx + oracle_chain(O, xx, net.voting)).reshape(2, -1, 259)
    assert relerr(N(out["votes_xyz"]), votes[..., :3]) < 1e-4
    assert relerr(N(out["votes_points"]), votes[..., 3:]) < 1e-4
    # proposal layer on the DEVICE votes (the neighbour lists depend on vote xyz to the last bit)
    vx, vp = N(out["votes_xyz"]), N(out["votes_points"])
    px, pout = oracle_sa(O, net.proposal, vx, vp, sample_xyz=l2x)
    assert (N(out["proposals_xyz"]) == px).all()  # utils.py:42-43: FPS on seeds, centres from votes
    assert relerr(N(out["proposals_output"]), pout) < 1e-4
    assert out["proposals_output"].shape == (2, 256, 79)


def test_predict_tail_nms_vs_oracle(hiplib, dev, O):
    """Predict tower (model.py:98-139): decoded boxes -> device NMS equals the oracle NMS on the same boxes."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = torch.from_numpy(synth.room_batch(2, 4096, 5)).to(dev)
    net = VM.VoteNetHotPath(dev, seed=7, npoints=(512, 256, 128, 64))
    r = net.predict(x, 0.25)
    boxes, score = N(r["bboxes"]), N(r["scores"])
    obj = N(r["proposals_output"][..., :2])
    assert boxes.shape == (2, 256, 8, 3)
    exp = O.nms3d(boxes, score, obj, 0.25)
    iou = np.stack([O.iou3d_matrix(boxes[s]) for s in range(2)])
    got = N(r["nms_idx"])
    if not (np.abs(iou - 0.25) < 1e-5).any() and len(np.unique(score)) == score.size:
        assert (got == exp).all()
    assert got.shape[1] == 2 and (obj[got[:, 0], got[:, 1], 1] > obj[got[:, 0], got[:, 1], 0]).all()
