"""GPU: the fused loss kernel (votenet_loss, csrc/loss.hip) against the numpy oracle of model.py:61-84,141-231 (values) and
against torch float64 autograd of the independent restatement (cotangents)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import loss_ref  # noqa: E402

pytestmark = pytest.mark.gpu
NAMES = ["total_cost", "vote_reg_loss", "obj_cls_loss", "center_loss", "heading_cls_loss", "heading_residual_loss", "size_cls_loss",
         "size_residual_loss", "sem_cls_loss", "box_loss"]


def run_device(dev, seeds, votes, prop, out, gt):
    from votenet_amd import loss as VL
    o = dict(seeds_xyz=torch.from_numpy(seeds).to(dev), votes_xyz=torch.from_numpy(votes).to(dev),
             proposals_xyz=torch.from_numpy(prop).to(dev), proposals_output=torch.from_numpy(out).to(dev))
    return VL.votenet_loss(o, VL.gt_to_device(gt, dev))


@pytest.mark.parametrize("seed,shape", [(0, {}), (1, {}), (2, dict(b=8, n=1024, p=256, bb=9)), (3, dict(b=1, n=100, p=33, bb=1)),
                                        (4, dict(b=3, n=64, p=16, bb=16))])
def test_loss_values_and_cotangents(hiplib, dev, seed, shape):
    from oracle import oracle_loss
    seeds, votes, prop, out, gt = loss_ref.random_case(seed, **shape)
    losses, cot = run_device(dev, seeds, votes, prop, out, gt)
    got = dict(zip(NAMES + ["n_pos", "n_neg"], losses.cpu().tolist()))
    o = oracle_loss.votenet_loss(seeds, votes, prop, out, gt)
    assert int(got["n_pos"]) == o["n_pos"] > 0 and int(got["n_neg"]) == o["n_neg"] > 0
    for k in NAMES:
        assert abs(got[k] - float(o[k])) <= 1e-5 * max(1.0, abs(float(o[k]))), (k, got[k], float(o[k]))
    # cotangents: autograd of the float64 restatement
    T = lambda a: torch.from_numpy(a).double() if a.dtype == np.float32 else torch.from_numpy(a)
    v, p, w = T(votes).requires_grad_(True), T(prop).requires_grad_(True), T(out).requires_grad_(True)
    t = loss_ref.votenet_loss(T(seeds), v, p, w, {k: T(x) for k, x in gt.items()})
    t["total_cost"].backward()
    for name, ref in (("votes_xyz", v.grad), ("proposals_xyz", p.grad), ("proposals_output", w.grad)):
        g = cot[name].double().cpu()
        assert float((g - ref).abs().max()) <= 1e-5 * max(1e-3, float(ref.abs().max())), name
    # reproducible bit for bit (fixed-order reductions)
    losses2, _ = run_device(dev, seeds, votes, prop, out, gt)
    assert torch.equal(losses, losses2)


def test_loss_reads_a_column_slice_in_place(hiplib, dev):
    """votenet_loss_pitched: proposals_output as the first 79 columns of a wider row-major tensor (what the proposal module's last GEMM
    leaves) gives the losses and cotangents of the contiguous copy, bit for bit; the cotangent stays dense."""
    from votenet_amd import loss as VL
    seeds, votes, prop, out, gt = loss_ref.random_case(5, b=4, n=256, p=64, bb=7)
    ref_l, ref_c = run_device(dev, seeds, votes, prop, out, gt)
    wide = torch.full((out.shape[0], out.shape[1], 128), 1e30, device=dev)
    wide[:, :, :out.shape[2]] = torch.from_numpy(out).to(dev)
    o = dict(seeds_xyz=torch.from_numpy(seeds).to(dev), votes_xyz=torch.from_numpy(votes).to(dev),
             proposals_xyz=torch.from_numpy(prop).to(dev), proposals_output=wide[:, :, :out.shape[2]])
    assert not o["proposals_output"].is_contiguous()
    got_l, got_c = VL.votenet_loss(o, VL.gt_to_device(gt, dev))
    assert torch.equal(got_l, ref_l)
    for k in ref_c:
        assert torch.equal(got_c[k], ref_c[k]) and got_c[k].is_contiguous(), k


def test_loss_without_positives_is_nan_like_tensorflow(hiplib, dev):
    seeds, votes, prop, out, gt = loss_ref.random_case(7)
    losses, cot = run_device(dev, seeds, votes, (prop + 100.0).astype(np.float32), out, gt)
    l = losses.cpu().numpy()
    assert l[10] == 0 and np.isnan(l[0]) and np.isfinite(l[1])  # vote loss unaffected
    assert torch.isfinite(cot["proposals_output"]).all() and torch.isfinite(cot["votes_xyz"]).all()


def test_train_step_with_the_loss_graph(hiplib, dev):
    """config 3 in miniature: forward -> loss kernel -> backward -> Adam with the ground truth of the synthetic scenes; the
    total cost goes down over a few steps on one fixed batch."""
    from votenet_amd import loss as VL
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = torch.from_numpy(synth.room_batch(4, 8192, 300)).to(dev)
    gt = VL.gt_to_device(synth.room_gt(4, 8192, 300), dev)
    net = VM.VoteNetHotPath(dev, seed=1)
    net.init_optimizer(1e-3)
    costs, npos = [], []
    for _ in range(16):
        net.train_step(x, gt=gt)
        l = net.last_losses.cpu().numpy()
        costs.append(float(l[0]))
        npos.append(int(l[10]))
    costs = np.array(costs)
    # a step without any positive proposal has a NaN cost (tf.reduce_mean of an empty tensor) and still finite parameters
    assert np.isfinite(costs[np.array(npos) > 0]).all() and np.isnan(costs[np.array(npos) == 0]).all()
    fin = costs[np.isfinite(costs)]
    assert len(fin) >= 8 and fin[-3:].mean() < fin[:3].mean()
    assert torch.isfinite(net.store.flat).all()


def test_decode_boxes_vs_oracle(hiplib, dev):
    """votenet_decode_boxes against the numpy restatement of model.py:100-129, and its corner order against what NMS3D
    expects (first four corners = top face, [0] / [4] span the height)."""
    from oracle import oracle_loss
    from votenet_amd import loss as VL
    from votenet_amd import synth
    rng = np.random.default_rng(11)
    prop = (rng.random((3, 50, 3)) * 4).astype(np.float32)
    out = rng.normal(0, 1.0, (3, 50, 79)).astype(np.float32)
    out[0, 0, 5 + 24 + 10:5 + 24 + 40] = -5.0  # residual below -1: the 1e-6 floor of model.py:119
    boxes, scores = VL.decode_boxes(torch.from_numpy(prop).to(dev), torch.from_numpy(out).to(dev))
    eb, es = oracle_loss.decode_boxes(prop, out, synth.MEAN_SIZES.astype(np.float32))
    assert np.abs(boxes.cpu().numpy() - eb).max() <= 1e-5 * max(1.0, np.abs(eb).max())
    assert (scores.cpu().numpy() == es).all()
    b = boxes.cpu().numpy()
    assert np.allclose(b[:, :, :4, 1], b[:, :, :1, 1], atol=1e-6) and (b[:, :, 0, 1] >= b[:, :, 4, 1]).all()
