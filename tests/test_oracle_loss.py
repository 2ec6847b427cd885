"""CPU: the numpy oracle of the loss graph (oracle/oracle_loss.py) against the independent torch float64 restatement."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loss_ref  # noqa: E402
from oracle import oracle_loss  # noqa: E402

NAMES = ["total_cost", "vote_reg_loss", "obj_cls_loss", "center_loss", "heading_cls_loss", "heading_residual_loss", "size_cls_loss",
         "size_residual_loss", "sem_cls_loss", "box_loss"]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_oracle_loss_vs_torch(seed):
    seeds, votes, prop, out, gt = loss_ref.random_case(seed)
    o = oracle_loss.votenet_loss(seeds, votes, prop, out, gt)
    T = lambda a: torch.from_numpy(a).double() if a.dtype == np.float32 else torch.from_numpy(a)
    t = loss_ref.votenet_loss(T(seeds), T(votes), T(prop), T(out), {k: T(v) for k, v in gt.items()})
    assert o["n_pos"] == t["n_pos"] > 0 and o["n_neg"] == t["n_neg"] > 0
    for k in NAMES:
        assert abs(float(o[k]) - float(t[k])) <= 2e-5 * max(1.0, abs(float(t[k]))), k


def test_oracle_loss_empty_sets_are_nan():
    seeds, votes, prop, out, gt = loss_ref.random_case(5)
    far = prop + 100.0  # no proposal within 0.3 of a centre: reduce_mean over an empty tensor
    o = oracle_loss.votenet_loss(seeds, votes, far.astype(np.float32), out, gt)
    assert o["n_pos"] == 0 and np.isnan(o["total_cost"]) and np.isfinite(o["vote_reg_loss"])


def test_room_gt_layout():
    from votenet_amd import synth
    g = synth.room_gt(3, 2048, 1000)
    bb = g["bboxes_xyz"].shape[1]
    assert g["bboxes_lwh"].shape == (3, bb, 3) and g["size_residuals"].shape == (3, bb, 3) and g["heading_labels"].dtype == np.int32
    assert (g["heading_labels"] >= 0).all() and (g["heading_labels"] < 12).all() and (np.abs(g["heading_residuals"]) <= 1.0 + 1e-6).all()
    # size = mean * (1 + residual), angle = class * 2pi/NH + residual * pi/NH  (dataset.py:52-90, 297-299)
    size = synth.MEAN_SIZES[g["size_labels"]] * (1 + g["size_residuals"])
    assert np.allclose(size, g["bboxes_lwh"], atol=1e-5)
    ang = g["heading_labels"] * (2 * np.pi / 12) + g["heading_residuals"] * (np.pi / 12)
    assert np.allclose(np.cos(ang), np.cos(g["bboxes_roty"]), atol=1e-5) and np.allclose(np.sin(ang), np.sin(g["bboxes_roty"]), atol=1e-5)
