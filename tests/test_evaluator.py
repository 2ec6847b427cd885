"""The detection evaluator (votenet_amd/evaluator.py) against a literal restatement of the reference's loops
(evaluator.py:42-161) -- CPU part; the device IoU table is covered in the gpu-marked tests below."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ref_eval_det_cls(pred, gt_iou, npos, ovthresh):
    """evaluator.py:76-161 with the IoU of detection d against its image's ground truth given as a table.
    pred: list of (img, score, iou_row); literal transcription of the loop structure (sort, scan, det flags)."""
    conf = np.array([p[1] for p in pred])
    order = np.argsort(-conf, kind="stable")
    det = {}
    tp, fp = np.zeros(len(pred)), np.zeros(len(pred))
    for d, k in enumerate(order):
        img, _, row = pred[k]
        ovmax, jmax = -np.inf, -1
        for j in range(len(row)):
            if row[j] > ovmax:
                ovmax, jmax = row[j], j
        if ovmax > ovthresh:
            if not det.get((img, jmax), False):
                tp[d] = 1.0
                det[(img, jmax)] = True
            else:
                fp[d] = 1.0
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_eval_det_cls_vs_reference_loops(seed):
    from votenet_amd import evaluator as E
    rng = np.random.default_rng(seed)
    gt_count = {img: int(rng.integers(0, 4)) for img in range(6)}
    pred = []
    for _ in range(40):
        img = int(rng.integers(0, 6))
        row = (rng.random(gt_count[img]) * (rng.random() < 0.6)).astype(np.float32)
        pred.append((img, float(rng.random()), row))
    npos = sum(gt_count.values())
    exp = ref_eval_det_cls(pred, None, npos, 0.25)
    rec, prec, ap = E.eval_det_cls([p[0] for p in pred], [p[1] for p in pred], [p[2] for p in pred], gt_count, 0.25)
    assert abs(ap - exp) < 1e-12
    assert len(rec) == 40 and (np.diff(rec) >= 0).all()


def test_voc_ap_known_values():
    from votenet_amd import evaluator as E
    assert E.voc_ap(np.array([0.5, 1.0]), np.array([1.0, 1.0])) == 1.0
    assert abs(E.voc_ap(np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2.0 / 3.0])) - (0.5 * 1.0 + 0.5 * 2.0 / 3.0)) < 1e-12
    assert abs(E.voc_ap(np.array([0.0, 1.0]), np.array([0.0, 0.5]), use_07_metric=True) - 0.5) < 1e-12


def test_box_corners_match_get_3d_box():
    from votenet_amd import evaluator as E
    c = E.box_corners(np.array([1.0, 2.0, 3.0]), np.array([2.0, 1.0, 4.0]), np.array(0.0))
    assert c.shape == (8, 3) and np.allclose(c[0], [2.0, 4.0, 3.5]) and np.allclose(c[6], [0.0, 0.0, 2.5])  # dataset.py:99-101
    assert (c[:4, 1] == c[0, 1]).all() and c[0, 1] > c[4, 1]  # evaluator.py:32


@pytest.mark.gpu
def test_iou3d_cross_vs_matrix_and_oracle(hiplib, dev, O):
    import torch
    from votenet_amd import evaluator as E
    from votenet_amd import tf_nms3d
    rng = np.random.default_rng(3)
    mk = lambda n: E.box_corners(rng.random((2, n, 3)) * 2, rng.random((2, n, 3)) + 0.3, rng.random((2, n)) * 6.28)
    a, b = mk(17), mk(9)
    cross = tf_nms3d.iou3d_cross(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    both = np.concatenate([a, b], 1)
    full = tf_nms3d.iou3d_matrix(torch.from_numpy(both).to(dev)).cpu().numpy()
    assert np.array_equal(cross, full[:, :17, 17:])
    for s in range(2):
        assert np.abs(cross[s] - O.iou3d_matrix(both[s])[:17, 17:]).max() < 1e-5


@pytest.mark.gpu
def test_eval_det_perfect_and_shifted_detections(hiplib, dev):
    """Detections on top of the ground truth (2 cm off: identical boxes are the degenerate all-edges-collinear case of the
    reference's polygon clipping) score AP 1 in every class; moving one class's detections away drops only that class."""
    import torch
    from votenet_amd import evaluator as E
    from votenet_amd import synth
    gt_np = synth.room_gt(4, 2048, 40)
    gt = E.gt_for_eval(gt_np)
    B, G = gt["labels"].shape
    cls = np.zeros((B, G, 10), np.float32)
    cls[np.arange(B)[:, None], np.arange(G)[None], gt["labels"]] = 5.0
    keep = np.array([[b, i] for b in range(B) for i in range(gt["count"][b])], np.int32)
    near = (gt["boxes"] + np.array([0.02, 0.01, -0.02], np.float32)).astype(np.float32)
    pred = dict(bboxes=torch.from_numpy(near).to(dev), nms_idx=torch.from_numpy(keep).to(dev),
                class_scores=torch.from_numpy(cls).to(dev))
    ap, m = E.eval_det(pred, gt)
    assert m == 1.0 and all(v == 1.0 for v in ap.values())
    c0 = int(gt["labels"][0, 0])
    moved = near.copy()
    moved[gt["labels"] == c0] += 10.0
    pred["bboxes"] = torch.from_numpy(moved).to(dev)
    ap2, m2 = E.eval_det(pred, gt)
    assert ap2[c0] == 0.0 and all(v == 1.0 for k, v in ap2.items() if k != c0) and m2 < 1.0
