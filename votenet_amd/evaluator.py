"""Detection evaluation (evaluator.py:26-200 of the reference): VOC average precision per class at a 3D-IoU threshold.
The box overlaps -- the reference's shapely polygon loop over every (detection, ground-truth) pair -- come from the device
(votenet_iou3d_cross, the IoU kernel of the NMS); matching and the precision / recall curve are the reference's host logic
restated with numpy (they are O(detections))."""
import numpy as np
import torch

from . import tf_nms3d
from .synth import MEAN_SIZES, NC


def box_corners(center, lwh, roty):
    """get_3d_box (dataset.py:92-109): (…,3), (…,3) l,w,h, (…) heading -> (…,8,3) corners, first four = top face."""
    c, s = np.cos(roty), np.sin(roty)
    l, w, h = lwh[..., 0], lwh[..., 1], lwh[..., 2]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) * 0.5
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1]) * 0.5
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) * 0.5
    x0, y0, z0 = l[..., None] * sx, h[..., None] * sy, w[..., None] * sz
    x = c[..., None] * x0 + s[..., None] * z0
    z = -s[..., None] * x0 + c[..., None] * z0
    return (np.stack([x, y0, z], -1) + center[..., None, :]).astype(np.float32)


def voc_ap(rec, prec, use_07_metric=False):
    """evaluator.py:42-73."""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def eval_det_cls(det_img, det_score, det_iou, gt_count, ovthresh=0.25, use_07_metric=False):
    """evaluator.py:76-161 for one class.  det_img[d]: image of detection d; det_score[d]; det_iou[d]: its IoU with every
    ground-truth box of this class in its image (1-D array, may be empty); gt_count {img: #gt boxes}.  -> rec, prec, ap."""
    npos = int(sum(gt_count.values()))
    order = np.argsort(-np.asarray(det_score, dtype=np.float64), kind="stable")
    taken = {img: np.zeros(c, bool) for img, c in gt_count.items()}
    tp, fp = np.zeros(len(order)), np.zeros(len(order))
    for r, d in enumerate(order):
        ov = det_iou[d]
        if len(ov) and ov.max() > ovthresh:
            j = int(ov.argmax())  # first maximum, as the reference's strict '>' scan
            if not taken[det_img[d]][j]:
                tp[r] = 1.0
                taken[det_img[d]][j] = True
            else:
                fp[r] = 1.0
        else:
            fp[r] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos) if npos else np.zeros_like(tp)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det(pred, gt, ovthresh=0.25, use_07_metric=False):
    """evaluator.py:164-200 on batched device tensors.
    pred: dict(bboxes (B,N,8,3) device, nms_idx (K,2) [scene, box] device, class_scores (B,N,NC) device) -- the predict
          tower's outputs; a kept box is ONE detection of its arg-max class with the max class score (evaluator.py:225-233).
    gt:   dict(boxes (B,G,8,3) numpy corners, labels (B,G) int, count (B,) valid boxes per scene).
    -> {class: ap}, mAP over the classes that occur in gt."""
    dev = pred["bboxes"].device
    gt_boxes = torch.from_numpy(np.ascontiguousarray(gt["boxes"], dtype=np.float32)).to(dev)
    iou = tf_nms3d.iou3d_cross(pred["bboxes"], gt_boxes).cpu().numpy()  # (B,N,G): every overlap the evaluation can ask for
    keep = pred["nms_idx"].cpu().numpy()
    cls_scores = pred["class_scores"].detach().cpu().numpy()
    labels, count = np.asarray(gt["labels"]), np.asarray(gt["count"])
    ap = {}
    for c in range(NC):
        gt_count, gt_cols = {}, {}
        for b in range(labels.shape[0]):
            cols = np.nonzero(labels[b, :count[b]] == c)[0]
            if len(cols):
                gt_count[b], gt_cols[b] = len(cols), cols
        if not gt_count:
            continue
        d_img, d_score, d_iou = [], [], []
        for b, i in keep:
            if int(cls_scores[b, i].argmax()) != c:
                continue
            d_img.append(int(b))
            d_score.append(float(cls_scores[b, i].max()))
            d_iou.append(iou[b, i, gt_cols[b]] if b in gt_cols else np.zeros(0, np.float32))
        for b in set(d_img):
            gt_count.setdefault(b, 0)
        ap[c] = eval_det_cls(d_img, d_score, d_iou, gt_count, ovthresh, use_07_metric)[2]
    return ap, (float(np.mean(list(ap.values()))) if ap else float("nan"))


def gt_for_eval(gt_np, counts=None):
    """synth.room_gt dict -> corners / labels / per-scene count (padding rows repeat the last box: counted once)."""
    b, g = gt_np["bboxes_roty"].shape
    if counts is None:
        counts = []
        for s in range(b):
            n = g
            while n > 1 and np.array_equal(gt_np["bboxes_xyz"][s, n - 1], gt_np["bboxes_xyz"][s, n - 2]):
                n -= 1
            counts.append(n)
    return dict(boxes=box_corners(gt_np["bboxes_xyz"], gt_np["bboxes_lwh"], gt_np["bboxes_roty"]), labels=gt_np["semantic_labels"],
                count=np.asarray(counts))
