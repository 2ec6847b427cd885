"""CPU restatement of the reference's loss graph -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Follows model.py:62-84 (vote assignment + vote regression loss) and model.py:141-231 (proposal assignment, objectness,
centre / Chamfer centre, heading, size and semantic losses, total cost) op for op in numpy, float32 where TensorFlow
computes in float32.  TensorFlow / Tensorpack are not in the image, so this restatement is **parity unpinned** against
the reference itself; it is pinned structurally (every line cites the reference line it follows) and numerically against
an independent torch float64 restatement whose autograd also checks the HIP kernel's gradients (tests/test_gpu_loss.py).
"""
import numpy as np

F = np.float32


def _rot_y(d, ang):
    """rotate_pc_along_y(pc, ang), model.py:63-72: rotation [[c,0,s],[0,1,0],[-s,0,c]] applied to the last axis.
    d (B,N,BB,3), ang (B,BB)."""
    c, s = np.cos(ang).astype(F)[:, None, :], np.sin(ang).astype(F)[:, None, :]
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    return np.stack([c * x + s * z, y, -s * x + c * z], -1).astype(F)


def _softmax_ce(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits."""
    m = logits.max(-1, keepdims=True)
    lse = np.log(np.exp(logits - m).sum(-1)) + m[..., 0]
    return (lse - np.take_along_axis(logits, labels[..., None].astype(np.int64), -1)[..., 0]).astype(F)


def _huber(labels, predictions, delta=1.0):
    """tf.losses.huber_loss(..., reduction=NONE)."""
    e = np.abs(predictions - labels)
    q = np.minimum(e, delta)
    return (0.5 * q * q + delta * (e - q)).astype(F)


def _mean(v):
    """tf.reduce_mean: NaN on an empty tensor, as TensorFlow."""
    return F(np.nan) if v.size == 0 else F(v.astype(np.float64).mean())


def votenet_loss(seeds_xyz, votes_xyz, proposals_xyz, proposals_output, gt, nh=12, ns=10, nc=10, pos_thr=0.3, neg_thr=0.6):
    """-> dict of the reference's named losses (float32 scalars) + assignments (for tests)."""
    bx, lwh, roty = gt["bboxes_xyz"].astype(F), gt["bboxes_lwh"].astype(F), gt["bboxes_roty"].astype(F)
    B, N = seeds_xyz.shape[:2]
    # ---- model.py:61-84: vote targets
    d2c = np.abs(seeds_xyz[:, :, None, :] - bx[:, None, :, :]).astype(F)                      # :61
    d2c = _rot_y(d2c, -roty)                                                                   # :74
    inside = (d2c < (lwh[:, None] / F(2.0))).sum(-1) == 3                                      # :75-76
    surface = inside.sum(-1) >= 1                                                              # :77
    norm = np.sqrt((d2c * d2c).sum(-1, dtype=F)).astype(F)                                     # :79
    vassign = norm.argmin(-1)                                                                  # :80
    gt_c = np.take_along_axis(bx, vassign[..., None].repeat(3, -1), 1)                         # :81-83
    vote_reg = _mean(np.abs(votes_xyz - gt_c).sum(-1, dtype=F) * surface.astype(F))           # :84
    # ---- model.py:147-153: proposal assignment
    dist = np.sqrt(((proposals_xyz[:, :, None, :] - bx[:, None, :, :]) ** 2).sum(-1, dtype=F)).astype(F)
    passign = dist.argmin(-1)
    mind = dist.min(-1)
    pos, neg = mind < F(pos_thr), mind > F(neg_thr)
    pb, pp = np.nonzero(pos)
    pg = passign[pb, pp]                                                                       # :153 positive_gt_idxes
    out = proposals_output.astype(F)
    # ---- :156-161 objectness
    obj = _mean(_softmax_ce(out[pb, pp, :2], np.ones(len(pb), np.int64))) + \
        _mean(_softmax_ce(out[neg][:, :2], np.zeros(int(neg.sum()), np.int64)))
    # ---- :167-170 centre, :172-180 Chamfer (dual) centre
    delta_gt = bx[pb, pg] - proposals_xyz[pb, pp]
    center = _mean(_huber(delta_gt, out[pb, pp, 2:5]).sum(-1, dtype=F))
    dual = dist.argmin(1)                                                                      # (B,BB) nearest proposal
    bi = np.arange(B)[:, None].repeat(dual.shape[1], 1)
    center_dual = _mean(_huber(bx - proposals_xyz[bi, dual], out[bi, dual, 2:5]).sum(-1, dtype=F))
    center = F(center + center_dual)
    # ---- :183-191 heading
    hcls_gt = gt["heading_labels"][pb, pg]
    hcls = _mean(_softmax_ce(out[pb, pp, 5:5 + nh], hcls_gt))
    hres_pred = np.take_along_axis(out[pb, pp, 5 + nh:5 + 2 * nh], hcls_gt[:, None].astype(np.int64), 1)[:, 0]
    hres = _mean(_huber(gt["heading_residuals"].astype(F)[pb, pg], hres_pred))
    # ---- :194-203 size
    scls_gt = gt["size_labels"][pb, pg]
    o = 5 + 2 * nh
    scls = _mean(_softmax_ce(out[pb, pp, o:o + ns], scls_gt))
    sres_all = out[pb, pp, o + ns:o + 4 * ns].reshape(-1, ns, 3)
    sres_pred = sres_all[np.arange(len(pb)), scls_gt]
    sres = _mean(_huber(gt["size_residuals"].astype(F)[pb, pg], sres_pred).sum(-1, dtype=F))
    box = F(center + F(0.1) * hcls + hres + F(0.1) * scls + sres)                              # :205
    # ---- :208-212 semantic
    sem = _mean(_softmax_ce(out[pb, pp, -nc:], gt["semantic_labels"][pb, pg]))
    total = F(vote_reg + F(0.5) * obj + box + F(0.1) * sem)                                    # :228
    return dict(total_cost=total, vote_reg_loss=vote_reg, obj_cls_loss=F(obj), center_loss=center, heading_cls_loss=hcls,
                heading_residual_loss=hres, size_cls_loss=scls, size_residual_loss=sres, sem_cls_loss=sem, box_loss=box,
                n_pos=int(pos.sum()), n_neg=int(neg.sum()), votes_assignment=vassign, surface_ind=surface,
                bboxes_assignment=passign, positive=pos, negative=neg, dual_assignment=dual)


def decode_boxes(proposals_xyz, proposals_output, class_mean_size, nh=12, ns=10, nc=10):
    """model.py:100-129: -> bboxes (B,P,8,3) float32, scores (B,P) = max class logit."""
    o = proposals_output.astype(F)
    B, P = o.shape[:2]
    size_cls = o[..., 5 + 2 * nh:5 + 2 * nh + ns].argmax(-1)                                           # :115
    res = o[..., 5 + 2 * nh + ns:5 + 2 * nh + 4 * ns].reshape(B, P, ns, 3)
    res = np.take_along_axis(res, size_cls[..., None, None].repeat(3, -1), 2)[:, :, 0]                 # :116-118
    size = (class_mean_size.astype(F)[size_cls] * np.maximum(F(1) + res, F(1e-6))).astype(F)           # :119
    center = (proposals_xyz + o[..., 2:5]).astype(F)                                                   # :121
    hcls = o[..., 5:5 + nh].argmax(-1)                                                                 # :122
    hres = np.take_along_axis(o[..., 5 + nh:5 + 2 * nh], hcls[..., None], -1)[..., 0]                  # :123-125
    heading = np.mod(((hcls.astype(F) * F(2) + hres) * F(np.pi / nh)).astype(F), F(2 * np.pi)).astype(F)  # :126 floormod
    c, s = np.cos(heading).astype(F), np.sin(heading).astype(F)
    l, w, h = size[..., 0], size[..., 1], size[..., 2]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], F) * F(0.5)                                            # :107-110
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1], F) * F(0.5)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], F) * F(0.5)
    x0, y0, z0 = l[..., None] * sx, h[..., None] * sy, w[..., None] * sz
    xr = c[..., None] * x0 + s[..., None] * z0                                                         # :111 einsum with the y rotation
    zr = -s[..., None] * x0 + c[..., None] * z0
    boxes = np.stack([xr, y0, zr], -1).astype(F) + center[:, :, None, :]
    return boxes.astype(F), o[..., -nc:].max(-1)
