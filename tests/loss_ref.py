"""An independent torch restatement of the reference's loss graph (model.py:61-84, 141-231), used (float64, autograd) to
check the numpy oracle's values and the HIP kernel's cotangents.  Test infrastructure."""
import torch
import torch.nn.functional as Fn


def rotate_pc_along_y(pc, ang):
    """model.py:63-72.  pc (B,N,BB,3), ang (B,BB)."""
    c, s = torch.cos(ang), torch.sin(ang)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    rot = torch.stack([c, z, s, z, o, z, -s, z, c], -1).reshape(ang.shape[0], -1, 3, 3)  # B,BB,3,3
    return torch.einsum("ijkl,imjl->imjk", rot, pc)


def huber(labels, predictions):
    return Fn.huber_loss(predictions, labels, reduction="none", delta=1.0)


def votenet_loss(seeds_xyz, votes_xyz, proposals_xyz, out, gt, nh=12, ns=10, nc=10, pos_thr=0.3, neg_thr=0.6):
    bx, lwh, roty = gt["bboxes_xyz"], gt["bboxes_lwh"], gt["bboxes_roty"]
    d2c = (seeds_xyz[:, :, None] - bx[:, None]).abs()
    d2c = rotate_pc_along_y(d2c, -roty)
    inside = ((d2c < lwh[:, None] / 2.0).sum(-1) == 3)
    surface = inside.sum(-1) >= 1
    vassign = d2c.norm(dim=-1).argmin(-1)
    gt_c = torch.gather(bx, 1, vassign[..., None].expand(-1, -1, 3))
    vote = ((votes_xyz - gt_c).abs().sum(-1) * surface.to(votes_xyz.dtype)).mean()
    dist = (proposals_xyz[:, :, None] - bx[:, None]).norm(dim=-1)
    passign = dist.argmin(-1)
    mind = dist.min(-1).values
    pb, pp = torch.nonzero(mind < pos_thr, as_tuple=True)
    nb, npp = torch.nonzero(mind > neg_thr, as_tuple=True)
    pg = passign[pb, pp]
    ce = lambda lg, lab: Fn.cross_entropy(lg, lab.long(), reduction="mean")
    obj = ce(out[pb, pp, :2], torch.ones_like(pb)) + ce(out[nb, npp, :2], torch.zeros_like(nb))
    center = huber(bx[pb, pg] - proposals_xyz[pb, pp], out[pb, pp, 2:5]).sum(-1).mean()
    dual = dist.argmin(1)
    bi = torch.arange(bx.shape[0], device=bx.device)[:, None].expand_as(dual)
    center = center + huber(bx - proposals_xyz[bi, dual], out[bi, dual, 2:5]).sum(-1).mean()
    hl = gt["heading_labels"][pb, pg].long()
    hcls = ce(out[pb, pp, 5:5 + nh], hl)
    hres = huber(gt["heading_residuals"][pb, pg], out[pb, pp, 5 + nh:5 + 2 * nh].gather(1, hl[:, None])[:, 0]).mean()
    sl = gt["size_labels"][pb, pg].long()
    o = 5 + 2 * nh
    scls = ce(out[pb, pp, o:o + ns], sl)
    sres_pred = out[pb, pp, o + ns:o + 4 * ns].reshape(-1, ns, 3)[torch.arange(len(pb), device=bx.device), sl]
    sres = huber(gt["size_residuals"][pb, pg], sres_pred).sum(-1).mean()
    sem = ce(out[pb, pp, -nc:], gt["semantic_labels"][pb, pg])
    box = center + 0.1 * hcls + hres + 0.1 * scls + sres
    total = vote + 0.5 * obj + box + 0.1 * sem
    return dict(total_cost=total, vote_reg_loss=vote, obj_cls_loss=obj, center_loss=center, heading_cls_loss=hcls,
                heading_residual_loss=hres, size_cls_loss=scls, size_residual_loss=sres, sem_cls_loss=sem, box_loss=box,
                n_pos=len(pb), n_neg=len(nb))


def random_case(seed, b=2, n=256, p=64, bb=5, nh=12, ns=10, nc=10):
    """Seeds / votes / proposals scattered around a few boxes so that positive and negative proposals both occur."""
    import numpy as np
    rng = np.random.default_rng(seed)
    F = np.float32
    bx = (rng.random((b, bb, 3)) * np.array([4, 1, 4]) + np.array([-2, -1, 1])).astype(F)
    gt = dict(bboxes_xyz=bx, bboxes_lwh=(rng.random((b, bb, 3)) * 1.5 + 0.4).astype(F), bboxes_roty=(rng.random((b, bb)) * 6.28).astype(F),
              semantic_labels=rng.integers(0, nc, (b, bb)).astype(np.int32), heading_labels=rng.integers(0, nh, (b, bb)).astype(np.int32),
              heading_residuals=(rng.random((b, bb)) - 0.5).astype(F), size_labels=rng.integers(0, ns, (b, bb)).astype(np.int32),
              size_residuals=((rng.random((b, bb, 3)) - 0.5) * 0.4).astype(F))
    pick = rng.integers(0, bb, (b, n))
    seeds = (bx[np.arange(b)[:, None], pick] + rng.normal(0, 0.6, (b, n, 3))).astype(F)
    votes = (seeds + rng.normal(0, 0.3, (b, n, 3))).astype(F)
    pick = rng.integers(0, bb, (b, p))
    prop = (bx[np.arange(b)[:, None], pick] + rng.normal(0, 0.35, (b, p, 3)) * rng.choice([0.3, 3.0], (b, p, 1))).astype(F)
    out = (rng.normal(0, 1.5, (b, p, 5 + 2 * nh + 4 * ns + nc))).astype(F)
    return seeds, votes, prop, out, gt
