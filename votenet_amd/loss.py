"""The reference's loss graph on the device: one kernel (votenet_loss, csrc/loss.hip) for the total cost of model.py:228
and its cotangents with respect to the three tensors through which it reaches the hot path."""
import torch

from . import _lib as L
from .synth import NC, NH, NS

POSITIVE_THRES, NEGATIVE_THRES = 0.3, 0.6  # config.py
NAMES = ("total_cost", "vote_reg_loss", "obj_cls_loss", "center_loss", "heading_cls_loss", "heading_residual_loss",
         "size_cls_loss", "size_residual_loss", "sem_cls_loss", "box_loss", "n_pos", "n_neg")
GT_KEYS = ("bboxes_xyz", "bboxes_lwh", "bboxes_roty", "semantic_labels", "heading_labels", "heading_residuals", "size_labels",
           "size_residuals")


def gt_to_device(gt, device):
    """numpy ground-truth dict (synth.room_gt) -> contiguous device tensors."""
    return {k: torch.from_numpy(gt[k]).contiguous().to(device) for k in GT_KEYS}


def loss_buffers(out, losses=None):
    """The output buffers of votenet_loss for the tensors of `out`, allocated once by a caller that wants them at fixed addresses
    (model.StretchGraph: the segments captured behind the loss read the cotangents there): -> (losses (12,), flat).  flat holds the three
    cotangents and the kernel's workspace and must be ZERO when votenet_loss(..., buffers=) runs (here: a carve-out of the pass's arena)."""
    from . import mlp as M
    votes, pxyz, pout = out["votes_xyz"], out["proposals_xyz"], out["proposals_output"]
    nws = int(L.lib().votenet_loss_workspace_floats(votes.shape[0]))
    if losses is None:
        losses = torch.empty(12, dtype=torch.float32, device=votes.device)
    return losses, M._zeros_f32((votes.numel() + pxyz.numel() + pout.numel() + nws,), votes.device)


def votenet_loss(out, gt, nh=NH, ns=NS, nc=NC, buffers=None):
    """out: the dict VoteNetHotPath.forward returns; gt: device tensors (gt_to_device).
    -> losses (12,) f32 on the device (NAMES), cotangents dict(votes_xyz, proposals_xyz, proposals_output).
    buffers: (losses, flat) from loss_buffers(out), flat zeroed by the caller -- the results land there instead of in fresh tensors."""
    seeds = L.dev_f32(out["seeds_xyz"], "loss seeds_xyz", 3, 3)
    votes = L.dev_f32(out["votes_xyz"], "loss votes_xyz", 3, 3)
    pxyz = L.dev_f32(out["proposals_xyz"], "loss proposals_xyz", 3, 3)
    pout = out["proposals_output"]
    if not (isinstance(pout, torch.Tensor) and pout.is_cuda and pout.dtype == torch.float32 and pout.dim() == 3 and pout.stride(2) == 1
            and pout.stride(1) >= pout.shape[2] and pout.stride(0) == pout.shape[1] * pout.stride(1)):
        pout = L.dev_f32(pout, "loss proposals_output", 3)  # anything but rows with unit column stride: a contiguous copy
    b, n = seeds.shape[:2]
    p = pxyz.shape[1]
    bb = gt["bboxes_xyz"].shape[1]
    if pout.shape[2] != 5 + 2 * nh + 4 * ns + nc:
        raise L.InvalidArgumentError("loss: proposals_output has %d channels, expected %d" % (pout.shape[2], 5 + 2 * nh + 4 * ns + nc))
    dev = seeds.device
    # the three cotangents and the kernel's workspace are ONE buffer cleared by ONE fill
    nv, npx, npo, nws = votes.numel(), pxyz.numel(), pout.numel(), int(L.lib().votenet_loss_workspace_floats(b))
    from . import mlp as M
    if buffers is not None:
        losses, flat = buffers
        if flat.numel() != nv + npx + npo + nws or losses.numel() != 12:
            raise L.InvalidArgumentError("loss: the buffers were made for other shapes (loss_buffers(out))")
    else:
        losses = torch.empty(12, dtype=torch.float32, device=dev)
        flat = M._zeros_f32((nv + npx + npo + nws,), dev)  # inside a train step: a carve-out of the pass's one zero fill
    d_votes, d_pxyz = flat[:nv].view_as(votes), flat[nv:nv + npx].view_as(pxyz)
    d_pout, ws = flat[nv + npx:nv + npx + npo].view_as(pout), flat[nv + npx + npo:]
    with L.device_guard(dev):
        L.check(L.lib().votenet_loss_pitched(b, n, p, bb, nh, ns, nc, L.ptr(seeds), L.ptr(votes), L.ptr(pxyz), L.ptr(pout), pout.stride(1),
                                     L.ptr(gt["bboxes_xyz"]), L.ptr(gt["bboxes_lwh"]), L.ptr(gt["bboxes_roty"]),
                                     L.ptr(gt["semantic_labels"]), L.ptr(gt["heading_labels"]), L.ptr(gt["heading_residuals"]),
                                     L.ptr(gt["size_labels"]), L.ptr(gt["size_residuals"]), POSITIVE_THRES, NEGATIVE_THRES,
                                     L.ptr(losses), L.ptr(d_votes), L.ptr(d_pxyz), L.ptr(d_pout), L.ptr(ws), L.stream_ptr()))
    return losses, dict(votes_xyz=d_votes, proposals_xyz=d_pxyz, proposals_output=d_pout)


def decode_boxes(proposals_xyz, proposals_output, nh=NH, ns=NS, nc=NC):
    """model.py:100-129 on the device: -> bboxes (B,P,8,3), scores (B,P) = max class logit (the inputs of NMS3D)."""
    from .synth import MEAN_SIZES
    pxyz = L.dev_f32(proposals_xyz, "decode_boxes proposals_xyz", 3, 3)
    pout = L.dev_f32(proposals_output, "decode_boxes proposals_output", 3)
    b, p = pxyz.shape[:2]
    if pout.shape[2] != 5 + 2 * nh + 4 * ns + nc:
        raise L.InvalidArgumentError("decode_boxes: proposals_output has %d channels" % pout.shape[2])
    mean = torch.tensor(MEAN_SIZES, dtype=torch.float32, device=pxyz.device).contiguous()
    boxes = torch.empty((b, p, 8, 3), dtype=torch.float32, device=pxyz.device)
    scores = torch.empty((b, p), dtype=torch.float32, device=pxyz.device)
    with L.device_guard(pxyz.device):
        L.check(L.lib().votenet_decode_boxes(b, p, nh, ns, nc, L.ptr(pxyz), L.ptr(pout), L.ptr(mean), L.ptr(boxes), L.ptr(scores),
                                             L.stream_ptr()))
    return boxes, scores
