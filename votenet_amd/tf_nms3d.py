"""Mirror of the reference's tf_ops/3d_nms/tf_nms3d.py on torch (ROCm) tensors."""
import torch

from . import _lib as L


def iou3d_matrix(bboxes):
    """(B,N,8,3) corner boxes -> (B,N,N) 3D IoU of every ordered pair (tf_nms3d.cpp:178-192)."""
    bboxes = L.dev_f32(bboxes.detach(), "3D NMS expects (batch_size, nbbox, 8, 3) bbox shape.", 4, 3)
    if bboxes.shape[2] != 8:
        raise L.InvalidArgumentError("3D NMS expects (batch_size, nbbox, 8, 3) bbox shape.")
    b, n = bboxes.shape[:2]
    iou = torch.empty((b, n, n), dtype=torch.float32, device=bboxes.device)
    with L.device_guard(bboxes.device):
        L.check(L.lib().votenet_iou3d_matrix(b, n, L.ptr(bboxes), L.ptr(iou), L.stream_ptr()))
    return iou


def iou3d_cross(boxes_a, boxes_b):
    """(B,n,8,3), (B,m,8,3) corner boxes -> (B,n,m) 3D IoU of every pair across the two sets (evaluator.py:26-39)."""
    a = L.dev_f32(boxes_a.detach(), "iou3d_cross expects (batch, n, 8, 3) boxes", 4, 3)
    bset = L.dev_f32(boxes_b.detach(), "iou3d_cross expects (batch, m, 8, 3) boxes", 4, 3)
    if a.shape[2] != 8 or bset.shape[2] != 8 or a.shape[0] != bset.shape[0]:
        raise L.InvalidArgumentError("iou3d_cross expects (batch, n, 8, 3) and (batch, m, 8, 3) boxes")
    b, n, m = a.shape[0], a.shape[1], bset.shape[1]
    iou = torch.empty((b, n, m), dtype=torch.float32, device=a.device)
    with L.device_guard(a.device):
        L.check(L.lib().votenet_iou3d_cross(b, n, m, L.ptr(a), L.ptr(bset), L.ptr(iou), L.stream_ptr()))
    return iou


def NMS3D(bboxes, scores, objectiveness, iou_threshold, padded=False):
    """tf_nms3d.py:11-12.  (B,N,8,3), (B,N), (B,N,2) f32, scalar in [0,1] -> (Nsel,2) int32 [batch, box]
    in descending-score visit order over the whole batch (tf_nms3d.cpp:202-273).  No gradient.
    padded=True: -> (rows (B*N,2), count (1,) int32) both on the device, rows[:count] valid -- no host synchronisation
    (the (Nsel,2) shape of the reference needs the count on the host, which stalls a pipelined inference loop)."""
    bboxes = L.dev_f32(bboxes.detach(), "3D NMS expects (batch_size, nbbox, 8, 3) bbox shape.", 4, 3)
    if bboxes.shape[2] != 8:
        raise L.InvalidArgumentError("3D NMS expects (batch_size, nbbox, 8, 3) bbox shape.")
    b, n = bboxes.shape[:2]
    scores = L.dev_f32(scores.detach(), "3D NMS expects (batch_size, nbbox) scores shape.", 2)
    if tuple(scores.shape) != (b, n):
        raise L.InvalidArgumentError("3D NMS expects (batch_size, nbbox) scores shape.")
    objectiveness = L.dev_f32(objectiveness.detach(), "3D NMS expects (batch_size, nbbox, 2) objectiveness shape.", 3, 2)
    if tuple(objectiveness.shape) != (b, n, 2):
        raise L.InvalidArgumentError("3D NMS expects (batch_size, nbbox, 2) objectiveness shape.")
    if isinstance(iou_threshold, torch.Tensor):
        if iou_threshold.numel() != 1:
            raise L.InvalidArgumentError("3D NMS expects scalar threshold")
        iou_threshold = float(iou_threshold.item())
    out = torch.empty((max(b * n, 1), 2), dtype=torch.int32, device=bboxes.device)
    count = torch.zeros(1, dtype=torch.int32, device=bboxes.device)
    wbytes = L.lib().votenet_nms3d_workspace_bytes(b, n)
    ws = torch.empty(wbytes, dtype=torch.uint8, device=bboxes.device)
    with L.device_guard(bboxes.device):
        L.check(L.lib().votenet_nms3d(b, n, L.ptr(bboxes), L.ptr(scores), L.ptr(objectiveness), float(iou_threshold),
                                      L.ptr(out), L.ptr(count), L.ptr(ws), wbytes, L.stream_ptr()))
    if padded:
        return out, count
    nsel = int(count.item())  # dynamic output length (the TF op allocates (Nsel,2) the same way)
    return out[:nsel]
