"""Mirror of the reference's tf_ops/grouping/tf_grouping.py on torch (ROCm) tensors.

query_ball_point has no gradient (tf_grouping.py:21); group_point's gradient w.r.t. points is
GroupPointGrad (tf_grouping.py:42-46).  select_top_k / knn_point (tf_grouping.py:22-32,47-73) are reachable only with
knn=True (utils.py:46-47), which model.py never sets; they are provided for tf_ops API parity.
"""
import torch

from . import _lib as L


# bench.py sets this to a list to bracket every ball-query launch with HIP events recorded on the launch stream:
# entries are (start, end, b, n, m, nsample).
PROFILE_EVENTS = None
USE_INDEX = True  # False: query_ball_point always scans all candidates (votenet_query_ball_point), for A/B and tests


def query_ball_point(radius, nsample, xyz1, xyz2):
    """tf_grouping.py:8-20.  float, int, (B,n,3), (B,m,3) -> (idx (B,m,nsample) i32, pts_cnt (B,m) i32).
    Large candidate clouds (4096 < n <= 131072) go through their spatial index -- the one the farthest-point sampling of the
    same tensor left behind, or one built here -- with identical results (USE_INDEX = False: always the full scan)."""
    from . import tf_sampling
    index = None
    if USE_INDEX and xyz1.dim() == 3 and tf_sampling.INDEX_MIN_N <= xyz1.shape[1] <= tf_sampling.INDEX_MAX_N and xyz1.is_cuda \
            and xyz1.dtype == torch.float32 and xyz1.is_contiguous() and xyz1.shape[2] == 3 and float(radius) > 0 and int(nsample) > 0:
        index = tf_sampling.cached_index(xyz1)
        if index is None:
            index = tf_sampling.spatial_index(xyz1)
    xyz1 = L.dev_f32(xyz1.detach(), "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.", 3, 3)
    xyz2 = L.dev_f32(xyz2.detach(), "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.", 3, 3)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    nsample = int(nsample)
    idx = torch.empty((b, m, max(nsample, 0)), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with L.device_guard(xyz1.device):
        if PROFILE_EVENTS is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if index is not None:
            L.check(L.lib().votenet_query_ball_point_indexed(b, n, m, float(radius), nsample, L.ptr(xyz1), L.ptr(xyz2), L.ptr(index),
                                                             L.ptr(idx), L.ptr(cnt), L.stream_ptr()))
        else:
            L.check(L.lib().votenet_query_ball_point(b, n, m, float(radius), nsample, L.ptr(xyz1), L.ptr(xyz2), L.ptr(idx),
                                                     L.ptr(cnt), L.stream_ptr()))
        if PROFILE_EVENTS is not None:
            e1.record()
            PROFILE_EVENTS.append((e0, e1, b, n, m, nsample))
    return idx, cnt


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        points = L.dev_f32(points, "GroupPoint expects (batch_size, num_points, channel) points shape", 3)
        idx = L.dev_i32(idx, "GroupPoint expects (batch_size, npoints, nsample) idx shape", 3)
        if idx.shape[0] != points.shape[0]:
            raise L.InvalidArgumentError("GroupPoint expects (batch_size, npoints, nsample) idx shape")
        b, n, c = points.shape
        _, m, k = idx.shape
        out = torch.empty((b, m, k, c), dtype=torch.float32, device=points.device)
        with L.device_guard(points.device):
            L.check(L.lib().votenet_group_point(b, n, c, m, k, L.ptr(points), L.ptr(idx), L.ptr(out), L.stream_ptr()))
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return group_point_grad_raw(ctx.n, idx, grad_out), None


def group_point_grad_raw(n, idx, grad_out):
    """GroupPointGrad (tf_grouping.cpp:173-208): zero-filled (B,n,c) buffer + scatter-add."""
    grad_out = L.dev_f32(grad_out, "GroupPointGrad expects (batch_size, npoints, nsample, channel) grad_out shape", 4)
    b, m, k, c = grad_out.shape
    g = torch.zeros((b, n, c), dtype=torch.float32, device=grad_out.device)  # tf_grouping.cpp:204
    with L.device_guard(grad_out.device):
        L.check(L.lib().votenet_group_point_grad(b, n, c, m, k, L.ptr(grad_out), L.ptr(idx), L.ptr(g), L.stream_ptr()))
    return g


def group_point(points, idx):
    """tf_grouping.py:33-41.  (B,n,c) f32, (B,m,nsample) i32 -> (B,m,nsample,c) f32."""
    return _GroupPoint.apply(points, idx)


def select_top_k(k, dist):
    """tf_grouping.py:22-32.  int, (b,m,n) f32 -> (idx (b,m,n) i32, dist_out (b,m,n) f32): the first k columns are the k
    smallest distances, ascending, with their indices; the rest is the remainder as SelectionSort's swaps leave it."""
    dist = L.dev_f32(dist.detach(), "SelectionSort expects (b,m,n) dist shape.", 3)
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    with L.device_guard(dist.device):
        L.check(L.lib().votenet_selection_sort(b, n, m, int(k), L.ptr(dist), L.ptr(outi), L.ptr(out), L.stream_ptr()))
    return outi, out


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:47-73.  int, xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries -> (val (b,m,k) squared L2 distances,
    idx (b,m,k) i32).  One kernel: the (b,m,n) distance tensor of the reference is never formed."""
    xyz1 = L.dev_f32(xyz1.detach(), "knn_point expects (batch_size, ndataset, c) xyz1 shape.", 3)
    xyz2 = L.dev_f32(xyz2.detach(), "knn_point expects (batch_size, npoint, c) xyz2 shape.", 3)
    b, n, c = xyz1.shape
    if xyz2.shape[0] != b or xyz2.shape[2] != c:
        raise L.InvalidArgumentError("knn_point expects xyz1 (b,n,c) and xyz2 (b,m,c) with the same b and c")
    m, k = xyz2.shape[1], int(k)
    val = torch.empty((b, m, max(k, 0)), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, m, max(k, 0)), dtype=torch.int32, device=xyz1.device)
    nws = int(L.lib().votenet_knn_workspace_bytes(b, n, m))
    ws = torch.empty(nws, dtype=torch.uint8, device=xyz1.device) if nws else None
    with L.device_guard(xyz1.device):
        L.check(L.lib().votenet_knn_point(b, n, m, c, k, L.ptr(xyz1), L.ptr(xyz2), L.ptr(val), L.ptr(idx), L.ptr(ws), L.stream_ptr()))
    return val, idx
