"""Mirror of the reference's tf_ops/3d_interpolation/tf_interpolate.py on torch (ROCm) tensors.

The reference runs these ops on the CPU (tf_interpolate.cpp:187,222,262); here they are HIP
kernels on device tensors.  three_nn has no gradient (tf_interpolate.py:18);
three_interpolate's gradient is w.r.t. points only (tf_interpolate.py:29-34).
"""
import torch

from . import _lib as L


def three_nn(xyz1, xyz2):
    """tf_interpolate.py:8-17.  (B,n,3) unknown, (B,m,3) known -> (dist (B,n,3) SQUARED, idx (B,n,3) i32)."""
    xyz1 = L.dev_f32(xyz1.detach(), "ThreeNN expects (b,n,3) xyz1 shape.", 3, 3)
    xyz2 = L.dev_f32(xyz2.detach(), "ThreeNN expects (b,m,3) xyz2 shape.", 3, 3)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    with L.device_guard(xyz1.device):
        L.check(L.lib().votenet_three_nn(b, n, m, L.ptr(xyz1), L.ptr(xyz2), L.ptr(dist), L.ptr(idx), L.stream_ptr()))
    return dist, idx


def three_nn_weights(dist):
    """utils.py:279-282 as one kernel: d=max(d,1e-10); w=(1/d)/sum(1/d).  No gradient (dist has none)."""
    dist = L.dev_f32(dist.detach(), "three_nn_weights expects (b,n,3) dist shape", 3, 3)
    b, n, _ = dist.shape
    w = torch.empty_like(dist)
    with L.device_guard(dist.device):
        L.check(L.lib().votenet_three_nn_weights(b, n, L.ptr(dist), L.ptr(w), L.stream_ptr()))
    return w


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        points = L.dev_f32(points, "ThreeInterpolate expects (b,m,c) points shape", 3)
        idx = L.dev_i32(idx, "ThreeInterpolate expects (b,n,3) idx shape", 3)
        weight = L.dev_f32(weight, "ThreeInterpolate expects (b,n,3) weight shape", 3, 3)
        b, m, c = points.shape
        n = idx.shape[1]
        if idx.shape[0] != b or idx.shape[2] != 3:
            raise L.InvalidArgumentError("ThreeInterpolate expects (b,n,3) idx shape")
        if weight.shape[0] != b or weight.shape[1] != n:
            raise L.InvalidArgumentError("ThreeInterpolate expects (b,n,3) weight shape")
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        with L.device_guard(points.device):
            L.check(L.lib().votenet_three_interpolate(b, m, c, n, L.ptr(points), L.ptr(idx), L.ptr(weight), L.ptr(out),
                                                      L.stream_ptr()))
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return three_interpolate_grad_raw(ctx.m, idx, weight, grad_out), None, None


def three_interpolate_grad_raw(m, idx, weight, grad_out):
    """ThreeInterpolateGrad (tf_interpolate.cpp:226-262): zero-filled (b,m,c) buffer + scatter-add."""
    grad_out = L.dev_f32(grad_out, "ThreeInterpolateGrad expects (b,n,c) grad_out shape", 3)
    b, n, c = grad_out.shape
    from . import mlp as M
    if M.DETERMINISTIC and c <= 256 and getattr(idx, "_inv", None) is not None:  # gather-sum over the taps' inverse index (csr.hip)
        return M.csr_gather_sum(grad_out.view(b * n, c), idx._inv, b * m, weight=weight.contiguous(), div=3).view(b, m, c)
    g = torch.zeros((b, m, c), dtype=torch.float32, device=grad_out.device)  # tf_interpolate.cpp:258
    with L.device_guard(grad_out.device):
        L.check(L.lib().votenet_three_interpolate_grad(b, n, c, m, L.ptr(grad_out), L.ptr(idx), L.ptr(weight), L.ptr(g),
                                                       L.stream_ptr()))
    return g


def three_interpolate(points, idx, weight):
    """tf_interpolate.py:19-28.  (b,m,c), (b,n,3) i32, (b,n,3) f32 -> (b,n,c)."""
    return _ThreeInterpolate.apply(points, idx, weight)
