"""Mirror of the reference's tf_ops/3d_interpolation/tf_interpolate.py on torch (ROCm) tensors.

The reference runs these ops on the CPU (tf_interpolate.cpp:187,222,262); here they are HIP
kernels on device tensors.  three_nn has no gradient (tf_interpolate.py:18);
three_interpolate's gradient is w.r.t. points only (tf_interpolate.py:29-34).
"""
import torch

from . import _lib as L

# ThreeInterpolateGrad as a gather-sum over the taps' inverse index wherever the index carries one (FPModule.geometry attaches it: it
# depends on coordinates only and is built with the geometry chain, a step ahead): 24.6 -> 10.4 us at 1024 <- 512 x 256 channels,
# 14.4 -> 8.0 us at 512 <- 256 against the scatter-add with atomics (tools/probe/interp_grad_time.py; in the train step the 20 us are
# within the run-to-run noise), no zero fill, and one fixed summation order
GATHER_GRAD = True


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
    """ThreeInterpolateGrad (tf_interpolate.cpp:226-262): zero-filled (b,m,c) buffer + scatter-add.
    grad_out: (b,n,c), or a column slice of a wider row-major tensor (a view with unit channel stride: the gradient of the FP
    layer's concat is [d interpolated | d skip], utils.py:286) -- read in place, no copy."""
    if not (isinstance(grad_out, torch.Tensor) and grad_out.is_cuda and grad_out.dtype == torch.float32 and grad_out.dim() == 3):
        grad_out = L.dev_f32(grad_out, "ThreeInterpolateGrad expects (b,n,c) grad_out shape", 3)
    b, n, c = grad_out.shape
    strided = not grad_out.is_contiguous()
    if strided and not (grad_out.stride(2) == 1 and grad_out.stride(0) == n * grad_out.stride(1) and grad_out.stride(1) >= c):
        grad_out, strided = grad_out.contiguous(), False
    from . import mlp as M
    if (M.DETERMINISTIC or GATHER_GRAD) and c <= 256 and getattr(idx, "_inv", None) is not None:
        # gather-sum over the taps' inverse index (csr.hip): no atomics, no zero fill, one fixed order; a column slice is read in place
        src = grad_out.as_strided((b * n, c), (grad_out.stride(1), 1)) if strided else grad_out.view(b * n, c)
        return M.csr_gather_sum(src, idx._inv, b * m, weight=weight.contiguous(), div=3).view(b, m, c)
    g = M._zeros_f32((b, m, c), grad_out.device)  # tf_interpolate.cpp:258 (inside a pass: a carve-out of its one zero fill)
    with L.device_guard(grad_out.device):
        if strided:
            L.check(L.lib().votenet_three_interpolate_grad_strided(b, n, c, m, L.ptr(grad_out), grad_out.stride(1), 0, L.ptr(idx),
                                                                   L.ptr(weight), L.ptr(g), L.stream_ptr()))
        else:
            L.check(L.lib().votenet_three_interpolate_grad(b, n, c, m, L.ptr(grad_out), L.ptr(idx), L.ptr(weight), L.ptr(g),
                                                           L.stream_ptr()))
    return g


def three_interpolate_concat(points, idx, weight, skip):
    """[three_interpolate(points, idx, weight) | skip] (b,n,c + c1) in ONE launch: the FP layer's concat of utils.py:283-286 written
    by the interpolation kernel itself (no gradient wiring: the hot path's backward is explicit)."""
    points = L.dev_f32(points, "ThreeInterpolate expects (b,m,c) points shape", 3)
    skip = L.dev_f32(skip, "ThreeInterpolate expects (b,n,c1) skip shape", 3)
    idx = L.dev_i32(idx, "ThreeInterpolate expects (b,n,3) idx shape", 3)
    weight = L.dev_f32(weight, "ThreeInterpolate expects (b,n,3) weight shape", 3, 3)
    b, m, c = points.shape
    n, c1 = idx.shape[1], skip.shape[2]
    if idx.shape[0] != b or weight.shape[:2] != idx.shape[:2] or skip.shape[:2] != idx.shape[:2]:
        raise L.InvalidArgumentError("ThreeInterpolate expects (b,n,3) idx / weight and (b,n,c1) skip shapes")
    out = torch.empty((b, n, c + c1), dtype=torch.float32, device=points.device)
    with L.device_guard(points.device):
        L.check(L.lib().votenet_three_interpolate_concat(b, m, c, n, L.ptr(points), L.ptr(idx), L.ptr(weight), L.ptr(skip), c1,
                                                         L.ptr(out), L.stream_ptr()))
    return out


def three_interpolate(points, idx, weight):
    """tf_interpolate.py:19-28.  (b,m,c), (b,n,3) i32, (b,n,3) f32 -> (b,n,c)."""
    return _ThreeInterpolate.apply(points, idx, weight)
