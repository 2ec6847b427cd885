"""Mirror of the reference's tf_ops/sampling/tf_sampling.py on torch (ROCm) tensors.

Same function names, positional order and return arity; autograd wiring mirrors
RegisterGradient('GatherPoint') / ops.NoGradient('FarthestPointSample')
(tf_sampling.py:43-47,57).  prob_sample (tf_sampling.py:13-22) is not used by VoteNet
(utils.py:17) and is out of scope.
"""
import torch

from . import _lib as L

# bench.py sets this to a list to bracket every FPS launch with HIP events recorded on the
# launch stream (torch's current stream): entries are (start, end, b, n, npoint).
PROFILE_EVENTS = None


# Spatial indices left behind by farthest_point_sample (its temp scratch, 4096 < n <= 262144) or built by spatial_index():
# id(cloud tensor) -> (tensor, version, data pointer, index scratch).  tf_grouping.query_ball_point looks its candidate cloud up
# here, so the ball query that follows the sampling of the same cloud (every sample_and_group, utils.py:42-49) reuses the index.
# An entry is trusted while the tensor object, its version counter and its storage are unchanged.  A kernel that rewrites a cloud
# THROUGH ITS RAW POINTER (any votenet_* op writing into a buffer the caller reuses) bumps no version counter: such a caller must
# forget_index(cloud) -- the library's own writers allocate fresh outputs, so the hot path never needs to.  The cache holds at most six
# clouds (strong references: an id cannot be recycled while its entry lives); clear_index_cache() drops them.
_INDEX_CACHE = {}
INDEX_MIN_N, INDEX_MAX_N = 4097, 131072


def _remember_index(x, scratch):
    while len(_INDEX_CACHE) >= 6:
        _INDEX_CACHE.pop(next(iter(_INDEX_CACHE)))
    _INDEX_CACHE[id(x)] = (x, x._version, x.data_ptr(), scratch)


def cached_index(x):
    e = _INDEX_CACHE.get(id(x))
    if e is not None and e[0] is x and e[1] == x._version and e[2] == x.data_ptr():
        if x.is_cuda:
            e[3].record_stream(torch.cuda.current_stream(x.device))  # built on one stream, maybe consumed on another: keep its memory
        return e[3]
    return None


def forget_index(x):
    """Drop the remembered spatial index of `x` (after rewriting the cloud in place through a raw pointer)."""
    _INDEX_CACHE.pop(id(x), None)


def clear_index_cache():
    _INDEX_CACHE.clear()


def spatial_index(xyz):
    """(B,n,3) f32 -> the cloud's spatial index (votenet_spatial_index), remembered for the calls that follow."""
    got = cached_index(xyz)
    if got is not None:
        return got
    x = L.dev_f32(xyz.detach(), "spatial_index expects (batch_size,num_points,3) xyz shape", 3, 3)
    b, n, _ = x.shape
    scratch = torch.empty(L.lib().votenet_spatial_index_floats(b, n), dtype=torch.float32, device=x.device)
    with L.device_guard(x.device):
        L.check(L.lib().votenet_spatial_index(b, n, L.ptr(x), L.ptr(scratch), L.stream_ptr()))
    _remember_index(xyz, scratch)
    return scratch


def farthest_point_sample(npoint, inp):
    """tf_sampling.py:48-56.  int, (B,n,3) f32 -> (B,npoint) int32.  No gradient."""
    inp_arg = inp
    inp = L.dev_f32(inp.detach(), "FarthestPointSample expects (batch_size,num_points,3) inp shape", 3, 3)
    b, n, _ = inp.shape
    npoint = int(npoint)
    out = torch.empty((b, max(npoint, 0)), dtype=torch.int32, device=inp.device)
    nt = L.lib().votenet_fps_temp_floats(b, n)
    temp = torch.empty(nt, dtype=torch.float32, device=inp.device) if nt else None
    with L.device_guard(inp.device):
        if PROFILE_EVENTS is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(L.lib().votenet_farthest_point_sample(b, n, npoint, L.ptr(inp), L.ptr(temp), L.ptr(out), L.stream_ptr()))
        if PROFILE_EVENTS is not None:
            e1.record()
            PROFILE_EVENTS.append((e0, e1, b, n, npoint))
    if temp is not None and INDEX_MIN_N <= n <= 262144 and inp_arg.is_contiguous() and inp_arg.dtype == torch.float32:
        _remember_index(inp_arg, temp)  # the scratch now holds the cloud's spatial index
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        inp = L.dev_f32(inp, "GatherPoint expects (batch_size,num_points,3) inp shape", 3, 3)
        idx = L.dev_i32(idx, "GatherPoint expects (batch_size,num_result) idx shape", 2)
        if idx.shape[0] != inp.shape[0]:
            raise L.InvalidArgumentError("GatherPoint expects (batch_size,num_result) idx shape")
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        with L.device_guard(inp.device):
            L.check(L.lib().votenet_gather_point(b, n, m, L.ptr(inp), L.ptr(idx), L.ptr(out), L.stream_ptr()))
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        return gather_point_grad_raw(ctx.n, idx, out_g), None


def gather_point_grad_raw(n, idx, out_g, into=None):
    """GatherPointGrad (tf_sampling.cpp:150-178): zero-filled (B,n,3) buffer + scatter-add.  into: an existing contiguous (B,n,3)
    gradient to accumulate on instead (d_xyz + GatherPointGrad(...) without the zero fill and the add)."""
    out_g = L.dev_f32(out_g, "GatherPointGradGpuOp expects (batch_size,num_result,3) out_g shape", 3, 3)
    b, m = idx.shape
    if into is not None:
        if tuple(into.shape) != (b, n, 3) or not into.is_contiguous():
            raise L.InvalidArgumentError("GatherPointGrad: `into` must be a contiguous (batch_size,n,3) tensor")
        inp_g = into
    else:
        from . import mlp as M
        inp_g = M._zeros_f32((b, n, 3), out_g.device)  # tf_sampling.cpp:174
    with L.device_guard(out_g.device):
        L.check(L.lib().votenet_gather_point_grad(b, n, m, L.ptr(out_g), L.ptr(idx), L.ptr(inp_g), L.stream_ptr()))
    return inp_g


def gather_point(inp, idx):
    """tf_sampling.py:29-36.  (B,n,3) f32, (B,m) int32 -> (B,m,3) f32; d/d inp registered."""
    return _GatherPoint.apply(inp, idx)


def prob_sample(inp, inpr):
    """tf_sampling.py:13-21.  (batch_size, ncategory) f32 weights, (batch_size, npoints) f32 uniform draws ->
    (batch_size, npoints) i32 category ids; no gradient (tf_sampling.py:22)."""
    inp = L.dev_f32(inp.detach(), "ProbSample expects (batch_size,num_choices) inp shape", 2)
    inpr = L.dev_f32(inpr.detach(), "ProbSample expects (batch_size,num_points) inpr shape", 2)
    b, n = inp.shape
    if inpr.shape[0] != b:
        raise L.InvalidArgumentError("ProbSample expects (batch_size,num_points) inpr shape")
    m = inpr.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)  # tf_sampling.cpp:83 allocate_temp
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    with L.device_guard(inp.device):
        L.check(L.lib().votenet_prob_sample(b, n, m, L.ptr(inp), L.ptr(inpr), L.ptr(temp), L.ptr(out), L.stream_ptr()))
    return out
