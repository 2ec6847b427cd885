"""Front-end of the REFERENCE's own device kernels built for gfx950 (oracle/_ref/libref_{sampling,grouping}_gpu.so: tf_sampling_g.cu
and tf_grouping_g.cu compiled where they lie, oracle/Makefile).

TEST INFRASTRUCTURE ONLY: loaded by tests/ (gpu marker) and tests/golden/make_ref_gpu_golden.py -- never by votenet_amd/, bench.py or
smoke().  Needs a GPU; numpy in, numpy out (torch is used for the device buffers only).  What the reference's TF op wrappers do
around a launch is done here as they do it: the 32*n / b*n temp buffers (tf_sampling.cpp:86,115), the zero fill before the two
scatter-adds (tf_sampling.cpp:174, tf_grouping.cpp:204).
"""
import ctypes
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_LIBS = {}


def _lib(name):
    if name not in _LIBS:
        path = os.path.join(_DIR, "libref_%s_gpu.so" % name)
        _LIBS[name] = ctypes.CDLL(path) if os.path.exists(path) else None
    return _LIBS[name]


def available():
    """Both libraries built (this container, reference tree mounted; they travel to the GPU box) and a GPU to run them on."""
    import torch
    return _lib("sampling") is not None and _lib("grouping") is not None and torch.cuda.is_available()


def _dev(a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to("cuda:0")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _call(fn, *args):
    import torch
    torch.cuda.synchronize()  # the launchers use the null stream
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("reference kernel failed: hipError %d" % rc)


def farthest_point_sample(npoint, inp):
    """tf_sampling.cpp:97-120 + tf_sampling_g.cu:105-170.  inp (b,n,3) f32 -> (b,npoint) i32."""
    import torch
    x = _dev(inp, np.float32)
    b, n, _ = x.shape
    temp = torch.empty((32, n), dtype=torch.float32, device=x.device)
    out = torch.zeros((b, npoint), dtype=torch.int32, device=x.device)
    _call(_lib("sampling").ref_gpu_farthest_point_sample, b, n, npoint, _p(x), _p(temp), _p(out))
    return out.cpu().numpy()


def gather_point(inp, idx):
    import torch
    x, i = _dev(inp, np.float32), _dev(idx, np.int32)
    b, n, _ = x.shape
    m = i.shape[1]
    out = torch.zeros((b, m, 3), dtype=torch.float32, device=x.device)
    _call(_lib("sampling").ref_gpu_gather_point, b, n, m, _p(x), _p(i), _p(out))
    return out.cpu().numpy()


def gather_point_grad(n, idx, out_g):
    """tf_sampling.cpp:174-175: zero fill, then the atomic scatter-add (the order of the float additions is the hardware's)."""
    import torch
    i, g = _dev(idx, np.int32), _dev(out_g, np.float32)
    b, m = i.shape
    inp_g = torch.zeros((b, n, 3), dtype=torch.float32, device=i.device)
    _call(_lib("sampling").ref_gpu_scatter_add_point, b, n, m, _p(g), _p(i), _p(inp_g))
    return inp_g.cpu().numpy()


def cumsum(inp):
    import torch
    x = _dev(inp, np.float32)
    b, n = x.shape
    out = torch.zeros_like(x)
    _call(_lib("sampling").ref_gpu_cumsum, b, n, _p(x), _p(out))
    return out.cpu().numpy()


def prob_sample(inp_p, inp_r):
    """tf_sampling.cpp:68-91 + tf_sampling_g.cu:7-104,198-201: (b,n) weights, (b,m) uniforms -> (b,m) i32."""
    import torch
    p, r = _dev(inp_p, np.float32), _dev(inp_r, np.float32)
    b, n = p.shape
    m = r.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device=p.device)
    out = torch.zeros((b, m), dtype=torch.int32, device=p.device)
    _call(_lib("sampling").ref_gpu_prob_sample, b, n, m, _p(p), _p(r), _p(temp), _p(out))
    return out.cpu().numpy()


def query_ball_point(radius, nsample, xyz1, xyz2, fill=0):
    """tf_grouping_g.cu:3-36.  A query without a neighbour keeps `fill` in its row (the reference leaves it unwritten)."""
    import torch
    a, q = _dev(xyz1, np.float32), _dev(xyz2, np.float32)
    b, n, _ = a.shape
    m = q.shape[1]
    idx = torch.full((b, m, nsample), fill, dtype=torch.int32, device=a.device)
    cnt = torch.zeros((b, m), dtype=torch.int32, device=a.device)
    _call(_lib("grouping").ref_gpu_query_ball_point, b, n, m, ctypes.c_float(np.float32(radius)), nsample, _p(a), _p(q), _p(idx), _p(cnt))
    return idx.cpu().numpy(), cnt.cpu().numpy()


def group_point(points, idx):
    import torch
    x, i = _dev(points, np.float32), _dev(idx, np.int32)
    b, n, c = x.shape
    _, m, k = i.shape
    out = torch.zeros((b, m, k, c), dtype=torch.float32, device=x.device)
    _call(_lib("grouping").ref_gpu_group_point, b, n, c, m, k, _p(x), _p(i), _p(out))
    return out.cpu().numpy()


def group_point_grad(n, idx, grad_out):
    """tf_grouping.cpp:204-205: zero fill, then atomic adds in the hardware's order."""
    import torch
    i, g = _dev(idx, np.int32), _dev(grad_out, np.float32)
    b, m, k = i.shape
    c = g.shape[3]
    out = torch.zeros((b, n, c), dtype=torch.float32, device=i.device)
    _call(_lib("grouping").ref_gpu_group_point_grad, b, n, c, m, k, _p(g), _p(i), _p(out))
    return out.cpu().numpy()


def selection_sort(k, dist):
    """tf_grouping_g.cu:83-123: -> (idx, val), both (b,m,n); the first k entries of a row are meaningful."""
    import torch
    d = _dev(dist, np.float32)
    b, m, n = d.shape
    outi = torch.zeros((b, m, n), dtype=torch.int32, device=d.device)
    out = torch.zeros_like(d)
    _call(_lib("grouping").ref_gpu_selection_sort, b, n, m, k, _p(d), _p(outi), _p(out))
    return outi.cpu().numpy(), out.cpu().numpy()
