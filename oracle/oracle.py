"""ctypes/numpy front-end of the CPU oracle (oracle/liboracle.so) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY.  Import this from tests/, from __graft_entry__.smoke() and from
bench.py's cpu_baseline leg -- never from votenet_amd/ (the product path must fail loudly
when the HIP library is missing instead of falling back to this).

Each wrapper takes/returns C-contiguous numpy arrays (float32 / int32) and mirrors the
reference operator it restates (file:line in oracle.h).
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = {}

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(quiet=True):
    """Compile liboracle.so (and oracle/_ref when the reference tree is mounted)."""
    out = subprocess.run(["make", "-C", _DIR], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _load(name):
    path = os.path.join(_DIR, name)
    if not os.path.exists(path):
        build()
    L = ctypes.CDLL(path)
    L.oracle_bev_intersection.restype = ctypes.c_float
    L.oracle_iou3d.restype = ctypes.c_float
    L.oracle_nms3d.restype = ctypes.c_int
    return L


_OMP = None
_THREADS = 1


def set_threads(n):
    """n == 1 (default): liboracle.so, the single-thread restatement with the reference's loop structure.  n > 1: the same
    sources built with -fopenmp (liboracle_omp.so) on n threads -- bit-identical results (tests/test_oracle_omp.py); this is
    bench.py's all-core cpu_baseline figure.  Returns the previous setting."""
    global _THREADS, _OMP
    prev, _THREADS = _THREADS, max(1, int(n))
    if _THREADS > 1:
        if _OMP is None:
            os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle workers sleep instead of spinning beside the GPU's host threads
            _OMP = _load("liboracle_omp.so")
            _OMP.oracle_set_threads.restype = None
        _OMP.oracle_set_threads(_THREADS)
    return prev


def lib():
    global _LIB
    if _THREADS > 1:
        return _OMP
    if _LIB is None:
        _LIB = _load("liboracle.so")
    return _LIB


def ref(name):
    """Load oracle/_ref/libref_<name>.so (the reference's own compiled code) or None."""
    if name not in _REF:
        path = os.path.join(_DIR, "_ref", "libref_%s.so" % name)
        _REF[name] = ctypes.CDLL(path) if os.path.exists(path) else None
    return _REF[name]


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---------------------------------------------------------------- sampling
def farthest_point_sample(npoint, inp, closed=False):
    inp = _f32(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    fn = lib().oracle_farthest_point_sample_closed if closed else lib().oracle_farthest_point_sample
    fn(b, n, npoint, _fp(inp), _ip(out))
    return out


def gather_point(inp, idx):
    inp, idx = _f32(inp), _i32(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.zeros((b, m, 3), np.float32)
    lib().oracle_gather_point(b, n, m, _fp(inp), _ip(idx), _fp(out))
    return out


def gather_point_grad(inp, idx, out_g):
    inp, idx, out_g = _f32(inp), _i32(idx), _f32(out_g)
    b, n, _ = inp.shape
    m = idx.shape[1]
    g = np.zeros((b, n, 3), np.float32)
    lib().oracle_gather_point_grad(b, n, m, _fp(out_g), _ip(idx), _fp(g))
    return g


# ---------------------------------------------------------------- grouping
def query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    cnt = np.zeros((b, m), np.int32)
    lib().oracle_query_ball_point(b, n, m, ctypes.c_float(np.float32(radius)), nsample,
                                  _fp(xyz1), _fp(xyz2), _ip(idx), _ip(cnt))
    return idx, cnt


def group_point(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, k = idx.shape
    out = np.zeros((b, m, k, c), np.float32)
    lib().oracle_group_point(b, n, c, m, k, _fp(points), _ip(idx), _fp(out))
    return out


def group_point_grad(points, idx, grad_out):
    points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
    b, n, c = points.shape
    _, m, k = idx.shape
    g = np.zeros((b, n, c), np.float32)
    lib().oracle_group_point_grad(b, n, c, m, k, _fp(grad_out), _ip(idx), _fp(g))
    return g


def group_concat(xyz, new_xyz, points, idx):
    xyz, new_xyz, idx = _f32(xyz), _f32(new_xyz), _i32(idx)
    b, n, _ = xyz.shape
    _, m, k = idx.shape
    c = 0 if points is None else points.shape[2]
    pts = None if points is None else _f32(points)
    out = np.zeros((b, m, k, 3 + c), np.float32)
    lib().oracle_group_concat(b, n, c, m, k, _fp(xyz), _fp(new_xyz),
                              _fp(pts) if pts is not None else None, _ip(idx), _fp(out))
    return out


# ---------------------------------------------------------------- interpolation
def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().oracle_three_nn(b, n, m, _fp(xyz1), _fp(xyz2), _fp(dist), _ip(idx))
    return dist, idx


def three_nn_weights(dist):
    dist = _f32(dist)
    b, n, _ = dist.shape
    w = np.zeros_like(dist)
    lib().oracle_three_nn_weights(b, n, _fp(dist), _fp(w))
    return w


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.zeros((b, n, c), np.float32)
    lib().oracle_three_interpolate(b, m, c, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def three_interpolate_grad(points, idx, weight, grad_out):
    points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)
    lib().oracle_three_interpolate_grad(b, n, c, m, _fp(grad_out), _ip(idx), _fp(weight), _fp(g))
    return g


# ---------------------------------------------------------------- 3D IoU / NMS
def bev_intersection(b1, b2):
    b1, b2 = _f32(b1), _f32(b2)
    return float(lib().oracle_bev_intersection(_fp(b1), _fp(b2)))


def iou3d(b1, b2):
    b1, b2 = _f32(b1), _f32(b2)
    return float(lib().oracle_iou3d(_fp(b1), _fp(b2)))


def iou3d_matrix(bboxes):
    bboxes = _f32(bboxes)
    n = bboxes.shape[0]
    out = np.zeros((n, n), np.float32)
    lib().oracle_iou3d_matrix(n, _fp(bboxes), _fp(out))
    return out


def nms3d(bboxes, scores, objectiveness, iou_threshold):
    bboxes, scores, objectiveness = _f32(bboxes), _f32(scores), _f32(objectiveness)
    b, n = scores.shape
    out = np.zeros((b * n, 2), np.int32)
    cnt = lib().oracle_nms3d(b, n, _fp(bboxes), _fp(scores), _fp(objectiveness),
                             ctypes.c_float(np.float32(iou_threshold)), _ip(out))
    return out[:cnt].copy()


# ---------------------------------------------------------------- grouped MLP
def linear(x, w, bias=None):
    x, w = _f32(x), _f32(w)
    rows, cin = x.shape
    cout = w.shape[1]
    z = np.zeros((rows, cout), np.float32)
    bptr = _fp(_f32(bias)) if bias is not None else None
    lib().oracle_linear(ctypes.c_long(rows), cin, cout, _fp(x), _fp(w), bptr, _fp(z))
    return z


def bn_stats(z):
    z = _f32(z)
    rows, c = z.shape
    mean = np.zeros(c, np.float32)
    var = np.zeros(c, np.float32)
    lib().oracle_bn_stats(ctypes.c_long(rows), c, _fp(z), _fp(mean), _fp(var))
    return mean, var


def bn_relu(z, mean, var, gamma, beta, eps=1e-5, relu=True):
    z = _f32(z)
    rows, c = z.shape
    y = np.zeros_like(z)
    lib().oracle_bn_relu(ctypes.c_long(rows), c, _fp(z), _fp(_f32(mean)), _fp(_f32(var)),
                         _fp(_f32(gamma)), _fp(_f32(beta)), ctypes.c_float(eps), int(relu), _fp(y))
    return y


def max_over_k(y, k):
    y = _f32(y)
    rows, c = y.shape
    groups = rows // k
    out = np.zeros((groups, c), np.float32)
    lib().oracle_max_over_k(ctypes.c_long(groups), k, c, _fp(y), _fp(out))
    return out


# ---------------------------------------------------------------- unused-by-VoteNet variants
def select_top_k(k, dist):
    dist = _f32(dist)
    b, m, n = dist.shape
    outi, out = np.zeros((b, m, n), np.int32), np.zeros((b, m, n), np.float32)
    lib().oracle_selection_sort(b, n, m, k, _fp(dist), _ip(outi), _fp(out))
    return outi, out


def knn_dist(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, m, n), np.float32)
    lib().oracle_knn_dist(b, n, m, c, _fp(xyz1), _fp(xyz2), _fp(dist))
    return dist


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:47-71 -> (val (b,m,k), idx (b,m,k))."""
    outi, out = select_top_k(k, knn_dist(xyz1, xyz2))
    return out[:, :, :k].copy(), outi[:, :, :k].copy()


def cumsum(inp):
    inp = _f32(inp)
    b, n = inp.shape
    out = np.zeros((b, n), np.float32)
    lib().oracle_cumsum(b, n, _fp(inp), _fp(out))
    return out


def prob_sample(inp, inpr):
    inp, inpr = _f32(inp), _f32(inpr)
    b, n = inp.shape
    m = inpr.shape[1]
    temp, out = np.zeros((b, n), np.float32), np.zeros((b, m), np.int32)
    lib().oracle_prob_sample(b, n, m, _fp(inp), _fp(inpr), _fp(temp), _ip(out))
    return out


# ---------------------------------------------------------------- reference (oracle/_ref)
def ref_select_top_k(k, dist):
    r = ref("selection_sort")
    dist = _f32(dist)
    b, m, n = dist.shape
    outi, out = np.zeros((b, m, n), np.int32), np.zeros((b, m, n), np.float32)
    r.ref_selection_sort(b, n, m, k, _fp(dist), _ip(outi), _fp(out))
    return outi, out


def ref_query_ball_point(radius, nsample, xyz1, xyz2):
    r = ref("grouping")
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)  # test/query_ball_point.cpp:95 memsets idx too
    r.ref_query_ball_point(b, n, m, ctypes.c_float(np.float32(radius)), nsample, _fp(xyz1), _fp(xyz2), _ip(idx))
    return idx


def ref_group_point(points, idx):
    r = ref("grouping")
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, k = idx.shape
    out = np.zeros((b, m, k, c), np.float32)
    r.ref_group_point(b, n, c, m, k, _fp(points), _ip(idx), _fp(out))
    return out


def ref_group_point_grad(points, idx, grad_out):
    r = ref("grouping")
    points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
    b, n, c = points.shape
    _, m, k = idx.shape
    g = np.zeros((b, n, c), np.float32)
    r.ref_group_point_grad(b, n, c, m, k, _fp(grad_out), _ip(idx), _fp(g))
    return g


def ref_three_nn(xyz1, xyz2):
    r = ref("interpolate")
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    r.ref_three_nn(b, n, m, _fp(xyz1), _fp(xyz2), _fp(dist), _ip(idx))
    return dist, idx


def ref_three_interpolate(points, idx, weight):
    r = ref("interpolate")
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.zeros((b, n, c), np.float32)
    r.ref_three_interpolate(b, m, c, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def ref_three_interpolate_grad(points, idx, weight, grad_out):
    r = ref("interpolate")
    points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)
    r.ref_three_interpolate_grad(b, n, c, m, _fp(grad_out), _ip(idx), _fp(weight), _fp(g))
    return g
