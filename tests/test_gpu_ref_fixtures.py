"""GPU: the PRODUCT against what the reference's own device kernels computed on MI355X, directly -- tests/golden/ref_gpu_*.npz
(written by tests/golden/make_ref_gpu_golden.py from tf_sampling_g.cu / tf_grouping_g.cu compiled for gfx950 where they lie).
No oracle/_ref library and no oracle in between: these comparisons run on any snapshot, with or without the reference tree.

Half of the tests go through the Python mirror (votenet_amd.tf_sampling / tf_grouping -> the C ABI); the other half drive the eight
launcher names the reference's wrappers declare (tf_sampling.cpp:65,94,125,150, tf_grouping.cpp:66,108,142,173) through
tests/link/launcher_link.cpp, linked against the product with -Wl,-z,defs (conftest.linklib): the drop-in seam itself."""
import ctypes
import hashlib

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def P(t):
    return ctypes.c_void_p(t.data_ptr())


@pytest.fixture(scope="module")
def ops(hiplib):
    from votenet_amd import tf_grouping, tf_sampling

    class Ops:
        pass
    o = Ops()
    o.s, o.g = tf_sampling, tf_grouping
    return o


# ---------------------------------------------------------------- through the Python mirror / C ABI
@pytest.mark.parametrize("name", sorted(cases.fps_cases()))
def test_fps_small_cases_are_the_reference_kernels_picks(ops, dev, golden, name):
    xyz, m = cases.fps_cases()[name]
    assert (N(ops.s.farthest_point_sample(m, T(xyz, dev))) == golden("ref_gpu_fps")[name]).all()


@pytest.mark.parametrize("name", ["room_8x20480", "uniform_2x20480", "scan_1x80000"])
def test_fps_full_size_clouds_are_the_reference_kernels_picks(ops, dev, golden, name):
    """Every pick of the headline launch (8 x 20480 -> 2048), the uniform cube and a config-5 scene (80000 -> 2048)."""
    xyz, m = cases.full_size_cases()[name]
    got = N(ops.s.farthest_point_sample(m, T(xyz, dev)))
    ref = golden("ref_gpu_fps")[name]
    assert got.shape == ref.shape and (got == ref).all(), "%d picks differ" % int((got != ref).sum())


def test_gather_and_scatter_add_are_the_reference_kernels(ops, dev, golden):
    g = golden("ref_gpu_gather")
    xyz, _ = cases.fps_cases()["small_n300"]
    idx = golden("ref_gpu_fps")["small_n300"]
    assert (N(ops.s.gather_point(T(xyz, dev), T(idx, dev))) == g["out"]).all()
    grad = ops.s.gather_point_grad_raw(xyz.shape[1], T(idx, dev), T(g["cot"], dev))  # integer cotangents: exact in any order
    assert (N(grad) == g["grad"]).all()


def test_prob_sample_is_the_reference_kernel(ops, dev, golden):
    g = golden("ref_gpu_prob_sample")
    for name, (p, r) in cases.prob_sample_cases().items():
        assert (N(ops.s.prob_sample(T(p, dev), T(r, dev))) == g[name]).all(), name


def test_ball_query_group_and_grad_are_the_reference_kernels(ops, dev, golden):
    g = golden("ref_gpu_grouping")
    for name, c in (("optest", cases.grouping_optest()), ("demo", cases.grouping_demo())):
        idx, cnt = ops.g.query_ball_point(c["radius"], c["nsample"], T(c["xyz1"], dev), T(c["xyz2"], dev))
        assert sha(N(idx)) == str(g[name + "_idx_sha"]) and (N(cnt) == g[name + "_cnt"]).all(), name
        assert sha(N(ops.g.group_point(T(c["points"], dev), idx))) == str(g[name + "_out_sha"]), name
    c = cases.grouping_optest()
    idx, _ = ops.g.query_ball_point(c["radius"], c["nsample"], T(c["xyz1"], dev), T(c["xyz2"], dev))
    grad = ops.g.group_point_grad_raw(128, idx, T(g["optest_grad_cot"], dev))
    assert (N(grad) == g["optest_grad"]).all()


def test_ball_query_sa1_full_size_is_the_reference_kernel(ops, dev, golden):
    """8 x 20480 candidates x 2048 centres, r = 0.2, K = 64 on the room scenes: digests of every index and count."""
    g = golden("ref_gpu_grouping")
    room, _ = cases.full_size_cases()["room_8x20480"]
    x = T(room, dev)
    centres = ops.s.gather_point(x, T(golden("ref_gpu_fps")["room_8x20480"], dev))
    idx, cnt = ops.g.query_ball_point(0.2, 64, x, centres)
    idx, cnt = N(idx), N(cnt)
    assert (idx[0, :4] == g["sa1_idx_head"]).all()
    assert sha(idx) == str(g["sa1_idx_sha"]) and sha(cnt) == str(g["sa1_cnt_sha"])


def test_selection_sort_is_the_reference_kernel(ops, dev, golden):
    g = golden("ref_gpu_selection_sort")
    for name, (dist, k) in cases.selection_sort_cases().items():
        outi, val = ops.g.select_top_k(k, T(dist, dev))
        assert sha(N(outi)[..., :k]) == str(g[name + "_idx_sha"]) and sha(N(val)[..., :k]) == str(g[name + "_val_sha"]), name


# ---------------------------------------------------------------- through the eight launcher names (the drop-in seam)
def test_launchers_prob_sample_and_selection_sort_through_the_link(linklib, dev, golden):
    """probsampleLauncher (tf_sampling.cpp:65, called :89 with a (b,n) float temp) and selectionSortLauncher (tf_grouping.cpp:108,
    called :134 with (b,m,n) outputs) -- null stream, as the reference."""
    g = golden("ref_gpu_prob_sample")
    for name, (p, r) in cases.prob_sample_cases().items():
        b, n = p.shape
        m = r.shape[1]
        dp, dr = T(p, dev), T(r, dev)
        temp = torch.empty((b, n), dtype=torch.float32, device=dev)
        out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        linklib.link_prob_sample(b, n, m, P(dp), P(dr), P(temp), P(out))
        torch.cuda.synchronize()
        assert (N(out) == g[name]).all(), name
    g = golden("ref_gpu_selection_sort")
    for name, (dist, k) in cases.selection_sort_cases().items():
        b, m, n = dist.shape
        d = T(dist, dev)
        outi = torch.full((b, m, n), -1, dtype=torch.int32, device=dev)
        val = torch.full((b, m, n), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        linklib.link_selection_sort(b, n, m, k, P(d), P(outi), P(val))
        torch.cuda.synchronize()
        assert sha(N(outi)[..., :k]) == str(g[name + "_idx_sha"]) and sha(N(val)[..., :k]) == str(g[name + "_val_sha"]), name


def test_launchers_sampling_through_the_link(linklib, dev, golden):
    """farthestpointsamplingLauncher with the reference's own 32*n-float temp (tf_sampling.cpp:115), gatherpointLauncher,
    scatteraddpointLauncher into a caller-zeroed buffer (tf_sampling.cpp:174)."""
    for name in ("cfg1", "n5000", "duplicates"):
        xyz, m = cases.fps_cases()[name]
        b, n, _ = xyz.shape
        x = T(xyz, dev)
        temp = torch.empty((32, n), dtype=torch.float32, device=dev)
        out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        linklib.link_fps(b, n, m, P(x), P(temp), P(out))
        torch.cuda.synchronize()
        assert (N(out) == golden("ref_gpu_fps")[name]).all(), name
    room, m = cases.full_size_cases()["uniform_2x20480"]
    x = T(room, dev)
    temp = torch.empty((32, 20480), dtype=torch.float32, device=dev)
    out = torch.full((2, m), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    linklib.link_fps(2, 20480, m, P(x), P(temp), P(out))
    torch.cuda.synchronize()
    assert (N(out) == golden("ref_gpu_fps")["uniform_2x20480"]).all()
    g = golden("ref_gpu_gather")
    xyz, _ = cases.fps_cases()["small_n300"]
    idx = golden("ref_gpu_fps")["small_n300"]
    b, n, _ = xyz.shape
    m = idx.shape[1]
    x, di = T(xyz, dev), T(idx, dev)
    o = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
    cot = T(g["cot"], dev)
    grad = torch.zeros((b, n, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    linklib.link_gather(b, n, m, P(x), P(di), P(o))
    linklib.link_scatter_add(b, n, m, P(cot), P(di), P(grad))
    torch.cuda.synchronize()
    assert (N(o) == g["out"]).all() and (N(grad) == g["grad"]).all()


def test_launchers_grouping_through_the_link(linklib, dev, golden):
    """queryBallPointLauncher, groupPointLauncher, groupPointGradLauncher (caller-zeroed, tf_grouping.cpp:204)."""
    g = golden("ref_gpu_grouping")
    for name, c in (("optest", cases.grouping_optest()), ("demo", cases.grouping_demo())):
        b, n, ch = c["points"].shape
        m, k = c["xyz2"].shape[1], c["nsample"]
        x1, x2, pts = T(c["xyz1"], dev), T(c["xyz2"], dev), T(c["points"], dev)
        idx = torch.full((b, m, k), -1, dtype=torch.int32, device=dev)
        cnt = torch.full((b, m), -1, dtype=torch.int32, device=dev)
        out = torch.empty((b, m, k, ch), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        linklib.link_query_ball(b, n, m, ctypes.c_float(c["radius"]), k, P(x1), P(x2), P(idx), P(cnt))
        linklib.link_group(b, n, ch, m, k, P(pts), P(idx), P(out))
        torch.cuda.synchronize()
        assert sha(N(idx)) == str(g[name + "_idx_sha"]) and (N(cnt) == g[name + "_cnt"]).all(), name
        assert sha(N(out)) == str(g[name + "_out_sha"]), name
        if name == "optest":
            cot = T(g["optest_grad_cot"], dev)
            grad = torch.zeros((b, n, ch), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            linklib.link_group_grad(b, n, ch, m, k, P(cot), P(idx), P(grad))
            torch.cuda.synchronize()
            assert (N(grad) == g["optest_grad"]).all()
