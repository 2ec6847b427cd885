"""CPU: the oracle against what the REFERENCE's own device kernels computed on MI355X (tests/golden/ref_gpu_*.npz, written on the GPU
box by tests/golden/make_ref_gpu_golden.py from oracle/_ref/libref_{sampling,grouping}_gpu.so = tf_sampling_g.cu / tf_grouping_g.cu
compiled for gfx950 where they lie).  This is what pins the FPS, gather / scatter-add and ProbSample restatements, which have no CPU
twin in the reference tree; the live three-way comparison (reference kernel, product kernel, oracle) is tests/test_gpu_reference_kernels.py."""
import hashlib

import numpy as np
import pytest

import cases


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fixtures_say_where_they_come_from(golden):
    for f in ("ref_gpu_fps", "ref_gpu_gather", "ref_gpu_prob_sample", "ref_gpu_grouping", "ref_gpu_selection_sort"):
        assert str(golden(f)["source"]).startswith("ref-gpu")


@pytest.mark.parametrize("name", sorted(cases.fps_cases()))
def test_fps_small_cases(O, golden, name):
    """tf_sampling_g.cu:105-170: n below the 512 lanes, past the 3072-point LDS buffer, exact ties (duplicates, lattice), m > n."""
    xyz, m = cases.fps_cases()[name]
    ref = golden("ref_gpu_fps")[name]
    assert ref.shape == (xyz.shape[0], m)
    assert (O.farthest_point_sample(m, xyz) == ref).all()
    assert (O.farthest_point_sample(m, xyz, closed=True) == ref).all()  # the closed form of the tie rule, too
    assert (golden("fps_cases")[name] == ref).all()  # the older oracle-made fixture holds the same picks


@pytest.mark.parametrize("name", ["room_8x20480", "uniform_2x20480", "scan_1x80000"])
def test_fps_full_size_clouds(O, golden, name):
    """The headline's own clouds: every one of the 8 x 2048 picks of the sa1 level, the uniform cube, a config-5 scene."""
    xyz, m = cases.full_size_cases()[name]
    assert (O.farthest_point_sample(m, xyz) == golden("ref_gpu_fps")[name]).all()


def test_gather_point_and_its_gradient(O, golden):
    g = golden("ref_gpu_gather")
    xyz, _ = cases.fps_cases()["small_n300"]
    idx = golden("ref_gpu_fps")["small_n300"]
    assert (O.gather_point(xyz, idx) == g["out"]).all()
    # integer-valued cotangents: the reference's atomic adds are exact in any order, so the comparison is bit for bit
    assert (O.gather_point_grad(xyz, idx, g["cot"]) == g["grad"]).all()


def test_prob_sample_and_cumsum(O, golden):
    """tf_sampling_g.cu:7-104: the float running sum with the kernel's scan-tree association and its compensated chunk carry, then
    the binary search -- category counts across the 8192-element chunk and the 4-element group boundaries."""
    g = golden("ref_gpu_prob_sample")
    for name, (p, r) in cases.prob_sample_cases().items():
        assert (O.prob_sample(p, r) == g[name]).all(), name
        assert sha(O.cumsum(p)) == str(g[name + "_cumsum_sha"]), name


def test_ball_query_group_and_gradient_device_kernels(O, golden):
    """tf_grouping_g.cu:3-78 as device code (the CPU twin under tf_ops/grouping/test is pinned in test_oracle_golden.py)."""
    g = golden("ref_gpu_grouping")
    for name, c in (("optest", cases.grouping_optest()), ("demo", cases.grouping_demo())):
        idx, cnt = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
        assert sha(idx) == str(g[name + "_idx_sha"]) and (cnt == g[name + "_cnt"]).all()
        assert sha(O.group_point(c["points"], idx)) == str(g[name + "_out_sha"])
    c = cases.grouping_optest()
    idx, _ = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    assert (O.group_point_grad(c["points"], idx, g["optest_grad_cot"]) == g["optest_grad"]).all()


def test_ball_query_full_size_sa1(O, golden):
    """8 x 20480 candidates x 2048 centres, r = 0.2, K = 64 on the room scenes: pts_cnt and every index."""
    g = golden("ref_gpu_grouping")
    room, m = cases.full_size_cases()["room_8x20480"]
    centres = O.gather_point(room, golden("ref_gpu_fps")["room_8x20480"])
    prev = O.set_threads(8)
    try:
        idx, cnt = O.query_ball_point(0.2, 64, room, centres)
    finally:
        O.set_threads(prev)
    assert sha(idx) == str(g["sa1_idx_sha"]) and sha(cnt) == str(g["sa1_cnt_sha"])
    assert (idx[0, :4] == g["sa1_idx_head"]).all()


def test_selection_sort_device_kernel(O, golden):
    g = golden("ref_gpu_selection_sort")
    for name, (dist, k) in cases.selection_sort_cases().items():
        outi, val = O.select_top_k(k, dist)
        assert sha(outi[..., :k]) == str(g[name + "_idx_sha"]) and sha(val[..., :k]) == str(g[name + "_val_sha"]), name
