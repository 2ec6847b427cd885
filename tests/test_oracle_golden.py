"""CPU: the oracle restatement against the committed golden vectors and, where the reference's
own compiled code is available (oracle/_ref), against that code directly."""
import hashlib

import numpy as np
import pytest

import cases


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_grouping_optest_golden(O, golden):
    c, g = cases.grouping_optest(), golden("grouping_optest")
    idx, cnt = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    assert (idx == g["idx"]).all() and (cnt == g["pts_cnt"]).all()
    assert (O.group_point(c["points"], idx) == g["out"]).all()
    assert (O.group_point_grad(c["points"], idx, c["grad_out"]) == g["grad"]).all()


def test_grouping_demo_golden(O, golden):
    c, g = cases.grouping_demo(), golden("grouping_demo")
    idx, cnt = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    assert sha(idx) == str(g["idx_sha"]) and (cnt == g["pts_cnt"]).all()
    assert sha(O.group_point(c["points"], idx)) == str(g["out_sha"])


def test_cfg1_golden(O, golden):
    g = golden("cfg1")
    xyz = cases.cfg1_cloud()
    fidx = O.farthest_point_sample(512, xyz)
    assert (fidx == g["fps_idx"]).all()
    new_xyz = O.gather_point(xyz, fidx)
    idx, cnt = O.query_ball_point(0.2, 32, xyz, new_xyz)
    assert (idx == g["idx"]).all() and (cnt == g["pts_cnt"]).all()
    assert sha(O.group_point(xyz, idx)) == str(g["grouped_xyz_sha"])


def test_interpolate_optest_golden(O, golden):
    c, g = cases.interpolate_optest(), golden("interpolate_optest")
    dist, idx = O.three_nn(c["xyz1"], c["xyz2"])
    assert (dist == g["dist"]).all() and (idx == g["idx"]).all()
    w = np.full_like(dist, 1.0 / 3.0)
    assert (O.three_interpolate(c["points"], idx, w) == g["out"]).all()
    assert (O.three_interpolate_grad(c["points"], idx, w, c["grad_out"]) == g["grad"]).all()
    assert (O.three_nn_weights(dist) == g["weights_idw"]).all()


def test_interpolate_demo_golden(O, golden):
    c, g = cases.interpolate_demo(), golden("interpolate_demo")
    dist, idx = O.three_nn(c["xyz1"], c["xyz2"])
    assert sha(dist) == str(g["dist_sha"]) and sha(idx) == str(g["idx_sha"])
    w = np.full_like(dist, 1.0 / 3.0)
    assert sha(O.three_interpolate(c["points"], idx, w)) == str(g["out_sha"])


def test_nms_smoke_known_answer(O, golden):
    """The reference's own smoke input (tf_nms3d.py:21-46) and its recorded output."""
    c, g = cases.nms_smoke(), golden("nms_smoke")
    assert (O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.5) == g["keep_050"]).all()
    assert (O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25) == g["keep_025"]).all()
    assert abs(O.bev_intersection(c["bboxes"][0, 0], c["bboxes"][0, 1]) - float(g["bev_intersection"])) < 1e-6
    # volumes 1.0 / 0.512 -> iou = I3/(VA+VB-I3), I3 = 0.8 * 0.6227418
    i3 = 0.8 * float(g["bev_intersection"])
    assert abs(O.iou3d(c["bboxes"][0, 0], c["bboxes"][0, 1]) - i3 / (1.0 + 0.512 - i3)) < 1e-6


def test_nms_random_golden(O, golden):
    c, g = cases.nms_random(), golden("nms_random")
    for s in range(c["bboxes"].shape[0]):
        assert (O.iou3d_matrix(c["bboxes"][s]) == g["iou"][s]).all()
    assert (O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25) == g["keep_025"]).all()
    assert (O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.5) == g["keep_050"]).all()


def test_fps_golden(O, golden):
    g = golden("fps_cases")
    for name, (xyz, m) in cases.fps_cases().items():
        assert (O.farthest_point_sample(m, xyz) == g[name]).all(), name


# ---- directly against the reference's compiled code, when oracle/_ref exists ----
def _need_ref(O, name):
    if O.ref(name) is None:
        pytest.skip("oracle/_ref/libref_%s.so not built (reference tree not mounted)" % name)


@pytest.mark.parametrize("seed,b,n,m,r,k", [(1, 2, 700, 90, 0.2, 16), (2, 1, 2048, 512, 0.2, 32), (3, 3, 513, 64, 0.4, 64),
                                            (4, 1, 100, 7, 0.05, 8), (5, 2, 64, 64, 2.0, 5)])
def test_ball_query_vs_reference_build(O, seed, b, n, m, r, k):
    _need_ref(O, "grouping")
    rng = np.random.default_rng(seed)
    xyz1 = rng.random((b, n, 3), dtype=np.float32)
    xyz2 = rng.random((b, m, 3), dtype=np.float32)
    idx, cnt = O.query_ball_point(r, k, xyz1, xyz2)
    ref = O.ref_query_ball_point(r, k, xyz1, xyz2)
    # a query without any hit is left untouched by the reference (pre-zeroed here) and 0 in the oracle
    assert (idx == ref).all()
    pts = rng.random((b, n, 5), dtype=np.float32)
    assert (O.group_point(pts, idx) == O.ref_group_point(pts, idx)).all()
    go = rng.random((b, m, k, 5), dtype=np.float32)
    assert (O.group_point_grad(pts, idx, go) == O.ref_group_point_grad(pts, idx, go)).all()


@pytest.mark.parametrize("seed,b,n,m,c", [(1, 2, 300, 40, 8), (2, 1, 1024, 512, 3), (3, 1, 50, 2, 4), (4, 1, 50, 1, 4)])
def test_three_nn_vs_reference_build(O, seed, b, n, m, c):
    _need_ref(O, "interpolate")
    rng = np.random.default_rng(seed)
    xyz1 = rng.random((b, n, 3), dtype=np.float32)
    xyz2 = rng.random((b, m, 3), dtype=np.float32)
    if seed == 2:
        xyz2[:, 5] = xyz2[:, 9]  # duplicated known points: equal distances, lower index first
    dist, idx = O.three_nn(xyz1, xyz2)
    rd, ri = O.ref_three_nn(xyz1, xyz2)
    assert (idx == ri).all() and (dist == rd).all()  # includes the (inf, 0) fill when m < 3
    pts = rng.random((b, m, c), dtype=np.float32)
    w = O.three_nn_weights(np.where(np.isfinite(dist), dist, 1.0).astype(np.float32))
    assert (O.three_interpolate(pts, idx, w) == O.ref_three_interpolate(pts, idx, w)).all()
    go = rng.random((b, n, c), dtype=np.float32)
    assert (O.three_interpolate_grad(pts, idx, w, go) == O.ref_three_interpolate_grad(pts, idx, w, go)).all()
