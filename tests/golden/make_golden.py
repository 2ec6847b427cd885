"""Generate tests/golden/*.npz.  Run in the build container (reference tree mounted):

    python tests/golden/make_golden.py

Expected outputs come from the REFERENCE's own compiled code (oracle/_ref, built by
oracle/Makefile from tf_ops/grouping/test/query_ball_point.cpp, tf_ops/grouping/test/selection_sort.cpp and
tf_ops/3d_interpolation/interpolate.cpp) wherever it exists, and are tagged source="ref".
Where the reference cannot run here (FPS / ProbSample: device kernels, no GPU in this container; NMS: needs TensorFlow
headers) they come from the oracle restatement and are tagged source="oracle" -- plus the known answer of the reference's
NMS smoke input recorded in SURVEY.md section 4.  The device kernels' own outputs on MI355X are separate fixtures
(ref_gpu_*.npz, written on the GPU box by make_ref_gpu_golden.py) that the oracle-made ones are tested against.
Large tensors are stored as sha256 digests of their bytes (integer / pure-copy results are
bit-exact, so a digest is a complete check).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    O.build()
    assert O.ref("grouping") is not None and O.ref("interpolate") is not None, "oracle/_ref missing (no reference tree?)"

    # ---- grouping, reference test shape (tf_grouping_op_test.py)
    c = cases.grouping_optest()
    idx = O.ref_query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    _, cnt = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    out = O.ref_group_point(c["points"], idx)
    grad = O.ref_group_point_grad(c["points"], idx, c["grad_out"])
    np.savez_compressed(os.path.join(HERE, "grouping_optest.npz"), idx=idx, pts_cnt=cnt, out=out, grad=grad,
                        source="ref (pts_cnt: oracle)")

    # ---- grouping, demo inputs (tf_grouping.py:79-88), ball-query branch
    c = cases.grouping_demo()
    idx = O.ref_query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    _, cnt = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    out = O.ref_group_point(c["points"], idx)
    np.savez_compressed(os.path.join(HERE, "grouping_demo.npz"), idx_sha=sha(idx), out_sha=sha(out), pts_cnt=cnt,
                        idx_head=idx[:2], source="ref (pts_cnt: oracle)")

    # ---- BASELINE config 1: 2048 pts -> FPS 512 -> ball r=0.2 K=32
    xyz = cases.cfg1_cloud()
    fidx = O.farthest_point_sample(512, xyz)
    new_xyz = O.gather_point(xyz, fidx)
    idx = O.ref_query_ball_point(0.2, 32, xyz, new_xyz)
    _, cnt = O.query_ball_point(0.2, 32, xyz, new_xyz)
    gx = O.ref_group_point(xyz, idx)
    np.savez_compressed(os.path.join(HERE, "cfg1.npz"), fps_idx=fidx, idx=idx, pts_cnt=cnt, grouped_xyz_sha=sha(gx),
                        source="ball query/group: ref; FPS, pts_cnt: oracle")

    # ---- interpolation, reference test shape (tf_interpolate_op_test.py)
    c = cases.interpolate_optest()
    dist, idx = O.ref_three_nn(c["xyz1"], c["xyz2"])
    w = np.full_like(dist, 1.0 / 3.0)  # tf.ones_like(dist)/3.0
    out = O.ref_three_interpolate(c["points"], idx, w)
    grad = O.ref_three_interpolate_grad(c["points"], idx, w, c["grad_out"])
    np.savez_compressed(os.path.join(HERE, "interpolate_optest.npz"), dist=dist, idx=idx, out=out, grad=grad,
                        weights_idw=O.three_nn_weights(dist), source="ref (weights_idw: oracle, utils.py:279-282)")

    # ---- interpolation, demo inputs (tf_interpolate.py:39-49)
    c = cases.interpolate_demo()
    dist, idx = O.ref_three_nn(c["xyz1"], c["xyz2"])
    w = np.full_like(dist, 1.0 / 3.0)
    out = O.ref_three_interpolate(c["points"], idx, w)
    np.savez_compressed(os.path.join(HERE, "interpolate_demo.npz"), dist_sha=sha(dist), idx_sha=sha(idx), out_sha=sha(out),
                        idx_head=idx[0, :16], dist_head=dist[0, :16], source="ref")

    # ---- NMS smoke (tf_nms3d.py:21-46).  Known answer recorded in SURVEY.md section 4 from the
    # reference itself: thr 0.5 -> [[0,1],[0,0]], thr 0.25 -> [[0,1]], BEV intersection 0.6227418
    c = cases.nms_smoke()
    np.savez_compressed(os.path.join(HERE, "nms_smoke.npz"), keep_050=np.array([[0, 1], [0, 0]], np.int32),
                        keep_025=np.array([[0, 1]], np.int32), bev_intersection=np.float32(0.6227418),
                        volumes=np.array([1.0, 0.512], np.float32), source="reference known answer (SURVEY.md 4)")
    assert O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.5).tolist() == [[0, 1], [0, 0]]
    assert O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25).tolist() == [[0, 1]]
    assert abs(O.bev_intersection(c["bboxes"][0, 0], c["bboxes"][0, 1]) - 0.6227418) < 1e-6

    # ---- NMS random boxes (oracle)
    c = cases.nms_random()
    iou = np.stack([O.iou3d_matrix(c["bboxes"][s]) for s in range(c["bboxes"].shape[0])])
    np.savez_compressed(os.path.join(HERE, "nms_random.npz"), iou=iou,
                        keep_025=O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25),
                        keep_050=O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.5), source="oracle")

    # ---- FPS cases (oracle; the same cases computed by the reference kernel itself: ref_gpu_fps.npz, make_ref_gpu_golden.py)
    out = {}
    for name, (xyz, m) in cases.fps_cases().items():
        a = O.farthest_point_sample(m, xyz)
        assert (a == O.farthest_point_sample(m, xyz, closed=True)).all(), name
        out[name] = a
    np.savez_compressed(os.path.join(HERE, "fps_cases.npz"), source="oracle", **out)
    # ---- SelectionSort (reference CPU twin, tf_ops/grouping/test/selection_sort.cpp)
    assert O.ref("selection_sort") is not None
    out = {}
    for name, (dist, k) in cases.selection_sort_cases().items():
        outi, val = O.ref_select_top_k(k, dist)
        if dist.size <= 4096:
            out[name + "_idx"], out[name + "_val"] = outi, val
        else:
            out[name + "_idx_sha"], out[name + "_val_sha"], out[name + "_idx_head"] = sha(outi), sha(val), outi[0, :2, :k]
    np.savez_compressed(os.path.join(HERE, "selection_sort.npz"), source="ref", **out)

    # ---- ProbSample (oracle; the same cases computed by the reference kernels themselves: ref_gpu_prob_sample.npz)
    out = {}
    for name, (p, r) in cases.prob_sample_cases().items():
        out[name] = O.prob_sample(p, r)
        out[name + "_cumsum_sha"] = sha(O.cumsum(p))
    np.savez_compressed(os.path.join(HERE, "prob_sample.npz"), source="oracle", **out)
    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  %-28s %7d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
