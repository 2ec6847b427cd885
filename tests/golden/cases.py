"""Seeded input generators shared by make_golden.py and the tests.

The shapes, seeds and value ranges are those of the reference's own smoke / test programs:
  tf_ops/grouping/tf_grouping_op_test.py:11-16     (1,128,16) points, (1,128,3)/(1,8,3) xyz, r=0.3 K=32
  tf_ops/grouping/tf_grouping.py:79-88             seed 100, (32,512,64)/(32,512,3)/(32,128,3), r=0.1 K=64
  tf_ops/3d_interpolation/tf_interpolate_op_test.py:11-16   (1,8,16) points, (1,128,3)/(1,8,3) xyz, w=1/3
  tf_ops/3d_interpolation/tf_interpolate.py:39-42  seed 100, (32,128,64)/(32,512,3)/(32,128,3)
  tf_ops/3d_nms/tf_nms3d.py:21-46                  two boxes, scores .5/.6, thr 0.5
plus BASELINE.json config 1 (2048-pt random cloud, FPS->512, r=0.2 K=32, seed 0).
The reference's two op tests draw unseeded np.random.random; a fixed legacy seed is used here.
"""
import numpy as np


def grouping_optest():
    rs = np.random.RandomState(1234)
    points = rs.random_sample((1, 128, 16)).astype("float32")
    xyz1 = rs.random_sample((1, 128, 3)).astype("float32")
    xyz2 = rs.random_sample((1, 8, 3)).astype("float32")
    grad_out = rs.random_sample((1, 8, 32, 16)).astype("float32")
    return dict(points=points, xyz1=xyz1, xyz2=xyz2, grad_out=grad_out, radius=0.3, nsample=32)


def grouping_demo():
    rs = np.random.RandomState(100)  # np.random.seed(100), tf_grouping.py:79
    pts = rs.random_sample((32, 512, 64)).astype("float32")
    tmp1 = rs.random_sample((32, 512, 3)).astype("float32")
    tmp2 = rs.random_sample((32, 128, 3)).astype("float32")
    return dict(points=pts, xyz1=tmp1, xyz2=tmp2, radius=0.1, nsample=64)


def interpolate_optest():
    rs = np.random.RandomState(4321)
    points = rs.random_sample((1, 8, 16)).astype("float32")
    xyz1 = rs.random_sample((1, 128, 3)).astype("float32")
    xyz2 = rs.random_sample((1, 8, 3)).astype("float32")
    grad_out = rs.random_sample((1, 128, 16)).astype("float32")
    return dict(points=points, xyz1=xyz1, xyz2=xyz2, grad_out=grad_out)


def interpolate_demo():
    rs = np.random.RandomState(100)  # tf_interpolate.py:39
    pts = rs.random_sample((32, 128, 64)).astype("float32")
    tmp1 = rs.random_sample((32, 512, 3)).astype("float32")
    tmp2 = rs.random_sample((32, 128, 3)).astype("float32")
    return dict(points=pts, xyz1=tmp1, xyz2=tmp2)


def cfg1_cloud(b=1, n=2048, seed=0):
    """BASELINE.json config 1: xyz ~ U[0,1)^3."""
    return np.random.default_rng(seed).random((b, n, 3), dtype=np.float32)


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def corner_box(l, w, h, angle=None, center=(0, 0, 0)):
    """Corner order of tf_nms3d.py:21-28 / model.py:108-110 (first four = top face, y=+h/2)."""
    x = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    y = [h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2]
    z = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    c = np.vstack([x, y, z])
    if angle:
        c = roty(angle) @ c
    return np.transpose(c) + np.asarray(center, dtype=np.float64)[None, :]


def nms_smoke():
    bboxes = np.array([[corner_box(1, 1, 1), corner_box(0.8, 0.8, 0.8, np.pi / 4 * 3)]]).astype("float32")
    scores = np.array([[0.5, 0.6]]).astype("float32")
    objectiveness = np.array([[[0.3, 0.7], [0.4, 0.6]]]).astype("float32")
    return dict(bboxes=bboxes, scores=scores, objectiveness=objectiveness)


def nms_random(b=2, n=128, seed=7, room=4.0):
    """Random upright oriented boxes crowded enough to overlap; distinct scores (no ties)."""
    rng = np.random.default_rng(seed)
    boxes = np.zeros((b, n, 8, 3), np.float32)
    for s in range(b):
        for i in range(n):
            lwh = rng.uniform(0.3, 1.5, 3)
            ang = rng.uniform(0, 2 * np.pi)
            ctr = (rng.uniform(0, room), rng.uniform(0, 1.0), rng.uniform(0, room))
            boxes[s, i] = corner_box(lwh[0], lwh[1], lwh[2], ang, ctr).astype("float32")
    scores = rng.permutation(b * n).reshape(b, n).astype("float32") / (b * n)
    obj = rng.normal(size=(b, n, 2)).astype("float32")
    return dict(bboxes=boxes, scores=scores, objectiveness=obj)


def fps_cases():
    rng = np.random.default_rng(11)
    cases = {}
    cases["cfg1"] = (cfg1_cloud(), 512)
    cases["small_n300"] = (rng.random((2, 300, 3), dtype=np.float32), 64)            # n < 512 lanes
    cases["n5000"] = (rng.random((1, 5000, 3), dtype=np.float32) * 5, 128)            # n > 3072 (past the reference's smem buffer)
    dup = rng.random((1, 700, 3), dtype=np.float32)
    dup = np.concatenate([dup, dup[:, ::-1]], axis=1)                                 # every point twice -> exact ties
    cases["duplicates"] = (dup, 900)
    grid = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(10), indexing="ij"), -1).reshape(1, -1, 3)
    cases["lattice"] = (grid.astype(np.float32), 200)                                 # many exact distance ties
    cases["exhaust"] = (rng.random((1, 40, 3), dtype=np.float32), 64)                 # m > n: index 0 repeats
    return cases


def full_size_cases():
    """The clouds of BASELINE.json's configs for the fixtures computed by the reference's device kernels (make_ref_gpu_golden.py):
    config 2/3 (8 x 20480 room scenes -> 2048), the uniform-cube variant, one scene of config 5 (80000 -> 2048).  (cloud, m) by name."""
    from votenet_amd import synth
    return {
        "room_8x20480": (synth.room_batch(8, 20480, seed0=1000), 2048),
        "uniform_2x20480": (synth.uniform_batch(2, 20480, seed0=1000), 2048),
        "scan_1x80000": (synth.room_batch(1, 80000, seed0=5000, size=(8.0, 3.0, 8.0), nbox=(15, 25)), 2048),
    }


def selection_sort_cases():
    """SelectionSort inputs: the reference twin's own main() (test/selection_sort.cpp:66-77: b=2,n=4,m=2,k=3,
    dist[i] = 10-i), the kNN demo of tf_grouping.py:75-90 (squared distances of the demo clouds, k = 64; first 4 scenes),
    and rows with many exact ties (the swap order decides which tied index comes first)."""
    out = {"twin_main": (np.array([10.0 - i for i in range(16)], np.float32).reshape(2, 2, 4), 3)}
    c = grouping_demo()
    a, q = c["xyz1"][:4], c["xyz2"][:4]
    d = np.zeros((4, 128, 512), np.float32)
    for ch in range(3):  # left-to-right channel sum in float32 (oracle_knn_dist)
        t = (a[:, None, :, ch] - q[:, :, None, ch]).astype(np.float32)
        d = (t * t).astype(np.float32) if ch == 0 else (d + t * t).astype(np.float32)
    out["knn_demo"] = (d, 64)
    rs = np.random.RandomState(5)
    out["ties"] = (rs.randint(0, 6, (3, 7, 50)).astype(np.float32), 20)
    out["k_equals_n"] = (rs.random_sample((1, 3, 33)).astype(np.float32), 33)
    return out


def prob_sample_cases():
    """ProbSample inputs: the triangle-area demo of tf_sampling.py:60-72 (5 categories, 8192 draws; the draws come from
    tf.random_uniform there, numpy here), and category counts that cross the kernel's 8192-element chunk and 4-element
    group boundaries."""
    rs = np.random.RandomState(100)  # np.random.seed(100), tf_sampling.py:62
    tri = rs.rand(1, 5, 3, 3).astype("float32")
    ab, ac = tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0]
    areas = np.sqrt((np.cross(ab, ac) ** 2).sum(2) + 1e-9).astype(np.float32)
    out = {"triangles": (areas, rs.rand(1, 8192).astype(np.float32))}
    for n in (1, 2, 3, 4, 5, 7, 8, 1000, 8191, 8192, 8193, 8195, 20000):
        out["n%d" % n] = (rs.rand(2, n).astype(np.float32) + 1e-3, rs.rand(2, 257).astype(np.float32))
    return out
