"""CPU: properties that pin the parts of the oracle no reference build or test reaches
(FPS: CUDA only; NMS geometry: needs TensorFlow headers) against independent computations."""
import numpy as np

import cases


def _fps_numpy(xyz, m):
    """Independent float32 FPS with the reference rule: max d2, ties -> min (k%512), then min k."""
    n = xyz.shape[0]
    td = np.full(n, np.float32(1e38), np.float32)
    out = np.zeros(m, np.int32)
    lane = np.arange(n) % 512
    old = 0
    for j in range(1, m):
        d = xyz - xyz[old]
        d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        td = np.minimum(d, td)
        best = td.max()
        cand = np.nonzero(td == best)[0]
        cand = cand[lane[cand] == lane[cand].min()]
        old = int(cand.min())
        out[j] = old
    return out


def test_fps_matches_independent_numpy(O):
    for name, (xyz, m) in cases.fps_cases().items():
        got = O.farthest_point_sample(m, xyz)
        for s in range(xyz.shape[0]):
            assert (got[s] == _fps_numpy(xyz[s], m)).all(), name


def test_fps_literal_equals_closed_form(O):
    rng = np.random.default_rng(5)
    for n, m in [(1, 3), (2, 2), (511, 100), (512, 100), (513, 100), (1500, 1500), (4097, 50)]:
        xyz = np.round(rng.random((2, n, 3), dtype=np.float32) * 4) / 4  # coarse grid: many exact ties
        assert (O.farthest_point_sample(m, xyz) == O.farthest_point_sample(m, xyz, closed=True)).all(), (n, m)


def test_fps_properties(O):
    xyz = cases.cfg1_cloud(b=2, n=1000, seed=3)
    idx = O.farthest_point_sample(200, xyz)
    assert (idx[:, 0] == 0).all()
    for s in range(2):
        assert len(set(idx[s].tolist())) == 200  # distinct until exhaustion
        # each pick attains the max of the running min-distance
        td = np.full(1000, np.inf)
        for j in range(1, 200):
            d = ((xyz[s].astype(np.float64) - xyz[s, idx[s, j - 1]]) ** 2).sum(1)
            td = np.minimum(td, d)
            assert td[idx[s, j]] >= td.max() * (1 - 1e-5)


def test_fps_tie_rule_on_crafted_ties(O):
    # four points at the same distance from point 0: indices 1, 513, 514, 2 -> lanes 1, 1, 2, 2.
    n = 600
    xyz = np.zeros((1, n, 3), np.float32)
    xyz[0, :, 0] = np.linspace(0, 1e-3, n)  # everything else close to the origin
    for k in (1, 513, 514, 2):
        xyz[0, k] = 0
    xyz[0, 1] = (0, 2, 0)
    xyz[0, 513] = (0, -2, 0)
    xyz[0, 514] = (2, 0, 0)
    xyz[0, 2] = (-2, 0, 0)
    idx = O.farthest_point_sample(2, xyz)
    assert idx[0, 1] == 1  # max distance tie: smallest (k mod 512) is lane 1 -> k in {1, 513} -> smallest k
    xyz[0, 1] = 0
    idx = O.farthest_point_sample(2, xyz)
    assert idx[0, 1] == 513  # lane 1 (k=513) beats lane 2 (k=2, k=514) although 2 < 513


def test_ball_query_properties(O):
    rng = np.random.default_rng(2)
    xyz1 = rng.random((2, 800, 3), dtype=np.float32)
    xyz2 = xyz1[:, :100].copy()
    r, k = np.float32(0.15), 16
    idx, cnt = O.query_ball_point(r, k, xyz1, xyz2)
    for s in range(2):
        d = np.sqrt(((xyz2[s][:, None, :].astype(np.float32) - xyz1[s][None]) ** 2).sum(-1, dtype=np.float32))
        for j in range(100):
            hits = np.nonzero(np.maximum(d[j], np.float32(1e-20)) < r)[0]
            # borderline pairs may differ by rounding of the numpy expression; compare away from the boundary
            safe = np.abs(d[j] - r) > 1e-5
            if safe.all():
                exp = hits[:k]
                assert cnt[s, j] == len(exp)
                assert (idx[s, j, :len(exp)] == exp).all()
                assert (idx[s, j, len(exp):] == exp[0]).all()
    assert (cnt >= 1).all()  # queries are points of the cloud: d = 1e-20 < r


def test_three_nn_against_sort(O):
    rng = np.random.default_rng(4)
    xyz1 = rng.random((2, 200, 3), dtype=np.float32)
    xyz2 = rng.random((2, 37, 3), dtype=np.float32)
    dist, idx = O.three_nn(xyz1, xyz2)
    for s in range(2):
        df = xyz2[s][None, :, :] - xyz1[s][:, None, :]
        d = (df[..., 0] * df[..., 0] + df[..., 1] * df[..., 1]) + df[..., 2] * df[..., 2]
        order = np.argsort(d, axis=1, kind="stable")[:, :3]
        assert (idx[s] == order).all()
        assert (dist[s] == np.take_along_axis(d, order, 1)).all()


def _clip_area(p, q):
    """Independent float64 Sutherland-Hodgman area of convex quad p clipped by convex quad q."""
    def area(poly):
        x, z = poly[:, 0], poly[:, 1]
        return 0.5 * abs(np.dot(x, np.roll(z, -1)) - np.dot(z, np.roll(x, -1)))

    def ccw(poly):
        x, z = poly[:, 0], poly[:, 1]
        return poly if (np.dot(x, np.roll(z, -1)) - np.dot(z, np.roll(x, -1))) > 0 else poly[::-1]
    out = ccw(p.astype(np.float64))
    q = ccw(q.astype(np.float64))
    for i in range(4):
        a, b = q[i], q[(i + 1) % 4]
        inp, out = out, []
        if len(inp) == 0:
            return 0.0
        for j in range(len(inp)):
            c, d = inp[j], inp[(j + 1) % len(inp)]
            sc = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
            sd = (b[0] - a[0]) * (d[1] - a[1]) - (b[1] - a[1]) * (d[0] - a[0])
            if sc >= 0:
                out.append(c)
            if (sc >= 0) != (sd >= 0):
                t = sc / (sc - sd)
                out.append(c + t * (d - c))
        out = np.array(out)
    return area(out) if len(out) >= 3 else 0.0


def test_iou_against_independent_clipping(O):
    c = cases.nms_random(b=1, n=48, seed=21)
    boxes = c["bboxes"][0]
    iou = O.iou3d_matrix(boxes)
    for i in range(48):
        for j in range(48):
            if i == j:
                continue
            a2 = _clip_area(boxes[i, :4][:, [0, 2]], boxes[j, :4][:, [0, 2]])
            h = max(min(boxes[i, 0, 1], boxes[j, 0, 1]) - max(boxes[i, 4, 1], boxes[j, 4, 1]), 0.0)

            def vol(b):
                e1 = np.linalg.norm(b[0, [0, 2]].astype(np.float64) - b[1, [0, 2]])
                e2 = np.linalg.norm(b[1, [0, 2]].astype(np.float64) - b[2, [0, 2]])
                return e1 * e2 * (float(b[0, 1]) - float(b[4, 1]))
            i3 = a2 * h
            exp = i3 / (vol(boxes[i]) + vol(boxes[j]) - i3)
            assert abs(iou[i, j] - exp) < 2e-5, (i, j, iou[i, j], exp)


def test_nms_semantics(O):
    c = cases.nms_random(b=3, n=40, seed=9)
    keep = O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25)
    flat = c["scores"][keep[:, 0], keep[:, 1]]
    assert (np.diff(flat) < 0).all()  # descending score over the whole batch
    obj = c["objectiveness"]
    assert (obj[keep[:, 0], keep[:, 1], 1] > obj[keep[:, 0], keep[:, 1], 0]).all()
    for s in range(3):
        ks = keep[keep[:, 0] == s][:, 1]
        iou = O.iou3d_matrix(c["bboxes"][s])
        for a in range(len(ks)):
            for b2 in range(a):
                assert not iou[ks[a], ks[b2]] > 0.25  # kept boxes of one scene do not overlap above thr
    # threshold 1.0 keeps every candidate; empty candidate set gives an empty result
    assert len(O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 1.0)) == int((obj[..., 1] > obj[..., 0]).sum())
    none = np.zeros_like(obj)
    assert O.nms3d(c["bboxes"], c["scores"], none, 0.25).shape == (0, 2)


def test_mlp_oracle_against_float64(O):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(640, 19)).astype(np.float32)
    w = rng.normal(size=(19, 24)).astype(np.float32)
    b = rng.normal(size=24).astype(np.float32)
    z = O.linear(x, w, b)
    assert np.allclose(z, x.astype(np.float64) @ w + b, rtol=1e-5, atol=1e-5)
    mean, var = O.bn_stats(z)
    assert np.allclose(mean, z.astype(np.float64).mean(0), rtol=1e-6, atol=1e-6)
    assert np.allclose(var, z.astype(np.float64).var(0), rtol=1e-6, atol=1e-6)
    g, be = rng.normal(size=24).astype(np.float32), rng.normal(size=24).astype(np.float32)
    y = O.bn_relu(z, mean, var, g, be)
    exp = np.maximum(0, g * (z - mean) / np.sqrt(var + 1e-5) + be)
    assert np.allclose(y, exp, rtol=1e-5, atol=1e-5)
    assert (O.max_over_k(y, 64) == y.reshape(10, 64, 24).max(1)).all()


def test_iou_and_nms_closed_form_known_answers(O):
    """Known answers that come from geometry, not from any code: they pin the restatement of tf_nms3d.cpp:43-273 independently of
    how the reference could (not) be compiled here.
    (a) The reference's own smoke input (tf_nms3d.py:21-46): a unit cube and a 0.8-cube rotated by 3*pi/4 about y, same
        centre.  The rotated footprint's half-diagonal 0.4*sqrt(2) overshoots the unit square by d = 0.4*sqrt(2) - 0.5 at each
        of its four corners, and each overshoot is a right isosceles triangle of area d^2:
            BEV intersection = 0.64 - 4 d^2 = 0.6227417...,  I3 = 0.8 * that,  IoU = I3 / (1 + 0.512 - I3) = 0.49143
        -> with thr 0.5 nothing is suppressed (visit order: scores 0.6 then 0.5 -> [[0,1],[0,0]]), with thr 0.25 the unit cube is.
    (b) Two axis-aligned unit cubes shifted by t along x: IoU = (1 - t) / (1 + t).
    (c) A unit cube and the same cube rotated by pi/4 about y (regular octagon): BEV intersection = 2 (sqrt(2) - 1)."""
    c = cases.nms_smoke()
    d = 0.4 * np.sqrt(2.0) - 0.5
    bev = 0.64 - 4 * d * d
    assert abs(O.bev_intersection(c["bboxes"][0, 0], c["bboxes"][0, 1]) - bev) < 1e-6
    i3 = 0.8 * bev
    assert abs(O.iou3d(c["bboxes"][0, 0], c["bboxes"][0, 1]) - i3 / (1 + 0.512 - i3)) < 1e-6
    assert O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.5).tolist() == [[0, 1], [0, 0]]
    assert O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25).tolist() == [[0, 1]]
    for t in (0.0, 0.125, 0.5, 0.75, 1.5):
        a, b = cases.corner_box(1, 1, 1), cases.corner_box(1, 1, 1, None, (t, 0, 0))
        exp = (1 - t) / (1 + t) if t < 1 else 0.0
        assert abs(O.iou3d(a.astype(np.float32), b.astype(np.float32)) - exp) < 1e-6, t
    a, b = cases.corner_box(1, 1, 1).astype(np.float32), cases.corner_box(1, 1, 1, np.pi / 4).astype(np.float32)
    oct_area = 2 * (np.sqrt(2.0) - 1)
    assert abs(O.bev_intersection(a, b) - oct_area) < 1e-6
    assert abs(O.iou3d(a, b) - oct_area / (2 - oct_area)) < 1e-6
