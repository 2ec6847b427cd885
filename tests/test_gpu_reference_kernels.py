"""GPU: the product's HIP kernels, through the C ABI, against the REFERENCE's own device kernels running beside them on the same
MI355X (oracle/_ref/libref_{sampling,grouping}_gpu.so: tf_sampling_g.cu / tf_grouping_g.cu compiled for gfx950 where they lie,
-ffp-contract=off = SURVEY appendix A.1; front-end oracle/ref_gpu.py), on fresh seeded inputs -- and the CPU oracle as the third
party.  Bit-exact for indices, counts and copies; the two scatter-adds with integer-valued cotangents (exact in any order of the
atomics) bit for bit, with real-valued ones to 1e-5.

The libraries are built in the build container (the reference tree exists only there) and travel with the snapshot; if they are
missing the tests are skipped -- unless VOTENET_REQUIRE_REF=1 (conftest.py sets it whenever oracle/_ref/ exists in the snapshot), in
which case that is a failure.  Either way the committed fixtures tests/golden/ref_gpu_*.npz hold the same kernels' outputs and the
product is compared with them directly in tests/test_gpu_ref_fixtures.py (no oracle/_ref, no oracle in between)."""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def R(hiplib):
    from oracle import ref_gpu
    if not ref_gpu.available():
        import os
        assert os.environ.get("VOTENET_REQUIRE_REF", "0") != "1", \
            "oracle/_ref/ is in the snapshot but libref_*_gpu.so does not load: the reference-kernel pins would silently not run"
        pytest.skip("oracle/_ref/libref_*_gpu.so not built (reference tree not mounted when the snapshot was made)")
    return ref_gpu


@pytest.fixture(scope="module")
def ops(hiplib):
    from votenet_amd import tf_grouping, tf_sampling

    class Ops:
        pass
    o = Ops()
    o.s, o.g = tf_sampling, tf_grouping
    return o


# every sampling kernel of the product: register resident (n <= 4096), bucket-pruned (<= 24576), L2-resident (<= 262144), streaming
@pytest.mark.parametrize("b,n,m", [(2, 64, 64), (3, 511, 77), (2, 513, 128), (2, 2048, 1024), (1, 4096, 512), (2, 8192, 300),
                                   (2, 20480, 2048), (1, 24577, 60), (2, 80000, 256), (1, 262145, 6)])
def test_fps_product_is_the_reference_kernel(ops, dev, R, O, b, n, m):
    xyz = np.random.default_rng(n * 13 + m).random((b, n, 3), dtype=np.float32) * 5
    ref = R.farthest_point_sample(m, xyz)
    assert (N(ops.s.farthest_point_sample(m, T(xyz, dev))) == ref).all()
    assert (O.farthest_point_sample(m, xyz) == ref).all()


def test_fps_room_scenes_ties_duplicates_and_exhaustion(ops, dev, R, O):
    from votenet_amd import synth
    rng = np.random.default_rng(17)
    clouds = [(synth.room_batch(3, 20480, 4242), 2048),                                         # fresh room scenes, the sa1 shape
              (np.round(rng.random((2, 9000, 3), dtype=np.float32) * 6) / 2, 400),              # coarse lattice: exact ties, duplicates
              (np.repeat(rng.random((1, 50, 3), dtype=np.float32), 100, axis=1), 120),          # 50 distinct points: index 0 repeats
              (rng.random((1, 40, 3), dtype=np.float32), 64)]                                   # m > n
    for xyz, m in clouds:
        ref = R.farthest_point_sample(m, xyz)
        assert (N(ops.s.farthest_point_sample(m, T(xyz, dev))) == ref).all(), xyz.shape
        assert (O.farthest_point_sample(m, xyz) == ref).all(), xyz.shape


def test_fps_prefix_shortcut_levels_are_the_reference_kernel(ops, dev, R):
    """The levels below the first sample from centres that are already in farthest-point order (fps_prefix_check_kernel): the
    reference kernel on the same gathered clouds gives the same picks."""
    from votenet_amd import synth
    xyz = synth.room_batch(2, 20480, 777)
    i1 = R.farthest_point_sample(2048, xyz)
    l1 = R.gather_point(xyz, i1)
    for m in (1024, 512, 256):
        ref = R.farthest_point_sample(m, l1)
        assert (N(ops.s.farthest_point_sample(m, T(l1, dev))) == ref).all(), m
        l1 = R.gather_point(l1, ref)


def test_gather_point_and_gradient(ops, dev, R):
    rng = np.random.default_rng(23)
    xyz = rng.random((3, 5000, 3), dtype=np.float32)
    idx = rng.integers(0, 5000, size=(3, 700)).astype(np.int32)
    idx[:, :50] = 7  # a hot target for the atomics
    assert (N(ops.s.gather_point(T(xyz, dev), T(idx, dev))) == R.gather_point(xyz, idx)).all()
    cot = rng.integers(-8, 9, size=(3, 700, 3)).astype(np.float32)
    assert (N(ops.s.gather_point_grad_raw(5000, T(idx, dev), T(cot, dev))) == R.gather_point_grad(5000, idx, cot)).all()
    cot = rng.standard_normal((3, 700, 3)).astype(np.float32)
    np.testing.assert_allclose(N(ops.s.gather_point_grad_raw(5000, T(idx, dev), T(cot, dev))), R.gather_point_grad(5000, idx, cot),
                               rtol=0, atol=1e-5 * 50)


@pytest.mark.parametrize("n,m", [(1, 33), (5, 257), (1000, 257), (8191, 300), (8193, 300), (20000, 1000)])
def test_prob_sample_is_the_reference_kernel(ops, dev, R, n, m):
    rng = np.random.default_rng(n + m)
    p = rng.random((3, n), dtype=np.float32) + 1e-3
    r = rng.random((3, m), dtype=np.float32)
    assert (N(ops.s.prob_sample(T(p, dev), T(r, dev))) == R.prob_sample(p, r)).all()


@pytest.mark.parametrize("b,n,m,r,k", [(2, 128, 8, 0.3, 32), (4, 512, 128, 0.1, 64), (2, 2048, 512, 0.2, 32), (2, 20480, 2048, 0.2, 64),
                                       (2, 2048, 1024, 0.4, 32), (2, 1024, 512, 0.8, 16), (2, 512, 256, 1.2, 16), (1, 80000, 512, 0.2, 64)])
def test_ball_query_and_group_are_the_reference_kernels(ops, dev, R, O, b, n, m, r, k):
    from votenet_amd import synth
    xyz1 = synth.room_batch(b, n, 99 + n)
    xyz2 = xyz1[:, np.random.default_rng(n).permutation(n)[:m]]
    xyz2[:, -1] += 50.0  # a centre with no neighbour: the reference leaves its row unwritten (pre-filled with 0 as the product writes it)
    ridx, rcnt = R.query_ball_point(r, k, xyz1, xyz2, fill=0)
    idx, cnt = ops.g.query_ball_point(r, k, T(xyz1, dev), T(xyz2, dev))
    assert (N(idx) == ridx).all() and (N(cnt) == rcnt).all()
    assert rcnt[:, -1].max() == 0
    if n <= 20480:
        oi, oc = O.query_ball_point(r, k, xyz1, xyz2)
        assert (oi == ridx).all() and (oc == rcnt).all()
    feats = np.random.default_rng(1).standard_normal((b, n, 5)).astype(np.float32)
    assert (N(ops.g.group_point(T(feats, dev), idx)) == R.group_point(feats, ridx)).all()
    cot = np.random.default_rng(2).integers(-4, 5, size=(b, m, k, 5)).astype(np.float32)
    assert (N(ops.g.group_point_grad_raw(n, idx, T(cot, dev))) == R.group_point_grad(n, ridx, cot)).all()


@pytest.mark.parametrize("radius", [0.2, 0.4, 0.8, 1.2])
def test_ball_query_boundary_is_sqrtf_not_r_squared(ops, dev, R, radius):
    """a hit is `max(sqrtf(s), 1e-20f) < radius`; the product decides `s < T(r)`.  Candidates placed within a
    few ulps of the sphere on both sides, evaluated by the reference kernel itself."""
    r32 = np.float32(radius)
    steps = np.arange(-40, 41)
    d = r32 * (np.float32(1.0) + steps.astype(np.float32) * np.float32(2.0 ** -23))
    xyz1 = np.zeros((1, d.size * 3, 3), np.float32)
    xyz1[0, :d.size, 0] = d
    xyz1[0, d.size:2 * d.size, 1] = -d
    xyz1[0, 2 * d.size:, 0] = d * np.float32(0.6)
    xyz1[0, 2 * d.size:, 2] = d * np.float32(0.8)
    xyz2 = np.zeros((1, 1, 3), np.float32)
    ridx, rcnt = R.query_ball_point(radius, 256, xyz1, xyz2)
    idx, cnt = ops.g.query_ball_point(radius, 256, T(xyz1, dev), T(xyz2, dev))
    assert (N(cnt) == rcnt).all() and (N(idx) == ridx).all()
    assert 0 < int(rcnt[0, 0]) < xyz1.shape[1]  # the sphere really cuts the candidates


def test_selection_sort_and_knn_are_the_reference_kernel(ops, dev, R):
    for name, (dist, k) in cases.selection_sort_cases().items():
        ri, rv = R.selection_sort(k, dist)
        oi, ov = ops.g.select_top_k(k, T(dist, dev))
        assert (N(oi)[..., :k] == ri[..., :k]).all() and (N(ov)[..., :k] == rv[..., :k]).all(), name
    rng = np.random.default_rng(31)
    a, q = rng.random((2, 700, 3), dtype=np.float32), rng.random((2, 90, 3), dtype=np.float32)
    d = np.zeros((2, 90, 700), np.float32)
    for ch in range(3):  # tf_grouping.py:62-66: the squared distances the reference feeds its SelectionSort
        t = (a[:, None, :, ch] - q[:, :, None, ch]).astype(np.float32)
        d = (t * t).astype(np.float32) if ch == 0 else (d + t * t).astype(np.float32)
    ri, rv = R.selection_sort(16, d)
    val, idx = ops.g.knn_point(16, T(a, dev), T(q, dev))
    assert (N(idx) == ri[..., :16]).all() and (N(val) == rv[..., :16]).all()


def test_edge_shapes_against_the_reference_kernels(ops, dev, R):
    """The corners the reference kernels define by construction: a one-point cloud, m = 1, nsample = 1, a radius that reaches nothing
    and one that reaches everything, one feature channel and many, k = 1 and k = n in SelectionSort."""
    rng = np.random.default_rng(41)
    one = rng.random((2, 1, 3), dtype=np.float32)
    assert (N(ops.s.farthest_point_sample(1, T(one, dev))) == R.farthest_point_sample(1, one)).all()
    assert (N(ops.s.farthest_point_sample(5, T(one, dev))) == R.farthest_point_sample(5, one)).all()  # m > n = 1: index 0 repeats
    xyz = rng.random((2, 777, 3), dtype=np.float32)
    q = rng.random((2, 33, 3), dtype=np.float32)
    for r, k in ((1e-6, 8), (10.0, 1), (10.0, 16), (0.15, 1), (0.15, 1000)):
        ridx, rcnt = R.query_ball_point(r, k, xyz, q)
        idx, cnt = ops.g.query_ball_point(r, k, T(xyz, dev), T(q, dev))
        assert (N(idx) == ridx).all() and (N(cnt) == rcnt).all(), (r, k)
    ridx, _ = R.query_ball_point(0.3, 12, xyz, q)
    for c in (1, 2, 67, 256):
        feats = rng.standard_normal((2, 777, c)).astype(np.float32)
        assert (N(ops.g.group_point(T(feats, dev), T(ridx, dev))) == R.group_point(feats, ridx)).all(), c
        cot = rng.integers(-3, 4, size=(2, 33, 12, c)).astype(np.float32)
        assert (N(ops.g.group_point_grad_raw(777, T(ridx, dev), T(cot, dev))) == R.group_point_grad(777, ridx, cot)).all(), c
    d = rng.random((2, 5, 40), dtype=np.float32)
    for k in (1, 40):
        ri, rv = R.selection_sort(k, d)
        oi, ov = ops.g.select_top_k(k, T(d, dev))
        assert (N(oi)[..., :k] == ri[..., :k]).all() and (N(ov)[..., :k] == rv[..., :k]).all(), k


def test_whole_geometry_chain_of_a_backbone_pass_is_the_reference_kernels(ops, dev, R):
    """sa1 ... sa4 of model.py's backbone on two room scenes, every level's sampling, gather, ball query and grouping computed by the
    reference's own kernels and by the product, each level fed with the reference's result of the level above: 20480 -> 2048 (r 0.2, K 64)
    -> 1024 (0.4, 32) -> 512 (0.8, 16) -> 256 (1.2, 16)."""
    from votenet_amd import synth
    xyz = synth.room_batch(2, 20480, 31337)
    for m, r, k in ((2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)):
        ref_fps = R.farthest_point_sample(m, xyz)
        assert (N(ops.s.farthest_point_sample(m, T(xyz, dev))) == ref_fps).all(), m
        ref_new = R.gather_point(xyz, ref_fps)
        assert (N(ops.s.gather_point(T(xyz, dev), T(ref_fps, dev))) == ref_new).all(), m
        ridx, rcnt = R.query_ball_point(r, k, xyz, ref_new)
        idx, cnt = ops.g.query_ball_point(r, k, T(xyz, dev), T(ref_new, dev))
        assert (N(idx) == ridx).all() and (N(cnt) == rcnt).all(), m
        assert rcnt.min() >= 1  # every centre is its own neighbour
        assert (N(ops.g.group_point(T(xyz, dev), idx)) == R.group_point(xyz, ridx)).all(), m
        xyz = ref_new


def test_non_finite_points_where_the_product_deliberately_leaves_the_reference(ops, dev, R):
    """DESIGN_HISTORY.md section 2, "Non-finite coordinates": what the reference kernels do with a NaN point is an artefact of fminf / max / '>'
    (tf_sampling_g.cu:142-146, tf_grouping_g.cu:24-25) -- shown here on the kernels themselves -- and the product DEFINES the case
    instead (a hole is never sampled and never a neighbour).  On finite clouds the two agree everywhere (every other test of this file).
    Also the canary that the "reference" of this file IS the reference: the product exports the same launcher names (the drop-in seam),
    and a reference library whose calls resolved to the product's launchers would show the product's behaviour here (oracle/Makefile
    links the drivers -Bsymbolic for that reason; round 5 saw it happen once with the product loaded RTLD_GLOBAL)."""
    rng = np.random.default_rng(43)
    xyz = rng.random((1, 600, 3), dtype=np.float32)
    xyz[0, 137] = np.nan
    ref = R.farthest_point_sample(8, xyz)
    # the hole keeps its initial running distance 1e38: picked at once, and as a centre it freezes every running distance
    # (min(NaN, td) = td), so the arg-max stays the hole itself
    assert ref[0, 0] == 0 and (ref[0, 1:] == 137).all()
    own = N(ops.s.farthest_point_sample(8, T(xyz, dev)))
    assert 137 not in own[0] and len(set(own[0].tolist())) == 8
    clean = xyz.copy()
    clean[0, 137] = clean[0, 0]  # the product reads a hole as a copy of point 0
    assert (own == R.farthest_point_sample(8, clean)).all()
    q = xyz[:, :5].copy()
    ridx, rcnt = R.query_ball_point(0.05, 64, xyz, q)
    assert all(137 in ridx[0, j, :rcnt[0, j]] for j in range(5))  # max(sqrtf(NaN), 1e-20f) = 1e-20 < r: a "neighbour" of every centre
    idx, cnt = ops.g.query_ball_point(0.05, 64, T(xyz, dev), T(q, dev))
    idx, cnt = N(idx), N(cnt)
    assert all(137 not in idx[0, j, :cnt[0, j]] for j in range(5))
    hole_free = np.delete(xyz, 137, axis=1)
    fidx, fcnt = R.query_ball_point(0.05, 64, hole_free, q)
    assert (cnt == fcnt).all()  # the product's neighbour sets are the reference's on the cloud without the hole
