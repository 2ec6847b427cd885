"""GPU: the HIP path, called through the C ABI (votenet_amd.tf_* -> libvotenet_hip.so), against
the CPU oracle on the same seeded inputs and against the committed golden vectors.
Bar: bit-exact for indices / counts / pure copies; 1e-5 for floating-point sums."""
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


@pytest.fixture(scope="module")
def ops(hiplib):
    from votenet_amd import tf_grouping, tf_interpolate, tf_nms3d, tf_sampling

    class Ops:
        pass
    o = Ops()
    o.s, o.g, o.i, o.n = tf_sampling, tf_grouping, tf_interpolate, tf_nms3d
    return o


# ------------------------------------------------------------------ FPS
def test_fps_golden_cases(ops, dev, golden):
    g = golden("fps_cases")
    for name, (xyz, m) in cases.fps_cases().items():
        got = N(ops.s.farthest_point_sample(m, T(xyz, dev)))
        assert got.dtype == np.int32 and (got == g[name]).all(), name


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 64, 64), (3, 511, 77), (2, 512, 128), (2, 513, 128), (1, 1024, 256),
                                   (2, 2048, 1024), (1, 2049, 100), (1, 4096, 512), (1, 8192, 300), (1, 12288, 200),
                                   (1, 16384, 200), (2, 20480, 300), (1, 24576, 150), (1, 24577, 60), (1, 40000, 40), (2, 65536, 50),
                                   (2, 80000, 120), (1, 140000, 30), (1, 262144, 12), (1, 262145, 6)])
def test_fps_vs_oracle(ops, dev, O, b, n, m):
    xyz = np.random.default_rng(n * 7 + m).random((b, n, 3), dtype=np.float32) * 5
    got = N(ops.s.farthest_point_sample(m, T(xyz, dev)))
    assert (got == O.farthest_point_sample(m, xyz)).all()


def test_fps_exact_ties_vs_oracle(ops, dev, O):
    rng = np.random.default_rng(5)
    for n, m in [(700, 300), (3000, 500), (20000, 200), (70000, 150)]:
        xyz = np.round(rng.random((2, n, 3), dtype=np.float32) * 6) / 2  # coarse lattice: many equal distances, duplicates
        got = N(ops.s.farthest_point_sample(m, T(xyz, dev)))
        assert (got == O.farthest_point_sample(m, xyz)).all(), (n, m)


def test_fps_two_samples_per_round_experiment_is_the_same_sequence(ops, dev, O, hiplib):
    """fps_bucket2_kernel (votenet_fps_debug_two_pick): up to two picks per round when the runner-up is provably the next
    arg-max.  Off by default (slower, DESIGN_HISTORY.md 4.1); the indices must be the oracle's, duplicates and exact ties included."""
    from votenet_amd import synth
    hiplib.votenet_fps_debug_two_pick.restype = None
    hiplib.votenet_fps_debug_two_pick(1)
    try:
        rng = np.random.default_rng(11)
        cases = [rng.random((2, 20480, 3), dtype=np.float32) * 5, synth.room_batch(2, 20480, 77),
                 np.round(rng.random((1, 9000, 3), dtype=np.float32) * 6) / 2,                       # lattice: ties, duplicates
                 np.repeat(rng.random((1, 50, 3), dtype=np.float32), 100, axis=1)]                   # 50 distinct points only
        for xyz, m in zip(cases, (700, 2048, 400, 120)):
            got = N(ops.s.farthest_point_sample(m, T(xyz, dev)))
            assert (got == O.farthest_point_sample(m, xyz)).all()
    finally:
        hiplib.votenet_fps_debug_two_pick(0)


def test_fps_of_an_fps_ordered_subset_short_cut_is_exact(ops, dev, O, hiplib):
    """Levels 2-4 sample from centres that are already in farthest-point order: the parallel check confirms 0..m-1 (or, after an
    exact tie that the subset's tie key resolves differently, lets the sampling rounds run).  Against the oracle on the
    gathered subset, with the check on and off; lattice clouds provoke the ties."""
    from votenet_amd import synth
    hiplib.votenet_fps_debug_prefix_check.restype = None
    rng = np.random.default_rng(3)
    clouds = [synth.room_batch(2, 20480, 5), rng.random((2, 6000, 3), dtype=np.float32) * 4,
              np.round(rng.random((2, 5000, 3), dtype=np.float32) * 8) / 2,    # lattice: exact ties and duplicates
              np.round(rng.random((1, 3000, 3), dtype=np.float32) * 3)]         # 64 distinct positions only
    identity = 0
    for xyz in clouds:
        for s_n, m in ((2048, 1024), (1024, 512), (512, 256), (1024, 256), (300, 300)):
            sub = O.gather_point(xyz, O.farthest_point_sample(s_n, xyz))
            exp = O.farthest_point_sample(m, sub)
            identity += int((exp == np.arange(m)).all())
            for on in (1, 0):
                hiplib.votenet_fps_debug_prefix_check(on)
                got = N(ops.s.farthest_point_sample(m, T(sub, dev)))
                assert (got == exp).all(), (s_n, m, on)
    hiplib.votenet_fps_debug_prefix_check(1)
    assert 8 <= identity < 20  # both outcomes were exercised: confirmed prefixes and ties that break them


def _with_holes(xyz, rng, frac=0.01, first=False):
    """NaN / +-Inf in single components, in whole points, optionally in point 0 (the first centre)."""
    x = xyz.copy()
    b, n = x.shape[:2]
    k = max(2, int(n * frac))
    for s in range(b):
        pts = rng.choice(np.arange(1, n), size=min(k, n - 1), replace=False)
        vals = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), size=len(pts))
        comp = rng.integers(0, 3, size=len(pts))
        x[s, pts, comp] = vals
        x[s, pts[: len(pts) // 3]] = np.nan  # whole points
        if first:
            x[s, 0, s % 3] = np.nan
    return x


@pytest.mark.parametrize("n,m", [(300, 100), (2048, 512), (4096, 700), (8192, 300), (20480, 400), (24577, 60), (80000, 150), (262145, 8)])
def test_fps_of_a_cloud_with_nan_and_inf_points_is_defined_in_range_and_equals_the_oracle(ops, dev, O, n, m):
    """Depth clouds have holes.  Every sampling kernel (register, bucket-pruned, L2-resident, streaming, the prefix shortcut) reads a
    point with a non-finite coordinate as a copy of point 0 (fps.hip: fps_get; oracle_sampling.c defines the same): never a hang,
    never an index outside [0, n), a hole is never sampled, and the picks are the oracle's."""
    rng = np.random.default_rng(n + m)
    base = rng.random((2, n, 3), dtype=np.float32) * 5
    for first in (False, True):
        xyz = _with_holes(base, rng, first=first)
        got = N(ops.s.farthest_point_sample(m, T(xyz, dev)))
        assert got.min() >= 0 and got.max() < n
        exp = O.farthest_point_sample(m, xyz)
        assert (got == exp).all()
        holes = ~np.isfinite(xyz).all(-1)
        for s in range(2):
            assert not holes[s, got[s, 1:]].any()
        assert (got[:, 0] == 0).all()


def test_geometry_chain_and_ball_query_with_holes(ops, dev, O):
    """The whole coordinate-only chain of sa1 / sa2 on a cloud with holes: FPS -> centres -> ball query over the spatial index the
    sampling left behind -> the next level (prefix shortcut on centres whose first entry may be a hole).  Indices in range; the
    ball query equals the full scan and the oracle (a non-finite distance never passes the radius test: holes are no neighbours)."""
    from votenet_amd import synth, tf_sampling as S
    rng = np.random.default_rng(9)
    for first in (False, True):
        xyz = _with_holes(synth.room_batch(2, 20480, 31), rng, frac=0.02, first=first)
        x = T(xyz, dev)
        S._INDEX_CACHE.clear()
        fidx = ops.s.farthest_point_sample(2048, x)
        assert S.cached_index(x) is not None
        assert (N(fidx) == O.farthest_point_sample(2048, xyz)).all()
        new_xyz = ops.s.gather_point(x, fidx)
        idx, cnt = ops.g.query_ball_point(0.2, 64, x, new_xyz)               # over the index
        oi, oc = O.query_ball_point(0.2, 64, xyz, N(new_xyz))
        assert (N(idx) == oi).all() and (N(cnt) == oc).all()
        ops.g.USE_INDEX = False
        try:
            bi, bc = ops.g.query_ball_point(0.2, 64, x, new_xyz)             # full scan
        finally:
            ops.g.USE_INDEX = True
        assert torch.equal(bi, idx) and torch.equal(bc, cnt)
        holes = ~np.isfinite(xyz).all(-1)
        ii, cc = N(idx), N(cnt)
        for s in range(2):
            used = np.concatenate([ii[s, j, :max(1, cc[s, j])] for j in range(2048) if cc[s, j] > 0])
            assert not holes[s, used].any()
        # next level: the centres (a hole at index 0 when `first`) through the prefix shortcut and the rounds
        f2 = ops.s.farthest_point_sample(1024, new_xyz)
        assert (N(f2) == O.farthest_point_sample(1024, N(new_xyz))).all()
        assert int(f2.min()) >= 0 and int(f2.max()) < 2048


def test_fps_full_size_properties(ops, dev):
    """BASELINE config 2 size (8 x 20480 -> 2048): size-independent properties, checked on the device."""
    xyz = T(np.random.default_rng(0).random((8, 20480, 3), dtype=np.float32) * 5, dev)
    idx = ops.s.farthest_point_sample(2048, xyz).long()
    assert (idx[:, 0] == 0).all()
    assert all(len(torch.unique(idx[s])) == 2048 for s in range(8))
    # idempotence of the prefix: FPS(512) is the first 512 picks of FPS(2048)
    assert (ops.s.farthest_point_sample(512, xyz).long() == idx[:, :512]).all()
    # each pick attains the maximum of the running min distance (fp64 recomputation, relative slack)
    s = 3
    p = xyz[s].double()
    td = torch.full((20480,), float("inf"), dtype=torch.float64, device=dev)
    for j in range(1, 300):
        td = torch.minimum(td, ((p - p[idx[s, j - 1]]) ** 2).sum(1))
        assert td[idx[s, j]] >= td.max() * (1 - 1e-5)


def _headline_cloud(kind):
    from votenet_amd import synth
    return synth.room_batch(8, 20480, 1000) if kind == "room" else synth.uniform_batch(8, 20480, 1000)


@pytest.mark.parametrize("kind", ["room", "uniform"])
def test_fps_sa1_full_size_default_mode_bit_exact(ops, dev, O, kind):
    """The headline launch itself -- fps_bucket_sort_kernel + fps_bucket_kernel<12,32> in its DEFAULT mode, 8 x 20480 -> 2048
    (BASELINE configs[1]/[2], the shape bench.py's roofline is quoted on) -- equals the oracle's restatement of
    tf_sampling_g.cu:105-170 on every one of the 8 x 2048 indices, on the bench's own room scenes and on the uniform cube."""
    xyz = _headline_cloud(kind)
    got = N(ops.s.farthest_point_sample(2048, T(xyz, dev)))
    exp = O.farthest_point_sample(2048, xyz)
    assert got.shape == (8, 2048) and (got == exp).all(), int((got != exp).sum())


@pytest.mark.parametrize("kind", ["room", "uniform"])
def test_ball_query_sa1_full_size_bit_exact(ops, dev, O, kind):
    """sa1's ball query at full size (8 x 2048 centres x 20480 candidates, r = 0.2, K = 64; tf_grouping_g.cu:3-36): every
    neighbour list and every pts_cnt equals the oracle (itself equal to the reference's compiled CPU twin,
    tests/test_oracle_golden.py).  Room scenes have full balls (10 % reach K), the uniform cube none (full scan)."""
    xyz = _headline_cloud(kind)
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(2048, xyz))
    idx, cnt = ops.g.query_ball_point(0.2, 64, T(xyz, dev), T(new_xyz, dev))
    oi, oc = O.query_ball_point(0.2, 64, xyz, new_xyz)
    assert (N(cnt) == oc).all() and (N(idx) == oi).all()
    assert oc.min() >= 1 and ((oc == 64).mean() > 0.05 if kind == "room" else (oc == 64).mean() < 0.01)  # full balls only in rooms


def test_config5_scene_fps_and_ball_query_bit_exact(ops, dev, O):
    """BASELINE config 5's first level on full-size scenes (80 000 points -> 2048 centres, r = 0.2, K = 64): the L2-resident
    FPS variant (fps_bucket_l2_kernel) and the ball query against the oracle, index for index, on two scenes of the batch
    (the oracle needs ~0.6 s per scene for these two ops; it is the MLP stack that takes minutes)."""
    from votenet_amd import synth
    xyz = synth.room_batch(2, 80000, 77, size=(8.0, 3.0, 8.0), nbox=(15, 25))
    x = T(xyz, dev)
    fidx = ops.s.farthest_point_sample(2048, x)
    exp = O.farthest_point_sample(2048, xyz)
    assert (N(fidx) == exp).all()
    new_xyz = ops.s.gather_point(x, fidx)
    idx, cnt = ops.g.query_ball_point(0.2, 64, x, new_xyz)
    oi, oc = O.query_ball_point(0.2, 64, xyz, O.gather_point(xyz, exp))
    assert (N(cnt) == oc).all() and (N(idx) == oi).all()
    # levels below: 2048 -> 1024 (seeds) through the prefix check, r = 0.4
    l1 = N(new_xyz)
    f2 = ops.s.farthest_point_sample(1024, new_xyz)
    e2 = O.farthest_point_sample(1024, l1)
    assert (N(f2) == e2).all()
    i2, c2 = ops.g.query_ball_point(0.4, 64, new_xyz, ops.s.gather_point(new_xyz, f2))
    oi2, oc2 = O.query_ball_point(0.4, 64, l1, O.gather_point(l1, e2))
    assert (N(c2) == oc2).all() and (N(i2) == oi2).all()


@pytest.mark.parametrize("mode", [1, 3, 5])
@pytest.mark.parametrize("b,n,m", [(2, 80000, 300), (1, 24577, 64), (3, 98304, 40), (9, 30001, 33), (1, 50000, 1)])
def test_fps_scene_over_four_workgroups_gives_the_same_indices(ops, dev, hiplib, b, n, m, mode):
    """fps_bucket_split_kernel (votenet_debug_fps_split(1); off by default, profiles/r05_fps_split.txt): a scene's buckets held in the
    registers of four workgroups that agree on every round's winner through L2 -- the indices of tf_sampling_g.cu:105-170 exactly as
    the default kernel gives them (that one is pinned to the oracle and to the reference's kernel above), more scenes than XCDs,
    ragged sizes, both ends of the kernel's range; no poll may have given up."""
    import ctypes
    from votenet_amd import synth
    hiplib.votenet_debug_fps_split_timeouts.restype = ctypes.c_uint
    x = T(synth.room_batch(b, n, 5 + n % 7), dev)
    ref = ops.s.farthest_point_sample(m, x)
    hiplib.votenet_debug_fps_split(mode)  # 1: 4 workgroups x 12 waves per scene, 3: 12 x 4, 5: 6 x 8
    try:
        got = ops.s.farthest_point_sample(m, x)
        again = ops.s.farthest_point_sample(m, x)  # the exchange buffer is zeroed per launch: a second call sees no stale round numbers
        torch.cuda.synchronize()
    finally:
        hiplib.votenet_debug_fps_split(0)
    assert (N(got) == N(ref)).all() and (N(again) == N(ref)).all()
    assert hiplib.votenet_debug_fps_split_timeouts() == 0


def test_fps_reference_launcher_shim_large_batch(ops, dev, hiplib):
    """tf_sampling.cpp:94,115: the reference calls farthestpointsamplingLauncher with a TensorShape{32,n} scratch whatever
    the batch.  The exported shim (C++ linkage, the reference's exact signature) must give the op's result with exactly
    that much scratch, also for a batch whose bucket tables would not fit at once."""
    import ctypes
    fn = getattr(hiplib, "_Z29farthestpointsamplingLauncheriiiPKfPfPi")
    fn.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
    fn.restype = None
    for b, n, m in [(40, 5000, 64), (3, 700, 50), (33, 24577, 8)]:
        xyz = T(np.random.default_rng(b + n).random((b, n, 3), dtype=np.float32) * 4, dev)
        temp = torch.full((32, n), float("nan"), dtype=torch.float32, device=dev)
        guard = torch.zeros(1024, dtype=torch.float32, device=dev)  # allocated right after: catches an overrun in practice
        out = torch.full((b, m), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        fn(b, n, m, xyz.data_ptr(), temp.data_ptr(), out.data_ptr())  # null stream, like the reference
        torch.cuda.synchronize()
        assert (out == ops.s.farthest_point_sample(m, xyz)).all(), (b, n, m)
        assert (guard == 0).all()


# ------------------------------------------------------------------ gather
def test_gather_point_and_grad(ops, dev, O):
    rng = np.random.default_rng(3)
    xyz = rng.random((3, 500, 3), dtype=np.float32)
    idx = rng.integers(0, 500, (3, 120)).astype(np.int32)
    idx[0, :10] = 7  # repeated index: the gradient sums
    x = T(xyz, dev).requires_grad_(True)
    out = ops.s.gather_point(x, T(idx, dev))
    assert (N(out) == O.gather_point(xyz, idx)).all()
    go = rng.random((3, 120, 3), dtype=np.float32)
    out.backward(T(go, dev))
    assert np.allclose(N(x.grad), O.gather_point_grad(xyz, idx, go), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ ball query / group
def test_cfg1_golden(ops, dev, golden):
    g = golden("cfg1")
    xyz = cases.cfg1_cloud()
    x = T(xyz, dev)
    fidx = ops.s.farthest_point_sample(512, x)
    assert (N(fidx) == g["fps_idx"]).all()
    new_xyz = ops.s.gather_point(x, fidx)
    idx, cnt = ops.g.query_ball_point(0.2, 32, x, new_xyz)
    assert (N(idx) == g["idx"]).all() and (N(cnt) == g["pts_cnt"]).all()
    assert sha(N(ops.g.group_point(x, idx))) == str(g["grouped_xyz_sha"])


def test_grouping_reference_test_shape_golden(ops, dev, golden):
    c, g = cases.grouping_optest(), golden("grouping_optest")
    idx, cnt = ops.g.query_ball_point(c["radius"], c["nsample"], T(c["xyz1"], dev), T(c["xyz2"], dev))
    assert (N(idx) == g["idx"]).all() and (N(cnt) == g["pts_cnt"]).all()
    pts = T(c["points"], dev).requires_grad_(True)
    out = ops.g.group_point(pts, idx)
    assert (N(out) == g["out"]).all()
    out.backward(T(c["grad_out"], dev))
    assert np.allclose(N(pts.grad), g["grad"], rtol=1e-5, atol=1e-6)


def test_grouping_demo_golden(ops, dev, golden):
    c, g = cases.grouping_demo(), golden("grouping_demo")
    idx, cnt = ops.g.query_ball_point(c["radius"], c["nsample"], T(c["xyz1"], dev), T(c["xyz2"], dev))
    assert sha(N(idx)) == str(g["idx_sha"]) and (N(cnt) == g["pts_cnt"]).all()
    assert sha(N(ops.g.group_point(T(c["points"], dev), idx))) == str(g["out_sha"])


@pytest.mark.parametrize("b,n,m,r,k", [(1, 1, 1, 0.5, 4), (2, 63, 5, 0.3, 8), (2, 700, 90, 0.2, 16), (1, 2048, 512, 0.2, 32),
                                       (3, 513, 64, 0.4, 64), (1, 100, 7, 0.05, 8), (2, 64, 64, 2.0, 5), (1, 2049, 65, 0.1, 100),
                                       (1, 4097, 130, 0.25, 64), (2, 20480, 256, 0.04, 64), (1, 9000, 64, 0.3, 64), (2, 512, 256, 0.3, 64),
                                       (8, 1024, 256, 0.3, 64), (1, 1025, 70, 0.2, 16), (2, 1024, 100, 0.02, 64)])
def test_ball_query_vs_oracle(ops, dev, O, b, n, m, r, k):
    rng = np.random.default_rng(n + m)
    xyz1 = rng.random((b, n, 3), dtype=np.float32)
    xyz2 = rng.random((b, m, 3), dtype=np.float32)
    idx, cnt = ops.g.query_ball_point(r, k, T(xyz1, dev), T(xyz2, dev))
    oi, oc = O.query_ball_point(r, k, xyz1, xyz2)
    assert (N(cnt) == oc).all()
    assert (N(idx) == oi).all()  # includes all-zero rows for queries with no neighbour


@pytest.mark.parametrize("n", [33, 512, 777, 1024, 2048])
def test_ball_query_small_cloud_forms_agree(ops, dev, hiplib, n):
    """n <= 2048: sixteen waves with the cloud as one super-chunk (default), the four-wave kernel and sixteen waves x eight groups give
    the same indices and counts (votenet_debug_ball_query_small)."""
    rng = np.random.default_rng(n)
    xyz1 = T(rng.random((3, n, 3), dtype=np.float32), dev)
    xyz2 = T(rng.random((3, 130, 3), dtype=np.float32), dev)
    hook = hiplib.votenet_debug_ball_query_small
    hook.restype = None
    got = []
    try:
        for form in (0, 4, 16):
            hook(form)
            idx, cnt = ops.g.query_ball_point(0.25, 64, xyz1, xyz2)
            got.append((N(idx), N(cnt)))
    finally:
        hook(0)
    for idx, cnt in got[1:]:
        assert (idx == got[0][0]).all() and (cnt == got[0][1]).all()


@pytest.mark.parametrize("b,n,m,r,k,scale", [(2, 4097, 130, 0.25, 64, 1.0), (1, 9000, 257, 0.3, 16, 2.0), (2, 20480, 500, 0.2, 64, 5.0),
                                             (1, 20480, 64, 9.0, 32, 5.0), (1, 24577, 100, 0.15, 8, 3.0), (1, 80000, 300, 0.2, 64, 8.0),
                                             (1, 131072, 40, 0.1, 200, 4.0), (1, 5000, 33, 0.01, 4, 1.0)])
def test_ball_query_over_the_spatial_index_equals_the_full_scan(ops, dev, O, b, n, m, r, k, scale):
    """votenet_query_ball_point_indexed (bucket culling + index-ordered read-out through an LDS bitmap) against the full scan
    and the oracle: same neighbour lists, same pts_cnt -- index built on its own, and the one a farthest-point sampling of
    the same tensor leaves behind (n <= 24576: register kernel; above: the L2 kernel, which rewrites the sorted copy)."""
    from votenet_amd import tf_sampling as S
    rng = np.random.default_rng(n + m)
    xyz1 = rng.random((b, n, 3), dtype=np.float32) * scale
    xyz1[:, 7] = xyz1[:, 3]  # duplicates
    xyz2 = rng.random((b, m, 3), dtype=np.float32) * scale
    xyz2[:, 0] = 50.0        # a query with no neighbour: all-zero row, count 0
    oi, oc = O.query_ball_point(r, k, xyz1, xyz2)
    x1, x2 = T(xyz1, dev), T(xyz2, dev)
    ops.g.USE_INDEX = False
    try:
        bi, bc = ops.g.query_ball_point(r, k, x1, x2)
    finally:
        ops.g.USE_INDEX = True
    assert (N(bi) == oi).all() and (N(bc) == oc).all()
    S._INDEX_CACHE.clear()
    ii, ic = ops.g.query_ball_point(r, k, x1, x2)          # builds the index itself
    assert S.cached_index(x1) is not None
    assert (N(ii) == oi).all() and (N(ic) == oc).all()
    S._INDEX_CACHE.clear()
    S.farthest_point_sample(min(64, n), x1)                  # leaves the index behind
    assert S.cached_index(x1) is not None
    fi, fc = ops.g.query_ball_point(r, k, x1, x2)
    assert (N(fi) == oi).all() and (N(fc) == oc).all()
    x1.add_(0.0)                                             # version bump: the cached index is not trusted
    assert S.cached_index(x1) is None


def test_ball_query_boundary_radius(ops, dev, O):
    """Pairs at distance exactly r, one ulp below, one ulp above: sqrtf(s) < r must be decided as the reference does."""
    for r in [0.2, 0.4, 0.8, 1.2, 0.3, 0.1]:
        r32 = np.float32(r)
        q = np.zeros((1, 1, 3), np.float32)
        xs = [r32, np.nextafter(r32, np.float32(0)), np.nextafter(r32, np.float32(9)), r32 * np.float32(0.5)]
        xyz1 = np.zeros((1, 64, 3), np.float32)
        for t, x in enumerate(xs):
            xyz1[0, t * 3, 0] = x
            xyz1[0, t * 3 + 1, 1] = x
            xyz1[0, t * 3 + 2, :] = x / np.sqrt(np.float32(3))
        xyz1[0, 12:] = 9.0
        idx, cnt = ops.g.query_ball_point(r, 16, T(xyz1, dev), T(q, dev))
        oi, oc = O.query_ball_point(r, 16, xyz1, q)
        assert (N(idx) == oi).all() and (N(cnt) == oc).all(), r


def test_sa1_full_size_ball_query_properties(ops, dev):
    """sa1 size (8 x 20480 candidates, 2048 queries, r=0.2, K=64) checked through properties on the device."""
    xyz = T(np.random.default_rng(1).random((8, 20480, 3), dtype=np.float32) * 5, dev)
    fidx = ops.s.farthest_point_sample(2048, xyz)
    new_xyz = ops.s.gather_point(xyz, fidx)
    idx, cnt = ops.g.query_ball_point(0.2, 64, xyz, new_xyz)
    assert (cnt >= 1).all() and (cnt <= 64).all()
    li = idx.long()
    g = torch.gather(xyz.unsqueeze(1).expand(-1, 2048, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, 3))
    d = (g - new_xyz.unsqueeze(2)).norm(dim=-1)
    assert (d < 0.2 + 1e-5).all()  # every listed neighbour is inside the ball
    valid = torch.arange(64, device=dev)[None, None, :] < cnt.unsqueeze(-1)
    inc = (li[..., 1:] > li[..., :-1]) | ~valid[..., 1:]
    assert inc.all()  # ascending candidate order within the valid prefix
    assert ((li == li[..., :1]) | valid).all()  # padding repeats the first hit
    # count == min(K, true neighbour count) on a sample of queries (fp64 recount with a safety band)
    for s, j in [(0, 0), (3, 100), (7, 2047)]:
        dd = (xyz[s].double() - new_xyz[s, j].double()).norm(dim=-1)
        lo, hi = int((dd < 0.2 - 1e-5).sum()), int((dd < 0.2 + 1e-5).sum())
        assert min(lo, 64) <= int(cnt[s, j]) <= min(hi, 64)


@pytest.mark.parametrize("c", [1, 3, 4, 16, 67, 128])
def test_group_point_and_grad(ops, dev, O, c):
    rng = np.random.default_rng(c)
    pts = rng.random((2, 300, c), dtype=np.float32)
    idx = rng.integers(0, 300, (2, 40, 16)).astype(np.int32)
    p = T(pts, dev).requires_grad_(True)
    out = ops.g.group_point(p, T(idx, dev))
    assert (N(out) == O.group_point(pts, idx)).all()
    go = rng.random((2, 40, 16, c), dtype=np.float32)
    out.backward(T(go, dev))
    assert np.allclose(N(p.grad), O.group_point_grad(pts, idx, go), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ three_nn / interpolate
def test_interpolate_reference_test_shape_golden(ops, dev, golden):
    c, g = cases.interpolate_optest(), golden("interpolate_optest")
    dist, idx = ops.i.three_nn(T(c["xyz1"], dev), T(c["xyz2"], dev))
    assert (N(dist) == g["dist"]).all() and (N(idx) == g["idx"]).all()
    w = torch.ones_like(dist) / 3.0
    pts = T(c["points"], dev).requires_grad_(True)
    out = ops.i.three_interpolate(pts, idx, w)
    assert np.allclose(N(out), g["out"], rtol=1e-6, atol=1e-7)
    assert (N(out) == g["out"]).all()  # same un-fused evaluation order -> bit-exact
    out.backward(T(c["grad_out"], dev))
    assert np.allclose(N(pts.grad), g["grad"], rtol=1e-5, atol=1e-6)
    # utils.py:279-282 on both sides as un-fused fp32 with correctly rounded division (v_div_scale / v_div_fmas / v_div_fixup): bit-exact
    assert (N(ops.i.three_nn_weights(dist)) == g["weights_idw"]).all()


def test_interpolate_demo_golden(ops, dev, golden):
    c, g = cases.interpolate_demo(), golden("interpolate_demo")
    dist, idx = ops.i.three_nn(T(c["xyz1"], dev), T(c["xyz2"], dev))
    assert sha(N(dist)) == str(g["dist_sha"]) and sha(N(idx)) == str(g["idx_sha"])
    out = ops.i.three_interpolate(T(c["points"], dev), idx, torch.ones_like(dist) / 3.0)
    assert sha(N(out)) == str(g["out_sha"])


@pytest.mark.parametrize("n,m", [(77, 3), (100, 9), (1000, 1024), (333, 1025), (2050, 2500), (64, 7)])
def test_three_nn_ties_and_tiles_vs_oracle(ops, dev, O, n, m):
    """three_nn splits the known points over eight lanes per query and over LDS tiles of 1024: the merge must rank equal
    distances by index exactly as the reference's serial strict-'<' cascade (tf_interpolate.cpp:74-89).  Lattice clouds:
    most queries see many exactly equal distances, in different lanes and in different tiles; duplicated known points."""
    rng = np.random.default_rng(n * 31 + m)
    xyz1 = (np.round(rng.random((2, n, 3), dtype=np.float32) * 4) / 2).astype(np.float32)
    xyz2 = (np.round(rng.random((2, m, 3), dtype=np.float32) * 4) / 2).astype(np.float32)
    dist, idx = ops.i.three_nn(T(xyz1, dev), T(xyz2, dev))
    od, oi = O.three_nn(xyz1, xyz2)
    assert (N(idx) == oi).all() and (N(dist) == od).all()


def test_three_nn_weights_bit_exact_over_the_whole_range(ops, dev, O):
    """Inverse-distance weights (utils.py:279-282) against the oracle, bit for bit, over everything a squared distance can be: zero and
    below the 1e-10 clamp, subnormal, ordinary (random exponents), huge, exact ties; plus the distances of a real cloud."""
    rng = np.random.default_rng(5)
    e = rng.integers(-140, 120, size=(4, 50000, 3))
    d = (rng.random((4, 50000, 3)) * np.exp2(e.astype(np.float64))).astype(np.float32)
    d[0, :1000] = 0.0
    d[0, 1000:2000, 1] = 1e-10
    d[0, 2000:3000] = np.float32(1e-10) * (1 + rng.integers(-3, 4, size=(1000, 3)) * np.float32(2.0 ** -23))
    d[1, :1000, 2] = d[1, :1000, 1]
    d[2, :1000] = np.float32(3e38)
    d[3, :1000] = np.float32(1e-45)
    w, ow = ops.i.three_nn_weights(T(d, dev)), O.three_nn_weights(d)
    assert (N(w).view(np.uint32) == ow.view(np.uint32)).all()
    xyz1, xyz2 = rng.random((2, 4096, 3), dtype=np.float32), rng.random((2, 512, 3), dtype=np.float32)
    dist, _ = ops.i.three_nn(T(xyz1, dev), T(xyz2, dev))
    assert (N(ops.i.three_nn_weights(dist)) == O.three_nn_weights(N(dist))).all()


@pytest.mark.parametrize("b,n,m,c", [(2, 300, 40, 8), (1, 1024, 512, 256), (1, 50, 2, 4), (1, 50, 1, 5), (8, 512, 256, 256)])
def test_three_nn_interpolate_vs_oracle(ops, dev, O, b, n, m, c):
    rng = np.random.default_rng(n + m)
    xyz1 = rng.random((b, n, 3), dtype=np.float32)
    xyz2 = rng.random((b, m, 3), dtype=np.float32)
    if m > 10:
        xyz2[:, 5] = xyz2[:, 9]  # equal distances: lower index ranks first
    dist, idx = ops.i.three_nn(T(xyz1, dev), T(xyz2, dev))
    od, oi = O.three_nn(xyz1, xyz2)
    assert (N(idx) == oi).all() and (N(dist) == od).all()  # inf fill when m < 3
    dsafe = np.where(np.isfinite(od), od, 1.0).astype(np.float32)
    w = ops.i.three_nn_weights(T(dsafe, dev))
    ow = O.three_nn_weights(dsafe)
    assert (N(w) == ow).all()  # bit-exact: same three operations in the same order, IEEE division
    pts = rng.random((b, m, c), dtype=np.float32)
    p = T(pts, dev).requires_grad_(True)
    out = ops.i.three_interpolate(p, idx, T(ow, dev))
    assert (N(out) == O.three_interpolate(pts, oi, ow)).all()
    go = rng.random((b, n, c), dtype=np.float32)
    out.backward(T(go, dev))
    assert np.allclose(N(p.grad), O.three_interpolate_grad(pts, oi, ow, go), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ 3D IoU / NMS
def test_nms_smoke_known_answer(ops, dev, golden):
    c, g = cases.nms_smoke(), golden("nms_smoke")
    bb, sc, ob = T(c["bboxes"], dev), T(c["scores"], dev), T(c["objectiveness"], dev)
    assert (N(ops.n.NMS3D(bb, sc, ob, 0.5)) == g["keep_050"]).all()
    assert (N(ops.n.NMS3D(bb, sc, ob, torch.tensor(0.25))) == g["keep_025"]).all()
    iou = N(ops.n.iou3d_matrix(bb))
    i3 = 0.8 * float(g["bev_intersection"])
    assert abs(iou[0, 0, 1] - i3 / (1.512 - i3)) < 1e-5


def test_iou_closed_form_known_answers(ops, dev):
    """The device IoU against answers that come from geometry alone (same cases as tests/test_oracle_properties.py): axis-aligned unit
    cubes shifted by t along x -> (1 - t) / (1 + t); a unit cube against itself rotated by pi/4 -> regular octagon 2 (sqrt 2 - 1);
    the reference's smoke pair -> 0.64 - 4 (0.4 sqrt 2 - 0.5)^2."""
    ts = (0.0, 0.125, 0.5, 0.75, 1.5)
    boxes = [cases.corner_box(1, 1, 1)] + [cases.corner_box(1, 1, 1, None, (t, 0, 0)) for t in ts] + [cases.corner_box(1, 1, 1, np.pi / 4)]
    iou = N(ops.n.iou3d_matrix(T(np.array([boxes]).astype(np.float32), dev)))[0]
    for i, t in enumerate(ts):
        assert abs(iou[0, 1 + i] - ((1 - t) / (1 + t) if t < 1 else 0.0)) < 1e-5, t
    oct_area = 2 * (np.sqrt(2.0) - 1)
    assert abs(iou[0, len(ts) + 1] - oct_area / (2 - oct_area)) < 1e-5
    c = cases.nms_smoke()
    i3 = 0.8 * (0.64 - 4 * (0.4 * np.sqrt(2.0) - 0.5) ** 2)
    assert abs(N(ops.n.iou3d_matrix(T(c["bboxes"], dev)))[0, 0, 1] - i3 / (1.512 - i3)) < 1e-5


def test_nms_random_golden(ops, dev, golden):
    c, g = cases.nms_random(), golden("nms_random")
    bb, sc, ob = T(c["bboxes"], dev), T(c["scores"], dev), T(c["objectiveness"], dev)
    iou = N(ops.n.iou3d_matrix(bb))
    assert np.allclose(iou, g["iou"], rtol=0, atol=1e-5, equal_nan=True)
    # keep lists are exact when no pair sits within 1e-5 of the threshold (checked, not assumed)
    for thr, key in [(0.25, "keep_025"), (0.5, "keep_050")]:
        assert not (np.abs(g["iou"] - thr) < 1e-5).any()
        assert (N(ops.n.NMS3D(bb, sc, ob, thr)) == g[key]).all()
        rows, count = ops.n.NMS3D(bb, sc, ob, thr, padded=True)  # no host synchronisation: padded rows + device count
        assert rows.shape == (bb.shape[0] * bb.shape[1], 2) and int(count) == len(g[key])
        assert (N(rows)[:int(count)] == g[key]).all()


def test_nms_full_size_vs_oracle(ops, dev, O):
    """Config-3 size: 8 scenes x 256 proposals."""
    c = cases.nms_random(b=8, n=256, seed=33, room=6.0)
    bb, sc, ob = T(c["bboxes"], dev), T(c["scores"], dev), T(c["objectiveness"], dev)
    keep = N(ops.n.NMS3D(bb, sc, ob, 0.25))
    exp = O.nms3d(c["bboxes"], c["scores"], c["objectiveness"], 0.25)
    iou = np.stack([O.iou3d_matrix(c["bboxes"][s]) for s in range(8)])
    assert not (np.abs(iou - 0.25) < 1e-5).any()  # the seed was chosen so: no pair within rounding of the threshold, the lists are exact
    assert (keep == exp).all()
    assert np.allclose(N(ops.n.iou3d_matrix(bb)), iou, rtol=0, atol=1e-5, equal_nan=True)


@pytest.mark.parametrize("b,n,thr", [(3, 64, 0.25), (2, 65, 0.1), (2, 300, 0.25), (1, 512, 0.3), (2, 513, 0.25), (1, 700, 0.5),
                                     (5, 2100 // 5, 0.35)])
def test_nms_greedy_by_bit_masks_and_by_the_wave_loop_vs_oracle(ops, dev, O, b, n, thr):
    """Both greedy kernels (bit masks in LDS up to 512 boxes per scene, the wave-per-scene loop beyond) and the LDS-staged rank
    (more than one 2048-element chunk at 5 x 420) against the oracle's loop (tf_nms3d.cpp:202-273): identical keep lists."""
    c = cases.nms_random(b=b, n=n, seed=7 * n + b, room=4.0)  # dense: plenty of overlaps
    ob = c["objectiveness"].copy()
    ob[:, ::5] = np.array([1.0, 0.0], np.float32)  # every fifth box is no candidate
    keep = N(ops.n.NMS3D(T(c["bboxes"], dev), T(c["scores"], dev), T(ob, dev), thr))
    exp = O.nms3d(c["bboxes"], c["scores"], ob, thr)
    iou = np.stack([O.iou3d_matrix(c["bboxes"][s]) for s in range(b)])
    assert len(keep) < int((ob[..., 1] > ob[..., 0]).sum())  # something was suppressed
    assert not (np.abs(iou - thr) < 1e-5).any()  # seeds chosen so (an IoU within rounding of the threshold could fall either side)
    assert keep.shape == exp.shape and (keep == exp).all()


def _greedy_on_matrix(M, order, thr):
    """tf_nms3d.cpp:237-262 on a given IoU matrix of one scene: M is read as M[candidate][kept] (suppress_check(candidate, selected))."""
    kept = []
    for c in order:
        if not any(M[c, k] > thr for k in kept):
            kept.append(int(c))
    return kept


@pytest.mark.parametrize("n_pad", [0, 460])
def test_nms_reads_the_pair_matrix_the_way_the_reference_does(ops, dev, n_pad):
    """iou3d_pair(a, b) and (b, a) may differ in the last bit (the first box's footprint is clipped by the second's).  With the
    threshold set between the two orientations of a pair, the decision depends on which one is read: both greedy kernels (bit masks
    up to 512 boxes, wave loop beyond: n_pad pushes the scene over 512) must read [later candidate][earlier box] like the reference."""
    c = cases.nms_random(b=1, n=64, seed=11, room=2.5)  # dense: many overlapping pairs
    boxes = c["bboxes"]
    if n_pad:  # far-away boxes that overlap nothing and score below everything else
        far = np.stack([cases.corner_box(0.3, 0.3, 0.3, None, (100.0 + 2 * i, 0, 0)) for i in range(n_pad)]).astype(np.float32)
        boxes = np.concatenate([boxes, far[None]], 1)
    n = boxes.shape[1]
    M = N(ops.n.iou3d_matrix(T(boxes, dev)))[0]
    asym = [(a, b) for a in range(64) for b in range(64) if a != b and M[a, b] > 0.05 and M[a, b] > M[b, a]]
    assert asym, "no asymmetric pair in this case: pick another seed"
    hits = 0
    for a, b in asym[:6]:
        thr = float(M[b, a])  # M[a, b] > thr is true, M[b, a] > thr is false
        for first, second in ((a, b), (b, a)):
            sc = np.full((1, n), -5.0, np.float32) - np.arange(n, dtype=np.float32)[None] * 1e-3
            sc[0, first], sc[0, second] = 10.0, 9.0
            obj = np.tile(np.array([1.0, 0.0], np.float32), (1, n, 1))
            obj[0, [first, second]] = np.array([0.0, 1.0], np.float32)
            if n_pad:
                obj[0, 64:] = np.array([0.0, 1.0], np.float32)
                sc[0, 64:] = -50.0 - np.arange(n_pad, dtype=np.float32)
            order = [i for i in np.argsort(-sc[0], kind="stable") if obj[0, i, 1] > obj[0, i, 0]]
            exp = _greedy_on_matrix(M, order, thr)
            got = N(ops.n.NMS3D(T(boxes, dev), T(sc, dev), T(obj, dev), thr))
            assert got[:, 1].tolist() == exp, (a, b, first)
            hits += (second in exp) == bool(M[first, second] > thr)  # reading [earlier][later] instead would have decided differently
    assert hits > 0


def test_nms_nan_scores_are_ordered_last_not_out_of_bounds(ops, dev, O):
    """A diverged model emits NaN logits.  The visit order must stay a permutation of the candidates (NaN ranks with -inf,
    ties by flat index): before, every NaN candidate took rank 0 and the tail of the order buffer was uninitialised memory
    used as an index.  Expected result = the oracle's on the same boxes with NaN replaced by -inf."""
    c = cases.nms_random(b=2, n=64, seed=5)
    sc = c["scores"].copy()
    sc[0, 3] = sc[0, 40] = sc[1, 7] = sc[1, 8] = sc[1, 63] = np.nan
    allobj = np.tile(np.array([0.0, 1.0], np.float32), (2, 64, 1))
    for _ in range(3):
        keep = N(ops.n.NMS3D(T(c["bboxes"], dev), T(sc, dev), T(allobj, dev), 0.25))
        assert len(keep) and (keep[:, 0] >= 0).all() and (keep[:, 0] < 2).all() and (keep[:, 1] >= 0).all() and (keep[:, 1] < 64).all()
        assert len({(int(a), int(b)) for a, b in keep}) == len(keep)
        finite = [(int(a), int(b)) for a, b in keep if np.isfinite(sc[a, b])]
        exp = O.nms3d(c["bboxes"], np.where(np.isnan(sc), -np.inf, sc).astype(np.float32), allobj, 0.25)
        exp_f = [(int(a), int(b)) for a, b in exp if np.isfinite(sc[a, b])]
        assert finite == exp_f  # the finite-score part of the visit order is untouched, NaN boxes come after it
        assert [tuple(r) for r in keep.tolist()][:len(finite)] == finite


def test_nms_edge_cases(ops, dev):
    c = cases.nms_random(b=2, n=16, seed=2)
    bb, sc = T(c["bboxes"], dev), T(c["scores"], dev)
    none = torch.zeros(2, 16, 2, device=dev)
    assert tuple(ops.n.NMS3D(bb, sc, none, 0.25).shape) == (0, 2)  # no candidate
    allobj = torch.tensor([0.0, 1.0], device=dev).expand(2, 16, 2).contiguous()
    assert len(ops.n.NMS3D(bb, sc, allobj, 1.0)) == 32  # nothing can exceed IoU 1
    from votenet_amd import InvalidArgumentError
    with pytest.raises(InvalidArgumentError):
        ops.n.NMS3D(bb, sc, allobj, 1.5)  # tf_nms3d.cpp:300
    with pytest.raises(InvalidArgumentError):
        ops.n.NMS3D(bb[:, :, :4], sc, allobj, 0.5)  # tf_nms3d.cpp:287


def test_argument_errors(ops, dev):
    from votenet_amd import InvalidArgumentError
    x = torch.zeros(1, 16, 3, device=dev)
    with pytest.raises(InvalidArgumentError):
        ops.s.farthest_point_sample(0, x)  # tf_sampling.cpp:99
    with pytest.raises(InvalidArgumentError):
        ops.g.query_ball_point(-1.0, 4, x, x)  # tf_grouping.cpp:71
    with pytest.raises(InvalidArgumentError):
        ops.g.query_ball_point(0.1, 0, x, x)  # tf_grouping.cpp:74
    with pytest.raises(InvalidArgumentError):
        ops.s.farthest_point_sample(4, torch.zeros(1, 16, 4, device=dev))  # tf_sampling.cpp:105


def test_a_cloud_rewritten_through_its_raw_pointer_needs_forget_index(ops, dev, O):
    """The remembered spatial index is trusted while the tensor's version counter is unchanged; a kernel that writes the cloud through
    its raw pointer bumps none.  forget_index() is the documented way out (the library's own writers allocate fresh outputs)."""
    from votenet_amd import mlp as M, tf_sampling as S
    rng = np.random.default_rng(4)
    a = rng.random((1, 6000, 3), dtype=np.float32) * 4
    b = rng.random((1, 6000, 3), dtype=np.float32) * 4
    x, other = T(a, dev), T(b, dev)
    q = T(rng.random((1, 50, 3), dtype=np.float32) * 4, dev)
    S.clear_index_cache()
    S.farthest_point_sample(64, x)                    # leaves the index of cloud `a` behind
    M.row_segments(6000, [(x.view(6000, 3), other.view(6000, 3), None)])  # x := b, through the raw pointer: no version bump
    assert S.cached_index(x) is not None              # ... so the stale index would still be used
    S.forget_index(x)
    assert S.cached_index(x) is None
    idx, cnt = ops.g.query_ball_point(0.3, 16, x, q)  # builds a fresh index of what x holds now
    oi, oc = O.query_ball_point(0.3, 16, b, N(q))
    assert (N(idx) == oi).all() and (N(cnt) == oc).all()
