"""GPU: the tf_ops VoteNet never reaches -- select_top_k / knn_point, prob_sample -- through the C ABI
against oracle/oracle_variants.c (SelectionSort pinned by the reference's CPU twin, tests/golden/selection_sort.npz), and
the kNN / multi-scale-grouping variants of the SA module against the oracle composed the same way.
Bar: bit-exact (indices, copied distances, the float running sum of ProbSample)."""
import hashlib

import numpy as np
import pytest
import torch

import cases
from test_gpu_model import N, oracle_chain

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_select_top_k_golden(hiplib, dev, golden):
    from votenet_amd import tf_grouping
    g = golden("selection_sort")
    for name, (dist, k) in cases.selection_sort_cases().items():
        outi, out = tf_grouping.select_top_k(k, T(dist, dev))
        outi, out = N(outi), N(out)
        if name + "_idx" in g:
            assert np.array_equal(outi, g[name + "_idx"]) and np.array_equal(out, g[name + "_val"]), name
        else:
            assert sha(outi) == str(g[name + "_idx_sha"]) and sha(out) == str(g[name + "_val_sha"]), name


@pytest.mark.parametrize("b,m,n,k", [(2, 9, 1000, 17), (1, 3, 2048, 5), (2, 3, 2049, 40), (1, 3, 16384, 5), (1, 2, 16385, 4), (1, 2, 40000, 3), (3, 5, 64, 64),
                                     (1, 1, 1, 1), (2, 300, 65, 2)])
def test_select_top_k_vs_oracle_with_ties_and_special_values(hiplib, dev, O, b, m, n, k):
    from votenet_amd import tf_grouping
    rs = np.random.RandomState(n + k)
    d = (np.round(rs.random_sample((b, m, n)) * 64) / 64).astype(np.float32)  # many exact ties
    d[0, 0, rs.randint(0, n, 3)] = 0.0
    d[0, 0, rs.randint(0, n)] = -0.0
    if n > 10:
        d[-1, -1, 1] = np.inf
        d[-1, -1, 3] = np.nan  # never '<' anything: stays where the swaps leave it
    outi, out = tf_grouping.select_top_k(k, T(d, dev))
    ei, eo = O.select_top_k(k, d)
    assert np.array_equal(N(outi), ei)
    assert np.array_equal(N(out).view(np.uint32), eo.view(np.uint32))


def test_knn_point_fused_vs_oracle(hiplib, dev, O):
    from votenet_amd import tf_grouping
    c = cases.grouping_demo()  # tf_grouping.py:75-90 runs exactly this with knn=True, k = 64
    val, idx = tf_grouping.knn_point(64, T(c["xyz1"], dev), T(c["xyz2"], dev))
    ev, ei = O.knn_point(64, c["xyz1"], c["xyz2"])
    assert np.array_equal(N(idx), ei) and np.array_equal(N(val), ev)
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "selection_sort.npz"))
    assert np.array_equal(N(idx)[0, :2], g["knn_demo_idx_head"])  # the reference twin's own picks
    rs = np.random.RandomState(2)
    for b, n, m, cc, k in ((2, 700, 33, 3, 16), (1, 17000, 5, 3, 8), (1, 21000, 3, 3, 4), (2, 2049, 9, 3, 33), (2, 64, 64, 5, 64), (1, 50, 7, 1, 3)):
        x1 = (np.round(rs.random_sample((b, n, cc)) * 32) / 32).astype(np.float32)  # grid points: tied distances
        x2 = (np.round(rs.random_sample((b, m, cc)) * 32) / 32).astype(np.float32)
        val, idx = tf_grouping.knn_point(k, T(x1, dev), T(x2, dev))
        ev, ei = O.knn_point(k, x1, x2)
        assert np.array_equal(N(idx), ei) and np.array_equal(N(val), ev), (b, n, m, cc, k)


def test_prob_sample_golden_and_oracle(hiplib, dev, O, golden):
    from votenet_amd import tf_sampling
    g = golden("prob_sample")
    for name, (p, r) in cases.prob_sample_cases().items():
        out = N(tf_sampling.prob_sample(T(p, dev), T(r, dev)))
        assert np.array_equal(out, g[name]), name
    rs = np.random.RandomState(9)
    p = (rs.random_sample((3, 30000)) * rs.choice([1e-4, 1.0, 30.0], (3, 30000))).astype(np.float32)
    r = rs.random_sample((3, 5000)).astype(np.float32)
    r[:, 0], r[:, 1] = 0.0, np.float32(1.0) - np.float32(2 ** -24)
    assert np.array_equal(N(tf_sampling.prob_sample(T(p, dev), T(r, dev))), O.prob_sample(p, r))


def test_variant_argument_errors(hiplib, dev):
    from votenet_amd import tf_grouping, tf_sampling, _lib
    d = torch.zeros(1, 2, 8, device=dev)
    for k in (0, 9):
        with pytest.raises(_lib.InvalidArgumentError):
            tf_grouping.select_top_k(k, d)
    with pytest.raises(_lib.InvalidArgumentError):
        tf_grouping.select_top_k(2, d[0])
    with pytest.raises(_lib.InvalidArgumentError):
        tf_grouping.knn_point(3, torch.zeros(1, 8, 3, device=dev), torch.zeros(1, 2, 2, device=dev))
    with pytest.raises(_lib.InvalidArgumentError):
        tf_sampling.prob_sample(torch.ones(2, 5, device=dev), torch.zeros(3, 4, device=dev))


def _oracle_branch(O, mod, xyz, pts, new_xyz, idx):
    g = O.group_concat(xyz, new_xyz, pts, idx).reshape(-1, 3 + pts.shape[2])
    return oracle_chain(O, g, mod.mlp, mod.nsample).reshape(xyz.shape[0], mod.npoint, -1)


def test_sa_module_knn_and_msg_vs_oracle_and_autograd(hiplib, dev, O):
    from votenet_amd import pointnet2 as P
    from test_gpu_backward import ref_sa, relerr
    rs = np.random.RandomState(0)
    xyz = rs.random_sample((2, 600, 3)).astype(np.float32)
    pts = rs.normal(size=(2, 600, 8)).astype(np.float32)
    store = P.ParamStore(dev)
    knn = P.SAModule(store, "knn", 64, None, 16, 8, [32, 64], knn=True)
    msg = P.SAModuleMSG(store, "msg", 64, [0.1, 0.2, 0.4], [8, 16, 32], 8, [[16, 32], [32, 64], [32, 128]])
    store.materialize(5)
    assert "msg/conv2_1/W" in store.views and store["msg/conv1_0/W"].shape == (11, 32)
    fidx = O.farthest_point_sample(64, xyz)
    new_xyz = O.gather_point(xyz, fidx)
    tol = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    # ---- knn=True (utils.py:46-47)
    tape = []
    nx, out, idx = knn.forward(T(xyz, dev), T(pts, dev), tape=tape)
    _, eidx = O.knn_point(16, xyz, new_xyz)
    assert np.array_equal(N(nx), new_xyz) and np.array_equal(N(idx), eidx)
    assert tol(N(out), _oracle_branch(O, knn, xyz, pts, new_xyz, eidx)) < 1e-4
    # ---- multi-scale grouping (utils.py:161-201)
    tape = []
    nx, out = msg.forward(T(xyz, dev), T(pts, dev), tape=tape)
    exp = np.concatenate([_oracle_branch(O, sc, xyz, pts, new_xyz, O.query_ball_point(sc.radius, sc.nsample, xyz, new_xyz)[0])
                          for sc in msg.scales], -1)
    assert out.shape == (2, 64, 32 + 64 + 128) and np.array_equal(N(nx), new_xyz)
    assert tol(N(out), exp) < 1e-4
    # backward of the concatenation against float64 autograd over the same neighbour lists
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    store.grad.zero_()
    d_feat = msg.backward(tape[0], gout)
    P.wgrad_join()
    params = {k: v.detach().double().clone().requires_grad_(True) for k, v in store.views.items()}
    xd, pd = T(xyz, dev).double(), T(pts, dev).double().requires_grad_(True)
    y = torch.cat([ref_sa(sc, params, xd, pd, sub)[1] for sc, sub in zip(msg.scales, tape[0]["subs"])], -1)
    (y * gout.double()).sum().backward()
    assert relerr(d_feat.double(), pd.grad) < 1e-4
    for name in store.views:
        if name.startswith("msg/") and not name.endswith("/b"):
            assert relerr(store.g(name).double(), params[name].grad) < 1e-4, name
    w = store["msg/conv0_0/W"]
    assert torch.equal(P.SAModuleMSG.from_reference_rows(P.SAModuleMSG.reference_rows(w)), w)
