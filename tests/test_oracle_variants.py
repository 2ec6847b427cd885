"""CPU: the oracle of the ops the reference ships but VoteNet never reaches (SelectionSort / kNN, ProbSample;
oracle/oracle_variants.c) against the reference's compiled CPU twin (golden vectors + oracle/_ref when present) and against
a literal simulation of the CUDA kernel's index loops."""
import hashlib

import numpy as np
import pytest

import cases

F = np.float32


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_selection_sort_golden_from_the_reference_twin(O, golden):
    g = golden("selection_sort")
    assert str(g["source"]) == "ref"
    for name, (dist, k) in cases.selection_sort_cases().items():
        outi, val = O.select_top_k(k, dist)
        if name + "_idx" in g:
            assert np.array_equal(outi, g[name + "_idx"]) and np.array_equal(val, g[name + "_val"]), name
        else:
            assert sha(outi) == str(g[name + "_idx_sha"]) and sha(val) == str(g[name + "_val_sha"]), name
            assert np.array_equal(outi[0, :2, :k], g[name + "_idx_head"])
        # properties: a permutation of the row; the first k ascending and no larger than anything after them
        assert np.array_equal(np.sort(outi, -1), np.broadcast_to(np.arange(dist.shape[2]), dist.shape))
        assert np.array_equal(np.take_along_axis(dist, outi.astype(np.int64), -1), val)
        assert np.all(np.diff(val[..., :k], axis=-1) >= 0)
        if k < dist.shape[2]:
            assert np.all(val[..., k - 1:k] <= val[..., k:])
    tw = O.select_top_k(3, cases.selection_sort_cases()["twin_main"][0])
    assert tw[0][0, 0].tolist() == [3, 2, 1, 0] and tw[1][0, 0].tolist() == [7.0, 8.0, 9.0, 10.0]


def test_selection_sort_against_live_reference_twin(O):
    if O.ref("selection_sort") is None:
        pytest.skip("oracle/_ref not built (no reference tree on this machine)")
    rs = np.random.RandomState(0)
    for b, m, n, k in ((2, 5, 40, 7), (1, 3, 9, 9), (3, 2, 100, 1), (1, 1, 1, 1)):
        d = np.round(rs.random_sample((b, m, n)) * 8).astype(F) / 8  # plenty of ties
        a, r = O.select_top_k(k, d), O.ref_select_top_k(k, d)
        assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])


def test_knn_point_is_the_k_nearest():
    from oracle import oracle as O
    rs = np.random.RandomState(1)
    x1, x2 = rs.random_sample((2, 300, 3)).astype(F), rs.random_sample((2, 17, 3)).astype(F)
    val, idx = O.knn_point(8, x1, x2)
    d = ((x1[:, None].astype(np.float64) - x2[:, :, None]) ** 2).sum(-1)
    exp = np.argsort(d, -1, kind="stable")[..., :8]
    assert np.array_equal(idx, exp) and np.allclose(val, np.take_along_axis(d, exp, -1), rtol=1e-6)


def _cumsum_kernel_literal(x):
    """tf_sampling_g.cu:7-86 simulated index for index (one batch row; threads of a phase run one after another, which is
    equivalent: phases are separated by barriers and touch disjoint elements)."""
    n = len(x)
    BS, PL = 2048, 5
    pad = lambda i: i + (i >> PL)
    out = np.zeros(n, F)
    run, run2 = F(0), F(0)
    for j in range(0, n, BS * 4):
        n24_i = min(n - j, BS * 4)
        n24 = (n24_i + 3) & ~3
        n2 = n24 >> 2
        buf4 = np.zeros(n24, F)
        buf = np.zeros(BS + (BS >> PL) + 2, F)
        for k in range(0, n24_i, 4):
            if k + 3 < n24_i:
                v1, v2, v3, v4 = (F(x[j + k + t]) for t in range(4))
                v2 = F(v2 + v1)
                v4 = F(v4 + v3)
                v3 = F(v3 + v2)
                v4 = F(v4 + v2)
                buf4[k:k + 4] = (v1, v2, v3, v4)
                buf[pad(k >> 2)] = v4
            else:
                v = F(0)
                for k2 in range(k, n24_i):
                    v = F(v + x[j + k2])
                    buf4[k2] = v
                buf4[n24_i:n24] = v
                buf[pad(k >> 2)] = v
        u = 0
        while (2 << u) <= n2:
            for k in range(n2 >> (u + 1)):
                i1, i2 = (((k << 1) + 2) << u) - 1, (((k << 1) + 1) << u) - 1
                buf[pad(i1)] = F(buf[pad(i1)] + buf[pad(i2)])
            u += 1
        u -= 1
        while u >= 0:
            for k in range((n2 - (1 << u)) >> (u + 1)):
                i1, i2 = (((k << 1) + 3) << u) - 1, (((k << 1) + 2) << u) - 1
                buf[pad(i1)] = F(buf[pad(i1)] + buf[pad(i2)])
            u -= 1
        for k in range(4, n24, 4):
            buf4[k:k + 4] = (buf4[k:k + 4] + buf[pad((k >> 2) - 1)]).astype(F)
        out[j:j + n24_i] = (buf4[:n24_i] + run).astype(F)
        t = F(buf[pad(n2 - 1)] + run2)
        r2 = F(run + t)
        run2 = F(t - F(r2 - run))
        run = r2
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 100, 1023, 1024, 1025, 4097, 8191, 8192, 8193, 8198, 17000])
def test_cumsum_recurrence_equals_the_kernels_scan_tree(O, n):
    rs = np.random.RandomState(n)
    x = (rs.random_sample(n).astype(F) * rs.choice([1e-3, 1.0, 50.0], n).astype(F)).astype(F)
    got = O.cumsum(x[None])[0]
    assert np.array_equal(got, _cumsum_kernel_literal(x))
    assert np.allclose(got, np.cumsum(x.astype(np.float64)), rtol=2e-6)


def test_prob_sample_golden_and_inverse_cdf_property(O, golden):
    g = golden("prob_sample")
    for name, (p, r) in cases.prob_sample_cases().items():
        out = O.prob_sample(p, r)
        assert np.array_equal(out, g[name]), name
        cs = O.cumsum(p)
        assert sha(cs) == str(g[name + "_cumsum_sha"])
        q = (r * cs[:, -1:]).astype(F)
        hit = np.take_along_axis(cs, out.astype(np.int64), 1)
        prev = np.take_along_axis(cs, np.maximum(out - 1, 0).astype(np.int64), 1)
        assert np.all((hit >= q) | (out == p.shape[1] - 1)) and np.all((prev < q) | (out == 0)), name
    areas, r = cases.prob_sample_cases()["triangles"]
    freq = np.bincount(O.prob_sample(areas, r)[0], minlength=5) / 8192.0
    assert np.allclose(freq, areas[0] / areas[0].sum(), atol=0.02)  # categories drawn in proportion to their weight
