"""CPU: liboracle_omp.so (the oracle's sources built with -fopenmp, bench.py's all-core cpu_baseline figure) gives
bit-identical results to the single-thread liboracle.so -- every output element is computed by one thread in the same order."""
import numpy as np


def test_omp_build_is_bit_identical(O):
    rng = np.random.default_rng(0)
    xyz = rng.random((2, 3000, 3), dtype=np.float32)
    feat = rng.random((2, 3000, 8), dtype=np.float32)
    w = rng.normal(size=(11, 16)).astype(np.float32)

    def run():
        f = O.farthest_point_sample(300, xyz)
        c = O.gather_point(xyz, f)
        idx, cnt = O.query_ball_point(0.15, 16, xyz, c)
        g = O.group_concat(xyz, c, feat, idx).reshape(-1, 11)
        z = O.linear(g, w, np.ones(16, np.float32))
        mean, var = O.bn_stats(z)
        y = O.max_over_k(O.bn_relu(z, mean, var, np.ones(16, np.float32), np.zeros(16, np.float32)), 16)
        d, i3 = O.three_nn(xyz, c)
        itp = O.three_interpolate(y.reshape(2, 300, 16), i3, O.three_nn_weights(d))
        return f, idx, cnt, z, mean, var, y, d, i3, itp
    one = run()
    prev = O.set_threads(4)
    try:
        assert O.lib() is not None and O._THREADS == 4
        four = run()
    finally:
        O.set_threads(prev)
    for a, b in zip(one, four):
        assert a.dtype == b.dtype and np.array_equal(a, b)
