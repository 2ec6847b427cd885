"""kNN grouping and ProbSample (ops VoteNet never reaches): the fused knn_point against the reference's data flow (a
materialised (b,m,n) distance tensor through select_top_k), and prob_sample."""
import sys, time, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tools")]
import numpy as np
from votenet_amd import tf_grouping, tf_sampling
from oracle import oracle as O
from bench_mlp_util import timeit
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
for b, n, m, k in ((32, 512, 128, 64), (8, 2048, 1024, 64), (8, 20480, 2048, 16)):
    x1n, x2n = rs.random_sample((b, n, 3)).astype(np.float32), rs.random_sample((b, m, 3)).astype(np.float32)
    x1, x2 = torch.from_numpy(x1n).to(dev), torch.from_numpy(x2n).to(dev)
    fused = timeit(lambda: tf_grouping.knn_point(k, x1, x2), it=5)
    if b * m * n * 12 < 8e9:
        dist = ((x1[:, None] - x2[:, :, None]) ** 2).sum(-1).contiguous()
        mat = timeit(lambda: tf_grouping.select_top_k(k, dist), it=5)
    else:
        mat = float("nan")
    t = time.perf_counter(); O.knn_point(k, x1n[:1], x2n[:1, :max(1, m // 16)]); cpu = (time.perf_counter() - t) * b * 16
    print("knn_point b=%d n=%d m=%d k=%d: fused %.3f ms | select_top_k on a materialised (b,m,n) tensor %.3f ms (+ forming it) | "
          "oracle, one core (scaled from a 1/(16b) sample) %.0f ms" % (b, n, m, k, fused, mat, cpu * 1e3))
p = torch.rand(8, 100000, device=dev); r = torch.rand(8, 8192, device=dev)
print("prob_sample 8 x 100000 categories, 8192 draws: %.3f ms" % timeit(lambda: tf_sampling.prob_sample(p, r), it=20))
