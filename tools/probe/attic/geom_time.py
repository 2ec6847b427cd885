"""sa1 geometry alone on the GPU: spatial index + FPS (one call), ball query over the index, and the north star's pair figure."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + "/tools"]
import torch
from votenet_amd import synth, tf_grouping as G, tf_sampling as S
from bench_legs import gpu_ms
dev = torch.device("cuda:0")
B, n, m, K = 8, 20480, 2048, 64
alg_f = B * (m - 1) * n * 16 + B * n * 12 + B * m * 4
alg_b = B * m * n * 12 + B * m * (K + 1) * 4
for kind, x in (("room", synth.room_batch(B, n, 1000)), ("uniform", synth.uniform_batch(B, n, 1000))):
    x = torch.from_numpy(x).to(dev)
    c = S.gather_point(x, S.farthest_point_sample(m, x))
    tf = gpu_ms(lambda: S.farthest_point_sample(m, x), it=20)
    ti = gpu_ms(lambda: (S._INDEX_CACHE.clear(), S.spatial_index(x)), it=20)
    S.farthest_point_sample(m, x)
    tb = gpu_ms(lambda: G.query_ball_point(0.2, K, x, c), it=20)
    print("%-8s index %.4f  fps (index + sampling) %.4f  ball query %.4f ms | fps frac %.3f  fps+bq frac %.3f"
          % (kind, ti, tf, tb, alg_f / tf / 1e6 / 8000, (alg_f + alg_b) / (tf + tb) / 1e6 / 8000))
