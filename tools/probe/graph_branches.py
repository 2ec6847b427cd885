"""Do the branches of a two-stream capture run CONCURRENTLY when the graph is replayed (ROCm 7 / torch 2.10)?  Two farthest-point samplings
of 8 x 20480 points (8 workgroups for 1.6 ms each: they can only overlap, never fill the GPU) on one stream and on two, eager and replayed."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import synth, tf_sampling
dev = torch.device("cuda:0")
xa = torch.from_numpy(synth.room_batch(8, 20480, 1)).to(dev)
xb = torch.from_numpy(synth.room_batch(8, 20480, 2)).to(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
keep = []
def two(streamed):
    if streamed:
        ev = torch.cuda.Event(); ev.record(); s2.wait_event(ev)
        keep.append(tf_sampling.farthest_point_sample(2048, xa))
        with torch.cuda.stream(s2):
            keep.append(tf_sampling.farthest_point_sample(2048, xb))
        ev2 = torch.cuda.Event(); ev2.record(s2); torch.cuda.current_stream().wait_event(ev2)
    else:
        keep.append(tf_sampling.farthest_point_sample(2048, xa)); keep.append(tf_sampling.farthest_point_sample(2048, xb))
def timeit(fn, n=10):
    torch.cuda.synchronize(); t = torch.cuda.Event(enable_timing=True); u = torch.cuda.Event(enable_timing=True)
    t.record()
    for _ in range(n): fn()
    u.record(); torch.cuda.synchronize(); keep.clear(); return t.elapsed_time(u) / n
with torch.cuda.stream(s1):
    two(True); two(False); torch.cuda.synchronize()
    print("eager: one stream %.3f ms, two streams %.3f ms" % (timeit(lambda: two(False)), timeit(lambda: two(True))))
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s1, capture_error_mode="thread_local"): two(False)
    with torch.cuda.graph(g2, stream=s1, capture_error_mode="thread_local"): two(True)
    print("graph replay: serial capture %.3f ms, two-stream capture %.3f ms" % (timeit(g1.replay), timeit(g2.replay)))
