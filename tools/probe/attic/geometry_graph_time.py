"""The coordinate-only chain of one batch alone on an idle GPU: enqueued launch by launch against replayed as one HIP graph
(model.GeometryGraph) -- GPU time from the start event to the chain's last kernel, and the host time of the call."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import model as VM, synth
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(4)]
for graphs in (False, True, False, True):
    VM.GEOMETRY_GRAPHS = graphs
    gpu, host = [], []
    for i in range(12):
        x = xs[i % 4]
        net.__dict__.setdefault("_prefetched", {}).clear()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t0 = time.perf_counter()
        net.prefetch_geometry(x)
        host.append(time.perf_counter() - t0)
        ev = net._prefetched[id(x)][3]["fp"]
        torch.cuda.current_stream().wait_event(ev)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1))
    gpu, host = sorted(gpu[4:]), sorted(host[4:])
    print("graphs %-5s: chain %.3f ms on the GPU (min %.3f), %.3f ms of host time per call" % (graphs, gpu[len(gpu) // 2], gpu[0], host[len(host) // 2] * 1e3), flush=True)
