"""Decomposition of the pipelined train step: (A) as is, (B) no weight-gradient launch at all, (C) B with the geometry given (no chain beside
the step), (D) geometry given but weight gradients on -- wrong numbers, right timing."""
import os, sys, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import importlib.util as _iu
_s = _iu.spec_from_file_location("hp", os.path.join(R, "votenet_amd", "hostpin.py")); hostpin = _iu.module_from_spec(_s); _s.loader.exec_module(hostpin); hostpin.pin(0)
import torch
from votenet_amd import loss as VL, model as VM, synth, mlp as M, pointnet2 as P
dev = torch.device("cuda:0")
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
real_on = P.on_wgrad_stream
def run(k, net, given=None):
    for i in range(k):
        x = xs[i % 3]
        if given is not None:
            net._prefetched[id(x)] = given[id(x)]
            net.train_step(x, gt=gts[i % 3])
        else:
            net.train_step(x, gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
def measure(name, net, given=None):
    run(8, net, given); torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter(); run(40, net, given); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
    print("%-70s %.3f ms per step" % (name, dt / 40 * 1e3), flush=True)
net = VM.VoteNetHotPath(dev, seed=0)
VM.GEOMETRY_GRAPHS = False
net2 = VM.VoteNetHotPath(dev, seed=0)
for x in xs:
    net2.prefetch_geometry(x)
torch.cuda.synchronize()
given = dict(net2._prefetched)
VM.GEOMETRY_GRAPHS = True
for rep in range(2):
    P.on_wgrad_stream = real_on; net.drop_graphs()
    measure("A  as is", net)
    P.on_wgrad_stream = lambda fn, *t: None; net.drop_graphs()
    measure("B  no weight-gradient launch at all", net)
    VM.GEOMETRY_GRAPHS = False
    P.on_wgrad_stream = lambda fn, *t: None; net2.drop_graphs(); net2._prefetched = {}
    measure("C  no weight gradients, geometry given (nothing beside the chain)", net2, given)
    P.on_wgrad_stream = real_on; net2.drop_graphs()
    measure("D  weight gradients on, geometry given", net2, given)
    VM.GEOMETRY_GRAPHS = True
