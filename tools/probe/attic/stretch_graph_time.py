"""The stretch of a train step (model.StretchGraph) alone on the GPU: launch by launch vs replayed, with and without the weight-gradient
side branch in the capture."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import synth, loss as VL, mlp as M, model as VM, pointnet2 as P
dev = torch.device("cuda:0")
net = VM.VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
VM.STRETCH_GRAPH = False
for i in range(4):
    net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
torch.cuda.synchronize()
# the inputs of one step's stretch
net.store.refresh_split(); net.store.refresh_transposes()
M.arena_begin(dev)
tape = []
lv, g = net.backbone_levels(xs[0], tape)
ins = net._stretch_inputs(lv, g, gts[0])
if net._wgrad_stream is None:
    net._wgrad_stream = torch.cuda.Stream(device=dev)

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = torch.cuda.Event(enable_timing=True); u = torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter(); t.record()
    for _ in range(n): fn()
    u.record(); h1 = time.perf_counter(); torch.cuda.synchronize()
    return t.elapsed_time(u) / n, (h1 - h0) / n * 1e3

def eager(ws):
    def f():
        net.store.grad.zero_()
        M._StatsArena.off = M._StatsArena.off32 = 0
        M._StatsArena.buf.zero_()
        net._stretch_body(ins, tape, ws)
    return f
a = M._StatsArena
off0, want0 = a.off, a.want32
net._stretch_body(ins, tape, net._wgrad_stream)
demand = (a.off - off0 + 1024, a.want32 - want0 + 4096)
print("arena demand of the stretch: %d doubles, %d floats" % demand)
print("eager, weight gradients on their stream: GPU %.3f ms, host %.3f ms" % timeit(eager(net._wgrad_stream)))
print("eager, one stream:                        GPU %.3f ms, host %.3f ms" % timeit(eager(None)))
for seg in (True, False):
    VM.STRETCH_SEGMENTS = seg
    sg = VM.StretchGraph(net, ins, tape, demand)
    def rp():
        net.store.grad.zero_(); M._StatsArena.off = M._StatsArena.off32 = 0; M._StatsArena.buf.zero_()
        sg.replay(ins, net._wgrad_stream); P.wgrad_join()
    print("graph replay, segments %s (%d graphs, %d thunks): GPU %.3f ms, host %.3f ms (arena used %s)" % (seg, len(sg.segments), sum(len(t) for _, t in sg.segments), *timeit(rp), sg.arena_used))
    print("   copy launch alone: GPU %.3f ms" % timeit(lambda: sg.copy_inputs(ins))[0])
M.arena_end()
