"""Train step with the side streams at different HIP priorities, and the chain itself on a high-priority stream (scratch, GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch, gc
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range())
B, n = 8, 20480
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000, 900000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000, 900000)]
def trial(side, wgrad, main_pri):
    VM.SIDE_PRIORITY, VM.WGRAD_PRIORITY = side, wgrad
    net = VM.VoteNetHotPath(dev, seed=0)
    ms = torch.cuda.Stream(device=dev, priority=main_pri) if main_pri is not None else torch.cuda.current_stream()
    def run(k):
        with torch.cuda.stream(ms):
            for i in range(k):
                net.train_step(xs[i % 3], gt=gts[i % 3], next_x=[xs[(i + 1) % 3]])
    run(8); torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0; gc.enable()
    print("side %r wgrad %r main %r: %.3f ms per step" % (side, wgrad, main_pri, dt / 40 * 1e3), flush=True)
lo, hi = torch.cuda.Stream.priority_range()
for rep in range(2):
    trial(0, 0, None)
    trial(0, 0, hi)
    trial(lo, lo, hi)
    trial(0, lo, hi)
    trial(lo, 0, hi)
