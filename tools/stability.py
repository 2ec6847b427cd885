import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from votenet_amd import synth
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
from votenet_amd import loss as VL
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(4)]
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(4)]
net.init_optimizer(1e-3)
t0 = time.time()
for i in range(300):
    out = net.train_step(xs[i % 4], gt=gts[i % 4], next_x=xs[(i + 1) % 4])
    if i % 50 == 49:
        torch.cuda.synchronize()
        print(i + 1, "steps, cost %.3f (pos %d), %.1f s, mem %.2f GB (peak %.2f), |param| %.4f, finite %s" % (
            float(net.last_losses[0]), int(net.last_losses[10]),
            time.time() - t0, torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9,
            float(net.store.flat.abs().mean()), bool(torch.isfinite(net.store.flat).all() and torch.isfinite(out["proposals_output"]).all())))
