import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from votenet_amd import synth
from votenet_amd.model import VoteNetHotPath
dev = torch.device("cuda:0")
net = VoteNetHotPath(dev, seed=0)
cot = net.make_cotangents(8, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(4)]
net.init_optimizer(1e-3)
t0 = time.time()
for i in range(300):
    out = net.train_step(xs[i % 4], cot, 1)
    if i % 50 == 49:
        torch.cuda.synchronize()
        print(i + 1, "steps, %.1f s, mem %.2f GB (peak %.2f), |param| %.4f, finite %s" % (
            time.time() - t0, torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9,
            float(net.store.flat.abs().mean()), bool(torch.isfinite(net.store.flat).all() and torch.isfinite(out["proposals_output"]).all())))
