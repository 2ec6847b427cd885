import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
b, n = 2, 4096
xs = [torch.from_numpy(synth.room_batch(b, n, 90 + i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(b, n, 90 + i), dev) for i in range(3)]
print("gt shapes", [tuple(g["bboxes_xyz"].shape) for g in gts], {k: v.dtype for k, v in gts[0].items()})
nets = [VM.VoteNetHotPath(dev, seed=5, npoints=(512, 256, 128, 64)) for _ in range(3)]
flags = [False, False, True]
for net in nets:
    net.init_optimizer(lr=1e-3); net._ema_state()
for i in range(10):
    for net in nets[1:]:
        net.store.flat.copy_(nets[0].store.flat); net.store.params_changed()
        net._m.copy_(nets[0]._m); net._v.copy_(nets[0]._v); net._ema_flat.copy_(nets[0]._ema_flat)
    res = []
    for net, flag in zip(nets, flags):
        VM.STRETCH_GRAPH = flag
        out = net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
        torch.cuda.synchronize()
        res.append((net.last_losses.clone(), net.store.grad.clone(), {k: v.clone() for k, v in out.items()}))
    l0, l1, l2 = res[0][0], res[1][0], res[2][0]
    print(i, "eager-eager loss equal", torch.equal(l0, l1), "eager-graph", torch.equal(l0, l2),
          "out equal", all(torch.equal(res[0][2][k], res[2][2][k]) for k in res[0][2]),
          "grad e-e %.2e e-g %.2e" % (float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max()), float((res[0][1] - res[2][1]).abs().max() / res[0][1].abs().max())))
    if not torch.equal(l0, l2):
        print("   ", l0.tolist()); print("   ", l2.tolist())
