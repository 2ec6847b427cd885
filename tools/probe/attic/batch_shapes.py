import sys; sys.path.insert(0, '/root/repo')
import torch
from votenet_amd import loss as VL, model as VM, synth
dev = torch.device("cuda:0")
for b, n in ((1, 20480), (3, 20480), (16, 20480), (2, 12000), (4, 40000)):
    x = torch.from_numpy(synth.room_batch(b, n, 5)).to(dev); gt = VL.gt_to_device(synth.room_gt(b, n, 5), dev)
    net = VM.VoteNetHotPath(dev, seed=1)
    net.init_optimizer(1e-3)
    for _ in range(3):
        net.train_step(x, gt=gt, next_x=[x])
    torch.cuda.synchronize()
    kinds = None
    tape = []; net.forward(x, tape)
    kinds = [t["recs"][0]["kind"] for t in tape if t.get("op") == "sa"]
    print(b, n, "cost %.3f" % float(net.last_losses[0]), "finite", bool(torch.isfinite(net.store.flat).all()), kinds)
