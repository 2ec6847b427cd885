"""probe: deterministic mode, two replicas from one seed stepping on the same batches -- parameters bit-equal after every step?
(what bench.py --gpus N's self-check asks, without the process group)   python tools/probe/det_step.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch
import votenet_amd
from votenet_amd import loss as VL, model as VM, synth, mlp as M
dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", "2")), 20480
votenet_amd.set_deterministic(True)
xs = [torch.from_numpy(synth.room_batch(B, n, s)).to(dev) for s in (1000, 500000)]
gts = [VL.gt_to_device(synth.room_gt(B, n, s), dev) for s in (1000, 500000)]
nets = [VM.VoteNetHotPath(dev, seed=0) for _ in range(2)]
for net in nets:
    net.init_optimizer()
bad = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for net in nets:
        net.train_step(xs[i % 2], None, 1, gt=gts[i % 2])
    torch.cuda.synchronize()
    same = torch.equal(nets[0].store.flat, nets[1].store.flat)
    gsame = torch.equal(nets[0].store.grad, nets[1].store.grad)
    if not same or not gsame:
        bad += 1
        d = (nets[0].store.grad != nets[1].store.grad)
        names = []
        for name, v in nets[0].store.views.items():
            g0, g1 = nets[0].store.g(name), nets[1].store.g(name)
            if not torch.equal(g0, g1):
                names.append("%s(%d)" % (name, int((g0 != g1).sum())))
        print("step %d: params equal %s, grads equal %s; %d grad values differ: %s" % (i, same, gsame, int(d.sum()), " ".join(names[:12])))
        nets[1].store.flat.copy_(nets[0].store.flat); nets[1].store.params_changed()
        nets[1]._m.copy_(nets[0]._m); nets[1]._v.copy_(nets[0]._v)
print("%d steps with a difference" % bad)
