"""Train step time with a process group alive (one-rank RCCL group on the 1-GPU box): no exchange, the overlapped gradient exchange
forced on (dp.GradSync force=True), with and without the geometry graphs -- the captures run while the group's watchdog thread is alive."""
import os, sys, time, socket, datetime
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R]
import torch, torch.distributed as dist
from votenet_amd import dp, loss as VL, model as VM, synth
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net = VM.VoteNetHotPath(dev, seed=0)
xs = [torch.from_numpy(synth.room_batch(8, 20480, 1000 + 8 * i)).to(dev) for i in range(3)]
gts = [VL.gt_to_device(synth.room_gt(8, 20480, 1000 + 8 * i), dev) for i in range(3)]
net.init_optimizer()
def run(label, k=40):
    for i in range(8):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3], world=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 8 + k):
        net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3], world=1)
    torch.cuda.synchronize()
    print("%-70s %.3f ms per step" % (label, (time.perf_counter() - t0) / k * 1e3), flush=True)
run("no process group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
run("one-rank RCCL group alive, no exchange in the step")
keep = net._gsync
for graphs in (True, False, True):
    VM.GEOMETRY_GRAPHS = graphs
    net._gsync = dp.GradSync(net.store, net.store.offset_of("sa3/"), overlap=True, force=True)
    run("exchange forced (tail overlapped + head), geometry graphs %s" % graphs)
    net._gsync = dp.GradSync(net.store, net.store.offset_of("sa3/"), overlap=False, force=True)
    run("exchange forced (one blocking all-reduce), geometry graphs %s" % graphs)
net._gsync = keep
ring = next(iter(net._geometry_rings.values()))
print("%d geometry graphs, generations %s; loss finite: %s" % (len(ring["graphs"]), [g.generation for g in ring["graphs"]], bool(torch.isfinite(net.last_losses).all())))
dist.destroy_process_group()
