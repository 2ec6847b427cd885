"""GPU: the data-parallel training step END TO END with the real kernels on two ranks.

The GPU box has one MI355X and RCCL refuses two ranks on one device, so the two ranks share cuda:0 and exchange the
gradient over gloo (which stages CUDA tensors through the host): everything except the transport is the product path --
VoteNetHotPath.train_step(world=2): forward, loss graph, backward, dp.GradSync (tail all-reduce issued after sa3's
backward from the communication stream, head after the last weight gradient), clip + Adam with the 1/world scale.
Checked: both ranks end with bit-identical parameters after every step, and the update equals the optimizer applied to
the mean of the two ranks' local gradients.  (N ranks over RCCL: bench.py --gpus N on the driver's 8-GPU node.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from votenet_amd import dp, synth
    from votenet_amd import loss as VL
    from votenet_amd import mlp as M
    from votenet_amd import model as VM
    net = VM.VoteNetHotPath(dev, seed=10 + rank, npoints=(512, 256, 128, 64))  # replicas start different ...
    dp.broadcast_params(net.store)                                               # ... and are made identical
    net.init_optimizer(1e-3)
    seeds = dp.scene_seeds(rank, 2, base=300)                                    # disjoint scene shards
    x = torch.from_numpy(synth.room_batch(2, 4096, seeds[0])).to(dev)
    gt = VL.gt_to_device(synth.room_gt(2, 4096, seeds[0]), dev)
    # capture the LOCAL gradient slices exactly as they enter the collectives
    local = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        local.append((t.storage_offset(), t.clone()))
        return real(t, *a, **k)
    ok_equal, ok_update, colls = [], [], []
    for step in (1, 2, 3):
        p0, m0, v0 = net.store.flat.clone(), net._m.clone(), net._v.clone()
        local.clear()
        dist.all_reduce = spy
        try:
            net.train_step(x, gt=gt, world=world)
        finally:
            dist.all_reduce = real
        torch.cuda.synchronize()
        colls.append(list(net._gsync.log))
        g_local = torch.zeros_like(net.store.grad)
        for off, t in local:
            g_local[off:off + t.numel()] = t
        both = [torch.zeros_like(g_local) for _ in range(world)]
        dist.all_gather(both, g_local)
        g_sum = both[0] + both[1]
        # the all-reduced bucket the optimizer saw is exactly the sum of the two local gradients
        same_sum = torch.equal(net.store.grad, g_sum)
        sumsq = torch.zeros_like(net._sumsq)
        M.clip_adam(net._seg, sumsq, p0, g_sum, m0, v0, 1e-3, step, grad_scale=1.0 / world)
        ok_update.append(bool(same_sum and torch.equal(p0, net.store.flat)))
        flats = [torch.zeros_like(p0) for _ in range(world)]
        dist.all_gather(flats, net.store.flat)
        ok_equal.append(bool(torch.equal(flats[0], flats[1])))
        differ = not torch.equal(both[0], both[1])
    q.put((rank, ok_equal, ok_update, colls, differ, net.store.grad.numel(), net.store.offset_of("sa3/"),
           bool(torch.isfinite(net.store.flat).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_train_step_two_ranks_real_kernels(hiplib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok_equal, ok_update, colls, differ, numel, split, finite in res:
        assert finite and differ                      # different scenes per rank -> different local gradients
        assert ok_equal == [True] * 3                 # bit-identical replicas after every step
        assert ok_update == [True] * 3                # = optimizer(mean of the local gradients), bit for bit
        assert colls == [[("tail", numel - split), ("head", split)]] * 3


def test_check_dp_one_rank_rccl_communicator(hiplib):
    """bench.py --gpus 1 --check-dp: a ONE-rank RCCL communicator (backend nccl) carries both collectives of dp.GradSync from the
    communication stream -- the three-stream event ordering in front of RCCL itself, on the GPU this box has -- and the result must
    be bit-equal to the blocking exchange.  The line reports what the communicator says, and the device's PCI address / uuid."""
    import json
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--check-dp", "--steps", "2"], env=e,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["check_dp"]["equal_everywhere"] and line["check_dp"]["equal_on_this_rank"]
    assert [c[0] for c in line["check_dp"]["collectives_per_step"]] == ["tail", "head"]
    assert line["communicator"]["backend"] == "nccl" and line["communicator"]["world_size"] == 1 and line["n_gpus"] == 1
    assert "tail_ms" in line["check_dp"]["timings_last_step"] and "head_ms" in line["check_dp"]["timings_last_step"]
    d0 = line["communicator"]["devices"][0]
    assert "pci_bus_id" in d0 or "uuid" in d0


def test_bench_gpus_2_line_validates_its_own_data_parallel_path():
    """`python bench.py --gpus 2` end to end with the real kernels -- the launcher, two ranks, the timed region with dp.GradSync inside
    train_step, the dp_collectives leg and the self-check appended to the line (round-3 verdict item 5).  The box has ONE GPU and RCCL
    refuses two ranks on a device: VOTENET_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 over gloo; everything else is the N-rank path
    the driver runs on the 8-GPU node."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VOTENET_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "2",
                        "--headline-only"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4 and line["value"] > 0
    c = line["check_dp"]
    assert c["equal_everywhere"] and c["equal_on_this_rank"] and c["ranks_identical"] and c["world_size"] == 2
    assert [w for w, _ in c["collectives_per_step"]] == ["tail", "head"] and c["tail_exposed_ms"] is not None
    assert line["communicator"]["world_size"] == 2 and line["communicator"]["backend"] == "gloo"
    assert c["distinct_devices"] == 1   # both ranks on cuda:0 HERE (the flag above); N on the driver's node
    # every rank's own clock and host placement in the one line (a straggler shows in a single SCALE record)
    pr = line["per_rank"]
    assert [r_["rank"] for r_ in pr["ranks"]] == [0, 1]
    for r_ in pr["ranks"]:
        assert r_["ms_per_step"] > 0 and r_["step_ms_min"] <= r_["step_ms_median"] <= r_["step_ms_max"]
        assert set(r_["hostpin"]) == {"cpus", "gpu_numa_node"}
    assert pr["ms_per_step_min"] <= pr["ms_per_step_max"] and abs(pr["ms_per_step_max"] - line["ms_per_step"]) < 1e-3 + 0.001 * line["ms_per_step"]


_RCCL_BESIDE_GRAPHS = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = %(port)r
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1)
from votenet_amd import dp, synth, loss as VL, model as VM
xs = [torch.from_numpy(synth.room_batch(4, 20480, s)).to(dev) for s in (11, 22, 33)]
gts = [VL.gt_to_device(synth.room_gt(4, 20480, s), dev) for s in (11, 22, 33)]
nets = [VM.VoteNetHotPath(dev, seed=5) for _ in range(2)]
for n in nets:
    # learning rate 0: the parameters stay where they are, so every step sees the same three batches in the same state and the number
    # of positive proposals of a batch (a handful on a freshly initialised net) cannot drift to 0 -- where the reference's loss, and
    # this one, is NaN by definition (reduce_mean of an empty set); everything the test is about still runs every step
    n.init_optimizer(0.0)
split = nets[0].store.offset_of("sa3/")
nets[0]._gsync = dp.GradSync(nets[0].store, split, force=True)   # a ONE-rank RCCL communicator carries both collectives of every step
logs, gdiff, finite, npos = [], [], True, []
for i in range(12):
    for n in nets:
        n.train_step(xs[i %% 3], gt=gts[i %% 3], next_x=[xs[(i + 1) %% 3]])
    torch.cuda.synchronize()
    logs.append(list(nets[0]._gsync.log))
    if i == 0:  # same state, same batch: the exchanged bucket is the local gradient (a sum over one rank), up to the atomics' order
        a, b = nets[0].store.grad, nets[1].store.grad
        gdiff.append(float((a - b).abs().max() / b.abs().max()))
    finite = finite and bool(torch.isfinite(nets[0].store.flat).all()) and bool(torch.isfinite(nets[0].last_losses).all())
    npos.append(float(nets[0].last_losses[10]))
    nets[1].store.flat.copy_(nets[0].store.flat); nets[1]._m.copy_(nets[0]._m); nets[1]._v.copy_(nets[0]._v)
sg = list(nets[0]._stretch_graphs.values())
print(json.dumps(dict(logs=logs, gdiff=gdiff, finite=finite, npos=npos, n_stretch=len(sg), replays=[g.replays for g in sg],
                      numel=nets[0].store.grad.numel(), split=split,
                      loss=[float(nets[0].last_losses[0]), float(nets[1].last_losses[0])])))
dist.destroy_process_group()
"""


def test_stretch_and_geometry_graphs_replay_beside_a_live_rccl_communicator(hiplib):
    """What `bench.py --gpus N` does on every rank, on the one GPU of this box: the DEFAULT train step -- geometry ring and stretch
    segments captured (thread_local capture mode) and replayed -- while a process group's RCCL communicator and its watchdog thread
    are alive and dp.GradSync issues the tail / head all-reduces of every step from the communication stream (force=True: a one-rank
    communicator still runs them).  The deterministic self-check of bench.py never replays a graph, and the two-rank tests run over
    gloo: this is the combination they leave out.  Checked: no capture is invalidated, both collectives in every step, the captured
    stretch replays, the exchanged bucket is the local gradient, losses finite and equal to a replica without a process group."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    code = _RCCL_BESIDE_GRAPHS % dict(root=ROOT, port=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert min(d["npos"]) > 0, d["npos"]                      # precondition: a batch without a positive proposal has a NaN loss by definition
    assert d["finite"]
    assert d["logs"] == [[["tail", d["numel"] - d["split"]], ["head", d["split"]]]] * 12
    assert d["n_stretch"] == 1 and d["replays"][0] >= 9      # step 1 measures the arena demand, step 2 captures, the rest replay
    assert d["gdiff"][0] < 1e-4
    assert abs(d["loss"][0] - d["loss"][1]) <= 1e-4 * abs(d["loss"][1])
