"""CPU, world_size 2 over gloo: the data-parallel path -- one flat gradient bucket, ONE all-reduce,
identical replicas after the broadcast, disjoint scene shards.  (The HIP kernels need a GPU; what runs
here is exactly the host logic bench.py / train_step use for N > 1.)"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from votenet_amd import dp
    from votenet_amd import pointnet2 as P
    store = P.ParamStore(torch.device("cpu"))
    P.SAModule(store, "sa", 16, 0.2, 8, 3, [8, 8, 16])
    P.make_mlp(store, "vote", 19, [16, 19], "fc", last_plain=True)
    store.materialize(seed=rank)  # replicas start DIFFERENT ...
    dp.broadcast_params(store)    # ... and are made identical by one broadcast
    flat0 = store.flat.clone()
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.numel())
        return real(t, *a, **k)
    dist.all_reduce = counting
    for name in store.views:
        store.g(name).fill_(float(rank + 1))
    scale = dp.sync_gradients(store)
    dist.all_reduce = real
    ok_sum = all(bool((store.g(n) == 3.0).all()) for n in store.views)  # 1 + 2
    seeds = dp.scene_seeds(rank, 8)
    q.put((rank, flat0, calls, scale, ok_sum, seeds, store.grad.numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, f0, c0, s0, ok0, seeds0, n0), (r1, f1, c1, s1, ok1, seeds1, n1) = res
    assert torch.equal(f0, f1)                      # broadcast made the replicas identical
    assert c0 == [n0] and c1 == [n1]                # exactly ONE all-reduce, over the whole flat bucket
    assert s0 == 0.5 and s1 == 0.5                  # mean = sum * 1/world, folded into the optimizer
    assert ok0 and ok1
    assert not set(seeds0) & set(seeds1) and len(seeds0) == 8  # disjoint scene shards, fixed per-GPU batch


def test_single_process_is_noop():
    sys.path.insert(0, ROOT)
    from votenet_amd import dp
    from votenet_amd import pointnet2 as P
    store = P.ParamStore(torch.device("cpu"))
    P.make_mlp(store, "m", 4, [4], "fc")
    store.materialize(0)
    assert dp.world_size() == 1 and dp.sync_gradients(store) == 1.0
