"""CPU, world_size 2 over gloo: the data-parallel path -- one flat gradient bucket, ONE all-reduce,
identical replicas after the broadcast, disjoint scene shards.  (The HIP kernels need a GPU; what runs
here is exactly the host logic bench.py / train_step use for N > 1.)"""
import os
import socket
import sys

import pytest
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
    # numpy payloads: a torch tensor through mp.Queue ships a storage fd the parent must fetch from THIS process, which may
    # have exited by then (round-3 verdict: ConnectionResetError in rebuild_storage_fd)
    q.put((rank, flat0.numpy().copy(), calls, scale, ok_sum, seeds, store.grad.numel()))
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
    assert f0.shape == f1.shape and (f0 == f1).all()  # broadcast made the replicas identical
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


# ---------------------------------------------------------------------------------------------------------------------
# The REAL train_step host path on two ranks (CPU, gloo), kernels stubbed: forward / the modules' backward / the optimizer
# kernel are replaced by torch stand-ins that produce rank-dependent gradients; everything between them -- the tape walk of
# _backward, dp.GradSync (tail all-reduce started after sa3's backward, head after the last weight gradient), the 1/world
# scale handed to the optimizer -- is the product code.
def _torch_clip_adam(seg, sumsq, p, g, m, v, lr, step, grad_scale=1.0, clip=0.5, beta1=0.9, beta2=0.999, eps=1e-8):
    """model.py:240-250 as votenet_clip_adam computes it (per-tensor clip_by_average_norm, then TensorFlow's Adam)."""
    seg = seg.tolist()
    lr_t = lr * (1 - beta2 ** step) ** 0.5 / (1 - beta1 ** step)
    for a, b in zip(seg[0::2], seg[1::2]):
        gg = g[a:b] * grad_scale
        avg = gg.norm() / (b - a)
        gg = gg * clip / max(float(avg), clip)
        m[a:b] = beta1 * m[a:b] + (1 - beta1) * gg
        v[a:b] = beta2 * v[a:b] + (1 - beta2) * gg * gg
        p[a:b] -= lr_t * m[a:b] / (v[a:b].sqrt() + eps)


def _torch_row_segments(rows, segs):
    """votenet_row_segments (csrc/glue.hip) as torch ops: dst = a (+ b), zeros without a -- the stand-in for the glue kernel."""
    for dst, a, b in segs:
        if a is None:
            dst.zero_()
        else:
            dst.copy_(a if b is None else a + b)


def _stub_grad(name, shape, rank, step):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000 + 1000 * rank + step)
    return torch.randn(shape, generator=g) * (3.0 if "sa1" in name else 0.01)


def _stub_net(rank, seed, events, step_no, B=2):
    """A VoteNetHotPath on the CPU whose kernels are stand-ins (see above); the chain-backward stub finds its net through the
    voting record, so several stubbed nets can live in one process."""
    from votenet_amd import mlp as M
    from votenet_amd import model as VM
    from votenet_amd import pointnet2 as P
    cpu = torch.device("cpu")
    net = VM.VoteNetHotPath(cpu, seed=seed)
    net.overlap_wgrad = False
    net._side_stream = lambda: None
    net.store.refresh_transposes = lambda stream=None: None
    net.update_moving_averages = lambda tape: None
    NS = net.sa2.npoint

    def fill(layers):
        for L in layers:
            for k in ("W", "b") + (("gamma", "beta") if L.bn else ()):
                L.gp(k).add_(_stub_grad(L.name + "/" + k, L.gp(k).shape, rank, step_no[0]))

    def stub_forward(x, tape=None, next_x=None):
        for op in ("sa", "sa", "sa", "sa", "fp", "fp"):
            tape.append(dict(op=op))
        tape.append(dict(op="vote", recs=[net], b=B, n=NS))
        tape.append(dict(op="sa", fps_idx=None))
        return dict(proposals_output=torch.zeros(B, 256, 79))

    def sa_backward(mod, name, n_in, c_in):
        def f(rec, g_out, need_feat_grad=True, need_xyz_grad=False):
            events.append(name + ".backward")
            fill(mod.mlp + (mod.mlp2 or []))
            return (torch.zeros(B, n_in, c_in) if need_feat_grad else None), (torch.zeros(B, n_in, 3) if need_xyz_grad else None)
        return f

    def fp_backward(mod, name, n1, c1, m, c2):
        def f(rec, dy):
            events.append(name + ".backward")
            fill(mod.mlp)
            return torch.zeros(B, n1, c1), torch.zeros(B, m, c2)
        return f

    net.forward = stub_forward
    net.proposal.backward = sa_backward(net.proposal, "proposal", NS, 256)
    net.sa4.backward = sa_backward(net.sa4, "sa4", net.sa3.npoint, 256)
    net.sa3.backward = sa_backward(net.sa3, "sa3", net.sa2.npoint, 256)
    net.sa2.backward = sa_backward(net.sa2, "sa2", net.sa1.npoint, 128)
    net.sa1.backward = sa_backward(net.sa1, "sa1", 64, 3)
    net.fp2.backward = fp_backward(net.fp2, "fp2", net.sa2.npoint, 256, net.sa3.npoint, 256)
    net.fp1.backward = fp_backward(net.fp1, "fp1", net.sa3.npoint, 256, net.sa4.npoint, 256)
    net._stub_fill_voting = lambda: (events.append("voting.backward"), fill(net.voting))

    def chain_backward(recs, g, mode, **kw):
        recs[0]._stub_fill_voting()
        return torch.zeros_like(g)
    P.mlp_chain_backward = chain_backward
    M.clip_adam = _torch_clip_adam
    M.row_segments = _torch_row_segments
    return net


def _train_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from votenet_amd import dp
    events, step_no = [], [0]
    B = 2
    net = _stub_net(rank, rank, events, step_no, B)  # replicas start different ...
    dp.broadcast_params(net.store)                   # ... one broadcast makes them identical
    p_start = net.store.flat.clone()
    net._gsync = dp.GradSync(net.store, net.store.offset_of("sa3/"))
    real_tail = net._gsync.start_tail
    net._gsync.start_tail = lambda streams=None: (events.append("tail all-reduce issued"), real_tail(streams))[1]

    net.init_optimizer(lr=1e-3)
    # expected: the same optimizer on the MEAN of the two ranks' gradients
    p_exp = p_start.clone()
    m_exp, v_exp = torch.zeros_like(p_exp), torch.zeros_like(p_exp)
    logs = []
    cot = dict(proposals_output=torch.zeros(B, 256, 79), votes_xyz=None)
    for step in (1, 2):
        step_no[0] = step
        events.clear()
        net.train_step(torch.zeros(B, 64, 3), cot, world)
        logs.append((list(events), list(net._gsync.log)))
        g_sum = torch.zeros_like(p_exp)
        for r in range(world):
            for name, v in net.store.gviews.items():
                off = v.storage_offset()
                g_sum[off:off + v.numel()] += _stub_grad(name, v.shape, r, step).reshape(-1)
        _torch_clip_adam(net._seg, None, p_exp, g_sum, m_exp, v_exp, 1e-3, step, grad_scale=1.0 / world)
    q.put((rank, net.store.flat.numpy().copy(), p_exp.numpy().copy(), p_start.numpy().copy(), logs, net.store.grad.numel(),
           net.store.offset_of("sa3/")))  # numpy, not tensors: see _worker
    dist.barrier()
    dist.destroy_process_group()


def _check_worker(rank, world, port, q, sabotage):
    """dp.check_overlap_against_blocking (what bench.py --check-dp runs) on two stubbed replicas per rank; sabotage: the overlapped
    replica's tail collective is issued BEFORE sa3's gradients exist -- the check has to see it."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from votenet_amd import dp
    events, step_no = [], [0]
    B = 2
    nets = [_stub_net(rank, 0, events, step_no, B) for _ in range(2)]
    for net in nets:
        dp.broadcast_params(net.store)
        net.init_optimizer(lr=1e-3)
    cot = dict(proposals_output=torch.zeros(B, 256, 79), votes_xyz=None)

    orig_sa3 = nets[0].sa3.backward

    def run(net, i):
        step_no[0] = i + 1
        if sabotage and net is nets[0]:
            gs = net._gsync

            def early(rec, g_out, **kw):
                gs.start_tail(None)                  # too early: sa3's own gradients are not in the bucket yet
                gs._work[-1].wait()                  # (make the race lose every time: the collective completes before sa3 writes)
                gs.start_tail = lambda streams=None: None  # the real call site, after sa3: nothing left to start
                return orig_sa3(rec, g_out, **kw)
            net.sa3.backward = early
            try:
                net.train_step(torch.zeros(B, 64, 3), cot, world)
            finally:
                gs.__dict__.pop("start_tail", None)
                net.sa3.backward = orig_sa3
            return
        net.train_step(torch.zeros(B, 64, 3), cot, world)
    res = dp.check_overlap_against_blocking(nets[0], nets[1], run, steps=2)
    info = dp.comm_info(torch.device("cpu"))
    q.put((rank, res, info))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_train_step_world2(world):
    """world 2, and world 8 = the node BASELINE config 4 names (eight ranks, 8 scenes each): the same host path per rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [tuple(torch.from_numpy(v) if hasattr(v, "dtype") else v for v in r) for r in res]
    (_, f0, e0, s0, logs0, numel, split) = res[0]
    logs_all = []
    for (_, f1, e1, s1, logs1, _, _) in res:
        assert torch.equal(s0, s1)                   # identical replicas at the start (broadcast)
        assert torch.equal(f0, f1)                   # ... and after two optimizer steps: bit-identical parameters on every rank
        # the mean-gradient update.  world 2: a + b has one order; world 8: the ring adds the eight terms in another order than the
        # expectation's rank loop, and Adam's g / sqrt(v) turns a last-bit difference of a nearly cancelled sum into ~1e-3 of one update (lr 1e-3)
        assert torch.allclose(f1, e1, rtol=1e-6, atol=1e-9 if world == 2 else 2e-6), float((f1 - e1).abs().max())
        logs_all += logs1
    assert not torch.equal(f0, s0)
    for events, colls in logs_all:
        # two collectives per step, contiguous slices of the one flat bucket: tail = sa3 ... proposal, head = sa1 + sa2
        assert colls == [("tail", numel - split), ("head", split)]
        order = [e for e in events if e.endswith(".backward") or e.startswith("tail")]
        assert order == ["proposal.backward", "voting.backward", "fp2.backward", "fp1.backward", "sa4.backward", "sa3.backward",
                         "tail all-reduce issued", "sa2.backward", "sa1.backward"]
    assert 0 < split < 0.1 * numel                   # the part that cannot be overlapped is small (sa1 + sa2: 8 % of the bucket)


def _run2(target, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_check_dp_overlapped_exchange_equals_blocking_world2():
    """bench.py --check-dp's comparison on the real train_step host path (stubbed kernels): overlapped == blocking on every rank,
    every rank == rank 0, and the communicator's own report (world size, backend, one entry per rank) is what goes into the line."""
    for rank, res, info in _run2(_check_worker, False):
        assert res["equal_on_this_rank"] and res["ranks_identical"] and res["equal_everywhere"]
        assert [c[0] for c in res["collectives_per_step"]] == ["tail", "head"]
        assert info["world_size"] == 2 and info["backend"] == "gloo" and info["distinct_devices"] == 2
        assert sorted(d["rank"] for d in info["devices"]) == [0, 1]


def test_check_dp_detects_a_collective_issued_too_early():
    """The check is not vacuous: a tail all-reduce issued before sa3's gradients are written gives different parameters, and
    every rank learns it (the verdict is an all-reduce MIN)."""
    for rank, res, info in _run2(_check_worker, True):
        assert not res["equal_on_this_rank"] and not res["equal_everywhere"]
