#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark (contract: see the task statement / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W

N>1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when WORLD_SIZE is
not set -- bench.py starts the N ranks itself as child processes before anything touches the GPU (launch_ranks).

One "step" = one pass of the hot path over one batch of 8 synthetic 20480-point scenes per GPU
(BASELINE.json configs[2]/[3]: VoteNet layer stack sa1..sa4, fp1, fp2, voting, proposal).
Workloads:
    train : forward + the reference's loss graph + backward + (N>1: one RCCL all-reduce of the flat gradient bucket) + Adam
    fwd   : forward only (BASELINE.json configs[1] plus voting + proposal)
Inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
import gc
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
T_PROCESS_START = time.perf_counter()
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate, dense
MFMA_BF3_EQUIV_TF = round(16 * 157.3 / 6, 1)  # dense bf16 MFMA peak (16 x the fp32 rate, 2.5 PFLOP/s) / six bf16 MFMAs per fp32 product = 419.5
MFMA_H2_EQUIV_TF = round(16 * 157.3 / 3, 1)   # dense fp16 MFMA peak (the same rate) / three fp16 MFMAs per fp32 product (round 6: forward operands) = 838.9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU")
    ap.add_argument("--points", type=int, default=20480, help="points per scene (config.py:1)")
    ap.add_argument("--workload", default=None, choices=["train", "fwd"])
    ap.add_argument("--scene", default="room", choices=["room", "uniform"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="launcher self-test on CPU (gloo rendezvous, no kernels, nothing measured)")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra unpipelined / alone-on-the-GPU measurements (profiling)")
    ap.add_argument("--no-gram", action="store_true", help="pooled layers: direct backward GEMMs on the stored z instead of the Gram form")
    ap.add_argument("--check-dp", action="store_true",
                    help="data-parallel self-check instead of a measurement: in deterministic mode, train steps with the overlapped "
                         "gradient exchange (dp.GradSync: tail under sa2/sa1's backward, head at the end) against the same steps with "
                         "blocking all-reduces of the same two slices after a device synchronise; parameters must be bit-equal on every rank, else exit code 4.  "
                         "Works at --gpus 1 too (a one-rank RCCL communicator still runs both collectives)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not compute the coordinate-only geometry of the next batch underneath the current step")
    return ap.parse_args()


def cpu_baseline(points, scene_kind, min_seconds=8.0, max_scenes=3, ops=True):
    """The CPU oracle timed on this host (tools/bench_legs.cpu_baseline): forward of whole scenes single-thread and on all cores
    (OpenMP), CPU model / core counts, per-op medians.  Bounded sample (~25 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_legs
    return bench_legs.cpu_baseline(points, scene_kind, min_seconds=min_seconds, max_scenes=max_scenes, ops=ops)


def visible_gpu_count(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process could use, counted WITHOUT any HIP / HSA call: the KFD topology in sysfs (a node with simd_count > 0 is a
    GPU; CPU nodes have 0), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one is set.
    torch.cuda.device_count() is not used: on a ROCm build without amdsmi it is hipGetDeviceCount, which initialises the runtime
    in the launcher -- and a process that has initialised the GPU must not start the ranks (fork + exec) on this pool.
    -> the count, or None when sysfs has no KFD topology (then the ranks find out themselves: rank r fails fast if r >= its own
    device_count())."""
    try:
        nodes = sorted(os.listdir(root), key=lambda v: int(v) if v.isdigit() else 0)
    except OSError:
        return None
    gpus = 0
    for nd in nodes:
        try:
            props = dict(ln.split(None, 1) for ln in open(os.path.join(root, nd, "properties")).read().splitlines() if " " in ln)
        except OSError:
            continue
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            gpus += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = [t for t in v.split(",") if t.strip() != ""]
            gpus = min(gpus, len(listed))
    return gpus


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as CHILD processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would) and forward rank 0's JSON line.  The parent
    never touches the GPU -- it imports neither torch nor the HIP library; GPUs are counted in sysfs (visible_gpu_count) -- and never
    re-execs itself.  A failed child ends the others and makes the whole run fail; a rendezvous port that was taken between
    choosing it and rank 0 binding it (EADDRINUSE) is retried on a new port."""
    import socket
    import subprocess
    import tempfile
    if not args.dry_run and not os.environ.get("VOTENET_BENCH_SHARE_GPU"):
        have = visible_gpu_count()
        if have is not None and have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible (KFD topology)\n" % (args.gpus, have))
            return 2
    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = []
        out0 = tempfile.TemporaryFile(mode="w+")
        err_lines = []  # rank 0's stderr: forwarded line by line as it arrives (an outer timeout that kills this launcher still leaves
        #                 the diagnostics on the terminal) and kept for the EADDRINUSE test below
        for r in range(args.gpus):
            # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's (and torch's)
            # cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it already; it is
            # repeated here so that a launcher started from a scrubbed environment still hands it to every rank.
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE if r == 0 else None,
                                          text=True if r == 0 else None))

        def tee(pipe):
            for ln in pipe:
                err_lines.append(ln)
                if "EADDRINUSE" not in ln and "address already in use" not in ln.lower():  # (a retry follows: said below)
                    sys.stderr.write(ln)
                    sys.stderr.flush()
        import threading
        th = threading.Thread(target=tee, args=(procs[0].stderr,), daemon=True)
        th.start()
        codes = [None] * len(procs)
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes):  # one rank died: the others would wait in the collective forever
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.kill()  # exactly the PIDs started above
                        codes[r] = p.wait()
                break
            time.sleep(0.05)
        th.join(10)
        err_text = "".join(err_lines)
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if bad and attempt < 2 and ("EADDRINUSE" in err_text or "address already in use" in err_text.lower()):
            sys.stderr.write("bench.py: rendezvous port %d was taken, retrying on another\n" % port)
            continue
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
        if bad:
            sys.stderr.write("bench.py: rank(s) failed (rank, exit code): %s\n" % bad)
            return 1
        return 0
    return 1


def dp_self_check(dev, nets, run_step, steps=2):
    """What `--gpus N` (N > 1) appends to its line after the timed region: dp.check_overlap_against_blocking on two replicas (nets[0]
    exchanges gradients the overlapped way -- tail all-reduce under the backward pass of sa2 / sa1, head after the last weight gradient --
    nets[1] with blocking all-reduces of the same two slices between device synchronisations), what the communicator says about its ranks, and how long the
    tail collective was still running when the main stream reached the end of the backward pass.  equal_everywhere is an all-reduce MIN:
    every rank learns the same verdict (and exits with code 4 when it is False)."""
    from votenet_amd import dp
    res = dp.check_overlap_against_blocking(nets[0], nets[1], run_step, steps=steps)
    info = dp.comm_info(dev)
    tm = res.get("timings_last_step") or {}
    return {"equal_everywhere": res["equal_everywhere"], "equal_on_this_rank": res["equal_on_this_rank"], "ranks_identical": res["ranks_identical"],
            "steps": res["steps"], "collectives_per_step": res["collectives_per_step"], "world_size": info["world_size"],
            "backend": info["backend"], "distinct_devices": info["distinct_devices"], "tail_exposed_ms": tm.get("tail_exposed_ms"),
            "tail_ms": tm.get("tail_ms"), "head_ms": tm.get("head_ms"),
            "what": "deterministic mode, %d train steps on two replicas per rank after the timed region: overlapped exchange (dp.GradSync) vs "
                    "blocking all-reduces of the same two slices; parameters torch.equal on every rank and equal to rank 0's" % res["steps"]}


class _DryReplica:
    """The dry run's stand-in for a VoteNetHotPath replica (CPU, gloo): a flat parameter / gradient bucket with the real layout names
    (head = sa1 | sa2, tail = sa3 ...), gradients that differ per rank and step, dp.GradSync exactly as train_step drives it
    (begin -> start_tail after the tail's gradients exist -> finish), a plain SGD update on the mean gradient.  No kernels."""

    def __init__(self, seed):
        import torch
        from votenet_amd import pointnet2 as P
        self.store = P.ParamStore(torch.device("cpu"))
        for name, cin, cout in (("sa1", 6, 16), ("sa2", 19, 16), ("sa3", 19, 32), ("proposal", 35, 8)):
            P.make_mlp(self.store, name, cin, [cout], "conv")
        self.store.materialize(seed)
        self._gsync = None

    def train_step(self, rank, world, i):
        import torch
        st, gs = self.store, self._gsync
        split = st.offset_of("sa3/")
        gs.begin()
        g = torch.Generator().manual_seed(1000 * rank + i)
        st.grad[split:] = torch.randn(st.grad.numel() - split, generator=g)   # the backward pass of proposal ... sa3
        gs.start_tail(None)
        if os.environ.get("VOTENET_BENCH_DRYRUN_DIVERGE") and gs.overlap and gs._work:
            gs._work[-1].wait()          # tests: a gradient written AFTER its collective ran (the bug class the check exists for)
            st.grad[split:] += 1.0
        st.grad[:split] = torch.randn(split, generator=g)                     # ... of sa2, sa1, beside the tail's all-reduce
        scale = gs.finish()
        st.flat.sub_(0.01 * scale * st.grad)


def per_rank_record(rank, local, world, ms_per_step, spread, pinned, hostpin):
    """N > 1: every rank's own clock and host placement gathered over the group into ONE record of the line (every rank calls this)."""
    import torch.distributed as dist
    numa = None
    try:
        numas = hostpin.gpu_numa_nodes()
        phys = hostpin.physical_gpu(local, len(numas))
        numa = numas[phys] if phys is not None and 0 <= phys < len(numas) else None
    except Exception:
        pass
    mine = {"rank": rank, "local_rank": local, "ms_per_step": round(ms_per_step, 3),
            "step_ms_min": spread["min"] if spread else None, "step_ms_median": spread["median"] if spread else None,
            "step_ms_max": spread["max"] if spread else None, "hostpin": {"cpus": pinned, "gpu_numa_node": numa}}
    got = [None] * world
    dist.all_gather_object(got, mine)
    ms = [g["ms_per_step"] for g in got]
    return {"ranks": got, "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
            "what": "each rank's own wall clock over the timed region (`value` uses the maximum), the spread of its step boundaries "
                    "(HIP events on its main stream) and where its host threads were pinned (votenet_amd/hostpin.py)"}


def dry_run(args):
    """Launcher self-test without a GPU (`--dry-run`, used by tests/): the ranks rendezvous over gloo exactly as the real run
    does over RCCL, all-reduce one number and rank 0 prints a line with n_gpus; no kernel runs, nothing is measured."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("VOTENET_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    check = None
    if world > 1:  # the self-check a real `--gpus N` run appends to its line, over gloo on stand-in replicas
        from votenet_amd import dp
        nets = [_DryReplica(seed=rank) for _ in range(2)]   # replicas start different: the broadcast makes them one model
        for net in nets:
            dp.broadcast_params(net.store)
        check = dp_self_check(torch.device("cpu"), nets, lambda net, i: net.train_step(rank, world, i), steps=2)
    per_rank = None
    if world > 1:  # the per-rank record of a real `--gpus N` line (clocks made up here: nothing is measured)
        import importlib.util
        spec = importlib.util.spec_from_file_location("votenet_hostpin", os.path.join(ROOT, "votenet_amd", "hostpin.py"))
        hostpin = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(hostpin)
        per_rank = per_rank_record(rank, int(os.environ.get("LOCAL_RANK", "0")), world, 1.0 + rank,
                                   {"min": 0.9 + rank, "median": 1.0 + rank, "max": 1.2 + rank}, None, hostpin)
    if rank == 0:
        emit(json.dumps({"metric": "dry run (launcher self-test, nothing measured)", "value": None, "n_gpus": world,
                         "rank_sum": float(t.item()), "check_dp": check, "per_rank": per_rank}))
    if world > 1:
        dist.destroy_process_group()
    if check is not None and not check["equal_everywhere"]:
        sys.exit(4)


def check_dp(args, dev, world, rank):
    """--check-dp (see parse()): two replicas from the same seed, deterministic mode, the same scenes; one exchanges gradients the
    overlapped way, the other with blocking all-reduces of the same slices.  Prints one JSON line on rank 0; exit code 4 when they differ."""
    import torch
    import torch.distributed as dist
    import votenet_amd
    from votenet_amd import dp, synth
    from votenet_amd import loss as vloss
    from votenet_amd import model as VM
    B, n = args.batch, args.points
    info = dp.comm_info(dev)
    votenet_amd.set_deterministic(True)
    seeds = [dp.scene_seeds(rank, B, base=bs)[0] for bs in (1000, 500000)]
    xs = [torch.from_numpy(synth.room_batch(B, n, sd)).to(dev) for sd in seeds]
    gts = [vloss.gt_to_device(synth.room_gt(B, n, sd), dev) for sd in seeds]
    nets = [VM.VoteNetHotPath(dev, seed=0) for _ in range(2)]
    for net in nets:
        dp.broadcast_params(net.store)
        net.init_optimizer()

    def run(net, i):
        net.train_step(xs[i % 2], None, world, gt=gts[i % 2])
    res = dp.check_overlap_against_blocking(nets[0], nets[1], run, steps=max(2, min(args.steps, 4)))
    if rank == 0:
        emit(json.dumps({"metric": "data-parallel self-check (--check-dp; nothing measured)", "value": None,
                         "n_gpus": info["world_size"], "communicator": info, "check_dp": res,
                         "what": "deterministic mode; parameters after %d train steps with dp.GradSync's overlapped exchange vs "
                                 "blocking all-reduces of the same two slices between device synchronisations: torch.equal on every rank, every rank equal to "
                                 "rank 0" % res["steps"]}))
    return 0 if res["equal_everywhere"] else 4


_REAL_STDOUT = None


def quiet_stdout():
    """From here on everything any library writes to file descriptor 1 (RCCL prints its version banner there when a communicator
    is created) goes to stderr; emit() writes the ONE JSON line of the contract to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


def load_hostpin():
    """votenet_amd/hostpin.py as a stand-alone module (no package __init__, hence no torch, no library): the pin has to precede the
    first thread torch or the HIP runtime creates."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("votenet_hostpin", os.path.join(ROOT, "votenet_amd", "hostpin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    quiet_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the rank's host threads on eight cores of its GPU's NUMA node (votenet_amd/hostpin.py: 3.1 -> 2.75 ms of enqueue per step);
    # before anything touches the GPU, so that the runtime's helper threads inherit the mask
    hostpin = load_hostpin()  # by path: `from votenet_amd import hostpin` would import torch (the package's __init__) BEFORE the pin
    assert "torch" not in sys.modules, "bench.py: torch was imported before the host threads were pinned"
    full_mask = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = hostpin.pin(local)
    import torch
    import torch.distributed as dist
    # VOTENET_BENCH_SHARE_GPU=1 (tests on the one-GPU box): every rank on cuda:0, gloo transport (RCCL refuses two ranks per device) --
    # the N-rank code path of this file with the real kernels; the line says so (communicator.backend "gloo", distinct_devices 1)
    share_gpu = bool(os.environ.get("VOTENET_BENCH_SHARE_GPU"))
    if share_gpu:
        local = 0
    if torch.cuda.device_count() <= local:  # the launcher counted in sysfs (or could not count at all): fail fast here
        sys.exit("bench.py: rank %d has no GPU %d (device_count %d)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.check_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # --gpus 1 --check-dp without a launcher: a one-rank communicator
            import socket
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.check_dp:
        code = check_dp(args, dev, world, rank)
        dist.destroy_process_group()
        sys.exit(code)

    from votenet_amd import model as VM
    from votenet_amd import mlp as vmlp
    from votenet_amd import synth, tf_grouping, tf_sampling
    have_train = hasattr(VM.VoteNetHotPath, "train_step")
    workload = args.workload or ("train" if have_train else "fwd")

    B, n = args.batch, args.points
    gen = synth.room_batch if args.scene == "room" else synth.uniform_batch
    from votenet_amd import dp
    # THREE different batches in rotation: step i works on batch i % 3 while the coordinate-only geometry (FPS, ball query,
    # three_nn) of the batches after it is computed on side streams -- every step computes one full geometry, none is
    # reused.  Lookahead: 1 for the train step; 2 for the forward pass, which is shorter than one geometry chain
    seeds = [dp.scene_seeds(rank, B, base=bs)[0] for bs in (1000, 500000, 900000)]
    xs = [torch.from_numpy(gen(B, n, sd)).to(dev) for sd in seeds]  # disjoint seeds per rank, resident in HBM
    net = VM.VoteNetHotPath(dev, seed=0)
    cot = None
    gts = [None] * len(seeds)
    if workload == "train":
        if args.scene == "room":  # the generating boxes are the ground truth: the reference's loss graph drives the backward
            from votenet_amd import loss as vloss
            gts = [vloss.gt_to_device(synth.room_gt(B, n, sd), dev) for sd in seeds]
        else:                     # uniform cubes have no objects: fixed cotangents instead
            cot = net.make_cotangents(B, seed=rank)
        dp.broadcast_params(net.store)
    pipeline = not args.no_pipeline
    counter = [0]
    pipe_on = [pipeline]

    def step():
        i = counter[0]
        counter[0] += 1
        nb = len(xs)
        x = xs[i % nb]
        nxt = ([xs[(i + 1) % nb]] + ([xs[(i + 2) % nb]] if workload == "fwd" else [])) if pipe_on[0] else None
        if workload == "train":
            net.train_step(x, cot, world, gt=gts[i % nb], next_x=nxt)
        else:
            net.forward(x, next_x=nxt)

    if args.no_gram:
        from votenet_amd import pointnet2
        pointnet2.POOL_GRAM_BACKWARD = False
    tf_sampling.PROFILE_EVENTS = None
    # setup, before the W warm-up steps the caller asked for: steps that create the streams, fill the caching allocator's
    # pools for all three rotating batches and start the geometry pipeline -- one launch-by-launch chain, then the capture of the ring
    # of geometry graphs (model.GEOMETRY_RING, a device synchronise each; the forward workload prefetches two batches per call) --
    # one-time work of the process, like loading the library; reported as "setup_steps"
    SETUP_STEPS = 8
    torch.cuda.synchronize()
    t_phase0 = time.perf_counter()  # from here to the end of the GPU legs the process keeps the GPU's queues fed (see gpu_busy_seconds)
    for k in range(SETUP_STEPS):
        # one set-up step is instrumented the way the timed region's will be (its prefetch enqueues the geometry chain launch by launch,
        # into buffers of the caching allocator instead of a graph's pool): the allocator then OWNS those blocks -- without this the
        # instrumented step of the timed region and its follower call hipMalloc (6 + 3 calls, milliseconds each: a 20-step run measured
        # 4.16 ms per step around a median of 3.79)
        if k == SETUP_STEPS - 3 and workload == "train" and pipeline:
            tf_sampling.PROFILE_EVENTS, tf_grouping.PROFILE_EVENTS = [], []
        if k == SETUP_STEPS - 2:
            # The interpreter's garbage pass and the step-boundary events two set-up steps BEFORE the warm-up: between the last untimed
            # step and the first timed one there is then nothing but the barrier, whatever W is.  (With a gc.collect() -- tens of
            # milliseconds of an idle GPU -- right in front of the timed region its first four steps took 4.1 / 3.8 / 3.65 / 3.67 ms instead
            # of 3.55: tools/probe/step_series.py, profiles/r05_bench_start.txt -- the device has to wake up again.)
            gc.collect()
            gc.disable()  # a cyclic-garbage pass of the interpreter in the middle of 20 steps shows up as a 30 ms step (measured)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # one per step boundary: the per-step spread
        step()
        tf_sampling.PROFILE_EVENTS = tf_grouping.PROFILE_EVENTS = None
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events (recorded on the launch stream) around every FPS and ball-query launch of one or two timed steps only (`roofline`:
    # the sa1 FPS; those steps enqueue the geometry chain launch by launch instead of replaying its graph, ~0.3 ms each).  The events
    # around every MFMA GEMM launch (`roofline_mlp`: ~150 pairs per step, 1.4 ms of host time -- the step turns host-bound and takes
    # 6-7 ms) are taken in two EXTRA steps right after the timed region: they would cost the headline 3-6 %
    t_setup = time.perf_counter() - t_phase0
    prof_steps = min(2 if args.steps >= 40 else 1, args.steps)  # short runs: one instrumented step
    gemm_steps = 2
    # ... taken in the MIDDLE of the timed region, where the host runs ~1.3 ms ahead of the GPU and absorbs most of the 0.6 ms the
    # launch-by-launch chain costs it (right after the barrier the queue is empty and the same steps took 6.1 and 4.9 ms)
    prof_first = (args.steps - prof_steps) // 2
    events, bq_events, gemm_events = [], [], []
    t0 = time.perf_counter()
    host_marks = [t0]
    marks[0].record()
    for i in range(args.steps):
        if i == prof_first:
            tf_sampling.PROFILE_EVENTS, tf_grouping.PROFILE_EVENTS = [], []
        if i == prof_first + prof_steps:
            events, tf_sampling.PROFILE_EVENTS = tf_sampling.PROFILE_EVENTS, None
            bq_events, tf_grouping.PROFILE_EVENTS = tf_grouping.PROFILE_EVENTS, None
        if os.environ.get("VOTENET_BENCH_PROFILE_STEP") == str(i):
            import cProfile, pstats, io
            pr = cProfile.Profile(); pr.enable(); step(); pr.disable()
            sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(14); sys.stderr.write(sio.getvalue()[:4000])
        else:
            step()
        marks[i + 1].record()
        host_marks.append(time.perf_counter())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if tf_sampling.PROFILE_EVENTS is not None:  # steps <= 2
        events, tf_sampling.PROFILE_EVENTS = tf_sampling.PROFILE_EVENTS, None
        bq_events, tf_grouping.PROFILE_EVENTS = tf_grouping.PROFILE_EVENTS, None
    if not args.headline_only:  # the GEMM launches of two more steps, outside the timed region (see above); every rank steps (collectives)
        vmlp.PROFILE_EVENTS = []
        vmlp.PROFILE_SHAPES = True  # entries carry (rows, cin, cout, note): the per-family table of roofline_mlp
        for _ in range(gemm_steps):
            step()
        torch.cuda.synchronize()
        gemm_events, vmlp.PROFILE_EVENTS = vmlp.PROFILE_EVENTS, None
    dt_own = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if os.environ.get("VOTENET_BENCH_STEP_TIMES"):
        sys.stderr.write("step times (ms): %s\n" % " ".join("%.2f" % marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)))
        sys.stderr.write("host times (ms): %s\n" % " ".join("%.2f" % ((host_marks[i + 1] - host_marks[i]) * 1e3) for i in range(args.steps)))
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)
                      if not prof_first <= i < prof_first + prof_steps + 1)  # un-instrumented steps (and not the one that follows them), main stream
    spread = ({"min": round(per_step[0], 3), "median": round(per_step[len(per_step) // 2], 3), "max": round(per_step[-1], 3),
               "steps": len(per_step), "what": "ms between consecutive step boundaries on the main stream (HIP events), steps without "
                                               "kernel-level timing events"} if per_step else None)

    # N > 1: every rank's own clock and host placement in ONE record of the line (a straggling rank -- a NUMA-remote host thread, a
    # throttled GPU -- is then visible in a single SCALE record: `value` only shows the maximum)
    per_rank = per_rank_record(rank, local, world, dt_own / args.steps * 1e3, spread, pinned, hostpin) if world > 1 else None

    # the same step with every batch's geometry computed inside its own step (nothing carried across steps): reported beside
    # the headline value, not instead of it
    in_step = None
    if pipeline and not args.headline_only:
        pipe_on[0] = False
        k2 = max(2, min(args.steps, 20))
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        in_step = {"value": round(B * world * k2 / dt2, 2), "ms_per_step": round(dt2 / k2 * 1e3, 3), "steps": k2,
                   "what": "same workload, no geometry prefetch: FPS / ball query / three_nn of a batch inside its own step"}
        pipe_on[0] = True

    # the same step in bit-reproducible mode (votenet_amd.set_deterministic: no fp32 atomics anywhere in the backward pass)
    det_step = None
    if workload == "train" and world == 1 and not args.headline_only:
        import votenet_amd
        prev = votenet_amd.set_deterministic(True)
        try:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
        finally:
            votenet_amd.set_deterministic(prev)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        det_step = {"value": round(B * 10 / dt3, 2), "ms_per_step": round(dt3 / 10 * 1e3, 3), "steps": 10,
                    "what": "same workload with votenet_amd.set_deterministic(True): weight gradients through ordered partial sums, "
                            "scatter-adds as gather-sums over the groupings' inverse index; two identical passes give bit-identical "
                            "gradients (tests/test_gpu_backward.py)"}

    # the same step with every GEMM on the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32) instead of bf16 x 3 split operands
    fp32_step = None
    if world == 1 and not args.headline_only:
        from votenet_amd import _lib as vlib
        vmlp.debug_switch("fast_bf3", 0)
        vmlp.debug_switch("gram_bf3", 0)
        vmlp.debug_switch("wgrad_bf3", 0)
        try:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            dt4 = time.perf_counter() - t1
        finally:
            vmlp.debug_switch("fast_bf3", 1)
            vmlp.debug_switch("gram_bf3", 1)
            vmlp.debug_switch("wgrad_bf3", 1)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        fp32_step = {"value": round(B * 10 / dt4, 2), "ms_per_step": round(dt4 / 10 * 1e3, 3), "steps": 10,
                     "what": "same workload, votenet_debug_fast_bf3(0) + votenet_debug_gram_bf3(0) + votenet_debug_wgrad_bf3(0): every GEMM product on "
                             "v_mfma_f32_32x32x2_f32 instead of six v_mfma_f32_32x32x16_bf16 on exactly split operands"}

    # the same step with the FORWARD operands on bf16 x 3 again (the round-3..5 form: six MFMAs per product instead of three fp16 ones)
    bf3_step = None
    if world == 1 and not args.headline_only and workload == "train" and vmlp.FORWARD_H2:
        vmlp.FORWARD_H2 = False
        net.store.rebuild_split()
        try:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            dt6 = time.perf_counter() - t1
        finally:
            vmlp.FORWARD_H2 = True
            net.store.rebuild_split()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        bf3_step = {"value": round(B * 10 / dt6, 2), "ms_per_step": round(dt6 / 10 * 1e3, 3), "steps": 10,
                    "what": "same workload with mlp.FORWARD_H2 = False: the forward matrices' images as bf16 x 3 (six v_mfma_f32_32x32x16_bf16 per "
                            "product) and the Gram-form dense input gradient on the fp32 MFMA kernel, as rounds 3-5 ran them"}

    # the same step on the FULL row layout (64 rows per ball, copies of slot 0 included: the reference's tensor shape), and what the
    # piece layout keeps of each level's grouped rows on these scenes
    full_step = row_layout = None
    if world == 1 and not args.headline_only and workload == "train":
        from votenet_amd import pointnet2 as vp2
        if vp2.HALF_GROUPS:
            tape = []
            net.forward(xs[0], tape)
            kept = {}
            for name, rec in list(zip(("sa1", "sa2", "sa3", "sa4"), tape[:4])) + [("proposal", tape[-1])]:
                half = rec["recs"][0].get("half")
                if half is not None:  # (the proposal module's count lives on the device: read here, outside the timed region)
                    kept[name] = round(half.true_count() * 16 / float(rec["idx"].numel()), 3)
            row_layout = {"piece_rows": 16, "grouped_rows_kept": kept,
                          "what": "a ball (tf_grouping_g.cu:26-29 pads it to 64 slots with copies of its first hit) keeps the 16-row pieces that hold "
                                  "a real neighbour, ceil(pts_cnt / 16) of 4; its slot 0 stands for the dropped copies with weight 1 + 16 * dropped "
                                  "pieces in the BatchNorm sums (csrc/half.hip): the same results up to summation order; every GEMM of sa1-sa4 runs "
                                  "on these rows, so executed flops and gemm time both shrink"}
            vp2.HALF_GROUPS = False
            try:
                for _ in range(4):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                dt5 = time.perf_counter() - t1
            finally:
                vp2.HALF_GROUPS = True
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            full_step = {"value": round(B * 10 / dt5, 2), "ms_per_step": round(dt5 / 10 * 1e3, 3), "steps": 10,
                         "what": "same workload with pointnet2.HALF_GROUPS = False (the piece layout off): all 64 rows of every ball through the grouped MLP"}

    # the same two kernels alone on the GPU (in the timed region they share it with the GEMMs of the previous batch)
    iso_fps = iso_bq = None
    if rank == 0 and not args.headline_only:
        tf_sampling.PROFILE_EVENTS, tf_grouping.PROFILE_EVENTS = [], []
        for _ in range(5):
            net.sa1.geometry(xs[0])
            torch.cuda.synchronize()
        iso_fps = sum(e0.elapsed_time(e1) for (e0, e1, *_r) in tf_sampling.PROFILE_EVENTS) / 5
        iso_bq = sum(e0.elapsed_time(e1) for (e0, e1, *_r) in tf_grouping.PROFILE_EVENTS) / 5
        tf_sampling.PROFILE_EVENTS = tf_grouping.PROFILE_EVENTS = None

    # what the COMMUNICATOR says about this run (world size, backend, every rank's PCI address / uuid gathered over the group) and
    # the latency of the two gradient collectives in three extra, untimed steps with HIP events on the communication stream.  At
    # --gpus 1 there is no group in the timed region; afterwards a ONE-rank RCCL communicator is created so that the same two
    # collectives run through RCCL and the three-stream ordering on this GPU (labelled: it measures the path, not xGMI)
    comm = dp.comm_info(dev)
    dp_coll = None
    if workload == "train" and not args.headline_only and not os.environ.get("VOTENET_BENCH_NO_DP_LEG"):
        try:
            one_rank = False
            if world == 1:
                import socket
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                s_ = socket.socket()
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
                s_.close()
                import datetime
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
                one_rank = True
            keep = net._gsync
            net._gsync = dp.GradSync(net.store, net.store.offset_of("sa3/"), overlap=True, force=True, profile=True)
            tms = []
            for _ in range(4):
                step()
                torch.cuda.synchronize()
                tms.append(net._gsync.timings())
            net._gsync = keep
            tms = tms[1:]
            dp_coll = {k: round(sum(t[k] for t in tms) / len(tms), 4) for k in tms[0]}
            dp_coll["collectives_per_step"] = [["tail", net.store.grad.numel() - net.store.offset_of("sa3/")], ["head", net.store.offset_of("sa3/")]]
            dp_coll["communicator"] = dp.comm_info(dev)
            dp_coll["what"] = ("mean of 3 steps, HIP events on the communication stream around each all-reduce (fp32 elements); tail_exposed = "
                               "how long the tail was still running when the main stream reached the end of the backward pass"
                               + ("; ONE-rank RCCL communicator created after the timed region: the path and its stream ordering, not xGMI"
                                  if one_rank else ""))
            if one_rank:
                dist.destroy_process_group()
        except Exception as e:  # one GPU: never lose the headline to this leg
            if world > 1:
                # N ranks: a rank that swallowed its exception would walk on to destroy_process_group while the others wait in the
                # collective it skipped -- the run would hang instead of failing.  Exit non-zero: the launcher ends the other ranks.
                raise
            dp_coll = {"error": repr(e)[:300]}
    # N > 1: the line validates its own data-parallel path -- overlapped exchange == blocking exchange on every rank, replicas identical
    # across ranks, N distinct devices in the communicator (check_dp; the same comparison as `--check-dp`, two steps)
    check = None
    if workload == "train" and world > 1 and args.scene == "room" and not os.environ.get("VOTENET_BENCH_NO_DP_LEG"):
        import votenet_amd
        prev_det = votenet_amd.set_deterministic(True)
        try:
            pair = [VM.VoteNetHotPath(dev, seed=0) for _ in range(2)]
            for rep in pair:
                dp.broadcast_params(rep.store)
                rep.init_optimizer()
            check = dp_self_check(dev, pair, lambda rep, i: rep.train_step(xs[i % len(xs)], None, world, gt=gts[i % len(xs)]), steps=2)
        finally:
            votenet_amd.set_deterministic(prev_det)
        del pair
    if rank == 0:
        # dominant kernel: the sa1 farthest-point-sampling launch (n=20480 -> 2048)
        m1 = net.sa1.npoint
        durs = [e0.elapsed_time(e1) for (e0, e1, b_, n_, m_) in events if n_ == n and m_ == m1]
        roof = None
        if durs:
            avg_ms = sum(durs) / len(durs)
            alg = B * (m1 - 1) * n * 16 + B * n * 12 + B * m1 * 4  # B*(m-1)*n*16 + B*n*12 + B*m*4
            # HBM bytes per launch come from a separate rocprofv3 --pmc pass (PMC counters cannot be read inside this process):
            # the committed summary profiles/pmc_latest.json, labelled as such -- null when the file is absent
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    traffic = pj.get("fps_sa1", {}).get("hbm_bytes_per_launch")
                    traffic_src = "profiles/pmc_latest.json (%s): separate rocprofv3 --pmc passes, not measured in this run" % pj.get("source", "")
                except Exception:
                    traffic = None
            ach = alg / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "sidx_* (spatial index: 5 launches) + fps_bucket_kernel<12,32> (sa1 FPS %d->%d, register "
                                             "resident, exact bucket pruning)" % (n, m1),
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "residency": "registers, %d CUs: one workgroup per scene keeps its cloud and running distances in the registers of ONE compute "
                                 "unit (%d of 256 CUs busy); `achieved` / `frac` price the reference algorithm's byte model (SURVEY 8d) against HBM -- "
                                 "on-chip, latency-bound per round, not an HBM rate; `traffic` is what the counters saw" % (B, B),
                    "model_only": True, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes": alg,
                    "alone_on_the_gpu": ({"avg_launch_ms": round(iso_fps, 4), "frac": round(alg / (iso_fps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                         if iso_fps else None)}
        # the sa1 ball query (n=20480 candidates, 2048 centres, K=64) and the pair the north star names: FPS + ball query
        bq = None
        K1 = net.sa1.nsample
        bdurs = [e0.elapsed_time(e1) for (e0, e1, b_, n_, m_, k_) in bq_events if n_ == n and m_ == m1]
        if bdurs and roof:
            bq_ms = sum(bdurs) / len(bdurs)
            bq_alg = B * m1 * n * 12 + B * m1 * (K1 + 1) * 4  # B*m*n*12 + B*m*(K+1)*4
            ach = bq_alg / (bq_ms * 1e-3) / 1e9
            both = (alg + bq_alg) / ((avg_ms + bq_ms) * 1e-3) / 1e9
            bq = {"bound": "hbm", "kernel": "ball_query_indexed_kernel<4> (sa1: %d candidates x %d centres, r=0.2, K=%d; only the buckets of "
                                            "the FPS's spatial index that reach into the ball, index-ordered read-out through an LDS bitmap)"
                                            % (n, m1, K1),
                  "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                  "avg_launch_ms": round(bq_ms, 4), "algorithmic_bytes": bq_alg,
                  "model_only": True,
                  "model_note": "achieved / frac price SURVEY 8d's ALL-PAIRS byte model (B m n 12 + B m (K+1) 4) at the launch time: the kernel tests "
                                "2-3 % of those pairs (scanned_pairs), so frac > 1 is the better algorithm, not an HBM rate -- see frac_scanned",
                  "fps_plus_ball_query": {"achieved": round(both, 1), "frac": round(both / HBM_PEAK_GBS, 4),
                                          "ms": round(avg_ms + bq_ms, 4), "algorithmic_bytes": alg + bq_alg,
                                          "alone_on_the_gpu": ({"ms": round(iso_fps + iso_bq, 4),
                                                                "frac": round((alg + bq_alg) / ((iso_fps + iso_bq) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                                               if iso_fps else None)}}
        mfma = None
        if gemm_events:
            gemm_events = [vmlp.resolve_event(ev) for ev in gemm_events]  # (launches sized for an upper bound: the rows that were there)
            # GEMMs run on two streams (weight gradients beside the input-gradient chain): the time the matrix pipes are
            # in use is the UNION of the launch intervals, not their sum.  All events are placed on one time axis by
            # their distance from the first one.
            base = gemm_events[0][0]
            iv = sorted((base.elapsed_time(e0), base.elapsed_time(e1)) for (e0, e1, *_r) in gemm_events)
            tot_ms, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for a, b in iv[1:]:
                if a > cur_e:
                    tot_ms += cur_e - cur_s
                    cur_s, cur_e = a, b
                else:
                    cur_e = max(cur_e, b)
            tot_ms += cur_e - cur_s
            sum_ms = sum(b - a for a, b in iv)
            tot_fl = sum(ev[3] for ev in gemm_events)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            mfma = {"bound": "mfma", "kernel": "mlp_linear_fast_kernel / mlp_linear_kernel / mlp_wgrad_fast_kernel (all %d GEMM launches of "
                                               "two steps right after the timed region, fp32 in / fp32 accumulate; executed flops / union of the "
                                               "launch intervals over both streams).  Forward GEMMs, Gram matrices and the Gram-form dense input gradient "
                                               "multiply each fp32 operand as two fp16 pieces (three v_mfma_f32_32x32x16_f16 per k-step, round 6); the "
                                               "input-gradient and weight-gradient GEMMs as three bf16 pieces (x = hi + mid + lo exactly: six "
                                               "v_mfma_f32_32x32x16_bf16 per k-step; row-major images, fragments through ds_read_b64_tr_b16 in the weight gradients); "
                                               "fp32 accumulate, error vs float64 equal to the fp32 MFMA kernel's.  flops = fp32 multiply-adds of the GEMM (2 rows cin "
                                               "cout), priced against the fp32 MFMA peak" % len(gemm_events),
                    "achieved": round(ach, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4),
                    "gemm_ms_per_step": round(tot_ms / gemm_steps, 3), "gemm_ms_per_step_summed": round(sum_ms / gemm_steps, 3),
                    "gflop_per_step": round(tot_fl / gemm_steps / 1e9, 1)}
            # the same launches by family (the note of mlp._Timed: forward with the pooling epilogue, forward with statistics, fused
            # BatchNorm-backward input-gradient GEMM with the reduce of the layer below, weight gradients, ...): executed flops / the
            # SUM of the family's launch durations inside the step (other streams run beside them)
            fam = {}
            for ev in gemm_events:
                note = ev[4][3] if len(ev) > 4 and ev[4] else ev[2]
                f = fam.setdefault(note, [0.0, 0.0, 0])
                f[0] += ev[3]
                f[1] += ev[0].elapsed_time(ev[1])
                f[2] += 1
            # rocprof's MFMA utilisation of each family's kernel (matrix-pipe busy share, kernel alone on the GPU) cannot be read inside
            # this process: quoted from the committed counter summary, labelled with its source like `traffic` above
            util, util_src = {}, None
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("mlp_families", {})
                util, util_src = pj.get("families", {}), pj.get("source")
            except Exception:
                pass

            def fam_entry(k, v):
                tf = v[0] / (v[1] * 1e-3) / 1e12
                e = {"launches_per_step": round(v[2] / gemm_steps, 1), "gflop_per_step": round(v[0] / gemm_steps / 1e9, 1),
                     "ms_per_step": round(v[1] / gemm_steps, 3), "tflops": round(tf, 1), "frac": round(tf / MFMA_F32_PEAK_TF, 3),
                     "frac_bf3_equiv": round(tf / MFMA_BF3_EQUIV_TF, 3)}
                u = util.get(k)
                if u:
                    e["mfma_util_alone"] = u.get("mfma_util")
                    e["tflops_alone"] = u.get("alone_tflops")
                    if u.get("hbm_gbs_alone") is not None:  # the piece-layout families: the same launch's HBM rate (FETCH + WRITE counters)
                        e["hbm_gbs_alone"], e["hbm_frac_alone"] = u["hbm_gbs_alone"], u.get("hbm_frac_alone")
                return e
            mfma["by_family"] = {k: fam_entry(k, v) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:10]}
            # ONE number for the north star's own metric (rocprof-reported MFMA utilisation of the grouped MLP): the families' alone-on-the-
            # GPU MfmaUtil weighted by the time each family takes inside the step, over the families the counter summary covers
            cov = [(v[1], util[k]["mfma_util"]) for k, v in fam.items() if util.get(k) and util[k].get("mfma_util") is not None]
            if cov:
                tsum = sum(t for t, _ in cov)
                mfma["mfma_util_step_weighted"] = {"value": round(sum(t * u for t, u in cov) / tsum, 3),
                                                   "covers_frac_of_gemm_time": round(tsum / sum(v[1] for v in fam.values()), 3),
                                                   "what": "sum over GEMM families of (rocprofv3 MfmaUtil of the family's kernel alone on the GPU) x "
                                                           "(the family's launch time inside the step) / that time; families without a counter "
                                                           "entry (the small layers of fp / voting / proposal) are left out"}
            mfma["by_family_note"] = ("tflops / frac: the family's executed flops / the SUM of its launch durations INSIDE the step (two GEMM streams and "
                                      "the next batch's geometry run beside them); mfma_util_alone / tflops_alone: rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES "
                                      "share and rate of the same kernel alone on the GPU at sa2's / sa1's shape, hbm_gbs_alone / hbm_frac_alone: "
                                      "its HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) over its duration and the share of 8 TB/s -- the piece-layout "
                                      "kernels are bound by their row traffic, not by the matrix pipe -- from %s and %s (separate --pmc passes, "
                                      "not measured in this run)" % (util_src, pj.get("source_pieces"))) if util_src else None
            mfma["peak_bf3_equiv"] = MFMA_BF3_EQUIV_TF
            mfma["frac_bf3_equiv"] = round(ach / MFMA_BF3_EQUIV_TF, 4)
            mfma["peak_h2_equiv"] = MFMA_H2_EQUIV_TF
            mfma["pricing"] = ("frac prices the fp32 multiply-adds against the fp32 MFMA peak (157.3 TFLOP/s: what a kernel on v_mfma_f32_32x32x2_f32 "
                               "could reach at most); the kernels issue six v_mfma_f32_32x32x16_bf16 per product on exactly split operands, whose "
                               "ceiling is the dense bf16 peak / 6 = %.1f TFLOP/s of fp32 multiply-adds: frac_bf3_equiv is the utilisation of the "
                               "pipe that is actually used by the gradient GEMMs; the forward GEMMs, the Gram matrices and the Gram-form dense input "
                               "gradient (round 6) issue THREE v_mfma_f32_32x32x16_f16 per product: ceiling %.1f (peak_h2_equiv)" % (MFMA_BF3_EQUIV_TF, MFMA_H2_EQUIV_TF))
            if workload == "train" and B == 8 and n == 20480:
                # SURVEY.md 8(d)'s ALGORITHMIC figure for the grouped MLP: flops = 2 rows sum(C_in C_out) of the reference's
                # formulation (conv over the materialised grouped tensor): 186.0 GFLOP forward at B = 8, backward = 2 x forward.
                # The path EXECUTES fewer (first SA layer as a per-point GEMM before the grouping, pooled layers' backward in
                # Gram form): `frac` above is executed flops; this is the same time priced at the reference's work.
                alg_fl = 3 * 186.0e9
                mfma["algorithmic"] = {"gflop_per_step": round(alg_fl / 1e9, 1), "achieved": round(alg_fl / (tot_ms / gemm_steps * 1e-3) / 1e12, 1),
                                       "frac": round(alg_fl / (tot_ms / gemm_steps * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                                       "what": "SURVEY 8d algorithmic flops of the reference's formulation (186.0 GFLOP forward at B=8, x3 for "
                                               "fwd+bwd) / the same GEMM time; frac above counts only the flops this path executes"}
        cpu = None
        t_legs0 = time.perf_counter()
        # every BASELINE.json configuration in this line (1: single SA layer, 2: backbone forward, 3b: predict tower + NMS,
        # 5: dense 80000-pt scan; 3a = the headline value, 4 = this command with --gpus 8) + what the ball query really scans
        cfgs = bq_detail = None
        if world == 1 and not args.headline_only:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_legs
            cfgs = bench_legs.config_legs(net, xs, gts, dev, B, n, cpu=not args.no_cpu_baseline)
            cfgs["config3_train_step"] = "= the headline value of this line (forward + loss graph + backward + clip/Adam)"
            cfgs["config4_dp8"] = "python bench.py --gpus 8 (one rank per GPU, RCCL all-reduce of the flat gradient bucket, tail overlapped with backward)"
            bq_detail = bench_legs.ball_query_detail(net.sa1, xs[0], torch.from_numpy(synth.uniform_batch(B, n, 1000)).to(dev))
            if bq is not None:
                bq["alone_on_the_gpu_detail"] = bq_detail
                # what the launch REALLY scans: the (query, candidate) pairs the indexed kernel tests (room scenes) x 12 bytes, over the
                # launch time -- the all-pairs figure above is SURVEY 8d's input-independent model, not a rate this kernel achieves
                sp = bq_detail["room"]["scanned_pairs"]["indexed_kernel"]
                bq["frac_scanned"] = round(sp * 12 / (bq["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                bq["frac_scanned_alone"] = round(sp * 12 / (bq_detail["room"]["ms_alone"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                bq["scanned_pairs"] = sp
                bq["scanned_note"] = ("frac_scanned = scanned_pairs x 12 B / avg_launch_ms / 8 TB/s (frac_scanned_alone: / the launch alone on the GPU); "
                                      "the pairs come out of L2 / LDS (one bucket of 64 points serves every lane of a wave), so this is not an HBM "
                                      "rate either: the launch is bound by its index walk and read-out, ~%d us" % round(bq_detail["room"]["ms_alone"] * 1e3))
        torch.cuda.synchronize()
        t_end_gpu = time.perf_counter()
        # config_legs' CPU-oracle timing of config 1 (a few seconds of host work with an idle GPU) is inside t_legs: stated below
        gpu_busy = {"setup_and_warmup_s": round(t_setup, 3), "timed_region_s": round(dt, 4),
                    "after_the_timed_region_s": round(t_end_gpu - t_phase0 - t_setup - dt, 3),
                    "total_s": round(t_end_gpu - t_phase0, 3), "run_wall_s_so_far": round(t_end_gpu - T_PROCESS_START, 3),
                    "what": "wall seconds (host clock between device synchronisations) of the phases in which this process kept the GPU's queues "
                            "fed: set-up + warm-up steps, the timed region, and everything after it (roofline / variant steps, the kernels alone, "
                            "the config legs: %.1f s, of which a few seconds are the CPU oracle's config-1 layer with an idle GPU); an UPPER bound of "
                            "GPU-busy time -- a 5 s utilisation sampler sees 0 %% because the timed region is %.2f s long and the rest of the run is "
                            "the CPU baseline and the interpreter's start" % (t_end_gpu - t_legs0, dt)}
        if world == 1 and not args.no_cpu_baseline:  # last: its OpenMP threads keep the host busy for a while after they finish
            if pinned and full_mask:
                hostpin.unpin(full_mask)  # the CPU oracle runs on ALL cores: every thread of the process gets the full mask back
            cpu = cpu_baseline(n, args.scene)
        out = {
            "metric": "SUN RGB-D 20k-pt scenes/sec (%s)" % ("fwd+bwd" if workload == "train" else "fwd"),
            "value": round(B * world * args.steps / dt, 2), "unit": "scenes/s", "n_gpus": comm["world_size"], "steps": args.steps,
            "warmup": args.warmup, "setup_steps": SETUP_STEPS, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "host_threads": {"cpus": pinned, "what": "CPUs this rank's host threads were confined to for the GPU legs (votenet_amd/hostpin.py: eight "
                                                     "cores of the GPU's NUMA node; null = left to the scheduler; VOTENET_NO_PIN=1 switches it off)"},
            "config": {"workload": ("VoteNet hot path %s: sa1-4 + fp1-2 + voting + proposal, %d scenes x %d pts per GPU, "
                                    "%s scenes" % (("train step (fwd + loss graph of model.py:61-84,141-231 + bwd + clip/Adam)" if args.scene == "room" else
                                                    "train step (fwd+bwd+Adam, fixed cotangents)")
                                                   if workload == "train" else "forward", B, n, args.scene)
                                    + ("; three batches rotate, the coordinate-only geometry of the next batch (FPS, ball query, "
                                       "three_nn) runs on a side stream underneath the current step, replayed as one HIP graph (model.GeometryGraph); the step's "
                                       "static stretch (fp1 forward ... fp1 backward: ~85 launches) replays as four HIP graphs (model.StretchGraph)" if pipeline else "")),
                       "global_batch": B * world, "points": n, "parallelism": "dp%d" % world},
            "gemm_arithmetic": "fp32 in / fp32 out / fp32 accumulate.  FORWARD operands (activations behind a BatchNorm, coordinates, weights) are "
                               "multiplied as fp16 x 2 split operands (round 6: x = hi + lo, 22 bits, three v_mfma_f32_32x32x16_f16 per product, both "
                               "operands scaled by exact powers of two so that the lo pieces stay normal fp16 numbers; measured error against float64 "
                               "equal to the bf16 x 3 form's and the fp32 chain's: profiles/r06_mfma_f16_denorm.txt, tests/test_gpu_h2.py) -- every "
                               "forward GEMM, the Gram matrices and the Gram-form dense input gradient (its on-the-fly matrix W diag(C) W^T is scaled "
                               "into fp16's range by powers of two from row norms and max|C|).  GRADIENT operands stay bf16 x 3 (exact split, 6 of the 9 "
                               "cross terms, six v_mfma_f32_32x32x16_bf16 per product): fp32's range, and an ablation shows nothing to gain there "
                               "(profiles/r06_ablate_h2_bwd.txt).  Tests hold every form to the same tolerances",
            "ms_per_step_spread": spread, "without_cross_step_pipelining": in_step, "deterministic_mode": det_step,
            "fp32_mfma_gemms": fp32_step, "bf16x3_forward_operands": bf3_step, "full_row_layout": full_step, "row_layout": row_layout, "configs": cfgs,
            "communicator": comm, "per_rank": per_rank, "dp_collectives": dp_coll, "check_dp": check,
            "gpu_busy_seconds": gpu_busy,
            "roofline": roof, "roofline_ball_query": bq, "roofline_mlp": mfma, "cpu_baseline": cpu,
        }
        emit(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if check is not None and not check["equal_everywhere"]:
        sys.stderr.write("bench.py: the data-parallel replicas diverged (check_dp: %s)\n" % json.dumps(check))
        sys.exit(4)


if __name__ == "__main__":
    main()
