"""Data-parallel glue: scenes shard across GPUs, one process per GPU, the gradient sum is the only exchange.

The reference is single-GPU (SimpleTrainer, run.py:136) and has no collective at all; every op of the
hot path is independent per scene (b is the outermost index of every reference loop), so the batch
dimension shards with no data-path collective.  The whole model is 955 k fp32 parameters (3.8 MB) in
ONE flat bucket (pointnet2.ParamStore), laid out in forward order sa1 | sa2 | sa3 | sa4 | fp1 | fp2 | voting |
proposal.  The backward pass runs the other way round, so when sa3's backward has been enqueued the whole
TAIL of the bucket (sa3 ... proposal: 92 % of the bytes) is final while the two largest layers' backward
(sa2, sa1: ~45 % of the backward time) has not started.  GradSync therefore issues

    * the tail all-reduce from a communication stream that waits for the main and the weight-gradient stream
      at that point -- it runs over xGMI underneath sa2's / sa1's backward GEMMs, and
    * the head all-reduce (sa1 + sa2, 0.3 MB: pure latency) after the last weight gradient,

and the optimizer waits for both.  Two collectives per step, each one contiguous slice of the same flat
bucket; at these sizes both are latency-bound over xGMI, so the bucket is never split further.  BatchNorm
statistics stay per replica.
"""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def scene_seeds(rank_, per_gpu, base=1000):
    """Disjoint synthetic-scene seeds per rank (weak scaling: per-GPU batch fixed)."""
    return [base + rank_ * per_gpu + i for i in range(per_gpu)]


def broadcast_params(store, src=0):
    """All replicas start from rank 0's parameters (one broadcast of the flat bucket)."""
    if world_size() > 1:
        dist.broadcast(store.flat, src)


def sync_gradients(store):
    """Sum the flat gradient bucket over ranks with a single collective; returns the scale (1/world)
    that the optimizer folds into its update (votenet_clip_adam grad_scale)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(store.grad, op=dist.ReduceOp.SUM)
    return 1.0 / w


def _device_identity(dev):
    """What tells two GPUs of one node apart: PCI address and uuid from the device properties (each guarded: older torch builds lack them)."""
    pr = torch.cuda.get_device_properties(dev)
    ident = {"index": dev.index if dev.index is not None else torch.cuda.current_device(), "name": pr.name}
    for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
        if hasattr(pr, k):
            ident[k] = int(getattr(pr, k))
    if hasattr(pr, "uuid"):
        ident["uuid"] = str(pr.uuid)
    return ident


def comm_info(dev=None):
    """What the COMMUNICATOR reports (not the environment): world size, backend, and every rank's device identity gathered over the
    process group itself -- so that a bench line can show N ranks on N distinct devices.  Single process: world 1, backend None."""
    if not (dist.is_available() and dist.is_initialized()):
        me = _device_identity(dev) if (dev is not None and dev.type == "cuda") else {"index": None, "name": "cpu"}
        return {"world_size": 1, "backend": None, "devices": [dict(me, rank=0)], "distinct_devices": 1}
    me = _device_identity(dev) if (dev is not None and dev.type == "cuda") else {"index": None, "name": "cpu", "pid": __import__("os").getpid()}
    me["rank"] = dist.get_rank()
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, me)
    key = lambda d: (d.get("uuid"), d.get("pci_domain_id"), d.get("pci_bus_id"), d.get("pci_device_id"), d.get("index"), d.get("pid"))
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices": got, "distinct_devices": len({key(d) for d in got})}


class GradSync:
    """The gradient exchange of one training step, overlapped with the backward pass (see the module docstring).

        gs = GradSync(store, split)        split = first element of the tail inside the flat bucket
        gs.start_tail(streams)             when every gradient of bucket[split:] has been ENQUEUED (on `streams` / the
                                           current stream): the all-reduce starts as soon as those streams get there
        scale = gs.finish(streams)         after the last weight gradient: head all-reduce, then the current stream waits
                                           for both; returns 1/world for the optimizer

    With world == 1 every call is a no-op (unless force=True and a process group exists: a one-rank communicator then still runs
    both collectives -- how the 1-GPU box puts the stream ordering in front of RCCL itself).  On CUDA tensors the collectives are
    issued from a dedicated HIP stream so that neither the main stream nor the weight-gradient stream ever waits for the network
    before finish().

    overlap=False is the control the overlapped form is checked against (bench.py --check-dp, tests): no tail under the backward, and
    finish() drains the device, runs BLOCKING all-reduces on the current stream and drains again -- nothing can race with them by
    construction.  The control reduces the SAME two slices (bucket[split:], then bucket[:split]) as the overlapped form: a ring /
    tree all-reduce adds an element's contributions in an order that depends on where the element sits in the reduced buffer, so with
    more than two ranks ONE all-reduce of the whole bucket gives sums that differ in the last bit from two all-reduces of its slices --
    a bit-equality check against it fails with nothing wrong (found by the world-8 gloo test, round 6; world 2 cannot see it: a + b
    has one order).  same_slices=False keeps the old single collective.
    profile=True records HIP events around both collectives: after finish(), timings() gives their latency on the communication
    stream and how much of the tail had finished before the main stream arrived at finish() (= hidden under sa2 / sa1)."""

    def __init__(self, store, split, overlap=True, force=False, profile=False, same_slices=True):
        self.store, self.split = store, int(split)
        self.overlap, self.force, self.profile = bool(overlap), bool(force), bool(profile)
        self.same_slices = bool(same_slices)
        self._work = []
        self._comm = None
        self._marks = {}
        self.log = []  # (what, numel) per collective of the last step: tests read it

    def _active(self):
        if self.force and dist.is_available() and dist.is_initialized():
            return True
        return world_size() > 1

    def _issue(self, t, what, streams):
        if t.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=t.device)
            cur = torch.cuda.current_stream()
            for s in [cur] + [s for s in (streams or []) if s is not None]:
                ev = torch.cuda.Event()
                ev.record(s)
                self._comm.wait_event(ev)
            with torch.cuda.stream(self._comm):
                if self.profile:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self._comm)
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
                if self.profile:
                    w.wait()  # the COMMUNICATION stream waits for the collective (device side), so that an event on it marks its end
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self._comm)
                    self._marks[what] = (e0, e1)
        else:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        self._work.append(w)
        self.log.append((what, t.numel()))

    def begin(self):
        self._work, self.log = [], []
        self._marks = {}
        self._tail_started = False

    def start_tail(self, streams=None):
        if self.overlap and self._active() and 0 < self.split < self.store.grad.numel():
            self._issue(self.store.grad[self.split:], "tail", streams)
            self._tail_started = True

    def finish(self, streams=None):
        w = world_size()
        if self._active():
            g = self.store.grad
            if not self.overlap:
                if g.is_cuda:
                    torch.cuda.synchronize(g.device)
                if self.same_slices and 0 < self.split < g.numel():
                    for what, t in (("blocking tail", g[self.split:]), ("blocking head", g[:self.split])):
                        dist.all_reduce(t, op=dist.ReduceOp.SUM)
                        self.log.append((what, t.numel()))
                else:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM)
                    self.log.append(("blocking", g.numel()))
                if g.is_cuda:
                    torch.cuda.synchronize(g.device)
                self._tail_started = False
                return 1.0 / w
            if g.is_cuda and self.profile:
                arrive = torch.cuda.Event(enable_timing=True)
                arrive.record(torch.cuda.current_stream())
                self._marks["main_arrives"] = arrive
            if getattr(self, "_tail_started", False):
                self._issue(g[:self.split], "head", streams)
            else:
                self._issue(g, "all", streams)
            for wk in self._work:
                wk.wait()  # NCCL/RCCL: the CURRENT stream waits for the collective (no host sync); gloo: the host waits
            self._work = []
        self._tail_started = False
        return 1.0 / w

    def timings(self):
        """After a profiled step (and a device synchronise): {'tail_ms', 'head_ms' | 'all_ms', 'tail_exposed_ms', 'tail_hidden_frac'}.
        tail_exposed = how long after the main stream reached finish() the tail collective was still running."""
        out = {}
        for what in ("tail", "head", "all"):
            if what in self._marks:
                e0, e1 = self._marks[what]
                out[what + "_ms"] = round(e0.elapsed_time(e1), 4)
        if "tail" in self._marks and "main_arrives" in self._marks:
            exposed = max(0.0, self._marks["main_arrives"].elapsed_time(self._marks["tail"][1]))
            out["tail_exposed_ms"] = round(exposed, 4)
            out["tail_hidden_frac"] = round(1.0 - min(1.0, exposed / max(out["tail_ms"], 1e-9)), 4)
        return out


def check_overlap_against_blocking(net_a, net_b, run_step, steps=2):
    """The overlapped exchange cannot be told from a blocking one: net_a and net_b are two replicas in the SAME state (same
    parameters, optimizer state, deterministic mode on); run_step(net, i) runs training step i on a net.  net_a exchanges its
    gradients with GradSync's overlap (tail issued after sa3's backward from the communication stream, head after the last weight
    gradient), net_b with blocking all-reduces of the same two slices between two device synchronisations (same slices: see GradSync --
    the reduction order inside a collective depends on the buffer it reduces).  Any missing stream dependency in the
    overlapped form (a collective that starts before a gradient is final, an optimizer that starts before a collective is done)
    shows as different parameters.  -> dict(equal_on_this_rank, ranks_identical, equal_everywhere, collectives, timings):
    equal_everywhere is the AND over all ranks (one MIN all-reduce)."""
    split = net_a.store.offset_of("sa3/")
    on_gpu = net_a.store.flat.is_cuda
    net_a._gsync = GradSync(net_a.store, split, overlap=True, force=True, profile=on_gpu)
    net_b._gsync = GradSync(net_b.store, split, overlap=False, force=True)
    logs = []
    for i in range(steps):
        run_step(net_a, i)
        logs.append(list(net_a._gsync.log))
        run_step(net_b, i)
    if on_gpu:
        torch.cuda.synchronize(net_a.store.flat.device)
    same = bool(torch.equal(net_a.store.flat, net_b.store.flat))
    identical = True
    everywhere = same
    if dist.is_available() and dist.is_initialized():
        ref = net_a.store.flat.clone()
        dist.broadcast(ref, 0)
        identical = bool(torch.equal(ref, net_a.store.flat))
        flag = torch.tensor([1.0 if (same and identical) else 0.0], device=net_a.store.flat.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        everywhere = bool(flag.item() == 1.0)
    return {"equal_on_this_rank": same, "ranks_identical": identical, "equal_everywhere": everywhere, "steps": steps,
            "collectives_per_step": logs[-1] if logs else [], "timings_last_step": net_a._gsync.timings() if on_gpu else {}}
