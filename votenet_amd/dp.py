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


class GradSync:
    """The gradient exchange of one training step, overlapped with the backward pass (see the module docstring).

        gs = GradSync(store, split)        split = first element of the tail inside the flat bucket
        gs.start_tail(streams)             when every gradient of bucket[split:] has been ENQUEUED (on `streams` / the
                                           current stream): the all-reduce starts as soon as those streams get there
        scale = gs.finish(streams)         after the last weight gradient: head all-reduce, then the current stream waits
                                           for both; returns 1/world for the optimizer

    With world == 1 every call is a no-op.  On CUDA tensors the collectives are issued from a dedicated HIP stream so that
    neither the main stream nor the weight-gradient stream ever waits for the network before finish()."""

    def __init__(self, store, split):
        self.store, self.split = store, int(split)
        self._work = []
        self._comm = None
        self.log = []  # (what, numel) per collective of the last step: tests read it

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
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        else:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        self._work.append(w)
        self.log.append((what, t.numel()))

    def begin(self):
        self._work, self.log = [], []
        self._tail_started = False

    def start_tail(self, streams=None):
        if world_size() > 1 and 0 < self.split < self.store.grad.numel():
            self._issue(self.store.grad[self.split:], "tail", streams)
            self._tail_started = True

    def finish(self, streams=None):
        w = world_size()
        if w > 1:
            g = self.store.grad
            if getattr(self, "_tail_started", False):
                self._issue(g[:self.split], "head", streams)
            else:
                self._issue(g, "all", streams)
            for wk in self._work:
                wk.wait()  # NCCL/RCCL: the CURRENT stream waits for the collective (no host sync); gloo: the host waits
            self._work = []
        self._tail_started = False
        return 1.0 / w
