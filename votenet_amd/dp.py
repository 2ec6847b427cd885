"""Data-parallel glue: scenes shard across GPUs, one process per GPU, ONE all-reduce per step.

The reference is single-GPU (SimpleTrainer, run.py:136) and has no collective at all; every op of the
hot path is independent per scene (b is the outermost index of every reference loop), so the batch
dimension shards with no data-path collective.  The only exchange is the gradient sum: the whole
model is 955 k fp32 parameters (3.8 MB), kept in ONE flat bucket (pointnet2.ParamStore), so a step
issues exactly one RCCL all-reduce -- latency-bound over xGMI, not bandwidth-bound; bucketing it
further would only add launches.  BatchNorm statistics stay per replica.
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
