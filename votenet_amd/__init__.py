"""votenet_amd -- MI355X (gfx950) implementation of the VoteNet / PointNet++ point-cloud hot path.

The package holds only what the path needs:
  csrc/            hand-written HIP kernels + the C ABI (include/votenet_hip.h)
  lib/             the built libvotenet_hip.so (in-tree, not in git)
  tf_sampling, tf_grouping, tf_interpolate, tf_nms3d
                   host-side mirrors of the reference's tf_ops Python modules: same function
                   names, argument order and return arity, on torch (ROCm) tensors
  pointnet2        mirror of the SA / FP layer code of the reference's utils.py on the fused kernels

There is no CPU fallback: importing an op module without the built library raises.
"""
from ._lib import InvalidArgumentError, VotenetError, build, lib_path  # noqa: F401



def set_deterministic(on=True):
    """Bit-reproducible backward pass (no fp32 atomics): see votenet_amd.mlp.DETERMINISTIC."""
    from . import mlp
    return mlp.set_deterministic(on)


__all__ = ["InvalidArgumentError", "VotenetError", "build", "lib_path", "set_deterministic"]
