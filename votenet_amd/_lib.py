"""ctypes binding of libvotenet_hip.so (the C ABI declared in include/votenet_hip.h).

torch is used for device memory and streams only: every call passes raw device pointers,
sizes and the current HIP stream across the C ABI.  No fallback path exists -- if the library
is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libvotenet_hip.so")
_lib = None


class VotenetError(RuntimeError):
    """A HIP runtime / launch / workspace error reported by libvotenet_hip.so."""


class InvalidArgumentError(ValueError):
    """Mirror of tf.errors.InvalidArgumentError raised by the reference's OP_REQUIRES checks."""


def lib_path():
    return _LIB_PATH


def build(force=False):
    """Compile libvotenet_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:  # a clean build: the library AND every cached object file (build.sh recompiles what is missing)
        import glob
        for f in [_LIB_PATH] + glob.glob(os.path.join(_HERE, "csrc", "obj", "*.o")):
            if os.path.exists(f):
                os.remove(f)
    out = subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], capture_output=True, text=True)
    if out.returncode != 0:
        raise VotenetError("libvotenet_hip.so build failed:\n" + out.stdout + out.stderr)
    return _LIB_PATH


_c_f = ctypes.c_void_p  # all device pointers cross the ABI as void*
_SIGS = {
    "votenet_farthest_point_sample": [ctypes.c_int] * 3 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_gather_point": [ctypes.c_int] * 3 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_gather_point_grad": [ctypes.c_int] * 3 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_query_ball_point": [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_query_ball_point_indexed": [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_spatial_index": [ctypes.c_int] * 2 + [_c_f] * 2 + [ctypes.c_void_p],
    "votenet_group_point": [ctypes.c_int] * 5 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_group_point_grad": [ctypes.c_int] * 5 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_three_nn": [ctypes.c_int] * 3 + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_three_nn_weights": [ctypes.c_int] * 2 + [_c_f] * 2 + [ctypes.c_void_p],
    "votenet_three_interpolate": [ctypes.c_int] * 4 + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_three_interpolate_grad": [ctypes.c_int] * 4 + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_iou3d_matrix": [ctypes.c_int] * 2 + [_c_f] * 2 + [ctypes.c_void_p],
    "votenet_nms3d": [ctypes.c_int] * 2 + [_c_f] * 3 + [ctypes.c_float] + [_c_f] * 3 + [ctypes.c_size_t, ctypes.c_void_p],
}


class BnRaw(ctypes.Structure):
    """struct votenet_bn_raw (include/votenet_hip.h)."""
    _fields_ = [("stats", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("rows", ctypes.c_long),
                ("eps", ctypes.c_float), ("out", ctypes.c_void_p)]


class CoefTail(ctypes.Structure):
    """struct votenet_coef_tail (include/votenet_hip.h)."""
    _fields_ = [("ticket", ctypes.c_void_p), ("rows", ctypes.c_long), ("gamma", ctypes.c_void_p), ("coef", ctypes.c_void_p),
                ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p)]


class MlpInput(ctypes.Structure):
    """struct votenet_mlp_input (include/votenet_hip.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("in_scale", ctypes.c_void_p), ("in_shift", ctypes.c_void_p),
                ("in_relu", ctypes.c_int), ("in_bn", ctypes.POINTER(BnRaw)),
                ("xyz", ctypes.c_void_p), ("new_xyz", ctypes.c_void_p), ("feat", ctypes.c_void_p), ("idx", ctypes.c_void_p),
                ("b", ctypes.c_int), ("n", ctypes.c_int), ("m", ctypes.c_int), ("nsample", ctypes.c_int), ("c", ctypes.c_int)]


_SIGS.update({
    "votenet_mlp_linear": [ctypes.POINTER(MlpInput), ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_group_linear": [ctypes.c_int] * 5 + [_c_f] * 8 + [ctypes.c_void_p],
    "votenet_group_linear_backward": [ctypes.c_int] * 5 + [_c_f] * 7 + [ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_mlp_linear_pool": [ctypes.POINTER(MlpInput), ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_int]
                               + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_bn_pool_finalize": [ctypes.c_long, ctypes.c_int] + [_c_f] * 6 + [ctypes.POINTER(BnRaw), ctypes.c_int] + [_c_f] * 3
                                + [ctypes.c_void_p],
    "votenet_pool_backward_supported": [ctypes.c_int] * 3,
    "votenet_bn_backward_reduce_pool": [ctypes.c_long, ctypes.c_int] + [_c_f] * 6 + [ctypes.c_float, ctypes.c_int, _c_f,
                                        ctypes.POINTER(CoefTail), ctypes.c_void_p],
    "votenet_pool_dgrad_prepare": [ctypes.c_int] * 2 + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_pool_dgrad_prepare_split": [ctypes.c_int] * 2 + [_c_f] * 5 + [ctypes.c_void_p, ctypes.c_void_p],
    "votenet_pool_dgrad_scatter": [ctypes.c_long] + [ctypes.c_int] * 3 + [_c_f] * 4 + [ctypes.c_int] + [_c_f] * 7
                                  + [ctypes.c_float, ctypes.c_int, _c_f, ctypes.POINTER(CoefTail), ctypes.c_void_p],
    "votenet_mlp_gram": [ctypes.c_long, ctypes.c_int] + [_c_f] * 2 + [ctypes.c_int, _c_f, _c_f, ctypes.c_void_p],
    "votenet_pool_wgrad_sparse": [ctypes.c_long] + [ctypes.c_int] * 3 + [_c_f] * 3 + [ctypes.c_int] + [_c_f] * 4 + [ctypes.c_int]
                                 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_pool_wgrad_finish": [ctypes.c_int] * 2 + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_loss": [ctypes.c_int] * 7 + [_c_f] * 12 + [ctypes.c_float] * 2 + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_loss_pitched": [ctypes.c_int] * 7 + [_c_f] * 4 + [ctypes.c_long] + [_c_f] * 8 + [ctypes.c_float] * 2 + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_decode_boxes": [ctypes.c_int] * 5 + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_iou3d_cross": [ctypes.c_int] * 3 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_selection_sort": [ctypes.c_int] * 4 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_knn_point": [ctypes.c_int] * 5 + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_prob_sample": [ctypes.c_int] * 3 + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_subsample_augment": [ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_int, _c_f, _c_f, ctypes.c_ulonglong,
                                  ctypes.c_long, ctypes.c_int] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_augment_boxes": [ctypes.c_int, ctypes.c_int] + [_c_f] * 11 + [ctypes.c_int, ctypes.c_int] + [_c_f] * 8 + [ctypes.c_void_p],
    "votenet_transpose_segments": [ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_split_weights": [ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_split_weights_h2": [ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_register_split_weights_pieces": [_c_f, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int],
    "votenet_register_split_weights_scaled": [_c_f, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, _c_f, _c_f],
    "votenet_pool_dgrad_prepare_h2": [ctypes.c_int, ctypes.c_int] + [_c_f] * 8 + [ctypes.c_void_p],
    "votenet_split_weights_one": [_c_f, ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_mlp_split_k_arm": [ctypes.c_void_p, ctypes.c_long],
    "votenet_mlp_split_k_tickets": [ctypes.c_void_p, ctypes.c_long],
    "votenet_register_split_weights": [_c_f, ctypes.c_int, ctypes.c_int, ctypes.c_void_p],
    "votenet_bn_finalize": [ctypes.c_long, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_float] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_bn_relu_max": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int] + [_c_f] * 2 + [ctypes.c_void_p],
    "votenet_bn_relu": [ctypes.c_long, ctypes.c_int] + [_c_f] * 3 + [ctypes.POINTER(BnRaw), ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_bn_backward_reduce": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 7 + [ctypes.c_float, ctypes.c_int, _c_f,
                                                                                             ctypes.POINTER(CoefTail), ctypes.c_void_p],
    "votenet_bn_backward_apply": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_bias_grad": [ctypes.c_long, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_mlp_wgrad": [ctypes.POINTER(MlpInput), ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_bn_backward_coef": [ctypes.c_long, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_float] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_mlp_wgrad_bn": [ctypes.POINTER(MlpInput), ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int]
                            + [_c_f] * 2 + [ctypes.c_int, _c_f, _c_f, ctypes.c_void_p],
    "votenet_mlp_dgrad_bn": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int] + [_c_f] * 2 + [ctypes.c_int]
                            + [_c_f] * 2 + [ctypes.c_void_p],
    "votenet_mlp_dgrad_bn_reduce": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int] + [_c_f] * 7
                                   + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(CoefTail), ctypes.c_void_p],
    "votenet_assemble_rows": [ctypes.c_int] * 4 + [_c_f] * 7 + [ctypes.c_void_p],
    "votenet_assemble_stats": [ctypes.c_long, ctypes.c_int] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_assemble_z0": [ctypes.c_long, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_assembled_linear": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 5 + [ctypes.POINTER(BnRaw), ctypes.c_int]
                                + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_assembled_wgrad_bn": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 5 + [ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int, _c_f, _c_f,
                                   ctypes.c_void_p],
    "votenet_assembled_dgrad_bn_reduce": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int] + [_c_f] * 9
                                         + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(CoefTail), ctypes.c_void_p],
    "votenet_group_linear_backward_assembled": [ctypes.c_int] * 5 + [_c_f] * 8 + [ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_narrow_rows": [ctypes.c_int] * 5 + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_narrow_z0": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_narrow_stats": [ctypes.c_long, ctypes.c_int, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_narrow_linear": [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_c_f] * 5 + [ctypes.POINTER(BnRaw), ctypes.c_int]
                             + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_narrow_wgrad_bn": [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_c_f] * 5 + [ctypes.c_int] + [_c_f] * 3
                               + [ctypes.c_int, _c_f, _c_f, ctypes.c_void_p],
    "votenet_narrow_dgrad_bn_reduce": [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_int] + [_c_f] * 8
                                      + [ctypes.c_float, ctypes.c_int, _c_f, _c_f, ctypes.POINTER(CoefTail), ctypes.c_void_p],
    "votenet_narrow_wgrad_first": [ctypes.c_int, ctypes.c_int] + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_group_concat_grad": [ctypes.c_int] * 5 + [_c_f] * 7 + [ctypes.c_void_p],
    "votenet_inverse_index": [ctypes.c_int] * 3 + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_csr_gather_sum": [ctypes.c_long, ctypes.c_int] + [_c_f] * 4 + [ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_csr_gather_sum_pitched": [ctypes.c_long, ctypes.c_int, _c_f, ctypes.c_long] + [_c_f] * 3 + [ctypes.c_int, _c_f, ctypes.c_void_p],
    "votenet_group_linear_backward_csr": [ctypes.c_int] * 5 + [_c_f] * 8 + [ctypes.c_int] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_rows_dot3": [ctypes.c_long, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_clip_adam": [ctypes.c_int] + [_c_f] * 6 + [ctypes.c_float] * 4 + [ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                                              ctypes.c_void_p],
})


class RowSegment(ctypes.Structure):
    """struct votenet_row_segment (include/votenet_hip.h)."""
    _fields_ = [("dst", ctypes.c_void_p), ("dst_pitch", ctypes.c_int), ("dst_off", ctypes.c_int), ("width", ctypes.c_int),
                ("a", ctypes.c_void_p), ("a_pitch", ctypes.c_int), ("a_off", ctypes.c_int),
                ("b", ctypes.c_void_p), ("b_pitch", ctypes.c_int), ("b_off", ctypes.c_int)]


class CopySegment(ctypes.Structure):
    """struct votenet_copy_segment (include/votenet_hip.h)."""
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p), ("bytes", ctypes.c_long)]


_SIGS["votenet_copy_segments"] = [ctypes.c_int, ctypes.POINTER(CopySegment), ctypes.c_void_p]
_SIGS["votenet_three_interpolate_concat"] = [ctypes.c_int] * 4 + [_c_f] * 4 + [ctypes.c_int, _c_f, ctypes.c_void_p]
_SIGS["votenet_three_interpolate_grad_strided"] = [ctypes.c_int] * 4 + [_c_f, ctypes.c_int, ctypes.c_int] + [_c_f] * 3 + [ctypes.c_void_p]
_SIGS["votenet_bias_grad_strided"] = [ctypes.c_long, ctypes.c_int, _c_f, ctypes.c_int, _c_f, _c_f, ctypes.c_void_p]
_SIGS["votenet_ema_update"] = [ctypes.c_long, ctypes.c_float, _c_f, _c_f, _c_f, ctypes.c_void_p]
_SIGS["votenet_row_segments"] = [ctypes.c_long, ctypes.c_int, ctypes.POINTER(RowSegment), ctypes.c_void_p]
# piece layout (csrc/half.hip)
_I, _L, _F = ctypes.c_int, ctypes.c_long, ctypes.c_float
_SIGS.update({
    "votenet_half_groups": [_I] + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_assemble_rows_half": [_I] * 3 + [_c_f] * 10 + [ctypes.c_void_p],
    "votenet_assembled_linear_half": [_L, _I, _I] + [_c_f] * 5 + [ctypes.POINTER(BnRaw), _I] + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_mlp_linear_half": [_c_f] * 3 + [_I, _L, _I, _I] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_half_centre_sums": [_L, _I] + [_c_f] * 7 + [_I, _c_f, ctypes.c_void_p],
    "votenet_mlp_linear_pool_half": [ctypes.POINTER(MlpInput), _L, _I, _I] + [_c_f] * 9 + [ctypes.c_void_p],
    "votenet_bn_pool_finalize_half": [_L, _I] + [_c_f] * 5 + [ctypes.POINTER(BnRaw), _I] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_pool_dgrad_scatter_half": [_L, _I, _I, _I] + [_c_f] * 4 + [_I] + [_c_f] * 9 + [_F, _I, _c_f, ctypes.POINTER(CoefTail), _c_f,
                                                                                       ctypes.c_void_p],
    "votenet_mlp_gram_half": [_L, _I, _c_f, _c_f, _I, _c_f, _c_f, _c_f, ctypes.c_void_p],
    "votenet_pool_wgrad_sparse_half": [_L, _I, _I, _I] + [_c_f] * 3 + [_I] + [_c_f] * 4 + [_I] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_pool_wgrad_sparse_half_centres": [_L, _I, _I, _I] + [_c_f] * 3 + [_I] + [_c_f] * 4 + [_I] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_assembled_wgrad_bn_half": [_L, _I, _I] + [_c_f] * 5 + [_I] + [_c_f] * 3 + [_I, _c_f, _c_f, _c_f, ctypes.c_void_p],
    "votenet_assembled_dgrad_bn_reduce_half": [_L, _I, _I] + [_c_f] * 3 + [_I] + [_c_f] * 9 + [_F, _I, ctypes.c_void_p,
                                                                                             ctypes.POINTER(CoefTail), _c_f, _c_f, ctypes.c_void_p],
    "votenet_half_piece_rows": [],
    "votenet_half_sort_rows": [_I, _I] + [_c_f] * 3 + [_I, _c_f, ctypes.c_void_p],
    "votenet_mlp_dgrad_bn_half": [_L, _I, _I] + [_c_f] * 3 + [_I] + [_c_f] * 4 + [ctypes.c_void_p],
    "votenet_group_linear_backward_masked": [_L, _I] + [_c_f] * 9 + [_F, _I] + [_c_f] * 4 + [ctypes.c_void_p, _c_f, ctypes.c_void_p],
    "votenet_assembled_point_grad": [_L, _I] + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_assembled_wx_finish": [_I] + [_c_f] * 3 + [_I] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_group_linear_backward_sorted": [_L, _I] + [_c_f] * 7 + [_I] + [_c_f] * 3 + [ctypes.c_void_p],
    "votenet_narrow_rows_half": [_I] * 4 + [_c_f] * 9 + [ctypes.c_void_p],
    "votenet_narrow_linear_masked": [_L, _I, _I, _I] + [_c_f] * 5 + [ctypes.POINTER(BnRaw), _I] + [_c_f] * 6 + [ctypes.c_void_p],
    "votenet_narrow_dgrad_bn_reduce_masked": [_L, _I, _I, _I] + [_c_f] * 3 + [_I] + [_c_f] * 8 + [_F, _I, _c_f, _c_f, ctypes.POINTER(CoefTail),
                                              _c_f, _c_f, ctypes.c_void_p],
    "votenet_narrow_linear_half": [_L, _I, _I, _I] + [_c_f] * 5 + [ctypes.POINTER(BnRaw), _I] + [_c_f] * 5 + [ctypes.c_void_p],
    "votenet_narrow_wgrad_bn_half": [_L, _I, _I, _I] + [_c_f] * 5 + [_I] + [_c_f] * 3 + [_I, _c_f, _c_f, ctypes.c_void_p],
    "votenet_narrow_dgrad_bn_reduce_half": [_L, _I, _I, _I] + [_c_f] * 3 + [_I] + [_c_f] * 8 + [_F, _I, _c_f, _c_f, ctypes.POINTER(CoefTail),
                                                                                            _c_f, ctypes.c_void_p],
})


class _Library(ctypes.CDLL):
    """The library's measurement / tuning switches (include/votenet_hip_debug.h) are inert until the host opts in.  This host opts in
    the first time something (a test, a profile tool, mlp.debug_switch) looks one up; code that never touches a switch never does."""

    def __getattr__(self, name):  # only reached for names not bound yet (CDLL caches what it has resolved)
        fn = super().__getattr__(name)
        if "debug" in name and name not in ("votenet_debug_enable", "votenet_debug_enabled", "votenet_debug_fps_split_timeouts"):
            super().__getattr__("votenet_debug_enable")(1)
        return fn


def lib():
    """Load the library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise VotenetError(
                "libvotenet_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % _LIB_PATH)
        L = _Library(_LIB_PATH)
        L.votenet_last_error.restype = ctypes.c_char_p
        L.votenet_version.restype = ctypes.c_char_p
        L.votenet_fps_temp_floats.restype = ctypes.c_size_t
        L.votenet_fps_temp_floats.argtypes = [ctypes.c_int, ctypes.c_int]
        L.votenet_mlp_wgrad_scratch_floats.restype = ctypes.c_size_t
        L.votenet_mlp_wgrad_scratch_floats.argtypes = [ctypes.POINTER(MlpInput), ctypes.c_long, ctypes.c_int, ctypes.c_int]
        L.votenet_pool_wgrad_scratch_floats.restype = ctypes.c_size_t
        L.votenet_pool_wgrad_scratch_floats.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int]
        L.votenet_group_linear_backward_scratch_floats.restype = ctypes.c_size_t
        L.votenet_group_linear_backward_scratch_floats.argtypes = [ctypes.c_int] * 3
        L.votenet_spatial_index_floats.restype = ctypes.c_size_t
        L.votenet_spatial_index_floats.argtypes = [ctypes.c_int, ctypes.c_int]
        L.votenet_loss_workspace_floats.restype = ctypes.c_size_t
        L.votenet_loss_workspace_floats.argtypes = [ctypes.c_int]
        L.votenet_knn_workspace_bytes.restype = ctypes.c_size_t
        L.votenet_knn_workspace_bytes.argtypes = [ctypes.c_int] * 3
        L.votenet_nms3d_workspace_bytes.restype = ctypes.c_size_t
        L.votenet_nms3d_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        L.votenet_mlp_split_k_floats.restype = ctypes.c_long
        L.votenet_mlp_split_k_floats.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int]
        L.votenet_ball_threshold.restype = ctypes.c_float
        L.votenet_ball_threshold.argtypes = [ctypes.c_float]
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = ctypes.c_int
            fn.argtypes = sig
        _lib = L
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().votenet_last_error().decode()
    if rc == 1:
        raise InvalidArgumentError(msg)
    raise VotenetError("libvotenet_hip error %d: %s" % (rc, msg))


# The host side of a train step is ~250 launches: the Python objects behind torch.cuda.current_stream() / torch.cuda.device(...)
# were a quarter of its enqueue time (cProfile, tools/probe/host_profile.py).  The raw-handle calls below are what those wrappers
# end up calling.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current HIP stream of the current device as a void* for the C ABI."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())  # (a plain int: ctypes converts it)
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def device_guard(device):
    """`with device_guard(t.device):` the launches inside run on t's device.  One process per GPU is the rule here, so the device
    almost always IS the current one: then this is a shared no-op object instead of a torch.cuda.device context."""
    idx = device.index
    if _cur_device is not None and (idx is None or idx == _cur_device()):
        return _NOGUARD
    return torch.cuda.device(device)


def ptr(t):
    """The device address of a tensor for a void* argument (a plain int: ctypes converts it; no c_void_p object per argument)."""
    return t.data_ptr() if t is not None else None


def dev_f32(t, name, rank=None, last=None):
    """Validate a float32 device tensor the way the TF wrappers validate shapes; return contiguous."""
    if not isinstance(t, torch.Tensor):
        raise InvalidArgumentError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise VotenetError("%s must live on the GPU (no CPU fallback in votenet_amd)" % name)
    if t.dtype != torch.float32:
        raise InvalidArgumentError("%s must be float32" % name)
    if rank is not None and t.dim() != rank:
        raise InvalidArgumentError("%s expects rank %d, got shape %s" % (name, rank, tuple(t.shape)))
    if last is not None and t.shape[-1] != last:
        raise InvalidArgumentError("%s expects last dimension %d, got shape %s" % (name, last, tuple(t.shape)))
    return t.contiguous()


def dev_i32(t, name, rank=None):
    if not isinstance(t, torch.Tensor):
        raise InvalidArgumentError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise VotenetError("%s must live on the GPU (no CPU fallback in votenet_amd)" % name)
    if t.dtype != torch.int32:
        raise InvalidArgumentError("%s must be int32" % name)
    if rank is not None and t.dim() != rank:
        raise InvalidArgumentError("%s expects rank %d, got shape %s" % (name, rank, tuple(t.shape)))
    return t.contiguous()
