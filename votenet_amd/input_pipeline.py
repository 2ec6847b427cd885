"""Host mirror of the reference's input pipeline for the step before the hot path.

The reference builds every training sample in numpy inside MyDataFlow.__iter__ (dataset.py:183-189 random subsample to
config.POINT_NUM points + depth->camera axes, :219-231 the augmentation draws, :262-276 the box side, :302-308 the point
side) and pads the ragged ground truth in BatchData2Biggest (run.py:14-24,60-64).  Here the DRAWS stay on the host, in the
reference's order, and the work is two kernels over the whole batch (votenet_subsample_augment, votenet_augment_boxes).
There is no CPU path: without libvotenet_hip.so these functions raise.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L
from .synth import MEAN_SIZES, NH

POINT_NUM = 20480  # config.py:1


class Augmentation:
    """Per-scene draws of dataset.py:219-231: flip_x, flip_z (bools), angle (rad), scale; float64 like numpy."""

    def __init__(self, flip_x, flip_z, angle, scale):
        self.flip_x = np.ascontiguousarray(flip_x, dtype=bool)
        self.flip_z = np.ascontiguousarray(flip_z, dtype=bool)
        self.angle = np.ascontiguousarray(angle, dtype=np.float64)
        self.scale = np.ascontiguousarray(scale, dtype=np.float64)
        self.b = len(self.angle)

    def host_arrays(self):
        flip = (self.flip_x.astype(np.int32) | (self.flip_z.astype(np.int32) << 1)).astype(np.int32)
        return flip, self.angle, np.cos(self.angle), np.sin(self.angle), self.scale  # np.cos / np.sin: sunutils.py:135-136


def draw_augmentation(b, rand=np.random):
    """The four np.random.rand() draws per scene in the reference's order (dataset.py:219-231)."""
    fx, fz, ang, sc = [], [], [], []
    for _ in range(b):
        fx.append(rand.rand() > 0.5)
        fz.append(rand.rand() > 0.5)
        ang.append((rand.rand() * 2 - 1.) * 5. / 180 * np.pi)
        sc.append((rand.rand() * 2 - 1.) * 0.1 + 1.)
    return Augmentation(fx, fz, ang, sc)


def draw_choice(rng, n_raw, n_out=POINT_NUM):
    """dataset.py:185-186: self.rng.choice(n, POINT_NUM, replace=False) per scene -> (b, n_out) int32."""
    return np.stack([rng.choice(int(n), n_out, replace=False) for n in n_raw]).astype(np.int32)


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def pack_ragged(arrays, device, dtype=None):
    """list of per-scene (n_s, ...) arrays -> (device tensor of the concatenation, host int64 offsets (b+1))."""
    off = np.zeros(len(arrays) + 1, np.int64)
    off[1:] = np.cumsum([len(a) for a in arrays])
    cat = np.ascontiguousarray(np.concatenate(arrays, 0))
    if dtype is not None:
        cat = cat.astype(dtype, copy=False)
    return torch.from_numpy(cat).to(device), off


def subsample_augment(raw, raw_offset, n_out=POINT_NUM, aug=None, choice=None, seed=0, scene0=0, depth_to_camera=True):
    """raw: device tensor (sum n_s, stride) float32 or float64 in upright-depth coordinates (or already camera frame with
    depth_to_camera=False); raw_offset: host int64 (b+1).  choice: (b, n_out) int32 (host or device) -- the caller's
    rng.choice -- or None: the device draws a keyed permutation (seed, scene0 + s).  aug: Augmentation or None (evaluation).
    -> (b, n_out, 3) float32 device tensor: the `points` input of model.py:22."""
    if raw.dim() != 2 or raw.dtype not in (torch.float32, torch.float64) or not raw.is_cuda:
        raise L.InvalidArgumentError("subsample_augment: raw must be a 2-D float32 / float64 device tensor")
    raw = raw.contiguous()
    off = np.ascontiguousarray(raw_offset, dtype=np.int64)
    b = len(off) - 1
    if off[0] < 0 or off[-1] > raw.shape[0] or np.any(np.diff(off) < 0):
        raise L.InvalidArgumentError("subsample_augment: raw_offset does not describe rows of raw")
    if aug is not None and aug.b != b:
        raise L.InvalidArgumentError("subsample_augment: %d scenes but %d augmentation draws" % (b, aug.b))
    ch = None
    if choice is not None:
        if tuple(choice.shape) != (b, n_out):
            raise L.InvalidArgumentError("subsample_augment: choice must be (b, n_out)")
        if torch.is_tensor(choice) and choice.is_cuda:  # already on the device: not read back (the kernel clamps the index)
            ch = choice.to(torch.int32).contiguous()
        else:
            chn = np.asarray(choice)
            if np.any(chn < 0) or np.any(chn >= np.diff(off)[:, None]):
                raise L.InvalidArgumentError("subsample_augment: choice index out of range")
            ch = torch.from_numpy(np.ascontiguousarray(chn, dtype=np.int32)).to(raw.device)
    out = torch.empty((b, n_out, 3), dtype=torch.float32, device=raw.device)
    flip = ang = c = s = sc = None
    if aug is not None:
        flip, ang, c, s, sc = aug.host_arrays()
    with L.device_guard(raw.device):
        L.check(L.lib().votenet_subsample_augment(b, n_out, L.ptr(raw), 1 if raw.dtype == torch.float64 else 0, raw.shape[1], _hp(off),
                                                  L.ptr(ch), int(seed) & (2 ** 64 - 1), int(scene0), 1 if depth_to_camera else 0,
                                                  _hp(flip), _hp(c), _hp(s), _hp(sc), L.ptr(out), L.stream_ptr()))
    return out


GT_FIELDS = (("bboxes_xyz", 3, torch.float32), ("bboxes_lwh", 3, torch.float32), ("bboxes_roty", 0, torch.float32),
             ("semantic_labels", 0, torch.int32), ("heading_labels", 0, torch.int32), ("heading_residuals", 0, torch.float32),
             ("size_labels", 0, torch.int32), ("size_residuals", 3, torch.float32))


def augment_boxes(center, size, heading, cls, box_offset, aug=None, mean_size=MEAN_SIZES, nh=NH):
    """center (nbox,3), size (nbox,3), heading (nbox) float64 and cls (nbox) int32 device tensors in the upright-camera
    frame (dataset.py:253-259), box_offset host int64 (b+1).  -> dict of the eight ground-truth inputs of model.py:23-32 on
    the device, every scene padded to the longest by repeating its last box (run.py:14-24)."""
    off = np.ascontiguousarray(box_offset, dtype=np.int64)
    b = len(off) - 1
    cnt = np.diff(off)
    if b < 1 or np.any(cnt < 1):
        raise L.InvalidArgumentError("augment_boxes: every scene needs at least one box (dataset.py:300 skips the others)")
    if aug is not None and aug.b != b:
        raise L.InvalidArgumentError("augment_boxes: %d scenes but %d augmentation draws" % (b, aug.b))
    dev = center.device
    center, size, heading = (t.to(torch.float64).contiguous() for t in (center, size, heading))
    cls = cls.to(torch.int32).contiguous()
    if off[-1] > center.shape[0] or center.shape != size.shape or heading.shape[0] != center.shape[0] or cls.shape[0] != center.shape[0]:
        raise L.InvalidArgumentError("augment_boxes: box arrays do not match box_offset")
    bb = int(cnt.max())
    ms = np.ascontiguousarray(mean_size, dtype=np.float64)
    out = {k: torch.empty((b, bb, w) if w else (b, bb), dtype=dt, device=dev) for k, w, dt in GT_FIELDS}
    flip = ang = c = s = sc = None
    if aug is not None:
        flip, ang, c, s, sc = aug.host_arrays()
    with L.device_guard(dev):
        L.check(L.lib().votenet_augment_boxes(b, bb, _hp(off), L.ptr(center), L.ptr(size), L.ptr(heading), L.ptr(cls), _hp(flip),
                                              _hp(ang), _hp(c), _hp(s), _hp(sc), _hp(ms), ms.shape[0], nh,
                                              *[L.ptr(out[k]) for k, _, _ in GT_FIELDS], L.stream_ptr()))
    return out
