"""CPU restatement of the reference's per-scene input code -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Follows MyDataFlow.__iter__ (dataset.py:183-189, 219-231, 262-308), angle2class / size2class (dataset.py:52-84), roty and
flip_axis_to_camera (sunutils.py:70-77,133-139) and the batch padding of run.py:14-24 in numpy float64, as the reference
computes them.  dataset.py imports mayavi / cv2 / tensorpack at module level, none of which is in the image, so the
reference functions cannot be imported to generate fixtures: **parity unpinned** against the reference itself; pinned
structurally (line citations) and by tests/test_oracle_input.py, which checks the element-wise form used here against the
literal `(roty(a) @ p.T).T` matrix form and the label encoding against its inverse (class2angle, dataset.py:70-78).
"""
import numpy as np

TWO_PI = 2 * np.pi


def roty(t):
    """sunutils.py:133-139."""
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def flip_axis_to_camera(pc):
    """sunutils.py:70-77: depth (x right, y forward, z up) -> camera (x right, y down, z forward)."""
    pc2 = np.copy(pc)
    pc2[:, [0, 1, 2]] = pc2[:, [0, 2, 1]]
    pc2[:, 1] *= -1
    return pc2


def angle2class(angle, num_class):
    """dataset.py:52-67."""
    angle = angle % TWO_PI
    per = TWO_PI / float(num_class)
    shifted = (angle + per / 2) % TWO_PI
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def augment_points(raw, choice, flip_x, flip_z, angle, scale, train=True, depth_to_camera=True, literal=False):
    """One scene.  raw (n, >=3) float64/float32, choice (n_out,) indices.  -> (n_out, 3) float32.
    literal=True evaluates the rotation as the reference writes it (a BLAS matrix product, whose summation order and FMA
    use are the library's); the default is the same sum written out term by term, which is what the kernel is held to."""
    pc = np.asarray(raw, dtype=np.float64)[choice][:, :3]                      # dataset.py:185-186
    if depth_to_camera:
        pc = flip_axis_to_camera(pc)                                             # :187-189
    else:
        pc = pc.copy()
    if train:
        if flip_x:
            pc[..., 0] = -pc[..., 0]                                             # :303-304
        if flip_z:
            pc[..., 2] = -pc[..., 2]                                             # :305-306
        if literal:
            pc = (roty(angle) @ pc.T).T                                          # :307
        else:
            c, s = np.cos(angle), np.sin(angle)
            pc = np.stack([c * pc[:, 0] + s * pc[:, 2], pc[:, 1], -s * pc[:, 0] + c * pc[:, 2]], 1)
        pc = pc * scale                                                          # :308
    return pc.astype(np.float32)


def augment_boxes(centers, sizes, headings, classes, flip_x, flip_z, angle, scale, mean_size, nh, train=True):
    """One scene's boxes (lists / arrays, float64).  -> tuple of the eight per-box arrays of dataset.py:310-311 (float64 /
    int), before padding."""
    xyz, lwh, rot, sem, hl, hr, sl, sr = [], [], [], [], [], [], [], []
    for c0, s0, h, k in zip(centers, sizes, headings, classes):
        c0, s0, h, k = np.array(c0, np.float64), np.array(s0, np.float64), float(h), int(k)
        if train:
            if flip_x:
                c0[0] = -c0[0]
                h = np.pi - h                                                    # :263-265
            if flip_z:
                c0[2] = -c0[2]
                h = -h                                                           # :266-268
            cs, sn = np.cos(angle), np.sin(angle)
            c0 = np.array([cs * c0[0] + sn * c0[2], c0[1], -sn * c0[0] + cs * c0[2]])   # :270 roty(a) @ centre
            h += angle                                                           # :271
            c0 = c0 * scale                                                      # :273
            s0 = s0 * scale                                                      # :274
        res = s0 - mean_size[k]                                                  # size2class, :80-84
        cid, ares = angle2class(h, nh)                                           # :279
        xyz.append(c0), lwh.append(s0), rot.append(h), sem.append(k), hl.append(cid)
        hr.append(ares / (np.pi / nh))                                           # :296
        sl.append(k), sr.append(res / mean_size[k])                              # :297-298
    return (np.array(xyz), np.array(lwh), np.asarray(rot), np.array(sem), np.array(hl), np.array(hr), np.array(sl), np.array(sr))


def pad_along_axis(array, target_length):
    """run.py:14-24: np.pad mode='edge' along axis 0."""
    pad = target_length - array.shape[0]
    if pad < 0:
        return array
    return np.pad(array, [(0, pad)] + [(0, 0)] * (array.ndim - 1), mode="edge")


def batch_boxes(per_scene):
    """run.py:60-64: pad every component to the longest scene, stack; cast like the model's placeholders (model.py:23-32)."""
    bb = max(len(s[0]) for s in per_scene)
    names = ("bboxes_xyz", "bboxes_lwh", "bboxes_roty", "semantic_labels", "heading_labels", "heading_residuals", "size_labels",
             "size_residuals")
    ints = {"semantic_labels", "heading_labels", "size_labels"}
    return {n: np.stack([pad_along_axis(s[i], bb) for s in per_scene]).astype(np.int32 if n in ints else np.float32)
            for i, n in enumerate(names)}


# ---- the device-side draw (NOT from the reference: the keyed permutation of votenet_subsample_augment, restated) ----
def _lowbias32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def scene_key(seed, scene):
    with np.errstate(over="ignore"):
        sp1 = np.uint32((scene + 1) & 0xFFFFFFFF)
        hi = _lowbias32(np.uint32((seed >> 32) & 0xFFFFFFFF) + np.uint32(0x632BE5AB) * sp1)
        return int(_lowbias32(np.uint32(seed & 0xFFFFFFFF) ^ hi ^ (np.uint32(0x85EBCA6B) * sp1)))


def feistel_choice(n, n_out, seed, scene):
    """rows perm(0..n_out-1) of the keyed permutation of [0, n)."""
    key = np.uint32(scene_key(seed, scene))
    bits = 2
    while (1 << bits) < n:
        bits += 2
    half = bits // 2
    mask = np.uint32((1 << half) - 1)
    x = np.arange(n_out, dtype=np.uint64)
    todo = np.ones(n_out, bool)
    with np.errstate(over="ignore"):
        while todo.any():
            v = x[todo]
            l = ((v >> np.uint64(half)).astype(np.uint32)) & mask
            r = v.astype(np.uint32) & mask
            for rnd in range(6):
                f = _lowbias32(r + key + np.uint32(0x9E3779B9) * np.uint32(rnd + 1)) & mask
                l, r = r, l ^ f
            v = (l.astype(np.uint64) << np.uint64(half)) | r.astype(np.uint64)
            x[todo] = v
            todo[todo] = v >= n
    return x.astype(np.int64)
