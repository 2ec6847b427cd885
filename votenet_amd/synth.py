"""Synthetic SUN RGB-D-like scenes (the dataset is not available).

"room" generator: 60 % of the points on the floor and two walls of a 5 x 5 x 3 m room (y up,
camera at the origin looking +z: the upright-camera frame of dataset.py:188), 40 % on the
surfaces of 6-10 random upright boxes with SUN RGB-D class mean sizes (dataset.py:36-45),
plus N(0, 5 mm) noise.  Deterministic per (seed, scene).
"""
import numpy as np

# type_mean_size of the ten SUN RGB-D classes (l, w, h), dataset.py:36-45
MEAN_SIZES = np.array([[2.114256, 1.620300, 0.927272], [0.791118, 1.279516, 0.718182], [0.923508, 1.867419, 0.845495],
                       [0.591958, 0.552978, 0.827272], [0.699104, 0.454178, 0.75625], [0.69519, 1.346299, 0.736364],
                       [0.528526, 1.002642, 1.172878], [0.500618, 0.632163, 0.683424], [0.404671, 1.071108, 1.688889],
                       [0.76584, 1.398258, 0.472728]], dtype=np.float64)


def room_scene(n, seed, size=(5.0, 3.0, 5.0), nbox=(6, 10)):
    rng = np.random.default_rng(seed)
    sx, sy, sz = size
    n_struct = int(n * 0.6)
    n_obj = n - n_struct
    pts = np.empty((n, 3), np.float64)
    # floor (y = -sy/2 ... camera height), back wall (z = sz), left wall (x = -sx/2)
    which = rng.integers(0, 3, n_struct)
    u, v = rng.random(n_struct), rng.random(n_struct)
    floor = np.stack([(u - 0.5) * sx, np.full(n_struct, -1.2), v * sz], 1)
    back = np.stack([(u - 0.5) * sx, v * sy - 1.2, np.full(n_struct, sz)], 1)
    left = np.stack([np.full(n_struct, -sx / 2), u * sy - 1.2, v * sz], 1)
    pts[:n_struct] = np.where(which[:, None] == 0, floor, np.where(which[:, None] == 1, back, left))
    k = int(rng.integers(nbox[0], nbox[1] + 1))
    per = np.full(k, n_obj // k)
    per[: n_obj - per.sum()] += 1
    o = n_struct
    boxes = []
    for i in range(k):
        cls = int(rng.integers(0, 10))  # SUN RGB-D class of the box (dataset.py:31-32); same draw as before: clouds unchanged
        l, w, h = MEAN_SIZES[cls] * rng.uniform(0.8, 1.2, 3)
        ang = rng.uniform(0, 2 * np.pi)
        cx, cz = rng.uniform(-sx / 2 + 0.8, sx / 2 - 0.8), rng.uniform(0.8, sz - 0.8)
        cy = -1.2 + h / 2
        q = rng.uniform(-0.5, 0.5, (per[i], 3)) * np.array([l, h, w])
        face = rng.integers(0, 3, per[i])
        sign = rng.choice([-0.5, 0.5], per[i])
        dims = np.array([l, h, w])
        for a in range(3):
            sel = face == a
            q[sel, a] = sign[sel] * dims[a]
        c, s = np.cos(ang), np.sin(ang)
        x = c * q[:, 0] + s * q[:, 2]
        z = -s * q[:, 0] + c * q[:, 2]
        pts[o:o + per[i]] = np.stack([x + cx, q[:, 1] + cy, z + cz], 1)
        o += per[i]
        boxes.append((cx, cy, cz, l, w, h, ang, cls))
    pts += rng.normal(0, 0.005, pts.shape)
    rng.shuffle(pts)  # the reference subsamples at random: no spatial order in the index
    return pts.astype(np.float32), np.array(boxes, np.float32)


def room_batch(b, n, seed0=1000, **kw):
    return np.stack([room_scene(n, seed0 + i, **kw)[0] for i in range(b)])


def uniform_batch(b, n, seed0=1000, extent=5.0):
    """Worst case for the ball query: uniform in a cube, no ball fills K -> full n-scan."""
    return np.stack([np.random.default_rng(seed0 + i).random((n, 3), dtype=np.float32) * extent for i in range(b)])


NH, NS, NC = 12, 10, 10  # config.py: heading bins, size classes, semantic classes


def angle2class(angle, num_class=NH):
    """dataset.py:52-67: heading angle -> (bin, residual from the bin centre)."""
    angle = angle % (2 * np.pi)
    per = 2 * np.pi / float(num_class)
    shifted = (angle + per / 2) % (2 * np.pi)
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def room_gt(b, n, seed0=1000, **kw):
    """Ground truth of room_batch(b, n, seed0) in the reference's input layout (model.py:22-32, dataset.py:279-299): the
    generating boxes, ragged lists padded to the longest by repeating the last box (run.py:14-24, np.pad mode='edge').
    -> dict of float32 / int32 arrays: bboxes_xyz (b,BB,3), bboxes_lwh (b,BB,3), bboxes_roty (b,BB), semantic_labels,
    heading_labels (b,BB), heading_residuals (b,BB), size_labels (b,BB), size_residuals (b,BB,3)."""
    per_scene = []
    for i in range(b):
        boxes = room_scene(n, seed0 + i, **kw)[1].astype(np.float64)
        rows = []
        for cx, cy, cz, l, w, h, ang, cls in boxes:
            cls = int(cls)
            hc, hr = angle2class(ang)
            size = np.array([l, w, h])
            rows.append((np.array([cx, cy, cz]), size, ang, cls, hc, hr / (np.pi / NH), cls, (size - MEAN_SIZES[cls]) / MEAN_SIZES[cls]))
        per_scene.append(rows)
    bb = max(len(r) for r in per_scene)
    for r in per_scene:
        r += [r[-1]] * (bb - len(r))
    col = lambda k, dt: np.array([[row[k] for row in r] for r in per_scene], dtype=dt)
    return dict(bboxes_xyz=col(0, np.float32), bboxes_lwh=col(1, np.float32), bboxes_roty=col(2, np.float32),
                semantic_labels=col(3, np.int32), heading_labels=col(4, np.int32), heading_residuals=col(5, np.float32),
                size_labels=col(6, np.int32), size_residuals=col(7, np.float32))
