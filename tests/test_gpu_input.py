"""GPU: votenet_subsample_augment / votenet_augment_boxes (the step before the hot path) against oracle/oracle_input.py.
Bar: bit-exact -- float64 arithmetic in the reference's order, one rounding to float32; integer labels exact."""
import numpy as np
import pytest
import torch

from oracle import oracle_input as OI

pytestmark = pytest.mark.gpu


def _scenes(b, seed, lo=2500, hi=6000, cols=6):
    rng = np.random.default_rng(seed)
    return [rng.normal(size=(int(rng.integers(lo, hi)), cols)) * np.array([2.0, 3.0, 1.0] + [1.0] * (cols - 3)) for _ in range(b)]


@pytest.mark.parametrize("b,n_out,f64", [(3, 2048, True), (19, 1024, False), (1, 2500, True)])
def test_points_with_host_choice_bit_exact(hiplib, dev, b, n_out, f64):
    from votenet_amd import input_pipeline as IP
    scenes = _scenes(b, 10 + b)
    if not f64:
        scenes = [s.astype(np.float32) for s in scenes]
    raw, off = IP.pack_ragged(scenes, dev)
    aug = IP.draw_augmentation(b, np.random.RandomState(b))
    ch = IP.draw_choice(np.random.RandomState(1), [len(s) for s in scenes], n_out)
    got = IP.subsample_augment(raw, off, n_out, aug, ch).cpu().numpy()
    for s in range(b):
        exp = OI.augment_points(scenes[s], ch[s], aug.flip_x[s], aug.flip_z[s], aug.angle[s], aug.scale[s])
        assert np.array_equal(got[s], exp), s
    ev = IP.subsample_augment(raw, off, n_out, None, ch, depth_to_camera=False).cpu().numpy()
    for s in range(b):
        assert np.array_equal(ev[s], np.asarray(scenes[s], np.float64)[ch[s], :3].astype(np.float32))


def test_points_with_device_draw_bit_exact_and_without_replacement(hiplib, dev):
    from votenet_amd import input_pipeline as IP
    b, n_out = 18, 2048  # two launches of <= 16 scenes: scene numbers keep counting
    scenes = _scenes(b, 3, cols=3)
    raw, off = IP.pack_ragged(scenes, dev)
    aug = IP.draw_augmentation(b, np.random.RandomState(5))
    got = IP.subsample_augment(raw, off, n_out, aug, None, seed=0x1234567890AB, scene0=40).cpu().numpy()
    for s in range(b):
        ch = OI.feistel_choice(len(scenes[s]), n_out, 0x1234567890AB, 40 + s)
        assert len(np.unique(ch)) == n_out
        exp = OI.augment_points(scenes[s], ch, aug.flip_x[s], aug.flip_z[s], aug.angle[s], aug.scale[s])
        assert np.array_equal(got[s], exp), s
    again = IP.subsample_augment(raw, off, n_out, aug, None, seed=0x1234567890AB, scene0=40).cpu().numpy()
    other = IP.subsample_augment(raw, off, n_out, aug, None, seed=0x1234567890AC, scene0=40).cpu().numpy()
    assert np.array_equal(got, again) and not np.array_equal(got, other)


def test_points_full_size_properties(hiplib, dev):
    """BASELINE size: 8 scenes x 50 000 raw points -> 20 480 (config.POINT_NUM): every output row is a distinct raw row
    (replace=False), isometry up to the scale, evaluation = pure row selection."""
    from votenet_amd import input_pipeline as IP
    b, n_raw, n_out = 8, 50000, 20480
    rng = np.random.default_rng(0)
    rawn = rng.normal(size=(b * n_raw, 3)).astype(np.float32)
    raw = torch.from_numpy(rawn).to(dev)
    off = np.arange(b + 1) * n_raw
    ev = IP.subsample_augment(raw, off, n_out, None, None, seed=1).cpu().numpy()
    for s in range(b):
        sc = rawn[s * n_raw:(s + 1) * n_raw]
        cam = np.stack([sc[:, 0], -sc[:, 2], sc[:, 1]], 1)
        keys = {r.tobytes() for r in cam}
        rows = [r.tobytes() for r in ev[s]]
        assert len(set(rows)) == n_out and all(r in keys for r in rows)
    aug = IP.draw_augmentation(b, np.random.RandomState(0))
    tr = IP.subsample_augment(raw, off, n_out, aug, None, seed=1).cpu().numpy().astype(np.float64)
    assert np.allclose(np.linalg.norm(tr, axis=2), np.linalg.norm(ev.astype(np.float64), axis=2) * aug.scale[:, None], rtol=1e-6, atol=1e-6)
    assert np.allclose(tr[..., 1], ev[..., 1] * aug.scale[:, None], rtol=1e-6)


def test_points_argument_errors(hiplib, dev):
    from votenet_amd import input_pipeline as IP, _lib
    raw = torch.zeros(100, 3, device=dev)
    off = np.array([0, 60, 100])
    with pytest.raises(_lib.InvalidArgumentError):  # 40 points, 50 wanted: numpy raises for replace=False too
        IP.subsample_augment(raw, off, 50)
    with pytest.raises(_lib.InvalidArgumentError):
        IP.subsample_augment(raw, off, 10, choice=np.full((2, 10), 70))
    with pytest.raises(_lib.InvalidArgumentError):
        IP.subsample_augment(raw, off, 10, aug=IP.draw_augmentation(3))
    with pytest.raises(_lib.InvalidArgumentError):
        IP.subsample_augment(torch.zeros(100, 2, device=dev), off, 10)
    assert IP.subsample_augment(raw, off, 40).shape == (2, 40, 3)


@pytest.mark.parametrize("b,train", [(4, True), (17, True), (3, False)])
def test_boxes_bit_exact(hiplib, dev, b, train):
    from votenet_amd import input_pipeline as IP, synth
    rng = np.random.default_rng(b)
    cnt = rng.integers(1, 12, b)
    cen = [rng.normal(size=(c, 3)) * 2 for c in cnt]
    siz = [np.abs(rng.normal(size=(c, 3))) + 0.3 for c in cnt]
    hed = [rng.uniform(-2 * np.pi, 2 * np.pi, c) for c in cnt]
    hed[0][0] = 0.0
    cls = [rng.integers(0, 10, c).astype(np.int32) for c in cnt]
    aug = IP.draw_augmentation(b, np.random.RandomState(b)) if train else None
    dc, off = IP.pack_ragged(cen, dev)
    ds, _ = IP.pack_ragged(siz, dev)
    dh, _ = IP.pack_ragged(hed, dev)
    dk, _ = IP.pack_ragged(cls, dev)
    got = IP.augment_boxes(dc, ds, dh, dk, off, aug)
    per = [OI.augment_boxes(cen[s], siz[s], hed[s], cls[s], train and aug.flip_x[s], train and aug.flip_z[s],
                            aug.angle[s] if train else 0.0, aug.scale[s] if train else 1.0, synth.MEAN_SIZES, synth.NH, train=train)
           for s in range(b)]
    exp = OI.batch_boxes(per)
    assert set(got) == set(exp)
    for k in exp:
        g = got[k].cpu().numpy()
        assert g.dtype == exp[k].dtype and g.shape == exp[k].shape, k
        assert np.array_equal(g, exp[k]), k
    # the result feeds the loss graph directly
    assert got["bboxes_xyz"].shape[1] == cnt.max()


def test_boxes_argument_errors(hiplib, dev):
    from votenet_amd import input_pipeline as IP, _lib
    z3 = torch.zeros(4, 3, dtype=torch.float64, device=dev)
    z1 = torch.zeros(4, dtype=torch.float64, device=dev)
    k = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.InvalidArgumentError):  # a scene without boxes (dataset.py:300 skips it)
        IP.augment_boxes(z3, z3, z1, k, np.array([0, 2, 2, 4]))
    with pytest.raises(_lib.InvalidArgumentError):
        IP.augment_boxes(z3, z3, z1, k, np.array([0, 2, 6]))
    out = IP.augment_boxes(z3, z3 + 1, z1, k, np.array([0, 1, 4]))
    assert out["bboxes_lwh"].shape == (2, 3, 3) and torch.equal(out["bboxes_lwh"][0, 2], out["bboxes_lwh"][0, 0])


def test_pipeline_output_feeds_the_train_step(hiplib, dev):
    """raw clouds + ragged boxes -> subsample / augment / encode / pad on the device -> one train step on the result: the
    eight ground-truth tensors have the layout and dtypes the loss graph takes, the points the layout the backbone takes."""
    from votenet_amd import input_pipeline as IP, synth
    from votenet_amd.model import VoteNetHotPath
    b, n_raw, n_out = 2, 30000, 20480
    raws, cen, siz, hed, cls = [], [], [], [], []
    for s in range(b):
        pts, boxes = synth.room_scene(n_raw, 4000 + s)
        cam = pts.astype(np.float64)
        raws.append(np.stack([cam[:, 0], cam[:, 2], -cam[:, 1]], 1))  # camera -> upright depth axes (sunutils.py:79-84)
        bx = boxes.astype(np.float64)
        cen.append(bx[:, :3]), siz.append(bx[:, 3:6]), hed.append(bx[:, 6]), cls.append(bx[:, 7].astype(np.int32))
    raw, off = IP.pack_ragged(raws, dev)
    aug = IP.draw_augmentation(b, np.random.RandomState(3))
    x = IP.subsample_augment(raw, off, n_out, aug, None, seed=9)
    dc, boff = IP.pack_ragged(cen, dev)
    gt = IP.augment_boxes(dc, IP.pack_ragged(siz, dev)[0], IP.pack_ragged(hed, dev)[0], IP.pack_ragged(cls, dev)[0], boff, aug)
    assert x.shape == (b, n_out, 3) and gt["bboxes_xyz"].shape[0] == b
    # augmented points still lie on their augmented boxes: every box centre has points within half its diagonal
    d = (x[:, :, None, :] - gt["bboxes_xyz"][:, None, :, :]).norm(dim=-1)
    assert bool((d.min(1).values < gt["bboxes_lwh"].norm(dim=-1) * 0.75).all())
    net = VoteNetHotPath(dev, seed=1)
    net.train_step(x, gt=gt)
    assert bool(torch.isfinite(net.last_losses[:10]).all()) and float(net.last_losses[10]) > 0  # positives were assigned
    assert bool(torch.isfinite(net.store.flat).all())
