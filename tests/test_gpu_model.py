"""GPU: the layer stack (votenet_amd.model / pointnet2) against the CPU oracle composed the same way."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def N(t):
    return t.detach().cpu().numpy()


def oracle_chain(O, x, layers, k=None):
    for L in layers:
        z = O.linear(x, N(L.p("W")), N(L.p("b")))
        if L.bn:
            mean, var = O.bn_stats(z)
            x = O.bn_relu(z, mean, var, N(L.p("gamma")), N(L.p("beta")), relu=L.relu)
        else:
            x = z
    return O.max_over_k(x, k) if k else x


def oracle_sa(O, mod, xyz, pts, sample_xyz=None):
    b = xyz.shape[0]
    fidx = O.farthest_point_sample(mod.npoint, sample_xyz if sample_xyz is not None else xyz)
    new_xyz = O.gather_point(xyz, fidx)
    idx, _ = O.query_ball_point(mod.radius, mod.nsample, xyz, new_xyz)
    g = O.group_concat(xyz, new_xyz, pts, idx).reshape(-1, 3 + pts.shape[2])
    out = oracle_chain(O, g, mod.mlp, mod.nsample)
    if mod.mlp2:
        out = oracle_chain(O, out, mod.mlp2)
    return new_xyz, out.reshape(b, mod.npoint, -1)


def oracle_fp(O, mod, x1, x2, p1, p2):
    dist, idx = O.three_nn(x1, x2)
    itp = O.three_interpolate(p2, idx, O.three_nn_weights(dist))
    x = np.concatenate([itp, p1], 2).reshape(-1, itp.shape[2] + p1.shape[2])
    return oracle_chain(O, x, mod.mlp).reshape(x1.shape[0], x1.shape[1], -1)


def test_forward_small_vs_oracle(hiplib, dev, O):
    """The full layer stack on a reduced cloud (2 x 4096 pts, 512/256/128/64 samples), layer by layer."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = synth.room_batch(2, 4096, 77)
    net = VM.VoteNetHotPath(dev, seed=3, npoints=(512, 256, 128, 64))
    # perturb BN affine parameters so they are exercised
    g = torch.Generator().manual_seed(1)
    for name, v in net.store.views.items():
        if name.endswith("gamma"):
            v.copy_((1 + 0.2 * torch.randn(v.shape, generator=g)).to(dev))
        if name.endswith("beta") or name.endswith("/b"):
            v.copy_((0.1 * torch.randn(v.shape, generator=g)).to(dev))
    # the proposal layer samples 256 of the seeds; with 256 seeds here use all of them
    out = net.forward(torch.from_numpy(x).to(dev))

    l1x, l1p = oracle_sa(O, net.sa1, x, x)
    l2x, l2p = oracle_sa(O, net.sa2, l1x, l1p)
    l3x, l3p = oracle_sa(O, net.sa3, l2x, l2p)
    l4x, l4p = oracle_sa(O, net.sa4, l3x, l3p)
    l3p2 = oracle_fp(O, net.fp1, l3x, l4x, l3p, l4p)
    seeds = oracle_fp(O, net.fp2, l2x, l3x, l2p, l3p2)
    assert (N(out["seeds_xyz"]) == l2x).all()

    def relerr(a, b):
        return np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert relerr(N(out["seeds_points"]), seeds) < 2e-5
    xx = np.concatenate([l2x, seeds], 2).reshape(-1, 259)
    votes = (xx + oracle_chain(O, xx, net.voting)).reshape(2, -1, 259)
    assert relerr(N(out["votes_xyz"]), votes[..., :3]) < 2e-5
    assert relerr(N(out["votes_points"]), votes[..., 3:]) < 2e-5
    # proposal layer on the DEVICE votes (the neighbour lists depend on vote xyz to the last bit)
    vx, vp = N(out["votes_xyz"]), N(out["votes_points"])
    px, pout = oracle_sa(O, net.proposal, vx, vp, sample_xyz=l2x)
    assert (N(out["proposals_xyz"]) == px).all()  # utils.py:42-43: FPS on seeds, centres from votes
    assert relerr(N(out["proposals_output"]), pout) < 2e-5
    assert out["proposals_output"].shape == (2, 256, 79)


_FULL = {}


def _perturbed_net(dev, seed):
    from votenet_amd import model as VM
    net = VM.VoteNetHotPath(dev, seed=seed)
    g = torch.Generator().manual_seed(1)
    for name, v in net.store.views.items():
        if name.endswith("gamma"):
            v.copy_((1 + 0.2 * torch.randn(v.shape, generator=g)).to(dev))
        if name.endswith("beta") or name.endswith("/b"):
            v.copy_((0.1 * torch.randn(v.shape, generator=g)).to(dev))
    return net


def _device_layers(net, x, dev):
    """The stack module by module (what VoteNetHotPath.forward runs, with every intermediate kept)."""
    from votenet_amd import mlp as M
    net.store.refresh_split()
    M.arena_begin(dev)
    try:
        l1x, l1p, _ = net.sa1.forward(x, x)
        l2x, l2p, _ = net.sa2.forward(l1x, l1p)
        l3x, l3p, _ = net.sa3.forward(l2x, l2p)
        l4x, l4p, _ = net.sa4.forward(l3x, l3p)
        l3p2 = net.fp1.forward(l3x, l4x, l3p, l4p)
        seeds = net.fp2.forward(l2x, l3x, l2p, l3p2)
        vx, vp = net.vote(l2x, seeds)
        px, pout = net.propose(vx, vp, l2x)
    finally:
        M.arena_end()
    torch.cuda.synchronize()
    return dict(l1x=l1x, l1p=l1p, l2x=l2x, l2p=l2p, l3x=l3x, l3p=l3p, l4x=l4x, l4p=l4p, l3p2=l3p2, seeds=seeds, vx=vx, vp=vp, px=px, pout=pout)


def test_forward_full_size_vs_oracle_layer_by_layer(hiplib, dev, O, gemm_form):
    """The layer stack at the REAL sizes of model.py:39-57,89-93 (2 x 20480 points -> 2048 -> 1024 -> 512 -> 256, K = 64, fp1, fp2,
    voting, proposal) against the CPU oracle, layer by layer, on both GEMM forms: sampled indices / centres bit-exact, features to
    2e-5 of the tensor maximum.  The oracle runs on all host cores (liboracle_omp.so: bit-identical to the single-thread build,
    tests/test_oracle_omp.py); its result is computed once for both forms."""
    import os
    from votenet_amd import synth
    x = synth.room_batch(2, 20480, 4242)
    net = _perturbed_net(dev, 3)
    got = {k: N(v) for k, v in _device_layers(net, torch.from_numpy(x).to(dev), dev).items()}
    if "ref" not in _FULL:
        prev = O.set_threads(max(1, min(128, len(os.sched_getaffinity(0)))))
        try:
            r = {}
            r["l1x"], r["l1p"] = oracle_sa(O, net.sa1, x, x)
            r["l2x"], r["l2p"] = oracle_sa(O, net.sa2, r["l1x"], r["l1p"])
            r["l3x"], r["l3p"] = oracle_sa(O, net.sa3, r["l2x"], r["l2p"])
            r["l4x"], r["l4p"] = oracle_sa(O, net.sa4, r["l3x"], r["l3p"])
            r["l3p2"] = oracle_fp(O, net.fp1, r["l3x"], r["l4x"], r["l3p"], r["l4p"])
            r["seeds"] = oracle_fp(O, net.fp2, r["l2x"], r["l3x"], r["l2p"], r["l3p2"])
            xx = np.concatenate([r["l2x"], r["seeds"]], 2).reshape(-1, 259)
            votes = (xx + oracle_chain(O, xx, net.voting)).reshape(2, -1, 259)
            r["vx"], r["vp"] = votes[..., :3], votes[..., 3:]
        finally:
            O.set_threads(prev)
        _FULL["ref"] = r
    r = _FULL["ref"]

    def relerr(a, b):
        return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    for k in ("l1x", "l2x", "l3x", "l4x"):
        assert (got[k] == r[k]).all(), k  # FPS picks (incl. the prefix shortcut of the lower levels) and gathered centres: exact
    errs = {k: relerr(got[k], r[k]) for k in ("l1p", "l2p", "l3p", "l4p", "l3p2", "seeds", "vx", "vp")}
    _record_parity("chained_stack", gemm_form, errs)
    assert max(errs.values()) < 2e-5, errs
    # proposal layer on the DEVICE votes (its neighbour lists depend on the vote coordinates to the last bit)
    prev = O.set_threads(max(1, min(128, len(os.sched_getaffinity(0)))))
    try:
        px, pout = oracle_sa(O, net.proposal, got["vx"], got["vp"], sample_xyz=r["l2x"])
    finally:
        O.set_threads(prev)
    assert (got["px"] == px).all()
    assert relerr(got["pout"], pout) < 2e-5 and got["pout"].shape == (2, 256, 79)


def _record_parity(name, form, errs):
    """Achieved per-module errors into gpurun_out/parity_errors.txt (copied to profiles/ per round)."""
    import os
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_errors.txt"), "a") as f:
            f.write("%s  %s  %s\n" % (name, {2: "fp16x2_forward_images", 1: "bf16x3_images", 0: "fp32_mfma"}[form],
                                      "  ".join("%s=%.2e" % (k, v) for k, v in errs.items())))
    except OSError:
        pass


def test_forward_full_size_each_module_on_the_oracles_input(hiplib, dev, O, gemm_form):
    """north_star's bar -- grouped features within 1e-5 (fp32) of the reference on IDENTICAL inputs -- module by module at the real
    sizes: every device module (sa1-4, fp1-2, voting, proposal: utils.py:125-132,286-293, model.py:53-57,89-93) is fed the ORACLE's
    input of that module, so an error is that module's own (three GEMM + BatchNorm layers and the pool), not what the chain
    accumulated up to it.  Sampled centres exact; features to 1e-5 of the tensor maximum.  (The chained stack is held to its own bar
    by test_forward_full_size_vs_oracle_layer_by_layer.)  The achieved maxima are recorded."""
    import os
    from votenet_amd import mlp as M
    from votenet_amd import synth
    x = synth.room_batch(2, 20480, 4242)
    net = _perturbed_net(dev, 3)
    if "ref" not in _FULL:
        test_forward_full_size_vs_oracle_layer_by_layer(hiplib, dev, O, gemm_form)
    r = _FULL["ref"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def relerr(a, b):
        return float(np.abs(N(a) - b).max() / max(1.0, np.abs(b).max()))
    errs = {}
    net.store.refresh_split()
    M.arena_begin(dev)
    try:
        xt = T(x)
        cx, cp, _ = net.sa1.forward(xt, xt)
        assert (N(cx) == r["l1x"]).all()
        errs["sa1"] = relerr(cp, r["l1p"])
        for name, mod, xin, pin, xo, po in (("sa2", net.sa2, "l1x", "l1p", "l2x", "l2p"), ("sa3", net.sa3, "l2x", "l2p", "l3x", "l3p"),
                                            ("sa4", net.sa4, "l3x", "l3p", "l4x", "l4p")):
            cx, cp, _ = mod.forward(T(r[xin]), T(r[pin]))
            assert (N(cx) == r[xo]).all(), name
            errs[name] = relerr(cp, r[po])
        errs["fp1"] = relerr(net.fp1.forward(T(r["l3x"]), T(r["l4x"]), T(r["l3p"]), T(r["l4p"])), r["l3p2"])
        errs["fp2"] = relerr(net.fp2.forward(T(r["l2x"]), T(r["l3x"]), T(r["l2p"]), T(r["l3p2"])), r["seeds"])
        vx, vp = net.vote(T(r["l2x"]), T(r["seeds"]))
        errs["vote_xyz"], errs["vote_feat"] = relerr(vx, r["vx"]), relerr(vp, r["vp"])
        if "prop" not in _FULL:
            prev = O.set_threads(max(1, min(128, len(os.sched_getaffinity(0)))))
            try:
                _FULL["prop"] = oracle_sa(O, net.proposal, np.ascontiguousarray(r["vx"]), np.ascontiguousarray(r["vp"]), sample_xyz=r["l2x"])
            finally:
                O.set_threads(prev)
        px, pout = net.propose(T(r["vx"]), T(r["vp"]), T(r["l2x"]))
        assert (N(px) == _FULL["prop"][0]).all()
        errs["proposal"] = relerr(pout, _FULL["prop"][1])
    finally:
        M.arena_end()
    _record_parity("module_on_oracle_input", gemm_form, errs)
    assert max(errs.values()) < 1e-5, errs


def test_config5_scene_through_sa1_vs_oracle(hiplib, dev, O, gemm_form):
    """One 80 000-point config-5 scene through sa1 -- the L2-resident bucket FPS, the indexed ball query and the narrow
    first layer + two GEMMs + max over K at 131 072 grouped rows -- against the oracle: centres exact, features to 1e-5."""
    import os
    from votenet_amd import synth
    x = synth.room_batch(1, 80000, 77, size=(8.0, 3.0, 8.0), nbox=(15, 25))
    net = _perturbed_net(dev, 3)
    from votenet_amd import mlp as M
    net.store.refresh_split()
    M.arena_begin(dev)
    try:
        xt = torch.from_numpy(x).to(dev)
        l1x, l1p, _ = net.sa1.forward(xt, xt)
    finally:
        M.arena_end()
    if "cfg5" not in _FULL:
        prev = O.set_threads(max(1, min(128, len(os.sched_getaffinity(0)))))
        try:
            _FULL["cfg5"] = oracle_sa(O, net.sa1, x, x)
        finally:
            O.set_threads(prev)
    rx, rp = _FULL["cfg5"]
    assert (N(l1x) == rx).all()
    err = float(np.abs(N(l1p) - rp).max() / max(1.0, np.abs(rp).max()))
    _record_parity("config5_sa1", gemm_form, {"sa1": err})
    assert err < 1e-5, err


def test_predict_tail_nms_vs_oracle(hiplib, dev, O):
    """Predict tower (model.py:98-139): decoded boxes -> device NMS equals the oracle NMS on the same boxes."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = torch.from_numpy(synth.room_batch(2, 4096, 5)).to(dev)
    net = VM.VoteNetHotPath(dev, seed=7, npoints=(512, 256, 128, 64))
    r = net.predict(x, 0.25)
    boxes, score = N(r["bboxes"]), N(r["scores"])
    obj = N(r["proposals_output"][..., :2])
    assert boxes.shape == (2, 256, 8, 3)
    exp = O.nms3d(boxes, score, obj, 0.25)
    iou = np.stack([O.iou3d_matrix(boxes[s]) for s in range(2)])
    got = N(r["nms_idx"])
    # the comparison below is exact only when no pair sits within round-off of the threshold and no two scores tie (the visit order of
    # equal scores is unspecified in the reference): the precondition is ASSERTED for this seed, not a silent way around the comparison
    assert not (np.abs(iou - 0.25) < 1e-5).any() and len(np.unique(score)) == score.size
    assert got.shape == exp.shape and (got == exp).all()
    assert got.shape[1] == 2 and (obj[got[:, 0], got[:, 1], 1] > obj[got[:, 0], got[:, 1], 0]).all()


def _brute_ball(xyz, centre, r, k):
    """First-k-in-index-order neighbours of one centre, float32 arithmetic as tf_grouping_g.cu:14-24 (torch reference)."""
    d = xyz - centre
    s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    hit = torch.nonzero(torch.sqrt(s).clamp_min(1e-20) < r)[:, 0]
    return hit[:k]


def test_config5_dense_scan_forward_properties(hiplib, dev):
    """BASELINE config 5 (4 x 80 000-point scenes, 2048 -> 1024 seeds -> 512 -> 256, 256 proposals) at full size through
    size-independent properties of the path's pieces (one scene of it goes through sa1 against the oracle in
    test_config5_scene_through_sa1_vs_oracle)."""
    from votenet_amd import model as VM
    from votenet_amd import synth, tf_sampling as S
    x = torch.from_numpy(synth.room_batch(4, 80000, 77, size=(8.0, 3.0, 8.0), nbox=(15, 25))).to(dev)
    net = VM.VoteNetHotPath(dev, seed=3)
    tape = []
    out = net.forward(x, tape)
    assert out["proposals_output"].shape == (4, 256, 79) and out["seeds_xyz"].shape == (4, 1024, 3)
    for v in out.values():
        assert torch.isfinite(v).all()
    sa1 = tape[0]
    fidx = sa1["fps_idx"].long()
    assert (fidx[:, 0] == 0).all() and all(len(torch.unique(fidx[s])) == 2048 for s in range(4))
    assert (S.farthest_point_sample(300, x).long() == fidx[:, :300]).all()  # FPS prefixes nest
    # every FPS pick attains the maximum of the running min distance (float64 recomputation, first 200 picks of a scene)
    p = x[1].double()
    td = torch.full((80000,), float("inf"), dtype=torch.float64, device=dev)
    for j in range(1, 200):
        td = torch.minimum(td, ((p - p[fidx[1, j - 1]]) ** 2).sum(1))
        assert td[fidx[1, j]] >= td.max() * (1 - 1e-5)
    # ball query rows = first-K-in-index-order neighbours, padded with the first hit (sample of centres)
    idx, cnt, new_xyz = sa1["idx"], sa1["pts_cnt"], sa1["new_xyz"]
    assert torch.equal(new_xyz, torch.gather(x, 1, fidx[..., None].expand(-1, -1, 3)))
    g = torch.Generator().manual_seed(0)
    for s, j in zip(torch.randint(0, 4, (40,), generator=g).tolist(), torch.randint(0, 2048, (40,), generator=g).tolist()):
        ref = _brute_ball(x[s], new_xyz[s, j], 0.2, 64)
        c = int(cnt[s, j])
        assert c == len(ref) and torch.equal(idx[s, j, :c].long(), ref)
        assert (idx[s, j, c:] == idx[s, j, 0]).all()
    # the pooled layer output is permutation-invariant inside a group only through max: recompute one group by hand
    r2 = sa1["recs"][-1]
    z2 = r2["z"]
    if z2 is None:  # the pooled layer's z is not stored when its backward runs in Gram form: same GEMM, z kept
        from votenet_amd import mlp as M
        z2, _ = M.linear_dense(r2["x"], r2["layer"].p("W"), r2["layer"].p("b"), r2["in_scale"], r2["in_shift"], r2["in_relu"],
                               want_stats=False)
    if r2.get("half") is not None:  # compact rows (csrc/half.hip): a dropped slot is a copy of slot 0
        z2 = r2["half"].full_rows(z2)
    act = z2.view(4 * 2048, 64, -1) * r2["scale"] + r2["shift"]
    pooled = torch.where(act > 0, act, torch.zeros_like(act)).amax(1)
    again = net.sa1.forward(x, x, tape=None, geom=(sa1["fps_idx"], new_xyz, idx, cnt))[1]  # (a 4-tuple geometry: the full layout)
    if r2.get("half") is None:
        assert torch.equal(pooled.view(4, 2048, -1), again)
    else:  # the two layouts associate the BatchNorm sums differently: equal to rounding; the same layout again: equal
        assert float((pooled.view(4, 2048, -1) - again).abs().max()) < 1e-5 * float(again.abs().max())
        assert torch.equal(pooled.view(4, 2048, -1), net.sa1.forward(x, x, tape=None)[1])


def test_full_size_train_steps(hiplib, dev):
    """BASELINE config 3 shape (8 x 20 480 points): two optimizer steps run, gradients are finite and non-trivial, the
    parameters move, and the forward output of an unchanged input changes accordingly."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = torch.from_numpy(synth.room_batch(8, 20480, 1000)).to(dev)
    from votenet_amd import loss as VL
    net = VM.VoteNetHotPath(dev, seed=0)
    gt = VL.gt_to_device(synth.room_gt(8, 20480, 1000), dev)
    net.init_optimizer(1e-3)
    before = net.store.flat.clone()
    out0 = net.forward(x)["proposals_output"].clone()
    for _ in range(2):
        net.train_step(x, gt=gt)
    assert torch.isfinite(net.last_losses[1]) and net.last_losses[11] > 0  # vote loss finite, negatives exist
    torch.cuda.synchronize()
    g = net.store.grad
    assert torch.isfinite(g).all() and torch.isfinite(net.store.flat).all()
    assert float((g != 0).float().mean()) > 0.5
    step = (net.store.flat - before).abs()
    assert float(step.max()) <= 2.0e-3 * 1.001 and float(step.mean()) > 1e-4  # Adam: |delta| <= lr per step
    out1 = net.forward(x)["proposals_output"]
    assert not torch.equal(out0, out1) and torch.isfinite(out1).all()


def test_prefetched_geometry_is_the_same_computation(hiplib, dev):
    """prefetch_geometry(next_x) during a step, then the step on next_x: bit-identical outputs to the unpipelined call; a
    prefetch for another tensor (or a tensor modified since) is not used."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    xa = torch.from_numpy(synth.room_batch(2, 4096, 11)).to(dev)
    xb = torch.from_numpy(synth.room_batch(2, 4096, 21)).to(dev)
    net = VM.VoteNetHotPath(dev, seed=3, npoints=(512, 256, 128, 64))
    ref_b = net.forward(xb)
    net.forward(xa, next_x=xb)
    assert id(xb) in net._prefetched and net._prefetched[id(xb)][0] is xb
    got_b = net.forward(xb)
    assert not net._prefetched
    for k in ref_b:
        assert torch.equal(ref_b[k], got_b[k]), k
    net.forward(xa, next_x=xb)
    xb.add_(0.0)  # bumps the version: the prefetched geometry is dropped, not trusted
    again = net.forward(xb)
    for k in ref_b:
        assert torch.equal(ref_b[k], again[k]), k
    net.forward(xa, next_x=xa)
    other = net.forward(xb)  # prefetched for xa: ignored
    assert torch.equal(other["proposals_output"], ref_b["proposals_output"])
    # a lookahead of two on alternating prefetch streams
    xc = torch.from_numpy(synth.room_batch(2, 4096, 31)).to(dev)
    ref_c = net.forward(xc)
    net._prefetched.clear()
    net.forward(xa, next_x=[xb, xc])
    assert len(net._prefetched) == 2
    b2 = net.forward(xb, next_x=[xc, xa])
    c2 = net.forward(xc)
    for k in ref_b:
        assert torch.equal(ref_b[k], b2[k]) and torch.equal(ref_c[k], c2[k]), k


def test_geometry_graph_replays_are_the_launch_by_launch_chain(hiplib, dev, monkeypatch):
    """model.GeometryGraph: the prefetched chain replayed as one HIP graph gives bit-identical passes to the chain enqueued launch by
    launch, batch after batch around the ring of graphs; a prefetch whose graph has served another batch since is not trusted; a
    lookahead longer than the ring never replays the graph whose buffers the running pass reads."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    xs = [torch.from_numpy(synth.room_batch(2, 4096, 50 + i)).to(dev) for i in range(5)]
    net = VM.VoteNetHotPath(dev, seed=6, npoints=(512, 256, 128, 64))
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", False)
    refs = [net.forward(x) for x in xs]
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", True)
    net.forward(xs[0], next_x=xs[1])
    for i in range(1, 11):  # warm-up pass, three captures, then replays
        got = net.forward(xs[i % 5], next_x=xs[(i + 1) % 5])
        for k in refs[0]:
            assert torch.equal(got[k], refs[i % 5][k]), (i, k)
    ring = next(iter(net._geometry_rings.values()))
    assert len(ring["graphs"]) == VM.GEOMETRY_RING and sum(g.generation for g in ring["graphs"]) >= 9
    # the tape of a pass on replayed geometry drives the same backward pass
    def grads(x, **kw):
        tape = []
        out = net.forward(x, tape, **kw)
        net.store.grad.zero_()
        net.backward(tape, net.make_cotangents(2, seed=1))
        return out, net.store.grad.clone()
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", False)
    net._prefetched.clear()
    o_ref, g_ref = grads(xs[2])
    monkeypatch.setattr(VM, "GEOMETRY_GRAPHS", True)
    net.forward(xs[0], next_x=xs[2])
    assert net._prefetched[id(xs[2])][4] is not None  # a graph served it
    o_got, g_got = grads(xs[2])
    assert torch.equal(o_got["proposals_output"], o_ref["proposals_output"])
    assert (g_got - g_ref).abs().max() <= 1e-5 * g_ref.abs().max()  # atomics: the order of a sum varies run to run
    # stale: xs[3]'s graph serves other batches before xs[3] is asked for
    net._prefetched.clear()
    net.prefetch_geometry(xs[3])
    held = net._prefetched[id(xs[3])]
    for x in (xs[0], xs[1], xs[2], xs[4]):
        net.prefetch_geometry(x)
    net._prefetched[id(xs[3])] = held  # (the pool had dropped it: put the stale entry back)
    assert held[4].generation != held[5]
    got = net.forward(xs[3])
    assert torch.equal(got["proposals_output"], refs[3]["proposals_output"])
    # a lookahead of four with a ring of three: the graph under the running pass is skipped (that prefetch runs launch by launch)
    net._prefetched.clear()
    net.forward(xs[0], next_x=xs[1])
    got = net.forward(xs[1], next_x=[xs[2], xs[3], xs[4], xs[0]])
    assert torch.equal(got["proposals_output"], refs[1]["proposals_output"])
    got = net.forward(xs[0])
    assert torch.equal(got["proposals_output"], refs[0]["proposals_output"])


def test_outputs_and_tapes_on_graph_geometry_do_not_silently_go_stale(hiplib, dev):
    """Round-3 advice: a GeometryGraph's buffers are overwritten in place by later replays.  (i) The geometry tensor handed to the CALLER
    (seeds_xyz) is a copy outside the graph's pool: it keeps its values while the ring goes round; (ii) a tape recorded on prefetched
    geometry is refused by backward() once its graph has served another batch, and accepted until then."""
    from votenet_amd import VotenetError
    from votenet_amd import model as VM
    from votenet_amd import synth
    xs = [torch.from_numpy(synth.room_batch(2, 4096, 70 + i)).to(dev) for i in range(4)]
    net = VM.VoteNetHotPath(dev, seed=6, npoints=(512, 256, 128, 64))
    for i in range(5):  # warm-up pass + three captures
        net.forward(xs[i % 4], next_x=xs[(i + 1) % 4])
    net._prefetched.clear()
    net.prefetch_geometry(xs[0])
    tape = []
    out = net.forward(xs[0], tape)
    gg, gen = tape[0]["geometry_stamp"]
    assert gg is net._geometry_current and gen == gg.generation
    sa2_centres = tape[1]["new_xyz"]
    assert out["seeds_xyz"].data_ptr() != sa2_centres.data_ptr() and torch.equal(out["seeds_xyz"], sa2_centres)
    seeds = out["seeds_xyz"].clone()
    cot = net.make_cotangents(2, seed=1)
    net.store.grad.zero_()
    net.backward(tape, cot)                       # the tape's geometry is intact: fine
    for x in xs[1:] + xs[1:]:                     # prefetches alone never touch the graph under the LAST pass: its tape stays valid
        net.prefetch_geometry(x)
    assert gg.generation == gen
    net.backward(tape, cot)
    for i in range(1, 7):                         # later passes take over; the ring goes round and the graph replays for another batch
        net.forward(xs[i % 4], next_x=xs[(i + 1) % 4])
    torch.cuda.synchronize()
    assert gg.generation > gen
    assert torch.equal(out["seeds_xyz"], seeds)   # the caller's tensor did not change ...
    assert not torch.equal(sa2_centres, seeds)    # ... the graph's buffer did
    with pytest.raises(VotenetError, match="overwritten"):
        net.backward(tape, cot)


def test_stretch_graph_replays_are_the_launch_by_launch_step(hiplib, dev, monkeypatch):
    """model.StretchGraph: the static stretch of a train step (fp1 forward ... fp1 backward, moving averages and loss included) replayed
    as one HIP graph against the same step enqueued launch by launch.  Two replicas on the same batches; before every step the graph
    replica takes the launch replica's whole state (parameters, Adam moments, moving averages) -- Adam turns the last-bit noise of the
    backward pass's fp32 atomics into +-lr on near-zero gradients, so free-running replicas part within steps whatever the mechanism.
    Per step: forward results and losses bit-equal (the forward pass is order independent), gradient buckets equal to the run-to-run
    spread of the atomics, moving averages bit-equal; and the graph replica really replayed (one measuring step, one capture per shape)."""
    from votenet_amd import loss as VL
    from votenet_amd import model as VM
    from votenet_amd import synth
    b, n = 2, 4096
    xs = [torch.from_numpy(synth.room_batch(b, n, 90 + i)).to(dev) for i in range(3)]
    gts = [VL.gt_to_device(synth.room_gt(b, n, 90 + i), dev) for i in range(3)]
    ref = VM.VoteNetHotPath(dev, seed=5, npoints=(512, 256, 128, 64))
    got = VM.VoteNetHotPath(dev, seed=5, npoints=(512, 256, 128, 64))
    for net in (ref, got):
        net.init_optimizer(lr=1e-3)
        net._ema_state()
    for i in range(8):
        got.store.flat.copy_(ref.store.flat)
        got.store.params_changed()
        got._m.copy_(ref._m), got._v.copy_(ref._v), got._ema_flat.copy_(ref._ema_flat)
        res = []
        for net, flag in ((ref, False), (got, True)):
            monkeypatch.setattr(VM, "STRETCH_GRAPH", flag)
            out = net.train_step(xs[i % 3], gt=gts[i % 3], next_x=xs[(i + 1) % 3])
            torch.cuda.synchronize()
            res.append(({k: v.clone() for k, v in out.items()}, net.last_losses.clone(), net.store.grad.clone(), net._ema_flat.clone()))
        (o0, l0, g0, e0), (o1, l1, g1, e1) = res
        for k in o0:
            assert torch.equal(o0[k], o1[k]), (i, k)
        nn = lambda t: torch.nan_to_num(t, nan=-12345.0)  # (a batch without positives: the reference's empty means are NaN, here too)
        assert torch.equal(nn(l0), nn(l1)), (i, l0.tolist(), l1.tolist())
        assert torch.equal(e0, e1), i
        assert float((g0 - g1).abs().max()) <= 1e-4 * float(g0.abs().max()), i
    assert not ref.__dict__.get("_stretch_graphs") and len(got._stretch_graphs) == 1  # one capture: the graphs do not depend on the ground truth's shape
    assert sum(g.replays for g in got._stretch_graphs.values()) == 8 - 1  # every step but the measuring one
    # the outputs of a replayed step live in the capture's pool: a handle kept across the next step raises instead of reading that step's values
    from votenet_amd import VotenetError
    monkeypatch.setattr(VM, "STRETCH_GRAPH", True)
    o1 = got.train_step(xs[0], gt=gts[0])
    keep = o1["proposals_output"].clone()
    o2 = got.train_step(xs[1], gt=gts[1])
    assert o2["proposals_output"].shape == keep.shape
    with pytest.raises(VotenetError, match="replayed for a later step"):
        o1["proposals_output"]


def test_stretch_graphs_of_two_batch_shapes_keep_their_moving_average_factors(hiplib, dev, monkeypatch):
    """A captured stretch has the ADDRESS of its (1 - momentum | unbiased-variance) factor tensor baked into its votenet_ema_update node.
    Train at batch shape a, then b (whose first, launch-by-launch step builds another factor tensor), then a again: the replayed graph of
    shape a must still read ITS factors -- the moving averages stay bit-equal to a launch-by-launch replica's through the switches."""
    from votenet_amd import loss as VL
    from votenet_amd import model as VM
    from votenet_amd import synth
    n = 4096
    shapes = [2, 2, 2, 3, 3, 2, 3, 2]  # a: measuring step, capture + replay, replay; b: measuring, capture + replay; a, b, a: replays
    ref = VM.VoteNetHotPath(dev, seed=6, npoints=(512, 256, 128, 64))
    got = VM.VoteNetHotPath(dev, seed=6, npoints=(512, 256, 128, 64))
    for net in (ref, got):
        net.init_optimizer(lr=1e-3)
        net._ema_state()
    for i, b in enumerate(shapes):
        x = torch.from_numpy(synth.room_batch(b, n, 300 + i)).to(dev)
        gt = VL.gt_to_device(synth.room_gt(b, n, 300 + i), dev)
        got.store.flat.copy_(ref.store.flat)
        got.store.params_changed()
        got._m.copy_(ref._m), got._v.copy_(ref._v), got._ema_flat.copy_(ref._ema_flat)
        # churn the allocator between steps: a factor tensor that had gone back to it would be handed out again here and overwritten
        junk = [torch.full((got._ema_flat.numel(),), float("nan"), device=dev) for _ in range(8)]
        del junk
        ema = []
        for net, flag in ((ref, False), (got, True)):
            monkeypatch.setattr(VM, "STRETCH_GRAPH", flag)
            net.train_step(x, gt=gt)
            torch.cuda.synchronize()
            ema.append(net._ema_flat.clone())
        assert torch.isfinite(ema[1]).all(), i
        assert torch.equal(ema[0], ema[1]), (i, b, float((ema[0] - ema[1]).abs().max()))
    assert len(got._stretch_graphs) == 2
    assert sorted(g.replays for g in got._stretch_graphs.values()) == [2, 4]
    assert len(got._ema_fac_by_rows) == 2  # one factor tensor per shape, both alive


def test_moving_averages_follow_tensorflows_update(hiplib, dev):
    """The BatchNorm moving averages (reference: Tensorpack BNReLU, momentum 0.9): after a training-mode forward pass
    moving = 0.9 * moving + 0.1 * (batch mean | unbiased batch variance) for every BatchNorm layer, starting from 0 / 1; the
    batch statistics themselves are checked against torch on the stored z of a layer."""
    from votenet_amd import model as VM
    from votenet_amd import synth
    x = torch.from_numpy(synth.room_batch(2, 4096, 9)).to(dev)
    net = VM.VoteNetHotPath(dev, seed=4, npoints=(512, 256, 128, 64))
    names = [L.name for L in net._bn_layers()]
    assert len(names) == 23 and len(set(names)) == 23
    exp = {n: (torch.zeros(L.cout, device=dev), torch.ones(L.cout, device=dev)) for n, L in zip(names, net._bn_layers())}
    for _ in range(2):
        tape = []
        net.forward(x, tape)
        seen = 0
        for r in net._bn_records(tape):
            rows = r["rows"]
            m, v = exp[r["layer"].name]
            exp[r["layer"].name] = (0.9 * m + 0.1 * r["mean"], 0.9 * v + 0.1 * r["var"] * (rows / (rows - 1.0)))
            if r["z"] is not None and r["kind"] == "dense":
                z = r["z"].double()
                if r.get("half") is not None:  # compact rows (csrc/half.hip): the statistics run over the full layout's rows
                    z = r["half"].full_rows(z)
                    assert z.shape[0] == rows
                assert torch.allclose(r["mean"].double(), z.mean(0), rtol=1e-4, atol=1e-6)
                assert torch.allclose(r["var"].double(), z.var(0, unbiased=False), rtol=1e-4, atol=1e-7)
            seen += 1
        assert seen == 23
        net.update_moving_averages(tape)
    for n in names:
        assert torch.allclose(net._ema[n][2], exp[n][0], rtol=1e-6, atol=1e-7), n
        assert torch.allclose(net._ema[n][3], exp[n][1], rtol=1e-6, atol=1e-7), n


def test_predict_uses_moving_averages_so_a_scene_does_not_depend_on_its_batch_mates(hiplib, dev):
    """model.py:98-139 runs under `not is_training`: BatchNorm uses the moving averages, so the detections of a scene are
    the same whatever shares its batch (round-1 advice: predict() normalised with the batch's own statistics)."""
    from votenet_amd import loss as VL
    from votenet_amd import model as VM
    from votenet_amd import synth
    sa, sb, sc = (synth.room_batch(1, 4096, s) for s in (41, 42, 43))
    net = VM.VoteNetHotPath(dev, seed=5, npoints=(512, 256, 128, 64))
    xt = torch.from_numpy(synth.room_batch(2, 4096, 7)).to(dev)
    gt = VL.gt_to_device(synth.room_gt(2, 4096, 7), dev)
    for _ in range(3):  # a few optimizer steps: moving averages and weights away from their initial values
        net.train_step(xt, gt=gt)
    T = lambda *s: torch.from_numpy(np.concatenate(s)).to(dev)
    r_ab, r_ac, r_a = net.predict(T(sa, sb)), net.predict(T(sa, sc)), net.predict(T(sa))
    for k in ("proposals_output", "bboxes", "scores", "votes_xyz"):
        assert torch.equal(r_ab[k][0], r_ac[k][0]), k
        assert torch.allclose(r_ab[k][0], r_a[k][0], rtol=1e-5, atol=1e-6), k
        assert torch.isfinite(r_ab[k]).all()
    keep = lambda r: [int(b) for s, b in N(r["nms_idx"]) if s == 0]
    assert keep(r_ab) == keep(r_ac) == keep(r_a)
    # the training-mode normalisation does depend on the batch (what predict() did before)
    o_ab = net.predict(T(sa, sb), batch_statistics=True)["proposals_output"][0]
    o_ac = net.predict(T(sa, sc), batch_statistics=True)["proposals_output"][0]
    assert not torch.equal(o_ab, o_ac)
    # frozen scale / shift are cached until the weights or the averages change
    f1 = net.inference_bn()
    assert net.inference_bn() is f1
    net.train_step(xt, gt=gt)
    assert net.inference_bn() is not f1
