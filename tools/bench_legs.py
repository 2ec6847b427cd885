"""The measurement legs bench.py adds to its JSON line besides the headline value:

  cpu_baseline   the CPU oracle (oracle/, the restatement of the reference's CPU path) timed on the GPU box's host:
                 single thread (the reference's loop structure) AND all cores (the same sources with OpenMP), CPU model and
                 core counts stated, per-op medians of >= 10 runs, forward pass of whole scenes (the oracle has no backward:
                 the like-for-like GPU figure is the forward-only one, reported next to it)
  configs        every BASELINE.json configuration in the driver-observed line: 1 single SA layer (GPU vs CPU oracle),
                 2 backbone forward, 3 train step (= the headline) + predict tower with 3D NMS, 5 dense 80 000-point scan
  ball query     scanned-pair counts (reference algorithm / this kernel / all pairs) and the uniform-cube (no early exit) timing

Only bench.py imports this; oracle/ is used here as the thing timed for the CPU baseline, never by the GPU path.
"""
import os
import statistics
import time

import numpy as np


# ------------------------------------------------------------------------------------------------ host
def host_info():
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("processor"):
                logical += 1
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical = len(phys) or logical or usable
    return dict(cpu_model=model, physical_cores=physical, logical_cpus=logical or usable, usable_cpus=usable,
                threads_all=max(1, min(physical, usable)))


# ------------------------------------------------------------------------------------------------ CPU oracle forward
def cpu_forward(xyz):
    """The VoteNet layer stack (sa1-4, fp1-2, voting, proposal + mlp2) on the CPU oracle for a batch xyz (b, n, 3): the same
    ops in the same order as VoteNetHotPath.forward, BatchNorm over the whole batch.  -> seconds."""
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    b = xyz.shape[0]

    def mlp(x, dims, k=None, last_plain=False):
        for i in range(len(dims) - 1):
            w = (rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i])).astype(np.float32)
            z = O.linear(x, w, np.zeros(dims[i + 1], np.float32))
            if last_plain and i == len(dims) - 2:
                x = z
            else:
                mean, var = O.bn_stats(z)
                x = O.bn_relu(z, mean, var, np.ones(dims[i + 1], np.float32), np.zeros(dims[i + 1], np.float32))
        return O.max_over_k(x, k) if k else x

    def sa(xyz_, pts, m, r, k, widths, sample_xyz=None):
        fidx = O.farthest_point_sample(m, sample_xyz if sample_xyz is not None else xyz_)
        new_xyz = O.gather_point(xyz_, fidx)
        idx, _ = O.query_ball_point(r, k, xyz_, new_xyz)
        g = O.group_concat(xyz_, new_xyz, pts, idx).reshape(-1, 3 + pts.shape[2])
        return new_xyz, mlp(g, [g.shape[1]] + widths, k).reshape(b, m, -1)

    def fp(x1, x2, p1, p2, widths):
        dist, idx = O.three_nn(x1, x2)
        itp = O.three_interpolate(p2, idx, O.three_nn_weights(dist))
        x = np.concatenate([itp, p1], 2).reshape(-1, itp.shape[2] + p1.shape[2])
        return mlp(x, [x.shape[1]] + widths).reshape(b, x1.shape[1], -1)

    t0 = time.perf_counter()
    l1x, l1p = sa(xyz, xyz, 2048, 0.2, 64, [64, 64, 128])
    l2x, l2p = sa(l1x, l1p, 1024, 0.4, 64, [128, 128, 256])
    l3x, l3p = sa(l2x, l2p, 512, 0.8, 64, [128, 128, 256])
    l4x, l4p = sa(l3x, l3p, 256, 1.2, 64, [128, 128, 256])
    l3p2 = fp(l3x, l4x, l3p, l4p, [256, 256])
    seeds = fp(l2x, l3x, l2p, l3p2, [256, 256])
    x = np.concatenate([l2x, seeds], 2).reshape(-1, 259)
    votes = (x + mlp(x, [259, 256, 256, 259], last_plain=True)).reshape(b, 1024, 259)
    vx, vp = np.ascontiguousarray(votes[..., :3]), np.ascontiguousarray(votes[..., 3:])
    _, pp = sa(vx, vp, 256, 0.3, 64, [128, 128, 128], sample_xyz=l2x)
    mlp(pp.reshape(-1, 128), [128, 128, 128, 79], last_plain=True)
    return time.perf_counter() - t0


def _median_ms(fn, runs=10, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def cpu_ops(points, scene_kind, threads_all, runs=10):
    """Per-op medians (>= 10 runs after 2 warm-ups) of the oracle at sa1 size, at config 1's size and the fp2 three_nn, in ms PER
    SCENE: single thread on one scene; all cores on a batch of 8 scenes (the workload's batch: FPS parallelises over scenes
    only, its rounds are a dependent chain), divided by 8."""
    from oracle import oracle as O
    from votenet_amd import synth
    gen = synth.room_batch if scene_kind == "room" else synth.uniform_batch
    res = {}
    for label, th, nb in (("1t", 1, 1), ("all", threads_all, 8)):
        xyz = gen(nb, points, 1000)
        c1 = np.stack([np.random.default_rng(s).random((2048, 3), dtype=np.float32) for s in range(nb)])
        prev = O.set_threads(th)
        try:
            ctr = O.gather_point(xyz, O.farthest_point_sample(2048, xyz))
            q1 = O.gather_point(c1, O.farthest_point_sample(512, c1))
            l2 = O.gather_point(ctr, O.farthest_point_sample(1024, ctr))
            l3 = O.gather_point(l2, O.farthest_point_sample(512, l2))
            ops = {
                "fps_sa1 (%d -> 2048)" % points: lambda: O.farthest_point_sample(2048, xyz),
                "ball_query_sa1 (2048 x %d, r 0.2, K 64)" % points: lambda: O.query_ball_point(0.2, 64, xyz, ctr),
                "three_nn_fp2 (1024 x 512)": lambda: O.three_nn(l2, l3),
                "config1_fps (2048 -> 512)": lambda: O.farthest_point_sample(512, c1),
                "config1_ball_query (512 x 2048, r 0.2, K 32)": lambda: O.query_ball_point(0.2, 32, c1, q1),
            }
            for name, fn in ops.items():
                res.setdefault(name, {})["ms_per_scene_" + label] = round(_median_ms(fn, runs) / nb, 4)
        finally:
            O.set_threads(prev)
    return res


def cpu_baseline(points, scene_kind, min_seconds=8.0, max_scenes=3, batch_all=4, ops=True):
    """-> the `cpu_baseline` object of the bench line.  Bounded: about 10 s single thread + about 10 s all cores + the per-op
    medians (a few seconds)."""
    from oracle import oracle as O
    from votenet_amd import synth
    gen = synth.room_batch if scene_kind == "room" else synth.uniform_batch
    h = host_info()
    # (i) single thread: the reference's loop structure, one scene at a time
    tot1, k1 = 0.0, 0
    O.set_threads(1)
    while k1 == 0 or (k1 < max_scenes and tot1 < min_seconds):
        tot1 += cpu_forward(gen(1, points, 1000 + k1))
        k1 += 1
    # (ii) all cores: OpenMP over independent queries / rows / FPS lanes, a batch of scenes per pass
    tota, ka = 0.0, 0
    prev = O.set_threads(h["threads_all"])
    try:
        while ka == 0 or (ka < max_scenes * batch_all and tota < min_seconds):
            tota += cpu_forward(gen(batch_all, points, 2000 + ka))
            ka += batch_all
    finally:
        O.set_threads(prev)
    v1, va = k1 / tot1, ka / tota
    out = {"value": va, "unit": "scenes/s", "cores": h["threads_all"], "kind": "port", "what": "forward only (the oracle has no backward); "
           "compare with configs.config2_backbone_forward.forward_stack_scenes_per_s, the GPU's forward-only figure",
           "value_1t": v1, "value_all": va, "cpu_model": h["cpu_model"], "physical_cores": h["physical_cores"],
           "logical_cpus": h["logical_cpus"], "usable_cpus": h["usable_cpus"],
           "sample": "forward pass (sa1-4, fp1-2, voting, proposal) of synthetic %d-pt %s scenes on the CPU oracle: %d scene(s) single "
                     "thread in %.1f s; %d scenes in batches of %d on %d OpenMP threads in %.1f s"
                     % (points, scene_kind, k1, tot1, ka, batch_all, h["threads_all"], tota)}
    if ops:
        out["ops_ms"] = cpu_ops(points, scene_kind, h["threads_all"])
        out["ops_note"] = ("ms per scene, median of 10 runs after 2 warm-ups: one scene per call single-thread, 8 scenes per call on all cores; FPS is the restatement of tf_sampling_g.cu:105-170 "
                           "(the reference has no CPU FPS), ball query / three_nn restate test/query_ball_point.cpp:19-84 and "
                           "tf_interpolate.cpp:60-103 and are bit-identical to those compiled (tests/test_oracle_golden.py)")
    return out


# ------------------------------------------------------------------------------------------------ GPU legs
def gpu_ms(fn, it=10, warm=3):
    """Average ms per call from HIP events on the current stream (inputs resident)."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def ball_query_pairs(idx, cnt, n, chunk=4096, wg_queries=64):
    """Scanned (query, candidate) pairs of a ball query, from its outputs:
      reference  tf_grouping_g.cu:13-35 stops a query at its K-th hit: sum over queries of (index of the K-th hit + 1, else n)
      full_scan  ball_query_kernel<16> (no index) scans super-chunks of 4096 candidates for a workgroup of 64 queries until all
                 64 are full
      all        m * n per scene."""
    import torch
    b, m, k = idx.shape
    full = cnt >= k
    last = torch.where(full, idx[..., k - 1].long() + 1, torch.full_like(cnt, n).long())
    ref = int(last.sum().item())
    pad = (-m) % wg_queries
    lw = torch.nn.functional.pad(last, (0, pad), value=0).view(b, -1, wg_queries).amax(2)
    scanned = torch.clamp((lw + chunk - 1) // chunk * chunk, max=n)
    nvalid = torch.full((lw.shape[1],), wg_queries, device=idx.device)
    if pad:
        nvalid[-1] = wg_queries - pad
    kern = int((scanned * nvalid[None, :]).sum().item())
    return dict(reference_algorithm=ref, kernel=kern, all_pairs=b * m * n)


def indexed_pairs(index, xyz2, n, radius):
    """(query, candidate) pairs ball_query_indexed_kernel tests: 64 per bucket whose box passes the kernel's conservative
    test, recomputed here from the index's boxes with the same fp32 expression."""
    import ctypes
    import torch
    from votenet_amd import _lib as L
    b, m, _ = xyz2.shape
    nb = (n + 63) // 64
    L.lib().votenet_ball_threshold.restype = ctypes.c_float
    L.lib().votenet_ball_threshold.argtypes = [ctypes.c_float]
    thr = float(L.lib().votenet_ball_threshold(float(radius)))
    box = index[b * n:b * n + b * nb * 6].view(b, 1, nb, 6)
    q = xyz2.view(b, m, 1, 3)
    e = torch.clamp(torch.maximum(box[..., :3] - q, q - box[..., 3:]), min=0.0)
    lb = ((e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]) * 0.99999
    return int((~(lb >= thr)).sum().item()) * 64


def ball_query_detail(sa, x_room, x_unif):
    """sa1's ball query alone on the GPU, room scenes and the uniform cube: over the spatial index (what the path runs) and
    the full scan, with the pairs each of them tests."""
    from votenet_amd import tf_grouping as G
    from votenet_amd import tf_sampling as S
    out = {}
    for name, x in (("room", x_room), ("uniform", x_unif)):
        b, n = x.shape[:2]
        ctr = S.gather_point(x, S.farthest_point_sample(sa.npoint, x))  # leaves the index of x behind
        ms_i = gpu_ms(lambda: G.query_ball_point(sa.radius, sa.nsample, x, ctr), it=20)
        idx, cnt = G.query_ball_point(sa.radius, sa.nsample, x, ctr)
        G.USE_INDEX = False
        try:
            ms_f = gpu_ms(lambda: G.query_ball_point(sa.radius, sa.nsample, x, ctr), it=10)
        finally:
            G.USE_INDEX = True
        p = ball_query_pairs(idx, cnt, n)
        pi = indexed_pairs(S.cached_index(x), ctr, n, sa.radius)
        alg = b * sa.npoint * n * 12 + b * sa.npoint * (sa.nsample + 1) * 4
        out[name] = dict(ms_alone=round(ms_i, 4), ms_alone_full_scan=round(ms_f, 4), mean_pts_cnt=round(float(cnt.float().mean()), 2),
                         scanned_pairs=dict(reference_algorithm=p["reference_algorithm"], full_scan_kernel=p["kernel"],
                                            indexed_kernel=pi, all_pairs=p["all_pairs"]),
                         indexed_pairs_per_s=round(pi / (ms_i * 1e-3), 1), hbm_model_frac=round(alg / (ms_i * 1e-3) / 8e12, 4),
                         hbm_model_frac_full_scan=round(alg / (ms_f * 1e-3) / 8e12, 4))
    out["note"] = ("hbm_model_frac is SURVEY 8d's all-pairs byte model (B m n 12 + B m (K+1) 4) / time / 8 TB/s: it exceeds 1 because the "
                   "kernel tests only the buckets of the candidates' spatial index whose box reaches into the ball (scanned_pairs."
                   "indexed_kernel of all_pairs), and those from L2; the full scan (no index) is the same figure for every pair tested")
    return out


def config_legs(net, xs, gts, dev, B, n, cpu=True):
    """BASELINE.json configs 1, 2, 3 (predict tower) and 5 on the GPU, each a short HIP-event timing with resident inputs."""
    import torch
    from votenet_amd import pointnet2 as P
    from votenet_amd import synth
    from votenet_amd import tf_grouping as G
    from votenet_amd import tf_sampling as S
    cfg = {}
    # ---- config 1: single SA layer, 2048-pt random cloud, FPS -> 512, ball query r 0.2 K 32, MLP 64,64,128 + max-pool
    store = P.ParamStore(dev)
    sa = P.SAModule(store, "cfg1", 512, 0.2, 32, 3, [64, 64, 128])
    store.materialize(1)
    c1 = {}
    for b1 in (1, 32):
        x1 = torch.from_numpy(np.stack([np.random.default_rng(s).random((2048, 3), dtype=np.float32) for s in range(b1)])).to(dev)
        ctr = S.gather_point(x1, S.farthest_point_sample(512, x1))
        c1["b%d" % b1] = dict(fps_ms=round(gpu_ms(lambda: S.farthest_point_sample(512, x1), it=20), 4),
                              ball_query_ms=round(gpu_ms(lambda: G.query_ball_point(0.2, 32, x1, ctr), it=20), 4),
                              sa_layer_ms=round(gpu_ms(lambda: sa.forward(x1, x1), it=20), 4))
    if cpu:
        from oracle import oracle as O
        xc = np.random.default_rng(0).random((1, 2048, 3), dtype=np.float32)

        def cpu_layer():
            f = O.farthest_point_sample(512, xc)
            q = O.gather_point(xc, f)
            idx, _ = O.query_ball_point(0.2, 32, xc, q)
            a = O.group_concat(xc, q, xc, idx).reshape(-1, 6)
            rng = np.random.default_rng(1)
            for ci, co in ((6, 64), (64, 64), (64, 128)):
                z = O.linear(a, (rng.normal(size=(ci, co)) * np.sqrt(2.0 / ci)).astype(np.float32), np.zeros(co, np.float32))
                mean, var = O.bn_stats(z)
                a = O.bn_relu(z, mean, var, np.ones(co, np.float32), np.zeros(co, np.float32))
            return O.max_over_k(a, 32)
        c1["cpu_oracle_sa_layer_ms_1t"] = round(_median_ms(cpu_layer, 10), 3)
    c1["what"] = "BASELINE configs[0]: FPS 2048 -> 512, ball query r 0.2 K 32, grouped MLP 64,64,128 + max over K (features = xyz)"
    cfg["config1_single_sa_layer"] = c1
    # ---- config 2: backbone forward (+ voting + proposal), batch 8 x 20480
    k = [0]

    def fwd(pipe):
        i = k[0]
        k[0] += 1
        return net.forward(xs[i % 3], next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]] if pipe else None)
    ms_p = gpu_ms(lambda: fwd(True), it=20, warm=6)
    net.__dict__.get("_prefetched", {}).clear()
    ms_u = gpu_ms(lambda: fwd(False), it=20, warm=3)
    cfg["config2_backbone_forward"] = dict(forward_stack_ms=round(ms_p, 3), forward_stack_scenes_per_s=round(B / ms_p * 1e3, 1),
                                           without_cross_step_pipelining_ms=round(ms_u, 3),
                                           what="BASELINE configs[1] + voting + proposal: forward of %d x %d-pt scenes, geometry of the next "
                                                "batches prefetched (three batches rotate, every call computes one full geometry)" % (B, n))
    # ---- config 3, second half: predict tower = forward (BatchNorm on moving averages) + box decode + 3D NMS at 0.25
    def pred(sync):
        i = k[0]
        k[0] += 1
        return net.predict(xs[i % 3], 0.25, next_x=[xs[(i + 1) % 3], xs[(i + 2) % 3]], sync=sync)
    ms_a = gpu_ms(lambda: pred(False), it=20, warm=6)
    ms_s = gpu_ms(lambda: pred(True), it=20, warm=3)
    r = pred(True)
    net.__dict__.get("_prefetched", {}).clear()
    cfg["config3_predict_tower"] = dict(ms_padded_output=round(ms_a, 3), scenes_per_s=round(B / ms_a * 1e3, 1), ms_host_sized_output=round(ms_s, 3),
                                        kept_boxes=int(r["nms_idx"].shape[0]),
                                        what="model.py:98-139 on %d x %d-pt scenes: forward in inference mode (moving-average BatchNorm after the "
                                             "bench's training steps) + decode + NMS3D(thr 0.25); padded = kept list left on the device with its "
                                             "count (no host sync), host_sized = the reference's (Nsel, 2) shape" % (B, n))
    # ---- config 5: dense scan, 4 x 80000 points (8 x 8 x 3 m rooms, 15-25 boxes), 2048 -> 1024 seeds -> 512 -> 256, 256 proposals
    x5 = torch.from_numpy(synth.room_batch(4, 80000, 77, size=(8.0, 3.0, 8.0), nbox=(15, 25))).to(dev)
    fps5 = gpu_ms(lambda: S.farthest_point_sample(2048, x5), it=5, warm=2)
    ctr5 = S.gather_point(x5, S.farthest_point_sample(2048, x5))
    bq5 = gpu_ms(lambda: G.query_ball_point(0.2, 64, x5, ctr5), it=5, warm=2)
    fw5 = gpu_ms(lambda: net.forward(x5), it=5, warm=2)
    alg5 = 4 * 2047 * 80000 * 16 + 4 * 80000 * 12 + 4 * 2048 * 4
    cfg["config5_dense_scan"] = dict(forward_ms=round(fw5, 3), scenes_per_s=round(4 / fw5 * 1e3, 1), fps_ms=round(fps5, 3), ball_query_ms=round(bq5, 4),
                                     fps_effective_GBs=round(alg5 / (fps5 * 1e-3) / 1e9, 1), fps_hbm_model_frac=round(alg5 / (fps5 * 1e-3) / 8e12, 4),
                                     what="BASELINE configs[4]: 4 x 80000-pt scenes, forward; forward_ms = one batch by itself (its own geometry "
                                          "inside the call), forward_pipelined_ms = three batches rotating with the next batches' geometry on the side "
                                          "stream, as the other legs run; FPS = fps_bucket_l2_kernel (L2-resident, exact bucket pruning), "
                                          "algorithmic bytes B(m-1)n16 + Bn12 + Bm4")
    # the same forward the way the other legs are measured: three batches rotate, the coordinate-only geometry of the next ones prefetched
    try:
        x5s = [x5] + [torch.from_numpy(synth.room_batch(4, 80000, 77 + 1000 * j, size=(8.0, 3.0, 8.0), nbox=(15, 25))).to(dev) for j in (1, 2)]
        k5 = [0]

        def fwd5():
            i = k5[0]
            k5[0] += 1
            return net.forward(x5s[i % 3], next_x=[x5s[(i + 1) % 3], x5s[(i + 2) % 3]])
        fw5p = gpu_ms(fwd5, it=9, warm=6)
        cfg["config5_dense_scan"].update(forward_pipelined_ms=round(fw5p, 3), scenes_per_s_pipelined=round(4 / fw5p * 1e3, 1))
    except Exception as e:  # never lose the line to this leg
        cfg["config5_dense_scan"]["forward_pipelined_error"] = repr(e)[:300]
    net.__dict__.get("_prefetched", {}).clear()
    return cfg
