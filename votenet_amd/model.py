"""Driver of the hot path in VoteNet's layer configuration (the caller side of the path).

It wires the SA / FP / voting / proposal layers with the exact shapes and hyper-parameters of
model.py:39-49,53-61,89-93, the reference's loss graph (loss.py: model.py:61-84,141-231 as one kernel),
the predict tower (box decode -> 3D NMS, model.py:98-139) and the optimizer (model.py:240-250), so that
the hot path can be driven, checked and timed end to end:

    sa1 20480->2048 r0.2 K64 [64,64,128]     sa2 ->1024 r0.4 [128,128,256]
    sa3 ->512 r0.8 [128,128,256]             sa4 ->256 r1.2 [128,128,256]
    fp1 (l3<-l4) [256,256]                   fp2 (l2<-l3) [256,256]         seeds = l2_xyz
    voting FC 259->256->256->259 (BNReLU,BNReLU,none); votes = [seed_xyz, seed_feat] + offset
    proposal SA on votes, FPS on seeds (utils.py:42-43), 256 x r0.3 K64 [128,128,128] + [128,128,79]
"""

import torch

from . import dp
from . import mlp as M
from . import pointnet2 as P

NH, NS, NC = 12, 10, 10  # config.py:2-3
PREFETCH_AFTER = 2  # when the geometry chain of the next batch is enqueued: -1 at the start of the pass, k after the forward of sa<k>, 5 at the start of the backward pass.  Its FPS holds 8 CUs for 1.7 ms and every GEMM beside it runs ~18 % longer (tools/probe/gemm_beside_fps.py): after sa2 it falls on the small-kernel stretch of the pass (7.03 -> 6.88 ms per step, same box)
SIDE_PRIORITY = 0   # HIP stream priorities of the geometry (prefetch) streams and of the weight-gradient stream (0 = normal)
WGRAD_PRIORITY = 0
PROPOSAL_NUM = 256       # config.py:6
PROPOSAL_OUT = 5 + 2 * NH + 4 * NS + NC  # model.py:91 -> 79


GEOMETRY_GRAPHS = True  # prefetch_geometry replays the coordinate-only chain of a batch as ONE HIP graph (GeometryGraph) instead of enqueuing its ~50 launches
GEOMETRY_MAX_EVICTIONS = 8  # a net whose input shape keeps changing stops capturing (two shapes / configurations are kept side by side)
GEOMETRY_RING = 3       # graphs (= sets of geometry buffers) per input shape: a prefetch may run two batches ahead of the step that consumes it


class GeometryGraph:
    """The coordinate-only chain of one batch -- four FPS + ball queries, the piece layouts and their compact rows, the proposal
    layer's FPS, both three_nn: ~50 launches that depend on nothing but the points -- captured once on a prefetch stream and replayed
    per batch.  The chain's buffers live in the graph's memory pool at fixed addresses: a replay overwrites the geometry of the batch
    the graph served before, which is why a net keeps a ring of GEOMETRY_RING of them and tags every hand-out with the graph's
    generation.  The piece counts still reach the host through mapped pinned ints (the graph's own), the layouts are re-armed with the
    event that follows the replay."""

    def __init__(self, net, x, side):
        self.x = torch.empty_like(x)
        self.slots = torch.zeros(16, dtype=torch.int32).pin_memory()
        self.generation = 0
        self.graph = torch.cuda.CUDAGraph()
        g = {}
        # thread_local: a process group's watchdog thread (RCCL, one rank per GPU) queries its events while this thread captures
        with M.layout_slots(self.slots), torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            net._geometry_chain(self.x, g, None, ("sa1", "sa2", "sa3", "sa4"))
        self.g = g
        self.layouts = [t for v in g.values() if isinstance(v, tuple) for t in v if isinstance(t, M.HalfLayout)]

    def replay(self, x, side):
        """On `side` (current): copy the points in, run the chain -> (g, ev) as _geometry_chain leaves them."""
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        done = torch.cuda.Event()
        done.record(side)
        self.generation += 1
        for h in self.layouts:
            h.rearm(done)
        return self.g, {name: done for name in ("sa1", "sa2", "sa3", "sa4", "fp")}


GRAM_EARLY = ()          # levels whose Gram matrix is launched at the start of the levels' backward pass (see _grams_early): measured below
GRAMS_AHEAD = False      # train_step launches the Gram matrices of the levels' pooled layers (forward data only) on the weight-gradient stream under the stretch instead of beside the levels' backward GEMMs.  Measured (tools/probe/variant_step.py, three alternations): 3.78-3.81 -> 3.86-3.89 ms -- the stretch is a chain of tiny latency-bound kernels ON the critical path, and 0.27 ms of GPU-filling kernels beside it cost it more than they save the backward pass.  Off.
INLINE_WGRAD_TAIL = False  # experiment (see _levels_backward): the weight gradients of sa2 / sa1 on the main stream
STRETCH_GRAPH = True     # train_step replays its static stretch (fp1 forward ... fp1 backward: ~85 launches) as HIP graph segments (StretchGraph)
STRETCH_SEGMENTS = True  # the stretch's input-gradient chain cut into graphs at the modules' ends, the weight gradients launched between them (False: two graphs around the loss, weight gradients inline)
STRETCH_MAX_GRAPHS = 4   # captures kept per net (one per input shape / configuration; least recently replayed goes first)


class _PrivateArena:
    """`with _PrivateArena(buf, nd):` every zero-initialised scratch request of mlp._StatsArena comes out of `buf` (nd doubles of fp64
    region, the rest fp32) instead of the step's arena; the enclosing pass's arena state is restored on exit.  The owner clears buf."""
    FIELDS = ("buf", "nd", "off", "cap32", "off32", "want32", "zeroed32", "depth", "active", "whole_step")

    def __init__(self, buf, nd):
        self.buf, self.nd = buf, nd

    def __enter__(self):
        a = M._StatsArena
        self.saved = {k: getattr(a, k) for k in self.FIELDS}
        a.buf, a.nd, a.off, a.off32, a.want32 = self.buf, self.nd, 0, 0, 0
        a.cap32 = a.zeroed32 = (self.buf.numel() - self.nd) * 2
        a.depth, a.active, a.whole_step = 1, True, True
        return self

    def __exit__(self, *exc):
        a = M._StatsArena
        self.used = (a.off, a.want32)
        for k, v in self.saved.items():
            setattr(a, k, v)
        return False


class _StretchOutputs(dict):
    """What a replayed train step returns: the forward results in the capture's memory pool, which the NEXT replay of that capture
    overwrites.  Reading an entry after that raises instead of handing out another batch's values (clone what has to outlive the step)."""

    def __init__(self, tensors, graph):
        super().__init__(tensors)
        self._graph, self._stamp = graph, graph.replays

    def _check(self, key="them"):
        if self._graph.replays != self._stamp:
            raise M.L.VotenetError("train_step outputs: the stretch graph that holds %r has been replayed for a later step (the tensors a "
                                   "replayed step returns are valid until the next step: clone them to keep them)" % key)

    def __getitem__(self, key):
        self._check(key)
        return super().__getitem__(key)

    def get(self, key, default=None):
        self._check(key)
        return super().get(key, default)

    def items(self):
        self._check()
        return super().items()

    def values(self):
        self._check()
        return super().values()


class StretchGraph:
    """The static stretch of a train step -- feature propagation, voting and the proposal module forward (model.py:48-61,89-93), the
    moving averages, the loss graph (model.py:61-84,141-231), and the backward pass of all of it down to the gradients of the level
    outputs: ~85 launches of 5-40 us on static shapes -- captured ONCE and replayed per step.  What it buys is host time: the host
    enqueued this stretch in 1.7 ms against 1.17 ms of GPU time, i.e. the GPU waited for the host here (tools/probe/stretch_time.py).

    Segments.  The input-gradient chain is captured as FOUR graphs sharing one memory pool, cut where a module's backward ends (proposal |
    voting | fp2 | fp1); the weight-gradient launches of a segment (14 in all) are not captured but recorded as thunks and run launch by
    launch on the weight-gradient stream after their segment's graph, beside the next segment -- where they ran before.  (One graph
    with the weight gradients on a side branch replays its branches concurrently -- tools/probe/graph_branches.py -- but the branch's
    internal stream shared a hardware queue with the geometry prefetch and the graph waited for the sampling kernel: 3.94 -> 5.8 ms per
    step; inline on the one branch: 4.27 ms; tools/probe/stretch_modes.py.)

    Inputs arrive at changing addresses (level outputs from the allocator, geometry from a ring of GeometryGraphs, the batch's ground
    truth): ONE copy launch (votenet_copy_segments) moves them into the graph's fixed input buffers.  Scratch that kernels accumulate
    into comes from a private arena cleared by a fill node of the first graph.  Outputs (the forward results, the losses, d_l2p / d_l3p /
    d_l4p) live in the pool: valid until the next replay."""

    def __init__(self, net, ins, tape_levels, arena_demand):
        dev = net.device
        self.names = list(ins)
        self.fixed = {k: torch.empty_like(v) for k, v in ins.items()}
        nd = (int(arena_demand[0] * 1.25) + 4096 + 1) & ~1
        n32 = int(arena_demand[1] * 1.25) + (1 << 18)
        self.nd = nd
        self.arena = torch.empty(nd + (n32 + 1) // 2, dtype=torch.float64, device=dev)
        self.segments = []  # (graph, [(thunk, tensors) ...]) in replay order
        self._losses = torch.empty(12, dtype=torch.float32, device=dev)
        self._loss_ring = [torch.empty(12, dtype=torch.float32, device=dev) for _ in range(8)]
        self.replays = 0
        self.last_used = 0
        if getattr(net, "_capture_stream", None) is None:
            net._capture_stream = torch.cuda.Stream(device=dev)
        cs = net._capture_stream
        net.store._fresh_wait()  # the wait for this step's W^T copies happens OUTSIDE the graphs (train_step repeats it before a replay)
        self.copy_inputs(ins)    # (the capture executes nothing; this keeps the buffers defined)
        import gc
        torch.cuda.synchronize(dev)
        gc.collect()
        pool = torch.cuda.graph_pool_handle()
        keep, defer = [], []
        prev = (P.CAPTURE_KEEP, P.CAPTURE_DEFER, P.WGRAD_STREAM)
        P.CAPTURE_KEEP, P.CAPTURE_DEFER = keep, (defer if STRETCH_SEGMENTS else None)
        cur = [None]

        def begin():
            cur[0] = torch.cuda.CUDAGraph()
            cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")  # thread_local: see GeometryGraph

        def end():
            cur[0].capture_end()
            self.segments.append((cur[0], list(defer)))
            defer.clear()
            cur[0] = None

        def cut():
            if STRETCH_SEGMENTS:
                end()
                begin()

        def loss_hook(out):
            # the loss graph (two launches on the batch's ground truth: its shape -- the padded box count -- varies per batch) is NOT
            # captured: it runs between the forward segment and the first backward segment, into buffers the latter reads at fixed
            # addresses (a carve-out of the private arena: cleared by the first segment's fill)
            from . import loss as VL
            end()
            self.loss_bufs = VL.loss_buffers(out, losses=self._losses)  # (the losses vector: allocated on the caller's stream, below)
            self.loss_at = len(self.segments)  # the loss runs before this segment
            losses, flat = self.loss_bufs
            nv, npx = out["votes_xyz"].numel(), out["proposals_xyz"].numel()
            npo = out["proposals_output"].numel()
            cot = dict(votes_xyz=flat[:nv].view_as(out["votes_xyz"]), proposals_xyz=flat[nv:nv + npx].view_as(out["proposals_xyz"]),
                       proposals_output=flat[nv + npx:nv + npx + npo].view_as(out["proposals_output"]))
            begin()
            return losses, cot
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            begin()
            try:
                self.arena.zero_()
                with _PrivateArena(self.arena, nd) as pa:
                    self.out, self.losses, self.grads = net._stretch_body(self.fixed, tape_levels, None, cut=cut, loss_hook=loss_hook)
            finally:
                if cur[0] is not None:
                    end()
                P.CAPTURE_KEEP, P.CAPTURE_DEFER, P.WGRAD_STREAM = prev
        self.arena_used = pa.used
        self.keep = keep

    def copy_inputs(self, ins):
        M.copy_segments([(self.fixed[k], ins[k]) for k in self.names])

    def replay(self, ins, gt, wgrad_stream=None):
        """One copy launch, then the segments in order -- the loss graph on gt as launches where it belongs --; a segment's
        weight-gradient thunks go to wgrad_stream (None: the current one) behind its graph.  The caller joins the weight-gradient stream
        (pointnet2.wgrad_join) before it reads the gradient bucket."""
        from . import loss as VL
        self.copy_inputs(ins)
        prev = P.WGRAD_STREAM
        P.WGRAD_STREAM = wgrad_stream
        try:
            for i, (graph, thunks) in enumerate(self.segments):
                if i == self.loss_at:
                    # the loss launch is not captured and no captured kernel reads the 12 losses: each replay writes them into the next
                    # vector of a small ring, so a handle to last_losses kept across a few steps still shows ITS step (no copy launch)
                    losses = self._loss_ring[self.replays % len(self._loss_ring)]
                    VL.votenet_loss(self.out, gt, buffers=(losses, self.loss_bufs[1]))
                graph.replay()
                if thunks and wgrad_stream is not None:
                    P._hand_over([f for f, _ in thunks], ())  # (their tensors live in the graphs' pool: nothing for the allocator to track)
                else:
                    for f, _ in thunks:
                        f()
        finally:
            P.WGRAD_STREAM = prev
        self.replays += 1
        return _StretchOutputs(self.out, self), losses, self.grads


SPLIT_BF16 = True  # fused GEMMs on bf16 x 3 split operands (fp32-accurate products, six bf16 MFMAs per k-step; mlp.SplitImages)


class VoteNetHotPath:
    def __init__(self, device, seed=0, npoints=(2048, 1024, 512, 256)):
        self.device = device
        self.overlap_wgrad = True  # weight gradients on a second stream, next to the input-gradient chain
        self._wgrad_stream = None
        s = P.ParamStore(device)
        self.store = s
        n1, n2, n3, n4 = npoints
        self.sa1 = P.SAModule(s, "sa1", n1, 0.2, 64, 3, [64, 64, 128], leaf=True)      # model.py:39 (l0_points = xyz, C=3)
        self.sa2 = P.SAModule(s, "sa2", n2, 0.4, 64, 128, [128, 128, 256])  # model.py:41
        self.sa3 = P.SAModule(s, "sa3", n3, 0.8, 64, 256, [128, 128, 256])  # model.py:43
        self.sa4 = P.SAModule(s, "sa4", n4, 1.2, 64, 256, [128, 128, 256])  # model.py:45
        self.fp1 = P.FPModule(s, "fp1", 256, 256, [256, 256])               # model.py:48
        self.fp2 = P.FPModule(s, "fp2", 256, 256, [256, 256])               # model.py:49
        self.voting = P.make_mlp(s, "voting", 259, [256, 256, 259], "fc", last_plain=True)  # model.py:53-57
        self.proposal = P.SAModule(s, "proposal", PROPOSAL_NUM, 0.3, 64, 256, [128, 128, 128],
                                   mlp2=[128, 128, PROPOSAL_OUT])           # model.py:89-93
        s.materialize(seed)
        s.enable_split(SPLIT_BF16)  # forward() refreshes the images at its start, refresh_transposes() those of the copies

    # ---- forward pieces -------------------------------------------------------------
    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device, priority=SIDE_PRIORITY)
        return self._side

    def _geometry_chain(self, x, g, ev, levels):
        """FPS / ball query of `levels`, the proposal layer's FPS (it samples the SEEDS, utils.py:42-43) and both three_nn on
        the current stream; an event per level so that a consumer waits only for what it needs."""
        xyz = x if "sa1" in levels else g["sa1"][1]
        for name in levels:
            g[name] = getattr(self, name).geometry(xyz, points=x if name == "sa1" else None)  # sa1's input features are the coordinates (model.py:39)
            xyz = g[name][1]
            if name == "sa2":  # seeds = l2_xyz: the proposal layer's FPS can start as soon as they exist
                g["prop_fps"] = P.tf_sampling.farthest_point_sample(self.proposal.npoint, xyz)
            if ev is not None:  # None: the chain is being captured (GeometryGraph) -- one event after the replay stands for all
                ev[name] = torch.cuda.Event()
                ev[name].record()
        g["fp1"] = P.FPModule.geometry(g["sa3"][1], g["sa4"][1])
        g["fp2"] = P.FPModule.geometry(g["sa2"][1], g["sa3"][1])
        if ev is not None:
            ev["fp"] = torch.cuda.Event()
            ev["fp"].record()

    @staticmethod
    def _hand_over(g, main):
        for v in g.values():  # tensors born on the side stream are consumed on the main stream
            for t in (v if isinstance(v, tuple) else (v,)):
                for u in getattr(t, "tensors", tuple)() if not isinstance(t, torch.Tensor) else ():  # mlp.HalfLayout
                    u.record_stream(main)
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
                    for u in getattr(t, "_inv", None) or ():  # the grouping's inverse index rides on idx (mlp.attach_inverse)
                        u.record_stream(main)

    def geometry_ahead(self, x):
        """Every FPS / ball query / three_nn of the backbone depends on coordinates only, never on features.
        Level 1 stays on the caller's stream (the sa1 MLP needs it first); levels 2-4, both three_nn and the
        proposal layer's FPS run on a side HIP stream underneath the MLP GEMMs -- they are latency-bound chains on 8
        workgroups and leave the other CUs to the matrix work."""
        main = torch.cuda.current_stream()
        side = self._side_stream()
        g, ev = {}, {}
        g["sa1"] = self.sa1.geometry(x, points=x)
        start = torch.cuda.Event()
        start.record(main)
        with torch.cuda.stream(side):
            side.wait_event(start)
            self._geometry_chain(x, g, ev, ("sa2", "sa3", "sa4"))
        self._hand_over(g, main)
        return g, ev

    def prefetch_geometry(self, next_x):
        """Software pipelining across steps: the WHOLE coordinate-only part of an upcoming batch (all four FPS + ball queries,
        the proposal FPS, both three_nn) is launched now on a side stream, underneath this step's GEMMs -- FPS is a
        latency chain on one workgroup per scene (8 of 256 CUs), the one thing a step cannot hide from itself because
        everything waits for sa1's centres.  backbone(next_x) picks the result up; every step still computes one full
        geometry.  With GEOMETRY_GRAPHS the chain is one graph launch (GeometryGraph) and its results live in one of GEOMETRY_RING
        fixed sets of buffers: the geometry a pass picked up -- and a tape recorded on it -- is valid until GEOMETRY_RING - 1
        further batches have been prefetched (a training step consumes its tape before the next one starts).  Several batches may be in flight (two prefetch streams, used alternately): a forward-only pass is shorter
        than one geometry chain, so it wants a lookahead of two.  The input pipeline knows the next batches ahead (the
        reference prefetches them through QueueInput, run.py:121-122)."""
        pool = self.__dict__.setdefault("_prefetched", {})
        if id(next_x) in pool and pool[id(next_x)][0] is next_x and pool[id(next_x)][1] == next_x._version:
            return
        main = torch.cuda.current_stream()
        if getattr(self, "_pf_streams", None) is None:
            self._pf_streams = [self._side_stream(), torch.cuda.Stream(device=self.device, priority=SIDE_PRIORITY)]
            self._pf_turn = 0
        side = self._pf_streams[self._pf_turn]
        self._pf_turn ^= 1
        gg = self._geometry_graph(next_x, side)
        g, ev = {}, {}
        start = torch.cuda.Event()
        start.record(main)  # next_x may have been produced on the main stream; a graph's buffers may still serve the step before
        with torch.cuda.stream(side):
            side.wait_event(start)
            if gg is not None:
                g, ev = gg.replay(next_x, side)
            else:
                self._geometry_chain(next_x, g, ev, ("sa1", "sa2", "sa3", "sa4"))
        if gg is None:
            self._hand_over(g, main)
        while len(pool) >= (GEOMETRY_RING - 1 if gg is not None else 4):  # never picked up: drop the oldest
            pool.pop(next(iter(pool)))
        pool[id(next_x)] = (next_x, next_x._version, g, ev, gg, gg.generation if gg is not None else 0)

    def _geometry_graph(self, x, side):
        """The next graph of the ring for inputs shaped like x (captured on first use), or None when the chain is enqueued launch by
        launch: graphs off, the deterministic mode (its inverse indices ride on tensors as attributes), a capture under way, per-launch
        profiling events switched on."""
        if not GEOMETRY_GRAPHS or M.DETERMINISTIC or getattr(self, "_geometry_graphs_off", False) or torch.cuda.is_current_stream_capturing():
            return None
        if P.tf_sampling.PROFILE_EVENTS is not None or P.tf_grouping.PROFILE_EVENTS is not None:
            return None  # HIP events around single launches of the chain are wanted (bench.py's roofline legs): they need the launches
        key = (tuple(x.shape), x.dtype, P.HALF_GROUPS, P.ASSEMBLE_INLINE, self.proposal.npoint)
        rings = self.__dict__.setdefault("_geometry_rings", {})
        ring = rings.get(key)
        if ring is None:
            if len(rings) >= 2:  # another shape / configuration: the old graphs' buffers go
                rings.clear()
                self._geometry_evictions = getattr(self, "_geometry_evictions", 0) + 1
                if self._geometry_evictions > GEOMETRY_MAX_EVICTIONS:  # shapes keep changing: every capture is a device synchronise
                    import warnings
                    warnings.warn("VoteNetHotPath: the input shape / configuration changed %d times: the prefetched geometry chain is "
                                  "enqueued launch by launch from now on (model.GEOMETRY_GRAPHS)" % self._geometry_evictions)
                    self._geometry_graphs_off = True
                    return None
            ring = rings[key] = dict(graphs=[], turn=0)
        if len(ring["graphs"]) < GEOMETRY_RING:
            # the chain has run launch by launch before (sizes its scratch, loads its code objects): at least once per shape
            if ring.setdefault("warm", 0) < 1:
                ring["warm"] += 1
                return None
            ring["graphs"].append(GeometryGraph(self, x, side))
            return ring["graphs"][-1]
        gg = ring["graphs"][ring["turn"]]
        ring["turn"] = (ring["turn"] + 1) % GEOMETRY_RING
        if gg is getattr(self, "_geometry_current", None):  # its buffers serve the pass being enqueued (a lookahead beyond the ring)
            return None
        return gg

    def _take_prefetched(self, x):
        pf = self.__dict__.setdefault("_prefetched", {}).pop(id(x), None)
        self._geometry_current = None
        if pf is not None and pf[0] is x and pf[1] == x._version and (pf[4] is None or pf[4].generation == pf[5]):
            self._geometry_current = pf[4]
            return pf[2], pf[3]
        return None

    def backbone(self, x, tape=None, overlap=True, next_x=None):
        """model.py:35-50.  x (B,n,3) -> seeds_xyz (B,1024,3), seeds_points (B,1024,256)."""
        lv, g = self.backbone_levels(x, tape, overlap, next_x)
        l3_p2 = self.fp1.forward(lv["l3_xyz"], lv["l4_xyz"], lv["l3_p"], lv["l4_p"], tape=tape, geom=g.get("fp1"))
        seeds_p = self.fp2.forward(lv["l2_xyz"], lv["l3_xyz"], lv["l2_p"], l3_p2, tape=tape, geom=g.get("fp2"))
        return lv["l2_xyz"], seeds_p

    def backbone_levels(self, x, tape=None, overlap=True, next_x=None):
        """The four set-abstraction levels of model.py:39-45 (the part of the pass whose row counts depend on the data: the piece layout).
        -> (dict l2_xyz, l2_p, l3_xyz, l3_p, l4_xyz, l4_p; the geometry dict g with the feature-propagation taps "fp1" / "fp2" and the
        proposal layer's "prop_fps" when they were computed ahead).  The current stream has waited for all of g."""
        main = torch.cuda.current_stream()
        pf = self._take_prefetched(x)
        if pf is not None:
            g, ev = pf
            main.wait_event(ev["sa1"])
        elif overlap:
            g, ev = self.geometry_ahead(x)
        else:
            g, ev = {}, {}
        nexts = next_x if isinstance(next_x, (list, tuple)) else ([next_x] if next_x is not None else [])
        if PREFETCH_AFTER < 0:
            for nx in nexts:
                self.prefetch_geometry(nx)
        overlap = overlap or pf is not None
        self._prop_fps = g.get("prop_fps")
        def launch_prefetch(after):
            if PREFETCH_AFTER == after:
                for nx in nexts:
                    self.prefetch_geometry(nx)
        l1_xyz, l1_p, _ = self.sa1.forward(x, x, tape=tape, geom=g.get("sa1"))
        launch_prefetch(1)
        if overlap:
            main.wait_event(ev["sa2"])
        l2_xyz, l2_p, _ = self.sa2.forward(l1_xyz, l1_p, tape=tape, geom=g.get("sa2"))
        launch_prefetch(2)
        if overlap:
            main.wait_event(ev["sa3"])
        l3_xyz, l3_p, _ = self.sa3.forward(l2_xyz, l2_p, tape=tape, geom=g.get("sa3"))
        launch_prefetch(3)
        if overlap:
            main.wait_event(ev["sa4"])
        l4_xyz, l4_p, _ = self.sa4.forward(l3_xyz, l3_p, tape=tape, geom=g.get("sa4"))
        launch_prefetch(4)
        if overlap:
            main.wait_event(ev["fp"])
        return dict(l2_xyz=l2_xyz, l2_p=l2_p, l3_xyz=l3_xyz, l3_p=l3_p, l4_xyz=l4_xyz, l4_p=l4_p), g

    def vote(self, seeds_xyz, seeds_points, tape=None, seeds_copy=None):
        """model.py:53-61: votes = [seeds_xyz, seeds_points] + FC(...).  seeds_copy (b, n, 3): also receives a copy of seeds_xyz (the
        launch that builds the voting input reads those rows anyway; forward() hands the caller that copy when seeds_xyz lives in a
        geometry graph's buffers)."""
        b, n = seeds_xyz.shape[:2]
        rows = b * n
        # [seeds_xyz | seeds_points | 0]: the concat of model.py:53 and the zero padding of the ragged 259-wide input in ONE launch
        # (csrc/glue.hip) instead of cat + fill + copy; votes = x + offset and its split into xyz / features in another one
        pad = (self.voting[0].cin_pad if (self.voting[0].cin_pad and P.PAD_RAGGED_IN) else 259)
        xp = torch.empty((rows, pad), dtype=torch.float32, device=seeds_xyz.device)
        segs = [(xp[:, :3], seeds_xyz.reshape(rows, 3), None), (xp[:, 3:259], seeds_points.reshape(rows, 256), None)]
        if pad > 259:
            segs.append((xp[:, 259:], None, None))
        if seeds_copy is not None:
            segs.append((seeds_copy.view(rows, 3), seeds_xyz.reshape(rows, 3), None))
        M.row_segments(rows, segs)
        x = xp[:, :259]
        recs = []
        off, _ = P.mlp_chain_forward(self.voting, rows, ("dense", x, xp) if pad > 259 else ("dense", x), recs)
        v_xyz = torch.empty((b, n, 3), dtype=torch.float32, device=x.device)
        v_p = torch.empty((b, n, 256), dtype=torch.float32, device=x.device)
        M.row_segments(rows, [(v_xyz.view(rows, 3), x[:, :3], off[:, :3]), (v_p.view(rows, 256), x[:, 3:], off[:, 3:])])
        if tape is not None:
            tape.append(dict(op="vote", recs=recs, b=b, n=n))
        return v_xyz, v_p

    def propose(self, votes_xyz, votes_points, seeds_xyz, tape=None):
        """model.py:89-93: SA on votes with FPS on the seeds -> proposals_xyz (B,256,3), output (B,256,79)."""
        geom = None
        if getattr(self, "_prop_fps", None) is not None:  # FPS on the seeds was computed ahead (side stream)
            geom = self.proposal.geometry(votes_xyz, fps_idx=self._prop_fps, ahead=False)  # the votes exist only now
        p_xyz, p_out, _ = self.proposal.forward(votes_xyz, votes_points, sample_xyz=seeds_xyz, tape=tape, geom=geom)
        return p_xyz, p_out

    def forward(self, x, tape=None, next_x=None):
        """next_x: the batch of the NEXT call (or a list of the next few), if known: their geometry is computed underneath this
        pass (prefetch_geometry)."""
        self.store.refresh_split()  # bf16 x 3 images of the weights as they are NOW (one launch; no-op unless enable_split())
        M.arena_begin(self.device)  # one fill for all BatchNorm statistics of the pass
        try:
            lv, g = self.backbone_levels(x, tape, next_x=next_x)
            self._stamp_tape(tape)
            out = self._head_forward(lv, g.get("fp1"), g.get("fp2"), g.get("prop_fps"), tape,
                                     copy_seeds=getattr(self, "_geometry_current", None) is not None)
        finally:
            M.arena_end()
        return out

    def _stamp_tape(self, tape):
        """The geometry of this pass lives in a GeometryGraph's fixed buffers, which a later replay overwrites: the tape is stamped with
        (graph, generation) so that backward() refuses a tape whose geometry is gone."""
        gg = getattr(self, "_geometry_current", None)
        if gg is not None and tape:
            tape[0]["geometry_stamp"] = (gg, gg.generation)

    def _head_forward(self, lv, fp1_geom, fp2_geom, prop_fps, tape, copy_seeds=False):
        """Everything of the forward pass behind the four levels -- feature propagation (model.py:48-49), voting (:53-61), the proposal
        module (:89-93) -- all of it on static shapes.  copy_seeds: seeds_xyz (= sa2's centres) is handed to the caller as a copy (it
        lives in a geometry graph's buffers)."""
        l3_p2 = self.fp1.forward(lv["l3_xyz"], lv["l4_xyz"], lv["l3_p"], lv["l4_p"], tape=tape, geom=fp1_geom)
        seeds_p = self.fp2.forward(lv["l2_xyz"], lv["l3_xyz"], lv["l2_p"], l3_p2, tape=tape, geom=fp2_geom)
        seeds_xyz = lv["l2_xyz"]
        seeds_out = torch.empty_like(seeds_xyz) if copy_seeds else None
        v_xyz, v_p = self.vote(seeds_xyz, seeds_p, tape, seeds_copy=seeds_out)
        self._prop_fps = prop_fps
        p_xyz, p_out = self.propose(v_xyz, v_p, seeds_xyz, tape)
        return dict(seeds_xyz=seeds_out if seeds_out is not None else seeds_xyz, seeds_points=seeds_p, votes_xyz=v_xyz, votes_points=v_p,
                    proposals_xyz=p_xyz, proposals_output=p_out)

    # ---- inference tail: box decode (caller side, torch glue) + 3D NMS (hot path) ------
    def decode_boxes(self, proposals_xyz, proposals_output):
        """model.py:100-129 (votenet_decode_boxes): -> bboxes (B,N,8,3) in get_3d_bbox's corner order, class score (B,N)."""
        from . import loss as VL
        return VL.decode_boxes(proposals_xyz, proposals_output)

    # ---- BatchNorm moving averages (training) and the inference-mode BatchNorm built from them ----------------------
    BN_MOMENTUM = 0.9  # Tensorpack BatchNorm default (`momentum=0.9`, epsilon 1e-5); Tensorpack is not in the reference tree

    def _bn_layers(self):
        mods = [self.sa1, self.sa2, self.sa3, self.sa4, self.fp1, self.fp2, self.proposal]
        layers = [L for m in mods for L in m.mlp + (getattr(m, "mlp2", None) or [])] + list(self.voting)
        return [L for L in layers if L.bn]

    def _ema_state(self):
        """name -> (4, c) tensor [unused | unused | moving_mean | moving_var] (the layout of PendingBN.out, so that one
        multi-tensor launch updates every layer); initial values 0 / 1 as TensorFlow initialises them.  Kept OUTSIDE the
        gradient bucket: the moving averages are not trained and never all-reduced (statistics stay per replica)."""
        if getattr(self, "_ema", None) is None:
            self._ema = {}
            st = self.store
            # one flat buffer in the layout of ParamStore.bn_flat (the blocks the consumers' prologues fill): one launch updates all
            self._ema_flat = torch.zeros_like(st.bn_flat) if st.bn_flat is not None else None
            base = st.bn_flat.data_ptr() if st.bn_flat is not None else 0
            for L in self._bn_layers():
                blk = st._bn_views.get(L.name)
                if blk is not None and self._ema_flat is not None:
                    o = (blk.data_ptr() - base) // 4
                    t = self._ema_flat[o:o + 4 * L.cout].view(4, L.cout)
                else:
                    t = torch.zeros((4, L.cout), dtype=torch.float32, device=self.device)
                t[3].fill_(1.0)
                self._ema[L.name] = t
            self._ema_fac = {}
            self._ema_fac_flat = None
            self._ema_fac_by_rows = {}
            self._ema_version = 0
        return self._ema

    @staticmethod
    def _bn_records(tape):
        for t in tape:
            for r in t.get("recs", []) + t.get("recs2", []):
                if r.get("bn_out") is not None:
                    yield r

    def update_moving_averages(self, tape):
        """moving = momentum * moving + (1 - momentum) * batch for the mean and the UNBIASED batch variance (what
        tf.nn.fused_batch_norm hands to the moving-average update) of every BatchNorm layer of this forward pass: two
        multi-tensor launches per step on the (scale | shift | mean | var) blocks the consumers' prologues left behind."""
        ema = self._ema_state()
        recs = list(self._bn_records(tape))
        for r in recs:
            P.check_bn_block(r)  # the blocks read below are the ones THIS tape's pass wrote
        st = self.store
        blocks = st._bn_views
        if self._ema_flat is not None and self._ema_flat.is_cuda and len(recs) == len(blocks) and \
                all(r["bn_out"] is blocks.get(r["layer"].name) for r in recs):
            # every BatchNorm layer of the model ran once and left its block in the persistent buffer: ONE launch (csrc/glue.hip)
            key = tuple(r["rows"] for r in recs)
            if self._ema_fac_flat is None or self._ema_fac_flat[0] != key:
                # One factor tensor per rows key, kept for the life of the model: a StretchGraph captured at another batch shape has
                # this tensor's ADDRESS baked into its votenet_ema_update node, so it must never go back to the allocator while a
                # graph may still be replayed (a handful of shapes x 0.1 MB).
                f = self._ema_fac_by_rows.get(key)
                if f is None:
                    f = torch.full_like(st.bn_flat, 1.0 - self.BN_MOMENTUM)
                    base = st.bn_flat.data_ptr()
                    for r in recs:
                        c, rows = r["layer"].cout, r["rows"]
                        o = (blocks[r["layer"].name].data_ptr() - base) // 4
                        f[o + 3 * c:o + 4 * c] *= rows / max(rows - 1.0, 1.0)
                    self._ema_fac_by_rows[key] = f
                self._ema_fac_flat = (key, f)
            from . import _lib as L_
            with L_.device_guard(self.device):
                L_.check(L_.lib().votenet_ema_update(st.bn_flat.numel(), self.BN_MOMENTUM, L_.ptr(self._ema_flat), L_.ptr(st.bn_flat),
                                                     L_.ptr(self._ema_fac_flat[1]), L_.stream_ptr()))
            self._ema_version += 1
            return
        dst, src, fac = [], [], []
        for r in recs:
            name, rows = r["layer"].name, r["rows"]
            key = (name, rows)
            if key not in self._ema_fac:
                f = torch.full((4, r["layer"].cout), 1.0 - self.BN_MOMENTUM, dtype=torch.float32, device=self.device)
                f[3].mul_(rows / max(rows - 1.0, 1.0))
                self._ema_fac[key] = f
            dst.append(ema[name])
            src.append(r["bn_out"])
            fac.append(self._ema_fac[key])
        if dst:
            torch._foreach_mul_(dst, self.BN_MOMENTUM)
            torch._foreach_addcmul_(dst, src, fac)
            self._ema_version += 1

    def inference_bn(self):
        """name -> mlp.FrozenBN (scale = gamma rsqrt(moving_var + eps), shift = beta - moving_mean scale); rebuilt only when
        the parameters or the moving averages changed since the last call."""
        ema = self._ema_state()
        key = (self._ema_version, self.store.generation, self.store.flat._version)  # generation: bumped by the optimizer's raw-pointer update
        if getattr(self, "_frozen_key", None) != key:
            self._frozen = {}
            for L in self._bn_layers():
                e = ema[L.name]
                sc = L.p("gamma") * torch.rsqrt(e[3] + M.BN_EPS)
                self._frozen[L.name] = M.FrozenBN(torch.stack([sc, L.p("beta") - e[2] * sc]).contiguous())
            self._frozen_key = key
        return self._frozen

    def predict(self, x, iou_threshold=0.25, next_x=None, sync=True, batch_statistics=False):
        """Predict tower of model.py:98-139: forward -> decode -> NMS3D(bboxes, max class logit, objectness, 0.25), every
        BatchNorm in inference mode (moving averages, as the reference's BNReLU under `not is_training`): a scene's
        detections do not depend on its batch-mates.  batch_statistics=True normalises with the current batch instead
        (what a model without trained moving averages needs, e.g. random-init benchmarks).
        next_x: the batch(es) of the next call(s), as in forward().  sync=False: nms_idx stays padded on the device with its
        length in nms_count (no host synchronisation: calls pipeline)."""
        from . import tf_nms3d
        if not batch_statistics and self._ema_state() is not None and self._ema_version == 0 and not getattr(self, "_warned_ema", False):
            import warnings
            warnings.warn("VoteNetHotPath.predict: no training step has updated the BatchNorm moving averages yet (mean 0, variance 1): "
                          "pass batch_statistics=True for a freshly initialised model", stacklevel=2)
            self._warned_ema = True
        with P.frozen_bn(None if batch_statistics else self.inference_bn()):
            out = self.forward(x, next_x=next_x)
        boxes, score = self.decode_boxes(out["proposals_xyz"], out["proposals_output"])
        keep = tf_nms3d.NMS3D(boxes, score, out["proposals_output"][..., :2].contiguous(), iou_threshold, padded=not sync)
        extra = {} if sync else dict(nms_count=keep[1])
        return dict(bboxes=boxes, scores=score, nms_idx=keep if sync else keep[0], class_scores=out["proposals_output"][..., -NC:].contiguous(),
                    **extra, **out)

    # ---- backward / training --------------------------------------------------------
    def make_cotangents(self, b, seed=0):
        """Fixed upstream gradients (tests of the backward pass, scenes without ground truth): d loss / d proposals_output
        (B,256,79) and d loss / d votes_xyz (B,1024,3).  Training uses the loss graph instead (train_step(gt=...))."""
        g = torch.Generator(device="cpu").manual_seed(1234 + seed)
        n_seed = self.sa2.npoint
        return dict(proposals_output=(torch.randn(b, PROPOSAL_NUM, PROPOSAL_OUT, generator=g) / (b * PROPOSAL_NUM)).to(self.device),
                    votes_xyz=(torch.randn(b, n_seed, 3, generator=g) / (b * n_seed)).to(self.device))

    def backward(self, tape, cot):
        """Reverse sweep over the tape of forward(); parameter gradients accumulate into store.grad."""
        self.check_tape(tape)
        M.arena_begin(self.device)  # one fill for all BatchNorm-backward reductions of the pass
        if self.overlap_wgrad:
            if self._wgrad_stream is None:
                self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=WGRAD_PRIORITY)
            P.WGRAD_STREAM = self._wgrad_stream
        try:
            self._backward(tape, cot)
            P.wgrad_join()  # the optimizer (and the next pass's arena fill) come after every weight gradient
        finally:
            P.WGRAD_STREAM = None
            M.arena_end()

    @staticmethod
    def check_tape(tape):
        """A tape recorded on prefetched geometry reads a GeometryGraph's buffers (idx, pts_cnt, the compact rows, fps_idx, the piece
        layouts): valid until that graph replays for another batch -- GEOMETRY_RING - 1 further prefetches.  Raises instead of
        differentiating through another batch's neighbour lists."""
        stamp = tape[0].get("geometry_stamp") if tape else None
        if stamp is not None and stamp[0].generation != stamp[1]:
            raise M.L.VotenetError("backward(): the geometry buffers this tape was recorded on have been overwritten by %d later prefetch(es) "
                                   "(model.GeometryGraph; a tape on prefetched geometry is valid until GEOMETRY_RING - 1 = %d further "
                                   "batches have been prefetched)" % (stamp[0].generation - stamp[1], GEOMETRY_RING - 1))

    def _backward(self, tape, cot):
        d_l2p, d_l3p, d_l4p = self._head_backward(tape[4:8], cot)
        self._levels_backward(tape[:4], d_l2p, d_l3p, d_l4p)

    def _head_backward(self, tail, cot, cut=None):
        """The backward pass of _head_forward (proposal, voting, fp2, fp1): -> the gradients of the level outputs d_l2p, d_l3p, d_l4p.
        cut: called where a module's backward ends (StretchGraph cuts its segments there)."""
        fp1, fp2, vote, prop = tail
        cut = cut if cut is not None else (lambda: None)
        # proposal layer: gradients reach the vote features AND the vote xyz (grouped xyz, gathered centres)
        d_vp, d_vx = self.proposal.backward(prop, cot["proposals_output"], need_feat_grad=True, need_xyz_grad=True)
        P.wgrad_flush()  # the module's weight gradients go to their stream together, underneath the next module's chain
        cut()
        if cot.get("proposals_xyz") is not None:  # proposals_xyz = gather(votes_xyz, fps_idx), utils.py:42-47: accumulated in place
            d_vx = P.tf_sampling.gather_point_grad_raw(d_vx.shape[1], prop["fps_idx"], cot["proposals_xyz"], into=d_vx.contiguous())
        # voting: votes = x + FC(x), x = [seeds_xyz, seeds_points].  d_votes = [d_vx (+ the loss's pull on the votes) | d_vp | 0]: the
        # concat, the sum and the zero padding of the ragged 259-wide layer in ONE launch (csrc/glue.hip)
        b, n = vote["b"], vote["n"]
        rows = b * n
        last = self.voting[-1]
        padw = last.cout_pad if last.cout_pad else 259
        d_votes_p = torch.empty((rows, padw), dtype=torch.float32, device=d_vp.device)
        cv = cot.get("votes_xyz")
        segs = [(d_votes_p[:, :3], d_vx.reshape(rows, 3), cv.reshape(rows, 3) if cv is not None else None),
                (d_votes_p[:, 3:259], d_vp.reshape(rows, 256), None)]
        if padw > 259:
            segs.append((d_votes_p[:, 259:], None, None))
        M.row_segments(rows, segs)
        d_votes = d_votes_p[:, :259]
        d_in = P.mlp_chain_backward(vote["recs"], d_votes, "plain", need_input_grad=True, g_padded=d_votes_p if padw > 259 else None)
        P.wgrad_flush()
        cut()
        # d x = d_votes + d_in; only its feature columns go on (the seeds' coordinates carry no gradient in the backbone)
        d_seeds_p = torch.empty((b, n, 256), dtype=torch.float32, device=d_vp.device)
        M.row_segments(rows, [(d_seeds_p.view(rows, 256), d_votes[:, 3:], d_in[:, 3:259])])
        # feature propagation
        d_l2p, d_l3p2 = self.fp2.backward(fp2, d_seeds_p)
        P.wgrad_flush()
        cut()
        d_l3p, d_l4p = self.fp1.backward(fp1, d_l3p2)
        P.wgrad_flush()
        return d_l2p, d_l3p, d_l4p

    def _levels_backward(self, levels, d_l2p, d_l3p, d_l4p):
        """The backward pass of the four levels, sa4 ... sa1 (xyz carries no gradient in the backbone: the cloud is the input)."""
        sa1, sa2, sa3, sa4 = levels
        self._grams_early(dict(sa1=sa1, sa2=sa2, sa3=sa3, sa4=sa4))
        g3, _ = self.sa4.backward(sa4, d_l4p)
        P.wgrad_flush()
        d_l3p = P.add_rows(d_l3p, g3)
        g2, _ = self.sa3.backward(sa3, d_l3p)
        P.wgrad_flush()
        d_l2p = P.add_rows(d_l2p, g2)
        # every gradient of sa3 ... proposal (the tail of the flat bucket) is enqueued: its all-reduce runs on the
        # communication stream underneath the backward pass of sa2 and sa1 (dp.GradSync; a no-op on one GPU)
        if getattr(self, "_gsync", None) is not None:
            self._gsync.start_tail([P.WGRAD_STREAM])
        # inline_wgrad_tail (experiment, default off): the weight gradients of the two largest modules on the MAIN stream -- by the time
        # their backward runs the next batch's sampling kernel has finished, and pairs of GPU-filling kernels gain nothing from two streams
        keep_stream = P.WGRAD_STREAM
        if getattr(self, "inline_wgrad_tail", INLINE_WGRAD_TAIL):
            P.WGRAD_STREAM = None
        try:
            g1, _ = self.sa2.backward(sa2, d_l2p)
            P.wgrad_flush()
            P.wgrad_fine(True)  # the last module: nothing follows to hide its weight gradients under, so they start layer by layer
            try:
                self.sa1.backward(sa1, g1, need_feat_grad=False)
            finally:
                P.wgrad_fine(False)
        finally:
            P.WGRAD_STREAM = keep_stream

    # ---- the static stretch of a train step as one HIP graph (StretchGraph) -------------------------------------------------------
    def _stretch_eligible(self, x, cot, gt):
        # (per-launch events around the GEMMs -- mlp.PROFILE_EVENTS -- need the launches; events around the sampling / ball-query launches
        # -- bench.py's roofline legs -- concern the geometry chain only: a captured stretch still replays, see _train_step_stretch)
        return bool(STRETCH_GRAPH and gt is not None and cot is None and x.is_cuda and not M.DETERMINISTIC and PREFETCH_AFTER < 5
                    and M.PROFILE_EVENTS is None and P._FROZEN.table is None and not getattr(self, "_stretch_off", False)
                    and not torch.cuda.is_current_stream_capturing())

    @staticmethod
    def _stretch_inputs(lv, g):
        """name -> tensor: everything the captured segments read that lives at an address of this step's making (the ground truth is
        not among them: the loss runs as launches between two segments)."""
        ins = dict(lv)
        for name in ("fp1", "fp2"):
            idx, w = g[name]
            ins[name + "_idx"], ins[name + "_w"] = idx, w
            inv = getattr(idx, "_inv", None)
            if inv is not None:
                ins[name + "_inv0"], ins[name + "_inv1"] = inv
        ins["prop_fps"] = g["prop_fps"]
        return ins

    def _stretch_body(self, I, tape_levels, wgrad_stream, gt=None, cut=None, loss_hook=None, copy_seeds=False):
        """The stretch on the tensors of I (StretchGraph captures this on its fixed buffers; the first step of a shape runs it eagerly):
        forward head, moving averages, loss (on gt, or through loss_hook(out) -> (losses, cotangents) when captured), backward head.
        -> (out, losses, (d_l2p, d_l3p, d_l4p))."""
        from . import loss as VL

        def geom(name):
            idx = I[name + "_idx"]
            if name + "_inv0" in I:
                idx._inv = (I[name + "_inv0"], I[name + "_inv1"])
            return idx, I[name + "_w"]
        tail = []
        lv = {k: I[k] for k in ("l2_xyz", "l2_p", "l3_xyz", "l3_p", "l4_xyz", "l4_p")}
        out = self._head_forward(lv, geom("fp1"), geom("fp2"), I["prop_fps"], tail, copy_seeds=copy_seeds)
        self.update_moving_averages(list(tape_levels) + tail)
        losses, cot = loss_hook(out) if loss_hook is not None else VL.votenet_loss(out, gt)
        P.WGRAD_STREAM = wgrad_stream
        try:
            grads = self._head_backward(tail, cot, cut=cut)
            if cut is None:
                P.wgrad_join()
        finally:
            P.WGRAD_STREAM = None
        self._stretch_tail = tail  # (kept: the records own the tensors the captured kernels read)
        return out, losses, grads

    def _train_step_stretch(self, x, gt, tape, next_x):
        """train_step's middle with the stretch replayed as a graph: levels forward (launches: their row counts are the data's) ->
        ONE copy launch + ONE graph launch -> levels backward (launches)."""
        self.store.refresh_split()
        lv, g = self.backbone_levels(x, tape, next_x=next_x)
        self._stamp_tape(tape)
        self._grams_ahead(tape)
        if not all(k in g for k in ("fp1", "fp2", "prop_fps")):  # geometry computed without the taps (overlap off): the launch path
            g = dict(g)
            g.setdefault("fp1", P.FPModule.geometry(lv["l3_xyz"], lv["l4_xyz"]))
            g.setdefault("fp2", P.FPModule.geometry(lv["l2_xyz"], lv["l3_xyz"]))
            g.setdefault("prop_fps", P.tf_sampling.farthest_point_sample(self.proposal.npoint, lv["l2_xyz"]))
        ins = self._stretch_inputs(lv, g)
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in ins.items()) + (P.HALF_GROUPS, P.ASSEMBLE_FIRST, P.ASSEMBLE_INLINE, P.POOL_GRAM_BACKWARD, P.ASSEMBLED_DECOMPOSED,
                                                                             self.overlap_wgrad, STRETCH_SEGMENTS, M.SPLIT_K, M.COEF_TAIL, P.WGRAD_BATCH, P.POOL_IN_EPILOGUE,
                                                                             self.store.split, bool(getattr(self, "inline_wgrad_tail", INLINE_WGRAD_TAIL)), M.FORWARD_H2, M.ADHOC_H2, M.CONFIG_EPOCH)
        graphs = self.__dict__.setdefault("_stretch_graphs", {})
        sg = graphs.get(key)
        self._gsync.begin()
        if sg is None:
            dkey = key[:-1]  # what the stretch asks of the arena does not depend on library switches: one measurement serves
            demand = self.__dict__.setdefault("_stretch_demand", {}).get(dkey)
            geometry_events = P.tf_sampling.PROFILE_EVENTS is not None or P.tf_grouping.PROFILE_EVENTS is not None
            if demand is None or geometry_events:
                # first step of this shape: the stretch launch by launch, measuring what it asks of the arena (also while somebody records
                # events around the geometry launches: the proposal module's ball query sits in the stretch -- no capture then)
                a = M._StatsArena
                off0, want0 = a.off, a.want32
                self.check_tape(tape)
                if self.overlap_wgrad and self._wgrad_stream is None:
                    self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=WGRAD_PRIORITY)
                # (launch by launch on the live inputs: seeds_xyz would alias sa2's centres inside a GeometryGraph buffer that a later
                # prefetch overwrites -- handed out as a copy, as forward() does)
                out, self.last_losses, grads = self._stretch_body(ins, tape, self._wgrad_stream if self.overlap_wgrad else None, gt=gt,
                                                                  copy_seeds=getattr(self, "_geometry_current", None) is not None)
                self._stretch_demand[dkey] = (a.off - off0, a.want32 - want0)
                self._backward_levels_pass(tape, grads)
                return out
            while len(graphs) >= STRETCH_MAX_GRAPHS:  # the least recently replayed graph (and its pool) goes
                graphs.pop(min(graphs, key=lambda k: graphs[k].last_used))
            sg = graphs[key] = StretchGraph(self, ins, tape, demand)
        self._stretch_clock = getattr(self, "_stretch_clock", 0) + 1
        sg.last_used = self._stretch_clock
        self.store._fresh_wait()
        if self.overlap_wgrad and self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=WGRAD_PRIORITY)
        out, self.last_losses, grads = sg.replay(ins, gt, self._wgrad_stream if self.overlap_wgrad else None)  # (a vector of the graph's ring: rewritten eight replays later)
        self._ema_version += 1  # (the replay ran votenet_ema_update: inference_bn() must not serve a table built before it)
        self._backward_levels_pass(tape, grads)
        return out

    def _grams_early(self, levels):
        """GRAM_EARLY: the Gram matrices of the named levels (forward data only) go to the weight-gradient stream at the START of the
        levels' backward pass instead of beside their own level's backward GEMMs.  The step's tail is sa1's backward, where the chain
        (arg-max scatter, dense input gradient, narrow input gradient: 0.31 ms alone) and the weight-gradient stream (Gram matrix, sparse
        gather, narrow weight gradient: 0.28 ms alone) run beside each other at 0.45 ms: sa1's matrix ahead takes a third of that
        stream's tail away and runs beside sa4's / sa3's smaller kernels instead."""
        if not GRAM_EARLY or M.DETERMINISTIC or P.WGRAD_STREAM is None:
            return
        todo = [levels[n]["recs"][-1] for n in GRAM_EARLY if n in levels]
        todo = [r for r in todo if r.get("gram_form") and r.get("in_affine") is not None and "gram_ahead" not in r]
        if not todo:
            return

        def run():
            for r in todo:
                r["gram_ahead"] = M.gram(r["x"], r["in_affine"][:2], r["in_relu"], half=r.get("half"))
        P.on_wgrad_stream(run, *[r["x"] for r in todo])
        P.wgrad_flush()

    def _grams_ahead(self, tape):
        """The Gram matrix a^T a of a pooled layer's input (pool_bwd.hip) depends on forward data only.  Launched here -- behind the
        levels' forward pass, on the weight-gradient stream -- the four of sa1..sa4 (0.27 ms of kernels) run under the static stretch,
        whose own kernels leave most of the GPU idle, instead of beside the backward GEMMs of their levels."""
        if not GRAMS_AHEAD or M.DETERMINISTIC or not self.overlap_wgrad or not torch.cuda.is_available():
            return
        todo = [t["recs"][-1] for t in tape[:4] if t.get("op") == "sa" and t["recs"][-1].get("gram_form") and t["recs"][-1].get("in_affine") is not None]
        if not todo:
            return
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=WGRAD_PRIORITY)
        prev = P.WGRAD_STREAM
        P.WGRAD_STREAM = self._wgrad_stream

        def run():
            for r in todo:
                r["gram_ahead"] = M.gram(r["x"], r["in_affine"][:2], r["in_relu"], half=r.get("half"))
        try:
            P.on_wgrad_stream(run, *[r["x"] for r in todo])
        finally:
            P.WGRAD_STREAM = prev

    def _backward_levels_pass(self, tape, grads):
        """backward() for the four levels only (the head's gradients given)."""
        self.check_tape(tape)
        M.arena_begin(self.device)
        if self.overlap_wgrad:
            if self._wgrad_stream is None:
                self._wgrad_stream = torch.cuda.Stream(device=self.device, priority=WGRAD_PRIORITY)
            P.WGRAD_STREAM = self._wgrad_stream
        try:
            self._levels_backward(tape[:4], *grads)
            P.wgrad_join()
        finally:
            P.WGRAD_STREAM = None
            M.arena_end()

    def drop_graphs(self):
        """Forget every captured graph (geometry rings, stretch graphs): after a configuration change the captures do not key on
        (library debug switches, hand-edited module state)."""
        self.__dict__.pop("_stretch_graphs", None)
        self.__dict__.pop("_stretch_demand", None)
        self.__dict__.pop("_geometry_rings", None)
        self.__dict__.setdefault("_prefetched", {}).clear()
        self._geometry_current = None

    def init_optimizer(self, lr=1e-3):
        s = self.store
        import math
        base = s.flat.data_ptr()
        seg = []
        for name, shape, _ in s._specs:  # [start, end) of every tensor inside the flat bucket
            a = (s.views[name].data_ptr() - base) // 4
            seg += [a, a + math.prod(shape)]
        self._seg = torch.tensor(seg, dtype=torch.int64, device=self.device)
        self._sumsq = torch.zeros(M.SUMSQ_SLICES * (len(seg) // 2), dtype=torch.float32, device=self.device)  # ordered partials per tensor
        self._m = torch.zeros_like(s.flat)
        self._v = torch.zeros_like(s.flat)
        self._step = 0
        self._lr = lr

    def train_step(self, x, cot=None, world=1, gt=None, next_x=None):
        """forward + loss + backward + (world>1: the RCCL all-reduce of the flat gradient bucket, its tail overlapped with the
        backward pass of sa2 / sa1: dp.GradSync) + clip/Adam.
        gt: ground truth on the device (loss.gt_to_device): the reference's total cost (model.py:228) drives the backward
        pass, its components are left in self.last_losses (device, loss.NAMES).  cot: fixed cotangents instead (tests)."""
        if not hasattr(self, "_seg"):
            self.init_optimizer()
        self.store.grad.zero_()
        # every W^T of the backward pass's input-gradient GEMMs: one launch on the geometry stream, under the sa1 FPS
        self.store.refresh_transposes(self._side_stream())
        tape = []
        # ONE zero fill for every accumulator and scatter target of the step (mlp._StatsArena): forward() and backward() join it
        M.arena_begin(self.device, whole_step=True)
        try:
            if getattr(self, "_gsync", None) is None:
                self._gsync = dp.GradSync(self.store, self.store.offset_of("sa3/"))
            if self._stretch_eligible(x, cot, gt):
                out = self._train_step_stretch(x, gt, tape, next_x)
            else:
                out = self.forward(x, tape, next_x=next_x)
                self._grams_ahead(tape)
                self.update_moving_averages(tape)
                if gt is not None:
                    from . import loss as VL
                    self.last_losses, cot = VL.votenet_loss(out, gt)
                self._gsync.begin()
                if PREFETCH_AFTER >= 5:  # the next batch's geometry chain under the BACKWARD pass
                    for nx in (next_x if isinstance(next_x, (list, tuple)) else ([next_x] if next_x is not None else [])):
                        self.prefetch_geometry(nx)
                self.backward(tape, cot)            # world > 1: starts the all-reduce of the bucket's tail after sa3's backward
        finally:
            M.arena_end()
        self.store.invalidate_transposes()      # the optimizer changes W
        gscale = self._gsync.finish()           # head all-reduce + wait for both; 1/world goes to the optimizer
        self._step += 1
        M.clip_adam(self._seg, self._sumsq, self.store.flat, self.store.grad, self._m, self._v, self._lr, self._step,
                    grad_scale=gscale)
        return out
