"""PointNet++ SA / FP layers of VoteNet on the HIP hot path -- mirror of the reference's utils.py.

    sample_and_group     utils.py:25-61
    pointnet_sa_module   utils.py:93-158   (group_all=False, pooling='max'; knn=False is the only
                                            configuration model.py uses, knn=True is served too)
    pointnet_sa_module_msg utils.py:161-201 (multi-scale grouping; never reached by model.py)
    pointnet_fp_module   utils.py:266-294

The reference builds these from ~6 TF graph nodes per MLP layer on a materialised
(B,m,K,3+C) tensor.  Here a layer is ONE fused launch (votenet_mlp_linear): the grouped
tensor is never written, BN+ReLU of layer l is applied while layer l+1 loads its input, and
BatchNorm batch statistics come out of the GEMM epilogue.  torch is used for buffers only;
the backward pass is explicit (tape + hand-written backward kernels), not torch.autograd.

Parameters live in one flat fp32 bucket (ParamStore) so that data-parallel training needs a
single RCCL all-reduce over one contiguous gradient buffer per step.
"""
import math
import threading

import torch

from . import mlp as M
from . import tf_grouping, tf_interpolate, tf_sampling


PAD_RAGGED = True  # False: ragged plain layers go through the generic bounds-checked GEMM (A/B, tests)
NARROW_FIRST = True  # leaf SA module with 3 + c <= 8 grouped channels: the first layer's output is rebuilt from 8 floats per row, never stored (csrc/narrow.hip)
PAD_RAGGED_IN = True  # ragged INPUT widths (voting's 259) padded as well (Layer.cin_pad); False: the bounds-checked GEMMs (A/B)
ASSEMBLE_FIRST = True  # other SA modules: the first layer's output z0 = P[idx] + dxyz W[0:3] is rebuilt inside the kernels that consume it, never stored (csrc/assemble.hip)
ASSEMBLE_INLINE = True  # also where the geo records were not computed ahead with the geometry (they are built in place)
HALF_GROUPS = True     # modules with nsample = 64 run their grouped MLP on the PIECE layout (csrc/half.hip): a ball keeps the 16-row pieces that hold a real neighbour, slot 0 stands for the dropped copies -- same results up to summation order, 28-68 % fewer grouped rows on room scenes
FUSE_BN_REDUCE = True  # dense input-gradient GEMMs reduce the BatchNorm backward of the layer below in their epilogue


# --------------------------------------------------------------------------- parameters
class ParamStore:
    """All trainable tensors as views into one flat fp32 buffer (and one flat gradient buffer)."""

    def __init__(self, device):
        self.device = device
        self._specs = []  # (name, shape, init)
        self.flat = None
        self.grad = None
        self.views = {}
        self.gviews = {}
        self._tspecs = []   # (tensor name, row_lo, row_hi): W^T blocks wanted by the input-gradient GEMMs
        self._tviews = {}
        self._tflat = self._ttable = None
        self.t_event = None  # recorded after the transposes of the current step; None = stale
        self.split = False      # enable_split(): bf16 x 3 images of the weight matrices for the fused GEMMs (mlp.SplitImages)
        self._split_flat = self._split_t = None
        # every derived copy of the parameters (bf16 x 3 images, transposes, padded copies, inference BatchNorm tables) belongs to a
        # GENERATION of the bucket: params_changed() opens a new one (the optimizer does; so must anyone who writes the parameters in
        # place), and a copy of an older generation is rebuilt before its next use instead of being multiplied by silently
        self.generation = 0
        self._split_gen = -1
        # the (scale | shift | mean | var) blocks the BatchNorm consumers leave behind, one per BatchNorm'ed layer, in ONE persistent
        # buffer (bn_block): the moving-average update of a train step is then one launch over it (votenet_ema_update)
        self._split_rows = []  # (tensor name, row_lo, row_hi): row blocks of a matrix that are GEMM operands of their own (W[3:] of an SA first layer)
        self._bn_specs = []   # (layer name, cout)
        self.bn_flat = None
        self._bn_views = {}
        self._bn_pass = {}     # layer name -> number of the forward pass that wrote its persistent block last (check_bn_block)
        self._bn_passes = 0

    def want_transpose(self, name, lo=0, hi=None):
        """Register rows [lo, hi) of the 2-D tensor `name`: transposed() then serves its transpose from one bucket that
        refresh_transposes() fills with a single launch per step."""
        self._tspecs.append((name, lo, hi, "T", 0))

    def want_padded(self, name, pad_to):
        """Register copies of the tensor `name` whose ragged width is padded to `pad_to` columns (16-byte aligned rows for the
        fast GEMMs): padded(name) = [W | 0] (rows x pad_to) and padded(name, True) = [W | 0]^T (pad_to x rows), refreshed by
        the same launch as the transposes.  A 1-D tensor (bias) is a 1 x n matrix."""
        self._tspecs.append((name, 0, None, "P", pad_to))
        self._tspecs.append((name, 0, None, "PT", pad_to))  # skipped for 1-D tensors when the table is built

    def want_padded_rows(self, name, pad_to):
        """Register copies of the 2-D tensor `name` with its ragged ROW count padded to `pad_to`: padded_rows(name) = [W ; 0]
        (pad_to x cols) for a GEMM whose input carries zero columns up to pad_to, padded_rows(name, True) = [W ; 0]^T
        (cols x pad_to) for the input-gradient GEMM.  Same launch as the transposes."""
        self._tspecs.append((name, 0, None, "R", pad_to))
        self._tspecs.append((name, 0, None, "RT", pad_to))

    def enable_split(self, on=True):
        """The owner promises to call refresh_split() whenever the parameters changed before the next GEMM reads them
        (VoteNetHotPath.forward does, at its start; refresh_transposes() covers the transposed / padded copies)."""
        self.split = bool(on)
        if not on:
            for im in (self._split_flat, self._split_t, getattr(self, "_split_tf", None)):
                if im is not None:
                    im.close()
            self._split_flat = self._split_t = self._split_tf = None

    def refresh_split(self):
        """One launch on the current stream: the images of every eligible 2-D tensor of the bucket."""
        if not self.split or self.flat is None or not self.flat.is_cuda:
            return
        if self._split_flat is None:
            mats = [v for v in self.views.values() if v.dim() == 2]
            mats += [self.views[name][lo:hi] for name, lo, hi in self._split_rows]  # e.g. W[3:]: the per-point GEMM P = feat W[3:]
            self._split_flat = M.SplitImages(mats, pieces=2 if (M.FORWARD_H2 and not M.SPLIT_K) else 3)
        self._split_flat.refresh()
        self._split_gen = self.generation

    def rebuild_split(self):
        """Forget the images of the forward matrices: the next refresh_split() builds them in the form mlp.FORWARD_H2 names NOW (fp16 x 2
        or bf16 x 3).  Captured graphs hold the old images' addresses: the configuration epoch their keys carry moves on."""
        for name in ("_split_flat", "_split_t", "_split_tf"):  # (the padded forward copies change form with the flat images)
            im = getattr(self, name, None)
            if im is not None:
                im.close()
            setattr(self, name, None)
        self._split_gen = -1
        self.t_event = None  # the copies' images are rebuilt with the next refresh_transposes()
        M.CONFIG_EPOCH += 1

    def ensure_split(self):
        """Images of the current generation before a GEMM reads them: a no-op inside a pass (forward() refreshed them at its start),
        one launch when a module is driven directly after the parameters changed (SAModule.forward after an optimizer step)."""
        if self.split and self._split_gen != self.generation:
            self.refresh_split()

    def refresh_transposes(self, stream=None):
        """One launch for every registered W^T block / padded copy, on `stream` (default: the current one); transposed() /
        padded() make the current stream wait for it.  Call again whenever the parameters changed (once per training step)."""
        if not self._tspecs:
            return
        if self._tflat is None:
            base = self.flat.data_ptr()
            table, total, shapes = [], 0, {}
            for name, lo, hi, kind, pad in self._tspecs:
                v = self.views[name]
                v2 = v if v.dim() == 2 else v.view(1, -1)
                if kind == "PT" and v.dim() != 2:
                    continue
                hi_ = v2.shape[0] if hi is None else hi
                rows, cols = hi_ - lo, v2.shape[1]
                src = (v.data_ptr() - base) // 4 + lo * cols
                if kind == "T":      # (cols x rows), leading dimension rows
                    table += [src, total, rows, cols, rows, 1]
                    shp = (cols, rows)
                elif kind == "R":    # (pad x cols): rows >= the tensor's stay zero
                    table += [src, total, rows, cols, cols, 0]
                    shp = (pad, cols)
                elif kind == "RT":   # (cols x pad): the transpose with leading dimension pad
                    table += [src, total, rows, cols, pad, 1]
                    shp = (cols, pad)
                elif kind == "P":    # (rows x pad), zero padding
                    table += [src, total, rows, cols, pad, 0]
                    shp = (rows, pad) if v.dim() == 2 else (pad,)
                else:                # "PT": (pad x rows): rows >= cols of the transpose stay zero
                    table += [src, total, rows, cols, rows, 1]
                    shp = (pad, rows)
                shapes[(name, lo, hi, kind)] = (total, shp)
                n_el = 1
                for d in shp:
                    n_el *= d
                total += (n_el + 3) // 4 * 4
            self._tflat = torch.zeros(total, dtype=torch.float32, device=self.device)  # the padding is never written: stays 0
            self._ttable = torch.tensor(table, dtype=torch.int64, device=self.device)
            self._nseg = len(table) // 6
            for key, (off, shp) in shapes.items():
                n_el = 1
                for d in shp:
                    n_el *= d
                self._tviews[key] = self._tflat[off:off + n_el].view(shp)
        from . import _lib as L_
        cur = torch.cuda.current_stream()
        st = stream if stream is not None else cur
        if st is not cur:
            st.wait_stream(cur)  # the optimizer's update of the parameters is on the caller's stream
        with M.L.device_guard(self.device), torch.cuda.stream(st):
            L_.check(L_.lib().votenet_transpose_segments(self._nseg, L_.ptr(self._ttable), L_.ptr(self.flat),
                                                         L_.ptr(self._tflat), L_.stream_ptr()))
            if self.split:  # the images of the copies, behind the copies on the same stream
                if self._split_t is None:
                    # transposes ("T", "RT", "PT") are read by the backward GEMMs (gradient operands: bf16 x 3); the zero-padded copies
                    # ("R", "P") stand in for W itself in FORWARD GEMMs of ragged widths (voting 259 -> 288 rows, mlp2 79 -> 128 columns):
                    # fp16 x 2 like every other forward matrix (mlp.FORWARD_H2)
                    fwd = [v for k, v in self._tviews.items() if v.dim() == 2 and k[3] in ("R", "P")]
                    bwd = [v for k, v in self._tviews.items() if v.dim() == 2 and k[3] not in ("R", "P")]
                    h2 = M.FORWARD_H2 and not M.SPLIT_K
                    self._split_t = M.SplitImages(bwd + ([] if h2 else fwd))
                    self._split_tf = M.SplitImages(fwd, pieces=2) if h2 else None
                self._split_t.refresh()
                if self._split_tf is not None:
                    self._split_tf.refresh()
            self.t_event = torch.cuda.Event()
            self.t_event.record(st)
        self._t_waited = False

    def params_changed(self):
        """The parameters were written (optimizer step, manual edit): every derived copy is stale from here on."""
        self.generation += 1
        self.t_event = None

    def invalidate_transposes(self):
        self.params_changed()

    def _fresh(self, key):
        if self.t_event is None and self._tspecs and self.flat is not None and self.flat.is_cuda:
            self.refresh_transposes()  # first use since the parameters last changed (inference: once)
        v = self._tviews.get(key) if self.t_event is not None else None
        if v is None:
            return None
        if not self._t_waited:
            torch.cuda.current_stream().wait_event(self.t_event)
            self._t_waited = True
        return v

    def _fresh_wait(self):
        """The current stream waits for this step's derived copies now (a captured stretch must not contain that wait: the event is
        recorded outside the capture)."""
        if self.t_event is None and self._tspecs and self.flat is not None and self.flat.is_cuda:
            self.refresh_transposes()
        if self.t_event is not None and not self._t_waited:
            torch.cuda.current_stream().wait_event(self.t_event)
            self._t_waited = True

    def transposed(self, name, lo=0, hi=None):
        """W[lo:hi]^T, contiguous: from the per-step bucket when registered and fresh, else an ad-hoc copy."""
        v = self._fresh((name, lo, hi, "T"))
        return v if v is not None else self.views[name][lo:hi].t().contiguous()

    def padded(self, name, pad_to, transpose=False):
        """[W | 0] (rows x pad_to), or its transpose: from the per-step bucket when registered and fresh, else an ad-hoc copy."""
        v = self._fresh((name, 0, None, "PT" if transpose else "P"))
        if v is not None:
            return v
        w = self.views[name]
        p = torch.nn.functional.pad(w, (0, pad_to - w.shape[-1]))
        return p.t().contiguous() if transpose else p

    def padded_rows(self, name, pad_to, transpose=False):
        """[W ; 0] (pad_to x cols), or its transpose (cols x pad_to): from the per-step bucket when registered and fresh, else ad hoc."""
        v = self._fresh((name, 0, None, "RT" if transpose else "R"))
        if v is not None:
            return v
        w = self.views[name]
        p = torch.nn.functional.pad(w, (0, 0, 0, pad_to - w.shape[0]))
        return p.t().contiguous() if transpose else p

    def declare(self, name, shape, init):
        self._specs.append((name, tuple(shape), init))

    def bn_block_written(self, name):
        """A forward pass is about to (re)write layer `name`'s persistent BatchNorm block: -> the stamp its record keeps."""
        self._bn_passes += 1
        self._bn_pass[name] = self._bn_passes
        return self._bn_passes

    def materialize(self, seed=0):
        # every view starts on a 16-byte boundary (float4 loads of scale/shift/W rows)
        offs, total = [], 0
        for _, shape, _ in self._specs:
            offs.append(total)
            total += (math.prod(shape) + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        nbn = sum(4 * c for _, c in self._bn_specs)
        self.bn_flat = torch.zeros(max(nbn, 1), dtype=torch.float32, device=self.device)
        o = 0
        for name, c in self._bn_specs:
            self._bn_views[name] = self.bn_flat[o:o + 4 * c].view(4, c)
            o += 4 * c
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for (name, shape, init), off in zip(self._specs, offs):
            n = math.prod(shape)
            v = self.flat[off:off + n].view(shape)
            if init == "he":
                v.copy_((torch.randn(shape, generator=gen) * math.sqrt(2.0 / shape[0])).to(self.device))
            elif init == "ones":
                v.fill_(1.0)
            self.views[name] = v
            self.gviews[name] = self.grad[off:off + n].view(shape)
        return self

    def offset_of(self, prefix):
        """Element offset inside the flat bucket of the first tensor whose name starts with `prefix` (declaration order)."""
        for name, _, _ in self._specs:
            if name.startswith(prefix):
                return self.views[name].storage_offset()
        raise KeyError(prefix)

    def __getitem__(self, name):
        return self.views[name]

    def g(self, name):
        return self.gviews[name]

    def numel(self):
        return sum(math.prod(s) for _, s, _ in self._specs)


class Layer:
    """One 1x1 Conv2D / FullyConnected (+ BatchNorm + ReLU): utils.py:126-127, model.py:56."""

    def __init__(self, store, name, cin, cout, bn=True, relu=True):
        self.store, self.name, self.cin, self.cout, self.bn, self.relu = store, name, cin, cout, bn, relu
        store.declare(name + "/W", (cin, cout), "he")
        store.want_transpose(name + "/W")
        store.declare(name + "/b", (cout,), "zeros")
        if bn:
            store.declare(name + "/gamma", (cout,), "ones")
            store.declare(name + "/beta", (cout,), "zeros")
            store._bn_specs.append((name, cout))
        # a plain (no BatchNorm) layer with a ragged width -- voting's 259, mlp2's 79 -- runs on copies padded to a multiple of 64
        # columns: the fast GEMMs want 16-byte aligned rows (the generic kernel they replace ran at 19 TFLOP/s)
        self.cout_pad = 0
        if PAD_RAGGED and not bn and cout % 64 != 0 and cin % 32 == 0:
            self.cout_pad = (cout + 63) // 64 * 64
            store.want_padded(name + "/W", self.cout_pad)
            store.want_padded(name + "/b", self.cout_pad)
        # a BatchNorm'ed layer with a ragged INPUT width -- voting's 259 -- takes its dense input zero-padded to a multiple of 64
        # columns against [W ; 0]: forward, weight-gradient and input-gradient GEMMs all leave the bounds-checked kernels
        self.cin_pad = 0
        if PAD_RAGGED and bn and cin % 32 != 0 and cin > 64 and cout % 64 == 0:
            self.cin_pad = (cin + 63) // 64 * 64
            store.want_padded_rows(name + "/W", self.cin_pad)

    def p(self, k):
        st = self.store
        if st.split and st._split_gen != st.generation:
            st.refresh_split()  # a weight is about to be handed to a GEMM: its image must be of this generation
        return st[self.name + "/" + k]

    def wT(self, lo=0, hi=None):
        return self.store.transposed(self.name + "/W", lo, hi)

    def gp(self, k):
        return self.store.g(self.name + "/" + k)


def make_mlp(store, scope, cin, widths, prefix="conv", last_plain=False):
    layers = []
    for i, co in enumerate(widths):
        plain = last_plain and i == len(widths) - 1
        layers.append(Layer(store, "%s/%s%d" % (scope, prefix, i), cin, co, bn=not plain, relu=not plain))
        cin = co
    return layers


# --------------------------------------------------------------------------- MLP chains
# First SA layer: linear map before the grouping (see mlp_chain_forward).  False = the fused GATHER GEMM over the grouped
# rows (votenet_mlp_linear with a GATHER input), kept for comparison and tests.
PRE_LINEAR = True
# Max-pool over the nsample rows of a group started in the last GEMM's epilogue (votenet_mlp_linear_pool) instead of a
# separate pass over z (votenet_bn_relu_max).
POOL_IN_EPILOGUE = True
# Backward of that pooled layer in Gram form (pool_bwd.hip): both GEMMs contract over cin x cin instead of cin x cout and z
# of the layer is neither stored nor read.  False = votenet_mlp_wgrad_bn / votenet_mlp_dgrad_bn on the stored z.
POOL_GRAM_BACKWARD = True
# Backward of an ASSEMBLED first layer on the piece layout decomposed over the points (csrc/half.hip, round 4): the input-gradient GEMM of
# the layer above is a plain one (no epilogue gathers of the per-point table), the pass over the rows bucketed by point reduces the
# first layer's BatchNorm backward and scatters the MASKED gradient, a pass over the points finishes S.  False = the epilogue reduce
# (votenet_assembled_dgrad_bn_reduce_half) + votenet_group_linear_backward_sorted.
ASSEMBLED_DECOMPOSED = True
# Inference mode of every BatchNorm (model.py:98-139 runs with is_training=False): dict layer name -> mlp.FrozenBN built from the
# moving averages (VoteNetHotPath.inference_bn); None = training mode (batch statistics).
class _FrozenBN(threading.local):
    """Inference-mode BatchNorm table (layer name -> mlp.FrozenBN) of the forward pass running on THIS thread, or None (batch
    statistics).  Set with frozen_bn(table) as a context manager (VoteNetHotPath.predict does): thread-local and nesting-safe, so two
    models used from two threads, or a predict() inside another model's pass, do not see each other's tables."""
    table = None


_FROZEN = _FrozenBN()


class frozen_bn:
    def __init__(self, table):
        self.table = table

    def __enter__(self):
        self.prev, _FROZEN.table = _FROZEN.table, self.table
        return self.table

    def __exit__(self, *a):
        _FROZEN.table = self.prev
        return False


def mlp_chain_forward(layers, rows, first, tape, pool_k=0, keep_z=True):
    """Run a chain of layers over `rows` rows.  first = ('gather', xyz, new_xyz, feat, idx) or ('dense', x).
    Returns (z_last, pend): the last layer's RAW output and its BatchNorm as an mlp.PendingBN (None for a plain last
    layer) -- nobody has launched a finalize: every BatchNorm is derived from the raw sums in the prologue of the kernel
    that consumes it (the next GEMM inside the chain; the pooling / activation kernel of the caller for the last layer).
    Appends one record per layer to `tape`.
    pool_k > 0: the chain is followed by a max over groups of pool_k rows (utils.py:132); where the shape allows, the last
    GEMM's epilogue starts the pool (raw max / min per group) and the last record carries 'pool' for bn_pool_finalize;
    keep_z=False then skips the store of the last layer's z altogether (nothing downstream reads it in inference)."""
    z = None
    pend = None  # BatchNorm of z whose finalize rides in the prologue of z's consumer (mlp.PendingBN)
    prev_relu = False
    for i, L in enumerate(layers):
        w, b = L.p("W"), L.p("b")
        pool = None
        sc = pend.scale if pend is not None else None  # views: filled when the consumer below has run
        sh = pend.shift if pend is not None else None
        if i == 0 and first[0] == "narrow":
            # narrow first layer (csrc/narrow.hip): no kernel writes z0 = u8 W0 + b0; its BatchNorm statistics follow from the moments
            # of u8 and the next layer's GEMM rebuilds it in its operand loader
            _, u8, mom = first[:3]
            half = first[3] if len(first) > 3 else None  # mlp.HalfLayout: u8 (and every row tensor of the chain) holds compact rows
            zn = None
            st = M.narrow_stats(rows, mom, w, b) if (L.bn and _FROZEN.table is None) else None
            rec = dict(layer=L, kind="narrow", u8=u8, mom=mom, half=half)
        elif i == 0 and first[0] == "assembled":
            # first layer assembled inside its consumers (csrc/assemble.hip): the per-point GEMM P = feat W[3:] + b is all that runs here;
            # the BatchNorm statistics of z0 = P[idx] + dxyz W[0:3] come from one pass over the points
            _, xyz, new_xyz, feat, idx, geo, cntv, mom = first[:8]
            half = first[8] if len(first) > 8 else None  # mlp.HalfLayout: geo (and every row tensor of the chain) holds compact rows
            bb, nn, cc = feat.shape
            P, _ = M.linear_dense(feat.reshape(bb * nn, cc), w[3:], b, want_stats=False)
            zn = None
            st = M.assemble_stats(P, cntv, w[:3], mom) if (L.bn and _FROZEN.table is None) else None
            rec = dict(layer=L, kind="assembled", xyz=xyz, new_xyz=new_xyz, feat=feat, idx=idx, geo=geo, P=P, wx=w[:3], half=half,
                       cntv=cntv, mom=mom)
        elif i == 1 and first[0] == "assembled":
            r0 = tape[-1]
            zn, st = M.assembled_linear(r0["geo"], r0["P"], r0["wx"], w, b, pend, prev_relu, want_stats=L.bn, half=r0["half"])
            rec = dict(layer=L, kind="dense", x=None, assembled=True, in_scale=sc, in_shift=sh, in_relu=prev_relu, half=r0["half"])
        elif i == 1 and first[0] == "narrow":
            half = tape[-1]["half"]
            # training on the piece layout: the GEMM also leaves the first layer's ReLU mask (2 bytes per row and 16 channels) for the
            # backward pass, whose input-gradient epilogue then needs no rebuild of z0 (mlp.NARROW_MASK)
            want_mask = bool(M.NARROW_MASK and keep_z and half is not None and prev_relu and _FROZEN.table is None and M.COEF_TAIL
                             and M.narrow_mask_supported(first[1].shape[0], layers[0].cout))
            res = M.narrow_linear(first[1], layers[0].p("W"), layers[0].p("b"), w, b, pend, prev_relu, want_stats=L.bn, half=half,
                                  want_mask=want_mask)
            zn, st = res[0], res[1]
            rec = dict(layer=L, kind="dense", x=None, narrow=True, in_scale=sc, in_shift=sh, in_relu=prev_relu, half=half,
                       mask0=res[2] if want_mask else None)
        elif i == 0 and first[0] == "gather":
            # conv over the sample_and_group concat [xyz[idx]-new_xyz | feat[idx]] (utils.py:50-57,125-127).  A gather
            # commutes with a per-point linear map, so the feature block is ONE GEMM over the b*n points (P = feat W[3:])
            # and the layer output is assembled per grouped row: z = P[idx] + dxyz W[0:3] + bias (votenet_group_linear).
            _, xyz, new_xyz, feat, idx = first
            co = w.shape[1]
            if feat is not None and PRE_LINEAR and co % 4 == 0 and co <= 1024 and 256 % (co // 4) == 0:
                bb, nn, cc = feat.shape
                P, _ = M.linear_dense(feat.reshape(bb * nn, cc), w[3:], None, want_stats=False)
                zn, st = M.group_linear(xyz, new_xyz, idx, P, w[:3], b, want_stats=L.bn)
            else:
                zn, st = M.linear_gather(xyz, new_xyz, feat, idx, w, b, want_stats=L.bn)
            rec = dict(layer=L, kind="gather", xyz=xyz, new_xyz=new_xyz, feat=feat, idx=idx)
        elif i == 0 and L.cin_pad and PAD_RAGGED_IN and first[1].is_cuda:
            # [x | 0] against [W ; 0]; first = ("dense", x, xp): the caller built the padded rows itself (x = xp[:, :cin])
            xp = first[2] if len(first) > 2 else torch.nn.functional.pad(first[1], (0, L.cin_pad - L.cin))
            zn, st = M.linear_dense(xp, L.store.padded_rows(L.name + "/W", L.cin_pad), b, want_stats=L.bn)
            rec = dict(layer=L, kind="dense", x=xp, in_scale=None, in_shift=None, in_relu=False, cin_padded=True)
        elif i == 0:
            zn, st = M.linear_dense(first[1], w, b, want_stats=L.bn)
            rec = dict(layer=L, kind="dense", x=first[1], in_scale=None, in_shift=None, in_relu=False)
        elif pool_k and i == len(layers) - 1 and L.bn and POOL_IN_EPILOGUE and \
                M.linear_pool_supported(rows, w.shape[0], w.shape[1], pool_k):
            # training does not store z of this layer either when its backward runs in Gram form (it never reads z)
            gram_form = POOL_GRAM_BACKWARD and pend is not None and M.pool_backward_supported(w.shape[0], w.shape[1], pool_k)
            half = tape[0].get("half") if first[0] in ("assembled", "narrow") else None
            if half is not None and not (gram_form or not keep_z):
                raise M.L.VotenetError("piece layout: the pooled layer's backward must be in Gram form")
            zn, st, pool = M.linear_dense_pool(z, w, pool_k, b, None, None, prev_relu, keep_z=keep_z and not gram_form, in_bn=pend, half=half,
                                               gamma=L.p("gamma") if half is not None else None)
            rec = dict(layer=L, kind="dense", x=z, in_scale=sc, in_shift=sh, in_relu=prev_relu, gram_form=gram_form,
                       in_affine=pend.out if pend is not None else None, cout=w.shape[1], half=half)
        elif L.cout_pad and i > 0:
            # ragged plain layer on zero-padded copies of W and b: the fast GEMM, output (rows, cout_pad), the layer's z = [:, :cout]
            zp, st = M.linear_dense(z, L.store.padded(L.name + "/W", L.cout_pad), L.store.padded(L.name + "/b", L.cout_pad), None, None,
                                    prev_relu, want_stats=False, in_bn=pend)
            zn = zp[:, :L.cout]
            rec = dict(layer=L, kind="dense", x=z, in_scale=sc, in_shift=sh, in_relu=prev_relu, padded=True)
        else:
            zn, st = M.linear_dense(z, w, b, None, None, prev_relu, want_stats=L.bn, in_bn=pend)
            rec = dict(layer=L, kind="dense", x=z, in_scale=sc, in_shift=sh, in_relu=prev_relu)
        if L.bn and _FROZEN.table is not None:
            pend = _FROZEN.table[L.name]  # moving averages: the batch sums of this launch are ignored
            rec.update(scale=pend.scale, shift=pend.shift)
        elif L.bn:
            blk = L.store._bn_views.get(L.name)
            pend = M.PendingBN(st, L.p("gamma"), L.p("beta"), rows, out=blk)
            rec.update(scale=pend.scale, shift=pend.shift, mean=pend.mean, var=pend.var, bn_out=pend.out)
            if blk is not None:
                # the block is PERSISTENT (one per layer, rewritten by every training-mode forward pass of the net): the record
                # remembers which pass wrote it, and whoever reads it later through this record checks that no other pass has since
                rec["bn_pass"] = L.store.bn_block_written(L.name)
        else:
            pend = None
        rec.update(z=zn, rows=rows, pool=pool)
        tape.append(rec)
        z, prev_relu = zn, L.relu
    return z, pend


# Weight gradients hang off the backward chain (reduce -> coef -> dgrad -> reduce ...): nothing downstream needs them before
# the optimizer.  When the owner sets WGRAD_STREAM (a second HIP stream) they are launched there, concurrently with the input
# gradient of the same layer; wgrad_join() makes the current stream wait for them.  Tensors a side-stream kernel reads are
# tagged with record_stream so the caching allocator does not hand their memory out early.
WGRAD_STREAM = None
_wgrad_pending = False


# Every hand-over to WGRAD_STREAM costs an event (a barrier packet in the main queue: a few microseconds in which the chain
# stands still).  WGRAD_BATCH = True collects the weight-gradient launches of a module and hands them over together at
# wgrad_flush() (one event per module: its weight gradients then run underneath the NEXT module's chain); modules named in
# WGRAD_FINE (the last one of the backward pass, whose weight gradients have nothing after them to hide under) keep one
# hand-over per launch.
WGRAD_BATCH = False  # measured (tools/ab_step.py, same box): 7.81 ms per step per launch vs 7.99 ms batched per module -- where the weight gradients run matters more than the ~30 events saved
_wgrad_deferred = []
_wgrad_fine = False


def wgrad_fine(on):
    global _wgrad_fine
    _wgrad_fine = bool(on)


# hand-over events, reused round-robin: a wait that has been enqueued keeps the record it saw, so an event may be recorded again as
# soon as its wait is in the queue (creating and destroying a HIP event per hand-over was ~3 us of host time, 25 times per step)
_EVENTS = []
_event_turn = 0


# Set to a list while a stretch of the step is being captured into a HIP graph (model.StretchGraph): hand-overs then use fresh events
# (a captured record must not be mixed with the eager life of a pooled event) and, instead of record_stream -- whose bookkeeping
# the caching allocator defers during a capture -- every tensor a side-branch kernel reads is kept alive until the capture ends, so that
# the graph's private pool cannot hand its memory to a later allocation of the main branch.
CAPTURE_KEEP = None
# A list while a stretch is being captured in SEGMENTS (model.StretchGraph): a weight-gradient launch is then not captured at all but
# recorded as (thunk, tensors) -- the owner replays the segment's graph and runs its thunks on the weight-gradient stream afterwards,
# launch by launch, beside the next segment.  (Captured on a side branch of the graph instead, the branch's internal stream shared a
# hardware queue with the geometry prefetch: the whole graph then waited for the 1.6 ms sampling kernel -- 3.94 -> 5.8 ms per step.)
CAPTURE_DEFER = None


def _pooled_event():
    global _event_turn
    if CAPTURE_KEEP is not None:
        return torch.cuda.Event()
    if len(_EVENTS) < 64:
        _EVENTS.append(torch.cuda.Event())
        return _EVENTS[-1]
    _event_turn = (_event_turn + 1) % 64
    return _EVENTS[_event_turn]


def _hand_over(thunks, tensors):
    global _wgrad_pending
    main = torch.cuda.current_stream()
    ev = _pooled_event()
    ev.record(main)
    WGRAD_STREAM.wait_event(ev)
    if CAPTURE_KEEP is not None:
        CAPTURE_KEEP.extend(tensors)
        CAPTURE_KEEP.append(thunks)  # (the closures hold what the kernels they launch read)
    else:
        for t in tensors:
            t.record_stream(WGRAD_STREAM)
    # set_stream both ways instead of the `with torch.cuda.stream(...)` context: the context manager looks the current stream up
    # twice on entry and exit (~40 us per hand-over on the host, ~24 hand-overs per step)
    torch.cuda.set_stream(WGRAD_STREAM)
    try:
        for f in thunks:
            f()
    finally:
        torch.cuda.set_stream(main)
    _wgrad_pending = True


def on_wgrad_stream(fn, *tensors):
    """Run fn() -- launches of weight-gradient kernels reading `tensors` -- on WGRAD_STREAM after everything issued so far on the
    current stream (now, or together with the module's other weight gradients at the next wgrad_flush()); on the current
    stream when there is no weight-gradient stream."""
    tensors = [t for t in tensors if isinstance(t, torch.Tensor)]
    if CAPTURE_DEFER is not None:
        CAPTURE_DEFER.append((fn, tensors))
    elif WGRAD_STREAM is None:
        fn()
    elif WGRAD_BATCH and not _wgrad_fine:
        _wgrad_deferred.append((fn, tensors))
    else:
        _hand_over([fn], tensors)


def wgrad_flush():
    if _wgrad_deferred:
        if WGRAD_STREAM is None:
            for f, _ in _wgrad_deferred:
                f()
        else:
            _hand_over([f for f, _ in _wgrad_deferred], [t for _, ts in _wgrad_deferred for t in ts])
        _wgrad_deferred.clear()


def wgrad_join():
    """The current stream waits for every weight-gradient kernel launched on WGRAD_STREAM so far."""
    global _wgrad_pending
    wgrad_flush()
    if WGRAD_STREAM is not None and _wgrad_pending:
        ev = _pooled_event()
        ev.record(WGRAD_STREAM)
        torch.cuda.current_stream().wait_event(ev)
        _wgrad_pending = False


def check_bn_block(rec):
    """A layer's BatchNorm vectors (scale | shift | mean | var) live in ONE persistent block per layer: a record of an earlier forward
    pass of the same net reads another pass's statistics once a second training-mode pass has run (predict(batch_statistics=True), a
    forward without a tape, a finite-difference probe between forward(tape) and backward(tape)).  Raises instead."""
    n = rec.get("bn_pass")
    if n is not None and rec["layer"].store._bn_pass.get(rec["layer"].name) != n:
        raise M.L.VotenetError("BatchNorm statistics of layer %s: the record belongs to forward pass %d, the layer's persistent block was "
                               "rewritten by a later training-mode pass (run backward / update_moving_averages before the next forward of "
                               "the same net)" % (rec["layer"].name, n))


def mlp_chain_backward(recs, g, mode, argmax=None, k=0, need_input_grad=True, zsel=None, g_padded=None):
    """Backward of mlp_chain_forward.  g / mode describe the gradient arriving at the LAST layer:
         'pool'  : g = gout (rows/k, c) of the max over k of relu(bn(z))      (SA layers, utils.py:132)
         'act'   : g = dy (rows, c) of y = relu(bn(z))                         (FP layers)
         'plain' : g = dz (rows, c) of a last layer without BN / activation    (mlp2, voting)
    Accumulates parameter gradients into the store's gradient bucket.  Returns the gradient with respect to the
    chain's input (rows, cin) for a dense input.  For a gather input it stops at the first layer and returns either
    dict(dz=...) (rows, cout) or dict(da=, coef=, relu=) when the fused first-layer backward kernel will form dz itself:
    SAModule.backward finishes the layer (weight gradient, point gradients)."""
    da = g
    coef_ahead = None  # BatchNorm-backward coefficients of layer i already produced by the layer above (the reduce rides in its
    #                    scatter pass / GEMM epilogue, the coefficient vector in that kernel's tail)

    def tail_of(rec):
        Lr = rec["layer"]
        return (rec["rows"], Lr.p("gamma"), Lr.gp("gamma"), Lr.gp("beta"))
    for r in recs:
        check_bn_block(r)
    for i in range(len(recs) - 1, -1, -1):
        r = recs[i]
        L = r["layer"]
        z = r["z"]
        pooled = (mode == "pool" and i == len(recs) - 1)
        want_da = i > 0 or need_input_grad
        if pooled and r.get("gram_form") and zsel is not None:
            # Gram form (pool_bwd.hip).  The short dependent chain reduce -> coef -> prepare goes first, alone on the GPU; the
            # weight-gradient stream (x^T x, arg-max rows, finish) starts beside the dense GEMM that follows it
            x, aff, W, b = r["x"], r["in_affine"], L.p("W"), L.p("b")
            bn = (r["scale"], r["shift"], r["mean"], r["var"])
            coef = M.bn_backward_reduce_pool(da, zsel, *bn, L.relu, tail=tail_of(r))
            mm = M.pool_dgrad_prepare(W, b, coef, x.shape[0]) if want_da else None
            half = r.get("half")
            def _pooled_wgrad(x=x, aff=aff, r=r, W=W, b=b, coef=coef, L=L, da=da, half=half):
                G = r.pop("gram_ahead", None)  # the Gram matrix depends on forward data only: train_step may have launched it ahead
                if G is None:
                    G = M.gram(x, aff[:2], r["in_relu"], half=half)
                M.pool_wgrad(x, r["in_scale"], r["in_shift"], r["in_relu"], G, W, b, coef, L.relu, da, argmax, zsel, k, L.gp("W"), half=half)
            on_wgrad_stream(_pooled_wgrad, x, aff, coef, da, argmax, zsel)
            if not want_da:
                return None
            below = recs[i - 1]
            if below["layer"].bn and below["z"] is x:
                da, coef_ahead = M.pool_dgrad(x, r["in_scale"], r["in_shift"], r["in_relu"], W, b, L.wT(), coef, L.relu, da, argmax,
                                              zsel, k, below=(below["scale"], below["shift"], below["mean"], below["var"],
                                                              below["layer"].relu), mm=mm, below_tail=tail_of(below), half=half)
            else:
                da = M.pool_dgrad(x, r["in_scale"], r["in_shift"], r["in_relu"], W, b, L.wT(), coef, L.relu, da, argmax, zsel, k, mm=mm,
                                  half=half)
            continue
        rows, c = r["rows"], L.cout  # z is None for a first layer that is never stored (narrow / assembled)
        if L.bn:
            bn = (r["scale"], r["shift"], r["mean"], r["var"])
            if coef_ahead is not None:
                coef, coef_ahead = coef_ahead, None
            else:
                coef = M.bn_backward_reduce(z, *bn, L.relu, da, argmax=argmax if pooled else None, k=k if pooled else 0,
                                            tail=tail_of(r))
            # d bias of a BatchNorm'ed layer is identically zero (BN removes the mean): left at 0
            if r.get("assembled"):
                # second layer above an ASSEMBLED first layer (i == 1): both GEMMs rebuild z0 from geo + P; the input-gradient GEMM's
                # epilogue reduces the first layer's BatchNorm backward on the rebuilt z0
                r0 = recs[0]
                geo, Pt, wx, half = r0["geo"], r0["P"], r0["wx"], r0["half"]
                on_wgrad_stream(lambda r=r, z=z, coef=coef, L=L, da=da: M.assembled_wgrad_bn(
                    geo, Pt, wx, r["in_scale"], r["in_shift"], r["in_relu"], z, coef, L.relu, da, L.gp("W"), half=half), geo, Pt, z, coef, da)
                if ASSEMBLED_DECOMPOSED and half is not None and r0["layer"].bn and r0.get("cntv") is not None:
                    # the first layer's backward decomposed over the points (csrc/half.hip): a PLAIN input-gradient GEMM here -- no epilogue
                    # that gathers the per-point table --; the pass over the rows bucketed by point that follows (SAModule.
                    # _first_layer_backward) reduces the first layer's BatchNorm backward itself
                    da = M.dgrad_bn_half(z, coef, L.relu, L.wT(), da, half)
                    return dict(da=da, decomposed=True, relu=r0["layer"].relu, tail=tail_of(r0),
                                bn=(r0["scale"], r0["shift"], r0["mean"], r0["var"]))
                da, coef_ahead = M.assembled_dgrad_bn_reduce(z, coef, L.relu, L.wT(), da, geo, Pt, wx,
                                                              (r0["scale"], r0["shift"], r0["mean"], r0["var"], r0["layer"].relu),
                                                              below_tail=tail_of(r0), half=half)
                continue
            if r.get("narrow"):
                # second layer above a NARROW first layer (i == 1): both GEMMs rebuild z0 from u8; the input-gradient GEMM stores
                # nothing -- its epilogue leaves the first layer's BatchNorm-backward sums and the data term of its weight gradient
                r0 = recs[0]
                L0, u8, mom, half = r0["layer"], r0["u8"], r0["mom"], r0["half"]
                w0, b0 = L0.p("W"), L0.p("b")
                on_wgrad_stream(lambda r=r, z=z, coef=coef, L=L, da=da: M.narrow_wgrad_bn(
                    u8, w0, b0, r["in_scale"], r["in_shift"], r["in_relu"], z, coef, L.relu, da, L.gp("W"), half=half), u8, z, coef, da)
                coef0, ug = M.narrow_dgrad_bn_reduce(z, coef, L.relu, L.wT(), da, u8, w0, b0,
                                                     (r0["scale"], r0["shift"], r0["mean"], r0["var"], L0.relu), tail=tail_of(r0), half=half,
                                                     mask=r.get("mask0") if M.NARROW_MASK else None)
                M.narrow_wgrad_first(mom, ug, coef0, w0, b0, L0.gp("W"))
                return None  # a leaf: nothing upstream takes a gradient
            if r.get("cin_padded") and not pooled and M.dgrad_bn_supported(rows, c, L.cin_pad):
                # the same layer on [x | 0] and [W ; 0]: dW through a padded scratch (its rows >= cin multiply zeros), da = dz [W ; 0]^T
                def _padded_in_wgrad(r=r, z=z, coef=coef, L=L, da=da):
                    G = M._zeros_f32((L.cin_pad, L.cout), z.device)
                    M.wgrad_dense_bn(r["x"], z, coef, L.relu, G, da=da)
                    gw = L.gp("W")
                    M.row_segments(L.cin, [(gw, gw, G[:L.cin])])  # dW += the scratch's first cin rows (in place, one launch)
                on_wgrad_stream(_padded_in_wgrad, r["x"], z, coef, da)
                if not want_da:
                    return None
                da = M.dgrad_bn(z, coef, L.relu, L.store.padded_rows(L.name + "/W", L.cin_pad, transpose=True), da=da)[:, :L.cin]
                continue
            if r["kind"] == "dense" and (not want_da or M.dgrad_bn_supported(rows, c, r["x"].shape[1])):
                # dz never materialised: both GEMMs rebuild it from (da | gout, z, coef) in their loaders
                src = dict(gout=da, argmax=argmax, k=k) if pooled else dict(da=da)
                on_wgrad_stream(lambda r=r, z=z, coef=coef, L=L, src=src: M.wgrad_dense_bn(
                    r["x"], z, coef, L.relu, L.gp("W"), in_scale=r["in_scale"], in_shift=r["in_shift"], in_relu=r["in_relu"], **src),
                    r["x"], z, coef, da, argmax if pooled else None)
                if not want_da:
                    return None
                below = recs[i - 1] if i > 0 else None
                if FUSE_BN_REDUCE and not pooled and below is not None and below["layer"].bn and below["z"] is r["x"]:
                    # the layer below's BatchNorm-backward sums come out of this GEMM's store epilogue
                    da, coef_ahead = M.dgrad_bn(z, coef, L.relu, L.wT(), da=da, below=(
                        below["z"], below["scale"], below["shift"], below["mean"], below["var"], below["layer"].relu),
                        below_tail=tail_of(below))
                else:
                    da = M.dgrad_bn(z, coef, L.relu, L.wT(), **src)
                continue
            if i == 0 and r["kind"] == "assembled":
                return dict(da=da, coef=coef, relu=L.relu)  # dz is formed inside votenet_group_linear_backward_assembled
            if i == 0 and r["kind"] == "gather" and PRE_LINEAR and not pooled and r["feat"] is not None and \
                    M.group_linear_backward_supported(c, r["idx"].shape[2]):
                return dict(da=da, coef=coef, relu=L.relu)  # dz is formed inside votenet_group_linear_backward
            dz = M.bn_backward_apply(z, coef, L.relu, da, argmax=argmax if pooled else None, k=k if pooled else 0)
        else:
            dz = da
            # (a weight gradient like the others: on their stream -- the two launches of a column sum were 31 us of the main chain per step)
            on_wgrad_stream(lambda dz=dz, L=L: M.bias_grad(dz, L.gp("b")), dz)
            if r.get("padded"):
                # the same layer on the padded copies: dz -> [dz | 0] (rows, cout_pad); dW through a padded scratch, da = dz_p [W | 0]^T
                # (g_padded: the caller already holds the last layer's gradient as [g | 0], g = g_padded[:, :cout])
                if g_padded is not None and i == len(recs) - 1 and g_padded.shape[1] == L.cout_pad:
                    dzp = g_padded
                else:
                    dzp = torch.empty((dz.shape[0], L.cout_pad), dtype=torch.float32, device=dz.device)
                    M.row_segments(dz.shape[0], [(dzp[:, :L.cout], dz, None), (dzp[:, L.cout:], None, None)])

                def _padded_wgrad(r=r, dzp=dzp, L=L):
                    G = M._zeros_f32((L.cin, L.cout_pad), dzp.device)
                    M.wgrad_dense(r["x"], dzp, G, r["in_scale"], r["in_shift"], r["in_relu"])
                    gw = L.gp("W")
                    M.row_segments(L.cin, [(gw, gw, G[:, :L.cout])])  # dW += the scratch's first cout columns (in place, one launch)
                on_wgrad_stream(_padded_wgrad, dzp, r.get("x"))
                if want_da:
                    da, _ = M.linear_dense(dzp, L.store.padded(L.name + "/W", L.cout_pad, transpose=True), want_stats=False)
                else:
                    da = None
                continue
        if i == 0 and r["kind"] == "gather":
            return dict(dz=dz)  # the caller (SAModule.backward) finishes the first layer: it owns idx / pts_cnt / the tables
        on_wgrad_stream(lambda r=r, dz=dz, L=L: M.wgrad_dense(r["x"], dz, L.gp("W"), r["in_scale"], r["in_shift"], r["in_relu"]),
                        dz, r.get("x"))
        if want_da:
            da, _ = M.linear_dense(dz, L.wT(), want_stats=False)  # da_prev = dz W^T
        else:
            da = None
    return da


def add_rows(a, b):
    """a + b for two (..., c) tensors of equal shape as ONE launch of the glue kernel; either may be a column slice of a wider
    row-major tensor (FPModule.backward returns d_points1 that way).  -> a new contiguous tensor."""
    c = a.shape[-1]
    rows = a.numel() // c
    out = torch.empty(a.shape, dtype=torch.float32, device=a.device)

    def two_d(t):
        if t.is_contiguous():
            return t.view(rows, c)
        if t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1):
            return t.as_strided((rows, c), (t.stride(1), 1), t.storage_offset())
        return t.contiguous().view(rows, c)
    M.row_segments(rows, [(out.view(rows, c), two_d(a), two_d(b))])
    return out


# --------------------------------------------------------------------------- SA / FP modules
def _group_indices(radius, nsample, xyz, new_xyz, knn):
    """utils.py:46-49: ball query, or the nsample nearest points with knn=True (every slot then holds a distinct point)."""
    if knn:
        _, idx = tf_grouping.knn_point(nsample, xyz, new_xyz)
        return M.attach_inverse(idx, xyz.shape[1]), torch.full(idx.shape[:2], nsample, dtype=torch.int32, device=idx.device)
    idx, cnt = tf_grouping.query_ball_point(radius, nsample, xyz, new_xyz)
    return M.attach_inverse(idx, xyz.shape[1]), cnt  # the grouping's inverse (idx._inv) for the deterministic backward pass


def sample_and_group(npoint, radius, nsample, xyz, sample_xyz=None, knn=False):
    """utils.py:42-49 (geometry part): FPS on sample_xyz if given, centres gathered from xyz."""
    fps_idx = tf_sampling.farthest_point_sample(npoint, sample_xyz if sample_xyz is not None else xyz)
    new_xyz = tf_sampling.gather_point(xyz, fps_idx)
    idx, pts_cnt = _group_indices(radius, nsample, xyz, new_xyz, knn)
    return fps_idx, new_xyz, idx, pts_cnt


class SAModule:
    """pointnet_sa_module (utils.py:93-158) with group_all=False, pooling='max', use_xyz=True."""

    def __init__(self, store, scope, npoint, radius, nsample, cin, mlp, mlp2=None, knn=False, prefix="conv", leaf=False):
        """leaf: the input points carry no gradient (the first module of a network): backward() then returns (None, None), which
        lets a narrow first layer (3 + cin <= 8) run without ever storing its output (NARROW_FIRST, csrc/narrow.hip)."""
        self.npoint, self.radius, self.nsample, self.knn = npoint, radius, nsample, knn
        self.leaf, self.cin = leaf, cin
        self.mlp = make_mlp(store, scope, 3 + cin, mlp, prefix)
        store.want_transpose(self.mlp[0].name + "/W", 3, None)  # W[3:]^T: the per-point feature gradient
        store._split_rows.append((self.mlp[0].name + "/W", 3, None))  # W[3:]: the per-point GEMM of the forward pass gets its image too
        store.want_transpose(self.mlp[0].name + "/W", 0, 3)     # W[:3]^T: the xyz gradient (proposal layer)
        self.mlp2 = make_mlp(store, scope, mlp[-1], mlp2, "conv_post_", last_plain=True) if mlp2 else None

    def narrow(self, rows):
        """True when this module runs its first layer in the narrow form for `rows` grouped rows."""
        m = self.mlp
        return bool(NARROW_FIRST and self.leaf and len(m) >= 3 and m[0].bn and m[1].bn and m[0].relu and
                    M.narrow_supported(rows, 3 + self.cin, m[0].cout, m[1].cout))

    def assembled(self, b, n):
        """True when this module's first layer is assembled inside its consumers (csrc/assemble.hip) for b scenes of n points."""
        m = self.mlp
        rows = b * self.npoint * self.nsample
        return bool(ASSEMBLE_FIRST and PRE_LINEAR and not M.DETERMINISTIC and self.cin > 0 and not self.narrow(rows) and len(m) >= 3
                    and m[0].bn and m[1].bn and m[0].relu and M.assembled_supported(rows, m[0].cout, m[1].cout)
                    and M.group_linear_backward_supported(m[0].cout, self.nsample) and b * n * m[0].cout * 4 < 2 ** 32)

    def half_groups(self, b, n):
        """True when this module's grouped MLP runs on the piece layout (csrc/half.hip: the rows that repeat slot 0 dropped by
        halves of a ball) for b scenes of n points: an assembled first layer, three BatchNorm'ed layers, the pooled one in Gram form."""
        m = self.mlp
        if not (HALF_GROUPS and not M.DETERMINISTIC and not self.knn and self.nsample == 64 and len(m) == 3 and m[2].bn and POOL_IN_EPILOGUE
                and POOL_GRAM_BACKWARD and (b * self.npoint) % 8 == 0 and M.pool_backward_supported(m[1].cout, m[2].cout, 64)):
            return False
        return bool(self.narrow(b * self.npoint * 64) or (m[1].cout == 128 and self.assembled(b, n)))

    def geometry(self, xyz, sample_xyz=None, fps_idx=None, points=None, ahead=True):
        """The weight-independent part of the layer (FPS, centres, ball query): can run ahead on a side stream.
        points: the module's input features; given for a narrow leaf module, the grouped rows u8 and their moments (which depend
        on coordinates and input features only) are appended to the result."""
        if fps_idx is None:
            geom = sample_and_group(self.npoint, self.radius, self.nsample, xyz, sample_xyz, self.knn)
        else:
            new_xyz = tf_sampling.gather_point(xyz, fps_idx)
            idx, pts_cnt = _group_indices(self.radius, self.nsample, xyz, new_xyz, self.knn)
            geom = (fps_idx, new_xyz, idx, pts_cnt)
        if (points is not None or self.cin == 0) and self.narrow(xyz.shape[0] * self.npoint * self.nsample):
            if ahead and self.half_groups(xyz.shape[0], xyz.shape[1]):
                half = M.half_groups(geom[3])
                geom = tuple(geom) + M.narrow_rows_half(xyz, geom[1], points, geom[2], geom[3], half) + (half,)
            else:
                geom = tuple(geom) + M.narrow_rows(xyz, geom[1], points, geom[2])
        elif (ahead or ASSEMBLE_INLINE) and self.half_groups(xyz.shape[0], xyz.shape[1]):
            # the layout and the count of its pieces: known on the host by the time the MLP runs when the geometry is computed ahead;
            # inside the step it serves (the proposal module: the votes exist only now) the count stays on the device -- waiting for it
            # would drain the queue -- and the kernels stop at it themselves
            half = M.half_groups(geom[3], device_count=not ahead)
            geom = tuple(geom) + M.assemble_rows_half(xyz, geom[1], geom[2], geom[3], half) + (half,)
            M.half_sort_rows(half, xyz.shape[0] * xyz.shape[1])  # the rows bucketed by point, for the first layer's backward
        elif self.assembled(xyz.shape[0], xyz.shape[1]) and (ahead or ASSEMBLE_INLINE):  # ahead=False: called inside the step it serves
            geom = tuple(geom) + M.assemble_rows(xyz, geom[1], geom[2], pts_cnt=geom[3], in_pass=not ahead)  # geo records + per-point sums: coordinates only
        return geom

    def forward(self, xyz, points, sample_xyz=None, tape=None, geom=None):
        """xyz (B,n,3), points (B,n,C) or None -> new_xyz (B,m,3), new_points (B,m,C'), idx (B,m,K).
        geom: the tuple returned by geometry() when it was computed ahead of time."""
        b = xyz.shape[0]
        geom = geom if geom is not None else self.geometry(xyz, sample_xyz, points=points)
        fps_idx, new_xyz, idx, pts_cnt = geom[:4]
        recs = []
        rows = b * self.npoint * self.nsample
        first = ("gather", xyz, new_xyz, points, idx)
        if self.narrow(rows):
            u8, mom = geom[4:6] if len(geom) >= 6 else M.narrow_rows(xyz, new_xyz, points, idx)
            first = ("narrow", u8, mom)
            if len(geom) >= 7:  # piece layout: the compact rows
                half = geom[6].resolve()
                first = ("narrow", half.u8, mom, half)
        elif points is not None and self.assembled(b, xyz.shape[1]) and (len(geom) >= 7 or ASSEMBLE_INLINE):
            geo, cntv, mom = geom[4:7] if len(geom) >= 7 else M.assemble_rows(xyz, new_xyz, idx, pts_cnt=pts_cnt, in_pass=True)
            first = ("assembled", xyz, new_xyz, points, idx, geo, cntv, mom)
            if len(geom) >= 8:  # piece layout: the compact rows
                half = geom[7].resolve()
                first = ("assembled", xyz, new_xyz, points, idx, half.geo, cntv, mom, half)
        z, pend = mlp_chain_forward(self.mlp, rows, first, recs, pool_k=self.nsample, keep_z=tape is not None)
        if recs[-1]["pool"] is not None:  # utils.py:132, the pass over z already done by the GEMM epilogue
            res = M.bn_pool_finalize(recs[-1]["pool"], None, None, True, want_argmax=tape is not None, bn=pend,
                                     want_zsel=tape is not None and bool(recs[-1].get("gram_form")), half=recs[-1].get("half"))
            pooled, argmax, zsel = res if len(res) == 3 else (res[0], res[1], None)
        else:
            zsel = None
            sc, sh = pend.finalize()
            pooled, argmax = M.bn_relu_max(z, self.nsample, sc, sh, True, want_argmax=tape is not None)
        recs2 = []
        out = pooled
        if self.mlp2:
            z2, _ = mlp_chain_forward(self.mlp2, b * self.npoint, ("dense", pooled), recs2)
            out = z2  # last conv_post layer has no activation (utils.py:153)
        if tape is not None:
            tape.append(dict(op="sa", module=self, recs=recs, recs2=recs2, argmax=argmax, zsel=zsel, fps_idx=fps_idx, idx=idx, pts_cnt=pts_cnt,
                             xyz=xyz, points=points, new_xyz=new_xyz, b=b))
        return new_xyz, out.reshape(b, self.npoint, -1), idx


    def backward(self, rec, g_out, need_feat_grad=True, need_xyz_grad=False):
        """g_out (B,m,C') -> d_points (B,n,C) or None, d_xyz (B,n,3) or None."""
        b = rec["b"]
        g = g_out.reshape(b * self.npoint, -1).contiguous()
        if self.mlp2:
            g = mlp_chain_backward(rec["recs2"], g, "plain", need_input_grad=True)
        need_feat = need_feat_grad and rec["points"] is not None
        h = mlp_chain_backward(rec["recs"], g, "pool", argmax=rec["argmax"], k=self.nsample, need_input_grad=need_feat,
                               zsel=rec.get("zsel"))
        if rec["recs"][0]["kind"] == "narrow":
            if need_feat or need_xyz_grad:
                raise ValueError("SAModule(leaf=True): the narrow first layer keeps no per-row gradient; build the module with leaf=False")
            return None, None
        return self._first_layer_backward(rec, h, need_feat, need_xyz_grad)

    def _first_layer_backward(self, rec, h, need_feat, need_xyz_grad):
        """Backward of z = P[idx] + dxyz W[0:3] (P = feat W[3:]) given dz (rows, cout):
             dW[0:3] += dxyz^T dz                     (over the grouped rows)
             S = scatter-add of dz rows by idx        (b, n, cout)  -- GroupPointGrad on the layer OUTPUT width
             dW[3:]  += feat^T S ,  d_feat = S W[3:]^T                (two GEMMs over the b*n points, not the grouped rows)
             d_xyz / d_new_xyz from dz W[0:3]^T       (proposal layer only)
        h = dict(dz=) or dict(da=, coef=, relu=): in the second form one kernel forms dz from (z, da), scatters it and
        reduces the xyz rows of dW in a single pass (votenet_group_linear_backward)."""
        r0 = rec["recs"][0]
        L0 = r0["layer"]
        W, gW = L0.p("W"), L0.gp("W")
        xyz, new_xyz, feat, idx, pts_cnt = r0["xyz"], r0["new_xyz"], r0["feat"], r0["idx"], rec["pts_cnt"]
        b, n = xyz.shape[:2]
        cout = W.shape[1]
        d_feat = d_xyz = None
        S = None
        if "dz" in h:
            dz = h["dz"]
            if feat is None or PRE_LINEAR:
                on_wgrad_stream(lambda: M.wgrad_gather(xyz, new_xyz, None, idx, dz, gW), dz)  # rows 0..2 of dW (no feature block)
            if feat is not None and PRE_LINEAR:
                S, _, _ = M.group_concat_grad(dz, None, idx, pts_cnt, n, cout)
        elif r0["kind"] == "assembled" and r0.get("half") is not None:
            half = r0["half"]
            if h.get("decomposed"):
                S, coef0 = M.group_linear_backward_decomposed(half, b, n, r0["P"], r0["wx"], h["da"], h["bn"], h["relu"], h["tail"],
                                                              r0["cntv"], r0["mom"], gW[:3], defer=on_wgrad_stream)
                h = dict(h, coef=coef0)
                dz = None
            else:
                S, dz = M.group_linear_backward_half(half, b, n, r0["P"], r0["wx"], h["da"], h["coef"], h["relu"], gW[:3]), None
            if need_xyz_grad:
                # dz0 W[0:3]^T is linear in dz0: the points receive S W[0:3]^T, a centre minus the sum of its rows' dz0 times W[0:3]^T
                d_xyz = M.rows_dot3(S.view(b * n, cout), W[:3]).view(b, n, 3)
                d_new = M.rows_dot3(M.half_centre_sums(half, r0["P"], r0["wx"], h["da"], h["coef"], h["relu"]), W[:3]).view(b, -1, 3)
                d_xyz = tf_sampling.gather_point_grad_raw(n, rec["fps_idx"], d_new, into=d_xyz)
                need_xyz_grad = False  # done
        elif r0["kind"] == "assembled":
            S, dz = M.group_linear_backward_assembled(xyz, new_xyz, idx, pts_cnt, r0["P"], r0["wx"], h["da"], h["coef"], h["relu"],
                                                      gW[:3], want_dz=need_xyz_grad)
        else:
            S, dz = M.group_linear_backward(xyz, new_xyz, idx, pts_cnt, r0["z"], h["da"], h["coef"], h["relu"], gW[:3],
                                            want_dz=need_xyz_grad)
        if feat is not None:
            c = feat.shape[2]
            if PRE_LINEAR:
                S2, feat2 = S.view(b * n, cout), feat.reshape(b * n, c)
                on_wgrad_stream(lambda: M.wgrad_dense(feat2, S2, gW[3:]), S2, feat2)
                if need_feat:
                    d2, _ = M.linear_dense(S2, L0.wT(3, None), want_stats=False)
                    d_feat = d2.view(b, n, c)
            else:  # the fused GATHER GEMMs over the grouped rows
                on_wgrad_stream(lambda: M.wgrad_gather(xyz, new_xyz, feat, idx, dz, gW), dz, feat)
                if need_feat:
                    d_rows_feat, _ = M.linear_dense(dz, L0.wT(3, None), want_stats=False)
                    d_feat, _, _ = M.group_concat_grad(d_rows_feat, None, idx, pts_cnt, n, c)
        if need_xyz_grad:
            d_rows_xyz = M.rows_dot3(dz, W[:3])  # dz W[0:3]^T, three columns: a streaming kernel, not a 128-wide GEMM tile
            _, d_xyz, d_new = M.group_concat_grad(None, d_rows_xyz, idx, pts_cnt, n, 0)
            d_xyz = tf_sampling.gather_point_grad_raw(n, rec["fps_idx"], d_new, into=d_xyz)  # new_xyz = gather(xyz, fps_idx): accumulated in place
        return d_feat, d_xyz


class SAModuleMSG:
    """pointnet_sa_module_msg (utils.py:161-201): one FPS, then per scale i a ball query (radius_list[i], nsample_list[i]),
    the MLP mlp_list[i] (layers "conv<i>_<j>") and a max-pool; the per-scale features are concatenated.
    Each scale is an SAModule sharing the sampled centres, so it runs through the same fused kernels.

    Weight layout: the reference concatenates [grouped_points, grouped_xyz] here (utils.py:186-187) -- features first --
    while the fused first layer takes its rows as [xyz (3) | features (c)] like sample_and_group.  The first-layer W of a
    scale is therefore stored with the three xyz rows FIRST; reference_rows() / from_reference_rows() convert."""

    def __init__(self, store, scope, npoint, radius_list, nsample_list, cin, mlp_list):
        self.npoint, self.cin = npoint, cin
        self.scales = [SAModule(store, scope, npoint, r, k, cin, mlp, prefix="conv%d_" % i)
                       for i, (r, k, mlp) in enumerate(zip(radius_list, nsample_list, mlp_list))]
        self.widths = [mlp[-1] for mlp in mlp_list]

    @staticmethod
    def reference_rows(w):
        """first-layer W stored here ([xyz | feat] rows) -> the reference's row order ([feat | xyz])."""
        return torch.cat([w[3:], w[:3]], 0)

    @staticmethod
    def from_reference_rows(w):
        return torch.cat([w[-3:], w[:-3]], 0)

    def forward(self, xyz, points, tape=None):
        """xyz (B,n,3), points (B,n,C) or None -> new_xyz (B,m,3), new_points (B,m,sum of the scales' last widths)."""
        fps_idx = tf_sampling.farthest_point_sample(self.npoint, xyz)  # utils.py:179
        subs, outs, new_xyz = [], [], None
        for sc in self.scales:
            sub = [] if tape is not None else None
            new_xyz, o, _ = sc.forward(xyz, points, tape=sub, geom=sc.geometry(xyz, fps_idx=fps_idx))
            outs.append(o)
            subs.append(sub[0] if sub else None)
        if tape is not None:
            tape.append(dict(op="sa_msg", module=self, subs=subs))
        return new_xyz, torch.cat(outs, -1)

    def backward(self, rec, g_out, need_feat_grad=True):
        """g_out (B,m,sum widths) -> d_points (B,n,C) or None (summed over the scales)."""
        d_feat, o = None, 0
        for sc, sub, w in zip(self.scales, rec["subs"], self.widths):
            d, _ = sc.backward(sub, g_out[..., o:o + w].contiguous(), need_feat_grad=need_feat_grad)
            o += w
            if d is not None:
                d_feat = d if d_feat is None else d_feat + d
        return d_feat


class FPModule:
    """pointnet_fp_module (utils.py:266-294): 3-NN inverse-distance interpolation + MLP."""

    def __init__(self, store, scope, cin1, cin2, mlp):
        self.mlp = make_mlp(store, scope, cin1 + cin2, mlp, "conv_")

    @staticmethod
    def geometry(xyz1, xyz2):
        """three_nn + inverse-distance weights (utils.py:278-282): feature independent."""
        dist, idx = tf_interpolate.three_nn(xyz1, xyz2)
        return M.attach_inverse(idx, xyz2.shape[1], always=tf_interpolate.GATHER_GRAD), tf_interpolate.three_nn_weights(dist)

    def forward(self, xyz1, xyz2, points1, points2, tape=None, geom=None):
        b, n1 = xyz1.shape[:2]
        idx, weight = geom if geom is not None else self.geometry(xyz1, xyz2)  # utils.py:278-282
        if points1 is not None:  # utils.py:283-286: the interpolation writes the concat [interpolated | points1] itself
            x = tf_interpolate.three_interpolate_concat(points2, idx, weight, points1)
        else:
            x = tf_interpolate.three_interpolate(points2, idx, weight)
        rows = b * n1
        recs = []
        z, pend = mlp_chain_forward(self.mlp, rows, ("dense", x.view(rows, -1)), recs)
        y = M.bn_relu(z, None, None, True, bn=pend)
        if tape is not None:
            tape.append(dict(op="fp", module=self, recs=recs, idx=idx, weight=weight, m=xyz2.shape[1],
                             c2=points2.shape[2], c1=0 if points1 is None else points1.shape[2], b=b, n1=n1))
        return y.view(b, n1, -1)

    def backward(self, rec, dy):
        """dy (B,n1,C) -> d_points1 (B,n1,c1) or None, d_points2 (B,m,c2)."""
        b, n1, c1, c2 = rec["b"], rec["n1"], rec["c1"], rec["c2"]
        d_x = mlp_chain_backward(rec["recs"], dy.reshape(b * n1, -1).contiguous(), "act", need_input_grad=True)
        # [d interpolated | d points1] are column slices of d_x, read in place: ThreeInterpolateGrad takes a row pitch, and
        # d_points1 is returned as a VIEW (its consumer adds it to another gradient: add_rows)
        d_x3 = d_x.view(b, n1, c2 + c1)
        d_p1 = d_x3[:, :, c2:] if c1 else None
        d_p2 = tf_interpolate.three_interpolate_grad_raw(rec["m"], rec["idx"], rec["weight"], d_x3[:, :, :c2])
        return d_p1, d_p2
