"""Thin torch-tensor front-end of the grouped-MLP entry points of the C ABI
(votenet_mlp_linear / votenet_bn_finalize / votenet_bn_relu_max / votenet_bn_relu).

These are the building blocks pointnet2.py composes into the reference's SA / FP layer MLPs
(utils.py:125-132,149-155,286-293).  No torch math happens here: tensors are device buffers.
"""
import ctypes

import torch

from . import _lib as L

# bench.py sets this to a list to bracket every GEMM launch with HIP events on the launch stream:
# entries are (start, end, kind, flops) with kind in {"linear_dense", "linear_gather", "wgrad_dense", "wgrad_gather"}.
PROFILE_EVENTS = None
PROFILE_SHAPES = None  # tools/gemm_table.py: not None -> entries also carry (rows, cin, cout, note)


class _Timed:
    def __init__(self, kind, flops, shape=None, limit=None):
        """limit: a HalfLayout whose piece count only the device knows -- the launch is sized for the upper bound and `flops` counts that
        bound; the event then carries (flops, count tensor, bound) and resolve_event() prices the rows that were really there."""
        self.kind, self.flops, self.shape = kind, flops, shape
        if limit is not None and getattr(limit, "nh_limit", None) is not None:
            self.flops = (flops, limit.nh_dev, BALL_PIECES * limit.G)

    def __enter__(self):
        if PROFILE_EVENTS is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if PROFILE_EVENTS is not None:
            self.e1.record()
            PROFILE_EVENTS.append((self.e0, self.e1, self.kind, self.flops) if PROFILE_SHAPES is None else (self.e0, self.e1, self.kind, self.flops, self.shape))


def resolve_event(ev):
    """A PROFILE_EVENTS entry with its flops as a number (reads the device's piece count where the launch was sized for an upper bound:
    a host synchronisation -- call it after the timed region)."""
    if isinstance(ev[3], tuple):
        fl, nh, bound = ev[3]
        ev = ev[:3] + (fl * float(nh.item()) / bound,) + tuple(ev[4:])
    return ev


class _StatsArena:
    """Zero-initialised scratch of one pass comes out of ONE device buffer cleared by ONE fill: the fp64 accumulators (BatchNorm
    statistics, backward reductions; int64 counters share the region) and the fp32 buffers that kernels accumulate into with
    atomics (per-point scatter targets of the grouping's backward, Gram matrices, padded weight-gradient scratch, the gradients of
    three_interpolate / gather_point) -- about 60 tiny fills per train step otherwise.  The owner (VoteNetHotPath.train_step, or
    forward / backward on their own) calls arena_begin() on the stream the layers run on; a nested begin (forward inside
    train_step) joins the enclosing pass.  Without an arena every request falls back to torch.zeros.
    Only for buffers that are produced AND consumed inside the pass, on the pass's stream or on streams that wait for it (the
    weight-gradient stream): the geometry prefetched for the NEXT step allocates its own.
    The fp32 region is sized by the previous pass's demand (the first pass of a shape falls back to torch.zeros)."""
    buf = None        # float64 storage: [ndoubles fp64 | cap32 / 2 doubles viewed as fp32]
    nd = 0            # doubles of the fp64 region
    off = 0           # next free double
    cap32 = 0         # floats of the fp32 region
    off32 = 0         # next free float
    want32 = 0        # floats requested in this pass (what the next pass's region must hold)
    zeroed32 = 0      # floats of the fp32 region this pass's fill cleared
    depth = 0
    active = False
    whole_step = False  # the pass that owns the arena spans forward AND backward (train_step, a captured stretch): a buffer the forward pass
    #                     fills may then be read by the backward pass (forward() and backward() on their own each clear the arena)


def arena_begin(device, ndoubles=1 << 17, whole_step=False):
    a = _StatsArena
    if a.active and a.buf is not None and a.buf.device == device:
        a.depth += 1  # a pass inside a pass (forward / backward inside train_step): one fill for both
        return
    need32 = (a.want32 + 1023) // 1024 * 1024
    if a.buf is None or a.buf.device != device or a.nd < ndoubles or a.cap32 < need32:
        a.nd, a.cap32 = ndoubles, max(need32, a.cap32)
        a.buf = torch.empty(a.nd + a.cap32 // 2, dtype=torch.float64, device=device)
    used32 = min(a.cap32, need32)
    a.buf[:a.nd + used32 // 2].zero_()
    a.zeroed32 = used32
    a.off = a.off32 = a.want32 = 0
    a.depth = 1
    a.active = True
    a.whole_step = bool(whole_step)


def arena_end():
    a = _StatsArena
    a.depth -= 1
    if a.depth <= 0:
        a.active = False
        a.depth = 0
        a.whole_step = False


def _zeros_f64(n, device):
    a = _StatsArena
    if a.active and a.buf.device == device and a.off + n <= a.nd:
        v = a.buf[a.off:a.off + n]
        a.off += (n + 1) & ~1  # keep 16-byte alignment
        return v
    return torch.zeros(n, dtype=torch.float64, device=device)


def _zeros_i64(shape, device):
    """Zero-initialised int64 counters from the fp64 region (same width)."""
    n = 1
    for d in shape:
        n *= int(d)
    a = _StatsArena
    if a.active and a.buf.device == device and a.off + n <= a.nd:
        v = a.buf[a.off:a.off + n].view(torch.int64).view(shape)
        a.off += (n + 1) & ~1
        return v
    return torch.zeros(shape, dtype=torch.int64, device=device)


def _zeros_f32(shape, device):
    """Zero-initialised fp32 scratch of the pass (see _StatsArena); torch.zeros outside a pass or beyond the region."""
    n = 1
    for d in shape:
        n *= int(d)
    a = _StatsArena
    if a.active and a.buf.device == device:
        n4 = (n + 3) & ~3  # 16-byte aligned carve-outs
        a.want32 += n4
        if a.off32 + n4 <= a.zeroed32:
            v = a.buf[a.nd:].view(torch.float32)[a.off32:a.off32 + n].view(shape)
            a.off32 += n4
            return v
    return torch.zeros(shape, dtype=torch.float32, device=device)


# Bit-reproducible training (set_deterministic): weight gradients through per-workgroup partial tiles + an ordered reduction,
# the scatter-adds of the backward pass (GroupPointGrad, ThreeInterpolateGrad) as gather-sums over the groupings' inverse index
# (csr.hip) -- no fp32 atomics on any path; two identical passes give bit-identical gradients
# (tests/test_gpu_backward.py::test_training_gradients_are_bit_reproducible).  Off by default: it costs 17 % of the train step
# (7.83 -> 9.14 ms, same box; the gather-sums read the gradient rows in point order, 1.8 TB/s against 2.7 TB/s for the
# streaming pass with atomics), and the reference itself sums with atomics (tf_grouping_g.cu:74, tf_sampling_g.cu:187-189).
DETERMINISTIC = False
CONFIG_EPOCH = 0  # bumped by whoever flips a library-side switch the captured graphs cannot see (debug_switch): part of their keys


def debug_switch(name, *args):
    """L.lib().votenet_debug_<name>(*args) for switches that change WHICH kernels a launch runs (fast_bf3, gram_bf3, wgrad_bf3 ...): graphs
    captured before the call keep the old kernels, so the configuration epoch their keys carry moves on."""
    global CONFIG_EPOCH
    CONFIG_EPOCH += 1
    return getattr(L.lib(), "votenet_debug_" + name)(*args)


def set_deterministic(on=True):
    """Switch the bit-reproducible backward pass on / off; returns the previous setting."""
    global DETERMINISTIC
    prev, DETERMINISTIC = DETERMINISTIC, bool(on)
    return prev


# BatchNorm-backward coefficients in the tail of the kernel that reduced their sums (struct votenet_coef_tail): the last workgroup
# to finish does votenet_bn_backward_coef's work, so no launch sits between a reduction and its consumers on the step's
# dependent chain (23 launches of ~5 us per train step: 219 -> 196 launches).  Time: neutral (tools/ab_step.py mlp.COEF_TAIL False True,
# round 3, three alternations: 6.038 / 6.049 / 6.059 -> 6.044 / 6.052 / 6.050 ms; round 2: 6.997 -> 6.964) -- the small launches overlap
# the dispatch of their neighbours; what it saves is host enqueue time and queue slots.
# On since round 3.  The hand-off is the form MI355X_MICROARCH.md lists as valid for cross-workgroup data ("8-byte agent-scope atomics
# on both sides"): every contribution to the sums is a device-scope fp64 atomic (executed at the memory side, coherent across XCDs);
# each thread drains its own atomics (asm volatile s_waitcnt vmcnt(0): invisible to the pass that may drop a compiler-generated wait)
# before the workgroup's barrier and its ticket (a returning device-scope atomic); whoever draws the last ticket reads the sums with
# device-scope atomic loads, which bypass the CU's L1.  No plain store or load takes part, so no cache needs a release or an acquire.
COEF_TAIL = True
_tickets = {}


def _coef_tail(tail, c, device):
    """tail = (rows, gamma, dgamma, dbeta) or None -> (ctypes struct or None, coef tensor or None)."""
    if tail is None or not COEF_TAIL:
        return None, None
    rows, gamma, dgamma, dbeta = tail
    tk = _tickets.get(device)
    if tk is None:
        tk = _tickets[device] = [torch.zeros(64, dtype=torch.int32, device=device), 0]
    tk[1] = (tk[1] + 1) % 64  # kernels that may overlap must not share a ticket: 64 in rotation, each reset by its kernel
    coef = torch.empty(5 * c, dtype=torch.float32, device=device)
    t = L.CoefTail()
    t.ticket = tk[0].data_ptr() + 4 * tk[1]
    t.rows = rows
    t.gamma, t.coef = gamma.data_ptr(), coef.data_ptr()
    t.dgamma = dgamma.data_ptr() if dgamma is not None else None
    t.dbeta = dbeta.data_ptr() if dbeta is not None else None
    return t, coef


def _coef_after(tail, bn, sums, eps):
    """The separate launch, for a reduction that ran without a tail."""
    rows, gamma, dgamma, dbeta = tail
    return bn_backward_coef(rows, bn[0], bn[1], bn[2], bn[3], gamma, sums, dgamma, dbeta, eps)


def _wgrad_scratch(desc, rows, cin, cout, device):
    if not DETERMINISTIC:
        return None
    nf = L.lib().votenet_mlp_wgrad_scratch_floats(ctypes.byref(desc) if desc is not None else None, rows, cin, cout)
    return torch.empty(nf, dtype=torch.float32, device=device) if nf else None


def inverse_index(idx, n):
    """The inverse of a grouping: idx (b, m, k) int32 indices into n points per scene -> (order (b*m*k) int32, offsets (b*n + 1)
    int32): the slots (flat positions of idx) that reference point p are order[offsets[p]:offsets[p+1]], ascending (stable
    sort: one fixed summation order for the gather-sums of csr.hip).  Coordinates only: SAModule / FPModule.geometry build it
    with the grouping, on the geometry stream; it travels as idx._inv."""
    b = idx.shape[0]
    slots = idx.numel()
    if idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and 0 < n <= 8192 and b > 0 and slots // b <= 8 * n:
        # a small grouping with short lists (three_nn's taps: ~6 per target; NOT the ball queries of the deterministic mode, whose lists run
        # to hundreds -- one thread sorts a list by insertion): one launch, counts and cursors in LDS (votenet_inverse_index)
        order = torch.empty(slots, dtype=torch.int32, device=idx.device)
        offsets = torch.empty(b * n + 1, dtype=torch.int32, device=idx.device)
        with L.device_guard(idx.device):
            L.check(L.lib().votenet_inverse_index(b, slots // b, n, L.ptr(idx), L.ptr(order), L.ptr(offsets), L.stream_ptr()))
        return order, offsets
    flat = (idx.reshape(b, -1).to(torch.int64) + (torch.arange(b, device=idx.device, dtype=torch.int64) * n)[:, None]).reshape(-1)
    # unique 64-bit keys (point, slot): any sort gives the one order; no host synchronisation anywhere (bincount would need max())
    keys, _ = torch.sort(flat * slots + torch.arange(slots, device=idx.device, dtype=torch.int64))
    pts = torch.div(keys, slots, rounding_mode="floor")
    order = (keys - pts * slots).to(torch.int32)
    offsets = torch.searchsorted(pts, torch.arange(b * n + 1, device=idx.device, dtype=torch.int64)).to(torch.int32)
    return order, offsets


def attach_inverse(idx, n, always=False):
    """idx._inv = inverse_index(idx, n) in the deterministic mode -- or always: the taps of three_interpolate, whose gradient is a
    gather-sum over the inverse in every mode (tf_interpolate.GATHER_GRAD: 25 -> 10 us at 1024 <- 512 against the atomics)."""
    if (DETERMINISTIC or always) and getattr(idx, "_inv", None) is None:
        idx._inv = inverse_index(idx, n)
    return idx


def _inverse_of(idx, n):
    inv = getattr(idx, "_inv", None)
    if inv is None or inv[1].numel() != idx.shape[0] * n + 1:
        inv = inverse_index(idx, n)
        idx._inv = inv
    return inv


def csr_gather_sum(src2d, inv, npts, weight=None, div=1):
    """out (npts, c) = for every point the sum, in slot order, of weight[slot] * src2d[slot // div] (votenet_csr_gather_sum).
    src2d: (rows, c) with unit column stride; its rows may be a column slice of a wider tensor (votenet_csr_gather_sum_pitched)."""
    c = src2d.shape[1]
    out = torch.empty((npts, c), dtype=torch.float32, device=src2d.device)
    with L.device_guard(src2d.device):
        if src2d.is_contiguous():
            L.check(L.lib().votenet_csr_gather_sum(npts, c, L.ptr(src2d), L.ptr(inv[0]), L.ptr(inv[1]), L.ptr(weight), div, L.ptr(out),
                                                   L.stream_ptr()))
        else:
            if src2d.stride(1) != 1 or src2d.stride(0) < c:
                raise L.InvalidArgumentError("csr_gather_sum expects rows with unit column stride")
            L.check(L.lib().votenet_csr_gather_sum_pitched(npts, c, L.ptr(src2d), src2d.stride(0), L.ptr(inv[0]), L.ptr(inv[1]), L.ptr(weight),
                                                           div, L.ptr(out), L.stream_ptr()))
    return out


BN_EPS = 1e-5  # TensorFlow / Tensorpack BatchNorm default epsilon (not in the reference tree; see oracle/oracle_mlp.c)


class PendingBN:
    """The BatchNorm of a layer output whose finalize has not been launched: raw sums + gamma / beta.  The kernel that
    consumes the layer output derives scale / shift in its prologue (struct votenet_bn_raw) and fills `out`
    (scale | shift | mean | var, kept for the backward pass); finalize() launches the stand-alone kernel instead."""

    def __init__(self, stats, gamma, beta, rows, eps=None, out=None):
        """out: the layer's persistent (4, c) block (ParamStore.bn_flat) -- the backward pass of the SAME step reads it, the next
        forward pass of the layer overwrites it; None: a fresh buffer."""
        self.stats, self.gamma, self.beta, self.rows = stats, gamma, beta, rows
        self.eps = BN_EPS if eps is None else eps
        self.out = out if out is not None else torch.empty((4, gamma.shape[0]), dtype=torch.float32, device=gamma.device)
        self.done = False

    scale = property(lambda self: self.out[0])
    shift = property(lambda self: self.out[1])
    mean = property(lambda self: self.out[2])
    var = property(lambda self: self.out[3])

    def raw(self):
        """ctypes struct for a consumer (None once the vectors exist: a second consumer reads them)."""
        r = L.BnRaw()
        r.stats, r.gamma, r.beta = self.stats.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr()
        r.rows, r.eps, r.out = self.rows, self.eps, self.out.data_ptr()
        self.done = True
        return r

    def finalize(self):
        if not self.done:
            c = self.gamma.shape[0]
            with L.device_guard(self.gamma.device):
                L.check(L.lib().votenet_bn_finalize(self.rows, c, L.ptr(self.stats), L.ptr(self.gamma), L.ptr(self.beta), float(self.eps),
                                                    L.ptr(self.out[0]), L.ptr(self.out[1]), L.ptr(self.out[2]), L.ptr(self.out[3]),
                                                    L.stream_ptr()))
            self.done = True
        return self.out[0], self.out[1]


class FrozenBN:
    """Inference-mode BatchNorm of a layer (the reference's BNReLU with is_training=False uses the moving averages):
    scale = gamma * rsqrt(moving_var + eps), shift = beta - moving_mean * scale, computed once per set of weights
    (VoteNetHotPath.inference_bn).  Same interface as a finalized PendingBN: every consumer takes `done` -> scale / shift."""
    done = True

    def __init__(self, out):
        self.out = out  # (2, c): scale | shift

    scale = property(lambda self: self.out[0])
    shift = property(lambda self: self.out[1])

    def finalize(self):
        return self.out[0], self.out[1]


def _desc_dense(x, scale=None, shift=None, relu=True, in_bn=None):
    d = L.MlpInput()
    if in_bn is not None:
        if in_bn.done:  # already finalized by an earlier consumer
            scale, shift = in_bn.scale, in_bn.shift
        else:
            d._raw = in_bn.raw()
            d.in_bn = ctypes.pointer(d._raw)
    d.x = x.data_ptr()
    d.in_scale = scale.data_ptr() if scale is not None else None
    d.in_shift = shift.data_ptr() if shift is not None else None
    d.in_relu = 1 if (relu and (scale is not None or bool(d.in_bn))) else 0
    return d


def _desc_gather(xyz, new_xyz, feat, idx):
    d = L.MlpInput()
    d.x = None
    d.xyz = xyz.data_ptr()
    d.new_xyz = new_xyz.data_ptr()
    d.feat = feat.data_ptr() if feat is not None else None
    d.idx = idx.data_ptr()
    d.b, d.n = xyz.shape[0], xyz.shape[1]
    d.m, d.nsample = idx.shape[1], idx.shape[2]
    d.c = feat.shape[2] if feat is not None else 0
    return d


# Matrices made on the fly (pool_dgrad's W diag(C) W^T) can get an image too: votenet_pool_dgrad_prepare_split writes it in the
# launch that forms the matrix (no extra launch), and the image is registered around the one GEMM that multiplies by it.  That GEMM
# alone is 0.253 -> 0.124 ms at sa2 on split operands -- and the train step is SLOWER with it, measured twice on the same box
# (tools/ab_step.py mlp.SPLIT_ADHOC False True, round 3: 6.07 / 6.12 / 6.13 -> 6.24 / 6.23 / 6.24 ms; round 2 with a separate image
# launch: 6.25 -> 6.35): in the step this GEMM runs beside the Gram matrix and the arg-max gather of the weight-gradient stream, all
# three streaming the same activation -- the phase is bound by HBM traffic, and the leaner fp32 kernel (37 KB LDS, 72 VGPRs against
# 54 KB, 112) shares the CUs better (tools/trace_ab.sh adhoc: main queue +38 us, the other queue -95 us, the step +0.1 ms).  Off.
SPLIT_ADHOC = False
SPLIT_ADHOC_ROWS = 65536
# Round 6: the same matrix as TWO fp16 pieces, scaled by powers of two chosen inside the launch that forms it (pool_bwd.hip:
# pool_dgrad_prepare_kernel): the activations it multiplies are forward operands, the matrix is brought into fp16's range exactly.  The
# GEMM then runs three fp16 MFMAs per product instead of the fp32 MFMA kernel's v_mfma_f32_32x32x2_f32 chain.
ADHOC_H2 = True


def split_eligible(cin, cout):
    """Shapes whose GEMMs can read a bf16 x 3 image of the weights (mlp_fast.hip, BF3): whole 16-row slabs, 64-column blocks."""
    return cin % 32 == 0 and cin <= 512 and cout % 64 == 0


class SplitImages:
    """bf16 x 3 images of a set of weight matrices (votenet_split_weights): one device buffer, one launch to (re)build all of them,
    registered with the library so that the GEMM entry points which receive one of the matrices read its image.  The owner calls
    refresh() after every change of the weights; close() (or deletion) withdraws the registrations."""

    def __init__(self, mats, pieces=3):
        """pieces = 3: bf16 x 3 (any operand); 2: fp16 x 2 (votenet_split_weights_h2: matrices that multiply FORWARD operands only)."""
        mats = [w for w in mats if w.dim() == 2 and w.is_contiguous() and split_eligible(*w.shape)]
        self.mats = mats
        self.nseg = len(mats)
        self.pieces = int(pieces)
        assert self.pieces in (2, 3)
        if not mats:
            return
        dev = mats[0].device
        total, offs = 0, []
        for w in mats:
            offs.append(total)
            total += w.shape[0] * w.shape[1] * 2 * self.pieces
            total = (total + 15) // 16 * 16
        self.buf = torch.empty(total, dtype=torch.uint8, device=dev)
        base = self.buf.data_ptr()
        table = []
        for w, o in zip(mats, offs):
            table += [w.data_ptr(), base + o, w.shape[0], w.shape[1]]
        self.table = torch.tensor(table, dtype=torch.int64, device=dev)
        self._reg = [(w.data_ptr(), w.shape[0], w.shape[1], base + o) for w, o in zip(mats, offs)]
        for wp, cin, cout, ip in self._reg:
            L.check(L.lib().votenet_register_split_weights_pieces(ctypes.c_void_p(wp), cin, cout, ctypes.c_void_p(ip), self.pieces))

    def refresh(self):
        if self.nseg:
            with L.device_guard(self.buf.device):
                fn = L.lib().votenet_split_weights_h2 if self.pieces == 2 else L.lib().votenet_split_weights
                L.check(fn(self.nseg, L.ptr(self.table), L.stream_ptr()))

    def close(self):
        for wp, cin, cout, _ in getattr(self, "_reg", []):
            L.lib().votenet_register_split_weights(ctypes.c_void_p(wp), cin, cout, None)
        self._reg = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


FORWARD_H2 = True    # the images of the FORWARD weight matrices as fp16 x 2 (three MFMAs per product instead of six; csrc/mlp_types.h: split2); the transposed copies the backward GEMMs read stay bf16 x 3 (gradients need fp32's range)
SPLIT_K = False      # the fused GEMMs of few row tiles (the static stretch: 2048-8192 rows) share an output tile's contraction between 2-4 workgroups (csrc/mlp_fast.hip, FastArgs::sk_ws)
_SK_TICKETS = {}     # device -> the zeroed ticket array registered with the library (kept alive here)


class _SplitK:
    """`with _SplitK(rows, cin, cout, device):` the fused-GEMM launch inside may split its contraction: a workspace of the size the
    library asks for (from the caching allocator -- inside a graph capture: the capture's pool) is armed for it and disarmed
    afterwards.  The ticket array is registered on first use, outside of any capture (the first step of a shape runs launch by launch)."""
    __slots__ = ("ws",)

    def __init__(self, rows, cin, cout, device):
        self.ws = None
        if not SPLIT_K or rows <= 0 or rows > 16384:
            return
        if device not in _SK_TICKETS:
            if torch.cuda.is_current_stream_capturing():
                return
            t = torch.zeros(1 << 16, dtype=torch.int32, device=device)
            torch.cuda.current_stream(device).synchronize()  # zeroed before any stream may launch a split GEMM
            L.check(L.lib().votenet_mlp_split_k_tickets(L.ptr(t), t.numel()))
            _SK_TICKETS[device] = t
        n = L.lib().votenet_mlp_split_k_floats(rows, cin, cout)
        if n > 0:
            self.ws = torch.empty(n, dtype=torch.float32, device=device)

    def __enter__(self):
        if self.ws is not None:
            L.lib().votenet_mlp_split_k_arm(L.ptr(self.ws), self.ws.numel())
        return self

    def __exit__(self, *exc):
        if self.ws is not None:
            L.lib().votenet_mlp_split_k_arm(None, 0)
        return False


def linear_dense(x, w, bias=None, in_scale=None, in_shift=None, in_relu=True, want_stats=True, in_bn=None):
    """z = act(x) @ w + bias with act = relu(x*in_scale+in_shift) folded into the load (or identity).
    x (rows, cin) f32 -> z (rows, cout), stats (2*cout) f64 [column sums of z, of z*z] or None."""
    x = L.dev_f32(x, "mlp_linear x", 2)
    w = L.dev_f32(w, "mlp_linear w", 2)
    rows, cin = x.shape
    if w.shape[0] != cin:
        raise L.InvalidArgumentError("mlp_linear: w has %d rows, input has %d channels" % (w.shape[0], cin))
    cout = w.shape[1]
    z = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    stats = _zeros_f64(2 * cout, x.device) if want_stats else None
    d = _desc_dense(x, in_scale, in_shift, in_relu, in_bn)
    with L.device_guard(x.device), _SplitK(rows, cin, cout, x.device), \
            _Timed("linear_dense", 2.0 * rows * cin * cout, (rows, cin, cout, "fwd" + ("+bn" if (in_scale is not None or in_bn is not None) else ""))):
        L.check(L.lib().votenet_mlp_linear(ctypes.byref(d), rows, cin, cout, L.ptr(w), L.ptr(bias), L.ptr(z), L.ptr(stats),
                                           L.stream_ptr()))
    return z, stats


def linear_gather(xyz, new_xyz, feat, idx, w, bias=None, want_stats=True):
    """First SA-layer linear with the sample_and_group concat (utils.py:50-57) folded into the load:
    row (b,j,k) = [xyz[b,idx]-new_xyz[b,j] (3), feat[b,idx] (c)].  -> z (b*m*nsample, cout), stats."""
    xyz = L.dev_f32(xyz, "mlp_linear xyz", 3, 3)
    new_xyz = L.dev_f32(new_xyz, "mlp_linear new_xyz", 3, 3)
    idx = L.dev_i32(idx, "mlp_linear idx", 3)
    feat = L.dev_f32(feat, "mlp_linear feat", 3) if feat is not None else None
    w = L.dev_f32(w, "mlp_linear w", 2)
    b, m, k = idx.shape
    c = feat.shape[2] if feat is not None else 0
    cin, cout = 3 + c, w.shape[1]
    if w.shape[0] != cin:
        raise L.InvalidArgumentError("mlp_linear: w has %d rows, grouped input has %d channels" % (w.shape[0], cin))
    rows = b * m * k
    z = torch.empty((rows, cout), dtype=torch.float32, device=xyz.device)
    stats = _zeros_f64(2 * cout, xyz.device) if want_stats else None
    d = _desc_gather(xyz, new_xyz, feat, idx)
    with L.device_guard(xyz.device), _Timed("linear_gather", 2.0 * rows * cin * cout, (rows, cin, cout, "gather")):
        L.check(L.lib().votenet_mlp_linear(ctypes.byref(d), rows, cin, cout, L.ptr(w), L.ptr(bias), L.ptr(z), L.ptr(stats),
                                           L.stream_ptr()))
    return z, stats


# ---- first SA layer assembled inside its consumers (csrc/assemble.hip): z0 = P[idx] + dxyz W[0:3] never stored ----
def assembled_supported(rows, c0, c1):
    return rows > 0 and rows % 128 == 0 and rows < 2 ** 31 and c0 % 64 == 0 and c0 <= 512 and c1 % 64 == 0


def assemble_rows(xyz, new_xyz, idx, want_sums=True, pts_cnt=None, in_pass=False):
    """-> geo (rows, 4) f32 = (dx, dy, dz, bits(scene*n + idx)), cntv (b*n, 4) i64 and moments (9,) f64 or None, None."""
    b, m, k = idx.shape
    n = xyz.shape[1]
    geo = torch.empty((b * m * k, 4), dtype=torch.float32, device=xyz.device)
    # not from the per-step arenas: geometry is computed one or two steps ahead of its use
    # count, sum dxyz in fixed point 2^-32.  in_pass: called inside the pass that consumes the result (the proposal layer, whose votes
    # exist only then): the counters come out of the pass's arena; geometry computed ahead for a later step allocates its own
    cntv = (_zeros_i64((b * n, 4), xyz.device) if in_pass else torch.zeros((b * n, 4), dtype=torch.int64, device=xyz.device)) if want_sums else None
    mom = (_zeros_f64(10, xyz.device)[:9] if in_pass else torch.zeros(9, dtype=torch.float64, device=xyz.device)) if want_sums else None
    with L.device_guard(xyz.device):
        L.check(L.lib().votenet_assemble_rows(b, n, m, k, L.ptr(xyz), L.ptr(new_xyz), L.ptr(idx), L.ptr(pts_cnt), L.ptr(geo), L.ptr(cntv), L.ptr(mom),
                                              L.stream_ptr()))
    return geo, cntv, mom


# ---- PIECE layout (csrc/half.hip): the grouped MLP of a level without (most of) the rows that are copies of slot 0 ----
PIECE = 16           # rows per piece (votenet_half_piece_rows(), checked when the first layout is built)
BALL_PIECES = 64 // PIECE
_nh_ring = None  # pinned ints the piece counts are copied into (one slot per layout, reused round-robin)
_nh_turn = 0


_slot_source = None


class layout_slots:
    """Context: HalfLayout.host_slot() hands out the elements of `pinned` (a pinned int32 tensor its owner keeps alive), in order."""

    def __init__(self, pinned):
        self.pinned, self.turn = pinned, 0

    def _next(self):
        if self.turn >= self.pinned.numel():
            raise L.VotenetError("layout_slots: more piece layouts than slots (%d)" % self.pinned.numel())
        self.turn += 1
        return self.pinned[self.turn - 1:self.turn]

    def __enter__(self):
        global _slot_source
        self._saved, _slot_source = _slot_source, self._next
        return self

    def __exit__(self, *exc):
        global _slot_source
        _slot_source = self._saved


class HalfLayout:
    """Row layout of one level (include/votenet_hip.h, 'PIECE layout'): G centres, nh pieces of PIECE = 16 compact rows.
    pos (G, 3), hc / wh (nh,), geo (16 nh, 4).  The count nh is produced on the device by the geometry chain -- typically a step ahead of
    its use -- and copied to pinned host memory; resolve() waits for that copy (a no-op once it has landed) and trims the views."""

    @staticmethod
    def host_slot():
        """A pinned int the kernel that makes the layout writes the count into (one per layout, reused round-robin)."""
        global _nh_ring, _nh_turn
        if _slot_source is not None:  # a captured geometry chain owns its slots (layout_slots): the ring's are handed out again
            return _slot_source()
        if _nh_ring is None:
            if L.lib().votenet_half_piece_rows() != PIECE:
                raise L.VotenetError("libvotenet_hip.so was built for pieces of %d rows, the host code for %d" % (L.lib().votenet_half_piece_rows(), PIECE))
            _nh_ring = torch.zeros(256, dtype=torch.int32).pin_memory()
        slot = _nh_ring[_nh_turn:_nh_turn + 1]
        _nh_turn = (_nh_turn + 1) % 256
        return slot

    def __init__(self, G, pos, hc, wh, nh_dev, device_count=False, slot=None):
        """device_count: the host never learns the count (a level whose geometry is made inside the step: waiting for it would drain
        the queue).  Every row tensor then has the maximal 64 G rows, launches are sized for that, and the kernels stop at the count
        they read on the device (nh_limit: the pointer the *_half entries take)."""
        global _nh_ring, _nh_turn
        self.G, self.pos, self._hc, self._wh, self.nh_dev = G, pos, hc, wh, nh_dev
        self._geo = self._u8 = None  # the level's compact rows: geo records (assembled first layer) or u8 (narrow first layer)
        self._order = None           # the compact rows bucketed by the point they gather (half_sort_rows)
        self.nh_limit = nh_dev if device_count else None
        if device_count:
            self.nh = None
            self._ev = None
            return
        self._slot = slot  # written by votenet_half_groups itself (pinned memory is mapped into the device's address space)
        if torch.cuda.is_current_stream_capturing():
            self._ev = None  # a layout inside a captured chain: rearm() gives it the event that follows each replay
        else:
            self._ev = torch.cuda.Event()
            self._ev.record()
        self.nh = None

    def rearm(self, ev):
        """The layout's buffers have been (or are being) rewritten by a replay of the graph that holds the kernels that made it: forget
        the count; ev = an event recorded after the replay."""
        self.nh = None
        self._ev = ev

    def tensors(self):
        return [t for t in (self.pos, self._hc, self._wh, self.nh_dev, self._geo, self._u8, self._order) if t is not None]

    def resolve(self):
        if self.nh is None and self.nh_limit is not None:  # the device's secret: the upper bound stands in on the host
            self.nh = BALL_PIECES * self.G
            self.hc, self.wh, self.geo, self.u8, self.order = self._hc, self._wh, self._geo, self._u8, self._order
        if self.nh is None:
            self._ev.synchronize()
            self.nh = int(self._slot.item())
            if not (self.G <= self.nh <= BALL_PIECES * self.G and self.nh % (128 // PIECE) == 0):
                raise L.VotenetError("piece layout: bad count %d for %d centres" % (self.nh, self.G))
            self.hc, self.wh = self._hc[:self.nh], self._wh[:self.nh]
            self.geo = self._geo[:self.nh * PIECE] if self._geo is not None else None
            self.u8 = self._u8[:self.nh * PIECE] if self._u8 is not None else None
            self.order = self._order[:self.nh * PIECE] if self._order is not None else None
        return self

    @property
    def rows(self):
        return self.resolve().nh * PIECE

    def true_count(self):
        """The number of pieces (a host synchronisation when only the device knows it: tests, debugging)."""
        return int(self.nh_dev.item()) if self.nh_limit is not None else self.resolve().nh

    def full_index(self):
        """(64 G,) for every slot of the full layout the compact row that holds it; a slot of a dropped piece -> the ball's slot 0."""
        self.resolve()
        G, dev = self.G, self.pos.device
        c = torch.arange(G, device=dev)[:, None]
        s = torch.arange(64, device=dev)[None, :]
        j = s // PIECE
        posx = torch.cat([torch.zeros(G, 1, dtype=torch.long, device=dev), self.pos.view(G, BALL_PIECES - 1).long()], 1)  # piece 0: at c
        p = torch.gather(posx, 1, j.expand(G, 64))
        src = torch.where(j == 0, c * PIECE + s, torch.where(p >= 0, (G + p) * PIECE + s % PIECE, c * PIECE))
        return src.reshape(-1)

    def full_rows(self, t):
        """A compact row tensor (16 nh, c) laid out as the full layout (64 G, c): slot s of centre c; a dropped slot holds a copy of
        slot 0.  (Row VALUES of the forward pass; gradients per compact row are totals and do not expand this way.)  Tests, debugging."""
        return t[self.full_index()]

    def row_weights(self):
        """(16 nh,) the number of full-layout rows every compact row stands for (row 0 of a ball's first piece: also its dropped copies)."""
        self.resolve()
        w = torch.ones(self.nh, PIECE, device=self.wh.device)
        w[:, 0] = self.wh
        return w.reshape(-1)


def half_groups(pts_cnt, device_count=False):
    """pts_cnt (b, m) int32 -> HalfLayout (count still on its way to the host; device_count: never sent, see HalfLayout)."""
    G = pts_cnt.numel()
    dev = pts_cnt.device
    ints = torch.empty((BALL_PIECES - 1) * G + BALL_PIECES * G + 1, dtype=torch.int32, device=dev)
    pos, hc, nh = ints[:(BALL_PIECES - 1) * G], ints[(BALL_PIECES - 1) * G:-1], ints[-1:]
    wh = torch.empty(BALL_PIECES * G, dtype=torch.float32, device=dev)
    slot = None if device_count else HalfLayout.host_slot()
    with L.device_guard(dev):
        L.check(L.lib().votenet_half_groups(G, L.ptr(pts_cnt), L.ptr(pos), L.ptr(hc), L.ptr(wh), L.ptr(nh), L.ptr(slot), L.stream_ptr()))
        return HalfLayout(G, pos, hc, wh, nh, device_count, slot)


def assemble_rows_half(xyz, new_xyz, idx, pts_cnt, half):
    """assemble_rows on the piece layout: -> geo buffer (64 G, 4) of which half.resolve().geo is the written part, cntv, moments."""
    b, m, k = idx.shape
    n = xyz.shape[1]
    if k != 64:
        raise L.InvalidArgumentError("assemble_rows_half expects nsample == 64")
    geo = torch.empty((b * m * 64, 4), dtype=torch.float32, device=xyz.device)
    # ONE zero-filled buffer: the per-point counters (b n x 4 int64), the moments (9 f64) and the rows-per-point counts of the bucketing
    npts = b * n
    nz = npts * 4 + 16 + (npts + 1) // 2
    if half.nh_limit is not None and _StatsArena.active and _StatsArena.whole_step:
        # geometry made inside the pass it serves (the proposal module): the pass's one zero fill covers it -- when that pass spans the
        # backward too, which reads cntv and the moments again (the first layer's backward decomposed over the points)
        zero = _zeros_i64((nz,), xyz.device)
    else:                          # computed ahead, on a geometry stream, for a later pass: its own fill
        zero = torch.zeros(nz, dtype=torch.int64, device=xyz.device)
    cntv = zero[:npts * 4].view(npts, 4)
    mom = zero[npts * 4:npts * 4 + 9].view(torch.float64)
    work = zero[npts * 4 + 16:].view(torch.int32)[:npts]
    with L.device_guard(xyz.device):
        L.check(L.lib().votenet_assemble_rows_half(b, n, m, L.ptr(half.nh_dev), L.ptr(xyz), L.ptr(new_xyz), L.ptr(idx), L.ptr(pts_cnt),
                                                   L.ptr(half._hc), L.ptr(geo), L.ptr(cntv), L.ptr(mom), L.ptr(work), L.stream_ptr()))
    half._geo, half._work = geo, work
    return geo, cntv, mom


def narrow_rows_half(xyz, new_xyz, feat, idx, pts_cnt, half):
    """narrow_rows on the piece layout: -> u8 buffer (64 G, 8) of which half.resolve().u8 is the written part, moments (72,)."""
    b, m, k = idx.shape
    n = xyz.shape[1]
    c = feat.shape[2] if feat is not None else 0
    if k != 64:
        raise L.InvalidArgumentError("narrow_rows_half expects nsample == 64")
    u8 = torch.empty((b * m * 64, 8), dtype=torch.float32, device=xyz.device)
    mom = torch.zeros(72, dtype=torch.float64, device=xyz.device)
    with L.device_guard(xyz.device):
        L.check(L.lib().votenet_narrow_rows_half(b, n, m, c, L.ptr(half.nh_dev), L.ptr(xyz), L.ptr(new_xyz), L.ptr(feat), L.ptr(idx),
                                                 L.ptr(pts_cnt), L.ptr(half._hc), L.ptr(u8), L.ptr(mom), L.stream_ptr()))
    half._u8 = u8
    return u8, mom


def half_sort_rows(half, npts):
    """Bucket the layout's compact rows by the point they gather (geo must be assembled): half.order.  Geometry only."""
    dev = half._geo.device
    work, half._work = getattr(half, "_work", None), None  # the rows per point, counted by assemble_rows_half (consumed here)
    counted = work is not None
    if not counted:
        work = torch.empty(npts, dtype=torch.int32, device=dev)
    order = torch.empty(half.G * 64, dtype=torch.int32, device=dev)
    with L.device_guard(dev):
        L.check(L.lib().votenet_half_sort_rows(npts, half.G, L.ptr(half.nh_dev), L.ptr(half._geo), L.ptr(work), 1 if counted else 0, L.ptr(order),
                                               L.stream_ptr()))
    half._order = order
    if half.nh is not None:  # the count is known already (or stands at its upper bound)
        half.order = order[:half.nh * PIECE]
    return order


def group_linear_backward_half(half, b, n, P, wx, da, coef, relu, dw_xyz):
    """group_linear_backward_assembled on the piece layout (da = total gradients per compact row) -> S (b, n, cout): the first layer's
    scatter to the points over the rows bucketed by point (half_sort_rows: with the geometry, or here if it has not run)."""
    cout = P.shape[1]
    if getattr(half, "order", None) is None:
        half_sort_rows(half, b * n)
    S = _zeros_f32((b, n, cout), P.device)
    with L.device_guard(P.device):
        L.check(L.lib().votenet_group_linear_backward_sorted(half.nh, cout, L.ptr(half.order), L.ptr(half.geo), L.ptr(half.wh), L.ptr(P),
                                                             L.ptr(wx), L.ptr(da), L.ptr(coef), 1 if relu else 0, L.ptr(S), L.ptr(dw_xyz),
                                                             L.ptr(half.nh_limit), L.stream_ptr()))
    return S


_masked_slots = None


def dgrad_bn_half(z, coef, relu, wT, da, half):
    """votenet_mlp_dgrad_bn_half: da_prev (rows, cout) = dz wT with dz rebuilt from (da, z, coef) on the piece layout (totals; the affine
    part weighted by half.wh) -- a plain GEMM, nothing in its epilogue but the store."""
    rows, c = z.shape
    cout = wT.shape[1]
    out = torch.empty((rows, cout), dtype=torch.float32, device=z.device)
    with L.device_guard(z.device), _Timed("linear_dense", 2.0 * rows * c * cout, (rows, c, cout, "dgrad_bn half"), limit=half):
        L.check(L.lib().votenet_mlp_dgrad_bn_half(rows, c, cout, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(wT), L.ptr(out),
                                                  L.ptr(half.wh), L.ptr(half.nh_limit), L.stream_ptr()))
    return out


def group_linear_backward_decomposed(half, b, n, P, wx, da, bn, relu, tail, cntv, mom, dw_xyz, eps=BN_EPS, defer=None):
    """The first layer's backward decomposed over the points (csrc/half.hip: group_linear_bwd_masked_kernel): da = the total gradients per
    compact row of the layer's ACTIVATION (what votenet_mlp_dgrad_bn_half stored); bn = (scale, shift, mean, var) of the layer; tail as
    _coef_tail takes it.  -> (S (b, n, cout), coef of the layer).  dw_xyz += the coordinate rows of the weight gradient (a one-workgroup
    launch handed to defer(fn, *tensors) -- pointnet2.on_wgrad_stream -- when given: it is not on the chain)."""
    cout = P.shape[1]
    dev = P.device
    if getattr(half, "order", None) is None:
        half_sort_rows(half, b * n)
    sc, sh, me, va = bn
    global _masked_slots, COEF_TAIL
    if _masked_slots is None:
        _masked_slots = int(L.lib().votenet_group_linear_backward_masked_slots())
    nparts = _masked_slots
    S = _zeros_f32((b, n, cout), dev)
    small = _zeros_f32((3 + 3 * nparts, cout), dev)  # ug (3, cout: written whole) | vp (slots, 3, cout): accumulated
    ug, vp = small[:3], small[3:]
    zeros = _zeros_f64(2 * cout + _masked_slots * 5 * cout, dev)
    sums, part = zeros[:2 * cout], zeros[2 * cout:]
    # the kernel's LAST workgroup folds the partial sums (a ticket): the coefficient vector comes out of that tail whatever COEF_TAIL says
    prev_ct, COEF_TAIL = COEF_TAIL, True
    try:
        t, coef = _coef_tail(tail, cout, dev)
    finally:
        COEF_TAIL = prev_ct
    with L.device_guard(dev):
        L.check(L.lib().votenet_group_linear_backward_masked(half.nh, cout, L.ptr(half.order), L.ptr(half.geo), L.ptr(P), L.ptr(wx), L.ptr(da),
                                                             L.ptr(sc), L.ptr(sh), L.ptr(me), L.ptr(va), eps, 1 if relu else 0, L.ptr(S),
                                                             L.ptr(ug), L.ptr(sums), L.ptr(part), ctypes.byref(t), L.ptr(half.nh_limit),
                                                             L.stream_ptr()))
        L.check(L.lib().votenet_assembled_point_grad(b * n, cout, L.ptr(P), L.ptr(cntv), L.ptr(wx), L.ptr(coef), L.ptr(S), L.ptr(vp),
                                                     L.stream_ptr()))

    def finish():
        with L.device_guard(dev):
            L.check(L.lib().votenet_assembled_wx_finish(cout, L.ptr(coef), L.ptr(ug), L.ptr(vp), nparts, L.ptr(mom), L.ptr(wx), L.ptr(dw_xyz),
                                                        L.stream_ptr()))
    if defer is not None:
        defer(finish, coef, small, mom)
    else:
        finish()
    return S, coef


def half_centre_sums(half, P, wx, da, coef, relu):
    """-> (G, cout): MINUS the sum of the total gradients dz0 over every centre's compact rows (what the centre's coordinates receive
    through dxyz = xyz[idx] - new_xyz, before the product with W[0:3]^T): votenet_half_centre_sums."""
    cout = P.shape[1]
    T = torch.empty((half.G, cout), dtype=torch.float32, device=P.device)
    with L.device_guard(P.device):
        L.check(L.lib().votenet_half_centre_sums(half.G, cout, L.ptr(half.pos), L.ptr(half.geo), L.ptr(half.wh), L.ptr(P), L.ptr(wx), L.ptr(da),
                                                 L.ptr(coef), 1 if relu else 0, L.ptr(T), L.stream_ptr()))
    return T


def assemble_stats(P, cntv, wx, mom):
    """BatchNorm statistics (2*c0 f64) of the never-stored z0 from one pass over the points."""
    npts, c0 = P.shape
    stats = _zeros_f64(2 * c0, P.device)
    with L.device_guard(P.device):
        L.check(L.lib().votenet_assemble_stats(npts, c0, L.ptr(P), L.ptr(cntv), L.ptr(wx), L.ptr(mom), L.ptr(stats), L.stream_ptr()))
    return stats


def assemble_z0(geo, P, wx):
    """The first-layer output the product path never stores, with the kernels' own arithmetic (tests)."""
    z0 = torch.empty((geo.shape[0], P.shape[1]), dtype=torch.float32, device=P.device)
    with L.device_guard(P.device):
        L.check(L.lib().votenet_assemble_z0(geo.shape[0], P.shape[1], L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(z0), L.stream_ptr()))
    return z0


def assembled_linear(geo, P, wx, w, bias, in_bn, in_relu=True, want_stats=True, half=None):
    """Second layer over the assembled first-layer output: z = relu(bn0(P[prow] + dxyz wx)) w + bias -> z (rows, cout), stats.
    half: geo is a HalfLayout's compact rows; the statistics weigh the rows that stand for a dropped half."""
    rows, c0, cout = geo.shape[0], P.shape[1], w.shape[1]
    z = torch.empty((rows, cout), dtype=torch.float32, device=P.device)
    stats = _zeros_f64(2 * cout, P.device) if want_stats else None
    scale = shift = raw = None
    if in_bn.done:
        scale, shift = in_bn.scale, in_bn.shift
    else:
        raw = in_bn.raw()
    if half is not None:
        with L.device_guard(P.device), _Timed("linear_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "fwd+bn assembled half"), limit=half):
            L.check(L.lib().votenet_assembled_linear_half(rows, c0, cout, L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(scale), L.ptr(shift),
                                                          ctypes.byref(raw) if raw is not None else None, 1 if in_relu else 0, L.ptr(w),
                                                          L.ptr(bias), L.ptr(z), L.ptr(stats), L.ptr(half.wh), L.ptr(half.nh_limit), L.stream_ptr()))
        return z, stats
    with L.device_guard(P.device), _Timed("linear_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "fwd+bn assembled")):
        L.check(L.lib().votenet_assembled_linear(rows, c0, cout, L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(scale), L.ptr(shift),
                                                 ctypes.byref(raw) if raw is not None else None, 1 if in_relu else 0, L.ptr(w),
                                                 L.ptr(bias), L.ptr(z), L.ptr(stats), L.stream_ptr()))
    return z, stats


def assembled_wgrad_bn(geo, P, wx, in_scale, in_shift, in_relu, z, coef, relu, da, dw, half=None):
    """dw (c0, cout) += relu(bn0(z0))^T dz1 with z0 rebuilt in the loader, dz1 = BatchNorm-backward(da, z, coef)."""
    rows, c0, cout = geo.shape[0], P.shape[1], z.shape[1]
    if half is not None:
        with L.device_guard(P.device), _Timed("wgrad_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "wgrad_bn assembled half"), limit=half):
            L.check(L.lib().votenet_assembled_wgrad_bn_half(rows, c0, cout, L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(in_scale), L.ptr(in_shift),
                                                            1 if in_relu else 0, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0,
                                                            L.ptr(half.wh), L.ptr(dw), L.ptr(half.nh_limit), L.stream_ptr()))
        return
    scr = _wgrad_scratch(None, rows, c0, cout, P.device)
    with L.device_guard(P.device), _Timed("wgrad_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "wgrad_bn assembled")):
        L.check(L.lib().votenet_assembled_wgrad_bn(rows, c0, cout, L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(in_scale), L.ptr(in_shift),
                                                   1 if in_relu else 0, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(dw),
                                                   L.ptr(scr), L.stream_ptr()))


def assembled_dgrad_bn_reduce(z, coef, relu, wT, da, geo, P, wx, below, eps=BN_EPS, below_tail=None, half=None):
    """dgrad_bn(..., below=...) for an assembled layer below: -> (da_prev, sums) or, with below_tail, (da_prev, coef of that layer)."""
    rows, c = z.shape
    cout = wT.shape[1]
    bsc, bsh, bme, bva, brelu = below
    out = torch.empty((rows, cout), dtype=torch.float32, device=z.device)
    sums = _zeros_f64(2 * cout, z.device)
    t, coef_b = _coef_tail(below_tail, cout, z.device)
    if half is not None:
        with L.device_guard(z.device), _Timed("linear_dense", 2.0 * rows * c * cout, (rows, c, cout, "dgrad_bn_reduce assembled half"), limit=half):
            L.check(L.lib().votenet_assembled_dgrad_bn_reduce_half(rows, c, cout, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(wT),
                                                                   L.ptr(out), L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(bsc), L.ptr(bsh),
                                                                   L.ptr(bme), L.ptr(bva), eps, 1 if brelu else 0, L.ptr(sums),
                                                                   ctypes.byref(t) if t is not None else None, L.ptr(half.wh), L.ptr(half.nh_limit),
                                                                   L.stream_ptr()))
    else:
        with L.device_guard(z.device), _Timed("linear_dense", 2.0 * rows * c * cout, (rows, c, cout, "dgrad_bn_reduce assembled")):
            L.check(L.lib().votenet_assembled_dgrad_bn_reduce(rows, c, cout, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(wT),
                                                              L.ptr(out), L.ptr(geo), L.ptr(P), L.ptr(wx), L.ptr(bsc), L.ptr(bsh), L.ptr(bme),
                                                              L.ptr(bva), eps, 1 if brelu else 0, L.ptr(sums),
                                                              ctypes.byref(t) if t is not None else None, L.stream_ptr()))
    if below_tail is None:
        return out, sums
    return out, (coef_b if coef_b is not None else _coef_after(below_tail, (bsc, bsh, bme, bva), sums, eps))


def group_linear_backward_assembled(xyz, new_xyz, idx, pts_cnt, P, wx, da, coef, relu, dw_xyz, want_dz=False):
    """group_linear_backward for a first layer whose output was never stored: z rebuilt from P (b*n, cout) and wx."""
    b, m, k = idx.shape
    n, cout = xyz.shape[1], P.shape[1]
    dz = torch.empty((b * m * k, cout), dtype=torch.float32, device=P.device) if want_dz else None
    S = _zeros_f32((b, n, cout), P.device)
    with L.device_guard(P.device):
        L.check(L.lib().votenet_group_linear_backward_assembled(b, n, m, k, cout, L.ptr(xyz), L.ptr(new_xyz), L.ptr(idx), L.ptr(pts_cnt),
                                                                L.ptr(P), L.ptr(wx), L.ptr(da), L.ptr(coef), 1 if relu else 0, L.ptr(S),
                                                                L.ptr(dw_xyz), L.ptr(dz), L.stream_ptr()))
    return S, dz


# ---- NARROW first layer (csrc/narrow.hip): 3 + c <= 8 grouped channels, z0 never stored ----
def narrow_supported(rows, k0, c0, c1):
    """Shapes the narrow-first-layer kernels serve (include/votenet_hip.h): k0 = 3 + c grouped input channels, c0 / c1 = widths
    of the first / second layer."""
    return 3 <= k0 <= 8 and rows > 0 and rows % 128 == 0 and rows < 2 ** 31 and c0 % 64 == 0 and c0 <= 128 and (c1 % 64 == 0)


def narrow_rows(xyz, new_xyz, feat, idx, want_moments=True):
    """-> u8 (rows, 8) f32 = (xyz[idx]-new_xyz | feat[idx] | 0...), moments (72,) f64 or None (votenet_narrow_rows)."""
    b, m, k = idx.shape
    n = xyz.shape[1]
    c = feat.shape[2] if feat is not None else 0
    u8 = torch.empty((b * m * k, 8), dtype=torch.float32, device=xyz.device)
    # not from the per-step statistics arena: the geometry of a batch is computed one or two steps ahead of its use
    mom = torch.zeros(72, dtype=torch.float64, device=xyz.device) if want_moments else None
    with L.device_guard(xyz.device):
        L.check(L.lib().votenet_narrow_rows(b, n, m, k, c, L.ptr(xyz), L.ptr(new_xyz), L.ptr(feat), L.ptr(idx), L.ptr(u8), L.ptr(mom),
                                            L.stream_ptr()))
    return u8, mom


def narrow_z0(u8, w0, b0):
    """The first-layer output the product path never stores, with the kernels' own arithmetic (tests: the device's active set)."""
    k0, c0 = w0.shape
    z0 = torch.empty((u8.shape[0], c0), dtype=torch.float32, device=u8.device)
    with L.device_guard(u8.device):
        L.check(L.lib().votenet_narrow_z0(u8.shape[0], k0, c0, L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(z0), L.stream_ptr()))
    return z0


def narrow_stats(rows, mom, w0, b0):
    """BatchNorm statistics (2*c0 f64: sum z0, sum z0^2) of the never-stored first-layer output, from the moments."""
    k0, c0 = w0.shape
    stats = torch.empty(2 * c0, dtype=torch.float64, device=w0.device)
    with L.device_guard(w0.device):
        L.check(L.lib().votenet_narrow_stats(rows, k0, c0, L.ptr(mom), L.ptr(w0), L.ptr(b0), L.ptr(stats), L.stream_ptr()))
    return stats


NARROW_MASK = True  # the narrow layer's ReLU mask is recorded by the forward pass and read by the input-gradient GEMM's epilogue (EPI 7 of mlp_fast.hip) instead of a rebuild of z0 per accumulator element


def narrow_mask_supported(rows, c0):
    return rows % 128 == 0 and c0 % 64 == 0


def narrow_linear(u8, w0, b0, w, bias, in_bn, in_relu=True, want_stats=True, half=None, want_mask=False):
    """Second layer over the rebuilt first-layer output: z = relu(bn0(u8 w0 + b0)) w + bias -> z (rows, cout), stats.
    want_mask: -> z, stats, mask (rows, c0 / 16) int16: bit k % 16 of word [row][k / 16] = [relu(bn0(z0[row, k])) > 0], for
    narrow_dgrad_bn_reduce(mask=)."""
    rows = u8.shape[0]
    k0, c0 = w0.shape
    cout = w.shape[1]
    z = torch.empty((rows, cout), dtype=torch.float32, device=u8.device)
    stats = _zeros_f64(2 * cout, u8.device) if want_stats else None
    scale = shift = raw = None
    if in_bn.done:
        scale, shift = in_bn.scale, in_bn.shift
    else:
        raw = in_bn.raw()
    if want_mask:
        if not narrow_mask_supported(rows, c0):
            raise L.InvalidArgumentError("narrow_linear: the mask needs rows %% 128 == 0 and c0 %% 64 == 0")
        mask = torch.empty((rows, c0 // 16), dtype=torch.int16, device=u8.device)
        with L.device_guard(u8.device), _Timed("linear_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "fwd+bn narrow" + (" half" if half else ""))):
            L.check(L.lib().votenet_narrow_linear_masked(rows, k0, c0, cout, L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(scale), L.ptr(shift),
                                                         ctypes.byref(raw) if raw is not None else None, 1 if in_relu else 0, L.ptr(w),
                                                         L.ptr(bias), L.ptr(z), L.ptr(stats), L.ptr(half.wh) if half is not None else None,
                                                         L.ptr(mask), L.stream_ptr()))
        return z, stats, mask
    if half is not None:
        with L.device_guard(u8.device), _Timed("linear_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "fwd+bn narrow half")):
            L.check(L.lib().votenet_narrow_linear_half(rows, k0, c0, cout, L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(scale), L.ptr(shift),
                                                       ctypes.byref(raw) if raw is not None else None, 1 if in_relu else 0, L.ptr(w),
                                                       L.ptr(bias), L.ptr(z), L.ptr(stats), L.ptr(half.wh), L.stream_ptr()))
        return z, stats
    with L.device_guard(u8.device), _Timed("linear_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "fwd+bn narrow")):
        L.check(L.lib().votenet_narrow_linear(rows, k0, c0, cout, L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(scale), L.ptr(shift),
                                              ctypes.byref(raw) if raw is not None else None, 1 if in_relu else 0, L.ptr(w),
                                              L.ptr(bias), L.ptr(z), L.ptr(stats), L.stream_ptr()))
    return z, stats


def narrow_wgrad_bn(u8, w0, b0, in_scale, in_shift, in_relu, z, coef, relu, da, dw, half=None):
    """dw (c0, cout) += relu(bn0(u8 w0 + b0))^T dz1, dz1 = BatchNorm-backward(da, z, coef) formed in the loader."""
    rows = u8.shape[0]
    k0, c0 = w0.shape
    cout = z.shape[1]
    if half is not None:
        with L.device_guard(u8.device), _Timed("wgrad_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "wgrad_bn narrow half")):
            L.check(L.lib().votenet_narrow_wgrad_bn_half(rows, k0, c0, cout, L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(in_scale), L.ptr(in_shift),
                                                         1 if in_relu else 0, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0,
                                                         L.ptr(half.wh), L.ptr(dw), L.stream_ptr()))
        return
    scr = _wgrad_scratch(None, rows, c0, cout, u8.device)
    with L.device_guard(u8.device), _Timed("wgrad_dense", 2.0 * rows * c0 * cout, (rows, c0, cout, "wgrad_bn narrow")):
        L.check(L.lib().votenet_narrow_wgrad_bn(rows, k0, c0, cout, L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(in_scale), L.ptr(in_shift),
                                                1 if in_relu else 0, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(dw),
                                                L.ptr(scr), L.stream_ptr()))


def narrow_dgrad_bn_reduce(z, coef, relu, wT, da, u8, w0, b0, below, eps=BN_EPS, tail=None, half=None, mask=None):
    """The input-gradient GEMM of the second layer with nothing stored: -> sums (2*c0 f64: BatchNorm-backward sums of the first
    layer), ug (8, c0) f64 = sum_r u8[r,:]^T da0'[r,:].  below = (scale, shift, mean, var, relu) of the first layer.
    mask (narrow_linear(want_mask=True); needs tail, the piece layout and mlp.COEF_TAIL): the first layer's ReLU mask from the forward
    pass instead of its rebuild in the epilogue (votenet_narrow_dgrad_bn_reduce_masked)."""
    rows, c = z.shape
    k0, c0 = w0.shape
    bsc, bsh, bme, bva, brelu = below
    out = _zeros_f64(10 * c0, z.device)
    sums, ug = out[:2 * c0], out[2 * c0:]
    t, coef0 = _coef_tail(tail, c0, z.device)
    if mask is not None and t is not None and half is not None:
        with L.device_guard(z.device), _Timed("linear_dense", 2.0 * rows * c * c0, (rows, c, c0, "dgrad_bn_reduce narrow half")):
            L.check(L.lib().votenet_narrow_dgrad_bn_reduce_masked(rows, c, c0, k0, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(wT),
                                                                  L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(bsc), L.ptr(bsh), L.ptr(bme), L.ptr(bva),
                                                                  eps, 1 if brelu else 0, L.ptr(sums), L.ptr(ug), ctypes.byref(t), L.ptr(half.wh),
                                                                  L.ptr(mask), L.stream_ptr()))
        return coef0, ug.view(8, c0)
    with L.device_guard(z.device), _Timed("linear_dense", 2.0 * rows * c * c0, (rows, c, c0, "dgrad_bn_reduce narrow" + (" half" if half else ""))):
        args = (rows, c, c0, k0, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(wT), L.ptr(u8), L.ptr(w0), L.ptr(b0), L.ptr(bsc),
                L.ptr(bsh), L.ptr(bme), L.ptr(bva), eps, 1 if brelu else 0, L.ptr(sums), L.ptr(ug), ctypes.byref(t) if t is not None else None)
        if half is not None:
            L.check(L.lib().votenet_narrow_dgrad_bn_reduce_half(*args, L.ptr(half.wh), L.stream_ptr()))
        else:
            L.check(L.lib().votenet_narrow_dgrad_bn_reduce(*args, L.stream_ptr()))
    if tail is None:
        return sums, ug.view(8, c0)
    return (coef0 if coef0 is not None else _coef_after(tail, (bsc, bsh, bme, bva), sums, eps)), ug.view(8, c0)  # tail: (coef of the first layer, ug)


def narrow_wgrad_first(mom, ug, coef, w0, b0, dw0):
    """dw0 (k0, c0) += the first layer's weight gradient from the sums alone (votenet_narrow_wgrad_first)."""
    k0, c0 = w0.shape
    with L.device_guard(w0.device):
        L.check(L.lib().votenet_narrow_wgrad_first(k0, c0, L.ptr(mom), L.ptr(ug), L.ptr(coef), L.ptr(w0), L.ptr(b0), L.ptr(dw0),
                                                   L.stream_ptr()))


def linear_pool_supported(rows, cin, cout, k):
    """Shapes votenet_mlp_linear_pool serves (see include/votenet_hip.h)."""
    return k == 64 and rows > 0 and rows % 128 == 0 and cin % 32 == 0 and cin <= 512 and cout % 128 == 0


def linear_dense_pool(x, w, k, bias=None, in_scale=None, in_shift=None, in_relu=True, keep_z=True, in_bn=None, half=None, gamma=None):
    """linear_dense whose epilogue also emits the raw max / min of every group of k rows: -> z or None, stats, pool where
    pool = (zmax, zmin, amax, amin), each (rows/k, cout); bn_pool_finalize(pool, scale, shift) completes the max-pool.
    half (piece layout): pool = (zbest, abest), each (pieces, cout) -- one candidate per 16-row piece, the max of z where gamma (the
    layer's own BatchNorm weight: its sign is the sign of the scale the pool applies) is >= 0, else the min."""
    rows, cin = x.shape
    cout = w.shape[1]
    z = torch.empty((rows, cout), dtype=torch.float32, device=x.device) if keep_z else None
    stats = _zeros_f64(2 * cout, x.device)
    d = _desc_dense(x, in_scale, in_shift, in_relu, in_bn)
    if half is not None:
        if gamma is None:
            raise L.InvalidArgumentError("linear_dense_pool(half=...): the layer's gamma decides which extreme a piece keeps")
        zbest = torch.empty((half.nh, cout), dtype=torch.float32, device=x.device)
        abest = torch.empty((half.nh, cout), dtype=torch.int32, device=x.device)
        with L.device_guard(x.device), _Timed("linear_dense", 2.0 * rows * cin * cout, (rows, cin, cout, "fwd+pool half"), limit=half):
            L.check(L.lib().votenet_mlp_linear_pool_half(ctypes.byref(d), rows, cin, cout, L.ptr(w), L.ptr(bias), L.ptr(z), L.ptr(stats),
                                                         L.ptr(half.wh), L.ptr(gamma), L.ptr(zbest), L.ptr(abest), L.ptr(half.nh_limit), L.stream_ptr()))
        return z, stats, (zbest, abest)
    g = rows // k
    vals = torch.empty((2, g, cout), dtype=torch.float32, device=x.device)
    args = torch.empty((2, g, cout), dtype=torch.int32, device=x.device)
    with L.device_guard(x.device), _Timed("linear_dense", 2.0 * rows * cin * cout, (rows, cin, cout, "fwd+pool")):
        L.check(L.lib().votenet_mlp_linear_pool(ctypes.byref(d), rows, cin, cout, L.ptr(w), L.ptr(bias), L.ptr(z), L.ptr(stats), k,
                                                L.ptr(vals[0]), L.ptr(vals[1]), L.ptr(args[0]), L.ptr(args[1]), L.stream_ptr()))
    return z, stats, (vals[0], vals[1], args[0], args[1])


def bn_pool_finalize(pool, scale, shift, relu=True, want_argmax=False, bn=None, want_zsel=False, half=None):
    """bn: a PendingBN instead of scale / shift (the kernel finalizes it).  -> out, argmax [, zsel = raw z at the arg-max]."""
    zmax = pool[0]
    raw = None
    if bn is not None:
        if bn.done:
            scale, shift = bn.scale, bn.shift
        else:
            raw = bn.raw()
    g, c = zmax.shape
    if half is not None:
        g = half.G  # pool holds one entry per piece; the result one per centre
    out = torch.empty((g, c), dtype=torch.float32, device=zmax.device)
    arg = torch.empty((g, c), dtype=torch.int32, device=zmax.device) if want_argmax else None
    zsel = torch.empty((g, c), dtype=torch.float32, device=zmax.device) if want_zsel else None
    if half is not None:
        with L.device_guard(zmax.device):
            L.check(L.lib().votenet_bn_pool_finalize_half(g, c, L.ptr(pool[0]), L.ptr(pool[1]), L.ptr(half.pos), L.ptr(scale), L.ptr(shift),
                                                          ctypes.byref(raw) if raw is not None else None, 1 if relu else 0, L.ptr(out),
                                                          L.ptr(arg), L.ptr(zsel), L.stream_ptr()))
        return (out, arg, zsel) if want_zsel else (out, arg)
    zmax, zmin, amax, amin = pool
    with L.device_guard(zmax.device):
        L.check(L.lib().votenet_bn_pool_finalize(g, c, L.ptr(zmax), L.ptr(zmin), L.ptr(amax), L.ptr(amin), L.ptr(scale), L.ptr(shift),
                                                 ctypes.byref(raw) if raw is not None else None, 1 if relu else 0, L.ptr(out),
                                                 L.ptr(arg), L.ptr(zsel), L.stream_ptr()))
    return (out, arg, zsel) if want_zsel else (out, arg)


# ---- backward of a pooled last layer in Gram form (pool_bwd.hip; see include/votenet_hip.h)
def pool_backward_supported(cin, cout, k):
    return bool(L.lib().votenet_pool_backward_supported(cin, cout, k))


def bn_backward_reduce_pool(gout, zsel, scale, shift, mean, var, relu, eps=BN_EPS, tail=None):
    """sums (2*c f64) of the pooled gradient at the arg-max entries; with tail = (rows, gamma, dgamma, dbeta) the coefficient
    vector of votenet_bn_backward_coef instead (computed in the kernel's tail)."""
    g, c = gout.shape
    sums = _zeros_f64(2 * c, gout.device)
    t, coef = _coef_tail(tail, c, gout.device)
    with L.device_guard(gout.device):
        L.check(L.lib().votenet_bn_backward_reduce_pool(g, c, L.ptr(gout), L.ptr(zsel), L.ptr(scale), L.ptr(shift), L.ptr(mean),
                                                        L.ptr(var), float(eps), 1 if relu else 0, L.ptr(sums),
                                                        ctypes.byref(t) if t is not None else None, L.stream_ptr()))
    if tail is None:
        return sums
    return coef if coef is not None else _coef_after(tail, (scale, shift, mean, var), sums, eps)


def pool_dgrad_prepare(w, bias, coef, rows=None):
    """-> (cin + 1, cin): [W diag(C) W^T ; (B + C.b) W^T], the weights / bias of the dense part of pool_dgrad.  With SPLIT_ADHOC
    (and a shape the split-operand GEMM serves; rows = the rows of that GEMM) the result carries the matrix's bf16 x 3 image (._img)."""
    cin, cout = w.shape
    mm = torch.empty((cin + 1, cin), dtype=torch.float32, device=w.device)
    want_img = SPLIT_ADHOC and split_eligible(cin, cin) and (rows is None or rows >= SPLIT_ADHOC_ROWS)
    with L.device_guard(w.device):
        if ADHOC_H2 and FORWARD_H2 and split_eligible(cin, cin):
            img = torch.empty(cin * cin * 4, dtype=torch.uint8, device=w.device)
            sv = torch.empty((2, cin), dtype=torch.float32, device=w.device)
            L.check(L.lib().votenet_pool_dgrad_prepare_h2(cin, cout, L.ptr(w), L.ptr(bias), L.ptr(coef), L.ptr(mm), L.ptr(mm[cin]),
                                                          L.ptr(img), L.ptr(sv[0]), L.ptr(sv[1]), L.stream_ptr()))
            mm._img, mm._h2 = img, sv
        elif want_img:
            img = torch.empty(cin * cin * 6, dtype=torch.uint8, device=w.device)
            L.check(L.lib().votenet_pool_dgrad_prepare_split(cin, cout, L.ptr(w), L.ptr(bias), L.ptr(coef), L.ptr(mm), L.ptr(mm[cin]),
                                                             L.ptr(img), L.stream_ptr()))
            mm._img = img
        else:
            L.check(L.lib().votenet_pool_dgrad_prepare(cin, cout, L.ptr(w), L.ptr(bias), L.ptr(coef), L.ptr(mm), L.ptr(mm[cin]), L.stream_ptr()))
    return mm


def pool_dgrad(xz, in_scale, in_shift, in_relu, w, bias, wT, coef, relu, gout, argmax, zsel, k, below=None, eps=BN_EPS, mm=None,
               below_tail=None, half=None):
    """da (rows, cin) of the pooled layer: x (W diag(C) W^T) + (B + C.b) W^T as ONE forward-type GEMM on the layer's input,
    then the cout scattered rows per group.  below = (scale, shift, mean, var, relu) of the layer that produced xz: the
    scatter pass then also reduces that layer's BatchNorm backward -> returns (da, sums).  mm: pool_dgrad_prepare's result
    when it was launched ahead."""
    rows, cin = xz.shape
    cout = w.shape[1]
    if mm is None:
        mm = pool_dgrad_prepare(w, bias, coef, rows)
    img = getattr(mm, "_img", None)
    h2 = getattr(mm, "_h2", None)

    def register():  # the matrix exists only for this launch: its image is registered around the GEMM's launch only
        if h2 is not None:
            L.check(L.lib().votenet_register_split_weights_scaled(L.ptr(mm), cin, cin, L.ptr(img), L.ptr(h2[0]), L.ptr(h2[1])))
        else:
            L.check(L.lib().votenet_register_split_weights(L.ptr(mm), cin, cin, L.ptr(img)))
    if half is not None and half.nh_limit is not None:
        # the count is the device's: the dense GEMM stops at it too (rows = the upper bound)
        da = torch.empty((rows, cin), dtype=torch.float32, device=xz.device)
        if h2 is not None:
            register()
        try:
            with L.device_guard(xz.device), _Timed("linear_dense", 2.0 * rows * cin * cin, (rows, cin, cin, "gram-form dense dgrad half"), limit=half):
                L.check(L.lib().votenet_mlp_linear_half(L.ptr(xz), L.ptr(in_scale), L.ptr(in_shift), 1 if in_relu else 0, rows, cin, cin, L.ptr(mm),
                                                        L.ptr(mm[cin]), L.ptr(da), L.ptr(half.nh_limit), L.stream_ptr()))
        finally:
            if h2 is not None:
                L.lib().votenet_register_split_weights(L.ptr(mm), cin, cin, None)
    elif img is not None:
        register()
        try:
            da, _ = linear_dense(xz, mm[:cin], mm[cin], in_scale, in_shift, in_relu, want_stats=False)
        finally:
            L.lib().votenet_register_split_weights(L.ptr(mm), cin, cin, None)
    else:
        da, _ = linear_dense(xz, mm[:cin], mm[cin], in_scale, in_shift, in_relu, want_stats=False)
    sums = _zeros_f64(2 * cin, xz.device) if below is not None else None
    bsc, bsh, bme, bva, brelu = below if below is not None else (None, None, None, None, False)
    t, coef_b = _coef_tail(below_tail if below is not None else None, cin, xz.device)
    if half is not None:
        with L.device_guard(xz.device):
            L.check(L.lib().votenet_pool_dgrad_scatter_half(half.nh, half.G, cin, cout, L.ptr(gout), L.ptr(argmax), L.ptr(zsel), L.ptr(coef),
                                                            1 if relu else 0, L.ptr(wT), L.ptr(da), L.ptr(half.hc), L.ptr(half.wh),
                                                            L.ptr(xz if below is not None else None), L.ptr(bsc), L.ptr(bsh), L.ptr(bme),
                                                            L.ptr(bva), float(eps), 1 if brelu else 0, L.ptr(sums),
                                                            ctypes.byref(t) if t is not None else None, L.ptr(half.nh_limit), L.stream_ptr()))
    else:
        with L.device_guard(xz.device):
            L.check(L.lib().votenet_pool_dgrad_scatter(rows // k, k, cin, cout, L.ptr(gout), L.ptr(argmax), L.ptr(zsel), L.ptr(coef),
                                                       1 if relu else 0, L.ptr(wT), L.ptr(da), L.ptr(xz if below is not None else None),
                                                       L.ptr(bsc), L.ptr(bsh), L.ptr(bme), L.ptr(bva), float(eps), 1 if brelu else 0,
                                                       L.ptr(sums), ctypes.byref(t) if t is not None else None, L.stream_ptr()))
    if below is None:
        return da
    if below_tail is None:
        return da, sums
    return da, (coef_b if coef_b is not None else _coef_after(below_tail, (bsc, bsh, bme, bva), sums, eps))  # below_tail: (da, coef of the layer below)


def gram(xz, scale_shift, relu, half=None):
    """(c, c) a^T a of the activation a = act(xz * scale + shift); scale_shift: contiguous (2, c).  half: a^T diag(w) a over compact rows."""
    rows, c = xz.shape
    g = _zeros_f32((c + 1, c), xz.device)  # [gram ; column sums (filled by pool_wgrad)]
    if half is not None:
        with L.device_guard(xz.device), _Timed("wgrad_dense", 2.0 * rows * c * c, (rows, c, c, "gram half"), limit=half):
            L.check(L.lib().votenet_mlp_gram_half(rows, c, L.ptr(xz), L.ptr(scale_shift), 1 if relu else 0, L.ptr(half.wh), L.ptr(g),
                                                  L.ptr(half.nh_limit), L.stream_ptr()))
        return g
    scr = _wgrad_scratch(None, rows, c, c, xz.device)
    with L.device_guard(xz.device), _Timed("wgrad_dense", 2.0 * rows * c * c, (rows, c, c, "gram")):
        L.check(L.lib().votenet_mlp_gram(rows, c, L.ptr(xz), L.ptr(scale_shift), 1 if relu else 0, L.ptr(g), L.ptr(scr), L.stream_ptr()))
    return g


POOL_WGRAD_CENTRES = True  # piece layout: the arg-max gather of a pooled layer's weight gradient walks centres (all kept pieces of a ball staged together), not pieces


def pool_wgrad(xz, in_scale, in_shift, in_relu, gram_buf, w, bias, coef, relu, gout, argmax, zsel, k, dw, half=None):
    """dw += x^T dz of the pooled layer from the Gram matrix (gram()), the gathered arg-max rows and the column sums."""
    rows, cin = xz.shape
    cout = w.shape[1]
    # (the centre-walking kernel addresses xz with 32-bit byte offsets: above 2 GiB of compact rows -- B >= 64 or so -- the piece form runs)
    if half is not None and POOL_WGRAD_CENTRES and half.nh * 16 * cin * 4 < 2 ** 31:
        with L.device_guard(xz.device):
            L.check(L.lib().votenet_pool_wgrad_sparse_half_centres(half.nh, half.G, cin, cout, L.ptr(xz), L.ptr(in_scale), L.ptr(in_shift),
                                                                   1 if in_relu else 0, L.ptr(gout), L.ptr(argmax), L.ptr(zsel), L.ptr(coef),
                                                                   1 if relu else 0, L.ptr(dw), L.ptr(gram_buf[cin]), L.ptr(half.pos), L.ptr(half.wh),
                                                                   L.ptr(half.nh_limit), L.stream_ptr()))
            L.check(L.lib().votenet_pool_wgrad_finish(cin, cout, L.ptr(gram_buf), L.ptr(gram_buf[cin]), L.ptr(w), L.ptr(bias), L.ptr(coef),
                                                      L.ptr(dw), L.stream_ptr()))
        return
    if half is not None:
        with L.device_guard(xz.device):
            L.check(L.lib().votenet_pool_wgrad_sparse_half(half.nh, half.G, cin, cout, L.ptr(xz), L.ptr(in_scale), L.ptr(in_shift),
                                                           1 if in_relu else 0, L.ptr(gout), L.ptr(argmax), L.ptr(zsel), L.ptr(coef),
                                                           1 if relu else 0, L.ptr(dw), L.ptr(gram_buf[cin]), L.ptr(half.hc), L.ptr(half.wh),
                                                           L.ptr(half.nh_limit), L.stream_ptr()))
            L.check(L.lib().votenet_pool_wgrad_finish(cin, cout, L.ptr(gram_buf), L.ptr(gram_buf[cin]), L.ptr(w), L.ptr(bias), L.ptr(coef),
                                                      L.ptr(dw), L.stream_ptr()))
        return
    scr = None
    if DETERMINISTIC:
        scr = torch.empty(L.lib().votenet_pool_wgrad_scratch_floats(rows // k, cin, cout), dtype=torch.float32, device=xz.device)
    with L.device_guard(xz.device):
        L.check(L.lib().votenet_pool_wgrad_sparse(rows // k, k, cin, cout, L.ptr(xz), L.ptr(in_scale), L.ptr(in_shift), 1 if in_relu else 0,
                                                  L.ptr(gout), L.ptr(argmax), L.ptr(zsel), L.ptr(coef), 1 if relu else 0, L.ptr(dw),
                                                  L.ptr(gram_buf[cin]), L.ptr(scr), L.stream_ptr()))
        L.check(L.lib().votenet_pool_wgrad_finish(cin, cout, L.ptr(gram_buf), L.ptr(gram_buf[cin]), L.ptr(w), L.ptr(bias), L.ptr(coef),
                                                  L.ptr(dw), L.stream_ptr()))


def group_linear(xyz, new_xyz, idx, P, w_xyz, bias=None, want_stats=True):
    """z (b*m*k, cout) = P[b, idx] + (xyz[idx] - new_xyz) @ w_xyz + bias with P (b, n, cout) = feat @ W[3:] computed per POINT
    (first SA layer, linear map before the grouping).  -> z, stats."""
    b, m, k = idx.shape
    n, cout = xyz.shape[1], w_xyz.shape[1]
    rows = b * m * k
    z = torch.empty((rows, cout), dtype=torch.float32, device=xyz.device)
    stats = _zeros_f64(2 * cout, xyz.device) if want_stats else None
    with L.device_guard(xyz.device):
        L.check(L.lib().votenet_group_linear(b, n, m, k, cout, L.ptr(xyz), L.ptr(new_xyz), L.ptr(idx), L.ptr(P), L.ptr(w_xyz),
                                             L.ptr(bias), L.ptr(z), L.ptr(stats), L.stream_ptr()))
    return z, stats


def group_linear_backward_supported(cout, nsample):
    return cout in (32, 64, 128, 256) and nsample <= 128


def group_linear_backward(xyz, new_xyz, idx, pts_cnt, z, da, coef, relu, dw_xyz, want_dz=False):
    """One pass over (z, da) of a BatchNorm'ed first SA layer: -> S (b, n, cout) = scatter-add of dz by idx, dz or None;
    accumulates the xyz rows of the weight gradient into dw_xyz (3, cout) (a view of the gradient bucket)."""
    b, m, k = idx.shape
    n, cout = xyz.shape[1], z.shape[1]
    dz = torch.empty_like(z) if want_dz else None
    if DETERMINISTIC:  # gather-sum over the grouping's inverse index: no atomics, S written once (no zero fill)
        inv = _inverse_of(idx, n)
        S = torch.empty((b, n, cout), dtype=torch.float32, device=z.device)
        scr = torch.empty(L.lib().votenet_group_linear_backward_scratch_floats(b, n, cout), dtype=torch.float32, device=z.device)
        with L.device_guard(z.device):
            L.check(L.lib().votenet_group_linear_backward_csr(b, n, m, k, cout, L.ptr(xyz), L.ptr(new_xyz), L.ptr(inv[0]), L.ptr(inv[1]),
                                                              L.ptr(inv[2]) if len(inv) > 2 else None, L.ptr(z), L.ptr(da), L.ptr(coef), 1 if relu else 0, L.ptr(S), L.ptr(dw_xyz),
                                                              L.ptr(dz), L.ptr(scr), L.stream_ptr()))
        return S, dz
    S = _zeros_f32((b, n, cout), z.device)
    with L.device_guard(z.device):
        L.check(L.lib().votenet_group_linear_backward(b, n, m, k, cout, L.ptr(xyz), L.ptr(new_xyz), L.ptr(idx), L.ptr(pts_cnt),
                                                      L.ptr(z), L.ptr(da), L.ptr(coef), 1 if relu else 0, L.ptr(S), L.ptr(dw_xyz),
                                                      L.ptr(dz), L.stream_ptr()))
    return S, dz


def bn_finalize(rows, stats, gamma, beta, eps=BN_EPS):
    """-> scale, shift, mean, var (each (c,) f32): scale=gamma*rsqrt(var+eps), shift=beta-mean*scale."""
    c = gamma.shape[0]
    out = torch.empty((4, c), dtype=torch.float32, device=gamma.device)
    with L.device_guard(gamma.device):
        L.check(L.lib().votenet_bn_finalize(rows, c, L.ptr(stats), L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(out[0]),
                                            L.ptr(out[1]), L.ptr(out[2]), L.ptr(out[3]), L.stream_ptr()))
    return out[0], out[1], out[2], out[3]


def bn_relu_max(z, k, scale, shift, relu=True, want_argmax=False):
    """(groups*k, c) raw z -> (groups, c) max over each group's k rows of act(z*scale+shift)."""
    rows, c = z.shape
    groups = rows // k
    out = torch.empty((groups, c), dtype=torch.float32, device=z.device)
    arg = torch.empty((groups, c), dtype=torch.int32, device=z.device) if want_argmax else None
    with L.device_guard(z.device):
        L.check(L.lib().votenet_bn_relu_max(groups, k, c, L.ptr(z), L.ptr(scale), L.ptr(shift), 1 if relu else 0, L.ptr(out),
                                            L.ptr(arg), L.stream_ptr()))
    return out, arg


def bn_relu(z, scale, shift, relu=True, bn=None):
    """bn: a PendingBN instead of scale / shift (the kernel finalizes it)."""
    rows, c = z.shape
    y = torch.empty_like(z)
    raw = None
    if bn is not None:
        if bn.done:
            scale, shift = bn.scale, bn.shift
        else:
            raw = bn.raw()
    with L.device_guard(z.device):
        L.check(L.lib().votenet_bn_relu(rows, c, L.ptr(z), L.ptr(scale), L.ptr(shift), ctypes.byref(raw) if raw is not None else None,
                                        1 if relu else 0, L.ptr(y), L.stream_ptr()))
    return y


# --------------------------------------------------------------------------- backward
def bn_backward(z, scale, shift, mean, var, gamma, relu, da, dgamma, dbeta, argmax=None, k=0, eps=BN_EPS):
    """Training-mode BatchNorm (+ReLU) backward.  da: dense (rows,c), or pooled gout (rows/k,c) with argmax.
    Returns dz (rows,c); accumulates dgamma / dbeta (views into the gradient bucket)."""
    sums = bn_backward_reduce(z, scale, shift, mean, var, relu, da, argmax, k, eps)
    coef = bn_backward_coef(z.shape[0], scale, shift, mean, var, gamma, sums, dgamma, dbeta, eps)
    return bn_backward_apply(z, coef, relu, da, argmax, k)


def bn_backward_apply(z, coef, relu, da, argmax=None, k=0):
    """dz (rows,c) = A*g' + B + C*z written out (the unfused path)."""
    rows, c = z.shape
    dz = torch.empty_like(z)
    with L.device_guard(z.device):
        L.check(L.lib().votenet_bn_backward_apply(rows, c, k, L.ptr(da), L.ptr(argmax), L.ptr(z), L.ptr(coef), 1 if relu else 0,
                                                  L.ptr(dz), L.stream_ptr()))
    return dz


def bn_backward_reduce(z, scale, shift, mean, var, relu, da, argmax=None, k=0, eps=BN_EPS, tail=None):
    """sums (2*c f64) = [sum g', sum g'*zhat] of a BatchNorm'ed layer (g' = ReLU / arg-max masked gradient); with
    tail = (rows, gamma, dgamma, dbeta) the coefficient vector of votenet_bn_backward_coef instead (kernel tail)."""
    rows, c = z.shape
    sums = _zeros_f64(2 * c, z.device)
    t, coef = _coef_tail(tail, c, z.device)
    with L.device_guard(z.device):
        L.check(L.lib().votenet_bn_backward_reduce(rows, c, k, L.ptr(da), L.ptr(argmax), L.ptr(z), L.ptr(scale), L.ptr(shift),
                                                   L.ptr(mean), L.ptr(var), float(eps), 1 if relu else 0, L.ptr(sums),
                                                   ctypes.byref(t) if t is not None else None, L.stream_ptr()))
    if tail is None:
        return sums
    return coef if coef is not None else _coef_after(tail, (scale, shift, mean, var), sums, eps)


def bn_backward_coef(rows, scale, shift, mean, var, gamma, sums, dgamma, dbeta, eps=BN_EPS):
    """coef (5*c) = [A|B|C|scale|shift] with dz = A*g' + B + C*z; accumulates dgamma / dbeta."""
    c = gamma.shape[0]
    coef = torch.empty(5 * c, dtype=torch.float32, device=gamma.device)
    with L.device_guard(gamma.device):
        L.check(L.lib().votenet_bn_backward_coef(rows, c, L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(var), float(eps),
                                                 L.ptr(gamma), L.ptr(sums), L.ptr(coef), L.ptr(dgamma), L.ptr(dbeta),
                                                 L.stream_ptr()))
    return coef


def dgrad_bn_supported(rows, c, cout):
    """Shapes the fused BatchNorm-backward dgrad kernel serves (see include/votenet_hip.h)."""
    return rows > 0 and rows % 128 == 0 and c % 32 == 0 and c <= 512 and cout % 64 == 0


def wgrad_dense_bn(x, z, coef, relu, dw, da=None, gout=None, argmax=None, k=0, in_scale=None, in_shift=None, in_relu=True):
    """dw += act(x)^T dz with dz = BatchNorm-backward(da | pooled gout, z, coef) formed in the loader."""
    rows, cin = x.shape
    cout = z.shape[1]
    d = _desc_dense(x, in_scale, in_shift, in_relu)
    scr = _wgrad_scratch(d, rows, cin, cout, x.device)
    with L.device_guard(x.device), _Timed("wgrad_dense", 2.0 * rows * cin * cout, (rows, cin, cout, "wgrad_bn")):
        L.check(L.lib().votenet_mlp_wgrad_bn(ctypes.byref(d), rows, cin, cout, L.ptr(da), L.ptr(gout), L.ptr(argmax), k, L.ptr(z),
                                             L.ptr(coef), 1 if relu else 0, L.ptr(dw), L.ptr(scr), L.stream_ptr()))


def dgrad_bn(z, coef, relu, wT, da=None, gout=None, argmax=None, k=0, below=None, eps=BN_EPS, below_tail=None):
    """da_prev = dz @ wT with dz = BatchNorm-backward(da | pooled gout, z, coef) formed in the loader.
    below = (z_prev, scale, shift, mean, var, relu) of the layer whose activation da_prev is the gradient of: the store
    epilogue then also reduces that layer's BatchNorm backward -> returns (da_prev, sums) (dense da only)."""
    rows, c = z.shape
    cout = wT.shape[1]
    out = torch.empty((rows, cout), dtype=torch.float32, device=z.device)
    if below is not None:
        zp, bsc, bsh, bme, bva, brelu = below
        sums = _zeros_f64(2 * cout, z.device)
        t, coef_b = _coef_tail(below_tail, cout, z.device)
        with L.device_guard(z.device), _SplitK(rows, c, cout, z.device), _Timed("linear_dense", 2.0 * rows * c * cout, (rows, c, cout, "dgrad_bn_reduce")):
            L.check(L.lib().votenet_mlp_dgrad_bn_reduce(rows, c, cout, L.ptr(da), L.ptr(z), L.ptr(coef), 1 if relu else 0, L.ptr(wT),
                                                        L.ptr(out), L.ptr(zp), L.ptr(bsc), L.ptr(bsh), L.ptr(bme), L.ptr(bva), eps,
                                                        1 if brelu else 0, L.ptr(sums), ctypes.byref(t) if t is not None else None,
                                                        L.stream_ptr()))
        if below_tail is None:
            return out, sums
        return out, (coef_b if coef_b is not None else _coef_after(below_tail, (bsc, bsh, bme, bva), sums, eps))  # (da_prev, coef of the layer below)
    with L.device_guard(z.device), _SplitK(rows if da is not None else 0, c, cout, z.device), _Timed("linear_dense", 2.0 * rows * c * cout, (rows, c, cout, "dgrad_bn")):
        L.check(L.lib().votenet_mlp_dgrad_bn(rows, c, cout, L.ptr(da), L.ptr(gout), L.ptr(argmax), k, L.ptr(z), L.ptr(coef),
                                             1 if relu else 0, L.ptr(wT), L.ptr(out), L.stream_ptr()))
    return out


def bias_grad(dz, dbias):
    """dbias += column sums of dz (rows, c); dz may be a column slice [:, :c] of a wider row-major tensor (read in place)."""
    rows, c = dz.shape
    pitch = dz.stride(0) if rows > 1 else c
    if dz.stride(1) != 1:
        dz, pitch = dz.contiguous(), c
    with L.device_guard(dz.device):
        L.check(L.lib().votenet_bias_grad_strided(rows, c, L.ptr(dz), pitch, L.ptr(_zeros_f64(c, dz.device)), L.ptr(dbias), L.stream_ptr()))


def wgrad_dense(x, dz, dw, in_scale=None, in_shift=None, in_relu=True):
    """dw += act(x)^T dz with the forward pass's folded BN+ReLU on x."""
    rows, cin = x.shape
    cout = dz.shape[1]
    d = _desc_dense(x, in_scale, in_shift, in_relu)
    scr = _wgrad_scratch(d, rows, cin, cout, x.device)
    with L.device_guard(x.device), _Timed("wgrad_dense", 2.0 * rows * cin * cout, (rows, cin, cout, "wgrad")):
        L.check(L.lib().votenet_mlp_wgrad(ctypes.byref(d), rows, cin, cout, L.ptr(dz), L.ptr(dw), L.ptr(scr), L.stream_ptr()))


def wgrad_gather(xyz, new_xyz, feat, idx, dz, dw):
    b, m, k = idx.shape
    c = feat.shape[2] if feat is not None else 0
    d = _desc_gather(xyz, new_xyz, feat, idx)
    scr = _wgrad_scratch(d, b * m * k, 3 + c, dz.shape[1], xyz.device)
    with L.device_guard(xyz.device), _Timed("wgrad_gather", 2.0 * b * m * k * (3 + c) * dz.shape[1], (b * m * k, 3 + c, dz.shape[1], "wgrad_gather")):
        L.check(L.lib().votenet_mlp_wgrad(ctypes.byref(d), b * m * k, 3 + c, dz.shape[1], L.ptr(dz), L.ptr(dw), L.ptr(scr), L.stream_ptr()))


def rows_dot3(dz, w3):
    """(rows, c) x (3, c)^T -> (rows, 3): the xyz columns of an input gradient, dz W[0:3]^T (votenet_rows_dot3)."""
    rows, c = dz.shape
    out = torch.empty((rows, 3), dtype=torch.float32, device=dz.device)
    with L.device_guard(dz.device):
        L.check(L.lib().votenet_rows_dot3(rows, c, L.ptr(dz), L.ptr(w3), L.ptr(out), L.stream_ptr()))
    return out


def group_concat_grad(d_rows_feat, d_rows_xyz, idx, pts_cnt, n, c):
    """Per-row input gradients of the first SA layer -> d_feat (b,n,c) or None, d_xyz (b,n,3) or None,
    d_new_xyz (b,m,3) or None (GroupPointGrad + the gradient of the centre subtraction, utils.py:50-57)."""
    b, m, k = idx.shape
    dev = idx.device
    if DETERMINISTIC and (d_rows_feat is None or d_rows_feat.shape[1] <= 256):
        inv = _inverse_of(idx, n)
        d_feat = csr_gather_sum(d_rows_feat, inv, b * n).view(b, n, c) if d_rows_feat is not None else None
        d_xyz = d_new = None
        if d_rows_xyz is not None:
            d_xyz = csr_gather_sum(d_rows_xyz, inv, b * n).view(b, n, 3)
            d_new = -d_rows_xyz.view(b, m, k, 3).sum(2)  # gradient of the centre subtraction (utils.py:55)
        return d_feat, d_xyz, d_new
    d_feat = _zeros_f32((b, n, c), dev) if d_rows_feat is not None else None
    d_xyz = _zeros_f32((b, n, 3), dev) if d_rows_xyz is not None else None
    d_new = _zeros_f32((b, m, 3), dev) if d_rows_xyz is not None else None
    with L.device_guard(dev):
        L.check(L.lib().votenet_group_concat_grad(b, n, c, m, k, L.ptr(d_rows_feat), L.ptr(d_rows_xyz), L.ptr(idx), L.ptr(pts_cnt),
                                                  L.ptr(d_feat), L.ptr(d_xyz), L.ptr(d_new), L.stream_ptr()))
    return d_feat, d_xyz, d_new


def _rows2d(t):
    """(base tensor, pitch) of a 2-D row-major view with unit column stride (a column slice of a wider tensor is fine)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise L.InvalidArgumentError("row_segments: operands are 2-D row-major views with unit column stride")
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


def row_segments(rows, segs):
    """votenet_row_segments (csrc/glue.hip): concat / slice / zero-pad / add of row-major tensors in one launch.
    segs: up to 8 tuples (dst, a, b): 2-D views of `rows` rows and equal width -- dst = a (+ b); a None writes zeros, b may be None.
    Views may be column slices of wider tensors (their row pitch is used)."""
    arr = (L.RowSegment * len(segs))()
    dev = None
    for i, (dst, a, b) in enumerate(segs):
        d, dp = _rows2d(dst)
        dev = d.device
        w = d.shape[1]
        arr[i].dst, arr[i].dst_pitch, arr[i].dst_off, arr[i].width = d.data_ptr(), dp, 0, w
        for name, t in (("a", a), ("b", b)):
            if t is None:
                setattr(arr[i], name, None)
                setattr(arr[i], name + "_pitch", w)
                setattr(arr[i], name + "_off", 0)
                continue
            tt, tp = _rows2d(t)
            if tt.shape[0] != rows or tt.shape[1] != w or d.shape[0] != rows:
                raise L.InvalidArgumentError("row_segments: segment %d: shapes %s / %s do not match (%d rows)" % (i, tuple(d.shape), tuple(tt.shape), rows))
            setattr(arr[i], name, tt.data_ptr())
            setattr(arr[i], name + "_pitch", tp)
            setattr(arr[i], name + "_off", 0)
    with L.device_guard(dev):
        L.check(L.lib().votenet_row_segments(rows, len(segs), arr, L.stream_ptr()))


def copy_segments(pairs):
    """votenet_copy_segments (csrc/glue.hip): dst.copy_(src) for up to 32 (dst, src) pairs of contiguous device tensors of equal byte size
    in ONE launch."""
    arr = (L.CopySegment * len(pairs))()
    dev = None
    for i, (dst, src) in enumerate(pairs):
        nb = dst.numel() * dst.element_size()
        if not (dst.is_contiguous() and src.is_contiguous() and nb == src.numel() * src.element_size() and dst.device == src.device):
            raise L.InvalidArgumentError("copy_segments: pair %d: tensors must be contiguous, on one device and of equal byte size (%s <- %s)"
                                         % (i, tuple(dst.shape), tuple(src.shape)))
        arr[i].dst, arr[i].src, arr[i].bytes = dst.data_ptr(), src.data_ptr(), nb
        dev = dst.device
    if not pairs:
        return
    with L.device_guard(dev):
        L.check(L.lib().votenet_copy_segments(len(pairs), arr, L.stream_ptr()))


SUMSQ_SLICES = 32  # VOTENET_SUMSQ_SLICES (include/votenet_hip.h): partial sums per tensor in votenet_clip_adam's scratch


def clip_adam(seg, sumsq, p, g, m, v, lr, step, grad_scale=1.0, clip=0.5, beta1=0.9, beta2=0.999, eps=1e-8):
    """model.py:240-250: per-tensor clip_by_average_norm(g, 0.5) then Adam(lr) on the flat bucket."""
    if sumsq.numel() < SUMSQ_SLICES * (seg.numel() // 2):
        raise L.InvalidArgumentError("clip_adam: the scratch holds %d floats, %d tensors need %d" % (sumsq.numel(), seg.numel() // 2, SUMSQ_SLICES * (seg.numel() // 2)))
    with L.device_guard(p.device):
        L.check(L.lib().votenet_clip_adam(seg.numel() // 2, L.ptr(seg), L.ptr(sumsq), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v),
                                          float(lr), float(beta1), float(beta2), float(eps), int(step), float(grad_scale),
                                          float(clip), L.stream_ptr()))
