// mlp_fast.hip -- lean MFMA GEMM (fp32 operands; products on fp32 MFMAs or on bf16 x 3 split operands) for the dense layers of the
// grouped-point MLP (gfx950).
//
// Same contract as mlp_linear_kernel (mlp.hip) for the case every dense VoteNet layer is in:
//   DENSE input, cin % 32 == 0, cout % 64 == 0, rows % 128 == 0, 16-byte aligned operands, cin <= 512.
// Written separately so that the hot loop carries no bounds checks, no mode branches and few live
// registers (the generic kernel needs > 220 VGPRs and spills its prefetch registers, which turns the
// "asynchronous" global loads into synchronous ones):
//   * operand addresses are two pointers per thread that advance by constants;
//   * two register sets hold the raw A / W quads of the next two (tile, k-slab) steps: half way through a step's
//     MFMAs the older set goes to the other LDS buffer -- the previous layer's folded BN scale/shift + ReLU are
//     applied at that point -- and is refilled with the slab three steps ahead, so every global load has two steps
//     of matrix work (~2 x 2048 MFMA cycles per wave) to arrive.  With one step of lead the loaded HBM latency was
//     longer than a step and memory time simply added to matrix time (sa2 L1: 68 -> 91 TFLOP/s with two);
//     the loop body has no memory operation under a branch, which keeps the compiler's vmcnt bookkeeping exact,
//     and the barrier waits for LDS only (lgkmcnt), never for vmcnt;
//   * the pipeline runs across row tiles of a persistent workgroup (no drain at tile boundaries);
//   * MFMA operand fragments are double-buffered in registers (ds_reads of sub-step k2+1 before the MFMAs
//     of k2); LDS images are [k][row] / [k][col], conflict-free for both operand reads.
//
// BF3 = true instantiations (the default path once the weight matrix has a registered image, votenet_split_weights): the same
// kernel with the products on bf16 MFMAs.  Every fp32 operand is split exactly into three bf16 pieces (split3), a product is the
// six terms hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (what is dropped
// is <= 2^-23 of the product); the weights arrive pre-split in LDS order, the activations are split where they are staged; one
// 16-deep slab is ONE k-step per piece pair, its barrier sits in the middle of the slab (see the slab body).  Loaders (SRC) and
// epilogues (EPI) are shared with the fp32 form: the 32 x 32 accumulator layout is the same.
#include "mlp_types.h"
#include <mutex>
#include <unordered_map>

namespace votenet {

constexpr int FG_BM = 128;
constexpr int FG_BK = 16;
constexpr int FG_LDA = FG_BM + 2;
#ifndef FWD_WAVES
#define FWD_WAVES 2 // EPI 0 / 2: minimum waves per SIMD
#endif
#ifndef EPI3_WAVES
#define EPI3_WAVES 2 // EPI 3: minimum waves per SIMD the register allocator is held to
#endif
#ifndef BF3_VPM
#define BF3_VPM 6 // BF3: vector instructions scheduled after each MFMA of a slab
#endif
#ifndef BF3_SETS
#define BF3_SETS 2 // BF3: register sets of raw operands in flight (slabs of lead); cin % (16 * BF3_SETS) == 0
#endif
#ifndef H2_SETS
#define H2_SETS 2 // H2: register sets of raw operands in flight (its slabs are shorter: the same lead in slabs is less lead in time); cin % (16 * H2_SETS) == 0
#endif
#ifndef BF3_PRIO
#define BF3_PRIO 0 // BF3: 1 = matrix loop at s_setprio 1, epilogue at 0; 2 = the reverse; 3 = prio 1 only around the MFMAs of a slab
#endif
#ifndef H2_SCALE
#define H2_SCALE 1 // H2: operands travel scaled by powers of two (see kH2A below); 0 in probe builds only
#endif
#ifndef BF3_ABL
#define BF3_ABL 0 // probe builds only (tools/probe/bf3_ablate.sh): 1 no MFMAs, 2 no epilogue, 4 no global loads after the prologue, 8 no staging,
                  // 16 the W image neither loaded nor staged after the prologue (upper bound of what LDS-resident weights could save)
#endif
#ifndef EPI6_ROWS
#define EPI6_ROWS 8 // EPI 6: rows of a 32 x 32 sub-tile column whose P gathers are in flight together (8 or 16)
#endif
#ifndef EPI3_CH
#define EPI3_CH 2 // EPI 3: 32x32 sub-tiles of z_prev loaded at a time
#endif

// Everything the kernel needs.  SRC selects how the A operand (rows x cin) is produced:
//   SRC 0: x, optionally through relu(x*in_scale+in_shift)                      (forward / plain dgrad)
//   SRC 1: dz = A*g + B + C*zsrc with g = da masked by [zsrc*S+H > 0]           (BatchNorm backward folded in,
//   SRC 2: same with g = gout[row/pool_k] where row%pool_k == argmax[row/pool_k]  dense / max-pooled upstream)
//          coef = [A | B | C | S | H], 5*cin floats (votenet_bn_backward_coef)
//   SRC 3: x = z0 of a NARROW first layer (narrow.hip), rebuilt per element from the row's eight floats u8[r] with narrow_z
//          (W0: k0 x cin, b0), then relu(x*in_scale+in_shift) as SRC 0: the loader reads 32 bytes per row and slab from L2
//          instead of 64 bytes of z0 from HBM, and z0 is never stored                                 (cin <= 128)
//   SRC 4: x = z0 of a first SA layer ASSEMBLED here (assemble.hip): P[prow(r), k] + dxyz(r) . Wx[:, k] from the row's geo record
//          (16 bytes: dx, dy, dz, bits(prow)) and a gather of the L2-resident per-point table P; then the folded BatchNorm +
//          ReLU as SRC 0.  The geo of a slab is loaded one refill BEFORE the P rows that need it and ahead of that refill's other
//          loads in program order, so neither the dependency nor vmcnt's in-order retirement exposes it.          (cin <= 512)
// EPI selects the statistics accumulated next to the store of the output z (rows x cout):
//   EPI 0: sum z, sum z^2                      (BatchNorm statistics of a forward layer)
//   EPI 1: no statistics (input-gradient GEMMs)
//   EPI 3: the output is the gradient da of the layer BELOW's activation; next to its store the epilogue reads that layer's
//          z tile (ez) and accumulates its BatchNorm-backward sums s1 = sum da', s2 = sum da' * zhat, da' = da [act > 0]
//          (what votenet_bn_backward_reduce would do in a pass over da and z of its own: here da never comes back from HBM)
//   EPI 2: EPI 0 plus the max-pool of utils.py:132 over groups of 64 rows, BEFORE BatchNorm: the layer's scale/shift
//          need the statistics of the whole launch, but max_k relu(s*z+h) = relu(s*max_k z + h) for s >= 0 and
//          relu(s*min_k z + h) for s < 0 (rounding is monotone), so the epilogue emits the raw max AND min of every
//          group (+ arg rows) and votenet_bn_pool_finalize picks by the sign of the scale.  A wave's 2 x 32 rows
//          are exactly one group (2x2 variant, WM = 2, MT = 2).
//   EPI 8: EPI 2 on the piece layout (half.hip), the pool per 16-row piece compiled in (A.pool32 is set): without the 64-row pool's
//          running max / min / arg registers and its compare-and-keep code beside it
//   EPI 6: EPI 3 for an ASSEMBLED layer below (assemble.hip): z_prev[r,c] = P[prow(r),c] + dxyz(r) . wx[:,c] is rebuilt per element
//          from the tile's geo records (staged in LDS by the loader) and a gather of the per-point table P through a buffer
//          descriptor (lane offset prow * pitch + column); da is stored as in EPI 3
//   EPI 4: EPI 3 for a narrow layer below: z_prev is rebuilt from u8 (staged per tile in LDS by the loader), the epilogue also
//          accumulates UG[d,c] = sum_r u[r,d] da'[r,c] (the data term of that layer's weight gradient) and stores NOTHING
//          (cout <= 128: the layer below's width)
//   EPI 7: EPI 4 without the rebuild of z_prev (round 4).  Per accumulator element EPI 4 spends 8 fmas on z_prev, 4 on the mask, 3 on
//          s2 and 8 on UG -- twice the cycles of the tile's MFMAs, 256 VGPRs and 26 spilled (108 bytes of scratch per lane: the
//          129.5 MB per launch rocprof saw written by a kernel that stores nothing).  Here the ReLU mask of the narrow layer comes from
//          the FORWARD pass (SRC 3 records, per row and slab, the 16 bits [relu(bn0(z0)) > 0]: mask_out; one u64 per row and 64-column
//          block here: mask_in, staged per tile in LDS beside the u rows), and s2 = sum da' zhat_prev is not accumulated at all: z_prev
//          is LINEAR in u, so sum da' z_prev[:, c] = sum_d W0[d, c] UG[d, c] + b0[c] s1[c], which the coefficient tail evaluates once
//          per column from the completed sums (needs tail.ticket).  Per element: a bit test, s1, 8 fmas.
struct FastArgs {
    const float *x, *in_scale, *in_shift;
    BnRaw in_raw; // alternative to in_scale / in_shift: derived here from the producer's raw sums
    int in_relu;
    const float *da, *gout;
    const int *argmax;
    int pool_k;
    const float *zsrc, *coef;
    int src_relu;
    double *stats;
    long rows;
    int cin, cout;
    const float *w, *bias;
    const unsigned *w3;    // BF3: w as three bf16 pieces in the kernel's LDS order (votenet_split_weights), or NULL
    int w3_np;             // pieces of that image: 3 (bf16 x 3) or 2 (fp16 x 2: H2 instantiations, forward families only)
    const float *h2_ascale, *h2_unscale; // H2 with a matrix made on the fly (votenet_pool_dgrad_prepare_h2): power-of-two scale per INPUT channel
                                         // (replaces the 2^4 of the staged activations) and per OUTPUT column (replaces 2^-12); NULL: the constants
    float *z;              // may be NULL with EPI 2 (inference: only the pooled result is wanted)
    float *zmax, *zmin;    // EPI 2: per 64-row group and channel, raw max / min of z ...
    int *amax, *amin;      //        ... and the row offsets (first occurrence) where they are attained
    const float *ez, *e_scale, *e_shift, *e_mean, *e_var; // EPI 3: z (rows x cout) and BatchNorm of the layer below
    float e_eps;
    int e_relu;
    const float *geo, *ptab, *wx; // SRC 4: geo (rows x 4 floats), P (points x cin), Wx (3 x cin)
    const float *u8, *w0, *b0; // SRC 3 / EPI 4: rows x 8 floats, W0 (k0 x c0), b0 (c0, may be NULL); c0 = cin (SRC 3) or cout (EPI 4)
    int k0;
    double *ug;                // EPI 4 / 7: [8][cout] doubles
    unsigned short *mask_out;  // SRC 3 (may be NULL): rows x (cin / 16) words, bit k % 16 of word [row][k / 16] = [relu(bn0(z0[row, k])) > 0]
    const unsigned long long *mask_in; // EPI 7: the same array read as rows x (cout / 64) u64
    CoefTail tail;             // EPI 3 / 4: the layer below's coefficient vector from the completed sums (last workgroup, common.h)
    // piece layout (half.hip): rows come in pieces of kPiece = 16, wh[row / 16] = the weight of the piece's row 0 (a ball's slot 0 also
    // stands for its dropped all-copy pieces; 1 for every other piece).  EPI 0 / 2: the statistics count that row wh times; SRC 5: the
    // affine part B + C z of the rebuilt dz is scaled by it (total gradients); pool32 (EPI 2): raw max / min per piece instead of per 64 rows
    const float *wh;
    int pool32;
    const int *nh_dev;       // piece layout with the count known on the device only (a level whose geometry is made inside the step):
                             // `rows` is the caller's upper bound, the kernel stops at 16 * nh_dev[0] rows; NULL: rows is exact
    int xcd_chunk;           // XCD x takes the x-th contiguous eighth of the row tiles (see the kernel)
    const float *pool_gamma; // pool32: the pooled layer's BatchNorm gamma -- its sign is the sign of the scale the pool will apply, so the
                             // epilogue keeps ONE candidate per piece and channel (the max where gamma >= 0, else the min) in zmax / amax
    // split-K (gridDim.z > 1; the launcher's choice for launches of few row tiles -- the static stretch's GEMMs of 2048-8192 rows, one
    // wave per SIMD on half of the CUs with a serial chain of cin / 16 slabs): workgroup (x, y, s) contracts the s-th part of cin for
    // output tile (x, y), stores its partial tile in sk_ws [tile][s][element][thread] and takes the tile's ticket; whoever takes the
    // LAST ticket adds the parts in the fixed order s = 0, 1, ... (bit-reproducible whichever workgroup that is) and runs the epilogue
    // -- statistics included -- on the complete tile; the others skip it.  Partial tiles travel as device-scope relaxed atomics (they
    // execute at the memory side: the XCDs' L2s are not coherent with each other) and a workgroup waits for its stores' acknowledgement
    // (vmcnt) before it takes the ticket: the same fence-free hand-off as coef_tail (common.h).
    float *sk_ws;
    unsigned *sk_ticket;
};

// WM x WN waves (WM*WN = 4), each MT x NT tiles of 32x32: BM = WM*MT*32 = 128, BN = WN*NT*32.
// amdgpu_waves_per_eu caps the occupancy the register allocator aims for: at 4 waves/SIMD (128 VGPRs) the
// 2x2 variant spills exactly its prefetch registers, which makes the prefetch synchronous.
template <int WM, int WN, int MT, int NT, int SRC, int EPI, bool BF3 = false, bool SK = false /* split-K launch: gridDim.z parts (FastArgs::sk_ws) */,
          bool H2 = false /* BF3 path on TWO fp16 pieces and three MFMAs per product (mlp_types.h: split2); forward families */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((EPI == 3 || EPI == 6) ? EPI3_WAVES : (EPI == 0 || EPI == 2 || EPI == 8) ? FWD_WAVES : 2, 3))) void mlp_linear_fast_kernel(FastArgs A)
{
    static_assert(!H2 || (BF3 && !SK && (SRC == 0 || SRC == 3 || SRC == 4)), "H2: a split-operand forward-type instantiation");
    constexpr int NPC = H2 ? 2 : 3;               // pieces per operand (split images, LDS planes)
    // H2: both operands travel scaled by a power of two (exact) so that their lo pieces stay NORMAL fp16 numbers for everyday magnitudes:
    // the weights by 2^8 in the image (split_weights_body_h2), the staged activations by 2^4 (folded into the BatchNorm scale / shift
    // table: no instruction), the accumulators are scaled back by 2^-12 inside the epilogue's bias add (an fma instead of an add: no
    // instruction).  Full 22 bits for |a| >= 2^-6 and |w| >= 2^-10 (unscaled: 2^-2 each); |a| < 4094, |w| < 255 or the result is inf.
    constexpr float kH2A = (H2 && H2_SCALE) ? 16.0f : 1.0f, kH2Un = (H2 && H2_SCALE) ? 1.0f / 4096.0f : 1.0f;
    constexpr bool POOL = (EPI == 2 || EPI == 8); // pooled forward layer
    constexpr bool P32 = (EPI == 8);              // ... per 16-row piece
    static_assert(WM * WN == 4 && WM * MT * 32 == FG_BM, "tile shape");
    constexpr int BN = WN * NT * 32;
    constexpr int LDB = BN + 4;
    constexpr int NB4 = FG_BK * BN / 4 / 256; // W float4 per thread per slab (1 or 2)
    __shared__ float As[BF3 ? 1 : 2][BF3 ? 1 : FG_BK][FG_LDA];
    __shared__ float Bs[BF3 ? 1 : 2][BF3 ? 1 : FG_BK][LDB];
    // BF3: both operands as three bf16 pieces (split3 below), one image per (piece, k-half): [row or column][8 bf16 = 4 dwords] --
    // a lane's MFMA fragment is ONE 16-byte read and consecutive lanes read consecutive 16 bytes (conflict-free ds_read_b128);
    // 16 dwords between the planes put the two k-halves one staging wave writes on different banks
    constexpr int PLA = FG_BM * 4 + 16, PLB = BN * 4 + 16;
    __shared__ __attribute__((aligned(16))) unsigned As3[BF3 ? 2 : 1][NPC][2][BF3 ? PLA : 4];
    __shared__ __attribute__((aligned(16))) unsigned Bs3[BF3 ? 2 : 1][NPC][2][BF3 ? PLB : 4];
    __shared__ __attribute__((aligned(16))) float Sco[(SRC == 0 ? 2 : 5)][512]; // per-input-channel coefficients
    constexpr bool NEPI = (EPI == 4 || EPI == 7); // the layer below is a NARROW one
    constexpr bool REDUCE_BELOW = (EPI == 3 || NEPI || EPI == 6);
    __shared__ float Eco[REDUCE_BELOW ? 4 : 1][REDUCE_BELOW ? BN : 1]; // EPI 3/4/6: scale, shift, mean, 1/std of this column block
    constexpr bool NARROW = (SRC == 3 || EPI == 4);
    __shared__ __attribute__((aligned(16))) float W0s[NARROW ? 9 : 1][NARROW ? 128 : 1]; // W0 rows 0..7 (zero padded) and b0: SRC 3 by input channel, EPI 4 by column of this block
    __shared__ __attribute__((aligned(16))) float Wxs[SRC == 4 ? 3 : 1][SRC == 4 ? 512 : 1];             // SRC 4: W[0:3] by input channel
    __shared__ __attribute__((aligned(16))) float Us[NEPI ? 2 : 1][NEPI ? FG_BM : 1][8];
    __shared__ __attribute__((aligned(8))) uint2 Ms[EPI == 7 ? 2 : 1][EPI == 7 ? FG_BM : 1]; // EPI 7: the tile's mask rows (the 64 columns of this block)
    __shared__ __attribute__((aligned(16))) float4 Gs[EPI == 6 ? 2 : 1][EPI == 6 ? FG_BM : 1]; // EPI 6: geo rows of the tile, by tile parity // EPI 4: u rows of the tile, double-buffered by tile parity

    const long rows = A.rows;
    const int cin = A.cin, cout = A.cout;
    const float *__restrict__ w = A.w;
    float *__restrict__ z = A.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const int n0 = blockIdx.y * BN;
    const int nsp = SK ? (int)gridDim.z : 1;       // split-K parts
    const int nk = cin / FG_BK / nsp;              // slabs this workgroup contracts ...
    const int ks0 = SK ? (int)blockIdx.z * nk : 0; // ... from this one on
    long ntiles = rows / FG_BM;
    if (A.nh_dev != nullptr) { // (a multiple of 8 pieces = whole tiles: half.hip)
        const long lim = (long)A.nh_dev[0] * kPiece / FG_BM;
        ntiles = lim < ntiles ? lim : ntiles;
    }
    const bool affine = (SRC == 0 || SRC == 3 || SRC == 4) && (A.in_scale != nullptr || A.in_raw.stats != nullptr);
    // which row tiles this workgroup takes: tile0, tile0 + tstride, ...  By default round-robin over the launch.  xcd_chunk (piece
    // layout): workgroups are dealt to the 8 XCDs round-robin (workgroup b -> XCD b % 8), and every kernel that gathers rows of the
    // per-point table P from a SCENE's 1 MB slice had all eight slices in flight on every XCD -- 8.4 MB against its 4 MB L2 (FETCH_SIZE
    // 1.6 x the algorithmic reads on the assembled input gradient).  The rows are in scene order, so XCD x takes the x-th contiguous
    // eighth of the tiles: its L2 then holds the one or two slices that eighth touches.
    long tile0 = blockIdx.x, tstride = gridDim.x, my_tiles = 0;
    if (A.xcd_chunk && (gridDim.x & 7) == 0) {
        const long per = (ntiles + 7) / 8, lo = (long)(blockIdx.x & 7) * per, hi = lo + per < ntiles ? lo + per : ntiles;
        tstride = gridDim.x >> 3;
        tile0 = lo + (blockIdx.x >> 3);
        if (tile0 < hi) my_tiles = (hi - 1 - tile0) / tstride + 1;
    } else if ((long)blockIdx.x < ntiles) {
        my_tiles = (ntiles - 1 - blockIdx.x) / gridDim.x + 1;
    }
    long steps_to_load = my_tiles * nk; // steps whose operands still have to be fetched

    // A staging: thread t -> tile rows (t>>2) and (t>>2)+64, k-quad (t&3); W staging: float4 #t (+256)
    const int a_row = tid >> 2, a_kq = tid & 3;
    const float *abase = (SRC == 0) ? A.x : (SRC == 3) ? A.u8 : (SRC == 4) ? A.ptab : A.zsrc; // the array the row pointers walk
    // SRC 3: the pointers stay on the row's eight floats for all slabs of a tile (re-read per slab from L2: no branch in the loop)
    const int arow_len = (SRC == 3) ? 8 : cin;
    const float *pa0 = abase + ((size_t)tile0 * FG_BM + a_row) * arow_len + (SRC == 3 ? 0 : ks0 * FG_BK + a_kq * 4);
    const float *pa1 = pa0 + (size_t)64 * arow_len;
    const ptrdiff_t da_off = (SRC == 1 || SRC == 5) ? (A.da - A.zsrc) : 0; // SRC 1 / 5: da has the layout of zsrc
    // SRC 5 (piece layout): this thread's rows a_row / a_row + 64 of a tile are row 0 of their piece iff a_row % kPiece == 0; the
    // weights of those two pieces travel with the slab (no load under a branch: every thread loads, most ignore)
    const bool sel31 = (a_row % kPiece) == 0;
    const float *pw = (SRC == 5) ? A.wh + (size_t)tile0 * (FG_BM / kPiece) + (a_row / kPiece) : nullptr;
    const size_t a_tile_jump = (SRC == 3) ? (size_t)tstride * FG_BM * 8 : (size_t)tstride * FG_BM * cin - (size_t)nk * FG_BK; // after the last slab of a tile
    const int a_slab_step = (SRC == 3) ? 0 : FG_BK;
    // EPI 4: thread t stages float4 #(t&1) of tile row t>>1 for the epilogue
    const uint2 *pm = (EPI == 7) ? reinterpret_cast<const uint2 *>(A.mask_in) + ((size_t)tile0 * FG_BM + (tid & 127)) * (cout / 64) + n0 / 64 : nullptr;
    const float *pu = NEPI ? A.u8 + ((size_t)tile0 * FG_BM + (tid >> 1)) * 8 + (tid & 1) * 4
                      : (EPI == 6) ? A.geo + ((size_t)tile0 * FG_BM + (tid & 127)) * 4 : nullptr; // EPI 6: both halves of the workgroup fetch the row's record (no load under a branch)
    int ltp = 0; // parity of the tile being LOADED
    // SRC 4: the geo cursor runs ONE SLAB AHEAD of the operand cursor (qn = the geo of this thread's two rows of the slab the next
    // issue_loads call fetches); dq0 / dq1 in a register set are the dxyz of the rows of ITS slab
    const float4 *pg = (SRC == 4) ? reinterpret_cast<const float4 *>(A.geo) + (size_t)tile0 * FG_BM + a_row : nullptr;
    float4 qn0 = make_float4(0.f, 0.f, 0.f, 0.f), qn1 = qn0;
    int glkt = 0;
    long gsteps = steps_to_load;
    auto geo_next = [&]() { // load the cursor's geo, then move the cursor one slab on (same no-branch-around-a-load rule as below)
        qn0 = pg[0];
        qn1 = pg[64];
        if (gsteps > 1 && ++glkt == nk) {
            glkt = 0;
            pg += (size_t)tstride * FG_BM;
        }
        --gsteps;
    };
    // SRC 2: pooled upstream gradient and arg-max of this thread's two rows (group = row / pool_k)
    const int pk = (SRC == 2) ? A.pool_k : 1;
    long g0 = 0, g1 = 0;    // groups of the two staged rows of the step being LOADED
    int ro0 = 0, ro1 = 0;   // their row offsets inside the group
    if (SRC == 2) {
        const long r0 = (long)tile0 * FG_BM + a_row;
        g0 = r0 / pk;
        ro0 = (int)(r0 - g0 * pk);
        g1 = (r0 + 64) / pk;
        ro1 = (int)(r0 + 64 - g1 * pk);
    }
    const float *pb[NB4];
#pragma unroll
    for (int u = 0; u < NB4; u++) {
        const int f = tid + u * 256;
        pb[u] = w + ((size_t)ks0 * FG_BK + (size_t)(f / (BN / 4))) * cout + n0 + (f % (BN / 4)) * 4;
    }
    const size_t b_step = (size_t)FG_BK * cout, b_wrap = (size_t)nk * FG_BK * cout;
    // BF3: W comes pre-split (votenet_split_weights): per slab [piece][k-half][column][8 bf16], i.e. the LDS image itself.  A slab
    // of this column block is 6 planes of BN x 16 bytes; thread t copies chunk t of planes (2u + t/128), u = 0..2 (16-byte chunks
    // at BN = 128, 8-byte chunks at BN = 64): one 32-bit lane offset that walks the slabs, the plane pair u as a scalar offset
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)A.w3, 0, 0x7fffffff, 0x00020000);
    const int b3_pl = tid >> 7, b3_c = (BN == 128) ? (tid & 127) : ((tid & 127) >> 1), b3_h = (BN == 128) ? 0 : (tid & 1);
    unsigned wvo = (((unsigned)b3_pl * (unsigned)cout + (unsigned)(n0 + b3_c)) * 4u + (unsigned)b3_h * 2u) * 4u;
    unsigned wpitch = (unsigned)cout * 32u; // bytes between plane pairs (= pieces)
    const unsigned w3_slab = (unsigned)cout * 32u * (unsigned)NPC; // bytes per slab of the split image
    wvo += (unsigned)ks0 * w3_slab;
    int lkt = 0; // k-slab index (from ks0) of the step being loaded
    // One k-slab step of raw operands in registers.  Two sets alternate: a set is filled two steps before its slab
    // is needed in LDS, so the global loads have two steps of matrix work (2 x 2048 MFMA cycles) to arrive -- one step
    // is less than the loaded HBM latency, which serialised memory time and matrix time.
    struct Regs {
        float4 a0, a1, b[BF3 ? 1 : NB4];
        uint4 bq[BF3 ? NPC : 1]; // BF3: this thread's chunk of the three (H2: two) pieces of the W slab (x, y only at BN = 64)
        float4 g0, g1; // SRC 1: da quads; SRC 2: gout quads; SRC 3: the second half of the rows' u
        int4 m0, m1;   // SRC 2: arg-max quads
        int k, ro0, ro1; // k / row offsets of the quads
        float4 uq;     // EPI 4: this thread's quad of the tile's u rows
        int tp;        //        and the tile's parity
        float4 dq0, dq1; // SRC 4: geo of the two rows (dxyz used when the slab goes to LDS)
        float mu0, mu1;  // SRC 5: weights of the two rows' pieces
        uint2 mq;        // EPI 7: the mask words of tile row tid & 127
        int row;         // SRC 3: the global row of a0 (a1: + 64), for mask_out
    };
    constexpr int NSETS = H2 ? H2_SETS : BF3 ? BF3_SETS : 2; // register sets = slabs in flight; the slab loop is unrolled by it (launcher: nk % NSETS == 0)
    Regs R[NSETS];
    bool abl_prologue = true;
    auto issue_loads = [&](Regs &r) {
        if (BF3 && (BF3_ABL & 4) && !abl_prologue) return;
        const int kq = (ks0 + lkt) * FG_BK + a_kq * 4;
        if (SRC == 4) {
            r.dq0 = qn0; // the geo of THIS slab's rows: loaded by the previous call, as the oldest of its loads
            r.dq1 = qn1;
            geo_next();  // first load of this call: the geo of the slab the next call fetches
            r.a0 = *reinterpret_cast<const float4 *>(A.ptab + (size_t)__float_as_uint(r.dq0.w) * cin + kq);
            r.a1 = *reinterpret_cast<const float4 *>(A.ptab + (size_t)__float_as_uint(r.dq1.w) * cin + kq);
        } else {
            r.a0 = *reinterpret_cast<const float4 *>(pa0);
            r.a1 = *reinterpret_cast<const float4 *>(pa1);
        }
        if (SRC == 1 || SRC == 5) {
            r.g0 = *reinterpret_cast<const float4 *>(pa0 + da_off);
            r.g1 = *reinterpret_cast<const float4 *>(pa1 + da_off);
            if (SRC == 5) {
                r.mu0 = pw[0];
                r.mu1 = pw[64 / kPiece];
            }
        } else if (SRC == 3) {
            r.g0 = *reinterpret_cast<const float4 *>(pa0 + 4);
            r.g1 = *reinterpret_cast<const float4 *>(pa1 + 4);
        } else if (SRC == 2) {
            r.g0 = *reinterpret_cast<const float4 *>(A.gout + (size_t)g0 * cin + kq);
            r.g1 = *reinterpret_cast<const float4 *>(A.gout + (size_t)g1 * cin + kq);
            r.m0 = *reinterpret_cast<const int4 *>(A.argmax + (size_t)g0 * cin + kq);
            r.m1 = *reinterpret_cast<const int4 *>(A.argmax + (size_t)g1 * cin + kq);
            r.ro0 = ro0;
            r.ro1 = ro1;
        }
        if constexpr (BF3) {
#pragma unroll
            for (int u = 0; u < NPC; u++) {
                if ((BF3_ABL & 16) && !abl_prologue) break; // probe: the W image neither loaded nor staged after the prologue
                if constexpr (BN == 128) {
                    r.bq[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, (unsigned)u * wpitch, 0));
                } else {
                    const uint2 t2 = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(wrs, wvo, (unsigned)u * wpitch, 0));
                    r.bq[u].x = t2.x;
                    r.bq[u].y = t2.y;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NB4; u++) r.b[u] = *reinterpret_cast<const float4 *>(pb[u]);
        }
        if (NEPI || EPI == 6) {
            r.uq = *reinterpret_cast<const float4 *>(pu);
            r.tp = ltp;
        }
        if (EPI == 7) r.mq = *pm;
        if (SRC == 3) r.row = (int)((pa0 - A.u8) >> 3);
        r.k = kq;
        if constexpr (BF3) {
            // the same cursor movement as below without a branch (scalar selects): the slab body stays ONE basic block, which is
            // what lets its MFMAs and the staging arithmetic be interleaved (sched_group_barrier in the loop)
            const bool adv = steps_to_load > 1;
            const bool wrap = adv && (lkt + 1 == nk);
            const ptrdiff_t astep = adv ? (ptrdiff_t)a_slab_step + (wrap ? (ptrdiff_t)a_tile_jump : 0) : 0;
            pa0 += astep;
            pa1 += astep;
            wvo += adv ? (wrap ? w3_slab - w3_slab * (unsigned)nk : w3_slab) : 0u;
            lkt = wrap ? 0 : (adv ? lkt + 1 : lkt);
            if (NEPI || EPI == 6) {
                pu += wrap ? (size_t)tstride * FG_BM * (NEPI ? 8 : 4) : 0;
                ltp ^= wrap ? 1 : 0;
            }
            if (EPI == 7) pm += wrap ? (size_t)tstride * FG_BM * (cout / 64) : 0;
            if (SRC == 5) pw += wrap ? (size_t)tstride * (FG_BM / kPiece) : 0;
            if (SRC == 2) {
                const long dg = wrap ? (long)tstride * FG_BM / pk : 0;
                g0 += dg;
                g1 += dg;
            }
            --steps_to_load;
            return;
        }
        // Advance to the next slab only if there is one: past the end the same (valid) slab is simply loaded again, so
        // the loop body has no memory operation under a branch and the compiler's s_waitcnt bookkeeping stays exact
        // (a conditional load makes it wait for vmcnt(0), which collapses the two-deep pipeline to one).
        if (steps_to_load > 1) {
            pa0 += a_slab_step;
            pa1 += a_slab_step;
#pragma unroll
            for (int u = 0; u < NB4; u++) pb[u] += b_step;
            if (++lkt == nk) {
                lkt = 0;
                pa0 += a_tile_jump;
                pa1 += a_tile_jump;
                if (NEPI || EPI == 6) {
                    pu += (size_t)tstride * FG_BM * (NEPI ? 8 : 4);
                    ltp ^= 1;
                }
                if (EPI == 7) pm += (size_t)tstride * FG_BM * (cout / 64);
                if (SRC == 5) pw += (size_t)tstride * (FG_BM / kPiece);
#pragma unroll
                for (int u = 0; u < NB4; u++) pb[u] -= b_wrap;
                if (SRC == 2) { // next tile: rows advance by gridDim.x*128, a multiple of pool_k (checked by the launcher)
                    const long dg = (long)tstride * FG_BM / pk;
                    g0 += dg;
                    g1 += dg;
                }
            }
        }
        --steps_to_load;
    };
    const float relu_floor = (affine && A.in_relu) ? 0.0f : -__builtin_inff(); // BF3: the folded ReLU as a select against a uniform floor
    auto act4 = [&](float4 v, const float4 &g, const int4 &am, int ro, int rk, float mu = 1.0f) {
        // rk is a multiple of 4 (the thread's channel quad): said aloud so that the table reads below are ONE 16-byte LDS read each with an
        // immediate offset (unproven, the compiler split every float4 of W0s / Sco / Wxs into two ds_read2_b32 behind their own address adds:
        // 18 + 18 instructions per slab in the narrow loader)
        __builtin_assume((rk & 3) == 0);
        if (SRC == 4) { // v = the P quad of the row, g = its geo: z0 of the four channels rk..rk+3
            const float4 w0 = *reinterpret_cast<const float4 *>(&Wxs[0][rk]), w1 = *reinterpret_cast<const float4 *>(&Wxs[1][rk]);
            const float4 w2 = *reinterpret_cast<const float4 *>(&Wxs[2][rk]);
            const f32x2 z01 = assembled_z2(v.x, v.y, g, w0.x, w0.y, w1.x, w1.y, w2.x, w2.y);
            const f32x2 z23 = assembled_z2(v.z, v.w, g, w0.z, w0.w, w1.z, w1.w, w2.z, w2.w);
            v.x = z01.x;
            v.y = z01.y;
            v.z = z23.x;
            v.w = z23.y;
        }
        if (SRC == 3) { // v, g = the row's u[0..4), u[4..8): z0 of the four channels rk..rk+3, then the folded BatchNorm + ReLU below
            const float uu[8] = {v.x, v.y, v.z, v.w, g.x, g.y, g.z, g.w};
            float wq[4][8];
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const float4 w4 = *reinterpret_cast<const float4 *>(&W0s[d][rk]);
                wq[0][d] = w4.x;
                wq[1][d] = w4.y;
                wq[2][d] = w4.z;
                wq[3][d] = w4.w;
            }
            const float4 b4 = *reinterpret_cast<const float4 *>(&W0s[8][rk]);
#ifdef NARROW_ABLATE
            v.x = uu[0] + b4.x + wq[0][0];
            v.y = uu[1] + b4.y + wq[1][1];
            v.z = uu[2] + b4.z + wq[2][2];
            v.w = uu[3] + uu[7] + b4.w + wq[3][3];
#else
            const f32x2 z01 = narrow_z2(uu, wq[0], wq[1], b4.x, b4.y), z23 = narrow_z2(uu, wq[2], wq[3], b4.z, b4.w);
            v.x = z01.x;
            v.y = z01.y;
            v.z = z23.x;
            v.w = z23.y;
#endif
        }
        if constexpr (BF3 && (SRC == 0 || SRC == 3 || SRC == 4)) {
            const float4 sc = *reinterpret_cast<const float4 *>(&Sco[0][rk]);
            const float4 sh = *reinterpret_cast<const float4 *>(&Sco[1][rk]);
            // the same two roundings as the fp32 loader (mul, add: every consumer of this BatchNorm sees one ReLU mask); the ReLU as
            // one v_max against the uniform floor (the staging arithmetic is issue-bound: T ~ 4 N_valu + 32 N_mfma per SIMD)
            // (as packed pairs: v_pk_mul_f32 + v_pk_add_f32 -- the same two roundings per element, half the instructions; every operand
            // a real register pair, no broadcast: nothing for op_sel to select)
            const f32x2 t01 = f32x2{v.x, v.y} * f32x2{sc.x, sc.y} + f32x2{sh.x, sh.y};
            const f32x2 t23 = f32x2{v.z, v.w} * f32x2{sc.z, sc.w} + f32x2{sh.z, sh.w};
            v.x = __builtin_fmaxf(t01.x, relu_floor);
            v.y = __builtin_fmaxf(t01.y, relu_floor);
            v.z = __builtin_fmaxf(t23.x, relu_floor);
            v.w = __builtin_fmaxf(t23.y, relu_floor);
            return v;
        }
        if (SRC == 0 || SRC == 3 || SRC == 4) {
            if (affine) {
                const float4 sc = *reinterpret_cast<const float4 *>(&Sco[0][rk]);
                const float4 sh = *reinterpret_cast<const float4 *>(&Sco[1][rk]);
                v.x = v.x * sc.x + sh.x;
                v.y = v.y * sc.y + sh.y;
                v.z = v.z * sc.z + sh.z;
                v.w = v.w * sc.w + sh.w;
                if (A.in_relu) {
                    v.x = v.x > 0.f ? v.x : 0.f;
                    v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f;
                    v.w = v.w > 0.f ? v.w : 0.f;
                }
            }
            return v;
        }
        // BatchNorm backward: v is the zsrc quad
        float gg[4] = {g.x, g.y, g.z, g.w};
        const float zz[4] = {v.x, v.y, v.z, v.w};
        if (SRC == 2) {
            gg[0] = (am.x == ro) ? gg[0] : 0.f;
            gg[1] = (am.y == ro) ? gg[1] : 0.f;
            gg[2] = (am.z == ro) ? gg[2] : 0.f;
            gg[3] = (am.w == ro) ? gg[3] : 0.f;
        }
        const float4 cA = *reinterpret_cast<const float4 *>(&Sco[0][rk]);
        const float4 cB = *reinterpret_cast<const float4 *>(&Sco[1][rk]);
        const float4 cC = *reinterpret_cast<const float4 *>(&Sco[2][rk]);
        const float cAa[4] = {cA.x, cA.y, cA.z, cA.w}, cBa[4] = {cB.x, cB.y, cB.z, cB.w}, cCa[4] = {cC.x, cC.y, cC.z, cC.w};
        float o[4];
        if (A.src_relu) {
            const float4 cS = *reinterpret_cast<const float4 *>(&Sco[3][rk]);
            const float4 cH = *reinterpret_cast<const float4 *>(&Sco[4][rk]);
            const float cSa[4] = {cS.x, cS.y, cS.z, cS.w}, cHa[4] = {cH.x, cH.y, cH.z, cH.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (!(zz[q] * cSa[q] + cHa[q] > 0.0f)) gg[q] = 0.0f;
        }
        if (SRC == 5) { // total gradient of a row that stands for mu true rows: the upstream part is a total already, the affine part is per row
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = cAa[q] * gg[q] + mu * (cBa[q] + cCa[q] * zz[q]);
            return make_float4(o[0], o[1], o[2], o[3]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) o[q] = cAa[q] * gg[q] + cBa[q] + cCa[q] * zz[q];
        return make_float4(o[0], o[1], o[2], o[3]);
    };
    auto store_regs = [&](int buf, const Regs &r) {
        if (BF3 && (BF3_ABL & 8) && !abl_prologue) return;
        const float4 v0 = act4(r.a0, SRC == 4 ? r.dq0 : r.g0, r.m0, r.ro0, r.k, (SRC == 5 && sel31) ? r.mu0 : 1.0f),
                     v1 = act4(r.a1, SRC == 4 ? r.dq1 : r.g1, r.m1, r.ro1, r.k, (SRC == 5 && sel31) ? r.mu1 : 1.0f);
        if (SRC == 3 && A.mask_out != nullptr) {
            // the ReLU mask of the narrow layer for the backward pass (EPI 7): this thread's four channels of its two rows; the four
            // threads of a row (a_kq = tid & 3: adjacent lanes) OR their nibbles into the slab's 16-bit word and all four store it
            // (same address, same value: no store under a lane-dependent branch)
            const unsigned m0 = (v0.x > 0.f ? 1u : 0u) | (v0.y > 0.f ? 2u : 0u) | (v0.z > 0.f ? 4u : 0u) | (v0.w > 0.f ? 8u : 0u);
            const unsigned m1 = (v1.x > 0.f ? 1u : 0u) | (v1.y > 0.f ? 2u : 0u) | (v1.z > 0.f ? 4u : 0u) | (v1.w > 0.f ? 8u : 0u);
            unsigned mm = (m0 | (m1 << 16)) << (4 * a_kq);
            mm |= __shfl_xor(mm, 1);
            mm |= __shfl_xor(mm, 2);
            const size_t wo = (size_t)r.row * nk + (r.k >> 4);
            A.mask_out[wo] = (unsigned short)(mm & 0xffffu);
            A.mask_out[wo + (size_t)64 * nk] = (unsigned short)(mm >> 16);
        }
        if constexpr (H2) {
            unsigned h[4], l[4];
            split2(v0.x, v0.y, h[0], l[0]);
            split2(v0.z, v0.w, h[1], l[1]);
            split2(v1.x, v1.y, h[2], l[2]);
            split2(v1.z, v1.w, h[3], l[3]);
            const int ao = a_row * 4 + (a_kq & 1) * 2, ap = a_kq >> 1;
            *reinterpret_cast<uint2 *>(&As3[buf][0][ap][ao]) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2 *>(&As3[buf][1][ap][ao]) = make_uint2(l[0], l[1]);
            *reinterpret_cast<uint2 *>(&As3[buf][0][ap][ao + 256]) = make_uint2(h[2], h[3]);
            *reinterpret_cast<uint2 *>(&As3[buf][1][ap][ao + 256]) = make_uint2(l[2], l[3]);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if ((BF3_ABL & 16) && !abl_prologue) break;
                if constexpr (BN == 128) *reinterpret_cast<uint4 *>(&Bs3[buf][u][b3_pl][b3_c * 4]) = r.bq[u];
                else *reinterpret_cast<uint2 *>(&Bs3[buf][u][b3_pl][b3_c * 4 + b3_h * 2]) = make_uint2(r.bq[u].x, r.bq[u].y);
            }
            if (NEPI) *reinterpret_cast<float4 *>(&Us[r.tp][tid >> 1][(tid & 1) * 4]) = r.uq;
            if (EPI == 7) Ms[r.tp][tid & 127] = r.mq;
            if (EPI == 6) Gs[r.tp][tid & 127] = r.uq;
            return;
        } else if constexpr (BF3) {
            unsigned h[4], m[4], l[4];
            split3(v0.x, v0.y, h[0], m[0], l[0]);
            split3(v0.z, v0.w, h[1], m[1], l[1]);
            split3(v1.x, v1.y, h[2], m[2], l[2]);
            split3(v1.z, v1.w, h[3], m[3], l[3]);
            const int ao = a_row * 4 + (a_kq & 1) * 2, ap = a_kq >> 1;
            *reinterpret_cast<uint2 *>(&As3[buf][0][ap][ao]) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2 *>(&As3[buf][1][ap][ao]) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2 *>(&As3[buf][2][ap][ao]) = make_uint2(l[0], l[1]);
            *reinterpret_cast<uint2 *>(&As3[buf][0][ap][ao + 256]) = make_uint2(h[2], h[3]);
            *reinterpret_cast<uint2 *>(&As3[buf][1][ap][ao + 256]) = make_uint2(m[2], m[3]);
            *reinterpret_cast<uint2 *>(&As3[buf][2][ap][ao + 256]) = make_uint2(l[2], l[3]);
#pragma unroll
            for (int u = 0; u < 3; u++) {
                if ((BF3_ABL & 16) && !abl_prologue) break;
                if constexpr (BN == 128) *reinterpret_cast<uint4 *>(&Bs3[buf][u][b3_pl][b3_c * 4]) = r.bq[u];
                else *reinterpret_cast<uint2 *>(&Bs3[buf][u][b3_pl][b3_c * 4 + b3_h * 2]) = make_uint2(r.bq[u].x, r.bq[u].y);
            }
            if (NEPI) *reinterpret_cast<float4 *>(&Us[r.tp][tid >> 1][(tid & 1) * 4]) = r.uq;
            if (EPI == 7) Ms[r.tp][tid & 127] = r.mq;
            if (EPI == 6) Gs[r.tp][tid & 127] = r.uq;
            return;
        }
        As[buf][a_kq * 4 + 0][a_row] = v0.x;
        As[buf][a_kq * 4 + 1][a_row] = v0.y;
        As[buf][a_kq * 4 + 2][a_row] = v0.z;
        As[buf][a_kq * 4 + 3][a_row] = v0.w;
        As[buf][a_kq * 4 + 0][a_row + 64] = v1.x;
        As[buf][a_kq * 4 + 1][a_row + 64] = v1.y;
        As[buf][a_kq * 4 + 2][a_row + 64] = v1.z;
        As[buf][a_kq * 4 + 3][a_row + 64] = v1.w;
#pragma unroll
        for (int u = 0; u < NB4; u++) {
            const int f = tid + u * 256;
            *reinterpret_cast<float4 *>(&Bs[buf][f / (BN / 4)][(f % (BN / 4)) * 4]) = r.b[u];
        }
        if (NEPI) *reinterpret_cast<float4 *>(&Us[r.tp][tid >> 1][(tid & 1) * 4]) = r.uq;
        if (EPI == 7) Ms[r.tp][tid & 127] = r.mq;
        if (EPI == 6) Gs[r.tp][tid & 127] = r.uq;
    };

    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) s1[j] = s2[j] = 0.0f;
    float ugs[NEPI ? 8 : 1][NT]; // EPI 4 / 7: sum_r u[r,d] da'[r,col] over this lane's rows
#pragma unroll
    for (int d = 0; d < (NEPI ? 8 : 1); d++)
#pragma unroll
        for (int j = 0; j < NT; j++) ugs[d][j] = 0.0f;
    if (my_tiles == 0) { // never with the launchers below (gridDim.x <= row tiles); a workgroup without work still takes its ticket
        if (REDUCE_BELOW) coef_tail(A.tail, gridDim.x * gridDim.y * (SK ? gridDim.z : 1u), cout, A.stats, A.e_scale, A.e_shift, A.e_mean, A.e_var, A.e_eps);
        return;
    }
    // Round 5: the FIRST operand loads go out before anything else touches memory.  The per-channel tables below (BatchNorm of the
    // input from raw sums / coefficient vector / the layer below's BatchNorm / bias) are a chain of small dependent round trips --
    // kernel arguments, table loads in loops, LDS, barrier -- that used to run BEFORE the first operand load was issued: 5-6 memory
    // latencies in a row ahead of the first MFMA, a third of a 15 us launch of the static stretch.  Now they wait together.
    if (SRC == 4) geo_next(); // qn = slab 0's geo (the cursor then stands on slab 1)
    issue_loads(R[0]);
    // (the bias and, pool32, the pooled layer's gamma -- the sign of the scale the pool will apply -- of this lane's columns: loaded
    // here, consumed below)
    float bvs[NT], sgs[NT], uns[NT]; // uns: what scales an accumulator back (1 unless H2)
#pragma unroll
    for (int j = 0; j < NT; j++) {
        bvs[j] = A.bias ? A.bias[n0 + (wv % WN * NT + j) * 32 + (lane & 31)] : 0.0f;
        sgs[j] = P32 ? A.pool_gamma[n0 + (wv % WN * NT + j) * 32 + (lane & 31)] : 1.0f;
        uns[j] = (H2 && A.h2_unscale) ? A.h2_unscale[n0 + (wv % WN * NT + j) * 32 + (lane & 31)] : kH2Un;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (SRC == 4)
        for (int t = tid; t < 3 * cin; t += 256) Wxs[t / cin][t % cin] = A.wx[t];
    if (NARROW) {
        const int c0 = (SRC == 3) ? cin : cout, cb = (SRC == 3) ? 0 : n0, cw = (SRC == 3) ? cin : BN;
        for (int t = tid; t < 9 * cw; t += 256) {
            const int d = t / cw, c = t % cw;
            W0s[d][c] = d < 8 ? (d < A.k0 ? A.w0[(size_t)d * c0 + cb + c] : 0.0f) : (A.b0 ? A.b0[cb + c] : 0.0f);
        }
    }
    if (SRC == 0 || SRC == 3 || SRC == 4) {
        if (A.in_raw.stats) { // the producer's BatchNorm, finalized here; workgroup (0,0) records it for the backward pass
            const bool writer = blockIdx.x == 0 && blockIdx.y == 0 && (!SK || blockIdx.z == 0);
            for (int k = tid; k < cin; k += 256) {
                float sc, sh;
                bn_raw_channel(A.in_raw, cin, k, writer, sc, sh);
                const float ka = (H2 && A.h2_ascale) ? A.h2_ascale[k] : kH2A;
                Sco[0][k] = sc * ka;
                Sco[1][k] = sh * ka;
            }
        } else if (affine)
            for (int k = tid; k < cin; k += 256) {
                const float ka = (H2 && A.h2_ascale) ? A.h2_ascale[k] : kH2A;
                Sco[0][k] = A.in_scale[k] * ka;
                Sco[1][k] = A.in_shift[k] * ka;
            }
        else if (BF3) // the BF3 loader has no branch on `affine`: x * 1 + 0
            for (int k = tid; k < cin; k += 256) {
                Sco[0][k] = (H2 && A.h2_ascale) ? A.h2_ascale[k] : kH2A;
                Sco[1][k] = 0.0f;
            }
    } else {
        for (int k = tid; k < 5 * cin; k += 256) Sco[k / cin][k % cin] = A.coef[k];
    }
    if (REDUCE_BELOW)
        for (int cidx = tid; cidx < BN; cidx += 256) {
            Eco[0][cidx] = A.e_scale[n0 + cidx];
            Eco[1][cidx] = A.e_shift[n0 + cidx];
            Eco[2][cidx] = A.e_mean[n0 + cidx];
            Eco[3][cidx] = 1.0f / sqrtf(A.e_var[n0 + cidx] + A.e_eps);
        }
    __syncthreads(); // Sco
    // bias of this lane's columns, loaded ONCE and consumed (the empty asm) before any prefetch is in flight: a load inside
    // the per-tile epilogue -- or one still pending in the compiler's bookkeeping -- costs an s_waitcnt vmcnt(0) there,
    // i.e. drains both prefetch sets at every tile boundary (every 4 slabs at cin = 64)
#pragma unroll
    for (int j = 0; j < NT; j++) {
        asm volatile("" : "+v"(bvs[j]));
        if (H2) asm volatile("" : "+v"(uns[j]));
        sgs[j] = (P32 && sgs[j] < 0.0f) ? -1.0f : 1.0f;
        asm volatile("" : "+v"(sgs[j]));
    }
    // EPI 0 / 2 (not the 64-row pool): the statistics as packed pairs (v_pk_add_f32 / v_pk_fma_f32: two accumulator elements per instruction)
    f32x2 s1p[NT], s2p[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) s1p[j] = s2p[j] = f32x2{0.0f, 0.0f};
    // prologue: slab 0 (in flight since the top) -> LDS buffer 0; slabs 1 and 2 in flight in register sets 1 and 0
    store_regs(0, R[0]);
#pragma unroll
    for (int q = 1; q <= NSETS; q++) { // same issue order as in the loop: sets 1 .. NSETS-1, then set 0
        issue_loads(R[q % NSETS]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    const int kh = lane >> 5, l31 = lane & 31;
    int buf = 0;
    abl_prologue = false;
    constexpr int MX = MT > NT ? MT : NT;
    uint4 fAlo[MX], fWhi[MX], fAhi[MX], fWmid[MX], fX[2][MX]; // BF3: fragments that live across slabs (see the slab body)
    if constexpr (H2) {
#pragma unroll
        for (int i = 0; i < MT; i++) fAlo[i] = *reinterpret_cast<const uint4 *>(&As3[0][1][kh][((wm * MT + i) * 32 + l31) * 4]);
#pragma unroll
        for (int j = 0; j < NT; j++) fWhi[j] = *reinterpret_cast<const uint4 *>(&Bs3[0][0][kh][((wn * NT + j) * 32 + l31) * 4]);
    } else if constexpr (BF3) {
#pragma unroll
        for (int i = 0; i < MT; i++) {
            fAlo[i] = *reinterpret_cast<const uint4 *>(&As3[0][NPC - 1][kh][((wm * MT + i) * 32 + l31) * 4]);
            fX[0][i] = *reinterpret_cast<const uint4 *>(&As3[0][1][kh][((wm * MT + i) * 32 + l31) * 4]);
        }
#pragma unroll
        for (int j = 0; j < NT; j++) fWhi[j] = *reinterpret_cast<const uint4 *>(&Bs3[0][0][kh][((wn * NT + j) * 32 + l31) * 4]);
    }
    bool sk_contrib = true; // split-K: false once this workgroup handed a partial tile over instead of running the epilogue
    for (long t = 0; t < my_tiles; t++) {
        if (BF3 && BF3_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (BF3 && BF3_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
        // nk is even (launcher): step parity == kt parity, so the register set of a step is a compile-time constant.
        // During step g the set (g+1)&1 holds slab g+1: half way through it goes to the other LDS buffer and the set is
        // refilled with slab g+3.
        for (int kt = 0; kt < nk; kt += NSETS) {
#pragma unroll
            for (int par = 0; par < NSETS; par++) {
                Regs &rs = R[(par + 1) % NSETS];
                if constexpr (BF3) {
                    // One slab = ONE k-step of v_mfma_f32_32x32x16_bf16 per piece pair, six per 32x32 sub-tile.  The slab's barrier
                    // sits in the MIDDLE: the products before it run while the next slab is staged into the other buffer, the
                    // products behind it while the first fragments of the next slab (A.lo, W.hi, A.mid) are read from that buffer
                    // into registers whose pieces are dead by then -- the matrix pipe never waits for an LDS read behind a
                    // barrier.  (With the barrier at the end of the slab the two waves of a SIMD fall into step: both read
                    // fragments, then both want the matrix pipe; the bare loop ran at 68 % of it.)
                    // Pieces: A.lo, W.hi, A.hi, W.mid in fixed registers; A.mid and W.lo swap two blocks by slab parity.
                    auto rdA = [&](uint4(&f)[MX], int piece, int b) {
#pragma unroll
                        for (int i = 0; i < MT; i++) f[i] = *reinterpret_cast<const uint4 *>(&As3[b][piece][kh][((wm * MT + i) * 32 + l31) * 4]);
                    };
                    auto rdW = [&](uint4(&f)[MX], int piece, int b) {
#pragma unroll
                        for (int j = 0; j < NT; j++) f[j] = *reinterpret_cast<const uint4 *>(&Bs3[b][piece][kh][((wn * NT + j) * 32 + l31) * 4]);
                    };
                    auto mm = [&](const uint4(&fa_)[MX], const uint4(&fw_)[MX]) {
                        if (BF3_ABL & 1) {
                            acc[0][0][0] += __uint_as_float(fa_[0].x ^ fw_[0].y);
                            return;
                        }
                        if (BF3_PRIO == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int i = 0; i < MT; i++)
#pragma unroll
                            for (int j = 0; j < NT; j++)
                                if constexpr (H2)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa_[i]),
                                                                                       __builtin_bit_cast(f16x8, fw_[j]), acc[i][j], 0, 0, 0);
                                else
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa_[i]),
                                                                                        __builtin_bit_cast(bf16x8, fw_[j]), acc[i][j], 0, 0, 0);
                        if (BF3_PRIO == 3) __builtin_amdgcn_s_setprio(0);
                    };
                    if constexpr (H2) {
                        // two pieces, three products: A.lo and W.hi of this slab are in registers since the previous slab's barrier
                        rdA(fAhi, 0, buf);
                        rdW(fWmid, 1, buf);      // W.lo (in the block the three-piece schedule keeps W.mid in)
                        mm(fAlo, fWhi);
                        store_regs(buf ^ 1, rs); // the other buffer was last read before the previous slab's barrier
                        mm(fAhi, fWhi);
                        lds_barrier();
                        rdA(fAlo, 1, buf ^ 1);   // the next slab's A.lo and W.hi: their blocks are free from here on
                        rdW(fWhi, 0, buf ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                        issue_loads(rs);
                        mm(fAhi, fWmid);
                        buf ^= 1;
                        continue;
                    }
                    rdA(fAhi, 0, buf);
                    rdW(fWmid, 1, buf);
                    rdW(fX[par ^ 1], 2, buf); // W.lo
                    mm(fAlo, fWhi);
                    if (!(BF3_ABL & 32)) mm(fX[par], fWhi); // A.mid   (BF3_ABL 32, probe builds: three of the six products, timing only)
                    store_regs(buf ^ 1, rs); // the other buffer was last read before the previous slab's barrier
                    mm(fAhi, fWhi);
                    lds_barrier();
                    rdA(fAlo, 2, buf ^ 1);
                    rdW(fWhi, 0, buf ^ 1);
                    mm(fAhi, fX[par ^ 1]); // W.lo: its block is free from here on
                    __builtin_amdgcn_sched_barrier(0);
                    rdA(fX[par ^ 1], 1, buf ^ 1); // the next slab's A.mid
                    issue_loads(rs);
                    if (!(BF3_ABL & 32)) {
                        mm(fAhi, fWmid);
                        mm(fX[par], fWmid);
                    }
                    buf ^= 1;
                    continue;
                }
                float fa[2][MT], fb[2][NT];
#pragma unroll
                for (int i = 0; i < MT; i++) fa[0][i] = As[buf][kh][(wm * MT + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < NT; j++) fb[0][j] = Bs[buf][kh][(wn * NT + j) * 32 + l31];
#pragma unroll
                for (int k2 = 0; k2 < FG_BK / 2; k2++) {
                    if (k2 == FG_BK / 4) {
                        store_regs(buf ^ 1, rs); // the other buffer was last read one step ago, behind a barrier
                        issue_loads(rs);
                    }
                    if (k2 + 1 < FG_BK / 2) {
#pragma unroll
                        for (int i = 0; i < MT; i++) fa[(k2 + 1) & 1][i] = As[buf][(k2 + 1) * 2 + kh][(wm * MT + i) * 32 + l31];
#pragma unroll
                        for (int j = 0; j < NT; j++) fb[(k2 + 1) & 1][j] = Bs[buf][(k2 + 1) * 2 + kh][(wn * NT + j) * 32 + l31];
                    }
                    __builtin_amdgcn_sched_barrier(0); // keep the reads of k2+1 ahead of the MFMAs of k2
#pragma unroll
                    for (int i = 0; i < MT; i++)
#pragma unroll
                        for (int j = 0; j < NT; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k2 & 1][i], fb[k2 & 1][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                lds_barrier(); // LDS only: the prefetched global loads stay in flight across it
                buf ^= 1;
            }
        }
        if (BF3 && BF3_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (BF3 && BF3_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        if constexpr (SK) {
            // split-K: hand the partial tile over, or complete it (see FastArgs::sk_ws)
            const size_t tile_id = (size_t)(tile0 + t * tstride) * gridDim.y + blockIdx.y;
            float *wst = A.sk_ws + tile_id * (size_t)nsp * (FG_BM * BN) + tid;
            float *mine = wst + (size_t)blockIdx.z * (FG_BM * BN);
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        __hip_atomic_store(mine + ((i * NT + j) * 16 + e) * 256, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __shared__ unsigned s_sk_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                s_sk_last = (__hip_atomic_fetch_add(A.sk_ticket + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nsp - 1) ? 1u : 0u;
            __syncthreads();
            if (!s_sk_last) {
                sk_contrib = false;
                continue;
            }
            if (tid == 0) __hip_atomic_store(A.sk_ticket + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next launch on this slot
            // fixed order s = 0, 1, ...: the sum does not depend on which part arrived last (this workgroup's own part comes back from
            // the workspace like the others: its stores are acknowledged)
            for (int sp = 0; sp < nsp; sp++) {
                const float *part = wst + (size_t)sp * (FG_BM * BN);
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        float pv[16];
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            pv[e] = __hip_atomic_load(part + ((i * NT + j) * 16 + e) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[i][j][e] = sp == 0 ? pv[e] : acc[i][j][e] + pv[e];
                    }
            }
        }
        // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
        const long m0 = (tile0 + t * tstride) * FG_BM;
        // row pitch in bytes as an opaque scalar: the per-row scalar offsets of the buffer accesses below are then formed here,
        // per tile (a few s_mul), instead of being hoisted out of the tile loop as 64 loop-invariant SGPRs that spill
        unsigned pitch = (unsigned)cout * 4u;
        asm volatile("" : "+s"(pitch));
        float pmaxv[MT][NT], pminv[MT][NT];
        int pmaxi[MT][NT], pmini[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++) {
                pmaxv[i][j] = pminv[i][j] = 0.0f;
                pmaxi[i][j] = pmini[i][j] = 0;
            }
        if (BF3 && (BF3_ABL & 2)) {
            float t_ = 0.0f;
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++) t_ += acc[i][j][e];
            if (t_ == 12345.678f) z[0] = t_;
            continue;
        }
        const bool store_z = !NEPI && (!POOL || z != nullptr); // wave-uniform: the stores sit in their own loop nest so that the
        if (store_z) {                                   // pooling arithmetic below is not scheduled around 64 addresses
            // buffer stores: a scalar descriptor of this tile's rows, a scalar byte offset per (sub-tile, row) and ONE 32-bit lane
            // offset.  With flat 64-bit addresses the loop-invariant parts of the 64 addresses were hoisted out of the tile loop
            // and stayed live across the matrix loop (about 60 VGPRs: 216 in all, two waves per SIMD instead of three)
            const unsigned svoff = (unsigned)(4 * kh) * pitch + (unsigned)l31 * 4u;
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void *)(z + (size_t)m0 * cout + n0), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float bv = (EPI == 3) ? 0.0f : bvs[j];
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    const unsigned sbase = (unsigned)((wm * MT + i) * 32) * pitch + (unsigned)((wn * NT + j) * 32) * 4u;
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(H2 ? __builtin_fmaf(acc[i][j][e], uns[j], bv) : acc[i][j][e] + bv), zr, svoff,
                                                              sbase + (unsigned)((e & 3) + 8 * (e >> 2)) * pitch, 0);
                }
            }
        }
        if (EPI == 0 || P32) {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const f32x2 bv2 = {bvs[j], bvs[j]};
                const f32x2 un2 = {uns[j], uns[j]};
#pragma unroll
                for (int i = 0; i < MT; i++) {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        // broadcast operand FIRST: op_sel lands on src0 (packed-f32 hazard)
                        const f32x2 v = H2 ? __builtin_elementwise_fma(un2, f32x2{acc[i][j][e], acc[i][j][e + 1]}, bv2)
                                           : bv2 + f32x2{acc[i][j][e], acc[i][j][e + 1]};
                        s1p[j] += v;
                        s2p[j] = __builtin_elementwise_fma(v, v, s2p[j]);
                    }
                }
            }
        }
        if (EPI == 2) {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float bv = bvs[j];
#pragma unroll
                for (int i = 0; i < MT; i++) {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const float v = H2 ? __builtin_fmaf(acc[i][j][e], uns[j], bv) : acc[i][j][e] + bv;
                        {
                            const int rloc = 4 * kh + (e & 3) + 8 * (e >> 2); // inside the 32-row block; ascending in e: strict compares
                            if (e == 0 || v > pmaxv[i][j]) {                  // keep the first occurrence
                                pmaxv[i][j] = v;
                                pmaxi[i][j] = rloc;
                            }
                            if (e == 0 || v < pminv[i][j]) {
                                pminv[i][j] = v;
                                pmini[i][j] = rloc;
                            }
                        }
                        s1[j] += v;
                        s2[j] += v * v;
                    }
                }
            }
        }
        if (EPI == 0 || POOL) {
            if (A.wh != nullptr) {
                // piece layout: rows 0 and 16 of every 32-row block (accumulator elements 0 and 8 of the lower half-wave) are row 0 of a
                // piece and stand for wh rows.  The 2 MT weights of this wave's blocks are consecutive: ONE scalar load per tile
                // (s_load through the constant address space: lgkmcnt, not vmcnt -- round 4 had four global loads here, each under a
                // branch with its own s_waitcnt vmcnt(0): four serial memory round trips per tile that also drained the operand prefetch)
                static_assert(kPiece == 16, "the 32 x 32 MFMA tile holds two pieces: elements e < 8 and e >= 8");
                static_assert(MT == 1 || MT == 2, "2 or 4 piece weights per wave");
                const int pc0 = __builtin_amdgcn_readfirstlane((int)(m0 / kPiece) + wm * MT * 2);
                float whs[2 * MT];
                if constexpr (MT == 2) {
                    const f32x4 q = *reinterpret_cast<const __attribute__((address_space(4))) f32x4 *>((unsigned long)(A.wh + pc0));
                    whs[0] = q.x, whs[1] = q.y, whs[2] = q.z, whs[3] = q.w;
                } else {
                    const f32x2 q = *reinterpret_cast<const __attribute__((address_space(4))) f32x2 *>((unsigned long)(A.wh + pc0));
                    whs[0] = q.x, whs[1] = q.y;
                }
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    const float wa = (kh == 0) ? whs[2 * i] - 1.0f : 0.0f, wb = (kh == 0) ? whs[2 * i + 1] - 1.0f : 0.0f;
#pragma unroll
                    for (int j = 0; j < NT; j++) {
                        const float va = H2 ? __builtin_fmaf(acc[i][j][0], uns[j], bvs[j]) : acc[i][j][0] + bvs[j];
                        const float vb = H2 ? __builtin_fmaf(acc[i][j][8], uns[j], bvs[j]) : acc[i][j][8] + bvs[j];
                        s1[j] += wa * va + wb * vb;
                        s2[j] += wa * (va * va) + wb * (vb * vb);
                    }
                }
            }
        }
        if (EPI == 3) {
            const float thr = A.e_relu ? 0.0f : -__builtin_inff(); // no ReLU below: every element passes
            // buffer loads: scalar descriptor of this tile's rows + a scalar byte offset per (sub-tile, row) + ONE 32-bit lane
            // offset, so the sixty-four loads of a tile cost no address registers (64-bit flat addresses cost two each and
            // pushed the kernel to 256 VGPRs with spills)
            const unsigned voff = (unsigned)(4 * kh) * pitch + (unsigned)l31 * 4u;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc((void *)(A.ez + (size_t)m0 * cout + n0), 0, 0x7fffffff, 0x00020000);
            unsigned vo = voff;
            constexpr int CH = EPI3_CH < MT ? EPI3_CH : MT;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int cl = (wn * NT + j) * 32 + l31;
                const float sc = Eco[0][cl], sf = Eco[1][cl], mu = Eco[2][cl], inv = Eco[3][cl];
#pragma unroll
                for (int i0 = 0; i0 < MT; i0 += CH) {
                    float zz[CH][16];
#pragma unroll
                    for (int ii = 0; ii < CH; ii++) {
                        const unsigned sbase = (unsigned)((wm * MT + i0 + ii) * 32) * pitch + (unsigned)((wn * NT + j) * 32) * 4u;
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            zz[ii][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                rs, vo, sbase + (unsigned)((e & 3) + 8 * (e >> 2)) * pitch, 0));
                    }
#pragma unroll
                    for (int ii = 0; ii < CH; ii++)
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            float g = acc[i0 + ii][j][e]; // no bias in an input-gradient GEMM
                            if (!(zz[ii][e] * sc + sf > thr)) g = 0.0f;
                            s1[j] += g;
                            s2[j] += g * ((zz[ii][e] - mu) * inv);
                        }
                    // the next chunk's loads wait for this chunk's sums (a made-up dependence through the lane offset): CH x 16
                    // z values in flight, not all 64 -- the accumulators and both prefetch sets are live here
                    asm volatile("" : "+v"(vo) : "v"(s2[j]));
                }
            }
        }
        if (EPI == 6) {
            const float thr = A.e_relu ? 0.0f : -__builtin_inff();
            const int tp = (int)(t & 1);
            const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void *)(A.ptab + n0), 0, 0xffffffff, 0x00020000);
            float sc[NT], sf[NT], mu[NT], inv[NT], x0[NT], x1[NT], x2[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int cl = (wn * NT + j) * 32 + l31;
                sc[j] = Eco[0][cl];
                sf[j] = Eco[1][cl];
                mu[j] = Eco[2][cl];
                inv[j] = Eco[3][cl];
                x0[j] = A.wx[n0 + cl];
                x1[j] = A.wx[cout + n0 + cl];
                x2[j] = A.wx[2 * cout + n0 + cl];
            }
            int urow = (wm * MT) * 32 + 4 * kh; // this lane's first row of the tile
            constexpr int ER = EPI6_ROWS; // rows of a sub-tile column gathered together (8 or 16): ER x NT loads in flight
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int h = 0; h < 16 / ER; h++) { // ER rows at a time: e = ER h .. ER h + ER - 1
                    // phase 1: the rows' table offsets, then ER x NT gathers in flight; phase 2 reads the dxyz again from LDS
                    unsigned vo[ER];
#pragma unroll
                    for (int q = 0; q < ER; q++) {
                        const int e = ER * h + q;
                        vo[q] = __float_as_uint(Gs[tp][urow + i * 32 + (e & 3) + 8 * (e >> 2)].w) * pitch + (unsigned)l31 * 4u;
                    }
                    float pv[NT][ER];
#pragma unroll
                    for (int j = 0; j < NT; j++)
#pragma unroll
                        for (int q = 0; q < ER; q++)
                            pv[j][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(pr, vo[q], (unsigned)((wn * NT + j) * 32) * 4u, 0));
#pragma unroll
                    for (int q = 0; q < ER; q++) {
                        const int e = ER * h + q;
                        const float4 g4 = Gs[tp][urow + i * 32 + (e & 3) + 8 * (e >> 2)];
#pragma unroll
                        for (int j = 0; j < NT; j++) {
                            const float zz = assembled_z(pv[j][q], g4, x0[j], x1[j], x2[j]);
                            float g = acc[i][j][e];
                            if (!(zz * sc[j] + sf[j] > thr)) g = 0.0f;
                            s1[j] += g;
                            s2[j] += g * ((zz - mu[j]) * inv[j]);
                        }
                    }
                    asm volatile("" : "+v"(urow) : "v"(s2[0])); // the next rows' gathers wait for these sums: 8 x NT in flight
                }
        }
        if (EPI == 4) {
            const float thr = A.e_relu ? 0.0f : -__builtin_inff();
            const int tp = (int)(t & 1);
            float wc[NT][8], bc[NT], sc[NT], sf[NT], mu[NT], inv[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int cl = (wn * NT + j) * 32 + l31;
#pragma unroll
                for (int d = 0; d < 8; d++) wc[j][d] = W0s[d][cl];
                bc[j] = W0s[8][cl];
                sc[j] = Eco[0][cl];
                sf[j] = Eco[1][cl];
                mu[j] = Eco[2][cl];
                inv[j] = Eco[3][cl];
            }
            int urow = (wm * MT) * 32 + 4 * kh; // this lane's first row of the tile; four rows (e & 3) are read at a time
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    float4 ua[4], ub[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ua[q] = *reinterpret_cast<const float4 *>(&Us[tp][urow + i * 32 + 8 * e4 + q][0]);
                        ub[q] = *reinterpret_cast<const float4 *>(&Us[tp][urow + i * 32 + 8 * e4 + q][4]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int e = e4 * 4 + q;
                        const float uu[8] = {ua[q].x, ua[q].y, ua[q].z, ua[q].w, ub[q].x, ub[q].y, ub[q].z, ub[q].w};
#pragma unroll
                        for (int j = 0; j < NT; j++) {
                            const float zz = narrow_z(uu, wc[j], bc[j]);
                            float g = acc[i][j][e];
                            if (!(zz * sc[j] + sf[j] > thr)) g = 0.0f;
                            s1[j] += g;
                            s2[j] += g * ((zz - mu[j]) * inv[j]);
#pragma unroll
                            for (int d = 0; d < 8; d++) ugs[d][j] = __builtin_fmaf(uu[d], g, ugs[d][j]);
                        }
                    }
                    // the next four rows' LDS reads wait for these sums (a made-up dependence through the row index): 32 registers of
                    // u in flight instead of 8 x 16 x MT
                    asm volatile("" : "+v"(urow) : "v"(s2[0]));
                }
        }
        if (EPI == 7) {
            const int tp = (int)(t & 1);
            const unsigned all = A.e_relu ? 0u : 0xffffffffu;
            const bool k0_wide = A.k0 > 6;
            int urow = (wm * MT) * 32 + 4 * kh; // this lane's first row of the tile; four rows (e & 3) are read at a time
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    float4 ua[4], ub[4];
                    uint2 mk[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ua[q] = *reinterpret_cast<const float4 *>(&Us[tp][urow + i * 32 + 8 * e4 + q][0]);
                        ub[q] = *reinterpret_cast<const float4 *>(&Us[tp][urow + i * 32 + 8 * e4 + q][4]);
                        mk[q] = Ms[tp][urow + i * 32 + 8 * e4 + q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int e = e4 * 4 + q;
                        const float uu[8] = {ua[q].x, ua[q].y, ua[q].z, ua[q].w, ub[q].x, ub[q].y, ub[q].z, ub[q].w};
#pragma unroll
                        for (int j = 0; j < NT; j++) {
                            static_assert(EPI != 7 || (WN == 1 && NT == 2), "EPI 7: a 64-column block = the two words of a mask row");
                            const unsigned word = (j == 0 ? mk[q].x : mk[q].y) | all;
                            const float g = ((word >> l31) & 1u) ? acc[i][j][e] : 0.0f;
                            s1[j] += g;
#pragma unroll
                            for (int d = 0; d < 6; d++) ugs[d][j] = __builtin_fmaf(uu[d], g, ugs[d][j]);
                            if (k0_wide) { // (wave-uniform: sa1's rows have six channels -- 3 + 3 --, the padding multiplies zeros)
                                ugs[6][j] = __builtin_fmaf(uu[6], g, ugs[6][j]);
                                ugs[7][j] = __builtin_fmaf(uu[7], g, ugs[7][j]);
                            }
                        }
                    }
                    asm volatile("" : "+v"(urow) : "v"(s1[0])); // as EPI 4: the next four rows' LDS reads wait for these sums
                }
        }
        if (POOL) {
            // the other half-wave holds the interleaved rows of the same 32-row block: combine, smaller row wins ties
            if (!P32)
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const float ov = __shfl_xor(pmaxv[i][j], 32), uv = __shfl_xor(pminv[i][j], 32);
                    const int oi = __shfl_xor(pmaxi[i][j], 32), ui = __shfl_xor(pmini[i][j], 32);
                    if (ov > pmaxv[i][j] || (ov == pmaxv[i][j] && oi < pmaxi[i][j])) {
                        pmaxv[i][j] = ov;
                        pmaxi[i][j] = oi;
                    }
                    if (uv < pminv[i][j] || (uv == pminv[i][j] && ui < pmini[i][j])) {
                        pminv[i][j] = uv;
                        pmini[i][j] = ui;
                    }
                }
            if constexpr (P32) {
                // piece layout: every 16-row piece is a group of its own (votenet_bn_pool_finalize_half joins a centre's pieces).  A 32-row
                // block is two pieces -- elements e < 8 / e >= 8 of both half-waves -- done one after the other from the accumulators.
                // Branch-free (round 5: the compare-and-keep chain compiled into 115 exec-mask regions per tile): the maximum by v_max,
                // its FIRST row by an equality scan from the last element down, the other half-wave through v_permlane32_swap.
                const int kh4 = 4 * kh;
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const long pc = m0 / kPiece + (wm * MT + i) * 2 + hh;
#pragma unroll
                        for (int j = 0; j < NT; j++) {
                            // the sign of the BatchNorm scale is the sign of gamma: the max of sg * z, first occurrence, is the entry the
                            // pool takes (the max where the scale is >= 0, the min where it is negative); sg * (z + b) as one fma (sg = +-1: exact)
                            const float sg = sgs[j], sb = sg * bvs[j], sgu = sg * uns[j]; // (H2: the accumulators are 2^12 x the sums)
                            float v[8];
#pragma unroll
                            for (int e8 = 0; e8 < 8; e8++) v[e8] = __builtin_fmaf(sgu, acc[i][j][hh * 8 + e8], sb);
                            float best = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), __builtin_fmaxf(v[2], v[3])),
                                                         __builtin_fmaxf(__builtin_fmaxf(v[4], v[5]), __builtin_fmaxf(v[6], v[7])));
                            int ibest = kh4 + 11; // rloc(e8) = 4 kh + (e8 & 3) + 8 (e8 >> 2), ascending in e8
#pragma unroll
                            for (int e8 = 6; e8 >= 0; e8--) ibest = (v[e8] == best) ? kh4 + (e8 & 3) + 8 * (e8 >> 2) : ibest;
                            // lanes < 32 take the other half-wave's candidate (rows 4..7 / 12..15 of the piece); smaller row wins ties
                            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                            const auto si = __builtin_amdgcn_permlane32_swap((unsigned)ibest, (unsigned)ibest, false, false);
                            const float ov = __uint_as_float(sv[1]);
                            const int oi = (int)si[1];
                            const bool take = ov > best || (ov == best && oi < ibest);
                            best = take ? ov : best;
                            ibest = take ? oi : ibest;
                            if (lane < 32) {
                                const size_t o = (size_t)pc * cout + n0 + (wn * NT + j) * 32 + l31;
                                A.zmax[o] = sg * best; // the raw z
                                A.amax[o] = ibest;
                            }
                        }
                    }
            } else {
                // a wave's MT = 2 blocks are one 64-row group: the later block replaces the earlier one only when strictly better
                const long grp = m0 / 64 + wm;
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    float vmax = pmaxv[0][j], vmin = pminv[0][j];
                    int imax = pmaxi[0][j], imin = pmini[0][j];
#pragma unroll
                    for (int i = 1; i < MT; i++) {
                        if (pmaxv[i][j] > vmax) {
                            vmax = pmaxv[i][j];
                            imax = i * 32 + pmaxi[i][j];
                        }
                        if (pminv[i][j] < vmin) {
                            vmin = pminv[i][j];
                            imin = i * 32 + pmini[i][j];
                        }
                    }
                    if (lane < 32) {
                        const size_t o = (size_t)grp * cout + n0 + (wn * NT + j) * 32 + l31;
                        A.zmax[o] = vmax;
                        A.zmin[o] = vmin;
                        A.amax[o] = imax;
                        A.amin[o] = imin;
                    }
                }
            }
        }
    }
    if (EPI == 0 || P32) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            s1[j] += s1p[j].x + s1p[j].y;
            s2[j] += s2p[j].x + s2p[j].y;
        }
    }
    if (EPI != 1 && A.stats && sk_contrib) {
        // combine the WM waves that share a column block in LDS (the operand buffers are free now: every wave is past
        // the last step's barrier), then one atomic per column and statistic per workgroup: a column's address takes
        // gridDim.x atomics instead of WM*gridDim.x, which is what bounds the tail of the narrow (BN = 64) variant
        constexpr int NS = NEPI ? 10 : 2; // statistics per column: s1, s2 (+ the eight rows of UG; EPI 7 leaves s2 to its tail)
        float *red = BF3 ? reinterpret_cast<float *>(&As3[0][0][0][0]) : &As[0][0][0]; // [NS][WM][BN]
        static_assert(NS * WM * BN <= 2 * FG_BK * FG_LDA, "reduction scratch exceeds the A buffers");
        static_assert(!BF3 || NS * WM * BN <= 2 * NPC * 2 * PLA, "reduction scratch exceeds the split A buffers");
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int c = (wn * NT + j) * 32 + l31;
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const float v = q == 0 ? s1[j] : q == 1 ? s2[j] : ugs[NEPI ? q - 2 : 0][j];
                const float t = v + __shfl_xor(v, 32);
                if (lane < 32) red[(q * WM + wm) * BN + c] = t;
            }
        }
        __syncthreads();
        for (int e = tid; e < NS * BN; e += 256) {
            const int which = e / BN, c = e % BN;
            float t = 0.0f;
#pragma unroll
            for (int i = 0; i < WM; i++) t += red[(which * WM + i) * BN + c];
            if (EPI == 7 && which == 1) continue; // s2 comes out of the tail
            if (which < 2) unsafeAtomicAdd(&A.stats[which * cout + n0 + c], (double)t);
            else unsafeAtomicAdd(&A.ug[(size_t)(which - 2) * cout + n0 + c], (double)t);
        }
    }
    if (EPI == 7) {
        // the coefficient tail of a narrow layer whose s2 was never accumulated: the last workgroup derives it per column from the
        // completed UG and s1 (z_prev = u W0 + b0 is linear in u), leaves it in stats[cout + col] for whoever reads the sums, then
        // computes the coefficient vector as coef_tail does.  Same hand-off as coef_tail (device-scope atomics only, vmcnt drained).
        const CoefTail &tl = A.tail;
        __shared__ unsigned s_last7;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            s_last7 = (__hip_atomic_fetch_add(tl.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y * (SK ? gridDim.z : 1u) - 1) ? 1u : 0u;
        __syncthreads();
        if (!s_last7) return;
        const double invn = 1.0 / (double)tl.rows;
        for (int col = threadIdx.x; col < cout; col += blockDim.x) {
            const double q1 = __hip_atomic_load(&A.stats[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double zs = (A.b0 ? (double)A.b0[col] : 0.0) * q1;
            for (int d = 0; d < A.k0; d++)
                zs += (double)A.w0[(size_t)d * cout + col] * __hip_atomic_load(&A.ug[(size_t)d * cout + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float inv = 1.0f / sqrtf(A.e_var[col] + A.e_eps);
            const double q2 = (zs - (double)A.e_mean[col] * q1) * (double)inv;
            A.stats[cout + col] = q2;
            const float m1 = (float)(q1 * invn), m2 = (float)(q2 * invn);
            const float ca = tl.gamma[col] * inv;
            const float cc = -ca * inv * m2;
            tl.coef[col] = ca;
            tl.coef[cout + col] = -ca * m1 - cc * A.e_mean[col];
            tl.coef[2 * cout + col] = cc;
            tl.coef[3 * cout + col] = A.e_scale[col];
            tl.coef[4 * cout + col] = A.e_shift[col];
            if (tl.dgamma) tl.dgamma[col] += (float)q2;
            if (tl.dbeta) tl.dbeta[col] += (float)q1;
        }
        if (threadIdx.x == 0) *tl.ticket = 0u;
        return;
    }
    if (REDUCE_BELOW) coef_tail(A.tail, gridDim.x * gridDim.y * (SK ? gridDim.z : 1u), cout, A.stats, A.e_scale, A.e_shift, A.e_mean, A.e_var, A.e_eps);
}

// ---- BF3: weights pre-split into the kernel's LDS order -----------------------------------------------------------------------
// image of a (cin x cout) matrix, cin % 16 == 0: [slab = k/16][piece hi, mid, lo][k-half][column][8 bf16 = 4 dwords], 6 bytes per
// weight.  One workgroup column per segment of `table` (4 longs each: source address, image address, cin, cout).
__device__ __forceinline__ void split_weights_body(const float *__restrict__ w, unsigned *__restrict__ out, int cin, int cout, int first,
                                                   int stride)
{
    const int items = (cin / 8) * cout; // one k-octet of one column each
    for (int it = first; it < items; it += stride) {
        const int o = it / cout, c = it - o * cout;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = w[(size_t)(o * 8 + i) * cout + c];
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; i++) split3(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
        const int sl = o >> 1, kh = o & 1;
        uint4 *dst = reinterpret_cast<uint4 *>(out) + ((size_t)(sl * 3) * 2 + kh) * cout + c;
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[(size_t)2 * cout] = make_uint4(m[0], m[1], m[2], m[3]);
        dst[(size_t)4 * cout] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}
// H2 image: [slab = k/16][piece hi, lo][k-half][column][8 fp16 = 4 dwords], 4 bytes per weight (mlp_types.h: split2)
__device__ __forceinline__ void split_weights_body_h2(const float *__restrict__ w, unsigned *__restrict__ out, int cin, int cout, int first,
                                                      int stride)
{
    const int items = (cin / 8) * cout;
    for (int it = first; it < items; it += stride) {
        const int o = it / cout, c = it - o * cout;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = w[(size_t)(o * 8 + i) * cout + c] * (H2_SCALE ? 256.0f : 1.0f); // 2^8: see kH2A / kH2Un in the kernel
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; i++) split2(v[2 * i], v[2 * i + 1], h[i], l[i]);
        const int sl = o >> 1, kh = o & 1;
        uint4 *dst = reinterpret_cast<uint4 *>(out) + ((size_t)(sl * 2) * 2 + kh) * cout + c;
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[(size_t)2 * cout] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}
__global__ __launch_bounds__(256) void split_weights_kernel(const long *__restrict__ table, int pieces)
{
    const long *e = table + (size_t)blockIdx.x * 4;
    if (pieces == 2)
        split_weights_body_h2(reinterpret_cast<const float *>(e[0]), reinterpret_cast<unsigned *>(e[1]), (int)e[2], (int)e[3],
                              blockIdx.y * 256 + threadIdx.x, gridDim.y * 256);
    else
        split_weights_body(reinterpret_cast<const float *>(e[0]), reinterpret_cast<unsigned *>(e[1]), (int)e[2], (int)e[3],
                           blockIdx.y * 256 + threadIdx.x, gridDim.y * 256);
}
__global__ __launch_bounds__(256) void split_weights_one_kernel(const float *__restrict__ w, unsigned *__restrict__ out, int cin, int cout)
{
    split_weights_body(w, out, cin, cout, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// which matrices have an image: keyed by the address the GEMM entry points receive as `w` (the caller keeps the image current:
// votenet_split_weights after every change of the weights -- pointnet2.ParamStore does, once per step)
int g_fast_h2 = 1;   // 0: two-piece (fp16 x 2) images are ignored -- those GEMMs then run on the fp32 MFMA kernel (votenet_debug_fast_h2)
struct W3Entry {
    int cin, cout;
    const unsigned *w3;
    int np; // pieces: 3 (bf16 x 3) or 2 (fp16 x 2)
    const float *ascale, *unscale; // np == 2, a matrix made on the fly: FastArgs::h2_ascale / h2_unscale (else NULL)
};
static std::mutex g_w3_mu;
static std::unordered_map<const void *, W3Entry> g_w3;
// -> the image of w and its piece count in `a`; h2_ok: the caller has a two-piece instantiation (an fp16 x 2 image is useless to the
// others: they run on w itself, the fp32 MFMA kernel -- never a three-piece kernel on a two-piece image)
template <class Args> static void w3_lookup(Args &a, const float *w, int cin, int cout, bool h2_ok)
{
    std::lock_guard<std::mutex> lk(g_w3_mu);
    auto it = g_w3.find(w);
    a.w3 = nullptr;
    a.w3_np = 3;
    a.h2_ascale = a.h2_unscale = nullptr;
    if (it == g_w3.end() || it->second.cin != cin || it->second.cout != cout) return;
    if (it->second.np == 2 && !(h2_ok && g_fast_h2)) return;
    a.w3 = it->second.w3;
    a.w3_np = it->second.np;
    a.h2_ascale = it->second.ascale;
    a.h2_unscale = it->second.unscale;
}

int g_fast_dyn_lds = 0; // probe (votenet_debug_fast_dyn_lds): unused dynamic LDS per BF3 workgroup, to lower the occupancy
int g_fast_bf3 = 63; // 1: the (SRC, EPI) pairs bf3_built() lists run on bf16 x 3 split operands (votenet_debug_fast_bf3)
template <int SRC, int EPI> constexpr bool bf3_built() { return true; }
// g_fast_bf3 is a mask over GEMM families (votenet_debug_fast_bf3): bit 0 forward with statistics / pooling (EPI 0, 2), 1 plain
// forward-type (EPI 1 from x / narrow / assembled), 2 BatchNorm-backward dgrad (SRC 1, 2 with EPI 1), 3 dgrad reducing the layer
// below (EPI 3), 4 the same over an assembled layer (EPI 6), 5 over a narrow layer (EPI 4)
template <int SRC, int EPI> constexpr int bf3_family()
{
    return (EPI == 0 || EPI == 2 || EPI == 8) ? 0 : EPI == 3 ? 3 : EPI == 6 ? 4 : (EPI == 4 || EPI == 7) ? 5 : (SRC == 1 || SRC == 2 || SRC == 5) ? 2 : 1;
}
// the families with a two-piece (fp16 x 2) instantiation: forward-type operands only (activations / coordinates x weights)
template <int SRC, int EPI> constexpr bool h2_built() { return (SRC == 0 || SRC == 3 || SRC == 4) && (EPI == 0 || EPI == 1 || EPI == 2 || EPI == 8); }
template <int SRC, int EPI> constexpr bool sk_built() { return (SRC == 0 || SRC == 1) && (EPI == 0 || EPI == 1 || EPI == 3); }
#define FAST_LAUNCH(WM_, WN_, MT_, NT_, SRC_, EPI_, GRID_, ST_, A_)                                                                  \
    do {                                                                                                                             \
        if constexpr (sk_built<SRC_, EPI_>()) {                                                                                      \
            if ((GRID_).z > 1) { /* split-K (sk_take): its own instantiations, so that the others keep their registers */            \
                if (((g_fast_bf3 >> bf3_family<SRC_, EPI_>()) & 1) && (A_).w3 != nullptr && (A_).w3_np == 3 && (A_).cin % (FG_BK * BF3_SETS) == 0) \
                    hipLaunchKernelGGL((mlp_linear_fast_kernel<WM_, WN_, MT_, NT_, SRC_, EPI_, true, true>), GRID_, dim3(256), g_fast_dyn_lds, ST_, A_); \
                else                                                                                                                 \
                    hipLaunchKernelGGL((mlp_linear_fast_kernel<WM_, WN_, MT_, NT_, SRC_, EPI_, false, true>), GRID_, dim3(256), 0, ST_, A_); \
                break;                                                                                                               \
            }                                                                                                                        \
        }                                                                                                                            \
        if constexpr (h2_built<SRC_, EPI_>()) {                                                                                      \
            if (((g_fast_bf3 >> bf3_family<SRC_, EPI_>()) & 1) && (A_).w3 != nullptr && (A_).w3_np == 2 && (A_).cin % (FG_BK * H2_SETS) == 0) { \
                hipLaunchKernelGGL((mlp_linear_fast_kernel<WM_, WN_, MT_, NT_, SRC_, EPI_, true, false, true>), GRID_, dim3(256), g_fast_dyn_lds, ST_, A_); \
                break;                                                                                                               \
            }                                                                                                                        \
        }                                                                                                                            \
        if constexpr (bf3_built<SRC_, EPI_>()) {                                                                                     \
            if (((g_fast_bf3 >> bf3_family<SRC_, EPI_>()) & 1) && (A_).w3 != nullptr && (A_).w3_np == 3 && (A_).cin % (FG_BK * BF3_SETS) == 0) { \
                hipLaunchKernelGGL((mlp_linear_fast_kernel<WM_, WN_, MT_, NT_, SRC_, EPI_, true>), GRID_, dim3(256), g_fast_dyn_lds, ST_, A_);    \
                break;                                                                                                               \
            }                                                                                                                        \
        }                                                                                                                            \
        hipLaunchKernelGGL((mlp_linear_fast_kernel<WM_, WN_, MT_, NT_, SRC_, EPI_, false>), GRID_, dim3(256), 0, ST_, A_);           \
    } while (0)
int g_fast_xcd_chunk = 1; // votenet_debug_fast_xcd_chunk: the piece-layout GEMMs that gather P take their row tiles in per-XCD chunks
// persistent workgroups per launch (votenet_debug_fast_workgroups: tuning hook).  Round 6: 512 / 1024 = what is RESIDENT at two workgroups per
// CU -- every workgroup starts at once and walks 2-5 row tiles, so the prologue (tables, first loads) is paid once per resident slot and no
// second round of workgroups is dispatched behind the first (1024 / 2048, the round-3 optimum of the six-MFMA kernels: 3.446 -> 3.390 ms per step)
int g_fast_cap22 = 512, g_fast_cap41 = 1024;

// ---- split-K for launches of few row tiles (FastArgs::sk_ws) ----------------------------------------------------------------------
// The caller arms the NEXT launch of its thread with a workspace (votenet_mlp_split_k_arm; size from votenet_mlp_split_k_floats) and
// registers one persistent zeroed ticket array per process (votenet_mlp_split_k_tickets: tiles of consecutive launches take
// consecutive slots of it as a ring; the last arriver of a tile puts its slot back to zero).  Unarmed launches never split.
static thread_local float *t_sk_ws = nullptr;
static thread_local size_t t_sk_floats = 0;
static std::mutex g_sk_mu;
static unsigned *g_sk_tickets = nullptr;
static long g_sk_nticket = 0, g_sk_tpos = 0;
int g_sk_target = 640, g_sk_max_parts = 4, g_sk_min_slabs = 4, g_sk_max_wgs = 400; // votenet_debug_split_k (tuning hook)
struct SkPlan {
    int parts, bn; // parts (1: no split), column-block width of the variant the launch takes
    long tiles;    // output tiles = row tiles x column blocks
};
// the variant fast_dispatch picks for (rows, cin, cout) and the parts it is split into; pooled / piece-layout / gather launches never split
static SkPlan sk_plan(long rows, int cin, int cout, bool pooled)
{
    SkPlan p = {1, 0, 0};
    if (pooled || rows <= 0 || rows % FG_BM != 0 || cin % (2 * FG_BK) != 0 || cin > 512 || cout % 64 != 0) return p;
    const long ntiles = rows / FG_BM;
    p.bn = (cout % 128 == 0 && ntiles * (cout / 128) >= 200) ? 128 : 64;
    p.tiles = ntiles * (cout / p.bn);
    if (p.tiles >= g_sk_max_wgs) return p;
    const int nk = cin / FG_BK;
    int parts = (int)((g_sk_target + p.tiles / 2) / p.tiles);
    parts = parts > g_sk_max_parts ? g_sk_max_parts : parts;
    while (parts > 1 && (nk % (2 * parts) != 0 || nk / parts < g_sk_min_slabs)) --parts;
    p.parts = parts < 1 ? 1 : parts;
    return p;
}
// -> parts for this launch (and the workspace / tickets in `a`), consuming the thread's armed workspace
template <int SRC, int EPI>
static int sk_take(FastArgs &a, long gx, long ntiles, int bn)
{
    float *ws = t_sk_ws;
    const size_t have = t_sk_floats;
    t_sk_ws = nullptr;
    t_sk_floats = 0;
    if (!sk_built<SRC, EPI>()) return 1;
    if (ws == nullptr || gx != ntiles || a.nh_dev != nullptr || a.wh != nullptr || a.xcd_chunk) return 1;
    const SkPlan p = sk_plan(a.rows, a.cin, a.cout, false);
    if (p.parts <= 1 || p.bn != bn || (size_t)p.tiles * p.parts * FG_BM * bn > have) return 1;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    if (g_sk_tickets == nullptr || p.tiles > g_sk_nticket) return 1;
    if (g_sk_tpos + p.tiles > g_sk_nticket) g_sk_tpos = 0;
    a.sk_ticket = g_sk_tickets + g_sk_tpos;
    g_sk_tpos += p.tiles;
    a.sk_ws = ws;
    return p.parts;
}

template <int SRC, int EPI>
static bool fast_dispatch(const FastArgs &a_in, hipStream_t st)
{
    FastArgs a = a_in;
    if (bf3_built<SRC, EPI>() && g_fast_bf3) w3_lookup(a, a.w, a.cin, a.cout, h2_built<SRC, EPI>());
    const float *abase = (SRC == 0) ? a.x : (SRC == 3) ? a.u8 : (SRC == 4) ? a.ptab : a.zsrc;
    if (SRC == 4 && ((uintptr_t)a.geo % 16 != 0 || (uintptr_t)a.wx % 16 != 0)) return false;
    if ((SRC == 3 && a.cin > 128) || ((SRC == 3 || EPI == 4 || EPI == 7) && (a.k0 < 1 || a.k0 > 8 || (uintptr_t)a.u8 % 16 != 0))) return false;
    if (EPI == 7 && (a.cout % 64 != 0 || a.mask_in == nullptr || a.tail.ticket == nullptr || (uintptr_t)a.mask_in % 8 != 0)) return false;
    const bool aligned = ((uintptr_t)abase % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.z % 16 == 0) &&
                         ((SRC != 1 && SRC != 5) || (uintptr_t)a.da % 16 == 0) && (SRC != 5 || a.wh != nullptr) &&
                         (SRC != 2 || ((uintptr_t)a.gout % 16 == 0 && (uintptr_t)a.argmax % 16 == 0));
    if (!aligned || a.cin % (2 * FG_BK) != 0 || a.cin > 512 || a.rows % FG_BM != 0 || a.rows == 0) return false;
    const long ntiles = a.rows / FG_BM;
    long gx;
    if constexpr (EPI == 4 || EPI == 7) { // 128 x 64 tiles only: the epilogue's per-column constants and the u rows fit the registers of that shape
        if (a.cout % 64 != 0) return false;
        const int ny = a.cout / 64;
        gx = ntiles < g_fast_cap41 / ny ? ntiles : g_fast_cap41 / ny;
        FAST_LAUNCH(4, 1, 1, 2, SRC, EPI, dim3((unsigned)gx, ny), st, a);
        return true;
    } else {
    // few row tiles (FP layers, voting, mlp2): 128 x 64 tiles double the number of workgroups
    if (EPI != 2 && a.cout % 128 == 0 && ntiles * (a.cout / 128) < 200) {
        const int ny = a.cout / 64;
        gx = ntiles;
        if (SRC == 2 && (gx * FG_BM) % a.pool_k != 0) return false;
        const dim3 grid((unsigned)gx, ny, sk_take<SRC, EPI>(a, gx, ntiles, 64));
        FAST_LAUNCH(4, 1, 1, 2, SRC, EPI, grid, st, a);
        return true;
    }
    if (a.cout % 128 == 0) {
        const int ny = a.cout / 128;
        gx = ntiles < g_fast_cap22 / ny ? ntiles : g_fast_cap22 / ny;
        if (a.xcd_chunk && gx >= 64) gx &= ~7L; // a whole number of workgroups per XCD
        if (SRC == 2 && (gx * FG_BM) % a.pool_k != 0) return false; // a tile jump must be a whole number of groups
        const dim3 grid((unsigned)gx, ny, EPI == 2 ? 1 : sk_take<SRC, EPI>(a, gx, ntiles, 128));
        FAST_LAUNCH(2, 2, 2, 2, SRC, EPI, grid, st, a);
        return true;
    }
    if (EPI != 2 && a.cout % 64 == 0) { // 64, and the odd multiples of 64 (320 = voting's 259 padded): 128 x 64 tiles, cout / 64 column blocks
        const int ny = a.cout / 64;
        gx = ntiles < g_fast_cap41 / ny ? ntiles : g_fast_cap41 / ny;
        if (SRC == 2 && (gx * FG_BM) % a.pool_k != 0) return false;
        const dim3 grid((unsigned)gx, ny, sk_take<SRC, EPI>(a, gx, ntiles, 64));
        FAST_LAUNCH(4, 1, 1, 2, SRC, EPI, grid, st, a);
        return true;
    }
    }
    return false;
}

// returns true when the fast kernel took the launch
bool mlp_linear_fast_launch(const float *x, const float *in_scale, const float *in_shift, const BnRaw &in_raw, int in_relu,
                            long rows, int cin, int cout, const float *w, const float *bias, float *z, double *stats, hipStream_t st)
{
    FastArgs a = {};
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_raw = in_raw;
    a.in_relu = in_relu;
    a.rows = rows;
    a.cin = cin;
    a.cout = cout;
    a.w = w;
    a.bias = bias;
    a.z = z;
    a.stats = stats;
    return stats ? fast_dispatch<0, 0>(a, st) : fast_dispatch<0, 1>(a, st); // no statistics wanted: skip their arithmetic
}

// z = act(x) w + bias over the first 16 * nh_dev[0] of `rows` rows (the piece layout with the count on the device: the dense part of a
// pooled layer's Gram-form input gradient); no statistics
extern "C" int votenet_mlp_linear_half(const float *x, const float *in_scale, const float *in_shift, int in_relu, long rows, int cin,
                                       int cout, const float *w, const float *bias, float *z, const int *nh_dev, void *stream)
{
    VN_REQUIRE(rows > 0 && cin > 0 && cout > 0 && x && w && z && nh_dev, "mlp_linear_half: bad arguments");
    VN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "mlp_linear_half: in_scale and in_shift go together");
    FastArgs a = {};
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_relu = in_relu;
    a.rows = rows;
    a.cin = cin;
    a.cout = cout;
    a.w = w;
    a.bias = bias;
    a.z = z;
    a.nh_dev = nh_dev;
    if (!fast_dispatch<0, 1>(a, as_stream(stream)))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "mlp_linear_half: shape not served (rows %% 128 == 0, cin %% 32 == 0, cout %% 64 == 0, 16-byte aligned)");
    return check_launch("mlp_linear_half");
}

// forward layer + raw max / min pooling over groups of 64 rows (EPI 2).  Returns false when the shape is not served.
bool mlp_linear_pool_launch(const float *x, const float *in_scale, const float *in_shift, const BnRaw &in_raw, int in_relu,
                            long rows, int cin, int cout, const float *w, const float *bias, float *z, double *stats, float *zmax,
                            float *zmin, int *amax, int *amin, hipStream_t st, const float *wh, const float *pool_gamma, const int *nh_dev)
{
    FastArgs a = {};
    a.nh_dev = nh_dev;
    a.wh = wh;                       // piece layout (half.hip): weighted statistics ...
    a.pool32 = wh != nullptr ? 1 : 0; // ... and the pool's candidate (max or min by the sign of gamma) per 16-row piece
    a.pool_gamma = pool_gamma;
    a.x = x;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_raw = in_raw;
    a.in_relu = in_relu;
    a.rows = rows;
    a.cin = cin;
    a.cout = cout;
    a.w = w;
    a.bias = bias;
    a.z = z;
    a.stats = stats;
    a.zmax = zmax;
    a.zmin = zmin;
    a.amax = amax;
    a.amin = amin;
    if (g_fast_bf3) w3_lookup(a, w, cin, cout, true);
    const bool aligned = ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)z % 16 == 0);
    if (!aligned || cin % (2 * FG_BK) != 0 || cin > 512 || rows % FG_BM != 0 || rows == 0 || cout % 128 != 0) return false;
    const long ntiles = rows / FG_BM;
    const int ny = cout / 128;
    const long gx = ntiles < g_fast_cap22 / ny ? ntiles : g_fast_cap22 / ny;
    if (a.pool32) FAST_LAUNCH(2, 2, 2, 2, 0, 8, dim3((unsigned)gx, ny), st, a);
    else FAST_LAUNCH(2, 2, 2, 2, 0, 2, dim3((unsigned)gx, ny), st, a);
    return true;
}

} // namespace votenet

using namespace votenet;

// da_prev (rows x cout) = dz (rows x c) * wT (c x cout) with dz = BatchNorm-backward(da | pooled gout, zsrc, coef)
// formed inside the A-operand loader.
extern "C" int votenet_mlp_dgrad_bn(long rows, int c, int cout, const float *da, const float *gout, const int *argmax,
                                    int pool_k, const float *zsrc, const float *coef, int relu, const float *wT,
                                    float *da_prev, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0 && cout > 0, "mlp_dgrad_bn expects rows > 0, c > 0, cout > 0");
    VN_REQUIRE((da != nullptr) != (gout != nullptr), "mlp_dgrad_bn: exactly one of da / gout");
    VN_REQUIRE(zsrc && coef && wT && da_prev, "mlp_dgrad_bn: null buffer");
    VN_REQUIRE(gout == nullptr || (argmax != nullptr && pool_k > 0 && rows % pool_k == 0), "mlp_dgrad_bn: pooled source needs argmax and k");
    FastArgs a = {};
    a.da = da;
    a.gout = gout;
    a.argmax = argmax;
    a.pool_k = pool_k;
    a.zsrc = zsrc;
    a.coef = coef;
    a.src_relu = relu;
    a.rows = rows;
    a.cin = c;
    a.cout = cout;
    a.w = wT;
    a.z = da_prev;
    hipStream_t st = as_stream(stream);
    const bool ok = da ? fast_dispatch<1, 1>(a, st) : fast_dispatch<2, 1>(a, st);
    if (!ok) return set_error(VOTENET_E_INVALID_ARGUMENT, "mlp_dgrad_bn: shape not supported by the fused kernel (use votenet_bn_backward_apply + votenet_mlp_linear)");
    return check_launch("mlp_dgrad_bn");
}

// votenet_mlp_dgrad_bn (dense upstream gradient) on the piece layout (half.hip): da / da_prev are TOTAL gradients per compact row, the
// affine part B + C z of the rebuilt dz is scaled by wh on the rows that stand for dropped pieces (SRC 5); rows: the caller's upper
// bound when nh_dev (the device's piece count) is given.  A plain GEMM: nothing but the store in its epilogue.
extern "C" int votenet_mlp_dgrad_bn_half(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                         const float *wT, float *da_prev, const float *wh, const int *nh_dev, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0 && cout > 0, "mlp_dgrad_bn_half expects rows > 0, c > 0, cout > 0");
    VN_REQUIRE(da && zsrc && coef && wT && da_prev && wh, "mlp_dgrad_bn_half: null buffer");
    FastArgs a = {};
    a.da = da;
    a.zsrc = zsrc;
    a.coef = coef;
    a.src_relu = relu;
    a.rows = rows;
    a.cin = c;
    a.cout = cout;
    a.w = wT;
    a.z = da_prev;
    a.wh = wh;
    a.nh_dev = nh_dev;
    if (!fast_dispatch<5, 1>(a, as_stream(stream)))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "mlp_dgrad_bn_half: shape not served (rows %% 128 == 0, c %% 32 == 0, cout %% 64 == 0, 16-byte aligned buffers)");
    return check_launch("mlp_dgrad_bn_half");
}

// The same GEMM whose epilogue also reduces the BatchNorm backward of the layer BELOW (the one whose activation da_prev is the
// gradient of): sums[0:cout] += sum da_prev', sums[cout:2cout] += sum da_prev' * zhat_prev -- votenet_bn_backward_reduce(rows,
// cout, 0, da_prev, NULL, z_prev, ...) without the pass over da_prev and z_prev of its own.  sums (2*cout doubles) is zeroed by
// the caller.  Dense upstream gradient only (da).
extern "C" int votenet_mlp_dgrad_bn_reduce(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                           const float *wT, float *da_prev, const float *z_prev, const float *scale_prev,
                                           const float *shift_prev, const float *mean_prev, const float *var_prev, float eps,
                                           int relu_prev, double *sums, const votenet_coef_tail *tail, void *stream)
{
    VN_REQUIRE(!tail || (tail->ticket && tail->gamma && tail->coef && tail->rows > 0), "mlp_dgrad_bn_reduce: incomplete coefficient tail");
    VN_REQUIRE(rows > 0 && c > 0 && cout > 0, "mlp_dgrad_bn_reduce expects rows > 0, c > 0, cout > 0");
    VN_REQUIRE(da && zsrc && coef && wT && da_prev, "mlp_dgrad_bn_reduce: null buffer");
    VN_REQUIRE(z_prev && scale_prev && shift_prev && mean_prev && var_prev && sums, "mlp_dgrad_bn_reduce: null buffer of the layer below");
    FastArgs a = {};
    a.da = da;
    a.zsrc = zsrc;
    a.coef = coef;
    a.src_relu = relu;
    a.rows = rows;
    a.cin = c;
    a.cout = cout;
    a.w = wT;
    a.z = da_prev;
    a.ez = z_prev;
    a.e_scale = scale_prev;
    a.e_shift = shift_prev;
    a.e_mean = mean_prev;
    a.e_var = var_prev;
    a.e_eps = eps;
    a.e_relu = relu_prev;
    a.stats = sums;
    a.tail = to_tail(tail);
    if (!fast_dispatch<1, 3>(a, as_stream(stream)))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "mlp_dgrad_bn_reduce: shape not supported by the fused kernel (use votenet_mlp_dgrad_bn + votenet_bn_backward_reduce)");
    return check_launch("mlp_dgrad_bn_reduce");
}

// Second layer of an SA chain whose first layer is NARROW (narrow.hip): z (rows x cout) = relu(bn0(z0)) w + bias with
// z0[r,:] = narrow_z(u8[r], W0, b0) rebuilt in the operand loader (z0 is never stored), bn0 from raw statistics (in_bn) or from
// in_scale / in_shift.  stats as for votenet_mlp_linear.
static int narrow_linear_impl(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0, const float *in_scale,
                              const float *in_shift, const votenet_bn_raw *in_bn, int in_relu, const float *w, const float *bias, float *z,
                              double *stats, const float *wh, void *stream, unsigned short *mask_out = nullptr)
{
    VN_REQUIRE(rows > 0 && k0 >= 3 && k0 <= 8 && c0 > 0 && cout > 0, "narrow_linear expects rows > 0, 3 <= k0 <= 8, c0 > 0, cout > 0");
    VN_REQUIRE(u8 && w0 && w && z, "narrow_linear: null buffer");
    VN_REQUIRE(in_bn != nullptr || (in_scale != nullptr && in_shift != nullptr), "narrow_linear: the first layer's BatchNorm is missing");
    FastArgs a = {};
    a.u8 = u8;
    a.w0 = w0;
    a.b0 = b0;
    a.k0 = k0;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_raw = to_raw(in_bn);
    a.in_relu = in_relu;
    a.rows = rows;
    a.cin = c0;
    a.cout = cout;
    a.w = w;
    a.bias = bias;
    a.z = z;
    a.stats = stats;
    a.wh = wh; // piece layout (half.hip): the statistics weigh row 0 of every piece
    a.mask_out = mask_out;
    VN_REQUIRE(mask_out == nullptr || (uintptr_t)mask_out % 8 == 0, "narrow_linear: the mask must be 8-byte aligned");
    hipStream_t st = as_stream(stream);
    const bool ok = stats ? fast_dispatch<3, 0>(a, st) : fast_dispatch<3, 1>(a, st);
    if (!ok) return set_error(VOTENET_E_INVALID_ARGUMENT, "narrow_linear: shape not served (rows %% 128 == 0, c0 %% 32 == 0, c0 <= 128, cout == 64 or cout %% 128 == 0, 16-byte aligned buffers)");
    return check_launch("narrow_linear");
}
extern "C" int votenet_narrow_linear(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                                     const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu,
                                     const float *w, const float *bias, float *z, double *stats, void *stream)
{
    return narrow_linear_impl(rows, k0, c0, cout, u8, w0, b0, in_scale, in_shift, in_bn, in_relu, w, bias, z, stats, nullptr, stream);
}
extern "C" int votenet_narrow_linear_half(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                                          const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu,
                                          const float *w, const float *bias, float *z, double *stats, const float *wh, void *stream)
{
    VN_REQUIRE(wh != nullptr && rows % 128 == 0, "narrow_linear_half: null weights / rows %% 128 != 0");
    return narrow_linear_impl(rows, k0, c0, cout, u8, w0, b0, in_scale, in_shift, in_bn, in_relu, w, bias, z, stats, wh, stream);
}
// The same (wh may be NULL: the full row layout) leaving the narrow layer's ReLU mask for the backward pass: mask (rows x c0 / 16 16-bit
// words, 8-byte aligned, c0 % 64 == 0): bit k % 16 of word [row][k / 16] = [relu(bn0(z0[row, k])) > 0] -- what
// votenet_narrow_dgrad_bn_reduce_masked reads instead of rebuilding z0.
extern "C" int votenet_narrow_linear_masked(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                                            const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu,
                                            const float *w, const float *bias, float *z, double *stats, const float *wh,
                                            unsigned short *mask, void *stream)
{
    VN_REQUIRE(mask != nullptr && c0 % 64 == 0 && rows % 128 == 0, "narrow_linear_masked: null mask / c0 %% 64 != 0 / rows %% 128 != 0");
    return narrow_linear_impl(rows, k0, c0, cout, u8, w0, b0, in_scale, in_shift, in_bn, in_relu, w, bias, z, stats, wh, stream, mask);
}

// Input-gradient GEMM of that second layer: da0 = dz1 wT (dz1 from (da, zsrc, coef) as votenet_mlp_dgrad_bn) is NOT stored; its
// epilogue reduces the first layer's BatchNorm backward (sums: 2*c0 doubles, as votenet_mlp_dgrad_bn_reduce, z0 rebuilt from u8)
// and ug[d*c0 + c] += sum_r u8[r,d] da0'[r,c] (8*c0 doubles), the data term of votenet_narrow_wgrad_first.  Both pre-zeroed.
static int narrow_dgrad_bn_reduce_impl(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef, int relu,
                                       const float *wT, const float *u8, const float *w0, const float *b0, const float *scale0,
                                       const float *shift0, const float *mean0, const float *var0, float eps, int relu0, double *sums,
                                       double *ug, const votenet_coef_tail *tail, const float *wh, void *stream,
                                       const void *mask = nullptr)
{
    VN_REQUIRE(!tail || (tail->ticket && tail->gamma && tail->coef && tail->rows > 0), "narrow_dgrad_bn_reduce: incomplete coefficient tail");
    VN_REQUIRE(rows > 0 && c > 0 && c0 > 0 && k0 >= 3 && k0 <= 8, "narrow_dgrad_bn_reduce expects rows > 0, c > 0, c0 > 0, 3 <= k0 <= 8");
    VN_REQUIRE(da && zsrc && coef && wT && u8 && w0, "narrow_dgrad_bn_reduce: null buffer");
    VN_REQUIRE(scale0 && shift0 && mean0 && var0 && sums && ug, "narrow_dgrad_bn_reduce: null buffer of the first layer");
    FastArgs a = {};
    a.da = da;
    a.zsrc = zsrc;
    a.coef = coef;
    a.src_relu = relu;
    a.rows = rows;
    a.cin = c;
    a.cout = c0;
    a.w = wT;
    a.u8 = u8;
    a.w0 = w0;
    a.b0 = b0;
    a.k0 = k0;
    a.e_scale = scale0;
    a.e_shift = shift0;
    a.e_mean = mean0;
    a.e_var = var0;
    a.e_eps = eps;
    a.e_relu = relu0;
    a.stats = sums;
    a.ug = ug;
    a.tail = to_tail(tail);
    a.wh = wh; // piece layout: da holds totals, the affine part of the rebuilt dz1 counts wh[q] times on row 16 q (SRC 5)
    if (mask != nullptr) { // EPI 7: the mask the forward pass recorded instead of the rebuild of z0; s2 from the tail
        VN_REQUIRE(tail != nullptr && wh != nullptr && c0 % 64 == 0 && (uintptr_t)mask % 8 == 0,
                   "narrow_dgrad_bn_reduce_masked needs the coefficient tail, the piece layout's weights, c0 %% 64 == 0 and an 8-byte aligned mask");
        a.mask_in = static_cast<const unsigned long long *>(mask);
        if (!fast_dispatch<5, 7>(a, as_stream(stream)))
            return set_error(VOTENET_E_INVALID_ARGUMENT, "narrow_dgrad_bn_reduce_masked: shape not served (as votenet_narrow_dgrad_bn_reduce)");
        return check_launch("narrow_dgrad_bn_reduce_masked");
    }
    if (!(wh ? fast_dispatch<5, 4>(a, as_stream(stream)) : fast_dispatch<1, 4>(a, as_stream(stream))))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "narrow_dgrad_bn_reduce: shape not served (rows %% 128 == 0, c %% 32 == 0, c <= 512, c0 == 64 or c0 %% 128 == 0, 16-byte aligned buffers)");
    return check_launch("narrow_dgrad_bn_reduce");
}
extern "C" int votenet_narrow_dgrad_bn_reduce(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef,
                                              int relu, const float *wT, const float *u8, const float *w0, const float *b0,
                                              const float *scale0, const float *shift0, const float *mean0, const float *var0,
                                              float eps, int relu0, double *sums, double *ug, const votenet_coef_tail *tail, void *stream)
{
    return narrow_dgrad_bn_reduce_impl(rows, c, c0, k0, da, zsrc, coef, relu, wT, u8, w0, b0, scale0, shift0, mean0, var0, eps, relu0, sums, ug,
                                       tail, nullptr, stream);
}
extern "C" int votenet_narrow_dgrad_bn_reduce_half(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef,
                                                   int relu, const float *wT, const float *u8, const float *w0, const float *b0,
                                                   const float *scale0, const float *shift0, const float *mean0, const float *var0,
                                                   float eps, int relu0, double *sums, double *ug, const votenet_coef_tail *tail,
                                                   const float *wh, void *stream)
{
    VN_REQUIRE(wh != nullptr, "narrow_dgrad_bn_reduce_half: null weights");
    return narrow_dgrad_bn_reduce_impl(rows, c, c0, k0, da, zsrc, coef, relu, wT, u8, w0, b0, scale0, shift0, mean0, var0, eps, relu0, sums, ug,
                                       tail, wh, stream);
}

// votenet_narrow_dgrad_bn_reduce_half with the narrow layer's ReLU mask from the forward pass (votenet_narrow_linear_masked) instead of the
// rebuild of z0 in the epilogue, and sums[c0:2 c0] derived in the coefficient tail (required) from ug and sums[0:c0]: same sums, same ug,
// same coefficient vector up to the association of the sums; a third of the epilogue's arithmetic, no register spills.
extern "C" int votenet_narrow_dgrad_bn_reduce_masked(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef,
                                                     int relu, const float *wT, const float *u8, const float *w0, const float *b0,
                                                     const float *scale0, const float *shift0, const float *mean0, const float *var0,
                                                     float eps, int relu0, double *sums, double *ug, const votenet_coef_tail *tail,
                                                     const float *wh, const void *mask, void *stream)
{
    VN_REQUIRE(mask != nullptr, "narrow_dgrad_bn_reduce_masked: null mask");
    return narrow_dgrad_bn_reduce_impl(rows, c, c0, k0, da, zsrc, coef, relu, wT, u8, w0, b0, scale0, shift0, mean0, var0, eps, relu0, sums, ug,
                                       tail, wh, stream, mask);
}

// ---- split-K (see sk_plan / FastArgs::sk_ws) -----------------------------------------------------------------------------------
// floats of workspace a launch of votenet_mlp_linear / votenet_mlp_dgrad_bn / votenet_mlp_dgrad_bn_reduce with these sizes would use
// split (0: it would not split: nothing to arm)
extern "C" long votenet_mlp_split_k_floats(long rows, int cin, int cout)
{
    {
        std::lock_guard<std::mutex> lk(votenet::g_sk_mu);
        if (votenet::g_sk_tickets == nullptr) return 0;
    }
    const votenet::SkPlan p = votenet::sk_plan(rows, cin, cout, false);
    return p.parts > 1 ? p.tiles * p.parts * votenet::FG_BM * p.bn : 0;
}
// the workspace (device memory, any contents, alive until the launch has run) for the NEXT fused-GEMM launch of the calling thread
extern "C" int votenet_mlp_split_k_arm(void *ws, long floats)
{
    VN_REQUIRE((ws == nullptr) == (floats <= 0) && (uintptr_t)ws % 16 == 0, "mlp_split_k_arm: a 16-byte aligned workspace and its size");
    votenet::t_sk_ws = static_cast<float *>(ws);
    votenet::t_sk_floats = ws ? (size_t)floats : 0;
    return VOTENET_OK;
}
// n ZEROED unsigned ints of device memory that stay alive (and untouched by anyone else) while split launches may run; NULL: split-K off
extern "C" int votenet_mlp_split_k_tickets(void *tickets, long n)
{
    VN_REQUIRE((tickets == nullptr) == (n <= 0), "mlp_split_k_tickets: an array and its length, or NULL and 0");
    std::lock_guard<std::mutex> lk(votenet::g_sk_mu);
    votenet::g_sk_tickets = static_cast<unsigned *>(tickets);
    votenet::g_sk_nticket = tickets ? n : 0;
    votenet::g_sk_tpos = 0;
    return VOTENET_OK;
}
extern "C" void votenet_debug_split_k(int target_wgs, int max_parts, int min_slabs, int max_wgs) // tuning hook: 0 keeps a value
{
    VN_DEBUG_GATE();
    if (target_wgs > 0) votenet::g_sk_target = target_wgs;
    if (max_parts > 0) votenet::g_sk_max_parts = max_parts;
    if (min_slabs > 0) votenet::g_sk_min_slabs = min_slabs;
    if (max_wgs > 0) votenet::g_sk_max_wgs = max_wgs;
}

extern "C" void votenet_debug_fast_dyn_lds(int bytes) { VN_DEBUG_GATE(); votenet::g_fast_dyn_lds = bytes; }
extern "C" void votenet_debug_fast_xcd_chunk(int on) { VN_DEBUG_GATE(); votenet::g_fast_xcd_chunk = on ? 1 : 0; }
extern "C" void votenet_debug_fast_bf3(int on) { VN_DEBUG_GATE(); votenet::g_fast_bf3 = (on == 1) ? 63 : on; } // 0 off, 1 every family, else a mask

// BF3 weight images.  table (device, 4 longs per segment): source address (cin x cout floats, row-major), image address
// (cin * cout * 6 bytes, 16-byte aligned), cin (% 16 == 0), cout.  One launch for all segments.
extern "C" int votenet_split_weights(int nseg, const long *table, void *stream)
{
    VN_REQUIRE(nseg >= 0, "split_weights expects nseg >= 0");
    if (nseg == 0) return VOTENET_OK;
    VN_REQUIRE(table != nullptr, "split_weights: null table");
    hipLaunchKernelGGL(votenet::split_weights_kernel, dim3(nseg, 8), dim3(256), 0, as_stream(stream), table, 3);
    return check_launch("split_weights");
}
// The same table as TWO fp16 pieces per weight (image address: cin * cout * 4 bytes; mlp_types.h: split2) -- for matrices whose GEMMs
// multiply forward operands (register them with votenet_register_split_weights_pieces(..., 2)).
extern "C" int votenet_split_weights_h2(int nseg, const long *table, void *stream)
{
    VN_REQUIRE(nseg >= 0, "split_weights_h2 expects nseg >= 0");
    if (nseg == 0) return VOTENET_OK;
    VN_REQUIRE(table != nullptr, "split_weights_h2: null table");
    hipLaunchKernelGGL(votenet::split_weights_kernel, dim3(nseg, 8), dim3(256), 0, as_stream(stream), table, 2);
    return check_launch("split_weights_h2");
}

// The image of ONE matrix, arguments by value (a matrix made on the fly, e.g. votenet_pool_dgrad_prepare's).
extern "C" int votenet_split_weights_one(const float *w, int cin, int cout, void *image, void *stream)
{
    VN_REQUIRE(w && image && cin > 0 && cin % 16 == 0 && cout > 0 && (uintptr_t)image % 16 == 0,
               "split_weights_one expects cin % 16 == 0, cout > 0 and a 16-byte aligned image");
    const int items = (cin / 8) * cout;
    hipLaunchKernelGGL(votenet::split_weights_one_kernel, dim3((items + 255) / 256), dim3(256), 0, as_stream(stream), w,
                       static_cast<unsigned *>(image), cin, cout);
    return check_launch("split_weights_one");
}

// Tell the GEMM entry points that the matrix they receive at address `w` (cin x cout) has a current image at `w3`
// (w3 == NULL: forget it).  While registered, the fused forward / input-gradient GEMMs read the image instead of w.
extern "C" int votenet_register_split_weights_pieces(const float *w, int cin, int cout, const void *w3, int pieces)
{
    VN_REQUIRE(w != nullptr, "register_split_weights: null w");
    VN_REQUIRE(w3 == nullptr || (cin > 0 && cin % 16 == 0 && cout > 0 && (uintptr_t)w3 % 16 == 0),
               "register_split_weights expects cin % 16 == 0, cout > 0 and a 16-byte aligned image");
    VN_REQUIRE(pieces == 2 || pieces == 3, "register_split_weights: pieces must be 3 (bf16 x 3) or 2 (fp16 x 2)");
    std::lock_guard<std::mutex> lk(votenet::g_w3_mu);
    if (w3) votenet::g_w3[w] = votenet::W3Entry{cin, cout, static_cast<const unsigned *>(w3), pieces, nullptr, nullptr};
    else votenet::g_w3.erase(w);
    return VOTENET_OK;
}
// A two-piece image whose scaling is NOT the standard one (weights x 2^8): ascale (cin floats, device) scales the staged input channels,
// unscale (cout floats, device) the output columns -- votenet_pool_dgrad_prepare_h2 writes the image and both vectors.
extern "C" int votenet_register_split_weights_scaled(const float *w, int cin, int cout, const void *w3, const float *ascale, const float *unscale)
{
    VN_REQUIRE(w != nullptr && w3 != nullptr && ascale != nullptr && unscale != nullptr, "register_split_weights_scaled: null argument");
    VN_REQUIRE(cin > 0 && cin % 16 == 0 && cout > 0 && (uintptr_t)w3 % 16 == 0,
               "register_split_weights_scaled expects cin % 16 == 0, cout > 0 and a 16-byte aligned image");
    std::lock_guard<std::mutex> lk(votenet::g_w3_mu);
    votenet::g_w3[w] = votenet::W3Entry{cin, cout, static_cast<const unsigned *>(w3), 2, ascale, unscale};
    return VOTENET_OK;
}
extern "C" int votenet_register_split_weights(const float *w, int cin, int cout, const void *w3)
{
    return votenet_register_split_weights_pieces(w, cin, cout, w3, 3);
}
extern "C" void votenet_debug_fast_h2(int on)
{
    VN_DEBUG_GATE();
    votenet::g_fast_h2 = on ? 1 : 0;
}
extern "C" void votenet_debug_fast_workgroups(int cap22, int cap41) // tuning hook: 0 keeps a value
{
    VN_DEBUG_GATE();
    if (cap22 > 0) votenet::g_fast_cap22 = cap22;
    if (cap41 > 0) votenet::g_fast_cap41 = cap41;
}

// Second layer of an SA chain whose first layer is ASSEMBLED in the operand loader (assemble.hip): z (rows x cout) =
// relu(bn0(z0)) w + bias with z0[r,:] = P[prow(r),:] + dxyz(r) . wx rebuilt from geo and the per-point table P (points x c0,
// bias included); bn0 from raw statistics (in_bn: votenet_assemble_stats) or in_scale / in_shift.  stats as votenet_mlp_linear.
extern "C" int votenet_assembled_linear(long rows, int c0, int cout, const float *geo, const float *P, const float *wx,
                                        const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu,
                                        const float *w, const float *bias, float *z, double *stats, void *stream)
{
    VN_REQUIRE(rows > 0 && c0 > 0 && cout > 0, "assembled_linear expects rows > 0, c0 > 0, cout > 0");
    VN_REQUIRE(geo && P && wx && w && z, "assembled_linear: null buffer");
    VN_REQUIRE(in_bn != nullptr || (in_scale != nullptr && in_shift != nullptr), "assembled_linear: the first layer's BatchNorm is missing");
    FastArgs a = {};
    a.geo = geo;
    a.ptab = P;
    a.wx = wx;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_raw = to_raw(in_bn);
    a.in_relu = in_relu;
    a.rows = rows;
    a.cin = c0;
    a.cout = cout;
    a.w = w;
    a.bias = bias;
    a.z = z;
    a.stats = stats;
    hipStream_t st = as_stream(stream);
    const bool ok = stats ? fast_dispatch<4, 0>(a, st) : fast_dispatch<4, 1>(a, st);
    if (!ok) return set_error(VOTENET_E_INVALID_ARGUMENT, "assembled_linear: shape not served (rows %% 128 == 0, c0 %% 32 == 0, c0 <= 512, cout %% 64 == 0, 16-byte aligned buffers)");
    return check_launch("assembled_linear");
}

// votenet_assembled_linear on the piece layout (half.hip): rows = 16 x pieces, wh = the weight of every piece's row 0 in
// the BatchNorm statistics (a ball's slot 0 also stands for its dropped all-copy pieces).
extern "C" int votenet_assembled_linear_half(long rows, int c0, int cout, const float *geo, const float *P, const float *wx,
                                             const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu,
                                             const float *w, const float *bias, float *z, double *stats, const float *wh, const int *nh_dev,
                                             void *stream)
{
    VN_REQUIRE(rows > 0 && c0 > 0 && cout > 0, "assembled_linear_half expects rows > 0, c0 > 0, cout > 0");
    VN_REQUIRE(geo && P && wx && w && z && wh, "assembled_linear_half: null buffer");
    VN_REQUIRE(in_bn != nullptr || (in_scale != nullptr && in_shift != nullptr), "assembled_linear_half: the first layer's BatchNorm is missing");
    FastArgs a = {};
    a.geo = geo;
    a.ptab = P;
    a.wx = wx;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.in_raw = to_raw(in_bn);
    a.in_relu = in_relu;
    a.rows = rows;
    a.cin = c0;
    a.cout = cout;
    a.w = w;
    a.bias = bias;
    a.z = z;
    a.stats = stats;
    a.wh = wh;
    a.nh_dev = nh_dev;
    a.xcd_chunk = g_fast_xcd_chunk; // the rows are in scene order: an XCD's L2 then holds the slices of P its tiles gather
    hipStream_t st = as_stream(stream);
    const bool ok = stats ? fast_dispatch<4, 0>(a, st) : fast_dispatch<4, 1>(a, st);
    if (!ok) return set_error(VOTENET_E_INVALID_ARGUMENT, "assembled_linear_half: shape not served (as votenet_assembled_linear)");
    return check_launch("assembled_linear_half");
}

// votenet_mlp_dgrad_bn_reduce for an ASSEMBLED layer below (assemble.hip): z_prev is not read but rebuilt from geo, the per-point
// table P (points x cout, bias included; points * cout * 4 < 2^32) and wx (3 x cout).  da_prev is stored.
extern "C" int votenet_assembled_dgrad_bn_reduce(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                                 const float *wT, float *da_prev, const float *geo, const float *P, const float *wx,
                                                 const float *scale_prev, const float *shift_prev, const float *mean_prev,
                                                 const float *var_prev, float eps, int relu_prev, double *sums,
                                                 const votenet_coef_tail *tail, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0 && cout > 0, "assembled_dgrad_bn_reduce expects rows > 0, c > 0, cout > 0");
    VN_REQUIRE(da && zsrc && coef && wT && da_prev && geo && P && wx, "assembled_dgrad_bn_reduce: null buffer");
    VN_REQUIRE(scale_prev && shift_prev && mean_prev && var_prev && sums, "assembled_dgrad_bn_reduce: null buffer of the layer below");
    VN_REQUIRE(!tail || (tail->ticket && tail->gamma && tail->coef && tail->rows > 0), "assembled_dgrad_bn_reduce: incomplete coefficient tail");
    VN_REQUIRE((uintptr_t)geo % 16 == 0, "assembled_dgrad_bn_reduce: geo must be 16-byte aligned");
    FastArgs a = {};
    a.da = da;
    a.zsrc = zsrc;
    a.coef = coef;
    a.src_relu = relu;
    a.rows = rows;
    a.cin = c;
    a.cout = cout;
    a.w = wT;
    a.z = da_prev;
    a.geo = geo;
    a.ptab = P;
    a.wx = wx;
    a.e_scale = scale_prev;
    a.e_shift = shift_prev;
    a.e_mean = mean_prev;
    a.e_var = var_prev;
    a.e_eps = eps;
    a.e_relu = relu_prev;
    a.stats = sums;
    a.tail = to_tail(tail);
    if (!fast_dispatch<1, 6>(a, as_stream(stream)))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "assembled_dgrad_bn_reduce: shape not served (as votenet_mlp_dgrad_bn_reduce)");
    return check_launch("assembled_dgrad_bn_reduce");
}

// The same on the piece layout (half.hip): da / da_prev are TOTAL gradients per compact row; the affine part B + C z of the rebuilt
// dz is scaled by wh on the rows that stand for dropped pieces (SRC 5); the epilogue's sums over totals need no weight.
extern "C" int votenet_assembled_dgrad_bn_reduce_half(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                                      const float *wT, float *da_prev, const float *geo, const float *P, const float *wx,
                                                      const float *scale_prev, const float *shift_prev, const float *mean_prev,
                                                      const float *var_prev, float eps, int relu_prev, double *sums,
                                                      const votenet_coef_tail *tail, const float *wh, const int *nh_dev, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0 && cout > 0, "assembled_dgrad_bn_reduce_half expects rows > 0, c > 0, cout > 0");
    VN_REQUIRE(da && zsrc && coef && wT && da_prev && geo && P && wx && wh, "assembled_dgrad_bn_reduce_half: null buffer");
    VN_REQUIRE(scale_prev && shift_prev && mean_prev && var_prev && sums, "assembled_dgrad_bn_reduce_half: null buffer of the layer below");
    VN_REQUIRE(!tail || (tail->ticket && tail->gamma && tail->coef && tail->rows > 0), "assembled_dgrad_bn_reduce_half: incomplete coefficient tail");
    VN_REQUIRE((uintptr_t)geo % 16 == 0, "assembled_dgrad_bn_reduce_half: geo must be 16-byte aligned");
    FastArgs a = {};
    a.da = da;
    a.zsrc = zsrc;
    a.coef = coef;
    a.src_relu = relu;
    a.rows = rows;
    a.cin = c;
    a.cout = cout;
    a.w = wT;
    a.z = da_prev;
    a.geo = geo;
    a.ptab = P;
    a.wx = wx;
    a.e_scale = scale_prev;
    a.e_shift = shift_prev;
    a.e_mean = mean_prev;
    a.e_var = var_prev;
    a.e_eps = eps;
    a.e_relu = relu_prev;
    a.stats = sums;
    a.tail = to_tail(tail);
    a.wh = wh;
    a.nh_dev = nh_dev;
    a.xcd_chunk = g_fast_xcd_chunk;
    if (!fast_dispatch<5, 6>(a, as_stream(stream)))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "assembled_dgrad_bn_reduce_half: shape not served (as votenet_mlp_dgrad_bn_reduce)");
    return check_launch("assembled_dgrad_bn_reduce_half");
}
