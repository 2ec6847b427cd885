// fps.hip -- farthest point sampling for gfx950.
//
// Replaces farthestpointsamplingKernel / Launcher (tf_ops/sampling/tf_sampling_g.cu:105-170,203-205).
//
// FPS is a chain of m-1 dependent arg-max rounds per scene; with B = 8 scenes only 8 chains exist,
// so the kernel is bound by the latency of ONE round on ONE compute unit, not by HBM.  The reference
// keeps the running distances in global memory, re-reads every point past the first 3072 each round
// and spends 10 barriers per round (tf_sampling_g.cu:111-165).  Three kernels here:
//
//   fps_reg_kernel     n <= 4096.  The scene (xyz + running distance, 4 VGPRs per point) lives in the
//                      register file of one workgroup (16 waves x 1, 2 or 4 points per lane); a round is
//                      a few un-fused distance updates per lane, a DPP wave arg-max and ONE barrier.
//   fps_bucket_kernel  4096 < n <= 24576.  Same residency, plus EXACT spatial pruning: points are
//                      pre-sorted into buckets of 64 (one register slot of one wave) by a Morton cell
//                      sort (build_spatial_index); every bucket keeps its bounding box and the max of
//                      its running distances.  A new sample can only lower distances of points closer
//                      than their current running distance, so a bucket whose box is farther than its
//                      max running distance is skipped -- nothing in it can change.  The skip test is
//                      conservative (box distance shrunk by 1e-5 relative), skipped buckets keep their
//                      cached arg-max, and ties are still resolved with the reference's key on ORIGINAL
//                      indices: the sampled indices are bit-identical to the brute-force scan while the
//                      VALU work per round drops from n points to a handful of buckets.  Spatially
//                      adjacent buckets are dealt round-robin to the waves so the few active buckets of a
//                      round are updated in parallel.
//   fps_stream_kernel  n > 24576: running distances in the caller's temp buffer (reference layout).
//
// In all of them the winner's coordinates travel through the reduction (registers -> readlane -> one
// LDS exchange), so a round touches no global memory on its critical path; register arrays are
// 16-wide vector chunks so a wave-uniform slot index is one s_set_gpr_idx access, not a select chain.
//
// Tie rule (part of the result): winner = max d2, then smallest (k mod 512), then smallest k --
// the reference's 512-thread stride + left-biased tree.  Encoded as a 32-bit key
// ((k & 511) << 23 | k >> 9) minimised among the candidates that hold the max.
// Distances are evaluated un-fused, left to right (-ffp-contract=off), matching the oracle.  They are
// non-negative, so min / max / compare run on their bit patterns as unsigned integers (same order, no
// NaN canonicalisation code); lanes or slots without a point carry distance +0 and key 0xFFFFFFFF, which
// loses every tie against a real point exactly as the reference's (best = -1) idle threads do.
#include "common.h"

namespace votenet {

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned fps_tiekey(unsigned k) { return ((k & 511u) << 23) | (k >> 9); }
__device__ __forceinline__ unsigned fps_key_to_index(unsigned key) { return ((key & 0x7FFFFFu) << 9) | (key >> 23); }
__device__ __forceinline__ unsigned fbits(float f) { return __float_as_uint(f); }

// Non-finite coordinates (NaN / Inf: holes of a depth camera).  The reference leaves them to fminf / compare artefacts (a NaN point
// keeps its initial distance 1e38, is picked at once and freezes every running distance); here the result is DEFINED on every path:
// a point with a non-finite coordinate is read as a copy of point 0 -- it is at distance 0 from the first centre and is never sampled
// (index 0 wins every all-zero tie) -- and point 0's own non-finite components are read as 0.  Integer tests: this file is built
// with -fno-honor-nans.  The spatial index marks such points in `perm` (sign bit): the ball query over the index never reports them,
// like the full scan, where a NaN / Inf distance fails the radius test (oracle/oracle_sampling.c defines the same).
// The round loop of the bucket-pruned kernels is a dependent chain on one CU: its time moves by ~2 % with the 4-byte phase of the
// loop's code (measured: an edit of the PROLOGUE alone took sa1 from 1.659 to 1.692 ms on one box).  FPS_PAD_A / FPS_PAD_B shift the
// loops of fps_bucket_kernel / fps_bucket_l2_kernel by that many s_nop (4 bytes each); the values are the measured best.
#ifndef FPS_PAD_A
#define FPS_PAD_A 1 // same box, sa1: 0 -> 1.688 ms, 1 -> 1.660, 2 -> 1.681, 3 -> 1.684 (round 2's build: 1.658)
#endif
#ifndef FPS_PAD_B
#define FPS_PAD_B 3 // same box, 4 x 80000 -> 2048: 0 -> 2.92 ms, 1 -> 2.93, 2 -> 2.87, 3 -> 2.855 (round 2's build: 2.94).  Re-measure after any edit of this file
#endif
#define FPS_NOPS_0
#define FPS_NOPS_1 asm volatile("s_nop 0");
#define FPS_NOPS_2 FPS_NOPS_1 FPS_NOPS_1
#define FPS_NOPS_3 FPS_NOPS_2 FPS_NOPS_1
#define FPS_NOPS_4 FPS_NOPS_2 FPS_NOPS_2
#define FPS_NOPS_5 FPS_NOPS_4 FPS_NOPS_1
#define FPS_NOPS_6 FPS_NOPS_4 FPS_NOPS_2
#define FPS_NOPS_7 FPS_NOPS_4 FPS_NOPS_3
#define FPS_CAT_(a, b) a##b
#define FPS_CAT(a, b) FPS_CAT_(a, b)
#define FPS_LOOP_PAD_A FPS_CAT(FPS_NOPS_, FPS_PAD_A)
#define FPS_LOOP_PAD_B FPS_CAT(FPS_NOPS_, FPS_PAD_B)
__device__ __forceinline__ bool fps_nonfinite(float v)
{
    unsigned u = __float_as_uint(v);
    asm volatile("" : "+v"(u)); // opaque: under no-nans-fp-math the optimiser recognises the exponent test as a class test and drops its NaN half
    return (u & 0x7f800000u) == 0x7f800000u;
}
// point 0 (the first centre): three loads and three selects, no branch
__device__ __forceinline__ void fps_get0(const float *__restrict__ pts, float &x, float &y, float &z)
{
    const float a = pts[0], b = pts[1], c = pts[2];
    x = fps_nonfinite(a) ? 0.0f : a;
    y = fps_nonfinite(b) ? 0.0f : b;
    z = fps_nonfinite(c) ? 0.0f : c;
}
__device__ __forceinline__ bool fps_get(const float *__restrict__ pts, size_t k, float &x, float &y, float &z)
{
    x = pts[k * 3 + 0];
    y = pts[k * 3 + 1];
    z = pts[k * 3 + 2];
    const bool bad = fps_nonfinite(x) || fps_nonfinite(y) || fps_nonfinite(z);
    if (bad || k == 0) { // rare
        const float a = pts[0], b = pts[1], c = pts[2];
        x = fps_nonfinite(a) ? 0.0f : a;
        y = fps_nonfinite(b) ? 0.0f : b;
        z = fps_nonfinite(c) ? 0.0f : c;
    }
    return bad;
}

// ---- wave64 reductions on unsigned keys: six dependent VOP2-DPP steps, result uniform
#define FPS_DPP_REDUCE(OP)                                                                                   \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                 OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"              \
                 OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                  \
                 OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                       \
                 OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                     \
                 OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"                         \
                 : "+v"(v))
#define FPS_DPP_REDUCE16(OP)                                                                                 \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                 OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"              \
                 OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                  \
                 OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"                           \
                 : "+v"(v))
__device__ __forceinline__ unsigned wmax_u32(unsigned v)
{
    FPS_DPP_REDUCE("v_max_u32_dpp");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wmin_u32(unsigned v)
{
    FPS_DPP_REDUCE("v_min_u32_dpp");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned r16max_u32(unsigned v) // every lane of a 16-lane row gets the row result
{
    FPS_DPP_REDUCE16("v_max_u32_dpp");
    return v;
}
__device__ __forceinline__ unsigned r16min_u32(unsigned v)
{
    FPS_DPP_REDUCE16("v_min_u32_dpp");
    return v;
}

// ---- register arrays are ONE 16- or 32-wide vector each (v16f32 / v32f32 register tuples), so that a
// wave-uniform dynamic slot index is a single s_set_gpr_idx access: no select chain, no scratch
typedef float f32v __attribute__((ext_vector_type(32)));
template <int VW>
struct SlotVec;
template <>
struct SlotVec<16> {
    typedef f16v type;
};
template <>
struct SlotVec<32> {
    typedef f32v type;
};

// Wave arg-max under (value descending, key ascending).  Exact ties are rare, so the key reduction only
// runs when more than one lane holds the maximum.  Returns the winning lane; vmax / kmin uniform.
__device__ __forceinline__ int wave_argmax(unsigned val, unsigned key, unsigned &vmax, unsigned &kmin)
{
    vmax = wmax_u32(val);
    unsigned long long hit = __ballot(val == vmax);
    if (hit & (hit - 1)) { // several lanes hold the max
        kmin = wmin_u32(val == vmax ? key : 0xFFFFFFFFu);
        hit = __ballot(val == vmax && key == kmin);
        return __ffsll((long long)hit) - 1;
    }
    const int l = __ffsll((long long)hit) - 1;
    kmin = (unsigned)__builtin_amdgcn_readlane((int)key, l);
    return l;
}

// Sampled indices are collected in the registers of wave 0 (lane j & 63 holds round j) and written
// 64 at a time, so no global store sits between two rounds.
struct FpsOut {
    int *__restrict__ o;
    int m;
    int val;
    __device__ __forceinline__ void put(int j, int k, int tid)
    {
        if (tid < 64) {
            if ((j & 63) == tid) val = k;
            if ((j & 63) == 63 || j == m - 1) {
                const int base = j & ~63;
                if (base + tid <= j) o[base + tid] = val;
            }
        }
    }
};

struct FpsWinner {
    unsigned k;    // original point index
    float x, y, z; // its coordinates (uniform)
    unsigned d2;   // its running distance (bits) and its tie key re-packed to 28 bits (fps_cross_wave): what a further exchange
    unsigned key28; // between workgroups compares (fps_bucket_split_kernel)
};

// Cross-wave stage: every wave contributes (wmax, wkey, wx, wy, wz); returns the block winner (the same value in every
// lane).  The exchange is ONE LDS atomic: lane 0 of each wave does ds_max_u64 on a packed word
//     [ d2 (32) | ~key28 (28) | wave (4) ]      key28 = the tie key re-packed to 28 bits: same order, k < 2^28
// whose maximum is exactly "largest distance, then smallest tie key"; after the barrier a wave needs two LDS reads (the
// word, then the winner's coordinates from its candidate row) instead of reading all NW candidates and reducing them with
// DPP steps, ballots and readlanes.  That tail is executed by every wave of the workgroup at the same moment, i.e. it
// competes for issue slots NW/4-fold: measured 0.45 us of a 0.9 us round before.
// Buffers: three atomic words by round % 3 (wave 0 clears the NEXT round's word before this round's barrier: its last
// readers passed the previous barrier, its next writers come after this one) and two candidate tables by round parity.
// One barrier per round.  Layout inside s_ex (160 dwords, 16-byte aligned): words at [0,6), tables at [8, 8 + 2*16*4).
__device__ __forceinline__ void fps_cross_init(unsigned *s_ex)
{
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(s_ex);
    slot[0] = slot[1] = slot[2] = 0ull;
}

template <int NW>
__device__ __forceinline__ FpsWinner fps_cross_wave(unsigned wmax, unsigned wkey, float wx, float wy, float wz, unsigned *s_ex,
                                                    int round)
{
    FpsWinner r;
    if (NW == 1) {
        r.k = fps_key_to_index(wkey);
        r.x = wx;
        r.y = wy;
        r.z = wz;
        r.d2 = wmax;
        r.key28 = ((wkey >> 23) << 19) | (wkey & 0x7FFFFu);
        return r;
    }
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(s_ex);
    float4 *cand = reinterpret_cast<float4 *>(s_ex + 8) + (round & 1) * 16;
    const int w = wave_id_uniform();
    const int lane = lane_id();
    const int cur = round % 3, nxt = cur == 2 ? 0 : cur + 1;
    if (lane == 0) {
        const unsigned key28 = ((wkey >> 23) << 19) | (wkey & 0x7FFFFu);
        const unsigned low = ((0xFFFFFFFu - key28) << 4) | (unsigned)w;
        cand[w] = make_float4(wx, wy, wz, 0.0f);
        atomicMax(&slot[cur], ((unsigned long long)wmax << 32) | low);
        if (w == 0) slot[nxt] = 0ull;
    }
    // LDS-only barrier: __syncthreads() would also wait for vmcnt(0), i.e. for the global store of the
    // previous round's index to complete -- hundreds of cycles on the critical path of every round
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned long long v = slot[cur]; // behind the compiler barrier above
    const unsigned low = (unsigned)v;
    const unsigned key28 = 0xFFFFFFFu - (low >> 4);
    const float4 c = cand[low & 15u];
    r.k = (key28 >> 19) | ((key28 & 0x7FFFFu) << 9);
    r.x = c.x;
    r.y = c.y;
    r.z = c.z;
    r.d2 = (unsigned)(v >> 32);
    r.key28 = key28;
    return r;
}

// ------------------------------------------------------------------ brute force, register resident
// NW waves x P points per lane: n <= 64 * NW * P.  Slots must be ordered by the tie key (k mod 512, k) so
// that the strict '>' below keeps the smallest key of the lane.  With T = 64*NW threads:
//   T >= 512 : k = tid + i*T              (k mod 512 is the same for every slot of a lane)
//   T <  512 : a lane owns R = 512/T residues; slot i = a*Q + b  ->  k = b*512 + a*T + tid
template <int NW, int P>
__device__ __forceinline__ int fps_slot_to_k(int tid, int i)
{
    constexpr int T = NW * 64;
    if (T >= 512) return tid + i * T;
    constexpr int R = 512 / (T < 512 ? T : 512);
    constexpr int Q = (P / R) > 0 ? (P / R) : 1;
    static_assert(T >= 512 || P % R == 0, "P must be a multiple of 512/T");
    return (i % Q) * 512 + (i / Q) * T + tid;
}

// ------------------------------------------------------------------ "is the input already in farthest-point order?"
// In PointNet++ every level below the first samples from the previous level's centres, which ARE in farthest-point order:
// greedy FPS restricted to its own first picks reproduces them, so the answer is 0, 1, ..., m-1 -- unless an exact tie is
// resolved differently by the tie key on the subset's indices.  Two kernels CHECK every greedy step in parallel, spread
// over the GPU, instead of performing the steps one after the other on one CU (n <= 2048):
//   fps_prefix_t_kernel:     T[k] = running distance of point k when it would be picked = min_{i<k} d(k, i)   (thread per k)
//   fps_prefix_check_kernel: every point q follows its own running distance td_k(q) = min_{i<k} d(q, i) along k and confirms
//                            that point k beats it at step k: td_k(q) < T[k], or equal with the larger tie key (thread per q)
// Same fp32 expression, same order as the sampling kernels.  T lives in the output buffer itself (out[1..m-1]; out[0] is the
// verdict word: 0 = confirmed so far).  The sampling kernel then reads out[0]: confirmed -> it writes 0..m-1 and returns;
// otherwise it runs its rounds.  The check is exact, not a heuristic: it IS the greedy algorithm, evaluated for one
// candidate answer.
constexpr int kFpsPrefixMax = 2048;
// Step 1 alone (is point 1 the farthest from point 0?) settles almost every input that is NOT in farthest-point order; every
// workgroup of both kernels evaluates it over all n points itself (a handful of distances per thread, no communication) and
// leaves at once when it fails: an unordered cloud pays two near-empty launches, not the full check.
__device__ __forceinline__ bool fps_step1_fails(const float *__restrict__ pts, int n, int tid, int nthreads)
{
    float x0, y0, z0, x1, y1, z1;
    fps_get(pts, 0, x0, y0, z0);
    fps_get(pts, 1, x1, y1, z1);
    const float ex = x1 - x0, ey = y1 - y0, ez = z1 - z0;
    const unsigned t1 = fbits(ex * ex + ey * ey + ez * ez), k1 = fps_tiekey(1u);
    int bad = 0;
    for (int q = tid; q < n; q += nthreads) {
        float qx, qy, qz;
        fps_get(pts, (size_t)q, qx, qy, qz);
        const float dx = qx - x0, dy = qy - y0, dz = qz - z0;
        const unsigned d = fbits(dx * dx + dy * dy + dz * dz);
        if (q != 1 && !(d < t1 || (d == t1 && fps_tiekey((unsigned)q) > k1))) bad = 1;
    }
    return __syncthreads_or(bad) != 0;
}

// Both kernels give every k (every q) to EIGHT lanes: a serial walk over a thousand centres is an instruction-latency chain
// (about 50 us per wavefront however few wavefronts there are); eight interleaved (T) or consecutive (check) slices of it
// and a shuffle reduction / scan cut it to an eighth.
__global__ __launch_bounds__(256) void fps_prefix_t_kernel(int n, int m, const float *__restrict__ xyz, int *__restrict__ out)
{
    const float *__restrict__ pts = xyz + (size_t)blockIdx.y * n * 3;
    int *__restrict__ o = out + (size_t)blockIdx.y * m;
    const int tid = threadIdx.x, sub = tid & 7;
    if (fps_step1_fails(pts, n, tid, 256)) {
        if (blockIdx.x == 0 && tid == 0) o[0] = 1; // verdict: not 0..m-1
        return;
    }
    // group g of 8 lanes owns k = g and k = m-1-g: every group walks about m centres
    const int k0 = blockIdx.x * 32 + (tid >> 3);
    const bool live = 2 * k0 < m;
    const int ka = live ? k0 : 0, kb = live ? m - 1 - k0 : 0;
    float ax0, ay0, az0, bx0, by0, bz0;
    fps_get(pts, (size_t)ka, ax0, ay0, az0);
    fps_get(pts, (size_t)kb, bx0, by0, bz0);
    unsigned ta = fbits(1e38f), tb = ta;
    for (int i = sub; i < kb; i += 8) {
        float cx, cy, cz;
        fps_get(pts, (size_t)i, cx, cy, cz);
        const float bx = bx0 - cx, by = by0 - cy, bz = bz0 - cz; // point minus centre, tf_sampling_g.cu:142
        tb = min(tb, fbits(bx * bx + by * by + bz * bz));
        if (i < ka) {
            const float ax = ax0 - cx, ay = ay0 - cy, az = az0 - cz;
            ta = min(ta, fbits(ax * ax + ay * ay + az * az));
        }
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        ta = min(ta, (unsigned)__shfl_xor((int)ta, d));
        tb = min(tb, (unsigned)__shfl_xor((int)tb, d));
    }
    if (live && sub == 0) {
        if (ka == 0) ta = 0u; // out[0] doubles as the verdict word: 0 = confirmed so far
        o[ka] = (int)ta;
        if (kb != ka) o[kb] = (int)tb;
    }
}

__global__ __launch_bounds__(128) void fps_prefix_check_kernel(int n, int m, const float *__restrict__ xyz, int *__restrict__ out)
{
    __shared__ float4 P4[kFpsPrefixMax]; // centres 0..m-1: x, y, z, T (bits): one ds_read_b128 per greedy step
    const float *__restrict__ pts = xyz + (size_t)blockIdx.y * n * 3;
    int *__restrict__ o = out + (size_t)blockIdx.y * m;
    const int tid = threadIdx.x, sub = tid & 7;
    if (fps_step1_fails(pts, n, tid, 128)) return; // fps_prefix_t_kernel has set the verdict
    for (int k = tid; k < m; k += 128) {
        float px, py, pz;
        fps_get(pts, (size_t)k, px, py, pz);
        P4[k] = make_float4(px, py, pz, __int_as_float(k ? o[k] : 0));
    }
    __syncthreads();
    const int q = blockIdx.x * 16 + (tid >> 3);
    const bool live = q < n;
    const int qq = live ? q : 0;
    float x, y, z;
    fps_get(pts, (size_t)qq, x, y, z);
    const unsigned qkey = fps_tiekey((unsigned)qq);
    // lane `sub` owns the steps k in [k_lo, k_hi): step k uses centre k-1
    const int L = (m - 1 + 7) / 8;
    const int k_lo = 1 + sub * L, k_hi = min(m, k_lo + L);
    unsigned cm = fbits(1e38f); // min over this slice's centres ...
    for (int k = k_lo; k < k_hi; k++) {
        const float4 c = P4[k - 1];
        const float dx = x - c.x, dy = y - c.y, dz = z - c.z;
        cm = min(cm, fbits(dx * dx + dy * dy + dz * dz));
    }
    unsigned td = fbits(1e38f); // ... exclusive prefix over the slices before it = the running distance at k_lo
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)cm, d, 8);
        if (sub >= d) cm = min(cm, t);
    }
    {
        const unsigned t = (unsigned)__shfl_up((int)cm, 1, 8);
        if (sub > 0) td = t;
    }
    bool bad = false;
    for (int k = k_lo; k < k_hi; k++) {
        const float4 c = P4[k - 1];
        const float dx = x - c.x, dy = y - c.y, dz = z - c.z;
        td = min(td, fbits(dx * dx + dy * dy + dz * dz));
        const unsigned tk = __float_as_uint(P4[k].w);
        if (qq != k && !(td < tk || (td == tk && qkey > fps_tiekey((unsigned)k)))) bad = true;
    }
    if (live && bad) atomicOr(&o[0], 1);
}

template <int NW, int P>
__global__ __launch_bounds__(NW * 64) void fps_reg_kernel(int n, int m, const float *__restrict__ xyz, int *__restrict__ out,
                                                          int check_prefix)
{
    __shared__ __attribute__((aligned(16))) unsigned s_ex[2 * 16 * 5];
    const float *__restrict__ pts = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ o = out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x;
    if (check_prefix && o[0] == 0) { // fps_prefix_check_kernel confirmed every greedy step of 0..m-1 (uniform: one scene per block)
        for (int jj = tid; jj < m; jj += NW * 64) o[jj] = jj;
        return;
    }

    float x[P], y[P], z[P];
    unsigned td[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int k = fps_slot_to_k<NW, P>(tid, i);
        const bool valid = k < n;
        x[i] = y[i] = z[i] = 0.0f;
        if (valid) fps_get(pts, (size_t)k, x[i], y[i], z[i]);
        td[i] = valid ? fbits(1e38f) : 0u; // tf_sampling_g.cu:118
    }
    if (tid == 0) fps_cross_init(s_ex);
    __syncthreads();
    FpsOut fo = {o, m, 0};
    fo.put(0, 0, tid); // tf_sampling_g.cu:114-116
    float cx, cy, cz;
    fps_get0(pts, cx, cy, cz);
    for (int j = 1; j < m; j++) {
        unsigned best = 0u;
        int bi = 0;
        float bx = x[0], by = y[0], bz = z[0];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const float dx = x[i] - cx, dy = y[i] - cy, dz = z[i] - cz;
            const float d = dx * dx + dy * dy + dz * dz; // tf_sampling_g.cu:142, un-fused
            const unsigned d2 = min(fbits(d), td[i]);    // :143
            td[i] = d2;
            if (d2 > best) { // :146 strict: the lowest slot (smallest tie key of this lane) wins ties
                best = d2;
                bi = i;
                bx = x[i];
                by = y[i];
                bz = z[i];
            }
        }
        const int bk = fps_slot_to_k<NW, P>(tid, bi);
        const unsigned key = bk < n ? fps_tiekey((unsigned)bk) : 0xFFFFFFFFu;
        unsigned wmax, wkey;
        const int fl = wave_argmax(best, key, wmax, wkey);
        const FpsWinner win = fps_cross_wave<NW>(wmax, wkey, readlane_f32(bx, fl), readlane_f32(by, fl), readlane_f32(bz, fl), s_ex, j);
        cx = win.x;
        cy = win.y;
        cz = win.z;
        fo.put(j, (int)win.k, tid);
    }
}

// ------------------------------------------------------------------ spatial bucketing (Morton cell counting sort)
__device__ __forceinline__ unsigned part1by2_4(unsigned v) // spread 4 bits: abcd -> a00b00c00d
{
    v &= 0xF;
    v = (v | (v << 4)) & 0xC3;  // ab0000cd
    v = (v | (v << 2)) & 0x249; // a00b00c00d
    return v;
}

// The index is built by five small launches spread over the GPU (one workgroup per scene took 65 us for 8 x 20480
// points: three latency-bound passes and two rounds of LDS atomics on 8 of 256 CUs; global atomics on one histogram per
// scene are no better -- device-scope atomics are executed behind the XCDs' L2s):
//   bounds   16 workgroups per scene: partial bounding boxes of the cloud
//   hist     16 workgroups per scene: Morton cells (16 x 16 x 16 grid over the box) of a slice of the points counted in
//            LDS -> one histogram row per workgroup
//   scan     per scene: exclusive scan of the 16 x 4096 counters in (cell, workgroup) order
//   scatter  the same slices: position = LDS atomic increment of the workgroup's own offset row; perm[position] = k,
//            sorted[position] = (x, y, z, k)  (the order inside a cell is arbitrary: no result depends on it, ties are
//            resolved on original indices)
//   boxes    per bucket of 64 consecutive sorted points: bounding box (one wave per bucket)
__device__ __forceinline__ unsigned sidx_cell(const float *__restrict__ pts, int k, const float *lo, const float *inv)
{
    unsigned c = 0;
    float p3[3];
    fps_get(pts, (size_t)k, p3[0], p3[1], p3[2]);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int q = (int)((p3[a] - lo[a]) * inv[a]);
        q = q < 0 ? 0 : (q > 15 ? 15 : q);
        c |= part1by2_4((unsigned)q) << a;
    }
    return c;
}

// work layout per scene: [kSidxParts rows of kSidxCells counters | kSidxParts x 6 partial bounds (min xyz, max xyz)]
__global__ __launch_bounds__(256) void sidx_bounds_kernel(int n, const float *__restrict__ xyz, int *__restrict__ work)
{
    __shared__ float sred[6][4];
    const float *__restrict__ pts = xyz + (size_t)blockIdx.y * n * 3;
    int *__restrict__ wk = work + (size_t)blockIdx.y * kSidxWork;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * 256 + tid; i < n * 3; i += 256 * kSidxParts) { // coalesced: component a = i % 3
        float v = pts[i];
        const int a = i % 3;
        if (fps_nonfinite(v)) v = pts[a]; // a hole reads as point 0 (fps_get): leaves the bounds alone
        if (fps_nonfinite(v)) v = 0.0f;
        mn[0] = a == 0 ? fminf(mn[0], v) : mn[0];
        mx[0] = a == 0 ? fmaxf(mx[0], v) : mx[0];
        mn[1] = a == 1 ? fminf(mn[1], v) : mn[1];
        mx[1] = a == 1 ? fmaxf(mx[1], v) : mx[1];
        mn[2] = a == 2 ? fminf(mn[2], v) : mn[2];
        mx[2] = a == 2 ? fmaxf(mx[2], v) : mx[2];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min_f32(mn[a]), hi = wave_max_f32(mx[a]);
        if (lane == 0) {
            sred[a][w] = lo;
            sred[3 + a][w] = hi;
        }
    }
    __syncthreads();
    if (tid < 6) {
        float v = sred[tid][0];
        for (int i = 1; i < 4; i++) v = tid < 3 ? fminf(v, sred[tid][i]) : fmaxf(v, sred[tid][i]);
        reinterpret_cast<float *>(wk + kSidxCells * kSidxParts)[blockIdx.x * 6 + tid] = v;
    }
}

// lower corner and cells-per-unit of the scene's grid from the partial bounds: 96 threads fetch one float each, three reduce
__device__ __forceinline__ void sidx_grid(const int *__restrict__ wk, float *lo, float *inv)
{
    __shared__ float s_part[6 * kSidxParts];
    __shared__ float s_grid[6];
    const float *part = reinterpret_cast<const float *>(wk + kSidxCells * kSidxParts);
    const int tid = threadIdx.x;
    if (tid < 6 * kSidxParts) s_part[tid] = part[tid];
    __syncthreads();
    if (tid < 3) {
        float l = s_part[tid], h = s_part[3 + tid];
        for (int i = 1; i < kSidxParts; i++) {
            l = fminf(l, s_part[i * 6 + tid]);
            h = fmaxf(h, s_part[i * 6 + 3 + tid]);
        }
        s_grid[tid] = l;
        s_grid[3 + tid] = (h > l) ? 16.0f / (h - l) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = s_grid[a];
        inv[a] = s_grid[3 + a];
    }
}

__global__ __launch_bounds__(256) void sidx_hist_kernel(int n, const float *__restrict__ xyz, int *__restrict__ work)
{
    __shared__ int s_cnt[kSidxCells];
    const float *__restrict__ pts = xyz + (size_t)blockIdx.y * n * 3;
    int *__restrict__ wk = work + (size_t)blockIdx.y * kSidxWork;
    float lo[3], inv[3];
    for (int c = threadIdx.x; c < kSidxCells; c += 256) s_cnt[c] = 0;
    sidx_grid(wk, lo, inv); // (its barriers also order the clear above)
    const int chunk = (n + kSidxParts - 1) / kSidxParts;
    const int k1 = min(n, (int)(blockIdx.x + 1) * chunk);
    for (int k = blockIdx.x * chunk + threadIdx.x; k < k1; k += 256) atomicAdd(&s_cnt[sidx_cell(pts, k, lo, inv)], 1);
    __syncthreads();
    int4 *__restrict__ row = reinterpret_cast<int4 *>(wk + (size_t)blockIdx.x * kSidxCells);
    for (int c = threadIdx.x; c < kSidxCells / 4; c += 256) row[c] = reinterpret_cast<const int4 *>(s_cnt)[c];
}

__global__ __launch_bounds__(1024) void sidx_scan_kernel(int *__restrict__ work)
{
    __shared__ unsigned wsum[16];
    int *__restrict__ cnt = work + (size_t)blockIdx.x * kSidxWork;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint4 c[kSidxParts]; // thread t: cells 4t .. 4t+3 of every workgroup row (coalesced 16-byte accesses)
    unsigned tsum = 0;
#pragma unroll
    for (int p = 0; p < kSidxParts; p++) {
        c[p] = *reinterpret_cast<const uint4 *>(cnt + (size_t)p * kSidxCells + tid * 4);
        tsum += c[p].x + c[p].y + c[p].z + c[p].w;
    }
    unsigned incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned run = incl - tsum;
    for (int i = 0; i < w; i++) run += wsum[i];
    uint4 o[kSidxParts];
#pragma unroll
    for (int p = 0; p < kSidxParts; p++) { o[p].x = run; run += c[p].x; }
#pragma unroll
    for (int p = 0; p < kSidxParts; p++) { o[p].y = run; run += c[p].y; }
#pragma unroll
    for (int p = 0; p < kSidxParts; p++) { o[p].z = run; run += c[p].z; }
#pragma unroll
    for (int p = 0; p < kSidxParts; p++) { o[p].w = run; run += c[p].w; }
#pragma unroll
    for (int p = 0; p < kSidxParts; p++) *reinterpret_cast<uint4 *>(cnt + (size_t)p * kSidxCells + tid * 4) = o[p];
}

__global__ __launch_bounds__(256) void sidx_scatter_kernel(int n, const float *__restrict__ xyz, int *__restrict__ work,
                                                           int *__restrict__ perm, float4 *__restrict__ sorted)
{
    __shared__ int s_off[kSidxCells];
    const float *__restrict__ pts = xyz + (size_t)blockIdx.y * n * 3;
    int *__restrict__ wk = work + (size_t)blockIdx.y * kSidxWork;
    float lo[3], inv[3];
    const int4 *__restrict__ row = reinterpret_cast<const int4 *>(wk + (size_t)blockIdx.x * kSidxCells);
    for (int c = threadIdx.x; c < kSidxCells / 4; c += 256) reinterpret_cast<int4 *>(s_off)[c] = row[c];
    sidx_grid(wk, lo, inv);
    const int nb = (n + 63) / 64;
    const int chunk = (n + kSidxParts - 1) / kSidxParts;
    const int k1 = min(n, (int)(blockIdx.x + 1) * chunk);
    for (int k = blockIdx.x * chunk + threadIdx.x; k < k1; k += 256) {
        const int pos = atomicAdd(&s_off[sidx_cell(pts, k, lo, inv)], 1);
        float px, py, pz;
        const bool hole = fps_get(pts, (size_t)k, px, py, pz);
        perm[(size_t)blockIdx.y * n + pos] = hole ? (int)((unsigned)k | 0x80000000u) : k; // sign bit: never a ball-query neighbour
        sorted[(size_t)blockIdx.y * nb * 64 + pos] = make_float4(px, py, pz, __int_as_float(k));
    }
}

__global__ __launch_bounds__(256) void sidx_boxes_kernel(int n, float4 *__restrict__ sorted, float *__restrict__ bbox)
{
    const int nb = (n + 63) / 64;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= nb) return;
    float4 *__restrict__ sp = sorted + ((size_t)blockIdx.y * nb + g) * 64;
    const int p = g * 64 + lane;
    const bool valid = p < n;
    float4 v = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (valid) v = sp[lane];
    else sp[lane] = v; // padding of the last bucket
    const float xl = wave_min_f32(valid ? v.x : INFINITY), xh = wave_max_f32(valid ? v.x : -INFINITY);
    const float yl = wave_min_f32(valid ? v.y : INFINITY), yh = wave_max_f32(valid ? v.y : -INFINITY);
    const float zl = wave_min_f32(valid ? v.z : INFINITY), zh = wave_max_f32(valid ? v.z : -INFINITY);
    if (lane == 0) {
        float *__restrict__ bb = bbox + ((size_t)blockIdx.y * nb + g) * 6;
        bb[0] = xl; bb[1] = yl; bb[2] = zl;
        bb[3] = xh; bb[4] = yh; bb[5] = zh;
    }
}

// ------------------------------------------------------------------ exact bucket-pruned FPS
// NW waves per scene, P = VW (16 or 32) slots per lane.  Bucket g (64 consecutive Morton-sorted points) is slot
// g / NW of wave g % NW, so spatial neighbours sit in different waves.  Lane i (< P) of a wave additionally
// holds the metadata of that wave's bucket i: bounding box, max running distance, arg-max key and lane.
#ifdef FPS_TRACE
__device__ unsigned long long g_fps_trace[8]; // cycles summed over rounds, per phase, wave 0 of block 0
#define FPS_T(i)                                                                        \
    do {                                                                                \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
        if (blockIdx.x == 0 && tid == 0) g_fps_trace[i] += _t - _tprev;                 \
        _tprev = _t;                                                                    \
    } while (0)
#else
#define FPS_T(i)
#endif
#ifndef FPS_PRIO
#define FPS_PRIO 0 // probe builds (tools/probe/fps_prio.sh): s_setprio of the sampling waves against whatever shares their SIMDs
#endif
template <int NW, int VW>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(int n, int m, const float *__restrict__ xyz,
                                                             const int *__restrict__ perm, const float *__restrict__ bbox,
                                                             const float4 *__restrict__ sorted, int *__restrict__ out)
{
    if (FPS_PRIO) __builtin_amdgcn_s_setprio(FPS_PRIO);
    constexpr int P = VW;
    typedef typename SlotVec<VW>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *s_key = reinterpret_cast<unsigned *>(smem);                          // NW*P*64 tie keys by (slot, wave, lane)
    unsigned *s_ex = reinterpret_cast<unsigned *>(smem + (size_t)NW * P * 64 * 4); // 2 x 16 x 5 exchange
    const float *__restrict__ pts = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ o = out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id_uniform();
    const int nb = (n + 63) / 64;

    vec_t X, Y, Z, TD;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int g = i * NW + w;
        const int p = g * 64 + lane;
        const bool valid = p < n;
        unsigned key = 0xFFFFFFFFu;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) { // one coalesced 16-byte load: the index's sorted copy carries (x, y, z, original index)
            const float4 v = sorted[(size_t)blockIdx.x * nb * 64 + p];
            px = v.x;
            py = v.y;
            pz = v.z;
            key = fps_tiekey((unsigned)__float_as_int(v.w));
        }
        X[i] = px;
        Y[i] = py;
        Z[i] = pz;
        TD[i] = valid ? 1e38f : 0.0f; // tf_sampling_g.cu:118; an empty slot never wins
        s_key[(size_t)(i * NW + w) * 64 + lane] = key;
    }
    // metadata of bucket (slot = lane) in lanes < P
    const int myg = lane * NW + w;
    const bool hasb = lane < P && myg < nb;
    const float *__restrict__ bb = bbox + ((size_t)blockIdx.x * nb + (hasb ? myg : 0)) * 6;
    const float bxl = hasb ? bb[0] : INFINITY, byl = hasb ? bb[1] : INFINITY, bzl = hasb ? bb[2] : INFINITY;
    const float bxh = hasb ? bb[3] : -INFINITY, byh = hasb ? bb[4] : -INFINITY, bzh = hasb ? bb[5] : -INFINITY;
    unsigned bmax = hasb ? fbits(1e38f) : 0u;
    unsigned bkey = 0xFFFFFFFFu;
    int blane = 0;
    if (tid == 0) fps_cross_init(s_ex);
    __syncthreads();
    FpsOut fo = {o, m, 0};
    fo.put(0, 0, tid); // tf_sampling_g.cu:114-116
    float cx, cy, cz;
    fps_get0(pts, cx, cy, cz);
    FPS_LOOP_PAD_A
    unsigned cw_max = 0u, cw_key = 0xFFFFFFFFu; // this wave's cached winner (uniform)
    int cw_slot = -1;                           // ... and the slot (bucket) it lives in
    float cw_x = 0.f, cw_y = 0.f, cw_z = 0.f;
#ifdef FPS_TRACE
    unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int j = 1; j < m; j++) {
        FPS_T(0); // loop back-edge + output bookkeeping
        // (1) which of this wave's buckets can change?  lower bound of the distance to the bucket box,
        //     shrunk by 1e-5 so that it is below every fp32-evaluated point distance of the bucket.
        const float ex = fmaxf(fmaxf(bxl - cx, cx - bxh), 0.0f);
        const float ey = fmaxf(fmaxf(byl - cy, cy - byh), 0.0f);
        const float ez = fmaxf(fmaxf(bzl - cz, cz - bzh), 0.0f);
#if defined(FPS_ABLATE) && FPS_ABLATE == 2 // 2: additionally no box tests
        const float lb = INFINITY;
        (void)ex; (void)ey; (void)ez;
#else
        const float lb = (ex * ex + ey * ey + ez * ez) * 0.99999f;
#endif
        unsigned long long act = __ballot(hasb && !(lb >= __uint_as_float(bmax)));
#if defined(FPS_ABLATE) && FPS_ABLATE >= 1 // timing ablation (tools/probe/fps_round_trace.sh): results are NOT the FPS indices
        if (j > 1) act = 0;                 // 1: no touched-bucket work after the first round
#endif
        FPS_T(1); // box tests + ballot
        // (2) update the active buckets, refresh their cached arg-max
        bool changed = false; // uniform: did the cached entry of this wave's WINNING bucket change this round?
        while (act) {
            const int i = __ffsll((long long)act) - 1;
            act &= act - 1;
            const float dx = X[i] - cx, dy = Y[i] - cy, dz = Z[i] - cz; // uniform i: s_set_gpr_idx reads
            const float d = dx * dx + dy * dy + dz * dz;                // tf_sampling_g.cu:142, un-fused
            const unsigned d2 = min(fbits(d), fbits(TD[i]));            // :143
            TD[i] = __uint_as_float(d2);
            // running distances only decrease: if the cached arg-max lane kept its value, the bucket's
            // (max, key, lane) entry is still exact and no reduction is needed
            const int ol = __builtin_amdgcn_readlane(blane, i);
            const unsigned omax = (unsigned)__builtin_amdgcn_readlane((int)bmax, i);
            if ((unsigned)__builtin_amdgcn_readlane((int)d2, ol) != omax) {
                const unsigned key = s_key[(size_t)(i * NW + w) * 64 + lane];
                unsigned nmax, nkey;
                const int nl = wave_argmax(d2, key, nmax, nkey);
                if (lane == i) {
                    bmax = nmax;
                    bkey = nkey;
                    blane = nl;
                }
                changed = changed || (i == cw_slot);
            }
        }
        FPS_T(2); // touched buckets
        // (3) wave winner over the cached bucket entries (lanes < P).  Bucket maxima only ever decrease, so the winner
        //     can only change when the winning bucket's OWN entry changed; otherwise last round's winner (uniform
        //     registers) is still the maximum -- the buckets touched in a round are the ones near the new centre, rarely it.
        if (changed || j == 1) {
            const int ws = wave_argmax(lane < P ? bmax : 0u, lane < P ? bkey : 0xFFFFFFFFu, cw_max, cw_key); // winning bucket = slot
            cw_slot = ws;
            const int fl = __builtin_amdgcn_readlane(blane, ws);                                            // winning lane inside it
            cw_x = readlane_f32(X[ws], fl);
            cw_y = readlane_f32(Y[ws], fl);
            cw_z = readlane_f32(Z[ws], fl);
        }
        FPS_T(3); // wave winner
#if defined(FPS_ABLATE) && FPS_ABLATE == 3 // 3: additionally no cross-wave exchange (every wave follows its own winner)
        FpsWinner win;
        win.k = fps_key_to_index(cw_key);
        win.x = cw_x;
        win.y = cw_y;
        win.z = cw_z;
#else
        const FpsWinner win = fps_cross_wave<NW>(cw_max, cw_key, cw_x, cw_y, cw_z, s_ex, j);
#endif
        FPS_T(4); // cross-wave stage (includes the wait for the slowest wave)
        cx = win.x;
        cy = win.y;
        cz = win.z;
        fo.put(j, (int)win.k, tid);
    }
}

// ------------------------------------------------------------------ exact bucket-pruned FPS, ONE SCENE OVER W WORKGROUPS
// 24 576 < n <= W * 24 576 (config 5: 80 000 points).  fps_bucket_l2_kernel keeps such a scene's points in L2 and pays one dependent
// L2 round trip per round for the touched buckets (1.40 us per round, 0.46 of the HBM model).  Here W workgroups hold the scene in
// REGISTERS, fps_bucket_kernel's way -- bucket g belongs to workgroup g % W (spatial neighbours in different workgroups AND different
// waves: a round's ~30 touched buckets spread over W x NW waves) -- and agree on the winner every round through L2:
//   every workgroup runs fps_bucket_kernel's round on its part up to its own winner (in-CU exchange: one LDS atomic, one barrier);
//   wave 0 PUBLISHES (distance, tie key, x, y, z) as five 64-bit words [round | payload] -- relaxed device-scope stores, nothing waits
//   for them; EVERY wave then polls the 5 W words of the round (one 8-byte load per lane, lanes < 5 W) until all carry the round's
//   number and picks the maximum itself: no second barrier, every wave of every workgroup derives the same winner from the same words.
// A word is single-copy atomic, so a lane never sees half a record; the round number in every word makes a record complete exactly
// when its five words match.  Two sets of slots by round parity: a workgroup can publish round j + 1 only after every workgroup
// published round j, i.e. after all of them finished reading round j - 1 -- the set it overwrites.  The buffer is zeroed before the
// launch (round numbers start at 1).
// MEASURED (round 5, profiles/r05_fps_split.txt), and why it stays behind votenet_debug_fps_split(1): 1.90 us per round at config 5
// against the L2-resident kernel's 1.33.  The exchange alone is 0.52-0.55 us per round (tools/probe/src/xwg_publish_poll.hip: the
// round-4 budget's 0.85 came from a probe that serialised four L2 round trips) -- but a workgroup's own round is 0.82 us, not the
// budgeted 0.35 (a quarter of an 80 000-point scene keeps its waves as busy as a whole 20 480-point scene: the dense scan touches
// several times as many buckets per round), and in the loop the exchange costs 1.08 us because every round now waits for the slowest
// of 48 waves instead of 12.  Indices are bit-identical to the other kernels' (tests/test_gpu_parity.py).
// Placement: workgroups go round-robin over the 8 XCDs, so blocks x, x + 8, x + 16, ... share an XCD and its L2: scene = XCD (+ 8 per
// further turn), part = turn.  Deadlock: the parts of a scene wait for each other, so the whole grid must be resident -- the launcher
// takes this kernel only for grids of at most 256 workgroups -- and a poll gives up after 2^22 tries (~1 s; counted in
// g_fps_split_timeouts, read by votenet_debug_fps_split_timeouts: the result is then garbage, the GPU is not hung).
__device__ unsigned g_fps_split_timeouts;

template <int NW, int VW, int W>
__global__ __launch_bounds__(NW * 64) void fps_bucket_split_kernel(int b, int n, int m, const float *__restrict__ xyz,
                                                                   const float *__restrict__ bbox, const float4 *__restrict__ sorted,
                                                                   int *__restrict__ out, unsigned long long *__restrict__ xch, int ablate)
{
    constexpr int P = VW;
    typedef typename SlotVec<VW>::type vec_t;
    const int xcd = blockIdx.x & 7, turn = blockIdx.x >> 3;
    const int scene = xcd + 8 * (turn / W), q = turn % W;
    if (scene >= b) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *s_key = reinterpret_cast<unsigned *>(smem);                          // NW*P*64 tie keys by (slot, wave, lane)
    unsigned *s_ex = reinterpret_cast<unsigned *>(smem + (size_t)NW * P * 64 * 4); // 2 x 16 x 5 exchange
    const float *__restrict__ pts = xyz + (size_t)scene * n * 3;
    int *__restrict__ o = out + (size_t)scene * m;
    unsigned long long *__restrict__ xs = xch + (size_t)scene * 2 * W * 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id_uniform();
    const int nb = (n + 63) / 64;

    vec_t X, Y, Z, TD;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int g = (i * NW + w) * W + q; // the scene's bucket in this slot
        const int p = g * 64 + lane;
        const bool valid = p < n;
        unsigned key = 0xFFFFFFFFu;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) {
            const float4 v = sorted[(size_t)scene * nb * 64 + p];
            px = v.x;
            py = v.y;
            pz = v.z;
            key = fps_tiekey((unsigned)__float_as_int(v.w));
        }
        X[i] = px;
        Y[i] = py;
        Z[i] = pz;
        TD[i] = valid ? 1e38f : 0.0f; // tf_sampling_g.cu:118; an empty slot never wins
        s_key[(size_t)(i * NW + w) * 64 + lane] = key;
    }
    const int myg = (lane * NW + w) * W + q;
    const bool hasb = lane < P && myg < nb;
    const float *__restrict__ bb = bbox + ((size_t)scene * nb + (hasb ? myg : 0)) * 6;
    const float bxl = hasb ? bb[0] : INFINITY, byl = hasb ? bb[1] : INFINITY, bzl = hasb ? bb[2] : INFINITY;
    const float bxh = hasb ? bb[3] : -INFINITY, byh = hasb ? bb[4] : -INFINITY, bzh = hasb ? bb[5] : -INFINITY;
    unsigned bmax = hasb ? fbits(1e38f) : 0u;
    unsigned bkey = 0xFFFFFFFFu;
    int blane = 0;
    if (tid == 0) fps_cross_init(s_ex);
    __syncthreads();
    FpsOut fo = {o, m, 0};
    if (q == 0) fo.put(0, 0, tid); // tf_sampling_g.cu:114-116
    float cx, cy, cz;
    fps_get0(pts, cx, cy, cz);
    unsigned cw_max = 0u, cw_key = 0xFFFFFFFFu;
    int cw_slot = -1;
    float cw_x = 0.f, cw_y = 0.f, cw_z = 0.f;
    bool dead = false; // a poll timed out: keep going without waiting (the result is garbage, the kernel ends)
    for (int j = 1; j < m; j++) {
        // (1)-(3): fps_bucket_kernel's round on this workgroup's buckets
        const float ex = fmaxf(fmaxf(bxl - cx, cx - bxh), 0.0f);
        const float ey = fmaxf(fmaxf(byl - cy, cy - byh), 0.0f);
        const float ez = fmaxf(fmaxf(bzl - cz, cz - bzh), 0.0f);
        const float lb = (ex * ex + ey * ey + ez * ez) * 0.99999f;
        unsigned long long act = __ballot(hasb && !(lb >= __uint_as_float(bmax)));
        bool changed = false;
        while (act) {
            const int i = __ffsll((long long)act) - 1;
            act &= act - 1;
            const float dx = X[i] - cx, dy = Y[i] - cy, dz = Z[i] - cz;
            const float d = dx * dx + dy * dy + dz * dz;     // tf_sampling_g.cu:142, un-fused
            const unsigned d2 = min(fbits(d), fbits(TD[i])); // :143
            TD[i] = __uint_as_float(d2);
            const int ol = __builtin_amdgcn_readlane(blane, i);
            const unsigned omax = (unsigned)__builtin_amdgcn_readlane((int)bmax, i);
            if ((unsigned)__builtin_amdgcn_readlane((int)d2, ol) != omax) {
                const unsigned key = s_key[(size_t)(i * NW + w) * 64 + lane];
                unsigned nmax, nkey;
                const int nl = wave_argmax(d2, key, nmax, nkey);
                if (lane == i) {
                    bmax = nmax;
                    bkey = nkey;
                    blane = nl;
                }
                changed = changed || (i == cw_slot);
            }
        }
        if (changed || j == 1) {
            const int ws = wave_argmax(lane < P ? bmax : 0u, lane < P ? bkey : 0xFFFFFFFFu, cw_max, cw_key);
            cw_slot = ws;
            const int fl = __builtin_amdgcn_readlane(blane, ws);
            cw_x = readlane_f32(X[ws], fl);
            cw_y = readlane_f32(Y[ws], fl);
            cw_z = readlane_f32(Z[ws], fl);
        }
        const FpsWinner mine = fps_cross_wave<NW>(cw_max, cw_key, cw_x, cw_y, cw_z, s_ex, j); // this workgroup's winner, in every wave
        // (4) between the workgroups of the scene
        unsigned long long *__restrict__ cur = xs + (size_t)(j & 1) * W * 5;
        if (w == 0 && lane < 5) {
            const unsigned lo = lane == 0   ? mine.d2
                                : lane == 1 ? 0xFFFFFFFu - mine.key28 // larger = smaller tie key
                                : lane == 2 ? __float_as_uint(mine.x)
                                : lane == 3 ? __float_as_uint(mine.y)
                                            : __float_as_uint(mine.z);
            __hip_atomic_store(&cur[q * 5 + lane], ((unsigned long long)(unsigned)j << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long got = ((unsigned long long)(unsigned)j << 32);
        if (ablate == 1) { // timing ablation (votenet_debug_fps_split(2)): no exchange, every workgroup follows its own winner -- NOT the FPS indices
            cx = mine.x;
            cy = mine.y;
            cz = mine.z;
            continue;
        }
        for (int spin = 0; !dead; spin++) {
            if (lane < 5 * W) got = __hip_atomic_load(&cur[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(lane < 5 * W && (unsigned)(got >> 32) != (unsigned)j) == 0ull) break;
            if (spin > (1 << 22)) {
                dead = true;
                if (tid == 0) atomicAdd(&g_fps_split_timeouts, 1u);
            }
        }
        const int glo = (int)(unsigned)got;
        unsigned long long best = 0ull;
        int bq = 0;
#pragma unroll
        for (int qq = 0; qq < W; qq++) {
            const unsigned long long c = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(glo, 5 * qq) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane(glo, 5 * qq + 1);
            if (qq == 0 || c > best) { // records are distinct (a point has one tie key)
                best = c;
                bq = qq;
            }
        }
        cx = __uint_as_float((unsigned)__builtin_amdgcn_readlane(glo, 5 * bq + 2));
        cy = __uint_as_float((unsigned)__builtin_amdgcn_readlane(glo, 5 * bq + 3));
        cz = __uint_as_float((unsigned)__builtin_amdgcn_readlane(glo, 5 * bq + 4));
        if (q == 0) {
            const unsigned key28 = 0xFFFFFFFu - (unsigned)best;
            fo.put(j, (int)((key28 >> 19) | ((key28 & 0x7FFFFu) << 9)), tid);
        }
    }
}

// ------------------------------------------------------------------ exact bucket-pruned FPS, L2-resident points
// ------------------------------------------------------------------ two samples per round
// fps_bucket_kernel with up to TWO picks per round.  If q1 is the arg-max of a round and q2 the runner-up (in the full order:
// running distance, then tie key), and the distance from q2 to q1 -- evaluated exactly as the update would evaluate it -- is
// not below q2's running distance, then adding q1 leaves q2's running distance untouched while every other one can only
// decrease: q2 IS the arg-max of the next round, so it is emitted now and the round after next starts with both centres.
// (q2's running distance must be positive: at 0 the just-picked q1, whose own distance is 0 too, still has the smaller
// key.)  The sampled indices are the same, in the same order; on room scenes 94 % of the rounds emit two.
// The price per round: every wave also keeps its runner-up POINT (second-best bucket entry or the second-best point inside
// its best bucket, recomputed only when one of those two buckets was touched), the update runs for two centres, and the
// exchange carries eight words instead of one (still one ds_max_u64 instruction and one barrier: see (4) in the kernel).
template <int NW>
__device__ __forceinline__ unsigned long long fps_pack(unsigned dmax, unsigned key, int w)
{
    const unsigned key28 = ((key >> 23) << 19) | (key & 0x7FFFFu);
    return ((unsigned long long)dmax << 32) | (((0xFFFFFFFu - key28) << 4) | (unsigned)w);
}
__device__ __forceinline__ unsigned fps_unpack_index(unsigned low)
{
    const unsigned key28 = 0xFFFFFFFu - (low >> 4);
    return (key28 >> 19) | ((key28 & 0x7FFFFu) << 9);
}

template <int NW, int VW>
__global__ __launch_bounds__(NW * 64) void fps_bucket2_kernel(int n, int m, const float *__restrict__ xyz,
                                                              const int *__restrict__ perm, const float *__restrict__ bbox,
                                                              int *__restrict__ out, int allow_two)
{
    static_assert(NW <= 16, "wave number in 4 bits");
    constexpr int P = VW;
    typedef typename SlotVec<VW>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *s_key = reinterpret_cast<unsigned *>(smem);                          // NW*P*64 tie keys by (slot, wave, lane)
    unsigned char *ex = smem + (size_t)NW * P * 64 * 4;
    // exchange area: [3 rounds][4 bits][2 classes] atomic words | per round parity: best xyz[16], runner xyz[16], runner word[16]
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(ex);             // 24 words = 192 B
    float4 *tab = reinterpret_cast<float4 *>(ex + 192);                                 // 2 x 32 float4 = 1024 B
    unsigned long long *rword = reinterpret_cast<unsigned long long *>(ex + 192 + 1024); // 2 x 16 words = 256 B
    const float *__restrict__ pts = xyz + (size_t)blockIdx.x * n * 3;
    const int *__restrict__ pm = perm + (size_t)blockIdx.x * n;
    int *__restrict__ o = out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id_uniform();
    const int nb = (n + 63) / 64;

    vec_t X, Y, Z, TD;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int g = i * NW + w;
        const int p = g * 64 + lane;
        const bool valid = p < n;
        unsigned key = 0xFFFFFFFFu;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) {
            const int k = pm[p] & 0x7fffffff;
            fps_get(pts, (size_t)k, px, py, pz);
            key = fps_tiekey((unsigned)k);
        }
        X[i] = px;
        Y[i] = py;
        Z[i] = pz;
        TD[i] = valid ? 1e38f : 0.0f;
        s_key[(size_t)(i * NW + w) * 64 + lane] = key;
    }
    const int myg = lane * NW + w;
    const bool hasb = lane < P && myg < nb;
    const float *__restrict__ bb = bbox + ((size_t)blockIdx.x * nb + (hasb ? myg : 0)) * 6;
    const float bxl = hasb ? bb[0] : INFINITY, byl = hasb ? bb[1] : INFINITY, bzl = hasb ? bb[2] : INFINITY;
    const float bxh = hasb ? bb[3] : -INFINITY, byh = hasb ? bb[4] : -INFINITY, bzh = hasb ? bb[5] : -INFINITY;
    unsigned bmax = hasb ? fbits(1e38f) : 0u;
    unsigned bkey = 0xFFFFFFFFu;
    int blane = 0;
    if (tid < 24) slot[tid] = 0ull;
    __syncthreads();
    FpsOut fo = {o, m, 0};
    fo.put(0, 0, tid);
    float cx, cy, cz; // first centre of the round
    fps_get0(pts, cx, cy, cz);
    float ex2 = cx, ey2 = cy, ez2 = cz;          // second centre (valid when two)
    bool two = false;
    unsigned cw_max = 0u, cw_key = 0xFFFFFFFFu; // this wave's best point (uniform) ...
    int cw_slot = 0;
    float cw_x = 0.f, cw_y = 0.f, cw_z = 0.f;
    unsigned r_max = 0u, r_key = 0xFFFFFFFFu;   // ... and its runner-up point
    int r_slot = 0;
    float r_x = 0.f, r_y = 0.f, r_z = 0.f;
    int round = 0;
#ifdef FPS_TRACE
    unsigned long long _tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int j = 1; j < m; round++) {
        FPS_T(0);
        // (1) buckets that can change, for either centre
        float ax = fmaxf(fmaxf(bxl - cx, cx - bxh), 0.0f);
        float ay = fmaxf(fmaxf(byl - cy, cy - byh), 0.0f);
        float az = fmaxf(fmaxf(bzl - cz, cz - bzh), 0.0f);
        float lb = (ax * ax + ay * ay + az * az) * 0.99999f;
        if (two) {
            ax = fmaxf(fmaxf(bxl - ex2, ex2 - bxh), 0.0f);
            ay = fmaxf(fmaxf(byl - ey2, ey2 - byh), 0.0f);
            az = fmaxf(fmaxf(bzl - ez2, ez2 - bzh), 0.0f);
            lb = fminf(lb, (ax * ax + ay * ay + az * az) * 0.99999f);
        }
        unsigned long long act = __ballot(hasb && !(lb >= __uint_as_float(bmax)));
        const bool first = round == 0;
        // the runner-up is stale as soon as its bucket or the best bucket was touched (running distances inside may have dropped)
        bool need_r = first || ((act >> cw_slot) & 1ull) || ((act >> r_slot) & 1ull);
        FPS_T(1);
        // (2) update them
        bool changed = false;
        while (act) {
            const int i = __ffsll((long long)act) - 1;
            act &= act - 1;
            const float px = X[i], py = Y[i], pz = Z[i];
            const float dx = px - cx, dy = py - cy, dz = pz - cz;
            const float d = dx * dx + dy * dy + dz * dz; // tf_sampling_g.cu:142, un-fused
            unsigned d2 = min(fbits(d), fbits(TD[i]));  // :143
            if (two) {
                const float fx = px - ex2, fy = py - ey2, fz = pz - ez2;
                d2 = min(d2, fbits(fx * fx + fy * fy + fz * fz));
            }
            TD[i] = __uint_as_float(d2);
            const int ol = __builtin_amdgcn_readlane(blane, i);
            const unsigned omax = (unsigned)__builtin_amdgcn_readlane((int)bmax, i);
            if ((unsigned)__builtin_amdgcn_readlane((int)d2, ol) != omax) {
                const unsigned key = s_key[(size_t)(i * NW + w) * 64 + lane];
                unsigned nmax, nkey;
                const int nl = wave_argmax(d2, key, nmax, nkey);
                if (lane == i) {
                    bmax = nmax;
                    bkey = nkey;
                    blane = nl;
                }
                changed = changed || (i == cw_slot);
            }
        }
        FPS_T(2);
        // (3) the wave's best point and its runner-up
        if (changed || first) {
            const int ws = wave_argmax(lane < P ? bmax : 0u, lane < P ? bkey : 0xFFFFFFFFu, cw_max, cw_key);
            cw_slot = ws;
            const int fl = __builtin_amdgcn_readlane(blane, ws);
            cw_x = readlane_f32(X[ws], fl);
            cw_y = readlane_f32(Y[ws], fl);
            cw_z = readlane_f32(Z[ws], fl);
            need_r = true;
        }
        if (need_r) {
            unsigned b2max, b2key; // second-best bucket entry
            const bool other = lane < P && lane != cw_slot;
            const int ws2 = wave_argmax(other ? bmax : 0u, other ? bkey : 0xFFFFFFFFu, b2max, b2key);
            const int fl = __builtin_amdgcn_readlane(blane, cw_slot); // second-best point inside the best bucket
            const unsigned tdb = fbits(TD[cw_slot]);
            const unsigned kb = s_key[(size_t)(cw_slot * NW + w) * 64 + lane];
            unsigned i2max, i2key;
            const int l2 = wave_argmax(lane == fl ? 0u : tdb, lane == fl ? 0xFFFFFFFFu : kb, i2max, i2key);
            const bool inner = i2max > b2max || (i2max == b2max && i2key < b2key);
            r_max = inner ? i2max : b2max;
            r_key = inner ? i2key : b2key;
            r_slot = inner ? cw_slot : (ws2 < P ? ws2 : P - 1); // no second bucket: r_max is 0 and the point is never used
            const int rl = inner ? l2 : __builtin_amdgcn_readlane(blane, r_slot);
            r_x = readlane_f32(X[r_slot], rl);
            r_y = readlane_f32(Y[r_slot], rl);
            r_z = readlane_f32(Z[r_slot], rl);
        }
        FPS_T(3);
        // (4) ONE exchange for the block's two best points.  The best of every wave goes, by a single ds_max_u64, into four
        // pairs of words: pair b is split by bit b of the wave number.  The overall maximum is max(pair 0); the best of the
        // OTHER waves is the maximum over b of the word of pair b on the opposite side of the winner's bit b (every other wave
        // differs from the winner in some bit, and none of those words contains the winner).  The block's second point is
        // that or the winner wave's runner-up.
        const int cur = round % 3, nxt = cur == 2 ? 0 : cur + 1;
        unsigned long long *sl = slot + cur * 8;
        float4 *best_t = tab + (round & 1) * 32, *run_t = best_t + 16;
        unsigned long long *rw = rword + (round & 1) * 16;
        const unsigned long long mine = fps_pack<NW>(cw_max, cw_key, w);
        if (lane < 4) atomicMax(&sl[lane * 2 + ((w >> lane) & 1)], mine);
        if (lane == 0) {
            best_t[w] = make_float4(cw_x, cw_y, cw_z, 0.0f);
            run_t[w] = make_float4(r_x, r_y, r_z, 0.0f);
            rw[w] = fps_pack<NW>(r_max, r_key, w);
        }
        if (w == 0 && lane < 8) slot[nxt * 8 + lane] = 0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        unsigned long long v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = sl[e];
        const unsigned long long g1 = v[0] > v[1] ? v[0] : v[1];
        const int w1 = (int)((unsigned)g1 & 15u);
        unsigned long long g2 = rw[w1];
#pragma unroll
        for (int b2 = 0; b2 < 4; b2++) {
            const unsigned long long c = ((w1 >> b2) & 1) ? v[2 * b2] : v[2 * b2 + 1];
            g2 = c > g2 ? c : g2;
        }
        const int w2 = (int)((unsigned)g2 & 15u);
        const unsigned td2 = (unsigned)(g2 >> 32);
        const float4 c1 = best_t[w1];
        const float4 c2 = (w2 == w1) ? run_t[w1] : best_t[w2];
        FPS_T(4);
        // (5) one pick or two
        const float gx = c2.x - c1.x, gy = c2.y - c1.y, gz = c2.z - c1.z; // q2 as a point, q1 as the centre: the update's expression
        const float d12 = gx * gx + gy * gy + gz * gz;
        const bool both = allow_two && (j + 1 < m) && td2 != 0u && fbits(d12) >= td2;
        fo.put(j, (int)fps_unpack_index((unsigned)g1), tid);
        if (both) fo.put(j + 1, (int)fps_unpack_index((unsigned)g2), tid);
        cx = c1.x;
        cy = c1.y;
        cz = c1.z;
        ex2 = c2.x;
        ey2 = c2.y;
        ez2 = c2.z;
        two = both;
        j += both ? 2 : 1;
    }
}

// 24 576 < n <= NW*64*64*NBL.  The same pruning as fps_bucket_kernel, but only the bucket METADATA lives in registers
// (NBL buckets per lane: box, cached max / key / lane and the coordinates of that arg-max point); the Morton-sorted
// points with their running distance are a float4 array in the caller's scratch (16 B per point, 1.3 MB for 80 000
// points: resident in the XCD's L2).  A touched bucket costs one coalesced 1 KB read, a distance update and a 256 B
// write-back of the changed running distances; the loads of up to FB touched buckets are issued together so their
// L2 latency overlaps.  Bucket g is slot (g / NW) / 64 of lane (g / NW) % 64 of wave g % NW.
// The reference scans all n points every round (tf_sampling_g.cu:130-147): 80 000 points -> a few hundred touched.
template <int NW, int NBL>
__global__ __launch_bounds__(NW * 64) void fps_bucket_l2_kernel(int n, int m, const float *__restrict__ xyz,
                                                                const int *__restrict__ perm, const float *__restrict__ bbox,
                                                                float4 *__restrict__ sorted, int *__restrict__ out)
{
    constexpr int FB = 4; // buckets fetched together
    __shared__ __attribute__((aligned(16))) unsigned s_ex[2 * 16 * 5];
    const float *__restrict__ pts = xyz + (size_t)blockIdx.x * n * 3;
    const int *__restrict__ pm = perm + (size_t)blockIdx.x * n;
    const int nb = (n + 63) / 64;
    float4 *__restrict__ sp = sorted + (size_t)blockIdx.x * nb * 64;
    int *__restrict__ o = out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = wave_id_uniform();

    // sorted copy: (x, y, z, running distance); padding points of the last bucket can never win (distance 0)
    for (int p = tid; p < nb * 64; p += NW * 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < n) {
            v = sp[p];    // the index's sorted copy already holds the point (holes read as point 0: sidx_scatter_kernel) ...
            v.w = 1e38f;  // ... its w, the original index (readers take indices from perm), becomes the running distance, tf_sampling_g.cu:118
        }
        sp[p] = v;
    }
    // metadata of this lane's buckets
    float bxl[NBL], byl[NBL], bzl[NBL], bxh[NBL], byh[NBL], bzh[NBL], wx[NBL], wy[NBL], wz[NBL];
    unsigned bmax[NBL], bkey[NBL];
    int blane[NBL];
    bool hasb[NBL];
#pragma unroll
    for (int s = 0; s < NBL; s++) {
        const int g = (s * 64 + lane) * NW + w;
        hasb[s] = g < nb;
        const float *__restrict__ bb = bbox + ((size_t)blockIdx.x * nb + (hasb[s] ? g : 0)) * 6;
        bxl[s] = hasb[s] ? bb[0] : INFINITY;
        byl[s] = hasb[s] ? bb[1] : INFINITY;
        bzl[s] = hasb[s] ? bb[2] : INFINITY;
        bxh[s] = hasb[s] ? bb[3] : -INFINITY;
        byh[s] = hasb[s] ? bb[4] : -INFINITY;
        bzh[s] = hasb[s] ? bb[5] : -INFINITY;
        bmax[s] = hasb[s] ? fbits(1e38f) : 0u;
        bkey[s] = 0xFFFFFFFFu;
        blane[s] = 0;
        wx[s] = wy[s] = wz[s] = 0.0f;
    }
    if (tid == 0) fps_cross_init(s_ex);
    __syncthreads(); // the sorted copy (written by other waves) is read below: block-scope release / acquire
    FpsOut fo = {o, m, 0};
    fo.put(0, 0, tid); // tf_sampling_g.cu:114-116
    float cx, cy, cz;
    fps_get0(pts, cx, cy, cz);
    FPS_LOOP_PAD_B
    unsigned cw_max = 0u, cw_key = 0xFFFFFFFFu; // this wave's cached winner (uniform)
    int cw_slot = -1;                           // ... and its bucket (slot * 64 + lane)
    float cw_x = 0.f, cw_y = 0.f, cw_z = 0.f;
    for (int j = 1; j < m; j++) {
        bool changed = false;
#pragma unroll
        for (int s = 0; s < NBL; s++) {
            // (1) which buckets of slot s can change?  (see fps_bucket_kernel)
            const float ex = fmaxf(fmaxf(bxl[s] - cx, cx - bxh[s]), 0.0f);
            const float ey = fmaxf(fmaxf(byl[s] - cy, cy - byh[s]), 0.0f);
            const float ez = fmaxf(fmaxf(bzl[s] - cz, cz - bzh[s]), 0.0f);
            const float lb = (ex * ex + ey * ey + ez * ez) * 0.99999f;
            unsigned long long act = __ballot(hasb[s] && !(lb >= __uint_as_float(bmax[s])));
            // (2) fetch up to FB touched buckets at once, then update them one by one
            while (act) {
                int li[FB];
                float4 pv[FB];
                int pk[FB];
                int cnt = 0;
#pragma unroll
                for (int f = 0; f < FB; f++) {
                    li[f] = -1;
                    if (act) {
                        li[f] = __ffsll((long long)act) - 1;
                        act &= act - 1;
                        const size_t base = (size_t)((s * 64 + li[f]) * NW + w) * 64 + lane;
                        pv[f] = sp[base];
                        pk[f] = (base < (size_t)n) ? pm[base] : -1; // a hole's entry is negative (sidx_scatter_kernel): like padding, it never wins a tie
                        cnt = f + 1;
                    }
                }
#pragma unroll
                for (int f = 0; f < FB; f++) {
                    if (f < cnt) {
                        const int i = li[f];
                        const float4 v = pv[f];
                        const float dx = v.x - cx, dy = v.y - cy, dz = v.z - cz;
                        const float d = dx * dx + dy * dy + dz * dz; // tf_sampling_g.cu:142, un-fused
                        const unsigned t0 = fbits(v.w);
                        const unsigned d2 = min(fbits(d), t0);       // :143
                        if (d2 != t0) sp[(size_t)((s * 64 + i) * NW + w) * 64 + lane].w = __uint_as_float(d2);
                        const int ol = __builtin_amdgcn_readlane(blane[s], i);
                        const unsigned omax = (unsigned)__builtin_amdgcn_readlane((int)bmax[s], i);
                        if ((unsigned)__builtin_amdgcn_readlane((int)d2, ol) != omax) {
                            const unsigned key = pk[f] >= 0 ? fps_tiekey((unsigned)pk[f]) : 0xFFFFFFFFu;
                            unsigned nmax, nkey;
                            const int nl = wave_argmax(d2, key, nmax, nkey);
                            const float nx = readlane_f32(v.x, nl), ny = readlane_f32(v.y, nl), nz = readlane_f32(v.z, nl);
                            if (lane == i) {
                                bmax[s] = nmax;
                                bkey[s] = nkey;
                                blane[s] = nl;
                                wx[s] = nx;
                                wy[s] = ny;
                                wz[s] = nz;
                            }
                            changed = changed || (s * 64 + i == cw_slot); // only the winning bucket's change matters
                        }
                    }
                }
            }
        }
        // (3) wave winner over the cached bucket entries; unchanged entries -> last round's winner is still valid
        if (changed || j == 1) {
            unsigned lm = bmax[0], lk = bkey[0];
            float lx = wx[0], ly = wy[0], lz = wz[0];
#pragma unroll
            for (int s = 1; s < NBL; s++) {
                const bool better = bmax[s] > lm || (bmax[s] == lm && bkey[s] < lk);
                lm = better ? bmax[s] : lm;
                lk = better ? bkey[s] : lk;
                lx = better ? wx[s] : lx;
                ly = better ? wy[s] : ly;
                lz = better ? wz[s] : lz;
            }
            int ls = 0; // the slot the lane's best entry comes from
#pragma unroll
            for (int s = 1; s < NBL; s++)
                if (bmax[s] == lm && bkey[s] == lk) ls = s;
            const int wl = wave_argmax(lm, lk, cw_max, cw_key);
            cw_slot = __builtin_amdgcn_readlane(ls, wl) * 64 + wl;
            cw_x = readlane_f32(lx, wl);
            cw_y = readlane_f32(ly, wl);
            cw_z = readlane_f32(lz, wl);
        }
#if defined(FPS_ABLATE) && FPS_ABLATE == 3 // 3: additionally no cross-wave exchange (every wave follows its own winner)
        FpsWinner win;
        win.k = fps_key_to_index(cw_key);
        win.x = cw_x;
        win.y = cw_y;
        win.z = cw_z;
#else
        const FpsWinner win = fps_cross_wave<NW>(cw_max, cw_key, cw_x, cw_y, cw_z, s_ex, j);
#endif
        cx = win.x;
        cy = win.y;
        cz = win.z;
        fo.put(j, (int)win.k, tid);
    }
}

// ------------------------------------------------------------------ streaming fallback (n > 24576)
template <int NW>
__global__ __launch_bounds__(NW * 64) void fps_stream_kernel(int b, int n, int m, const float *__restrict__ xyz,
                                                              float *__restrict__ temp, int *__restrict__ out)
{
    constexpr int T = NW * 64;
    static_assert(T % 512 == 0, "k mod 512 must be constant per lane");
    __shared__ __attribute__((aligned(16))) unsigned s_ex[2 * 16 * 5];
    const int tid = threadIdx.x;
    unsigned *__restrict__ td = reinterpret_cast<unsigned *>(temp) + (size_t)blockIdx.x * n;
    for (int scene = blockIdx.x; scene < b; scene += gridDim.x) {
        if (tid == 0) fps_cross_init(s_ex);
        __syncthreads();
        const float *__restrict__ pts = xyz + (size_t)scene * n * 3;
        int *__restrict__ o = out + (size_t)scene * m;
        for (int k = tid; k < n; k += T) td[k] = fbits(1e38f);
        FpsOut fo = {o, m, 0};
        fo.put(0, 0, tid);
        float cx, cy, cz;
        fps_get0(pts, cx, cy, cz);
        for (int j = 1; j < m; j++) {
            unsigned best = 0u, bk = (unsigned)tid;
            float bx = 0.f, by = 0.f, bz = 0.f;
            bool any = false;
            for (int k = tid; k < n; k += T) { // ascending k, k mod 512 constant per lane
                float px, py, pz;
                fps_get(pts, (size_t)k, px, py, pz);
                const float dx = px - cx, dy = py - cy, dz = pz - cz;
                const float d = dx * dx + dy * dy + dz * dz;
                const unsigned t0 = td[k];
                const unsigned d2 = min(fbits(d), t0);
                if (d2 != t0) td[k] = d2;
                if (!any || d2 > best) {
                    best = d2;
                    bk = (unsigned)k;
                    bx = px;
                    by = py;
                    bz = pz;
                    any = true;
                }
            }
            const unsigned key = any ? fps_tiekey(bk) : 0xFFFFFFFFu;
            unsigned wmax, wkey;
            const int fl = wave_argmax(best, key, wmax, wkey);
            const FpsWinner win = fps_cross_wave<NW>(wmax, wkey, readlane_f32(bx, fl), readlane_f32(by, fl), readlane_f32(bz, fl),
                                                     s_ex, j);
            cx = win.x;
            cy = win.y;
            cz = win.z;
            fo.put(j, (int)win.k, tid);
        }
        __syncthreads(); // td[] is re-initialised by other lanes for the next scene
    }
}

#ifdef FPS_TRACE
extern "C" void votenet_fps_trace_read(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(votenet::g_fps_trace), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(votenet::g_fps_trace), z, sizeof(z));
    }
}
#endif
static const int kFpsRegMax = 4096;         // brute-force register kernel
static const int kFpsBucketMax = 1024 * 24; // bucket-pruned register kernel
static const int kFpsL2Max = 16 * 64 * 64 * 4; // bucket-pruned kernel with L2-resident points (262 144)
static const int kFpsSplitW = 12;                          // fps_bucket_split_kernel: at most this many workgroups per scene (exchange words) ...
static const int kFpsSplitMax = 4 * 12 * 32 * 64;          // ... together holding 48 waves x 32 slots x 64 points in registers (98 304)
static const int kFpsSplitScenes = 16;                     // 8 XCDs x 2 turns x 12 parts = 192 workgroups: all resident

} // namespace votenet

int votenet::build_spatial_index(int b, int n, const float *xyz, float *index, hipStream_t st)
{
    const SpatialIndex v = spatial_index_view(index, b, n);
    const int nb = (n + 63) / 64;
    hipLaunchKernelGGL(sidx_bounds_kernel, dim3(kSidxParts, b), dim3(256), 0, st, n, xyz, v.work);
    hipLaunchKernelGGL(sidx_hist_kernel, dim3(kSidxParts, b), dim3(256), 0, st, n, xyz, v.work);
    hipLaunchKernelGGL(sidx_scan_kernel, dim3(b), dim3(1024), 0, st, v.work);
    hipLaunchKernelGGL(sidx_scatter_kernel, dim3(kSidxParts, b), dim3(256), 0, st, n, xyz, v.work, v.perm, v.sorted);
    hipLaunchKernelGGL(sidx_boxes_kernel, dim3((nb + 3) / 4, b), dim3(256), 0, st, n, v.sorted, v.bbox);
    return check_launch("spatial_index");
}

using namespace votenet;

extern "C" size_t votenet_spatial_index_floats(int b, int n) { return spatial_index_floats(b, n); }

extern "C" int votenet_spatial_index(int b, int n, const float *xyz, float *index, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0, "spatial_index expects (batch_size, num_points, 3) xyz shape");
    if (b == 0) return VOTENET_OK;
    VN_REQUIRE(xyz && index, "spatial_index: null buffer");
    return build_spatial_index(b, n, xyz, index, as_stream(stream));
}

static size_t fps_split_floats(int b) { return (size_t)b * 2 * kFpsSplitW * 5 * 2 + 4; } // [b][2][W][5] 64-bit words behind the index (+ alignment)

extern "C" size_t votenet_fps_temp_floats(int b, int n)
{
    if (n <= kFpsRegMax) return 0;
    if (n > kFpsBucketMax && n <= kFpsSplitMax) return spatial_index_floats(b, n) + fps_split_floats(b); // the index + the split kernel's exchange
    if (n <= kFpsL2Max) return spatial_index_floats(b, n); // the spatial index: permutation, bucket boxes, sorted float4, work
    return (size_t)(b < 32 ? b : 32) * (size_t)n;                                          // running distances, tf_sampling.cpp:115
}

#define FPS_LAUNCH(NW, P) hipLaunchKernelGGL((fps_reg_kernel<NW, P>), dim3(b), dim3(NW * 64), 0, st, n, m, inp, out, check_prefix)
static int g_fps_dbg_nw = 0, g_fps_dbg_p = 0;
static int g_fps_prefix_check = 1;
extern "C" void votenet_fps_debug_prefix_check(int on) // measurement hook: 0 = always run the sampling rounds
{
    VN_DEBUG_GATE();
    g_fps_prefix_check = on;
}
static int g_fps_two_pick = 0; // 1: two samples per round; 2: that kernel with the second pick disabled (measurement)
extern "C" void votenet_fps_debug_two_pick(int on) // experiment hook: fps_bucket2_kernel (two samples per round) for 4096 < n <= 24576
{
    VN_DEBUG_GATE();
    g_fps_two_pick = on;
}
extern "C" void votenet_fps_debug_config(int nw, int p) // tuning hook: force a brute-force configuration (0,0 = automatic)
{
    VN_DEBUG_GATE();
    g_fps_dbg_nw = nw;
    g_fps_dbg_p = p;
}
#define FPS_BUCKET2_LAUNCH(NW, VW)                                                                                 \
    do {                                                                                                           \
        constexpr size_t lds = (size_t)NW * VW * 64 * 4 + 192 + 1024 + 256;                                        \
        static bool attr_set2 = false;                                                                             \
        if (!attr_set2) {                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_bucket2_kernel<NW, VW>),                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
            attr_set2 = true;                                                                                      \
        }                                                                                                          \
        hipLaunchKernelGGL((fps_bucket2_kernel<NW, VW>), dim3(b), dim3(NW * 64), lds, st, n, m, inp, (const int *)sidx.perm, \
                           (const float *)sidx.bbox, out, g_fps_two_pick == 1 ? 1 : 0);                            \
    } while (0)
#define FPS_BUCKET_LAUNCH(NW, VW)                                                                                  \
    do {                                                                                                           \
        constexpr size_t lds_need = (size_t)NW * VW * 64 * 4 + 2 * 16 * 5 * 4;                                     \
        /* g_fps_lds_floor: ask for more LDS than the kernel uses, so that no other workgroup fits on its CU (see  \
           votenet_debug_fps_lds_floor) */                                                                        \
        const size_t lds = lds_need > (size_t)g_fps_lds_floor ? lds_need : (size_t)g_fps_lds_floor;                \
        static size_t attr_set = 0;                                                                                \
        if (attr_set != lds) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_bucket_kernel<NW, VW>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
            attr_set = lds;                                                                                        \
        }                                                                                                          \
        hipLaunchKernelGGL((fps_bucket_kernel<NW, VW>), dim3(b), dim3(NW * 64), lds, st, n, m, inp, (const int *)sidx.perm, \
                           (const float *)sidx.bbox, (const float4 *)sidx.sorted, out);                            \
    } while (0)

static int g_fps_split = 0; // 1 / 3 / 5: 24 576 < n <= 98 304 samples one scene over 4 / 12 / 6 workgroups (fps_bucket_split_kernel) -- measured SLOWER than
                            // the L2-resident kernel (1.9-2.6 vs 1.33 us per round at config 5, profiles/r05_fps_split.txt): off; 2 / 4 / 6: without the exchange (timing)
extern "C" void votenet_debug_fps_split(int on) { VN_DEBUG_GATE(); g_fps_split = on; } // A/B and test hook
extern "C" unsigned votenet_debug_fps_split_timeouts(void)            // polls of the split kernel that gave up (0 unless a part never ran)
{
    unsigned v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fps_split_timeouts), sizeof(v));
    return v;
}
static int g_fps_lds_floor = 0;
extern "C" void votenet_debug_fps_lds_floor(int bytes) { VN_DEBUG_GATE(); g_fps_lds_floor = bytes > 0 ? bytes : 0; } // tuning hook

template <int NW, int VW, int W>
static int fps_split_launch(int b, int n, int m, const float *inp, float *temp, const float *boxes, const float4 *sorted, int *out, int ablate,
                            hipStream_t st)
{
    static_assert(NW * VW * W * 64 >= kFpsSplitMax && 5 * W <= 64 && W <= kFpsSplitW, "split shape");
    constexpr size_t lds = (size_t)NW * VW * 64 * 4 + 2 * 16 * 5 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_bucket_split_kernel<NW, VW, W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    unsigned long long *xch =
        reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(temp + spatial_index_floats(b, n)) + 15) & ~(uintptr_t)15);
    const hipError_t me = hipMemsetAsync(xch, 0, (fps_split_floats(b) - 4) * sizeof(float), st);
    if (me != hipSuccess) return set_error(VOTENET_E_HIP, "FarthestPointSample (split): clearing the exchange words: %s", hipGetErrorString(me));
    hipLaunchKernelGGL((fps_bucket_split_kernel<NW, VW, W>), dim3(8 * W * ((b + 7) / 8)), dim3(NW * 64), lds, st, b, n, m, inp, boxes, sorted,
                       out, xch, ablate);
    return VOTENET_OK;
}

extern "C" int votenet_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream)
{
    VN_REQUIRE(m > 0, "FarthestPointSample expects positive npoint");                               // tf_sampling.cpp:99
    VN_REQUIRE(b >= 0 && n > 0, "FarthestPointSample expects (batch_size,num_points,3) inp shape"); // :105
    VN_REQUIRE(inp && out, "FarthestPointSample: null buffer");
    if (b == 0) return VOTENET_OK;
    hipStream_t st = as_stream(stream);
    if (n > kFpsRegMax)
        VN_REQUIRE(temp != nullptr, "FarthestPointSample: temp scratch of %zu floats required for n=%d",
                   votenet_fps_temp_floats(b, n), n);
    // inputs that may already be in farthest-point order (every PointNet++ level below the first): check all greedy steps in
    // parallel first; the sampling kernel then returns at once or, after an exact tie, runs as usual
    const int check_prefix = (g_fps_prefix_check && n <= kFpsPrefixMax && m <= n && m > 1) ? 1 : 0;
    if (check_prefix) {
        hipLaunchKernelGGL(fps_prefix_t_kernel, dim3((m / 2 + 32) / 32, b), dim3(256), 0, st, n, m, inp, out);
        hipLaunchKernelGGL(fps_prefix_check_kernel, dim3((n + 15) / 16, b), dim3(128), 0, st, n, m, inp, out);
    }
    if (g_fps_dbg_nw && n <= 64 * g_fps_dbg_nw * g_fps_dbg_p) {
#define FPS_DBG(NW, P) if (g_fps_dbg_nw == NW && g_fps_dbg_p == P) { FPS_LAUNCH(NW, P); return check_launch("farthest_point_sample"); }
        FPS_DBG(1, 8) FPS_DBG(1, 16) FPS_DBG(2, 4) FPS_DBG(2, 8) FPS_DBG(2, 16) FPS_DBG(4, 2) FPS_DBG(4, 4) FPS_DBG(4, 8) FPS_DBG(4, 16)
        FPS_DBG(8, 1) FPS_DBG(8, 2) FPS_DBG(8, 4) FPS_DBG(8, 8) FPS_DBG(16, 1) FPS_DBG(16, 2) FPS_DBG(16, 4)
    }
    if (n <= 1024) {
        FPS_LAUNCH(4, 4);
    } else if (n <= 2048) {
        FPS_LAUNCH(4, 8);
    } else if (n <= 4096) {
        FPS_LAUNCH(8, 8);
    } else if (n <= kFpsBucketMax) {
        const SpatialIndex sidx = spatial_index_view(temp, b, n);
        if (build_spatial_index(b, n, inp, temp, st) != VOTENET_OK) return VOTENET_E_HIP;
        const bool single = !g_fps_two_pick; // measured: the two-pick rounds are 1.9x as long as the plain ones (DESIGN_HISTORY.md 4.1)
        if (n <= 16 * 16 * 64) {
            if (single) FPS_BUCKET_LAUNCH(16, 16); // 16 waves x 16 slots
            else FPS_BUCKET2_LAUNCH(16, 16);
        } else {
            if (single) FPS_BUCKET_LAUNCH(12, 32); // 12 waves x 32 slots: 3 waves per SIMD, 4 x 32 data VGPRs of the 168 available
            else FPS_BUCKET2_LAUNCH(12, 32);
        }
    } else if (n <= kFpsL2Max) {
        const int nb = (n + 63) / 64;
        const SpatialIndex sidx = spatial_index_view(temp, b, n);
        if (build_spatial_index(b, n, inp, temp, st) != VOTENET_OK) return VOTENET_E_HIP;
        float *boxes = sidx.bbox;
        float4 *sorted = sidx.sorted;
        if (g_fps_split && n <= kFpsSplitMax && b <= kFpsSplitScenes) {
            // 1 (and 2 = its timing ablation): 4 workgroups x 12 waves; 3 (4): 12 workgroups x 4 waves -- one wave per SIMD, every
            // instruction of the round issued once per SIMD instead of three times; 5 (6): 6 workgroups x 8 waves
            const int abl = (g_fps_split % 2 == 0) ? 1 : 0;
            int rc;
            if (g_fps_split <= 2) rc = fps_split_launch<12, 32, 4>(b, n, m, inp, temp, boxes, sorted, out, abl, st);
            else if (g_fps_split <= 4) rc = fps_split_launch<4, 32, 12>(b, n, m, inp, temp, boxes, sorted, out, abl, st);
            else rc = fps_split_launch<8, 32, 6>(b, n, m, inp, temp, boxes, sorted, out, abl, st);
            if (rc != VOTENET_OK) return rc;
        } else if (nb <= 16 * 64)
            hipLaunchKernelGGL((fps_bucket_l2_kernel<16, 1>), dim3(b), dim3(1024), 0, st, n, m, inp, (const int *)temp, boxes, sorted, out);
        else if (nb <= 16 * 64 * 2)
            hipLaunchKernelGGL((fps_bucket_l2_kernel<16, 2>), dim3(b), dim3(1024), 0, st, n, m, inp, (const int *)temp, boxes, sorted, out);
        else
            hipLaunchKernelGGL((fps_bucket_l2_kernel<16, 4>), dim3(b), dim3(1024), 0, st, n, m, inp, (const int *)temp, boxes, sorted, out);
    } else {
        const int grid = b < 32 ? b : 32; // tf_sampling_g.cu:204
        hipLaunchKernelGGL((fps_stream_kernel<16>), dim3(grid), dim3(1024), 0, st, b, n, m, inp, temp, out);
    }
    return check_launch("farthest_point_sample");
}

// the reference's own launcher name, C++ linkage, exact signature (tf_sampling.cpp:94)
// The reference's scratch is TensorShape{32, n} whatever the batch (tf_sampling.cpp:115): the bucket path needs about
// 1.1*n floats per scene, so larger batches go through in slices that fit 32*n floats (launches are stream-ordered).
VN_EXPORT void farthestpointsamplingLauncher(int b, int n, int m, const float *inp, float *temp, int *out)
{
    int chunk = b;
    while (chunk > 1 && votenet_fps_temp_floats(chunk, n) > (size_t)32 * (size_t)n) chunk--;
    for (int b0 = 0; b0 < b; b0 += chunk) {
        const int nb = b - b0 < chunk ? b - b0 : chunk;
        votenet_farthest_point_sample(nb, n, m, inp + (size_t)b0 * n * 3, temp, out + (size_t)b0 * m, nullptr);
    }
}
