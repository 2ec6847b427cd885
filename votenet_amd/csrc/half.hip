// half.hip -- the PIECE layout of a set-abstraction level: the grouped MLP on (nearly) only the rows that are not copies (gfx950).
//
// A ball with fewer than K = 64 neighbours repeats its first hit in the remaining slots (tf_grouping_g.cu:26-29), and a repeated slot is
// an identical row through every layer of the grouped MLP: same point, same centre, same input row, same z at every layer; the max-pool
// takes the first occurrence, so a repeated row is never the arg-max, and its gradient is the same B + C z at every layer.  On room
// scenes 39-76 % of the grouped rows are such copies (tools/probe/pts_cnt_stats.py).  Here a level's rows are laid out in PIECES of
// kPiece = 16 rows (mlp_types.h); a ball keeps the pieces that hold at least one real neighbour -- pieces 0 .. kc-1, kc = ceil(pts_cnt / 16):
//     piece q <  G (G = b * m centres):  slots 0..15 of centre q                                   -- always present
//     piece q >= G:                      slots 16 j .. 16 j + 15 of centre c, hc[q] = 4 c + j       -- j = 1 .. kc - 1
// The dropped pieces hold copies of slot 0 only, and slot 0 -- row 0 of the ball's first piece -- stands for them: it carries WEIGHT
// wh[q] = 1 + 16 (4 - kc) wherever a sum runs over the true rows: the BatchNorm statistics of the forward pass and the affine part
// B + C z of every BatchNorm backward (1 for every other piece).  With TOTAL gradients per row (the sum over the true rows a row stands
// for) the backward reductions, the weight gradients and the scatter to the points need no weight.  The number of pieces is kept a
// multiple of 8 (a 128-row GEMM tile) by keeping up to seven all-copy pieces, which is exact.  The result differs from the full layout
// only in the association of those sums.
#include "mlp_types.h"

namespace votenet {

constexpr int PS = kPiece, NP = kBallPieces, TP = kTilePieces;

__device__ __forceinline__ int kept_pieces(int cnt)
{
    const int c = cnt < 1 ? 1 : cnt;
    const int k = (c + PS - 1) / PS;
    return k > NP ? NP : k;
}

// One workgroup: pos[c * (NP - 1) + j - 1] = index (from G) of piece j >= 1 of centre c, or -1; hc[q] = NP * centre + piece number;
// wh[q] = weight of piece q's row 0; nh[0] = number of pieces.  Centres in ascending order (a prefix scan, no atomics): the layout -- and
// with it the order of every sum over the rows -- is the same in every run.
__global__ __launch_bounds__(1024) void half_groups_kernel(int G, const int *__restrict__ pts_cnt, int *__restrict__ pos, int *__restrict__ hc,
                                                           float *__restrict__ wh, int *__restrict__ nh_out, int *nh_host)
{
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < G; c0 += 1024) { // rounds of 1024 consecutive centres: coalesced, ascending
        const int c = c0 + tid;
        const int kc = c < G ? kept_pieces(pts_cnt[c]) : 1;
        const int mine = kc - 1;
        int x = mine; // inclusive scan over the wavefront, then over the 16 wavefronts
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off);
            if (lane >= off) x += t;
        }
        if (lane == 63) s_wave[wv] = x;
        __syncthreads();
        int base = s_carry;
        for (int w = 0; w < wv; w++) base += s_wave[w];
        int p = base + x - mine;
        if (c < G) {
            hc[c] = c * NP;
            for (int j = 1; j < NP; j++) {
                if (j < kc) {
                    hc[G + p] = c * NP + j;
                    pos[c * (NP - 1) + j - 1] = p++;
                } else {
                    pos[c * (NP - 1) + j - 1] = -1;
                }
            }
        }
        __syncthreads();
        if (tid == 1023) s_carry = base + x;
        __syncthreads();
    }
    if (tid == 0) {
        // a GEMM tile is 128 rows = TP pieces: round up with all-copy pieces (the next piece of a ball that dropped some), which is exact
        // (G % TP == 0: there are enough)
        int n2 = s_carry;
        for (int c = 0; c < G && ((G + n2) % TP) != 0; c++)
            for (int j = 1; j < NP && ((G + n2) % TP) != 0; j++)
                if (pos[c * (NP - 1) + j - 1] < 0) {
                    pos[c * (NP - 1) + j - 1] = n2;
                    hc[G + n2] = c * NP + j;
                    n2++;
                }
        s_carry = n2;
        nh_out[0] = G + n2;
        if (nh_host != nullptr) { // pinned host memory, mapped into the device's address space: the host reads it after the event behind
            nh_host[0] = G + n2;  // this kernel (no copy launch)
            __threadfence_system();
        }
    }
    __syncthreads();
    const int nh = G + s_carry;
    for (int q = tid; q < nh; q += 1024) {
        float w = 1.0f;
        if (q < G) {
            int kc = 1;
            for (int j = 1; j < NP; j++) kc += pos[q * (NP - 1) + j - 1] >= 0 ? 1 : 0;
            w = 1.0f + (float)(PS * (NP - kc));
        }
        wh[q] = w;
    }
}

__device__ __forceinline__ double half_shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long half_fixed(float v) { return (long long)((double)v * 4294967296.0); }

// A ball's copies of slot 0 are consecutive rows with one point, and a few points are slot 0 of hundreds of balls: one atomic per RUN of
// equal points inside a wavefront (the run's first lane adds the run's length), not one per row, or those addresses serialise the pass.
__device__ __forceinline__ void half_sort_runs(unsigned prow, bool live, int lane, int &run_start, int &run_len)
{
    const unsigned prev = __shfl_up(prow, 1);
    const bool prev_live = __shfl_up((int)live, 1) != 0;
    const bool starts = live && (lane == 0 || !prev_live || prev != prow);
    const unsigned long long sm = __ballot(starts), lm = __ballot(live);
    // the run this lane belongs to starts at the highest start bit at or below the lane; it ends before the next start / the first dead lane
    const unsigned long long below = sm & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    run_start = below ? 63 - __builtin_clzll(below) : 0;
    const unsigned long long after = (sm | ~lm) & ~((run_start == 63) ? ~0ull : ((1ull << (run_start + 1)) - 1ull));
    const int end = after ? __builtin_ctzll(after) : 64;
    run_len = end - run_start;
}
// votenet_assemble_rows on the piece layout: thread = compact row r = q * PS + s <-> (centre hc[q] / NP, slot PS (hc[q] % NP) + s).
// The per-point counters and the moments run over the TRUE rows exactly as in assemble_rows_kernel: slot k < pts_cnt adds itself, slot 0
// also the 64 - pts_cnt copies.
__global__ __launch_bounds__(256) void assemble_rows_half_kernel(const int *__restrict__ nh_dev, long max_rows, int n, int groups_per_scene, const float *__restrict__ xyz,
                                                                 const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                                 const int *__restrict__ pts_cnt, const int *__restrict__ hc,
                                                                 float4 *__restrict__ geo, long long *__restrict__ cntv,
                                                                 double *__restrict__ moments, int *__restrict__ count)
{
    __shared__ double red[4][9];
    double acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = 0.0;
    const long rows = (long)nh_dev[0] * PS; // the number of pieces is known on the device only when this is enqueued
    for (long r0 = (long)blockIdx.x * 256; r0 < max_rows; r0 += (long)gridDim.x * 256) {
        const long r = r0 + threadIdx.x;
        const bool live = r < rows;
        unsigned my_prow = 0;
        // past the count the buffer holds records of point 0 (a reader that does not know the count -- a caller with the count on the
        // device only sizes everything for the upper bound -- gathers a valid row)
        if (!live && r < max_rows) geo[r] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
        if (live) {
        const int q = (int)(r / PS), s = (int)(r % PS);
        const int code = hc[q];
        const int c = code / NP;
        const int k = (code % NP) * PS + s;
        const int id = idx[(size_t)c * 64 + k];
        const unsigned prow = ((unsigned)c / (unsigned)groups_per_scene) * (unsigned)n + (unsigned)id;
        const float dx = xyz[(size_t)prow * 3 + 0] - new_xyz[(size_t)c * 3 + 0]; // utils.py:55
        const float dy = xyz[(size_t)prow * 3 + 1] - new_xyz[(size_t)c * 3 + 1];
        const float dz = xyz[(size_t)prow * 3 + 2] - new_xyz[(size_t)c * 3 + 2];
        geo[r] = make_float4(dx, dy, dz, __uint_as_float(prow));
        my_prow = prow;
        int cnt = pts_cnt[c];
        if (cnt < 1) cnt = 1;
        if (k < cnt) {
            const long long mult = (k == 0) ? (long long)(64 - cnt + 1) : 1LL;
            if (cntv) {
                unsigned long long *cv = reinterpret_cast<unsigned long long *>(cntv + (size_t)prow * 4);
                atomicAdd(cv + 0, (unsigned long long)mult);
                atomicAdd(cv + 1, (unsigned long long)(mult * half_fixed(dx)));
                atomicAdd(cv + 2, (unsigned long long)(mult * half_fixed(dy)));
                atomicAdd(cv + 3, (unsigned long long)(mult * half_fixed(dz)));
            }
            const double m = (double)mult, x = dx, y = dy, z = dz;
            acc[0] += m * x; acc[1] += m * y; acc[2] += m * z;
            acc[3] += m * x * x; acc[4] += m * x * y; acc[5] += m * x * z; acc[6] += m * y * y; acc[7] += m * y * z; acc[8] += m * z * z;
        }
        }
        if (count) { // the first pass of votenet_half_sort_rows rides here: rows per point, one atomic per run of equal points
            int rs, rl;
            half_sort_runs(my_prow, live, threadIdx.x & 63, rs, rl);
            if (live && rs == (int)(threadIdx.x & 63)) atomicAdd(&count[my_prow], rl);
        }
    }
    if (!moments) return;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        double v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += half_shfl_xor_f64(v, m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        unsafeAtomicAdd(&moments[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// votenet_narrow_rows on the piece layout: thread = compact row; u (8 floats) of the row's slot; the moments run over the TRUE rows
// (slot k < pts_cnt counts once, slot 0 also for the 64 - pts_cnt copies of it).
__global__ __launch_bounds__(256) void narrow_rows_half_kernel(const int *__restrict__ nh_dev, int n, int groups_per_scene, int c,
                                                               const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                               const float *__restrict__ feat, const int *__restrict__ idx,
                                                               const int *__restrict__ pts_cnt, const int *__restrict__ hc,
                                                               float *__restrict__ u8, double *__restrict__ moments)
{
    __shared__ double red[4][44];
    double acc[44]; // m[0..8), then the upper triangle of M row by row
#pragma unroll
    for (int i = 0; i < 44; i++) acc[i] = 0.0;
    const long rows = (long)nh_dev[0] * PS;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const int q = (int)(r / PS), s = (int)(r % PS);
        const int code = hc[q];
        const int ctr = code / NP;
        const int k = (code % NP) * PS + s;
        const int id = idx[(size_t)ctr * 64 + k];
        const size_t prow = (size_t)((unsigned)ctr / (unsigned)groups_per_scene) * n + id;
        float u[8];
        u[0] = xyz[prow * 3 + 0] - new_xyz[(size_t)ctr * 3 + 0]; // utils.py:55
        u[1] = xyz[prow * 3 + 1] - new_xyz[(size_t)ctr * 3 + 1];
        u[2] = xyz[prow * 3 + 2] - new_xyz[(size_t)ctr * 3 + 2];
#pragma unroll
        for (int d = 0; d < 5; d++) u[3 + d] = d < c ? feat[prow * c + d] : 0.0f;
        *reinterpret_cast<float4 *>(u8 + (size_t)r * 8) = make_float4(u[0], u[1], u[2], u[3]);
        *reinterpret_cast<float4 *>(u8 + (size_t)r * 8 + 4) = make_float4(u[4], u[5], u[6], u[7]);
        int cnt = pts_cnt[ctr];
        if (cnt < 1) cnt = 1;
        if (k < cnt) {
            const double mult = (k == 0) ? (double)(64 - cnt + 1) : 1.0;
            int t = 8;
#pragma unroll
            for (int d = 0; d < 8; d++) {
                acc[d] += mult * (double)u[d];
#pragma unroll
                for (int e = d; e < 8; e++) acc[t++] += mult * ((double)u[d] * (double)u[e]);
            }
        }
    }
    if (!moments) return;
#pragma unroll
    for (int i = 0; i < 44; i++) {
        double v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += half_shfl_xor_f64(v, m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 44) {
        const double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const int i = threadIdx.x;
        if (i < 8) {
            unsafeAtomicAdd(&moments[i], v);
        } else { // triangle position -> (d, e), written to both halves of the full matrix
            int d = 0, t = 8;
            while (i >= t + (8 - d)) {
                t += 8 - d;
                d++;
            }
            const int e = d + (i - t);
            unsafeAtomicAdd(&moments[8 + d * 8 + e], v);
            if (e != d) unsafeAtomicAdd(&moments[8 + e * 8 + d], v);
        }
    }
}

// votenet_bn_pool_finalize over pieces: zbest / abest hold every piece's candidate (the max of z where the scale is >= 0, the min where it
// is negative: votenet_mlp_linear_pool_half decides by the sign of gamma, which is the sign of the scale); a centre's pooled value = the
// best of its kept pieces (ties -> the earlier piece: the first occurrence, as the 64-row epilogue decides); arg-max = the slot inside
// the 64-slot ball.
__global__ void bn_pool_finalize_half_kernel(long total, int G, int c, const float *__restrict__ zbest, const int *__restrict__ abest,
                                             const int *__restrict__ pos, const float *__restrict__ scale, const float *__restrict__ shift,
                                             BnRaw raw, int relu, float *__restrict__ out, int *__restrict__ argmax, float *__restrict__ zsel)
{
    // The launcher makes the grid's stride a multiple of c: a thread's channel never changes, so its BatchNorm scale / shift -- from raw
    // sums: two fp64 divisions and a square root -- are computed ONCE, not per element (round 6: they were most of this kernel's time; it
    // sits on the forward chain of every level)
    const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = (int)(e0 % c);
    float s = 0.0f, h = 0.0f;
    if (e0 < total) {
        if (raw.stats) bn_raw_channel(raw, c, ch, e0 < c, s, h);
        else {
            s = scale[ch];
            h = shift[ch];
        }
    }
    for (long e = e0; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / c;
        const float sg = s >= 0.0f ? 1.0f : -1.0f;
        // no load under a branch (round 5): the three piece slots, then all candidates together -- a piece the ball did not keep reads the
        // ball's first piece again and is ignored.  (With the loads under `if (p >= 0)` the compiler put an s_waitcnt vmcnt(0) behind each:
        // up to four dependent memory round trips per element on a kernel that sits on the forward chain of every level.)
        int pj[NP - 1];
#pragma unroll
        for (int j = 1; j < NP; j++) pj[j - 1] = pos[g * (NP - 1) + j - 1];
        float zb[NP];
        int ab[NP];
        zb[0] = zbest[e];
        ab[0] = abest[e];
#pragma unroll
        for (int j = 1; j < NP; j++) {
            const size_t e2 = pj[j - 1] >= 0 ? (size_t)(G + pj[j - 1]) * c + ch : (size_t)e;
            zb[j] = zbest[e2];
            ab[j] = abest[e2];
        }
        float best = sg * zb[0];
        int ibest = ab[0];
#pragma unroll
        for (int j = 1; j < NP; j++) {
            const float b = sg * zb[j];
            if (pj[j - 1] >= 0 && b > best) {
                best = b;
                ibest = j * PS + ab[j];
            }
        }
        const float zr = sg * best;
        float v = zr * s + h;
        if (relu && !(v > 0.0f)) v = 0.0f;
        out[e] = v;
        if (argmax) argmax[e] = ibest;
        if (zsel) zsel[e] = zr;
    }
}

// ---- the first layer's scatter to the points WITHOUT one atomic per row: the compact rows sorted by the point they gather -----------------
// An fp32 atomic costs an L2 channel ~14 cycles per 64-byte line whatever the number of active lanes (tools/probe/src/atomic_scope.hip:
// 300 G adds/s), which made a row-major pass with one atomic per real row atomic-bound at twice the time of its loads (110 -> 79 us at sa2).  The grouping depends on coordinates
// only, so with the geometry (a step ahead, off the chain) the compact rows are bucketed by point: count, scan, fill -> order (16 nh).
// The backward pass then walks CHUNKS of 64 consecutive entries: the rows of a point are consecutive, a thread (one channel) sums them
// in a register and stores the point's row of S once; only a chunk's first and last point can be shared with a neighbour chunk and are
// added with atomics -- 2 per 64 rows instead of ~30.
__global__ __launch_bounds__(256) void half_sort_count_kernel(const int *__restrict__ nh_dev, const float4 *__restrict__ geo,
                                                              int *__restrict__ count)
{
    const long rows = (long)nh_dev[0] * PS;
    const int lane = threadIdx.x & 63;
    for (long r0 = (long)blockIdx.x * 256; r0 < rows; r0 += (long)gridDim.x * 256) {
        const long r = r0 + threadIdx.x;
        const bool live = r < rows;
        const unsigned prow = live ? __float_as_uint(geo[r].w) : 0u;
        int rs, rl;
        half_sort_runs(prow, live, lane, rs, rl);
        if (live && rs == lane) atomicAdd(&count[prow], rl);
    }
}
// one workgroup: count[p] -> the exclusive prefix (the fill's cursor of point p).  A thread owns ceil(npts / 1024) CONSECUTIVE points: its
// own sum, one wave scan, one barrier, then it writes its points' cursors (the first version walked the points 1024 at a time with three
// barriers per trip: 10 us for the 8192 votes of the proposal module, on the train step's critical chain).
__global__ __launch_bounds__(1024) void half_sort_scan_kernel(int npts, int *__restrict__ count)
{
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (npts + 1023) / 1024;
    const int b0 = tid * per < npts ? tid * per : npts, e0 = b0 + per < npts ? b0 + per : npts;
    constexpr int KEEP = 16; // points a thread keeps in registers between its two walks (npts <= 16384: every level of the model)
    int v[KEEP];
    int tot = 0;
    if (per <= KEEP) {
#pragma unroll
        for (int i = 0; i < KEEP; i++) v[i] = b0 + i < e0 ? count[b0 + i] : 0; // independent loads, all in flight
#pragma unroll
        for (int i = 0; i < KEEP; i++) tot += v[i];
    } else {
        for (int p = b0; p < e0; p++) tot += count[p];
    }
    int x = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(x, off);
        if (lane >= off) x += t;
    }
    if (lane == 63) s_wave[wv] = x;
    __syncthreads();
    int run = x - tot;
    for (int w = 0; w < wv; w++) run += s_wave[w];
    if (per <= KEEP) {
#pragma unroll
        for (int i = 0; i < KEEP; i++) {
            if (b0 + i < e0) count[b0 + i] = run;
            run += v[i];
        }
        return;
    }
    for (int p = b0; p < e0; p++) {
        const int c = count[p];
        count[p] = run;
        run += c;
    }
}
__global__ __launch_bounds__(256) void half_sort_fill_kernel(const int *__restrict__ nh_dev, const float4 *__restrict__ geo,
                                                             int *__restrict__ cursor, int *__restrict__ order)
{
    const long rows = (long)nh_dev[0] * PS;
    const int lane = threadIdx.x & 63;
    for (long r0 = (long)blockIdx.x * 256; r0 < rows; r0 += (long)gridDim.x * 256) {
        const long r = r0 + threadIdx.x;
        const bool live = r < rows;
        const unsigned prow = live ? __float_as_uint(geo[r].w) : 0u;
        int rs, rl;
        half_sort_runs(prow, live, lane, rs, rl);
        int base = 0;
        if (live && rs == lane) base = atomicAdd(&cursor[prow], rl);
        base = __shfl(base, rs);
        if (live) order[base + (lane - rs)] = (int)r;
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void group_linear_bwd_sorted_kernel(long rows, const int *__restrict__ order, const float4 *__restrict__ geo,
                                                                      const float *__restrict__ wh, const float *__restrict__ ptab,
                                                                      const float *__restrict__ wx, const float *__restrict__ da,
                                                                      const float *__restrict__ coef, int relu, float *__restrict__ spt,
                                                                      float *__restrict__ dw_xyz, const int *__restrict__ nh_dev)
{
    if (nh_dev != nullptr) { // the count is the device's (rows: the caller's upper bound)
        const long lim = (long)nh_dev[0] * PS;
        rows = lim < rows ? lim : rows;
    }
    constexpr int CPB = 256 / COUT; // chunks per workgroup pass
    constexpr int CH = 64;          // entries per chunk
    __shared__ int s_row[CPB][CH];
    __shared__ float4 s_geo[CPB][CH];
    __shared__ float s_w[CPB][CH];
    __shared__ float red[256][3];
    const int tid = threadIdx.x;
    const int ch = tid % COUT, cl = tid / COUT;
    const float kA = coef[ch], kB = coef[COUT + ch], kC = coef[2 * COUT + ch], kS = coef[3 * COUT + ch], kH = coef[4 * COUT + ch];
    const float wx0 = wx[ch], wx1 = wx[COUT + ch], wx2 = wx[2 * COUT + ch];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const long nchunk = (rows + CH - 1) / CH;
    for (long c0 = (long)blockIdx.x * CPB; c0 < nchunk; c0 += (long)gridDim.x * CPB) {
        lds_barrier(); // the previous pass's records are consumed
        if (tid < CPB * CH) {
            const long e = c0 * CH + tid;
            int r = -1;
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float w = 1.0f;
            if (e < rows) {
                r = order[e];
                g4 = geo[r];
                if (r % PS == 0) w = wh[r / PS]; // row 0 of a piece: a ball's slot 0 stands for its dropped copies
            }
            s_row[tid / CH][tid % CH] = r;
            s_geo[tid / CH][tid % CH] = g4;
            s_w[tid / CH][tid % CH] = w;
        }
        lds_barrier();
        if (c0 + cl >= nchunk) continue;
        unsigned cur = 0xffffffffu;
        float acc = 0.0f;
        bool first = true;
#pragma unroll 1
        for (int e0 = 0; e0 < CH; e0 += 8) {
            float gg[8], zz[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int r = s_row[cl][e0 + u];
                const int rr = r < 0 ? 0 : r;
                gg[u] = da[(size_t)rr * COUT + ch];
                zz[u] = ptab[(size_t)__float_as_uint(s_geo[cl][e0 + u].w) * COUT + ch];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (s_row[cl][e0 + u] < 0) continue; // past the end of the list (the last chunk)
                const float4 g4 = s_geo[cl][e0 + u];
                const unsigned prow = __float_as_uint(g4.w);
                if (prow != cur) { // the rows of a point are consecutive: its sum is complete
                    if (cur != 0xffffffffu) {
                        if (first) unsafeAtomicAdd(&spt[(size_t)cur * COUT + ch], acc); // may continue the previous chunk's last point
                        else spt[(size_t)cur * COUT + ch] = acc;
                        first = false;
                    }
                    cur = prow;
                    acc = 0.0f;
                }
                const float z = assembled_z(zz[u], g4, wx0, wx1, wx2);
                float gq = gg[u];
                if (relu && !(z * kS + kH > 0.0f)) gq = 0.0f;
                const float d = kA * gq + s_w[cl][e0 + u] * (kB + kC * z);
                a0 += g4.x * d;
                a1 += g4.y * d;
                a2 += g4.z * d;
                acc += d;
            }
        }
        if (cur != 0xffffffffu) unsafeAtomicAdd(&spt[(size_t)cur * COUT + ch], acc); // may continue in the next chunk
    }
    red[tid][0] = a0;
    red[tid][1] = a1;
    red[tid][2] = a2;
    __syncthreads();
    if (tid < COUT) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float t = 0.0f;
            for (int q = 0; q < CPB; q++) t += red[q * COUT + tid][d];
            unsafeAtomicAdd(&dw_xyz[(size_t)d * COUT + tid], t);
        }
    }
}

// ---- the first layer's backward DECOMPOSED over the points (round 4) -------------------------------------------------------------
// dz0[r] = A g[r] + B + C z0[r] with g = da0 . relu'(bn0(z0)) couples every row to the layer's BatchNorm-backward sums  (sum g, sum g zhat0),
// which is why the input-gradient GEMM above used to reduce them in its epilogue -- gathering the per-point table P for every element of
// its tile (half of its wave cycles parked, rocprof MfmaUtil 0.12).  But everything the points and the weights receive is LINEAR in dz0:
//     S[p]      = sum_{r -> p} dz0[r]       = A Sg[p] + cnt_p B + C (cnt_p P[p] + V_p Wx)            Sg[p] = sum_{r -> p} g[r]
//     dWx[d]    = sum_r dxyz_d(r) dz0[r]    = A UG[d] + B (sum dxyz_d) + C (sum_p V_p[d] P[p] + (sum dxyz_d dxyz) Wx)     UG[d] = sum_r dxyz_d(r) g[r]
// with cnt_p / V_p (rows per point, the sum of their dxyz: votenet_assemble_rows_half's cntv) and the moments known from the geometry.
// So ONE pass over the rows bucketed by point (this kernel: the former scatter pass without its coefficients) leaves Sg, UG and the
// BatchNorm-backward sums -- it rebuilds z0 for the mask anyway, a point's P row once per run of rows --, the coefficient vector comes
// out of its tail, a pass over the POINTS (assembled_point_grad_kernel) finishes S, and the GEMM above is a plain one.
constexpr int kMaskedSlots = 32;
template <int COUT>
__global__ __launch_bounds__(256) void group_linear_bwd_masked_kernel(long rows, const int *__restrict__ order, const float4 *__restrict__ geo,
                                                                      const float *__restrict__ ptab, const float *__restrict__ wx,
                                                                      const float *__restrict__ da, const float *__restrict__ scale,
                                                                      const float *__restrict__ shift, const float *__restrict__ mean,
                                                                      const float *__restrict__ var, float eps, int relu,
                                                                      float *__restrict__ sg, float *__restrict__ ug, double *__restrict__ sums,
                                                                      double *__restrict__ part /* kMaskedSlots x 5 x COUT, zeroed */,
                                                                      const int *__restrict__ nh_dev, CoefTail tail)
{
    if (nh_dev != nullptr) { // the count is the device's (rows: the caller's upper bound)
        const long lim = (long)nh_dev[0] * PS;
        rows = lim < rows ? lim : rows;
    }
    constexpr int CPB = 256 / COUT; // chunks per workgroup pass
    constexpr int CH = 64;          // entries per chunk
    constexpr int GB = 8;           // entries whose two loads are in flight together (16 measured slower: 54.6 -> 59.9 us at sa2)
    __shared__ int s_row[CPB][CH];
    __shared__ float4 s_geo[CPB][CH];
    __shared__ float red[256][5];
    const int tid = threadIdx.x;
    const int ch = tid % COUT, cl = tid / COUT;
    const float kS = scale[ch], kH = shift[ch], kM = mean[ch], kR = 1.0f / sqrtf(var[ch] + eps);
    const float wx0 = wx[ch], wx1 = wx[COUT + ch], wx2 = wx[2 * COUT + ch];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, s1 = 0.f, s2 = 0.f;
    const long nchunk = (rows + CH - 1) / CH;
    for (long c0 = (long)blockIdx.x * CPB; c0 < nchunk; c0 += (long)gridDim.x * CPB) {
        lds_barrier(); // the previous pass's records are consumed
        if (tid < CPB * CH) {
            const long e = c0 * CH + tid;
            int r = -1;
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < rows) {
                r = order[e];
                g4 = geo[r];
            }
            s_row[tid / CH][tid % CH] = r;
            s_geo[tid / CH][tid % CH] = g4;
        }
        lds_barrier();
        if (c0 + cl >= nchunk) continue;
        unsigned cur = 0xffffffffu;
        float acc = 0.0f;
        bool first = true;
#pragma unroll 1
        for (int e0 = 0; e0 < CH; e0 += GB) {
            float gg[GB], zz[GB];
#pragma unroll
            for (int u = 0; u < GB; u++) {
                const int r = s_row[cl][e0 + u];
                const int rr = r < 0 ? 0 : r;
                gg[u] = da[(size_t)rr * COUT + ch];
                zz[u] = ptab[(size_t)__float_as_uint(s_geo[cl][e0 + u].w) * COUT + ch];
            }
#pragma unroll
            for (int u = 0; u < GB; u++) {
                if (s_row[cl][e0 + u] < 0) continue; // past the end of the list (the last chunk)
                const float4 g4 = s_geo[cl][e0 + u];
                const unsigned prow = __float_as_uint(g4.w);
                if (prow != cur) { // the rows of a point are consecutive: its sum is complete
                    if (cur != 0xffffffffu) {
                        if (first) unsafeAtomicAdd(&sg[(size_t)cur * COUT + ch], acc); // may continue the previous chunk's last point
                        else sg[(size_t)cur * COUT + ch] = acc;
                        first = false;
                    }
                    cur = prow;
                    acc = 0.0f;
                }
                const float z = assembled_z(zz[u], g4, wx0, wx1, wx2);
                float gq = gg[u];
                if (relu && !(z * kS + kH > 0.0f)) gq = 0.0f;
                a0 += g4.x * gq;
                a1 += g4.y * gq;
                a2 += g4.z * gq;
                s1 += gq;
                s2 += gq * ((z - kM) * kR);
                acc += gq;
            }
        }
        if (cur != 0xffffffffu) unsafeAtomicAdd(&sg[(size_t)cur * COUT + ch], acc); // may continue in the next chunk
    }
    red[tid][0] = a0;
    red[tid][1] = a1;
    red[tid][2] = a2;
    red[tid][3] = s1;
    red[tid][4] = s2;
    __syncthreads();
    if (tid < COUT) {
        float t[5];
#pragma unroll
        for (int d = 0; d < 5; d++) {
            t[d] = 0.0f;
            for (int q = 0; q < CPB; q++) t[d] += red[q * COUT + tid][d];
        }
        // every workgroup ends in 5 atomics per channel: on ONE set of addresses 2048 workgroups queue up behind each other at the L2
        // (~7 ns per atomic and address); kMaskedSlots sets share the queue, the last workgroup adds them up
        double *ps = part + (size_t)(blockIdx.x % kMaskedSlots) * 5 * COUT;
#pragma unroll
        for (int d = 0; d < 5; d++) unsafeAtomicAdd(&ps[(size_t)d * COUT + tid], (double)t[d]);
    }
    // the tail (as coef_tail, common.h): the last workgroup to take a ticket folds the slots into ug / sums and computes the layer's
    // coefficient vector
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const double invn = 1.0 / (double)tail.rows;
    for (int col = threadIdx.x; col < COUT; col += blockDim.x) {
        double q[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int sl = 0; sl < kMaskedSlots; sl++)
#pragma unroll
            for (int d = 0; d < 5; d++)
                q[d] += __hip_atomic_load(&part[((size_t)sl * 5 + d) * COUT + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int d = 0; d < 3; d++) ug[(size_t)d * COUT + col] = (float)q[d];
        sums[col] = q[3];
        sums[COUT + col] = q[4];
        const float inv = 1.0f / sqrtf(var[col] + eps);
        const float m1 = (float)(q[3] * invn), m2 = (float)(q[4] * invn);
        const float cA = tail.gamma[col] * inv;
        const float cC = -cA * inv * m2;
        tail.coef[col] = cA;
        tail.coef[COUT + col] = -cA * m1 - cC * mean[col];
        tail.coef[2 * COUT + col] = cC;
        tail.coef[3 * COUT + col] = scale[col];
        tail.coef[4 * COUT + col] = shift[col];
        if (tail.dgamma) tail.dgamma[col] += (float)q[4];
        if (tail.dbeta) tail.dbeta[col] += (float)q[3];
    }
    if (threadIdx.x == 0) *tail.ticket = 0u;
}

// S[p, c] = A Sg[p, c] + cnt_p B + C (cnt_p P[p, c] + V_p . Wx[:, c]) in place over Sg (points x COUT), and vp[d, c] += sum_p V_p[d] P[p, c]
// (what the coordinate rows of the weight gradient still need).  Thread = (point lane, channel); cntv as votenet_assemble_rows leaves it
// (count, then the dxyz sums in fixed point 2^-32).
template <int COUT>
__global__ __launch_bounds__(256) void assembled_point_grad_kernel(long npts, const float *__restrict__ ptab, const long long *__restrict__ cntv,
                                                                   const float *__restrict__ wx, const float *__restrict__ coef,
                                                                   float *__restrict__ s, float *__restrict__ vp /* kMaskedSlots x 3 x COUT, zeroed */)
{
    constexpr int PPB = 256 / COUT;
    constexpr int U = 4; // points in flight per thread: the pass is a latency chain otherwise (33 us for 26 MB with one)
    __shared__ float red[256][3];
    const int tid = threadIdx.x, ch = tid % COUT, pl = tid / COUT;
    const float kA = coef[ch], kB = coef[COUT + ch], kC = coef[2 * COUT + ch];
    const float wx0 = wx[ch], wx1 = wx[COUT + ch], wx2 = wx[2 * COUT + ch];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    const long stride = (long)gridDim.x * PPB;
    for (long p0 = (long)blockIdx.x * PPB + pl; p0 < npts; p0 += U * stride) {
        long long c4[U][4];
        float pv[U], sv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long p = p0 + u * stride < npts ? p0 + u * stride : p0; // (clamped: skipped below)
            const long long *cv = cntv + (size_t)p * 4;
            c4[u][0] = cv[0];
            c4[u][1] = cv[1];
            c4[u][2] = cv[2];
            c4[u][3] = cv[3];
            pv[u] = ptab[(size_t)p * COUT + ch];
            sv[u] = s[(size_t)p * COUT + ch];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long p = p0 + u * stride;
            if (p >= npts || c4[u][0] == 0) continue; // past the end / a point no ball contains: Sg is zero there and stays
            const float fx = (float)((double)c4[u][1] * (1.0 / 4294967296.0)), fy = (float)((double)c4[u][2] * (1.0 / 4294967296.0)),
                        fz = (float)((double)c4[u][3] * (1.0 / 4294967296.0));
            const float fc = (float)c4[u][0];
            const float zsum = fc * pv[u] + (fx * wx0 + fy * wx1 + fz * wx2);
            s[(size_t)p * COUT + ch] = kA * sv[u] + fc * kB + kC * zsum;
            v0 += fx * pv[u];
            v1 += fy * pv[u];
            v2 += fz * pv[u];
        }
    }
    red[tid][0] = v0;
    red[tid][1] = v1;
    red[tid][2] = v2;
    __syncthreads();
    if (tid < COUT) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float t = 0.0f;
            for (int q = 0; q < PPB; q++) t += red[q * COUT + tid][d];
            // kMaskedSlots address sets share the workgroups' atomics (on one set they queue up at the L2: 33 -> 54 us with 2048 workgroups)
            unsafeAtomicAdd(&vp[((size_t)(blockIdx.x % kMaskedSlots) * 3 + d) * COUT + tid], t);
        }
    }
}

// dWx[d, c] += A[c] UG[d, c] + B[c] (sum dxyz_d) + C[c] (VP[d, c] + sum_e (sum dxyz_d dxyz_e) Wx[e, c]); moments as assemble_rows leaves them
// (sum dx, dy, dz, then xx, xy, xz, yy, yz, zz).  One workgroup.
__global__ __launch_bounds__(256) void assembled_wx_finish_kernel(int cout, const float *__restrict__ coef, const float *__restrict__ ug,
                                                                  const float *__restrict__ vp, int nparts, const double *__restrict__ mom,
                                                                  const float *__restrict__ wx, float *__restrict__ dw_xyz)
{
    for (int c = threadIdx.x; c < cout; c += blockDim.x) {
        double vs[3] = {0.0, 0.0, 0.0}; // the slots of votenet_assembled_point_grad
        for (int q = 0; q < nparts; q++)
#pragma unroll
            for (int d = 0; d < 3; d++) vs[d] += (double)vp[((size_t)q * 3 + d) * cout + c];
        const double kA = coef[c], kB = coef[cout + c], kC = coef[2 * cout + c];
        const double w0 = wx[c], w1 = wx[cout + c], w2 = wx[2 * cout + c];
        const double m[3][3] = {{mom[3], mom[4], mom[5]}, {mom[4], mom[6], mom[7]}, {mom[5], mom[7], mom[8]}};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double mw = m[d][0] * w0 + m[d][1] * w1 + m[d][2] * w2;
            dw_xyz[(size_t)d * cout + c] += (float)(kA * ug[(size_t)d * cout + c] + kB * mom[d] + kC * (vs[d] + mw));
        }
    }
}

// T[c, ch] = MINUS the sum of the total gradients dz0 = A g' + w (B + C z0) over the compact rows of centre c (its first piece at c, the others
// through pos): what the centre's coordinates receive through dxyz = xyz[idx] - new_xyz.  Thread = (centre, channel).
template <int COUT>
__global__ __launch_bounds__(256) void half_centre_sums_kernel(long G, const int *__restrict__ pos, const float4 *__restrict__ geo,
                                                               const float *__restrict__ wh, const float *__restrict__ ptab,
                                                               const float *__restrict__ wx, const float *__restrict__ da,
                                                               const float *__restrict__ coef, int relu, float *__restrict__ T)
{
    constexpr int CPB = 256 / COUT;
    const int tid = threadIdx.x, ch = tid % COUT, cl = tid / COUT;
    const float kA = coef[ch], kB = coef[COUT + ch], kC = coef[2 * COUT + ch], kS = coef[3 * COUT + ch], kH = coef[4 * COUT + ch];
    const float wx0 = wx[ch], wx1 = wx[COUT + ch], wx2 = wx[2 * COUT + ch];
    for (long c = (long)blockIdx.x * CPB + cl; c < G; c += (long)gridDim.x * CPB) {
        float acc = 0.0f;
#pragma unroll 1
        for (int j = 0; j < NP; j++) {
            long q = c;
            if (j > 0) {
                const int p = pos[c * (NP - 1) + j - 1];
                if (p < 0) continue;
                q = G + p;
            }
            const float w0 = wh[q];
            // eight rows' loads in flight (the record, then the gather it addresses, and the gradient row): with four the pass was a
            // latency chain of ~8 round trips per centre (30 us for 2048 centres)
#pragma unroll
            for (int s0 = 0; s0 < PS; s0 += 8) {
                float4 g8[8];
                float p8[8], d8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) g8[u] = geo[q * PS + s0 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    p8[u] = ptab[(size_t)__float_as_uint(g8[u].w) * COUT + ch];
                    d8[u] = da[(size_t)(q * PS + s0 + u) * COUT + ch];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float z = assembled_z(p8[u], g8[u], wx0, wx1, wx2);
                    float gq = d8[u];
                    if (relu && !(z * kS + kH > 0.0f)) gq = 0.0f;
                    acc += kA * gq + ((s0 + u) == 0 ? w0 : 1.0f) * (kB + kC * z);
                }
            }
        }
        T[(size_t)c * COUT + ch] = -acc; // dxyz = xyz[idx] - new_xyz: the centre receives minus the sum
    }
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_half_groups(int G, const int *pts_cnt, int *pos, int *hc, float *wh, int *nh_out, int *nh_host, void *stream)
{
    VN_REQUIRE(G > 0 && G % TP == 0, "half_groups expects a positive number of centres, a multiple of %d", TP);
    VN_REQUIRE(pts_cnt && pos && hc && wh && nh_out, "half_groups: null buffer");
    hipLaunchKernelGGL(half_groups_kernel, dim3(1), dim3(1024), 0, as_stream(stream), G, pts_cnt, pos, hc, wh, nh_out, nh_host);
    return check_launch("half_groups");
}
extern "C" int votenet_half_piece_rows(void) { return PS; }

extern "C" int votenet_assemble_rows_half(int b, int n, int m, const int *nh, const float *xyz, const float *new_xyz, const int *idx,
                                          const int *pts_cnt, const int *hc, float *geo, long long *cntv, double *moments, int *count,
                                          void *stream)
{
    VN_REQUIRE(b > 0 && n > 0 && m > 0, "assemble_rows_half: bad shape");
    const long max_rows = 64L * b * m;
    VN_REQUIRE(max_rows < (1L << 31) && (long)b * n < (1L << 31), "assemble_rows_half: row and point counts must be below 2^31");
    VN_REQUIRE(nh && xyz && new_xyz && idx && pts_cnt && hc && geo, "assemble_rows_half: null buffer");
    VN_REQUIRE((uintptr_t)geo % 16 == 0 && (!cntv || (uintptr_t)cntv % 16 == 0), "assemble_rows_half: geo / cntv must be 16-byte aligned");
    long gx = (max_rows / 2 + 256 * 8 - 1) / (256 * 8); // sized for a typical fill; grid-stride covers the rest
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(assemble_rows_half_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), nh, max_rows, n, m, xyz, new_xyz, idx, pts_cnt, hc,
                       reinterpret_cast<float4 *>(geo), cntv, moments, count);
    return check_launch("assemble_rows_half");
}

extern "C" int votenet_narrow_rows_half(int b, int n, int m, int c, const int *nh, const float *xyz, const float *new_xyz, const float *feat,
                                       const int *idx, const int *pts_cnt, const int *hc, float *u8, double *moments, void *stream)
{
    VN_REQUIRE(b > 0 && n > 0 && m > 0 && c >= 0 && c <= 5, "narrow_rows_half expects 0 <= c <= 5 feature channels (3 + c <= 8)");
    const long max_rows = 64L * b * m;
    VN_REQUIRE(max_rows < (1L << 31), "narrow_rows_half: b*m*64 must be below 2^31");
    VN_REQUIRE(nh && xyz && new_xyz && idx && pts_cnt && hc && u8 && (c == 0 || feat), "narrow_rows_half: null buffer");
    VN_REQUIRE((uintptr_t)u8 % 16 == 0, "narrow_rows_half: u8 must be 16-byte aligned");
    long gx = (max_rows / 2 + 256 * 16 - 1) / (256 * 16);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(narrow_rows_half_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), nh, n, m, c, xyz, new_xyz, feat, idx,
                       pts_cnt, hc, u8, moments);
    return check_launch("narrow_rows_half");
}

extern "C" int votenet_bn_pool_finalize_half(long G, int c, const float *zbest, const int *abest, const int *pos, const float *scale,
                                             const float *shift, const votenet_bn_raw *bn, int relu, float *out, int *argmax, float *zsel,
                                             void *stream)
{
    VN_REQUIRE(G >= 0 && c > 0, "bn_pool_finalize_half expects G >= 0, c > 0");
    if (G == 0) return VOTENET_OK;
    const BnRaw raw = to_raw(bn);
    VN_REQUIRE(zbest && abest && pos && out && ((scale && shift) || (raw.stats && raw.gamma && raw.beta && raw.rows > 0)),
               "bn_pool_finalize_half: null buffer");
    VN_REQUIRE(256 % c == 0, "bn_pool_finalize_half: c must divide 256 (a thread keeps its channel over the grid's stride), got %d", c);
    long grid = (G * c + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(bn_pool_finalize_half_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), G * c, (int)G, c, zbest, abest, pos,
                       scale, shift, raw, relu, out, argmax, zsel);
    return check_launch("bn_pool_finalize_half");
}

// order (64 G ints; 16 nh[0] written) = the level's compact rows bucketed by the point they gather (geo[r].w); work: npts ints.
// Coordinates only: runs with the geometry.  The order inside a bucket is whatever the fill's atomics decide.
extern "C" int votenet_half_sort_rows(int npts, int G, const int *nh, const float *geo, int *work, int counted, int *order, void *stream)
{
    VN_REQUIRE(npts > 0 && G > 0 && nh && geo && work && order, "half_sort_rows: bad arguments");
    VN_REQUIRE((uintptr_t)geo % 16 == 0, "half_sort_rows: geo must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const long max_rows = 64L * G;
    long gx = (max_rows / 2 + 256 * 4 - 1) / (256 * 4);
    if (gx > 2048) gx = 2048;
    const float4 *g4 = reinterpret_cast<const float4 *>(geo);
    if (!counted) { // (counted: work holds the rows per point already -- votenet_assemble_rows_half's count argument)
        if (hipMemsetAsync(work, 0, (size_t)npts * sizeof(int), st) != hipSuccess) return set_error(VOTENET_E_HIP, "half_sort_rows: memset failed");
        hipLaunchKernelGGL(half_sort_count_kernel, dim3((unsigned)gx), dim3(256), 0, st, nh, g4, work);
    }
    hipLaunchKernelGGL(half_sort_scan_kernel, dim3(1), dim3(1024), 0, st, npts, work);
    hipLaunchKernelGGL(half_sort_fill_kernel, dim3((unsigned)gx), dim3(256), 0, st, nh, g4, work, order);
    return check_launch("half_sort_rows");
}

// votenet_group_linear_backward_assembled on the piece layout, over the rows bucketed by point: da holds TOTAL gradients per compact row, so
//   dz_total = A g' + w (B + C z),  w = wh[q] on row 0 of piece q (the rows it stands for share z, hence the mask of g');
// S written point by point (atomics only where a chunk of 64 entries shares a point with its neighbour).  s_points pre-zeroed (points
// nobody gathers keep their zeros).
extern "C" int votenet_group_linear_backward_sorted(long nh, int cout, const int *order, const float *geo, const float *wh, const float *P,
                                                    const float *wx, const float *da, const float *coef, int relu, float *s_points,
                                                    float *dw_xyz, const int *nh_dev, void *stream)
{
    VN_REQUIRE(nh > 0 && (cout == 64 || cout == 128 || cout == 256), "group_linear_backward_sorted expects nh > 0, cout in {64, 128, 256}");
    VN_REQUIRE(order && geo && wh && P && wx && da && coef && s_points && dw_xyz, "group_linear_backward_sorted: null buffer");
    VN_REQUIRE((uintptr_t)geo % 16 == 0, "group_linear_backward_sorted: geo must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const long rows = nh * PS, nchunk = (rows + 63) / 64;
    const int cpb = 256 / cout;
    long gx = (nchunk + cpb - 1) / cpb;
    if (gx > 8192) gx = 8192;
    const float4 *g4 = reinterpret_cast<const float4 *>(geo);
    if (cout == 128)
        hipLaunchKernelGGL(group_linear_bwd_sorted_kernel<128>, dim3((unsigned)gx), dim3(256), 0, st, rows, order, g4, wh, P, wx, da, coef, relu,
                           s_points, dw_xyz, nh_dev);
    else if (cout == 64)
        hipLaunchKernelGGL(group_linear_bwd_sorted_kernel<64>, dim3((unsigned)gx), dim3(256), 0, st, rows, order, g4, wh, P, wx, da, coef, relu,
                           s_points, dw_xyz, nh_dev);
    else
        hipLaunchKernelGGL(group_linear_bwd_sorted_kernel<256>, dim3((unsigned)gx), dim3(256), 0, st, rows, order, g4, wh, P, wx, da, coef, relu,
                           s_points, dw_xyz, nh_dev);
    return check_launch("group_linear_backward_sorted");
}

// The first layer's backward decomposed over the points, on the piece layout (see group_linear_bwd_masked_kernel): da = the TOTAL
// gradients per compact row the plain input-gradient GEMM (votenet_mlp_dgrad_bn_half) stored.
//   votenet_group_linear_backward_masked: one pass over the rows bucketed by point -> sg (points x cout, pre-zeroed) = the scatter of the
//     MASKED gradient, ug (3 x cout, pre-zeroed) += sum_r dxyz(r)^T g[r], sums (2 cout doubles, pre-zeroed) += (sum g, sum g zhat0): the
//     BatchNorm-backward sums of the first layer; tail: its coefficient vector from the completed sums (votenet_coef_tail).
//   votenet_assembled_point_grad: sg -> S in place, vp (3 x cout, pre-zeroed) += sum_p V_p^T P[p].
//   votenet_assembled_wx_finish: dw_xyz += the coordinate rows of the first layer's weight gradient.
extern "C" int votenet_group_linear_backward_masked(long nh, int cout, const int *order, const float *geo, const float *P, const float *wx,
                                                    const float *da, const float *scale, const float *shift, const float *mean,
                                                    const float *var, float eps, int relu, float *sg, float *ug, double *sums, double *part,
                                                    const votenet_coef_tail *tail, const int *nh_dev, void *stream)
{
    VN_REQUIRE(nh > 0 && (cout == 64 || cout == 128 || cout == 256), "group_linear_backward_masked expects nh > 0, cout in {64, 128, 256}");
    VN_REQUIRE(order && geo && P && wx && da && scale && shift && mean && var && sg && ug && sums && part, "group_linear_backward_masked: null buffer");
    VN_REQUIRE(tail != nullptr, "group_linear_backward_masked needs the coefficient tail (votenet_coef_tail): its last workgroup folds the partial sums");
    VN_REQUIRE((uintptr_t)geo % 16 == 0, "group_linear_backward_masked: geo must be 16-byte aligned");
    VN_REQUIRE(!tail || (tail->ticket && tail->gamma && tail->coef && tail->rows > 0), "group_linear_backward_masked: incomplete coefficient tail");
    hipStream_t st = as_stream(stream);
    const long rows = nh * PS, nchunk = (rows + 63) / 64;
    const int cpb = 256 / cout;
    long gx = (nchunk + cpb - 1) / cpb;
    if (gx > 2048) gx = 2048;
    const float4 *g4 = reinterpret_cast<const float4 *>(geo);
    const CoefTail t = to_tail(tail);
#define GLBM(C) hipLaunchKernelGGL(group_linear_bwd_masked_kernel<C>, dim3((unsigned)gx), dim3(256), 0, st, rows, order, g4, P, wx, da, scale, \
                                   shift, mean, var, eps, relu, sg, ug, sums, part, nh_dev, t)
    if (cout == 128) GLBM(128);
    else if (cout == 64) GLBM(64);
    else GLBM(256);
#undef GLBM
    return check_launch("group_linear_backward_masked");
}

static long point_grad_grid(long npts, int cout)
{
    const long ppb = (cout > 0 && cout <= 256) ? 256 / cout : 1;
    const long gx = (npts + ppb * 4 - 1) / (ppb * 4); // four points in flight per thread
    return gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
}
extern "C" int votenet_group_linear_backward_masked_slots(void) { return kMaskedSlots; }

extern "C" int votenet_assembled_point_grad(long npts, int cout, const float *P, const long long *cntv, const float *wx, const float *coef,
                                            float *s, float *vp, void *stream)
{
    VN_REQUIRE(npts > 0 && (cout == 64 || cout == 128 || cout == 256), "assembled_point_grad expects npts > 0, cout in {64, 128, 256}");
    VN_REQUIRE(P && cntv && wx && coef && s && vp && (uintptr_t)cntv % 16 == 0, "assembled_point_grad: null / unaligned buffer");
    hipStream_t st = as_stream(stream);
    const int ppb = 256 / cout;
    (void)ppb;
    const long gx = point_grad_grid(npts, cout);
    if (cout == 128) hipLaunchKernelGGL(assembled_point_grad_kernel<128>, dim3((unsigned)gx), dim3(256), 0, st, npts, P, cntv, wx, coef, s, vp);
    else if (cout == 64) hipLaunchKernelGGL(assembled_point_grad_kernel<64>, dim3((unsigned)gx), dim3(256), 0, st, npts, P, cntv, wx, coef, s, vp);
    else hipLaunchKernelGGL(assembled_point_grad_kernel<256>, dim3((unsigned)gx), dim3(256), 0, st, npts, P, cntv, wx, coef, s, vp);
    return check_launch("assembled_point_grad");
}

extern "C" int votenet_assembled_wx_finish(int cout, const float *coef, const float *ug, const float *vp, int nparts, const double *moments,
                                           const float *wx, float *dw_xyz, void *stream)
{
    VN_REQUIRE(cout > 0 && nparts > 0 && coef && ug && vp && moments && wx && dw_xyz, "assembled_wx_finish: bad arguments");
    hipLaunchKernelGGL(assembled_wx_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), cout, coef, ug, vp, nparts, moments, wx, dw_xyz);
    return check_launch("assembled_wx_finish");
}

extern "C" int votenet_half_centre_sums(long G, int cout, const int *pos, const float *geo, const float *wh, const float *P, const float *wx,
                                        const float *da, const float *coef, int relu, float *T, void *stream)
{
    VN_REQUIRE(G > 0 && (cout == 64 || cout == 128 || cout == 256), "half_centre_sums expects G > 0, cout in {64, 128, 256}");
    VN_REQUIRE(pos && geo && wh && P && wx && da && coef && T && (uintptr_t)geo % 16 == 0, "half_centre_sums: null / unaligned buffer");
    hipStream_t st = as_stream(stream);
    const int cpb = 256 / cout;
    long gx = (G + cpb - 1) / cpb;
    if (gx > 4096) gx = 4096;
    const float4 *g4 = reinterpret_cast<const float4 *>(geo);
    if (cout == 128)
        hipLaunchKernelGGL(half_centre_sums_kernel<128>, dim3((unsigned)gx), dim3(256), 0, st, G, pos, g4, wh, P, wx, da, coef, relu, T);
    else if (cout == 64)
        hipLaunchKernelGGL(half_centre_sums_kernel<64>, dim3((unsigned)gx), dim3(256), 0, st, G, pos, g4, wh, P, wx, da, coef, relu, T);
    else
        hipLaunchKernelGGL(half_centre_sums_kernel<256>, dim3((unsigned)gx), dim3(256), 0, st, G, pos, g4, wh, P, wx, da, coef, relu, T);
    return check_launch("half_centre_sums");
}
