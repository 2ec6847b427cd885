// pool_bwd.hip -- backward of the LAST layer of an SA chain (conv + BatchNorm + ReLU + max over the nsample rows of a group)
// in "Gram form".  With x = the layer's input activation (rows x cin), z = x W + b, and the folded BatchNorm backward
//     dz[r,c] = A[c] g'[r,c] + B[c] + C[c] z[r,c]
// the pooled upstream gradient g' has ONE non-zero per (group, channel) -- the arg-max row -- while B + C z is dense only
// because BatchNorm couples the rows.  Substituting z = x W + b turns the dense part into products with cin x cin matrices:
//     da  = dz W^T   = x (W diag(C) W^T) + (B + C.b) W^T                    + scatter of A g' W^T rows
//     dW  = x^T dz   = ((x^T x) W) . C  + (sum_r x)^T (B + C.b)             + gather of x rows by arg-max
// The two big GEMMs of the layer shrink from (rows x cout x cin) to (rows x cin x cin) -- half the flops for VoteNet's
// 128 -> 256 layers -- z of the layer is never read again (training does not store it any more), and x^T x depends on the
// forward pass only: it belongs to the weight-gradient stream, off the critical chain.  The sparse parts touch cout values
// per group.
#include "mlp_types.h"
#include <mutex>
#include <set>

namespace votenet {

bool wgrad_fast_launch(int mode, const MlpIn &d, long rows, int cin, int cout, const float *dz, const BnSrc &bs, int bsrc, float *dw,
                       hipStream_t st, float *scratch); // mlp_wgrad_fast.hip
void wgrad_reduce(int nslice, long pstride, long e0, long e1, const float *part, float *dw, hipStream_t st); // mlp_bwd.hip

static inline int pb_grid(long total, int block, int cap)
{
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

// sums = [sum g', sum g' zhat] of the pooled gradient: g' lives at the arg-max rows, whose raw z is zsel
__global__ __launch_bounds__(256) void bn_bwd_reduce_zsel_kernel(long groups, int c, const float *__restrict__ gout,
                                                                 const float *__restrict__ zsel, const float *__restrict__ scale,
                                                                 const float *__restrict__ shift, const float *__restrict__ mean,
                                                                 const float *__restrict__ var, float eps, int relu,
                                                                 double *__restrict__ sums, CoefTail tail)
{
    __shared__ float sh1[4][64], sh2[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cx;
    float s1 = 0, s2 = 0;
    if (col < c) {
        const float sc = scale[col], sf = shift[col], mu = mean[col], inv = 1.0f / sqrtf(var[col] + eps);
        // eight groups' loads in flight per trip: the pass sits on the backward chain of every SA level and was one loaded-memory latency
        // per 4 groups and workgroup (22 us for 8 MB)
        const long stride = (long)gridDim.x * 4;
        long g = (long)blockIdx.x * 4 + ry;
        for (; g + 7 * stride < groups; g += 8 * stride) {
            float zz[8], gg[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                zz[u] = zsel[(size_t)(g + u * stride) * c + col];
                gg[u] = gout[(size_t)(g + u * stride) * c + col];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (relu && !(zz[u] * sc + sf > 0.0f)) gg[u] = 0.0f;
                s1 += gg[u];
                s2 += gg[u] * ((zz[u] - mu) * inv);
            }
        }
        for (; g < groups; g += stride) {
            const float zz = zsel[(size_t)g * c + col];
            float gg = gout[(size_t)g * c + col];
            if (relu && !(zz * sc + sf > 0.0f)) gg = 0.0f;
            s1 += gg;
            s2 += gg * ((zz - mu) * inv);
        }
    }
    sh1[ry][cx] = s1;
    sh2[ry][cx] = s2;
    __syncthreads();
    if (ry == 0 && col < c) {
        const float t1 = (sh1[0][cx] + sh1[1][cx]) + (sh1[2][cx] + sh1[3][cx]);
        const float t2 = (sh2[0][cx] + sh2[1][cx]) + (sh2[2][cx] + sh2[3][cx]);
        unsafeAtomicAdd(&sums[col], (double)t1);
        unsafeAtomicAdd(&sums[c + col], (double)t2);
    }
    coef_tail(tail, gridDim.x * gridDim.y, c, sums, scale, shift, mean, var, eps);
}

// mmat (cin x cin) = W diag(C) W^T ; cvec (cin) = (B + C.b) W^T.  W is cin x cout row-major; coef = [A|B|C|S|H].
// A 16 x 16 output tile per workgroup; its two 16 x cout panels of W go to LDS in ONE round of loads (this kernel sits on the
// step's critical chain and runs beside the weight-gradient GEMMs: every dependent load phase costs a loaded-memory latency).
template <int COUT>
__global__ __launch_bounds__(256) void pool_dgrad_prepare_kernel(int cin, const float *__restrict__ w, const float *__restrict__ bias,
                                                                 const float *__restrict__ coef, float *__restrict__ mmat,
                                                                 float *__restrict__ cvec, unsigned *__restrict__ image,
                                                                 float *__restrict__ h2_ascale = nullptr, float *__restrict__ h2_unscale = nullptr)
{
    constexpr int LD = COUT + 4;
    __shared__ __attribute__((aligned(16))) float Wj[16][LD], Wk[16][LD];
    __shared__ float Cs[COUT], Ds[COUT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int j0 = blockIdx.y * 16, k0 = blockIdx.x * 16;
    constexpr int Q = COUT / 4;
    for (int e = tid; e < 16 * Q; e += 256) {
        const int r = e / Q, q = e % Q;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(&Wj[r][q * 4]) = j0 + r < cin ? *reinterpret_cast<const float4 *>(w + (size_t)(j0 + r) * COUT + q * 4) : z4;
        *reinterpret_cast<float4 *>(&Wk[r][q * 4]) = k0 + r < cin ? *reinterpret_cast<const float4 *>(w + (size_t)(k0 + r) * COUT + q * 4) : z4;
    }
    for (int c = tid; c < COUT; c += 256) {
        const float Cc = coef[2 * COUT + c];
        Cs[c] = Cc;
        Ds[c] = coef[COUT + c] + Cc * (bias ? bias[c] : 0.0f);
    }
    __syncthreads();
    float acc = 0.0f, accv = 0.0f;
#pragma unroll 8
    for (int c = 0; c < COUT; c++) {
        const float wk = Wk[tx][c];
        acc += Wj[ty][c] * Cs[c] * wk;
        accv += Ds[c] * wk;
    }
    if (j0 + ty < cin && k0 + tx < cin) mmat[(size_t)(j0 + ty) * cin + k0 + tx] = acc;
    if (blockIdx.y == 0 && ty == 0 && k0 + tx < cin) cvec[k0 + tx] = accv;
    if (image != nullptr && h2_ascale != nullptr) {
        // Round 6: the image as TWO fp16 pieces (mlp_types.h: split2).  The matrix is gradient-sized (C is a BatchNorm-backward
        // coefficient), far below fp16's range: it travels scaled by powers of two chosen from what THIS workgroup holds --
        //   |M[j][k]| <= max|C| * |W[j,:]| * |W[k,:]|   (Cauchy-Schwarz)
        // so M'[j][k] = M[j][k] * S / (rn[j] rn[k]) with rn[r] = the power of two >= |W[r,:]| and S = 2^13 / (the power of two >= max|C|)
        // is below 2^13 everywhere.  The GEMM multiplies input channel j by 16 rn[j] while it stages it (h2_ascale, folded into the
        // BatchNorm table) and scales output column k back by rn[k] / (16 S) in its bias add (h2_unscale): every factor a power of
        // two, the product exact.
        __shared__ float nj[16], nk[16], cmax[4];
        __syncthreads();
        {
            float pj = 0.0f, pk = 0.0f; // thread (ty, tx): partial squared norms of rows j0 + ty (Wj) and k0 + ty (Wk)
            for (int c = tx; c < COUT; c += 16) {
                pj += Wj[ty][c] * Wj[ty][c];
                pk += Wk[ty][c] * Wk[ty][c];
            }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) {
                pj += __shfl_xor(pj, o);
                pk += __shfl_xor(pk, o);
            }
            if (tx == 0) {
                nj[ty] = sqrtf(pj);
                nk[ty] = sqrtf(pk);
            }
            float cm = 0.0f;
            for (int c = tid; c < COUT; c += 256) cm = fmaxf(cm, fabsf(Cs[c]));
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) cm = fmaxf(cm, __shfl_xor(cm, o));
            if ((tid & 63) == 0) cmax[tid >> 6] = cm;
        }
        __syncthreads();
        auto pow2_ge = [](float v) { // the power of two >= v (1 for 0; clamped: a dead layer must not produce inf)
            int e = 0;
            if (v > 0.0f) (void)frexpf(v, &e);
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            return ldexpf(1.0f, e);
        };
        const float S = ldexpf(1.0f, 13) / pow2_ge(fmaxf(fmaxf(cmax[0], cmax[1]), fmaxf(cmax[2], cmax[3])));
        const float rj = pow2_ge(nj[ty]), rk = pow2_ge(nk[tx]);
        if (blockIdx.y == 0 && ty == 0 && k0 + tx < cin) {
            h2_ascale[k0 + tx] = 16.0f * rk;
            h2_unscale[k0 + tx] = rk / (16.0f * S);
        }
        float(*T)[LD] = Wj; // the panels are dead: the tile goes through them
        __syncthreads();
        T[ty][tx] = acc * (S / (rj * rk));
        __syncthreads();
        if (tid < 32) {
            const int kh = tid >> 4, c = tid & 15;
            unsigned h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; i++) split2(T[kh * 8 + 2 * i][c], T[kh * 8 + 2 * i + 1][c], h[i], l[i]);
            uint4 *dst = reinterpret_cast<uint4 *>(image) + ((size_t)(blockIdx.y * 2) * 2 + kh) * cin + k0 + c;
            dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
            dst[(size_t)2 * cin] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    } else if (image != nullptr) {
        // the matrix is the weight operand of ONE forward-type GEMM (mlp_fast.hip): its bf16 x 3 image goes out with it, in that
        // kernel's LDS order [slab = row / 16][piece][k-half][column][8 bf16] (cin % 16 == 0: this tile is one slab of 16 columns)
        __syncthreads();
        float(*T)[LD] = Wj; // the panels are dead: the tile goes through them
        T[ty][tx] = acc;
        __syncthreads();
        if (tid < 32) {
            const int kh = tid >> 4, c = tid & 15;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; i++) split3(T[kh * 8 + 2 * i][c], T[kh * 8 + 2 * i + 1][c], h[i], m[i], l[i]);
            uint4 *dst = reinterpret_cast<uint4 *>(image) + ((size_t)(blockIdx.y * 3) * 2 + kh) * cin + k0 + c;
            dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
            dst[(size_t)2 * cin] = make_uint4(m[0], m[1], m[2], m[3]);
            dst[(size_t)4 * cin] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    }
}

// da[g*k + argmax[g,c], :] += A[c] g'[g,c] W[:,c]^T.  One persistent workgroup per CU keeps W^T (cout x CIN) in LDS and walks
// groups: the group's cout channels are bucketed by their arg-max row (counting sort in LDS), then every wavefront takes
// rows, sums the W^T rows of the row's channels in registers and adds the result to the da row in one coalesced
// read-modify-write -- no floating-point atomics, rows without an arg-max are not touched.
// RED: the same pass also reduces the BatchNorm backward of the layer BELOW (whose output gradient da is): with pz = that
// layer's raw output, sums += [sum g', sum g' zhat], g' = da masked by its ReLU -- the separate votenet_bn_backward_reduce
// pass over (da, z) disappears.
struct PoolBelow {
    const float *z, *scale, *shift, *mean, *var;
    float eps;
    int relu;
    double *sums;
    CoefTail tail; // that layer's coefficient vector from the completed sums, by the last workgroup (common.h)
    int reverse;   // walk the groups back to front (votenet_debug_scatter_reverse)
    // piece layout (half.hip; K = kPiece = 16): a "group" is a piece q of 16 compact rows; gout / argmax / zsel are per CENTRE hc[q] / 4,
    // a channel belongs to this piece when its arg-max slot lies in [16 (hc[q] % 4), + 16); the dense part of row 0 is scaled by wh[q]
    const int *hc;
    const float *wh;
    int G;
    const int *nh_dev; // the number of pieces when only the device knows it (`groups` is then an upper bound), or NULL
};

template <int CIN, int COUT, int K, bool RED, int NWV = 8 /* wavefronts: 16 when W^T leaves room for one workgroup per CU only */>
__global__ __launch_bounds__(NWV * 64) void pool_dgrad_scatter_kernel(long groups, const float *__restrict__ gout,
                                                                 const int *__restrict__ argmax, const float *__restrict__ zsel,
                                                                 const float *__restrict__ coef, int relu, const float *__restrict__ wT,
                                                                 float *__restrict__ da, PoolBelow pb)
{
    static_assert(K == 64, "one lane per row in the prefix scan");
    constexpr bool HALF = false; // (the piece layout is served by pool_dgrad_scatter_wave_kernel only)
    constexpr int PL = CIN / 64; // floats per lane of a row
    constexpr int RW = K / NWV;  // rows per wavefront
    extern __shared__ __attribute__((aligned(16))) float pds_smem[];
    float *Wl = pds_smem;                                    // [COUT][CIN]
    float *sv = Wl + COUT * CIN;                              // [2][COUT]   (everything below is double-buffered by group parity)
    int *lst = reinterpret_cast<int *>(sv + 2 * COUT);        // [2][COUT] channels ordered by row
    int *cnt = lst + 2 * COUT;                                // [2][K]
    int *start = cnt + 2 * K;                                 // [2][K]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < COUT * CIN / 4; e += NWV * 64)
        reinterpret_cast<float4 *>(Wl)[e] = reinterpret_cast<const float4 *>(wT)[e];
    float cA = 0.f, cS = 0.f, cH = 0.f;
    const bool own = tid < COUT;
    if (own) {
        cA = coef[tid];
        cS = coef[3 * COUT + tid];
        cH = coef[4 * COUT + tid];
    }
    if (tid < 2 * K) cnt[tid] = 0;
    float bS[PL], bH[PL], bM[PL], bI[PL], s1[PL], s2[PL];
#pragma unroll
    for (int q = 0; q < PL; q++) {
        s1[q] = s2[q] = 0.0f;
        if (RED) {
            const int j = lane * PL + q;
            bS[q] = pb.scale[j];
            bH[q] = pb.shift[j];
            bM[q] = pb.mean[j];
            bI[q] = 1.0f / sqrtf(pb.var[j] + pb.eps);
        }
    }
    float n_z = 0.f, n_g = 0.f, n_w = 1.0f;
    int n_a = 0;
    auto fetch = [&](long g) {
        const long ctr = HALF ? (long)pb.hc[g] : g;
        if (HALF) n_w = pb.wh[g];
        if (own) {
            n_z = zsel[(size_t)ctr * COUT + tid];
            n_g = gout[(size_t)ctr * COUT + tid];
            n_a = argmax[(size_t)ctr * COUT + tid];
            if (HALF) {
                n_a -= g >= pb.G ? 32 : 0;
                if (n_a < 0 || n_a >= 32) n_a = -1; // the centre's other half holds this channel's arg-max
            }
        }
    };
    // pb.reverse: walk the groups from the last to the first.  The kernel before this one streamed da (and z) front to back, so the
    // Infinity Cache holds their TAILS when this pass starts: a pass that starts at the tail finds them there
    auto grp = [&](long i) { return pb.reverse ? groups - 1 - i : i; };
    if ((long)blockIdx.x < groups) fetch(grp(blockIdx.x));
    __syncthreads();
    int par = 0;
    for (long gi = blockIdx.x; gi < groups; gi += gridDim.x, par ^= 1) {
        const long g = grp(gi);
        float *svp = sv + par * COUT;
        int *lstp = lst + par * COUT, *cntp = cnt + par * K, *startp = start + par * K;
        const float zz = n_z, w31 = n_w;
        float gg = n_g;
        const int myrow = n_a;
        const bool mine = own && (!HALF || myrow >= 0);
        // this wavefront's da rows travel while the channels are bucketed
        float pre[RW][PL], zpre[RW][PL];
#pragma unroll
        for (int ri = 0; ri < RW; ri++)
#pragma unroll
            for (int q = 0; q < PL; q++) {
                const size_t off = ((size_t)g * K + wv + NWV * ri) * CIN + lane * PL + q;
                pre[ri][q] = da[off];
                if (RED) zpre[ri][q] = pb.z[off];
            }
        const long gn = gi + gridDim.x;
        if (gn < groups) fetch(grp(gn));
        int mypos = 0;
        if (own) {
            if (relu && !(zz * cS + cH > 0.0f)) gg = 0.0f;
            svp[tid] = cA * gg;
        }
        // slots are claimed wavefront by wavefront (lanes of one ds_add_rtn are served in lane order), so a row's channels sit in
        // ascending order and the floating-point sum below has ONE order: da is reproducible bit for bit
#pragma unroll
        for (int w2 = 0; w2 < COUT / 64; w2++) {
            if (wv == w2 && (!HALF || myrow >= 0)) mypos = atomicAdd(&cntp[myrow], 1);
            __syncthreads();
        }
        if (wv == 0) { // exclusive prefix of the K counters: one lane per row
            const int c0 = lane < K ? cntp[lane] : 0;
            int x = c0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(x, off);
                if (lane >= off) x += t;
            }
            if (lane < K) startp[lane] = x - c0;
        } else if (wv == 1) {
            if (lane < K) cnt[(par ^ 1) * K + lane] = 0; // the other parity's counters: last read before this group's first barrier
        }
        __syncthreads();
        if (mine) lstp[startp[myrow] + mypos] = tid;
        __syncthreads();
#pragma unroll
        for (int ri = 0; ri < RW; ri++) {
            const int r = wv + NWV * ri;
            const int n = cntp[r], s0 = startp[r];
            const bool scaled = HALF && r == 31 && w31 != 1.0f;
            if (!RED && n == 0 && !scaled) continue;
            float acc[PL];
#pragma unroll
            for (int q = 0; q < PL; q++) acc[q] = scaled ? pre[ri][q] * w31 : pre[ri][q];
            for (int i = 0; i < n; i += 4) {
                int c[4];
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool ok = i + u < n;
                    c[u] = lstp[s0 + (ok ? i + u : i)];
                    v[u] = ok ? svp[c[u]] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int q = 0; q < PL; q++) acc[q] += v[u] * Wl[c[u] * CIN + lane * PL + q];
            }
            if (n != 0 || scaled) {
                float *drow = da + ((size_t)g * K + r) * CIN + lane * PL;
#pragma unroll
                for (int q = 0; q < PL; q++) drow[q] = acc[q];
            }
            if (RED) {
#pragma unroll
                for (int q = 0; q < PL; q++) {
                    const float zz2 = zpre[ri][q];
                    const float gp = (pb.relu && !(zz2 * bS[q] + bH[q] > 0.0f)) ? 0.0f : acc[q];
                    s1[q] += gp;
                    s2[q] += gp * ((zz2 - bM[q]) * bI[q]);
                }
            }
        }
    }
    if (RED) { // combine the 8 wavefronts' column sums in LDS (W^T is no longer needed), one fp64 atomic per column and workgroup
        __syncthreads();
        float *red = Wl; // [NWV][2][CIN]
#pragma unroll
        for (int q = 0; q < PL; q++) {
            red[(wv * 2 + 0) * CIN + lane * PL + q] = s1[q];
            red[(wv * 2 + 1) * CIN + lane * PL + q] = s2[q];
        }
        __syncthreads();
        for (int e = tid; e < 2 * CIN; e += NWV * 64) {
            float t = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < NWV; w8++) t += red[w8 * 2 * CIN + e];
            unsafeAtomicAdd(&pb.sums[e], (double)t);
        }
        coef_tail(pb.tail, gridDim.x, CIN, pb.sums, pb.scale, pb.shift, pb.mean, pb.var, pb.eps);
    }
}

// The same pass with ONE WAVEFRONT per group and no workgroup barrier inside the loop (the default; the kernel above stays behind
// votenet_debug_scatter_form(0)).  A group's fixed work -- bucketing cout channels by their arg-max row, ~cout W^T rows to add -- does
// not shrink with the rows of a group, and with one group per 16-wave workgroup at a time every LDS round trip and every barrier
// (6 per group) was exposed: 5.4 us per group and CU whatever K.  Here the 8 / 16 wavefronts of a workgroup share only the read-only
// W^T image; each buckets its own group in a private LDS scratch (LDS operations of one wavefront execute in order: the lanes of one
// ds_add_rtn are served in lane order and the four are issued in channel order, so a row's channels sit in ascending order and da is
// reproducible bit for bit), scans the K counters with lane shuffles, then walks the rows -- their da / z rows prefetched eight at a
// time, one batch ahead, across groups -- adding the W^T rows of each row's channels.  Channels whose masked gradient is zero are not
// listed at all.
template <int CIN, int COUT, int K, bool RED, int NWV>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(NWV == 12 ? 3 : 4))) void pool_dgrad_scatter_wave_kernel(long groups, const float *__restrict__ gout,
                                                                           const int *__restrict__ argmax, const float *__restrict__ zsel,
                                                                           const float *__restrict__ coef, int relu,
                                                                           const float *__restrict__ wT, float *__restrict__ da, PoolBelow pb)
{
    static_assert(K == 64 || K == kPiece, "one lane per row in the prefix scan");
    constexpr bool HALF = K == kPiece; // piece layout (half.hip)
    constexpr int PL = CIN / 64;  // floats per lane of a row
    constexpr int NJ = COUT / 64; // channels per lane
    constexpr int RB = 4;         // rows per prefetch batch (two batches in flight; 128 VGPRs per wavefront)
    constexpr int NB = K / RB;    // batches per group (even: batch 0 of every group lives in register set 0)
    // a wavefront's scratch: the rows' lists of (value, channel), every row's list PADDED to a multiple of four entries with (0, channel 0)
    // -- the row loop then reads four entries with one 16-byte and one 8-byte LDS read and has no tail predicate (round 5: the loop over
    // single entries with its per-entry bounds branches was ~10 vector + 8 scalar instructions per entry: the pass was issue-bound) --
    // and the K counters
    constexpr int LCAP = COUT + 3 * K;   // entries incl. padding (at most three per row)
    constexpr int WS = LCAP * 6 + K * 4; // bytes: values (fp32), channels (16 bit), counters
    static_assert(WS % 16 == 0 && (LCAP * 4) % 16 == 0, "the value lists are read 16 bytes at a time");
    extern __shared__ __attribute__((aligned(16))) float pds_smem[];
    float *Wl = pds_smem; // [COUT][CIN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *scratch = reinterpret_cast<unsigned char *>(Wl + COUT * CIN) + (size_t)wv * WS;
    float *lv = reinterpret_cast<float *>(scratch);                             // [LCAP] A g' of the listed channels, row by row
    unsigned short *lc = reinterpret_cast<unsigned short *>(scratch + LCAP * 4); // [LCAP] their channel numbers
    int *cnt = reinterpret_cast<int *>(scratch + LCAP * 6);                     // [K]
    for (int e = tid; e < COUT * CIN / 4; e += NWV * 64) reinterpret_cast<float4 *>(Wl)[e] = reinterpret_cast<const float4 *>(wT)[e];
    if (lane < K) cnt[lane] = 0;
    float cA[NJ], cS[NJ], cH[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        cA[j] = coef[j * 64 + lane];
        cS[j] = coef[3 * COUT + j * 64 + lane];
        cH[j] = coef[4 * COUT + j * 64 + lane];
    }
    float bS[PL], bH[PL], bM[PL], bI[PL], s1[PL], s2[PL];
#pragma unroll
    for (int q = 0; q < PL; q++) {
        s1[q] = s2[q] = 0.0f;
        bS[q] = bH[q] = bM[q] = bI[q] = 0.0f;
        if (RED) {
            const int j = lane * PL + q;
            bS[q] = pb.scale[j];
            bH[q] = pb.shift[j];
            bM[q] = pb.mean[j];
            bI[q] = 1.0f / sqrtf(pb.var[j] + pb.eps);
        }
    }
    __syncthreads(); // W^T is in place; from here on the wavefronts run on their own
    if (HALF && pb.nh_dev != nullptr) groups = pb.nh_dev[0] < groups ? pb.nh_dev[0] : groups;
    const long stride = (long)gridDim.x * NWV;
    float nz[NJ], ng[NJ], nw = 1.0f;
    int na[NJ];
    auto fetch = [&](long g) {
        const int code = HALF ? pb.hc[g] : 0;
        const long ctr = HALF ? (long)(code / kBallPieces) : g;
        if (HALF) nw = pb.wh[g];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            nz[j] = zsel[(size_t)ctr * COUT + j * 64 + lane];
            ng[j] = gout[(size_t)ctr * COUT + j * 64 + lane];
            int a = argmax[(size_t)ctr * COUT + j * 64 + lane];
            if (HALF) {
                a -= (code % kBallPieces) * kPiece;
                if (a < 0 || a >= kPiece) a = -1; // another piece of the centre holds this channel's arg-max
            }
            na[j] = a;
        }
    };
    struct Rows {
        float d[RB][PL], z[RB][PL];
    };
    Rows R0, R1;
    auto load_rows = [&](Rows &r, long g, int rb) {
#pragma unroll
        for (int u = 0; u < RB; u++)
#pragma unroll
            for (int q = 0; q < PL; q++) {
                const size_t off = ((size_t)g * K + rb * RB + u) * CIN + lane * PL + q;
                r.d[u][q] = da[off];
                if (RED) r.z[u][q] = pb.z[off];
            }
    };
    long gi = (long)blockIdx.x * NWV + wv;
    if (gi < groups) {
        fetch(gi);
        load_rows(R0, gi, 0);
    }
    for (; gi < groups; gi += stride) {
        const long g = gi, gn = gi + stride;
        float v[NJ];
        int row[NJ], pos[NJ];
        const float w31 = nw;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            float gg = ng[j];
            if (relu && !(nz[j] * cS[j] + cH[j] > 0.0f)) gg = 0.0f;
            v[j] = cA[j] * gg;
            row[j] = v[j] != 0.0f ? na[j] : -1; // nothing to add: not listed
        }
        fetch(gn < groups ? gn : g); // (never a load under a branch: past the last group its own records are read again, unused)
#pragma unroll
        for (int j = 0; j < NJ; j++) pos[j] = row[j] >= 0 ? atomicAdd(&cnt[row[j]], 1) : 0;
        const int c0 = lane < K ? cnt[lane] : 0;
        const int c4 = (c0 + 3) & ~3; // the row's list, padded
        int x = c4; // exclusive prefix of the K padded counts: one lane per row
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(x, off);
            if (lane >= off) x += t;
        }
        const int start = x - c4;
        if (lane < K) {
            cnt[lane] = 0; // ready for the next group
            for (int t = c0; t < c4; t++) { // the padding: nothing times W^T row 0
                lc[start + t] = 0;
                lv[start + t] = 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int st = __shfl(start, row[j] >= 0 ? row[j] : 0);
            if (row[j] >= 0) {
                lc[st + pos[j]] = (unsigned short)(j * 64 + lane);
                lv[st + pos[j]] = v[j];
            }
        }
        auto batch = [&](const Rows &r, int rb) {
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int rr = rb * RB + u;
                const int n = __builtin_amdgcn_readlane(c4, rr), s0 = __builtin_amdgcn_readlane(start, rr);
                const bool scaled = HALF && rr == 0 && w31 != 1.0f; // the ball's slot 0 also stands for its dropped copies
                float acc[PL];
#pragma unroll
                for (int q = 0; q < PL; q++) acc[q] = scaled ? r.d[u][q] * w31 : r.d[u][q];
                for (int i = 0; i < n; i += 4) { // four entries per trip: their channels in 8 bytes, their values in 16 (all lanes read the same words)
                    // (element types as written -- no type punning between the 16-bit stores and a wider read; the alignment lets the
                    // compiler merge the four reads of each list into one 8-byte / 16-byte LDS read)
                    const unsigned short *pc = static_cast<const unsigned short *>(__builtin_assume_aligned(&lc[s0 + i], 8));
                    const float *pv = static_cast<const float *>(__builtin_assume_aligned(&lv[s0 + i], 16));
                    const unsigned cc[4] = {pc[0], pc[1], pc[2], pc[3]};
                    const float vv[4] = {pv[0], pv[1], pv[2], pv[3]};
                    constexpr bool kPacked = PL == 2;
                    if constexpr (kPacked) {
                        f32x2 wr[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) wr[t] = *reinterpret_cast<const f32x2 *>(&Wl[cc[t] * CIN + lane * PL]);
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            // THE BROADCAST OPERAND FIRST.  Written second -- fma(row, {v, v}, acc) -- the compiler folds the broadcast into
                            // op_sel on src1: v_pk_fma_f32 acc, row, [v_t, v_t+1], acc op_sel:[0,1,0] for the odd entries, whose LOW half
                            // takes src1's HIGH register -- and that form returns a wrong low half in lanes 48-63 now and then while MFMA
                            // wavefronts of another kernel (the weight-gradient stream) run on the same compute unit: one list entry's
                            // contribution missing in 16 columns of a row, 10-25 % of the launches with three processes on the GPU
                            // (tools/probe/src/pk_opsel_hazard.hip, profiles/r05_pk_opsel_hazard.txt; tools/check_isa_hazards.py keeps
                            // the form out of the library).  op_sel on src0 is exact.
#ifdef SCATTER_PK_SRC1 /* probe build: the hazardous operand order, for tools/probe/scatter_repeat.py */
                            const f32x2 a2 = __builtin_elementwise_fma(wr[t], f32x2{vv[t], vv[t]}, f32x2{acc[0], acc[1]});
#else
                            const f32x2 a2 = __builtin_elementwise_fma(f32x2{vv[t], vv[t]}, wr[t], f32x2{acc[0], acc[1]});
#endif
                            acc[0] = a2.x;
                            acc[1] = a2.y;
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; t++)
#pragma unroll
                            for (int q = 0; q < PL; q++) acc[q] = __builtin_fmaf(vv[t], Wl[cc[t] * CIN + lane * PL + q], acc[q]);
                    }
                }
                if (n != 0 || scaled) {
                    float *drow = da + ((size_t)g * K + rr) * CIN + lane * PL;
#pragma unroll
                    for (int q = 0; q < PL; q++) drow[q] = acc[q];
                }
                if (RED) {
#pragma unroll
                    for (int q = 0; q < PL; q++) {
                        const float zz2 = r.z[u][q];
                        const float gp = (pb.relu && !(zz2 * bS[q] + bH[q] > 0.0f)) ? 0.0f : acc[q];
                        s1[q] += gp;
                        s2[q] += gp * ((zz2 - bM[q]) * bI[q]);
                    }
                }
            }
        };
#pragma unroll 1
        for (int rb = 0; rb < NB; rb += 2) {
            load_rows(R1, g, rb + 1);
            batch(R0, rb);
            const bool more = rb + 2 < NB;
            load_rows(R0, more ? g : (gn < groups ? gn : g), more ? rb + 2 : 0); // (the next group's first rows; at the very end: unused)
            batch(R1, rb + 1);
        }
    }
    if (RED) { // combine the wavefronts' column sums in LDS (W^T is no longer needed), one fp64 atomic per column and workgroup
        __syncthreads();
        float *red = Wl; // [NWV][2][CIN]
#pragma unroll
        for (int q = 0; q < PL; q++) {
            red[(wv * 2 + 0) * CIN + lane * PL + q] = s1[q];
            red[(wv * 2 + 1) * CIN + lane * PL + q] = s2[q];
        }
        __syncthreads();
        for (int e = tid; e < 2 * CIN; e += NWV * 64) {
            float t = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < NWV; w8++) t += red[w8 * 2 * CIN + e];
            unsafeAtomicAdd(&pb.sums[e], (double)t);
        }
        coef_tail(pb.tail, gridDim.x, CIN, pb.sums, pb.scale, pb.shift, pb.mean, pb.var, pb.eps);
    }
}

// dW[:, c] += sum_g x[g*k + argmax[g,c], :] * A[c] g'[g,c]   and   colsum[j] += sum_r x[r, j]
// x = act(xz * in_scale + in_shift) staged per group in LDS; thread c owns output column c (CIN accumulators).
// K = kPiece = 16: the piece layout (half.hip) -- a "group" is a piece q of 16 compact rows of xz, gout / argmax / zsel are per centre
// hc[q] / 4 and a channel counts here when its arg-max slot lies in this piece; the column sums weigh row 0 by wh[q].
// TEAMS = 2 (piece layout): two 256-thread teams per workgroup, each walking its own groups with its own tiles; at the end the second
// team's accumulators go through LDS and ONE team flushes.  What bounds this pass is a * groups / teams + b * workgroups with b = the flush
// of a workgroup's cin x cout partial by atomics (tools/probe/sparse_time.py: a = 1.2 us, b = 0.11 us): two teams halve b per unit of
// parallelism.
template <int CIN, int K, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS) void pool_wgrad_sparse_kernel(long groups, int cout, const float *__restrict__ xz,
                                                                const float *__restrict__ in_scale, const float *__restrict__ in_shift,
                                                                int in_relu, const float *__restrict__ gout,
                                                                const int *__restrict__ argmax, const float *__restrict__ zsel,
                                                                const float *__restrict__ coef, int relu, float *__restrict__ dw,
                                                                float *__restrict__ colsum, float *__restrict__ part,
                                                                const int *__restrict__ hc, const float *__restrict__ wh, int G,
                                                                const int *__restrict__ nh_dev)
{
    constexpr bool HALF = K == kPiece;
    constexpr int LD = CIN + 4;
    constexpr int NBUF = HALF ? 2 : 1;
    if (HALF && nh_dev != nullptr) groups = nh_dev[0] < groups ? nh_dev[0] : groups; // (uniform: every thread reads the same count)
    extern __shared__ __attribute__((aligned(16))) float sparse_smem[]; // [TEAMS][NBUF][K][LD]; at the end [CIN + 1][cout] of team 1
    const int team = TEAMS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    const int tid = threadIdx.x & 255; // inside the team
    float(*xs)[K][LD] = reinterpret_cast<float(*)[K][LD]>(sparse_smem + (size_t)team * NBUF * K * LD);
    float acc[CIN];
#pragma unroll
    for (int i = 0; i < CIN; i++) acc[i] = 0.0f;
    float csum = 0.0f;
    const bool own = tid < cout;
    const float cA = own ? coef[tid] : 0.0f, cS = own ? coef[3 * cout + tid] : 0.0f, cH = own ? coef[4 * cout + tid] : 0.0f;
    constexpr int Q = CIN / 4;
    const int q = tid % Q;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_scale) {
        sc = *reinterpret_cast<const float4 *>(in_scale + q * 4);
        sh = *reinterpret_cast<const float4 *>(in_shift + q * 4);
    }
    const float fl = (in_scale && in_relu) ? 0.0f : -__builtin_inff();
    constexpr int NL = K * Q / 256; // float4 per thread per group tile
    // A piece's tile is 8 KB and its work a fraction of a microsecond: with one tile in flight per team the pass ran at one loaded-memory
    // latency per group.  D tiles (and their channels' gout / zsel / argmax) travel ahead in registers; the LDS tile is double-buffered,
    // one barrier per group.
    constexpr int D = HALF ? 4 : 1;
    float4 nxt[D][NL];
    float n_g[D], n_z[D], n_w[D]; // the group's pooled gradient / raw arg-max value of this thread's channel, the piece's weight
    int n_a[D];                   // ... and its arg-max row
    auto fetch = [&](int d, long g) {
        const float4 *src = reinterpret_cast<const float4 *>(xz + (size_t)g * K * CIN);
#pragma unroll
        for (int h = 0; h < NL; h++) nxt[d][h] = src[tid + h * 256];
        const int code = HALF ? hc[g] : 0;
        const long ctr = HALF ? (long)(code / kBallPieces) : g;
        n_w[d] = HALF ? wh[g] : 1.0f;
        n_z[d] = n_g[d] = 0.0f;
        n_a[d] = 0;
        if (own) {
            n_z[d] = zsel[(size_t)ctr * cout + tid];
            n_g[d] = gout[(size_t)ctr * cout + tid];
            n_a[d] = argmax[(size_t)ctr * cout + tid];
            if (HALF) {
                n_a[d] -= (code % kBallPieces) * kPiece;
                if (n_a[d] < 0 || n_a[d] >= kPiece) { // another piece of the centre holds this channel's arg-max
                    n_a[d] = 0;
                    n_g[d] = 0.0f;
                }
            }
        }
    };
    const long first = (long)blockIdx.x * TEAMS + team, stride = (long)gridDim.x * TEAMS; // this team's groups: first, first + stride, ...
#pragma unroll
    for (int d = 0; d < D; d++) {
        const long g = first + (long)d * stride;
        fetch(d, g < groups ? g : groups - 1);
    }
    int buf = 0;
    // every team of the workgroup runs the same number of rounds (the barriers are the workgroup's): a team past its last group idles
    for (long g0 = (long)blockIdx.x * TEAMS; g0 < groups; g0 += (long)D * stride) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (g0 + (long)d * stride >= groups) break; // (uniform over the workgroup: no team has a group in this round)
            const long g = g0 + team + (long)d * stride;
            const bool live = g < groups;
            float(*xt)[LD] = xs[HALF ? buf : 0];
#pragma unroll
            for (int h = 0; h < NL; h++) { // (tid + h*256) % Q == q: 256 % Q == 0
                float4 v = nxt[d][h];
                v.x = fmaxf(v.x * sc.x + sh.x, fl);
                v.y = fmaxf(v.y * sc.y + sh.y, fl);
                v.z = fmaxf(v.z * sc.z + sh.z, fl);
                v.w = fmaxf(v.w * sc.w + sh.w, fl);
                *reinterpret_cast<float4 *>(&xt[(tid + h * 256) / Q][q * 4]) = v;
            }
            float gg = live ? n_g[d] : 0.0f;
            const float zz = n_z[d], w31 = n_w[d];
            const int ar = n_a[d];
            const long gn = g + (long)D * stride;
            fetch(d, gn < groups ? gn : (live ? g : groups - 1)); // this stage's registers travel again while the tile is used
            __syncthreads();
            if (own) {
                if (relu && !(zz * cS + cH > 0.0f)) gg = 0.0f;
                const float v = cA * gg;
                if (v != 0.0f) {
                    const float4 *row = reinterpret_cast<const float4 *>(&xt[ar][0]);
                    // eight 16-byte LDS reads in flight, then their 32 multiply-adds
#pragma unroll
                    for (int i0 = 0; i0 < Q; i0 += 8) {
                        float4 a[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) a[i] = row[i0 + i];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            acc[4 * (i0 + i)] += a[i].x * v;
                            acc[4 * (i0 + i) + 1] += a[i].y * v;
                            acc[4 * (i0 + i) + 2] += a[i].z * v;
                            acc[4 * (i0 + i) + 3] += a[i].w * v;
                        }
                    }
                }
            }
            if (live && tid >= 256 - CIN) { // the column sums ride on the waves that own no (or the last) output columns
                const int jc = tid - (256 - CIN);
#pragma unroll 8
                for (int r = (HALF ? 1 : 0); r < K; r++) csum += xt[r][jc];
                if (HALF) csum += w31 * xt[0][jc]; // the ball's slot 0 also stands for its dropped copies
            }
            if (HALF) buf ^= 1; // the other tile was last read before this group's barrier: the next group may overwrite it
            else __syncthreads();
        }
    }
    if (TEAMS > 1) { // team 1 hands its sums to team 0 through LDS (the tiles are dead), team 0 flushes
        __syncthreads();
        float *comb = sparse_smem; // [CIN + 1][cout]
        if (team == 1) {
            if (own) {
#pragma unroll
                for (int i = 0; i < CIN; i++) comb[(size_t)i * cout + tid] = acc[i];
            }
            if (tid >= 256 - CIN) comb[(size_t)CIN * cout + tid - (256 - CIN)] = csum;
        }
        __syncthreads();
        if (team == 1) return;
        if (own) {
#pragma unroll
            for (int i = 0; i < CIN; i++) acc[i] += comb[(size_t)i * cout + tid];
        }
        if (tid >= 256 - CIN) csum += comb[(size_t)CIN * cout + tid - (256 - CIN)];
    }
    if (part) { // this workgroup's slice [(CIN + 1) x cout]: dW rows, then the column sums; added in workgroup order afterwards
        float *__restrict__ mine = part + (size_t)blockIdx.x * (CIN + 1) * cout;
        if (own) {
#pragma unroll
            for (int i = 0; i < CIN; i++) mine[(size_t)i * cout + tid] = acc[i];
        }
        if (tid >= 256 - CIN) mine[(size_t)CIN * cout + tid - (256 - CIN)] = csum;
        return;
    }
    if (own) {
#pragma unroll
        for (int i = 0; i < CIN; i++) unsafeAtomicAdd(&dw[(size_t)i * cout + tid], acc[i]);
    }
    if (tid >= 256 - CIN) unsafeAtomicAdd(&colsum[tid - (256 - CIN)], csum);
}

// The piece-layout pass above walks PIECES: a channel counts in the one piece of its centre that holds its arg-max slot, so a centre of
// kc pieces sends every wavefront through the 32-read / 128-multiply-add row loop kc times with 1 / kc of its lanes active (sa1, sa3,
// sa4: 2.5-2.9 pieces per centre).  Here a group is a CENTRE (round 5): the tiles of all its kept pieces (piece 0 at compact rows
// 16 c, piece j >= 1 at 16 (G + pos[3 c + j - 1]), half.hip) are staged together -- up to 4 x 16 rows, double-buffered -- and every
// channel reads its arg-max row once: the row loop runs once per centre with every lane of a live channel active.  Same sums as the
// piece pass in another association.  colsum: every row of every kept piece, the ball's slot 0 weighted by wh[c].
template <int CIN, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS) void pool_wgrad_sparse_centre_kernel(int G, long nh, int cout, const float *__restrict__ xz,
                                                                       const float *__restrict__ in_scale, const float *__restrict__ in_shift,
                                                                       int in_relu, const float *__restrict__ gout,
                                                                       const int *__restrict__ argmax, const float *__restrict__ zsel,
                                                                       const float *__restrict__ coef, int relu, float *__restrict__ dw,
                                                                       float *__restrict__ colsum, const int *__restrict__ pos,
                                                                       const float *__restrict__ wh, const int *__restrict__ nh_dev)
{
    constexpr int K = kPiece, NP = kBallPieces;
    constexpr int LD = CIN + 4;
    if (nh_dev != nullptr) nh = nh_dev[0] < nh ? nh_dev[0] : nh; // (uniform: every thread reads the same count)
    extern __shared__ __attribute__((aligned(16))) float sparse_smem[]; // [TEAMS][2][NP][K][LD]; at the end [CIN + 1][cout] of team 1
    const int team = TEAMS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    const int tid = threadIdx.x & 255; // inside the team
    float(*xs)[NP][K][LD] = reinterpret_cast<float(*)[NP][K][LD]>(sparse_smem + (size_t)team * 2 * NP * K * LD);
    float acc[CIN];
#pragma unroll
    for (int i = 0; i < CIN; i++) acc[i] = 0.0f;
    float csum = 0.0f;
    const bool own = tid < cout;
    const float cA = own ? coef[tid] : 0.0f, cS = own ? coef[3 * cout + tid] : 0.0f, cH = own ? coef[4 * cout + tid] : 0.0f;
    constexpr int Q = CIN / 4;
    const int q = tid % Q;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_scale) {
        sc = *reinterpret_cast<const float4 *>(in_scale + q * 4);
        sh = *reinterpret_cast<const float4 *>(in_shift + q * 4);
    }
    const float fl = (in_scale && in_relu) ? 0.0f : -__builtin_inff();
    constexpr int NL = K * Q / 256; // float4 per thread per piece tile (2 at CIN = 128, 1 at CIN = 64)
    constexpr int D = 2;            // centres in flight in registers (their tiles and their channels' gout / zsel / argmax)
    float4 nxt[D][NP][NL];
    long n_row[D][NP];              // first compact row of piece j of the staged centre, -1: not kept (uniform over the team)
    float n_g[D], n_z[D], n_w[D];
    int n_a[D];
    // No load under a branch: a conditional load makes the compiler wait for vmcnt(0) at the next use of ANY loaded value, which
    // turned the two-centre prefetch into one full memory round trip per centre (25 such waits in the loop's code).  The tiles of
    // pieces a ball did not keep are "loaded" through a buffer descriptor at an offset past its end: such a read returns zeros and
    // moves no data; the channels beyond cout read channel 0 of the centre and ignore it.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)xz, 0, (int)((size_t)nh * K * CIN * 4), 0x00020000);
    const int ch = own ? tid : 0;
    auto fetch = [&](int d, long c) {
        n_row[d][0] = c * K;
#pragma unroll
        for (int j = 1; j < NP; j++) {
            const int p = pos[(size_t)c * (NP - 1) + j - 1];
            n_row[d][j] = (p >= 0 && (long)G + p < nh) ? ((long)G + p) * K : -1;
        }
#pragma unroll
        for (int j = 0; j < NP; j++) {
            // (the whole offset in the lane offset: that is what the descriptor's range check sees)
            const unsigned base = n_row[d][j] < 0 ? 0x80000000u : (unsigned)(n_row[d][j] * CIN * 4);
#pragma unroll
            for (int h = 0; h < NL; h++)
                nxt[d][j][h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, base + (unsigned)(tid + h * 256) * 16u, 0, 0));
        }
        n_w[d] = wh[c];
        n_z[d] = zsel[(size_t)c * cout + ch];
        const float gv = gout[(size_t)c * cout + ch];
        n_g[d] = own ? gv : 0.0f;
        n_a[d] = argmax[(size_t)c * cout + ch];
    };
    const long first = (long)blockIdx.x * TEAMS + team, stride = (long)gridDim.x * TEAMS; // this team's centres: first, first + stride, ...
#pragma unroll
    for (int d = 0; d < D; d++) {
        const long c = first + (long)d * stride;
        fetch(d, c < G ? c : G - 1);
    }
    int buf = 0;
    // every team of the workgroup runs the same number of rounds (the barriers are the workgroup's): a team past its last centre idles
    for (long c0 = (long)blockIdx.x * TEAMS; c0 < G; c0 += (long)D * stride) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (c0 + (long)d * stride >= G) break; // (uniform over the workgroup: no team has a centre in this round)
            const long c = c0 + team + (long)d * stride;
            const bool live = c < G;
            float(*xt)[K][LD] = xs[buf];
            bool kept[NP];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                kept[j] = n_row[d][j] >= 0;
                if (!kept[j]) continue;
#pragma unroll
                for (int h = 0; h < NL; h++) { // (tid + h*256) % Q == q: 256 % Q == 0
                    float4 v = nxt[d][j][h];
                    v.x = fmaxf(v.x * sc.x + sh.x, fl);
                    v.y = fmaxf(v.y * sc.y + sh.y, fl);
                    v.z = fmaxf(v.z * sc.z + sh.z, fl);
                    v.w = fmaxf(v.w * sc.w + sh.w, fl);
                    *reinterpret_cast<float4 *>(&xt[j][(tid + h * 256) / Q][q * 4]) = v;
                }
            }
            float gg = live ? n_g[d] : 0.0f;
            const float zz = n_z[d], w31 = n_w[d];
            const int ar = n_a[d];
            const long cn = c + (long)D * stride;
            fetch(d, cn < G ? cn : (live ? c : G - 1)); // this stage's registers travel again while the tiles are used
            __syncthreads();
            if (own) {
                if (relu && !(zz * cS + cH > 0.0f)) gg = 0.0f;
                const float v = cA * gg;
                const int pj = ar >> 4;
                const bool there = pj == 0 ? kept[0] : pj == 1 ? kept[1] : pj == 2 ? kept[2] : kept[3]; // (an arg-max is a real neighbour: always)
                if (v != 0.0f && there) {
                    const float4 *row = reinterpret_cast<const float4 *>(&xt[pj][ar & (K - 1)][0]);
                    // eight 16-byte LDS reads in flight, then their 32 multiply-adds
#pragma unroll
                    for (int i0 = 0; i0 < Q; i0 += 8) {
                        float4 a[8];
#pragma unroll
                        for (int i = 0; i < 8; i++) a[i] = row[i0 + i];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            acc[4 * (i0 + i)] += a[i].x * v;
                            acc[4 * (i0 + i) + 1] += a[i].y * v;
                            acc[4 * (i0 + i) + 2] += a[i].z * v;
                            acc[4 * (i0 + i) + 3] += a[i].w * v;
                        }
                    }
                }
            }
            if (live && tid >= 256 - CIN) { // the column sums ride on the waves that own no (or the last) output columns
                const int jc = tid - (256 - CIN);
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    if (!kept[j]) continue;
#pragma unroll 8
                    for (int r = (j == 0 ? 1 : 0); r < K; r++) csum += xt[j][r][jc];
                }
                csum += w31 * xt[0][0][jc]; // the ball's slot 0 also stands for its dropped copies
            }
            buf ^= 1; // the other buffer was last read before this centre's barrier: the next centre may overwrite it
        }
    }
    if (TEAMS > 1) { // team 1 hands its sums to team 0 through LDS (the tiles are dead), team 0 flushes
        __syncthreads();
        float *comb = sparse_smem; // [CIN + 1][cout]
        if (team == 1) {
            if (own) {
#pragma unroll
                for (int i = 0; i < CIN; i++) comb[(size_t)i * cout + tid] = acc[i];
            }
            if (tid >= 256 - CIN) comb[(size_t)CIN * cout + tid - (256 - CIN)] = csum;
        }
        __syncthreads();
        if (team == 1) return;
        if (own) {
#pragma unroll
            for (int i = 0; i < CIN; i++) acc[i] += comb[(size_t)i * cout + tid];
        }
        if (tid >= 256 - CIN) csum += comb[(size_t)CIN * cout + tid - (256 - CIN)];
    }
    if (own) {
#pragma unroll
        for (int i = 0; i < CIN; i++) unsafeAtomicAdd(&dw[(size_t)i * cout + tid], acc[i]);
    }
    if (tid >= 256 - CIN) unsafeAtomicAdd(&colsum[tid - (256 - CIN)], csum);
}

// dW[j,c] += C[c] * (gram[j,:] . W[:,c]) + colsum[j] * (B[c] + C[c] b[c])
__global__ __launch_bounds__(256) void pool_wgrad_finish_kernel(int cin, int cout, const float *__restrict__ gram,
                                                                const float *__restrict__ colsum, const float *__restrict__ w,
                                                                const float *__restrict__ bias, const float *__restrict__ coef,
                                                                float *__restrict__ dw)
{
    __shared__ float Gs[16][33], Ws[32][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j0 = blockIdx.y * 16, c0 = blockIdx.x * 16;
    float acc = 0.0f;
    for (int i0 = 0; i0 < cin; i0 += 32) {
        for (int e = threadIdx.x; e < 16 * 32; e += 256) {
            const int r = e >> 5, ii = e & 31;
            Gs[r][ii] = (j0 + r < cin && i0 + ii < cin) ? gram[(size_t)(j0 + r) * cin + i0 + ii] : 0.0f;
            const int wr = e >> 4, wc = e & 15;
            Ws[wr][wc] = (i0 + wr < cin && c0 + wc < cout) ? w[(size_t)(i0 + wr) * cout + c0 + wc] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int ii = 0; ii < 32; ii++) acc += Gs[ty][ii] * Ws[ii][tx];
        __syncthreads();
    }
    const int j = j0 + ty, c = c0 + tx;
    if (j < cin && c < cout) {
        const float Cc = coef[2 * cout + c];
        dw[(size_t)j * cout + c] += Cc * acc + colsum[j] * (coef[cout + c] + Cc * (bias ? bias[c] : 0.0f));
    }
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_pool_backward_supported(int cin, int cout, int k)
{
    return k == 64 && ((cin == 128 && (cout == 256 || cout == 128)) || (cin == 64 && cout == 128));
}

static int g_zsel_groups = 32, g_zsel_cap = 256;
extern "C" void votenet_debug_zsel_grid(int groups_per_wg, int cap) // tuning hook
{
    VN_DEBUG_GATE();
    g_zsel_groups = groups_per_wg > 0 ? groups_per_wg : 32;
    g_zsel_cap = cap > 0 ? cap : 256;
}
extern "C" int votenet_bn_backward_reduce_pool(long groups, int c, const float *gout, const float *zsel, const float *scale,
                                               const float *shift, const float *mean, const float *var, float eps, int relu,
                                               double *sums, const votenet_coef_tail *tail, void *stream)
{
    VN_REQUIRE(groups > 0 && c > 0, "bn_backward_reduce_pool expects groups > 0, c > 0");
    VN_REQUIRE(gout && zsel && scale && shift && mean && var && sums, "bn_backward_reduce_pool: null buffer");
    VN_REQUIRE(!tail || (tail->ticket && tail->gamma && tail->coef && tail->rows > 0), "bn_backward_reduce_pool: incomplete coefficient tail");
    const int ny = (c + 63) / 64;
    hipLaunchKernelGGL(bn_bwd_reduce_zsel_kernel, dim3(pb_grid(groups, g_zsel_groups, g_zsel_cap / ny), ny), dim3(256), 0, as_stream(stream), groups,
                       c, gout, zsel, scale, shift, mean, var, eps, relu, sums, to_tail(tail));
    return check_launch("bn_backward_reduce_pool");
}

static int pool_dgrad_prepare_launch(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat, float *cvec,
                                     void *image, void *stream, float *h2_ascale = nullptr, float *h2_unscale = nullptr)
{
    VN_REQUIRE(cin > 0 && w && coef && mmat && cvec, "pool_dgrad_prepare: bad arguments");
    VN_REQUIRE((cout == 128 || cout == 256) && (uintptr_t)w % 16 == 0, "pool_dgrad_prepare: cout must be 128 or 256 (got %d), w 16-byte aligned", cout);
    VN_REQUIRE(image == nullptr || (cin % 16 == 0 && (uintptr_t)image % 16 == 0), "pool_dgrad_prepare_split: cin %% 16 == 0 and a 16-byte aligned image");
    const dim3 grid((cin + 15) / 16, (cin + 15) / 16);
    unsigned *img = static_cast<unsigned *>(image);
    if (cout == 256)
        hipLaunchKernelGGL(pool_dgrad_prepare_kernel<256>, grid, dim3(256), 0, as_stream(stream), cin, w, bias, coef, mmat, cvec, img, h2_ascale, h2_unscale);
    else
        hipLaunchKernelGGL(pool_dgrad_prepare_kernel<128>, grid, dim3(256), 0, as_stream(stream), cin, w, bias, coef, mmat, cvec, img, h2_ascale, h2_unscale);
    return check_launch("pool_dgrad_prepare");
}
extern "C" int votenet_pool_dgrad_prepare(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat,
                                          float *cvec, void *stream)
{
    return pool_dgrad_prepare_launch(cin, cout, w, bias, coef, mmat, cvec, nullptr, stream);
}
// The same, and the matrix's image as TWO fp16 pieces (cin * cin * 4 bytes) scaled by powers of two, with the two vectors (cin floats
// each) the GEMM needs to undo the scaling: register all three with votenet_register_split_weights_scaled around the one GEMM that
// multiplies by mmat (see the kernel for the bound the scales come from).
extern "C" int votenet_pool_dgrad_prepare_h2(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat,
                                             float *cvec, void *image, float *ascale, float *unscale, void *stream)
{
    VN_REQUIRE(image != nullptr && ascale != nullptr && unscale != nullptr, "pool_dgrad_prepare_h2: null image / scale vectors");
    return pool_dgrad_prepare_launch(cin, cout, w, bias, coef, mmat, cvec, image, stream, ascale, unscale);
}
// The same, and the bf16 x 3 image of mmat (cin * cin * 6 bytes, votenet_split_weights' layout) written by the same launch: the
// caller registers it (votenet_register_split_weights) around the one GEMM that multiplies by mmat.
extern "C" int votenet_pool_dgrad_prepare_split(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat,
                                                float *cvec, void *image, void *stream)
{
    VN_REQUIRE(image != nullptr, "pool_dgrad_prepare_split: null image");
    return pool_dgrad_prepare_launch(cin, cout, w, bias, coef, mmat, cvec, image, stream);
}

static int g_scatter_reverse = 0;
extern "C" void votenet_debug_scatter_reverse(int on) { VN_DEBUG_GATE(); g_scatter_reverse = on ? 1 : 0; }
static int g_scatter_nwv = 16; // votenet_debug_scatter_waves (tuning hook): wavefronts per workgroup of the 128 -> 256 piece-layout scatter (12 or 16)
extern "C" void votenet_debug_scatter_waves(int n) { VN_DEBUG_GATE(); g_scatter_nwv = n == 12 ? 12 : 16; }
static int g_scatter_wgs = 0; // votenet_debug_scatter_workgroups (tuning hook): 0 = one workgroup per CU and LDS share
extern "C" void votenet_debug_scatter_workgroups(int n) { VN_DEBUG_GATE(); g_scatter_wgs = n > 0 ? n : 0; }
static int g_scatter_form = 1; // 1: one wavefront per group (pool_dgrad_scatter_wave_kernel), 0: one workgroup per group
extern "C" void votenet_debug_scatter_form(int form) { VN_DEBUG_GATE(); g_scatter_form = form ? 1 : 0; }
static int pool_dgrad_scatter_launch(long groups, int k, int cin, int cout, const float *gout, const int *argmax, const float *zsel,
                                     const float *coef, int relu, const float *wT, float *da, const float *below_z,
                                     const float *below_scale, const float *below_shift, const float *below_mean, const float *below_var,
                                     float eps, int below_relu, double *below_sums, const votenet_coef_tail *below_tail, const int *hc,
                                     const float *wh, int G, void *stream, const int *nh_dev = nullptr);
extern "C" int votenet_pool_dgrad_scatter(long groups, int k, int cin, int cout, const float *gout, const int *argmax,
                                          const float *zsel, const float *coef, int relu, const float *wT, float *da,
                                          const float *below_z, const float *below_scale, const float *below_shift,
                                          const float *below_mean, const float *below_var, float eps, int below_relu,
                                          double *below_sums, const votenet_coef_tail *below_tail, void *stream)
{
    VN_REQUIRE(!below_tail || (below_z && below_tail->ticket && below_tail->gamma && below_tail->coef && below_tail->rows > 0),
               "pool_dgrad_scatter: incomplete coefficient tail");
    VN_REQUIRE(groups > 0 && gout && argmax && zsel && coef && wT && da, "pool_dgrad_scatter: bad arguments");
    VN_REQUIRE(votenet_pool_backward_supported(cin, cout, k), "pool_dgrad_scatter: unsupported shape cin=%d cout=%d k=%d", cin, cout, k);
    VN_REQUIRE((uintptr_t)wT % 16 == 0, "pool_dgrad_scatter: wT must be 16-byte aligned");
    VN_REQUIRE(!below_z || (below_scale && below_shift && below_mean && below_var && below_sums),
               "pool_dgrad_scatter: below_z given without the layer's BatchNorm vectors / sums");
    return pool_dgrad_scatter_launch(groups, k, cin, cout, gout, argmax, zsel, coef, relu, wT, da, below_z, below_scale, below_shift, below_mean,
                                     below_var, eps, below_relu, below_sums, below_tail, nullptr, nullptr, 0, stream);
}

// The same pass on the piece layout (half.hip): da and below_z have 16 * nh compact rows; gout / argmax / zsel stay per centre.
extern "C" int votenet_pool_dgrad_scatter_half(long nh, int G, int cin, int cout, const float *gout, const int *argmax, const float *zsel,
                                               const float *coef, int relu, const float *wT, float *da, const int *hc, const float *wh,
                                               const float *below_z, const float *below_scale, const float *below_shift,
                                               const float *below_mean, const float *below_var, float eps, int below_relu,
                                               double *below_sums, const votenet_coef_tail *below_tail, const int *nh_dev, void *stream)
{
    VN_REQUIRE(hc && wh && G > 0 && nh >= G && nh <= (long)kBallPieces * G, "pool_dgrad_scatter_half: bad piece-layout arguments");
    VN_REQUIRE(gout && argmax && zsel && coef && wT && da && (uintptr_t)wT % 16 == 0, "pool_dgrad_scatter_half: null / unaligned buffer");
    VN_REQUIRE(votenet_pool_backward_supported(cin, cout, 64), "pool_dgrad_scatter_half: unsupported shape cin=%d cout=%d", cin, cout);
    VN_REQUIRE(!below_z || (below_scale && below_shift && below_mean && below_var && below_sums),
               "pool_dgrad_scatter_half: below_z given without the layer's BatchNorm vectors / sums");
    VN_REQUIRE(!below_tail || (below_z && below_tail->ticket && below_tail->gamma && below_tail->coef && below_tail->rows > 0),
               "pool_dgrad_scatter_half: incomplete coefficient tail");
    return pool_dgrad_scatter_launch(nh, 64, cin, cout, gout, argmax, zsel, coef, relu, wT, da, below_z, below_scale, below_shift, below_mean,
                                     below_var, eps, below_relu, below_sums, below_tail, hc, wh, G, stream, nh_dev);
}

static int pool_dgrad_scatter_launch(long groups, int k, int cin, int cout, const float *gout, const int *argmax, const float *zsel,
                                     const float *coef, int relu, const float *wT, float *da, const float *below_z,
                                     const float *below_scale, const float *below_shift, const float *below_mean, const float *below_var,
                                     float eps, int below_relu, double *below_sums, const votenet_coef_tail *below_tail, const int *hc,
                                     const float *wh, int G, void *stream, const int *nh_dev)
{
    hipStream_t st = as_stream(stream);
    const PoolBelow pb = {below_z, below_scale, below_shift, below_mean, below_var, eps, below_relu, below_sums, to_tail(below_tail), g_scatter_reverse, hc, wh, G, nh_dev};
    auto go = [&](auto kern, int ci, int co, int threads = 512) {
        const size_t smem = ((size_t)co * ci + 4 * co + 4 * k) * 4;
        const int per_cu = smem > 80 * 1024 ? 1 : (smem > 40 * 1024 ? 2 : 4);
        static std::set<const void *> raised; // the attribute is per kernel: set once
        static std::mutex raised_mu;          // entry points may be called from several host threads
        bool fresh;
        {
            std::lock_guard<std::mutex> lock(raised_mu);
            fresh = raised.insert(reinterpret_cast<const void *>(kern)).second;
        }
        if (fresh)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3(pb_grid(groups, 2, 256 * per_cu)), dim3(threads), smem, st, groups, gout, argmax, zsel, coef, relu,
                           wT, da, pb);
    };
    // one wavefront per group (the default)
    auto gow = [&](auto kern, int ci, int co, int kk, int nwv) {
        const size_t smem = (size_t)co * ci * 4 + (size_t)nwv * ((co + 3 * kk) * 6 + kk * 4);
        const int per_cu = smem > 80 * 1024 ? 1 : (smem > 40 * 1024 ? 2 : 4);
        static std::set<const void *> raised_w;
        static std::mutex raised_w_mu;
        bool fresh;
        {
            std::lock_guard<std::mutex> lock(raised_w_mu);
            fresh = raised_w.insert(reinterpret_cast<const void *>(kern)).second;
        }
        if (fresh)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3(pb_grid(groups, nwv, g_scatter_wgs > 0 ? g_scatter_wgs : 256 * per_cu)), dim3(nwv * 64), smem, st, groups, gout,
                           argmax, zsel, coef, relu, wT, da, pb);
    };
    if (g_scatter_form == 1 || hc) {
#define VN_SCATTER_WAVE(CI, CO, KK, NW)                                                            \
    do {                                                                                           \
        if (below_z) gow(pool_dgrad_scatter_wave_kernel<CI, CO, KK, true, NW>, CI, CO, KK, NW);   \
        else gow(pool_dgrad_scatter_wave_kernel<CI, CO, KK, false, NW>, CI, CO, KK, NW);          \
    } while (0)
        if (hc) {
            if (cin == 64) VN_SCATTER_WAVE(64, 128, kPiece, 8);
            else if (cout == 256) {
                // (round 5: with 16 wavefronts the kernel is held to 128 VGPRs; it spilled 19 of them -- a reload inside the group loop is a
                // scratch load with an s_waitcnt vmcnt(0) behind it -- until the prefetch loads lost their branches (1 spill now).  12
                // wavefronts, 132 VGPRs, no spill: measured equal, hook kept)
                if (g_scatter_nwv == 12) VN_SCATTER_WAVE(128, 256, kPiece, 12);
                else VN_SCATTER_WAVE(128, 256, kPiece, 16);
            }
            else VN_SCATTER_WAVE(128, 128, kPiece, 8);
        } else {
            // (64-row groups: the row lists need 8 wavefronts' scratch beside W^T)
            if (cin == 64) VN_SCATTER_WAVE(64, 128, 64, 8);
            else if (cout == 256) VN_SCATTER_WAVE(128, 256, 64, 8);
            else VN_SCATTER_WAVE(128, 128, 64, 8);
        }
#undef VN_SCATTER_WAVE
        return check_launch("pool_dgrad_scatter");
    }
    if (below_z) {
        if (cin == 128 && cout == 256)
            go(pool_dgrad_scatter_kernel<128, 256, 64, true, 16>, 128, 256, 1024);
        else if (cin == 128)
            go(pool_dgrad_scatter_kernel<128, 128, 64, true>, 128, 128);
        else
            go(pool_dgrad_scatter_kernel<64, 128, 64, true>, 64, 128);
    } else {
        if (cin == 128 && cout == 256)
            go(pool_dgrad_scatter_kernel<128, 256, 64, false, 16>, 128, 256, 1024);
        else if (cin == 128)
            go(pool_dgrad_scatter_kernel<128, 128, 64, false>, 128, 128);
        else
            go(pool_dgrad_scatter_kernel<64, 128, 64, false>, 64, 128);
    }
    return check_launch("pool_dgrad_scatter");
}

// ---- Gram matrix on bf16 x 3 split operands ---------------------------------------------------------------------------------------
// G (C x C) += a^T a, a = act(z * scale + shift), over the rows of a workgroup's range.  The contraction runs over the ROWS, so an
// MFMA fragment is 8 consecutive rows of one channel: thread (channel c, row group) loads its KPT = C / 16 rows of the slab as
// scalars (coalesced over c), applies the activation with ITS channel's scale / shift, splits the values exactly into three bf16
// pieces (split3, mlp_types.h) and writes them as one 16-byte (8-byte at C = 64) row of the LDS image [piece][row-half][channel][8
// rows] -- which is the A^T AND the B operand image: both operands of a^T a are the same matrix.  Six v_mfma_f32_32x32x16_bf16
// per 32 x 32 sub-tile and 16-row slab instead of eight v_mfma_f32_32x32x2_f32 of twice the length; two register sets of raw
// rows in flight, LDS double-buffered, one barrier per slab.  Partial tiles are added with atomics (the deterministic mode keeps
// the fp32 kernel with its ordered reduction).
#ifndef GRAM_ABL
#define GRAM_ABL 0 // probe builds only (tools/probe/ablate_half.sh): 1 no MFMAs, 2 no epilogue (atomics), 4 no global loads after the prologue,
                   // 8 no staging (activation, split, LDS writes) after the prologue -- results wrong by construction, only the time is read
#endif
// NP = 2 (round 6, the default): the same on fp16 x 2 pieces (mlp_types.h: split2) -- both operands are ACTIVATIONS behind a BatchNorm,
// inside fp16's range; staged scaled by 2^4 (folded into scale / shift: no instruction), the tile scaled back by 2^-8 before its
// atomics; three v_mfma_f32_32x32x16_f16 per sub-tile and slab, two LDS planes per operand instead of three.
template <int C, int NP = 3>
__global__ __launch_bounds__(256) void gram_bf3_kernel(long rows, const float *__restrict__ x, const float *__restrict__ scale_shift,
                                                       int relu, float *__restrict__ gram, long rows_per_block,
                                                       const float *__restrict__ wh /* piece layout: row 16 q counts wh[q] times */,
                                                       const int *__restrict__ nh_dev /* or NULL: rows is exact */)
{
    if (nh_dev != nullptr) {
        const long lim = (long)nh_dev[0] * kPiece;
        rows = lim < rows ? lim : rows;
    }
    constexpr int KPT = C / 16;          // rows of a slab per thread (8 or 4)
    constexpr int T = C / 64;            // 32 x 32 sub-tiles per wave and direction (waves 2 x 2)
    constexpr int PL = C * 4 + 16;       // dwords per (piece, row-half) plane; the pad keeps a wave's two halves on different banks
    static_assert(NP == 2 || NP == 3, "pieces per operand");
    constexpr float kA = NP == 2 ? 16.0f : 1.0f, kUn = NP == 2 ? 1.0f / 256.0f : 1.0f;
    __shared__ __attribute__((aligned(16))) unsigned Ts[2][NP][2][PL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    if (r_begin >= rows) return;
    const int nrow = (int)((r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows) - r_begin);
    const int nslab = (nrow + 15) / 16;
    const int c = tid % C, rg = tid / C; // C = 128: rg = row-half; C = 64: rg = quarter (row-half rg >> 1, dword pair rg & 1)
    const int kh_w = (C == 128) ? rg : (rg >> 1), d_w = (C == 128) ? 0 : (rg & 1) * 2;
    const float sc = scale_shift[c] * kA, sh = scale_shift[C + c] * kA;
    const float floor_ = relu ? 0.0f : -__builtin_inff();
    const float *xb = x + (size_t)r_begin * C + c;
    float R[2][KPT];
    // a slab is a piece (16 rows, r_begin % 16 == 0): a thread's first row is row 0 of the piece exactly when its row group is the first
    static_assert(kPiece == 16, "one slab = one piece");
    const bool first_rg = rg == 0;
    const float *whb = wh ? wh + r_begin / kPiece : nullptr;
    const float *whp = whb ? whb : scale_shift; // (a valid address either way: no load under a branch)
    float Wq[2] = {1.0f, 1.0f}; // sqrt of that row's weight, travelling with the register set
    bool abl_prologue = true;
    auto load = [&](float (&r)[KPT], int s, float &wq) {
        if ((GRAM_ABL & 4) && !abl_prologue) return;
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            int lr = s * 16 + rg * KPT + i;
            lr = lr < nrow ? lr : nrow - 1; // past the end: a valid row, stored as zero below
            r[i] = xb[(size_t)lr * C];
        }
        // the piece's weight travels RAW with the register set and every thread loads it (round 5): under `if (whb && first_rg)` with the
        // square root right behind it this was a conditional load followed by s_waitcnt vmcnt(0) -- it drained the eight row loads just
        // issued, once per 16-row slab: the kernel ran at one exposed memory latency per slab
        // (no weight array -- the full-row layout -- reads element 0 of scale_shift: always inside that array, never used)
        const int sw = (whb != nullptr && s * 16 < nrow) ? s : 0;
        wq = whp[sw];
    };
    auto store = [&](int buf, const float (&r)[KPT], int s, float wq) {
        if ((GRAM_ABL & 8) && !abl_prologue) return;
        float v[KPT];
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            v[i] = fmaxf(r[i] * sc + sh, floor_);
            if (s * 16 + rg * KPT + i >= nrow) v[i] = 0.0f; // padding rows contribute nothing
        }
        if (whb != nullptr && first_rg && s * 16 < nrow) v[0] *= sqrtf(wq); // a^T diag(w) a = (sqrt(w) a)^T (sqrt(w) a)
        if constexpr (NP == 2) {
            unsigned h[KPT / 2], l[KPT / 2];
#pragma unroll
            for (int i = 0; i < KPT / 2; i++) split2(v[2 * i], v[2 * i + 1], h[i], l[i]);
            if constexpr (C == 128) {
                *reinterpret_cast<uint4 *>(&Ts[buf][0][kh_w][c * 4]) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4 *>(&Ts[buf][1][kh_w][c * 4]) = make_uint4(l[0], l[1], l[2], l[3]);
            } else {
                *reinterpret_cast<uint2 *>(&Ts[buf][0][kh_w][c * 4 + d_w]) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2 *>(&Ts[buf][1][kh_w][c * 4 + d_w]) = make_uint2(l[0], l[1]);
            }
            return;
        }
        unsigned h[KPT / 2], m[KPT / 2], l[KPT / 2];
#pragma unroll
        for (int i = 0; i < KPT / 2; i++) split3(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
        if constexpr (C == 128) {
            *reinterpret_cast<uint4 *>(&Ts[buf][0][kh_w][c * 4]) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4 *>(&Ts[buf][1][kh_w][c * 4]) = make_uint4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<uint4 *>(&Ts[buf][NP - 1][kh_w][c * 4]) = make_uint4(l[0], l[1], l[2], l[3]);
        } else {
            *reinterpret_cast<uint2 *>(&Ts[buf][0][kh_w][c * 4 + d_w]) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2 *>(&Ts[buf][1][kh_w][c * 4 + d_w]) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2 *>(&Ts[buf][NP - 1][kh_w][c * 4 + d_w]) = make_uint2(l[0], l[1]);
        }
    };
    f32x16 acc[T][T];
#pragma unroll
    for (int a = 0; a < T; a++)
#pragma unroll
        for (int b = 0; b < T; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[a][b][e] = 0.0f;
    // prologue: slab 0 -> buffer 0; slabs 1 and 2 in flight
    load(R[0], 0, Wq[0]);
    store(0, R[0], 0, Wq[0]);
    load(R[1], 1, Wq[1]);
    __builtin_amdgcn_sched_barrier(0);
    load(R[0], 2, Wq[0]);
    __syncthreads();
    const int kh = lane >> 5, l31 = lane & 31;
    int buf = 0;
    abl_prologue = false;
    const int nslab2 = (nslab + 1) & ~1; // the loop runs slab pairs; a padding slab multiplies zeros
    for (int s = 0; s < nslab2; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; par++) {
            uint4 fa[NP][T], fb[NP][T];
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int t = 0; t < T; t++) {
                    fa[p][t] = *reinterpret_cast<const uint4 *>(&Ts[buf][p][kh][((wi * T + t) * 32 + l31) * 4]);
                    fb[p][t] = *reinterpret_cast<const uint4 *>(&Ts[buf][p][kh][((wj * T + t) * 32 + l31) * 4]);
                }
            }
            auto mm = [&](int pa, int pb) {
                if (GRAM_ABL & 1) {
                    acc[0][0][0] += __uint_as_float(fa[pa][0].x ^ fb[pb][0].y);
                    return;
                }
#pragma unroll
                for (int a = 0; a < T; a++)
#pragma unroll
                    for (int b = 0; b < T; b++) {
                        if constexpr (NP == 2)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[pa][a]),
                                                                               __builtin_bit_cast(f16x8, fb[pb][b]), acc[a][b], 0, 0, 0);
                        else
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[pa][a]),
                                                                                __builtin_bit_cast(bf16x8, fb[pb][b]), acc[a][b], 0, 0, 0);
                    }
            };
            if constexpr (NP == 2) {
                mm(1, 0);
                mm(0, 1);
                store(buf ^ 1, R[par ^ 1], s + par + 1, Wq[par ^ 1]); // the other buffer was last read one slab ago, behind a barrier
                load(R[par ^ 1], s + par + 3, Wq[par ^ 1]);
                mm(0, 0);
            } else {
                mm(2, 0);
                mm(0, 2);
                mm(1, 1);
                store(buf ^ 1, R[par ^ 1], s + par + 1, Wq[par ^ 1]); // the other buffer was last read one slab ago, behind a barrier
                load(R[par ^ 1], s + par + 3, Wq[par ^ 1]);
                mm(1, 0);
                mm(0, 1);
                mm(0, 0);
            }
            lds_barrier();
            buf ^= 1;
        }
    }
    if (GRAM_ABL & 2) {
        float t_ = 0.0f;
#pragma unroll
        for (int a = 0; a < T; a++)
#pragma unroll
            for (int b = 0; b < T; b++)
#pragma unroll
                for (int e = 0; e < 16; e++) t_ += acc[a][b][e];
        if (t_ == 12345.678f) gram[0] = t_;
        return;
    }
    // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < T; a++)
#pragma unroll
        for (int b = 0; b < T; b++) {
            const int j = (wj * T + b) * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = (wi * T + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                unsafeAtomicAdd(&gram[(size_t)i * C + j], acc[a][b][e] * kUn);
            }
        }
}

int g_gram_bf3 = 1; // votenet_debug_gram_bf3: 0 = the fp32 MFMA kernel always, 1 (default) = split operands as fp16 x 2 pieces, 3 = as bf16 x 3 pieces
static int g_gram_wgs = 384; // votenet_debug_gram_workgroups (tuning hook)
template <int C>
static void gram_bf3_launch(long rows, const float *z, const float *scale_shift, int relu, float *gram, hipStream_t st, const float *wh = nullptr,
                            const int *nh_dev = nullptr)
{
    // 384 workgroups as the fp32 weight-gradient kernels (mlp_wgrad_fast.hip, plan_fast): the launch runs beside the input-gradient chain
    long rpb = (rows + g_gram_wgs - 1) / g_gram_wgs;
    rpb = (rpb + 31) / 32 * 32;
    if (rpb < 128) rpb = 128;
    const unsigned gx = (unsigned)((rows + rpb - 1) / rpb);
    if (g_gram_bf3 != 3) hipLaunchKernelGGL((gram_bf3_kernel<C, 2>), dim3(gx), dim3(256), 0, st, rows, z, scale_shift, relu, gram, rpb, wh, nh_dev);
    else hipLaunchKernelGGL((gram_bf3_kernel<C, 3>), dim3(gx), dim3(256), 0, st, rows, z, scale_shift, relu, gram, rpb, wh, nh_dev);
}
extern "C" void votenet_debug_gram_bf3(int on) { VN_DEBUG_GATE(); g_gram_bf3 = on; }
extern "C" void votenet_debug_gram_workgroups(int n) { VN_DEBUG_GATE(); g_gram_wgs = n > 0 ? n : 384; }

extern "C" int votenet_mlp_gram(long rows, int c, const float *z, const float *scale_shift, int relu, float *gram, float *scratch,
                                void *stream)
{
    if (g_gram_bf3 && scratch == nullptr && (c == 64 || c == 128) && rows > 0 && rows < (1L << 31) / c && z && scale_shift && gram &&
        (uintptr_t)z % 16 == 0) {
        if (c == 128) gram_bf3_launch<128>(rows, z, scale_shift, relu, gram, as_stream(stream));
        else gram_bf3_launch<64>(rows, z, scale_shift, relu, gram, as_stream(stream));
        return check_launch("mlp_gram");
    }
    VN_REQUIRE(rows > 0 && c > 0 && z && scale_shift && gram, "mlp_gram: bad arguments");
    MlpIn d = {};
    d.x = z;
    d.in_scale = scale_shift;
    d.in_shift = scale_shift + c;
    d.in_relu = relu;
    BnSrc bs = {};
    bs.z = z;
    bs.coef = scale_shift;
    bs.relu = relu;
    VN_REQUIRE(wgrad_fast_launch(0, d, rows, c, c, nullptr, bs, 3, gram, as_stream(stream), scratch),
               "mlp_gram: shape not served (c %% 64 == 0, 16-byte aligned operands), got c = %d", c);
    return check_launch("mlp_gram");
}

static int g_sparse_wgs = 384, g_sparse_wgs2 = 256, g_sparse_teams = 2;
extern "C" void votenet_debug_sparse_workgroups(int n) { VN_DEBUG_GATE(); g_sparse_wgs = n > 0 ? n : 384; } // tuning hook
extern "C" void votenet_debug_sparse_teams(int teams, int wgs) // tuning hook: 1 or 2 teams per workgroup (piece layout), workgroups of the 2-team form
{
    VN_DEBUG_GATE();
    g_sparse_teams = teams == 1 ? 1 : 2;
    if (wgs > 0) g_sparse_wgs2 = wgs;
}
static int pool_wgrad_sparse_launch(long groups, int cin, int cout, const float *xz, const float *in_scale, const float *in_shift, int in_relu,
                                    const float *gout, const int *argmax, const float *zsel, const float *coef, int relu, float *dw,
                                    float *colsum, float *scratch, const int *hc, const float *wh, int G, void *stream, const int *nh_dev = nullptr);
extern "C" int votenet_pool_wgrad_sparse(long groups, int k, int cin, int cout, const float *xz, const float *in_scale,
                                         const float *in_shift, int in_relu, const float *gout, const int *argmax, const float *zsel,
                                         const float *coef, int relu, float *dw, float *colsum, float *scratch, void *stream)
{
    VN_REQUIRE(groups > 0 && xz && gout && argmax && zsel && coef && dw && colsum, "pool_wgrad_sparse: bad arguments");
    VN_REQUIRE(cin <= cout, "pool_wgrad_sparse expects cin <= cout");
    VN_REQUIRE(votenet_pool_backward_supported(cin, cout, k), "pool_wgrad_sparse: unsupported shape cin=%d cout=%d k=%d", cin, cout, k);
    VN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pool_wgrad_sparse: in_scale and in_shift go together");
    VN_REQUIRE((uintptr_t)xz % 16 == 0 && (!in_scale || ((uintptr_t)in_scale % 16 == 0 && (uintptr_t)in_shift % 16 == 0)),
               "pool_wgrad_sparse: operands must be 16-byte aligned");
    return pool_wgrad_sparse_launch(groups, cin, cout, xz, in_scale, in_shift, in_relu, gout, argmax, zsel, coef, relu, dw, colsum, scratch,
                                    nullptr, nullptr, 0, stream);
}

// The same on the piece layout (half.hip): xz has 16 * nh compact rows, gout / argmax / zsel stay per centre.
// votenet_mlp_gram over the piece layout (half.hip): G += a^T diag(w) a with w = wh[q] on row 16 q, 1 elsewhere.
extern "C" int votenet_mlp_gram_half(long rows, int c, const float *z, const float *scale_shift, int relu, const float *wh, float *gram,
                                     const int *nh_dev, void *stream)
{
    VN_REQUIRE(rows > 0 && rows % kPiece == 0 && rows < (1L << 31) / c && (c == 128 || c == 64), "mlp_gram_half expects rows %% 16 == 0 and c in {64, 128}");
    VN_REQUIRE(z && scale_shift && wh && gram && (uintptr_t)z % 16 == 0, "mlp_gram_half: null / unaligned buffer");
    if (c == 128) gram_bf3_launch<128>(rows, z, scale_shift, relu, gram, as_stream(stream), wh, nh_dev);
    else gram_bf3_launch<64>(rows, z, scale_shift, relu, gram, as_stream(stream), wh, nh_dev);
    return check_launch("mlp_gram_half");
}

extern "C" int votenet_pool_wgrad_sparse_half(long nh, int G, int cin, int cout, const float *xz, const float *in_scale, const float *in_shift,
                                              int in_relu, const float *gout, const int *argmax, const float *zsel, const float *coef, int relu,
                                              float *dw, float *colsum, const int *hc, const float *wh, const int *nh_dev, void *stream)
{
    VN_REQUIRE(nh > 0 && G > 0 && nh >= G && nh <= (long)kBallPieces * G && hc && wh, "pool_wgrad_sparse_half: bad piece-layout arguments");
    VN_REQUIRE(xz && gout && argmax && zsel && coef && dw && colsum, "pool_wgrad_sparse_half: bad arguments");
    VN_REQUIRE(votenet_pool_backward_supported(cin, cout, 64), "pool_wgrad_sparse_half: unsupported shape cin=%d cout=%d", cin, cout);
    VN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pool_wgrad_sparse_half: in_scale and in_shift go together");
    VN_REQUIRE((uintptr_t)xz % 16 == 0 && (!in_scale || ((uintptr_t)in_scale % 16 == 0 && (uintptr_t)in_shift % 16 == 0)),
               "pool_wgrad_sparse_half: operands must be 16-byte aligned");
    return pool_wgrad_sparse_launch(nh, cin, cout, xz, in_scale, in_shift, in_relu, gout, argmax, zsel, coef, relu, dw, colsum, nullptr, hc, wh,
                                    G, stream, nh_dev);
}

static int g_sparse_centre_wgs = 192, g_sparse_centre_teams = 2; // votenet_debug_sparse_centre_workgroups (tuning hook)
extern "C" void votenet_debug_sparse_centre_workgroups(int n) { VN_DEBUG_GATE(); g_sparse_centre_wgs = n > 0 ? n : 192; }
extern "C" void votenet_debug_sparse_centre_teams(int t) { VN_DEBUG_GATE(); g_sparse_centre_teams = t == 1 ? 1 : 2; }
// votenet_pool_wgrad_sparse_half walking CENTRES instead of pieces (pool_wgrad_sparse_centre_kernel): pos = the layout's (G, 3) table of
// the pieces j >= 1 of every centre (votenet_half_groups), wh[0:G] the weights of the balls' slot 0.  Same results up to the association
// of the sums.
extern "C" int votenet_pool_wgrad_sparse_half_centres(long nh, int G, int cin, int cout, const float *xz, const float *in_scale,
                                                      const float *in_shift, int in_relu, const float *gout, const int *argmax,
                                                      const float *zsel, const float *coef, int relu, float *dw, float *colsum,
                                                      const int *pos, const float *wh, const int *nh_dev, void *stream)
{
    VN_REQUIRE(nh > 0 && G > 0 && nh >= G && nh <= (long)kBallPieces * G && pos && wh, "pool_wgrad_sparse_half_centres: bad piece-layout arguments");
    VN_REQUIRE(xz && gout && argmax && zsel && coef && dw && colsum, "pool_wgrad_sparse_half_centres: bad arguments");
    VN_REQUIRE(votenet_pool_backward_supported(cin, cout, 64), "pool_wgrad_sparse_half_centres: unsupported shape cin=%d cout=%d", cin, cout);
    VN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "pool_wgrad_sparse_half_centres: in_scale and in_shift go together");
    // the kernel addresses xz through a buffer descriptor with 32-bit byte offsets and uses offset 2^31 as its "piece not kept" marker:
    // above this size call votenet_pool_wgrad_sparse_half (the piece-walking form, 64-bit addressing)
    VN_REQUIRE((size_t)nh * kPiece * (size_t)cin * 4 < ((size_t)1 << 31),
               "pool_wgrad_sparse_half_centres: nh*16*cin*4 = %zu bytes exceeds the 2 GiB the centre-walking kernel addresses; "
               "use votenet_pool_wgrad_sparse_half", (size_t)nh * kPiece * (size_t)cin * 4);
    VN_REQUIRE((uintptr_t)xz % 16 == 0 && (!in_scale || ((uintptr_t)in_scale % 16 == 0 && (uintptr_t)in_shift % 16 == 0)),
               "pool_wgrad_sparse_half_centres: operands must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int teams = g_sparse_centre_teams;
    auto go = [&](auto kern, int ci) {
        const size_t tiles = (size_t)teams * 2 * kBallPieces * kPiece * (ci + 4) * 4;
        const size_t comb = teams > 1 ? (size_t)(ci + 1) * cout * 4 : 0;
        const size_t smem = tiles > comb ? tiles : comb;
        static std::set<const void *> raised_c;
        static std::mutex raised_c_mu;
        bool fresh;
        {
            std::lock_guard<std::mutex> lock(raised_c_mu);
            fresh = raised_c.insert(reinterpret_cast<const void *>(kern)).second;
        }
        if (fresh) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        const int grid = pb_grid(G, 8 * teams, g_sparse_centre_wgs * 2 / teams);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * teams), smem, st, G, nh, cout, xz, in_scale, in_shift, in_relu, gout, argmax, zsel, coef, relu, dw,
                           colsum, pos, wh, nh_dev);
    };
    if (teams == 1) {
        if (cin == 64) go(pool_wgrad_sparse_centre_kernel<64, 1>, 64);
        else go(pool_wgrad_sparse_centre_kernel<128, 1>, 128);
    } else if (cin == 64) go(pool_wgrad_sparse_centre_kernel<64, 2>, 64);
    else go(pool_wgrad_sparse_centre_kernel<128, 2>, 128);
    return check_launch("pool_wgrad_sparse_half_centres");
}

static int pool_wgrad_sparse_launch(long groups, int cin, int cout, const float *xz, const float *in_scale, const float *in_shift, int in_relu,
                                    const float *gout, const int *argmax, const float *zsel, const float *coef, int relu, float *dw,
                                    float *colsum, float *scratch, const int *hc, const float *wh, int G, void *stream, const int *nh_dev)
{
    hipStream_t st = as_stream(stream);
    int grid;
    auto go = [&](auto kern, int ci, int kk, int teams, int cap) {
        const size_t tiles = (size_t)teams * (kk == kPiece ? 2 : 1) * kk * (ci + 4) * 4;
        const size_t comb = teams > 1 ? (size_t)(ci + 1) * cout * 4 : 0;
        const size_t smem = tiles > comb ? tiles : comb;
        static std::set<const void *> raised_s;
        static std::mutex raised_s_mu;
        bool fresh;
        {
            std::lock_guard<std::mutex> lock(raised_s_mu);
            fresh = raised_s.insert(reinterpret_cast<const void *>(kern)).second;
        }
        if (fresh) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        grid = pb_grid(groups, (kk == kPiece ? 32 : 8) * teams, cap);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * teams), smem, st, groups, cout, xz, in_scale, in_shift, in_relu, gout, argmax, zsel, coef,
                           relu, dw, colsum, scratch, hc, wh, G, nh_dev);
    };
    // off the critical chain (weight-gradient stream): the caps leave CUs to the chain beside it
    if (hc && g_sparse_teams == 2 && !scratch) {
        if (cin == 64) go(pool_wgrad_sparse_kernel<64, kPiece, 2>, 64, kPiece, 2, g_sparse_wgs2);
        else go(pool_wgrad_sparse_kernel<128, kPiece, 2>, 128, kPiece, 2, g_sparse_wgs2);
    } else if (hc) {
        if (cin == 64) go(pool_wgrad_sparse_kernel<64, kPiece, 1>, 64, kPiece, 1, g_sparse_wgs);
        else go(pool_wgrad_sparse_kernel<128, kPiece, 1>, 128, kPiece, 1, g_sparse_wgs);
    } else if (cin == 128) {
        go(pool_wgrad_sparse_kernel<128, 64, 1>, 128, 64, 1, g_sparse_wgs);
    } else {
        go(pool_wgrad_sparse_kernel<64, 64, 1>, 64, 64, 1, g_sparse_wgs);
    }
    if (scratch) { // ordered reduction of the workgroups' slices: dW rows, then the column sums (row cin of a slice)
        const long ps = (long)(cin + 1) * cout;
        wgrad_reduce(grid, ps, 0, (long)cin * cout, scratch, dw, st);
        wgrad_reduce(grid, ps, (long)cin * cout, (long)cin * cout + cin, scratch, colsum - (long)cin * cout, st);
    }
    return check_launch("pool_wgrad_sparse");
}

extern "C" size_t votenet_pool_wgrad_scratch_floats(long groups, int cin, int cout)
{
    if (groups <= 0) return 0;
    return (size_t)pb_grid(groups, 8, g_sparse_wgs) * (size_t)(cin + 1) * cout;
}

extern "C" int votenet_pool_wgrad_finish(int cin, int cout, const float *gram, const float *colsum, const float *w, const float *bias,
                                         const float *coef, float *dw, void *stream)
{
    VN_REQUIRE(cin > 0 && cout > 0 && gram && colsum && w && coef && dw, "pool_wgrad_finish: bad arguments");
    hipLaunchKernelGGL(pool_wgrad_finish_kernel, dim3((cout + 15) / 16, (cin + 15) / 16), dim3(256), 0, as_stream(stream), cin, cout,
                       gram, colsum, w, bias, coef, dw);
    return check_launch("pool_wgrad_finish");
}
