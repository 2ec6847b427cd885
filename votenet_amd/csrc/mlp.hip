// mlp.hip -- grouped-point MLP for gfx950: fp32 MFMA GEMM with a fused gather prologue and a
// BatchNorm-statistics epilogue, plus the small BN / max-pool kernels around it.
//
// Replaces the SA/FP-layer MLP path of the reference (utils.py:50-57 sample_and_group concat,
// :125-127 Conv2D 1x1 + BNReLU loop, :132 max-pool, :149-155 mlp2, :286-293 FP MLP), which the
// reference runs as cuDNN 1x1 convolutions over a MATERIALISED (B,m,K,3+C) tensor with
// BN / ReLU / max as separate graph nodes.  Here:
//   * the (B*m*K, 3+C) input matrix never exists: the GEMM's A-operand loader gathers
//     feat[idx[r]] rows and forms xyz[idx[r]] - new_xyz[r/K] on the fly (GATHER mode);
//   * BNReLU of layer l is folded into the A-operand load of layer l+1 (DENSE mode with
//     per-channel scale/shift + relu), so activations are written once, raw, and read once;
//   * the epilogue accumulates the per-channel sum / sum of squares that BatchNorm needs
//     (fp32 partials per workgroup, fp64 atomics across workgroups);
//   * math runs on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, exact fp32 products --
//     the reference computes this path in fp32); 128-row x 64/128-column workgroup tiles,
//     4 waves, operands staged through LDS in [k][row] order so both MFMA operand reads are
//     conflict-free ds_read_b32, global loads of the next k-slab in flight during the MFMAs.
//
// Internal K order of GATHER mode is [feat(c), dxyz(3)] (features first, so feature rows load
// as aligned float4); row k of the caller's W (whose order is the reference's [dxyz, feat],
// utils.py:55) is fetched through the same permutation, so the result is the reference's.
#include "mlp_types.h"

namespace votenet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MLP_MIN_WAVES
#define MLP_MIN_WAVES 3
#endif
constexpr int MLP_BM = 128; // rows per workgroup tile
#ifndef MLP_BK_DEF
#define MLP_BK_DEF 16
#endif
constexpr int MLP_BK = MLP_BK_DEF;  // k-slab
constexpr int MLP_KQ = MLP_BK / 4;        // float4 per staged row
constexpr int MLP_RPP = 256 / MLP_KQ;     // rows staged per pass of the 256 threads
constexpr int MLP_AP = MLP_BM / MLP_RPP;  // passes (float4 per thread) for the A slab
constexpr int MLP_LDA = MLP_BM + 2; // [k][row] image; +2 -> conflict-free 4-lane-strided writes
constexpr int MLP_MAXC = 512;       // input channels whose folded BN scale/shift are staged in LDS


// One element of the implicit A matrix, internal k order.  Bounds are the caller's job.
template <int MODE>
__device__ __forceinline__ float a_elem(const MlpIn &in, long r, int k, int cin, int src /*gather: idx[r]*/, long scene)
{
    if (MODE == 0) {
        float v = in.x[(size_t)r * cin + k];
        if (in.in_scale) {
            v = v * in.in_scale[k] + in.in_shift[k];
            if (in.in_relu) v = v > 0.0f ? v : 0.0f;
        }
        return v;
    } else {
        if (k < in.c) return in.feat[((size_t)scene * in.n + src) * in.c + k];
        const int d = k - in.c;
        return in.xyz[((size_t)scene * in.n + src) * 3 + d] - in.new_xyz[(size_t)(r / in.nsample) * 3 + d]; // utils.py:51
    }
}

// caller's W row for internal k (GATHER: features first internally, dxyz first in W)
template <int MODE>
__device__ __forceinline__ int w_row(int k, int c)
{
    if (MODE == 0) return k;
    return k < c ? k + 3 : k - c;
}

// z = A(rows x cin) * W(cin x cout) + bias, stats += column sums of z and z^2.
// WM x WN waves, each wave MT x NT tiles of 32x32.  BM = WM*MT*32 = 128, BN = WN*NT*32.
// FAST: dense input, cin % 16 == 0, cout % BN == 0, rows % 128 == 0, 16-byte aligned operands -- no bounds
// checks, operand addresses advance by pointer increments (every dense VoteNet layer qualifies).
template <int MODE, int WM, int WN, int MT, int NT, bool FAST>
__global__ __launch_bounds__(256, (MT * NT >= 4 ? 2 : 3)) void mlp_linear_kernel(MlpIn in, long rows, int cin, int cout,
                                                         const float *__restrict__ w, const float *__restrict__ bias,
                                                         float *__restrict__ z, double *__restrict__ stats)
{
    static_assert(WM * WN == 4 && WM * MT * 32 == MLP_BM, "tile shape");
    constexpr int BN = WN * NT * 32;
    constexpr int LDB = BN + 4;
    __shared__ float As[2][MLP_BK][MLP_LDA];
    __shared__ float Bs[2][MLP_BK][LDB];
    __shared__ __attribute__((aligned(16))) float Ssc[MLP_MAXC], Ssh[MLP_MAXC]; // folded BN of the previous layer

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool affine = (MODE == 0) && in.in_scale != nullptr;
    const bool affine_lds = affine && cin <= MLP_MAXC;
    if (affine_lds)
        for (int k = tid; k < cin; k += 256) {
            Ssc[k] = in.in_scale[k];
            Ssh[k] = in.in_shift[k];
        }
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const int n0 = blockIdx.y * BN;
    const int nk = (cin + MLP_BK - 1) / MLP_BK;
    const long ntiles = (rows + MLP_BM - 1) / MLP_BM;
    const bool a_vec4 = (MODE == 0) ? ((cin & 3) == 0) : ((in.c & 3) == 0 && in.c > 0);
    const bool b_vec4 = ((cout & 3) == 0);

    // A staging: thread t loads rows (t>>2) and (t>>2)+64 of the tile, k-quad (t&3) of the slab
    const int a_row = tid / MLP_KQ, a_kq = tid % MLP_KQ;
    // B staging: BK x BN floats = 4*BN float4; thread t loads float4 #t and #t+256 (if BN=128)
    constexpr int B_F4 = MLP_BK * BN / 4;       // 256 (BN=64) or 512 (BN=128)
    constexpr int B_PER_T = B_F4 / 256;         // 1 or 2

    float s1[NT], s2[NT]; // per-lane partial column sums over all tiles of this workgroup
#pragma unroll
    for (int j = 0; j < NT; j++) s1[j] = s2[j] = 0.0f;

    // The (tile, k-slab) iteration space is flattened into one sequence of steps so that the software
    // pipeline never drains between row tiles: while step s runs its MFMAs, the operands of step s+1
    // (possibly the first slab of the NEXT tile) are loaded to registers, and they are written to the
    // other LDS buffer half way through the MFMAs -- one barrier per step, no exposed global latency.
    long ar[MLP_AP];     // rows this thread stages for the step being PREFETCHED
    int asrc[MLP_AP];
    long ascene[MLP_AP];
    const float *pa[MLP_AP]; // FAST: this thread's A pointers for the step being prefetched
    const float *pb[B_PER_T];
    auto set_rows = [&](long tile) {
        const long m0p = tile * MLP_BM;
        if constexpr (FAST) {
#pragma unroll
            for (int h = 0; h < MLP_AP; h++) pa[h] = in.x + (size_t)(m0p + a_row + h * MLP_RPP) * cin + a_kq * 4;
#pragma unroll
            for (int u = 0; u < B_PER_T; u++) {
                const int f = tid + u * 256;
                pb[u] = w + (size_t)(f / (BN / 4)) * cout + n0 + (f % (BN / 4)) * 4;
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < MLP_AP; h++) {
            ar[h] = m0p + a_row + h * MLP_RPP;
            asrc[h] = 0;
            ascene[h] = 0;
            if (MODE == 1 && ar[h] < rows) {
                asrc[h] = in.idx[ar[h]];
                ascene[h] = ar[h] / ((long)in.m * in.nsample);
            }
        }
    };
    float4 ra[MLP_AP];
    float4 rb[B_PER_T];
    int rk = 0; // first k of the A quad held in ra (FAST path: needed by the deferred BN+ReLU)
    auto load_slab = [&](int kt) {
        if constexpr (FAST) {
            const int kq = kt * MLP_BK + a_kq * 4;
#pragma unroll
            for (int h = 0; h < MLP_AP; h++) {
                ra[h] = *reinterpret_cast<const float4 *>(pa[h]); // raw: the folded BN+ReLU is applied when the slab
                pa[h] += MLP_BK;                                  // is written to LDS, so the load stays in flight
            }
            rk = kq;
#pragma unroll
            for (int u = 0; u < B_PER_T; u++) {
                rb[u] = *reinterpret_cast<const float4 *>(pb[u]);
                pb[u] += (size_t)MLP_BK * cout;
            }
            return;
        }
        const int k0 = kt * MLP_BK + a_kq * 4;
#pragma unroll
        for (int h = 0; h < MLP_AP; h++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ar[h] < rows) {
                bool done = false;
                if (a_vec4) {
                    if (MODE == 0 && k0 + 3 < cin) {
                        v = *reinterpret_cast<const float4 *>(in.x + (size_t)ar[h] * cin + k0);
                        if (affine) {
                            const float4 sc = affine_lds ? *reinterpret_cast<const float4 *>(&Ssc[k0])
                                                         : *reinterpret_cast<const float4 *>(in.in_scale + k0);
                            const float4 sh = affine_lds ? *reinterpret_cast<const float4 *>(&Ssh[k0])
                                                         : *reinterpret_cast<const float4 *>(in.in_shift + k0);
                            v.x = v.x * sc.x + sh.x;
                            v.y = v.y * sc.y + sh.y;
                            v.z = v.z * sc.z + sh.z;
                            v.w = v.w * sc.w + sh.w;
                            if (in.in_relu) {
                                v.x = v.x > 0.f ? v.x : 0.f;
                                v.y = v.y > 0.f ? v.y : 0.f;
                                v.z = v.z > 0.f ? v.z : 0.f;
                                v.w = v.w > 0.f ? v.w : 0.f;
                            }
                        }
                        done = true;
                    } else if (MODE == 1 && k0 + 3 < in.c) {
                        v = *reinterpret_cast<const float4 *>(in.feat + ((size_t)ascene[h] * in.n + asrc[h]) * in.c + k0);
                        done = true;
                    }
                }
                if (!done) {
                    float t[4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        t[q] = (k0 + q < cin) ? a_elem<MODE>(in, ar[h], k0 + q, cin, asrc[h], ascene[h]) : 0.0f;
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            ra[h] = v;
        }
#pragma unroll
        for (int u = 0; u < B_PER_T; u++) {
            const int f = tid + u * 256;
            const int kk = f / (BN / 4), nq = f % (BN / 4);
            const int k = kt * MLP_BK + kk;
            const int nn = n0 + nq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < cin) {
                const float *wr = w + (size_t)w_row<MODE>(k, in.c) * cout;
                if (b_vec4 && nn + 3 < cout) {
                    v = *reinterpret_cast<const float4 *>(wr + nn);
                } else {
                    if (nn + 0 < cout) v.x = wr[nn + 0];
                    if (nn + 1 < cout) v.y = wr[nn + 1];
                    if (nn + 2 < cout) v.z = wr[nn + 2];
                    if (nn + 3 < cout) v.w = wr[nn + 3];
                }
            }
            rb[u] = v;
        }
    };
    auto store_slab = [&](int buf) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST && affine) {
            sc = *reinterpret_cast<const float4 *>(&Ssc[rk]);
            sh = *reinterpret_cast<const float4 *>(&Ssh[rk]);
        }
#pragma unroll
        for (int h = 0; h < MLP_AP; h++) {
            const int r = a_row + h * MLP_RPP;
            float4 v = ra[h];
            if (FAST && affine) {
                v.x = v.x * sc.x + sh.x;
                v.y = v.y * sc.y + sh.y;
                v.z = v.z * sc.z + sh.z;
                v.w = v.w * sc.w + sh.w;
                if (in.in_relu) {
                    v.x = v.x > 0.f ? v.x : 0.f;
                    v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f;
                    v.w = v.w > 0.f ? v.w : 0.f;
                }
            }
            As[buf][a_kq * 4 + 0][r] = v.x;
            As[buf][a_kq * 4 + 1][r] = v.y;
            As[buf][a_kq * 4 + 2][r] = v.z;
            As[buf][a_kq * 4 + 3][r] = v.w;
        }
#pragma unroll
        for (int u = 0; u < B_PER_T; u++) {
            const int f = tid + u * 256;
            const int kk = f / (BN / 4), nq = f % (BN / 4);
            *reinterpret_cast<float4 *>(&Bs[buf][kk][nq * 4]) = rb[u];
        }
    };

    f32x16 acc[MT][NT];
    const int kh = lane >> 5, l31 = lane & 31;
    if (affine_lds) __syncthreads(); // Ssc / Ssh visible before the first staged load uses them
    // Pipeline over steps s = (tile, kt):  registers hold step s+1 while step s computes; at mid-step they are
    // written to the other LDS buffer and immediately re-used to fetch step s+2, so every global load has a
    // full step of MFMAs (not half) to land.
    long tile = blockIdx.x;  // step being computed
    int kt = 0;
    long ptile = tile;       // step whose operands are fetched next
    int pkt = 0;
    auto advance = [&](long &t, int &k) {
        if (++k == nk) {
            k = 0;
            t += gridDim.x;
        }
    };
    if (tile < ntiles) {
        set_rows(ptile);
        load_slab(pkt);
        store_slab(0);
        advance(ptile, pkt);
        if (ptile < ntiles) {
            if (pkt == 0) set_rows(ptile);
            load_slab(pkt); // step 1 in flight
        }
    }
    __syncthreads();
    int buf = 0;
    while (tile < ntiles) {
        const bool last_k = (kt + 1 == nk);
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
        }
        const bool have_next = ptile < ntiles; // registers hold step s+1
        // MFMA operand fragments are double-buffered in registers: the ds_reads of sub-step k2+1 are issued
        // before the MFMAs of sub-step k2, so LDS latency hides under the matrix pipe
        float fa[2][MT], fb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; i++) fa[0][i] = As[buf][kh][(wm * MT + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < NT; j++) fb[0][j] = Bs[buf][kh][(wn * NT + j) * 32 + l31];
#pragma unroll
        for (int k2 = 0; k2 < MLP_BK / 2; k2++) {
            if (k2 == MLP_BK / 4 && have_next) {
                store_slab(buf ^ 1); // other buffer: last read one step ago, behind a barrier
                advance(ptile, pkt);
                if (ptile < ntiles) {
                    if (pkt == 0) set_rows(ptile);
                    load_slab(pkt); // step s+2
                }
            }
            if (k2 + 1 < MLP_BK / 2) {
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
        if (last_k) {
            // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
            const long m0 = tile * MLP_BM;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int col = n0 + (wn * NT + j) * 32 + (lane & 31);
                const bool cok = col < cout;
                const float bv = (bias && cok) ? bias[col] : 0.0f;
#pragma unroll
                for (int i = 0; i < MT; i++) {
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const long row = m0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        if (FAST || (cok && row < rows)) {
                            const float v = acc[i][j][e] + bv;
                            z[(size_t)row * cout + col] = v;
                            s1[j] += v;
                            s2[j] += v * v;
                        }
                    }
                }
            }
        }
        advance(tile, kt);
    }
    if (stats) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32);
            const float t2 = s2[j] + __shfl_xor(s2[j], 32);
            const int col = n0 + (wn * NT + j) * 32 + (lane & 31);
            if (lane < 32 && col < cout) {
                unsafeAtomicAdd(&stats[col], (double)t1);
                unsafeAtomicAdd(&stats[cout + col], (double)t2);
            }
        }
    }
}

// scale/shift from accumulated sums (biased variance, utils.py BNReLU training mode)
__global__ void bn_finalize_kernel(long rows, int c, const double *__restrict__ stats, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, float eps, float *__restrict__ scale,
                                   float *__restrict__ shift, float *__restrict__ mean, float *__restrict__ var)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= c) return;
    const double mu = stats[o] / (double)rows;
    double v = stats[c + o] / (double)rows - mu * mu;
    if (v < 0) v = 0;
    const float muf = (float)mu, vf = (float)v;
    const float sc = gamma[o] / sqrtf(vf + eps);
    scale[o] = sc;
    shift[o] = beta[o] - muf * sc;
    if (mean) mean[o] = muf;
    if (var) var[o] = vf;
}

// out[g,c] = max_k act(z[g*k..,c]*scale+shift); argmax optional.  One thread per (group, channel quad).
__global__ void bn_relu_max_kernel(long groups, int k, int c, const float *__restrict__ z, const float *__restrict__ scale,
                                   const float *__restrict__ shift, int relu, float *__restrict__ out,
                                   int *__restrict__ argmax)
{
    const long total = groups * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / c;
        const int ch = (int)(e - g * c);
        const float sc = scale[ch], sh = shift[ch];
        const float *__restrict__ p = z + (size_t)g * k * c + ch;
        float best = 0;
        int bi = 0;
        for (int j = 0; j < k; j++) {
            float v = p[(size_t)j * c] * sc + sh;
            if (relu && !(v > 0.0f)) v = 0.0f;
            if (j == 0 || v > best) {
                best = v;
                bi = j;
            }
        }
        out[e] = best;
        if (argmax) argmax[e] = bi;
    }
}

// c % 4 == 0: one thread per (group, channel quad), 16-byte loads, eight rows in flight.  Same comparison order as above
// (first maximum wins), so out / argmax are identical.
__global__ __launch_bounds__(256) void bn_relu_max_vec_kernel(long groups, int k, int c, const float *__restrict__ z,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              int relu, float *__restrict__ out, int *__restrict__ argmax)
{
    const int qc = c >> 2;
    const long total = groups * qc;
    const float floor_v = relu ? 0.0f : -__builtin_inff();
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / qc;
        const int q = (int)(e - g * qc);
        const float4 sc = *reinterpret_cast<const float4 *>(scale + 4 * q), sh = *reinterpret_cast<const float4 *>(shift + 4 * q);
        const float *__restrict__ p = z + (size_t)g * k * c + 4 * q;
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 bi = make_int4(0, 0, 0, 0);
        auto take = [&](const float4 &raw, int j) {
            float4 v;
            v.x = raw.x * sc.x + sh.x;
            v.y = raw.y * sc.y + sh.y;
            v.z = raw.z * sc.z + sh.z;
            v.w = raw.w * sc.w + sh.w;
            if (!(v.x > floor_v)) v.x = floor_v; // ReLU (NaN -> 0 like the scalar kernel); no ReLU: the floor is -inf
            if (!(v.y > floor_v)) v.y = floor_v;
            if (!(v.z > floor_v)) v.z = floor_v;
            if (!(v.w > floor_v)) v.w = floor_v;
            if (j == 0 || v.x > best.x) { best.x = v.x; bi.x = j; }
            if (j == 0 || v.y > best.y) { best.y = v.y; bi.y = j; }
            if (j == 0 || v.z > best.z) { best.z = v.z; bi.z = j; }
            if (j == 0 || v.w > best.w) { best.w = v.w; bi.w = j; }
        };
        int j = 0;
        for (; j + 8 <= k; j += 8) {
            float4 r[8];
#pragma unroll
            for (int u = 0; u < 8; u++) r[u] = *reinterpret_cast<const float4 *>(p + (size_t)(j + u) * c);
#pragma unroll
            for (int u = 0; u < 8; u++) take(r[u], j + u);
        }
        for (; j < k; j++) take(*reinterpret_cast<const float4 *>(p + (size_t)j * c), j);
        *reinterpret_cast<float4 *>(out + (size_t)g * c + 4 * q) = best;
        if (argmax) *reinterpret_cast<int4 *>(argmax + (size_t)g * c + 4 * q) = bi;
    }
}

__global__ void bn_relu_kernel(long total, int c, const float *__restrict__ z, const float *__restrict__ scale,
                               const float *__restrict__ shift, BnRaw raw, int relu, float *__restrict__ y)
{
    const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    // When the grid's stride is a multiple of c (the launcher arranges it whenever c divides 256) a thread's channel never changes: its
    // scale / shift -- from raw sums: two fp64 divisions and a square root -- are computed once, not per element (round 6)
    const bool fixed = stride % c == 0;
    float sc = 0.0f, sh = 0.0f;
    if (fixed && e0 < total) {
        const int ch = (int)(e0 % c);
        if (raw.stats) bn_raw_channel(raw, c, ch, e0 < c, sc, sh); // the first c elements also record the four vectors
        else {
            sc = scale[ch];
            sh = shift[ch];
        }
    }
    for (long e = e0; e < total; e += stride) {
        if (!fixed) {
            const int ch = (int)(e % c);
            if (raw.stats) bn_raw_channel(raw, c, ch, e < c, sc, sh);
            else {
                sc = scale[ch];
                sh = shift[ch];
            }
        }
        float v = z[e] * sc + sh;
        if (relu && !(v > 0.0f)) v = 0.0f;
        y[e] = v;
    }
}

// Second half of the max-pool fused into the last GEMM's epilogue (mlp_fast.hip, EPI 2): with the layer's BatchNorm
// scale/shift known, out = act(s * (s >= 0 ? zmax : zmin) + h) and argmax = the matching row offset.  s == 0: every row
// gives act(h), the first row is the arg-max (what the separate pass returns).
__global__ void bn_pool_finalize_kernel(long total, int c, const float *__restrict__ zmax, const float *__restrict__ zmin,
                                        const int *__restrict__ amax, const int *__restrict__ amin, const float *__restrict__ scale,
                                        const float *__restrict__ shift, BnRaw raw, int relu, float *__restrict__ out,
                                        int *__restrict__ argmax, float *__restrict__ zsel)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(e % c);
        float s, h;
        if (raw.stats) bn_raw_channel(raw, c, ch, e < c, s, h);
        else {
            s = scale[ch];
            h = shift[ch];
        }
        const float zr = s >= 0.0f ? zmax[e] : zmin[e]; // a zero scale ties every row: the row of the raw maximum stands for them
        float v = zr * s + h;
        if (relu && !(v > 0.0f)) v = 0.0f;
        out[e] = v;
        if (argmax) argmax[e] = s >= 0.0f ? amax[e] : amin[e];
        if (zsel) zsel[e] = zr;
    }
}

static inline int grid_for(long total, int block)
{
    long g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

bool mlp_linear_fast_launch(const float *x, const float *in_scale, const float *in_shift, const BnRaw &in_raw, int in_relu,
                            long rows, int cin, int cout, const float *w, const float *bias, float *z, double *stats,
                            hipStream_t st); // mlp_fast.hip

template <int MODE>
static int launch_linear(const MlpIn &in_, long rows, int cin, int cout, const float *w, const float *bias, float *z,
                         double *stats, const BnRaw &in_raw, hipStream_t st)
{
    if (MODE == 0 && mlp_linear_fast_launch(in_.x, in_.in_scale, in_.in_shift, in_raw, in_.in_relu, rows, cin, cout, w, bias, z, stats, st))
        return check_launch("mlp_linear");
    MlpIn in = in_;
    if (in_raw.stats) { // the generic kernel takes the affine as vectors: finalize into the caller's out first (one more launch)
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((cin + 255) / 256), dim3(256), 0, st, in_raw.rows, cin, in_raw.stats, in_raw.gamma,
                           in_raw.beta, in_raw.eps, in_raw.out, in_raw.out + cin, in_raw.out + 2 * cin, in_raw.out + 3 * cin);
        in.in_scale = in_raw.out;
        in.in_shift = in_raw.out + cin;
    }
    const long ntiles = (rows + MLP_BM - 1) / MLP_BM;
    const bool aligned = ((uintptr_t)in.x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)z % 16 == 0);
    const bool fast_in = MODE == 0 && aligned && (cin % MLP_BK == 0) && (rows % MLP_BM == 0) &&
                         (in.in_scale == nullptr || cin <= MLP_MAXC);
    if (cout > 64) {
        const int ny = (cout + 127) / 128;
        long gx = ntiles < 1024 / ny ? ntiles : 1024 / ny;
        if (gx < 1) gx = 1;
        if (fast_in && cout % 128 == 0) {
            if constexpr (MODE == 0)
                hipLaunchKernelGGL((mlp_linear_kernel<0, 2, 2, 2, 2, true>), dim3((unsigned)gx, ny), dim3(256), 0, st, in, rows, cin,
                                   cout, w, bias, z, stats);
        } else {
            hipLaunchKernelGGL((mlp_linear_kernel<MODE, 2, 2, 2, 2, false>), dim3((unsigned)gx, ny), dim3(256), 0, st, in, rows, cin,
                               cout, w, bias, z, stats);
        }
    } else {
        long gx = ntiles < 2048 ? ntiles : 2048;
        if (fast_in && cout == 64) {
            if constexpr (MODE == 0)
                hipLaunchKernelGGL((mlp_linear_kernel<0, 4, 1, 1, 2, true>), dim3((unsigned)gx, 1), dim3(256), 0, st, in, rows, cin, cout,
                                   w, bias, z, stats);
        } else {
            hipLaunchKernelGGL((mlp_linear_kernel<MODE, 4, 1, 1, 2, false>), dim3((unsigned)gx, 1), dim3(256), 0, st, in, rows, cin, cout,
                               w, bias, z, stats);
        }
    }
    return check_launch("mlp_linear");
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_mlp_linear(const votenet_mlp_input *in, long rows, int cin, int cout, const float *w,
                                  const float *bias, float *z, double *stats, void *stream)
{
    VN_REQUIRE(in != nullptr, "mlp_linear: null input descriptor");
    VN_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "mlp_linear expects rows >= 0, cin > 0, cout > 0");
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(w && z, "mlp_linear: null buffer");
    MlpIn d = {};
    d.x = in->x;
    d.in_scale = in->in_scale;
    d.in_shift = in->in_shift;
    d.in_relu = in->in_relu;
    const BnRaw raw = to_raw(in->in_bn);
    VN_REQUIRE(in->in_bn == nullptr || (in->x && raw.stats && raw.gamma && raw.beta && raw.out && raw.rows > 0 && in->in_scale == nullptr),
               "mlp_linear: in_bn needs a DENSE input, stats, gamma, beta, out, rows > 0 and no in_scale");
    d.xyz = in->xyz;
    d.new_xyz = in->new_xyz;
    d.feat = in->feat;
    d.idx = in->idx;
    d.n = in->n;
    d.m = in->m;
    d.nsample = in->nsample;
    d.c = in->feat ? in->c : 0;
    hipStream_t st = as_stream(stream);
    if (in->x) {
        VN_REQUIRE((in->in_scale == nullptr) == (in->in_shift == nullptr), "mlp_linear: in_scale and in_shift go together");
        return launch_linear<0>(d, rows, cin, cout, w, bias, z, stats, raw, st);
    }
    VN_REQUIRE(in->xyz && in->new_xyz && in->idx, "mlp_linear: GATHER input needs xyz, new_xyz and idx");
    VN_REQUIRE(in->b > 0 && in->n > 0 && in->m > 0 && in->nsample > 0, "mlp_linear: GATHER input needs b, n, m, nsample > 0");
    VN_REQUIRE(rows == (long)in->b * in->m * in->nsample, "mlp_linear: rows must equal b*m*nsample for a GATHER input");
    VN_REQUIRE(cin == 3 + d.c, "mlp_linear: cin must equal 3 + c for a GATHER input (utils.py:55)");
    return launch_linear<1>(d, rows, cin, cout, w, bias, z, stats, BnRaw{}, st);
}

namespace votenet {
bool mlp_linear_pool_launch(const float *x, const float *in_scale, const float *in_shift, const BnRaw &in_raw, int in_relu,
                            long rows, int cin, int cout, const float *w, const float *bias, float *z, double *stats, float *zmax,
                            float *zmin, int *amax, int *amin, hipStream_t st, const float *wh = nullptr,
                            const float *pool_gamma = nullptr, const int *nh_dev = nullptr); // mlp_fast.hip
}

extern "C" int votenet_mlp_linear_pool(const votenet_mlp_input *in, long rows, int cin, int cout, const float *w,
                                       const float *bias, float *z, double *stats, int pool_k, float *zmax, float *zmin,
                                       int *amax, int *amin, void *stream)
{
    VN_REQUIRE(in != nullptr && in->x != nullptr, "mlp_linear_pool: DENSE input descriptor required");
    VN_REQUIRE(rows > 0 && cin > 0 && cout > 0, "mlp_linear_pool expects rows > 0, cin > 0, cout > 0");
    VN_REQUIRE(w && zmax && zmin && amax && amin, "mlp_linear_pool: null buffer");
    VN_REQUIRE((in->in_scale == nullptr) == (in->in_shift == nullptr), "mlp_linear_pool: in_scale and in_shift go together");
    const BnRaw raw = to_raw(in->in_bn);
    VN_REQUIRE(in->in_bn == nullptr || (raw.stats && raw.gamma && raw.beta && raw.rows > 0 && in->in_scale == nullptr),
               "mlp_linear_pool: in_bn needs stats, gamma, beta, rows > 0 and no in_scale");
    if (pool_k != 64 || !votenet::mlp_linear_pool_launch(in->x, in->in_scale, in->in_shift, raw, in->in_relu, rows, cin, cout, w, bias, z,
                                                         stats, zmax, zmin, amax, amin, as_stream(stream)))
        return votenet::set_error(VOTENET_E_INVALID_ARGUMENT,
                                  "mlp_linear_pool: shape not served (pool_k == 64, rows % 128 == 0, cin % 32 == 0, cin <= 512, "
                                  "cout % 128 == 0, 16-byte aligned): use votenet_mlp_linear + votenet_bn_relu_max");
    return check_launch("mlp_linear_pool");
}

// votenet_mlp_linear_pool on the piece layout (half.hip): rows = 16 x pieces; zbest / abest (pieces x cout) are the pool's candidate of
// every 16-row piece -- the raw max of z where gamma (the pooled layer's BatchNorm weight, whose sign is the sign of the scale the pool
// applies) is >= 0, the raw min where it is negative, first occurrence -- joined per centre by votenet_bn_pool_finalize_half; the
// statistics count row 0 of a piece wh times.
extern "C" int votenet_mlp_linear_pool_half(const votenet_mlp_input *in, long rows, int cin, int cout, const float *w, const float *bias,
                                            float *z, double *stats, const float *wh, const float *gamma, float *zbest, int *abest,
                                            const int *nh_dev, void *stream)
{
    float *zmax = zbest, *zmin = zbest;
    int *amax = abest, *amin = abest;
    VN_REQUIRE(gamma != nullptr, "mlp_linear_pool_half: the pooled layer's gamma is missing");
    VN_REQUIRE(in != nullptr && in->x != nullptr, "mlp_linear_pool_half: DENSE input descriptor required");
    VN_REQUIRE(rows > 0 && cin > 0 && cout > 0, "mlp_linear_pool_half expects rows > 0, cin > 0, cout > 0");
    VN_REQUIRE(w && wh && zmax && zmin && amax && amin, "mlp_linear_pool_half: null buffer");
    VN_REQUIRE((in->in_scale == nullptr) == (in->in_shift == nullptr), "mlp_linear_pool_half: in_scale and in_shift go together");
    const BnRaw raw = to_raw(in->in_bn);
    VN_REQUIRE(in->in_bn == nullptr || (raw.stats && raw.gamma && raw.beta && raw.rows > 0 && in->in_scale == nullptr),
               "mlp_linear_pool_half: in_bn needs stats, gamma, beta, rows > 0 and no in_scale");
    if (!votenet::mlp_linear_pool_launch(in->x, in->in_scale, in->in_shift, raw, in->in_relu, rows, cin, cout, w, bias, z, stats, zmax, zmin,
                                         amax, amin, as_stream(stream), wh, gamma, nh_dev))
        return votenet::set_error(VOTENET_E_INVALID_ARGUMENT, "mlp_linear_pool_half: shape not served (as votenet_mlp_linear_pool)");
    return check_launch("mlp_linear_pool_half");
}

extern "C" int votenet_bn_pool_finalize(long groups, int c, const float *zmax, const float *zmin, const int *amax, const int *amin,
                                        const float *scale, const float *shift, const votenet_bn_raw *bn, int relu, float *out,
                                        int *argmax, float *zsel, void *stream)
{
    VN_REQUIRE(groups >= 0 && c > 0, "bn_pool_finalize expects groups >= 0, c > 0");
    if (groups == 0) return VOTENET_OK;
    const BnRaw raw = to_raw(bn);
    VN_REQUIRE(zmax && zmin && amax && amin && out && ((scale && shift) || (raw.stats && raw.gamma && raw.beta && raw.rows > 0)),
               "bn_pool_finalize: null buffer");
    hipLaunchKernelGGL(bn_pool_finalize_kernel, dim3(grid_for(groups * c, 256)), dim3(256), 0, as_stream(stream), groups * c, c, zmax,
                       zmin, amax, amin, scale, shift, raw, relu, out, argmax, zsel);
    return check_launch("bn_pool_finalize");
}

extern "C" int votenet_bn_finalize(long rows, int c, const double *stats, const float *gamma, const float *beta, float eps,
                                   float *scale, float *shift, float *mean, float *var, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0, "bn_finalize expects rows > 0, c > 0");
    VN_REQUIRE(stats && gamma && beta && scale && shift, "bn_finalize: null buffer");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), rows, c, stats, gamma, beta,
                       eps, scale, shift, mean, var);
    return check_launch("bn_finalize");
}

extern "C" int votenet_bn_relu_max(long groups, int k, int c, const float *z, const float *scale, const float *shift,
                                   int relu, float *out, int *argmax, void *stream)
{
    VN_REQUIRE(groups >= 0 && k > 0 && c > 0, "bn_relu_max expects groups >= 0, k > 0, c > 0");
    if (groups == 0) return VOTENET_OK;
    VN_REQUIRE(z && scale && shift && out, "bn_relu_max: null buffer");
    const bool vec = c % 4 == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)scale % 16 == 0 && (uintptr_t)shift % 16 == 0 &&
                     (uintptr_t)out % 16 == 0 && (uintptr_t)argmax % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(bn_relu_max_vec_kernel, dim3(grid_for(groups * (c / 4), 256)), dim3(256), 0, as_stream(stream), groups, k,
                           c, z, scale, shift, relu, out, argmax);
    else
        hipLaunchKernelGGL(bn_relu_max_kernel, dim3(grid_for(groups * c, 256)), dim3(256), 0, as_stream(stream), groups, k, c, z,
                           scale, shift, relu, out, argmax);
    return check_launch("bn_relu_max");
}

extern "C" int votenet_bn_relu(long rows, int c, const float *z, const float *scale, const float *shift, const votenet_bn_raw *bn,
                               int relu, float *y, void *stream)
{
    VN_REQUIRE(rows >= 0 && c > 0, "bn_relu expects rows >= 0, c > 0");
    if (rows == 0) return VOTENET_OK;
    const BnRaw raw = to_raw(bn);
    VN_REQUIRE(z && y && ((scale && shift) || (raw.stats && raw.gamma && raw.beta && raw.rows > 0)), "bn_relu: null buffer");
    hipLaunchKernelGGL(bn_relu_kernel, dim3(256 % c == 0 && rows * c > 1024L * 256 ? 1024 : grid_for(rows * c, 256)), dim3(256), 0, as_stream(stream), rows * c, c, z, scale,
                       shift, raw, relu, y);
    return check_launch("bn_relu");
}
