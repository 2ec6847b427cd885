// mlp_bwd.hip -- backward pass of the grouped-point MLP for gfx950.
//
// The reference gets these gradients from TensorFlow autodiff over Conv2D / BatchNorm / ReLU /
// reduce_max graph nodes plus its registered GroupPointGrad (tf_grouping.py:42-46).  Here:
//   votenet_bn_backward_reduce : s1 = sum(da'), s2 = sum(da' * zhat) per channel, da' = da * [act > 0];
//                                da is either dense or the max-pool scatter of (gout, argmax)
//   votenet_bn_backward_apply  : dz = gamma*invstd * (da' - s1/N - zhat*s2/N)   (training-mode BN)
//   votenet_mlp_wgrad          : dW += A^T dz on v_mfma_f32_32x32x2_f32 with the SAME fused A loaders
//                                as the forward pass (gather / dense + folded BN+ReLU); the
//                                contraction runs over rows, split across workgroups
//   (input gradients da = dz W^T reuse votenet_mlp_linear with W^T)
//   votenet_group_concat_grad  : scatter of the first layer's input gradient back to the feature
//                                and xyz tables (GroupPointGrad + the tile/subtract of utils.py:51)
//   votenet_clip_adam          : per-tensor clip_by_average_norm + Adam over the flat bucket (model.py:240-250)
#include "mlp_types.h"

namespace votenet {


// ---------------------------------------------------------------- BN backward: reductions
// block = 64 columns x 4 row-lanes; grid.x strides over rows, grid.y over column tiles of 64
__global__ __launch_bounds__(256) void bn_bwd_reduce_dense_kernel(long rows, int c, const float *__restrict__ da,
                                                                  const float *__restrict__ z, const float *__restrict__ scale,
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
        for (long r = (long)blockIdx.x * 4 + ry; r < rows; r += (long)gridDim.x * 4) {
            const float zz = z[(size_t)r * c + col];
            float g = da[(size_t)r * c + col];
            if (relu && !(zz * sc + sf > 0.0f)) g = 0.0f;
            s1 += g;
            s2 += g * ((zz - mu) * inv);
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

// The same reduction for c % 4 == 0 (c <= 1024), 16-byte accesses: thread = one channel quad, QC = c/4 threads per row,
// 256/QC rows per pass, four passes in flight per loop trip.  Partial sums meet in LDS, one fp64 atomic per channel,
// statistic and workgroup.
__global__ __launch_bounds__(256) void bn_bwd_reduce_dense_vec_kernel(long rows, int c, const float *__restrict__ da,
                                                                      const float *__restrict__ z, const float *__restrict__ scale,
                                                                      const float *__restrict__ shift, const float *__restrict__ mean,
                                                                      const float *__restrict__ var, float eps, int relu,
                                                                      double *__restrict__ sums, CoefTail tail)
{
    __shared__ float red[2][256][4];
    const int qc = c >> 2;             // quads per row (a divisor of 256, checked by the launcher)
    const int rpp = 256 / qc;          // rows per pass
    const int q = threadIdx.x % qc, rl = threadIdx.x / qc;
    const float4 sc = *reinterpret_cast<const float4 *>(scale + 4 * q), sf = *reinterpret_cast<const float4 *>(shift + 4 * q);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + 4 * q), vr = *reinterpret_cast<const float4 *>(var + 4 * q);
    const float4 inv = make_float4(1.0f / sqrtf(vr.x + eps), 1.0f / sqrtf(vr.y + eps), 1.0f / sqrtf(vr.z + eps), 1.0f / sqrtf(vr.w + eps));
    const float thr = relu ? 0.0f : -__builtin_inff(); // no ReLU: every element passes
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    auto acc = [&](const float4 &zz, float4 g) {
        if (!(zz.x * sc.x + sf.x > thr)) g.x = 0.0f;
        if (!(zz.y * sc.y + sf.y > thr)) g.y = 0.0f;
        if (!(zz.z * sc.z + sf.z > thr)) g.z = 0.0f;
        if (!(zz.w * sc.w + sf.w > thr)) g.w = 0.0f;
        s1.x += g.x;
        s1.y += g.y;
        s1.z += g.z;
        s1.w += g.w;
        s2.x += g.x * ((zz.x - mu.x) * inv.x);
        s2.y += g.y * ((zz.y - mu.y) * inv.y);
        s2.z += g.z * ((zz.z - mu.z) * inv.z);
        s2.w += g.w * ((zz.w - mu.w) * inv.w);
    };
    const long stride = (long)gridDim.x * rpp;
    long r = (long)blockIdx.x * rpp + rl;
    for (; r + 7 * stride < rows; r += 8 * stride) { // sixteen 16-byte loads in flight per thread: the pass is a latency chain on few workgroups
        float4 zz[8], gg[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            zz[u] = *reinterpret_cast<const float4 *>(z + (size_t)(r + u * stride) * c + 4 * q);
            gg[u] = *reinterpret_cast<const float4 *>(da + (size_t)(r + u * stride) * c + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc(zz[u], gg[u]);
    }
    for (; r < rows; r += stride)
        acc(*reinterpret_cast<const float4 *>(z + (size_t)r * c + 4 * q), *reinterpret_cast<const float4 *>(da + (size_t)r * c + 4 * q));
    *reinterpret_cast<float4 *>(&red[0][threadIdx.x][0]) = s1;
    *reinterpret_cast<float4 *>(&red[1][threadIdx.x][0]) = s2;
    __syncthreads();
    // thread t < 2*c: statistic t / c, channel t % c; sum over the rpp row-lanes
    for (int t = threadIdx.x; t < 2 * c; t += 256) {
        const int which = t / c, ch = t % c;
        float v = 0.0f;
        for (int i = 0; i < rpp; i++) v += red[which][i * qc + (ch >> 2)][ch & 3];
        unsafeAtomicAdd(&sums[which * c + ch], (double)v);
    }
    coef_tail(tail, gridDim.x, c, sums, scale, shift, mean, var, eps);
}

// max-pool mode: da'[g*k+argmax[g,col], col] = gout[g,col] * [act > 0], zero elsewhere
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(long groups, int k, int c, const float *__restrict__ gout,
                                                                 const int *__restrict__ argmax, const float *__restrict__ z,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 const float *__restrict__ mean, const float *__restrict__ var,
                                                                 float eps, int relu, double *__restrict__ sums, CoefTail tail)
{
    __shared__ float sh1[4][64], sh2[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cx;
    float s1 = 0, s2 = 0;
    if (col < c) {
        const float sc = scale[col], sf = shift[col], mu = mean[col], inv = 1.0f / sqrtf(var[col] + eps);
        for (long g = (long)blockIdx.x * 4 + ry; g < groups; g += (long)gridDim.x * 4) {
            const int a = argmax[(size_t)g * c + col];
            const float zz = z[((size_t)g * k + a) * c + col];
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

// plain column sums (bias gradient of a layer without BatchNorm)
__global__ __launch_bounds__(256) void colsum_kernel(long rows, int c, const float *__restrict__ x, int pitch, double *__restrict__ sums)
{
    __shared__ float sh1[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cx;
    float s1 = 0;
    if (col < c)
        for (long r = (long)blockIdx.x * 4 + ry; r < rows; r += (long)gridDim.x * 4) s1 += x[(size_t)r * pitch + col];
    sh1[ry][cx] = s1;
    __syncthreads();
    if (ry == 0 && col < c) unsafeAtomicAdd(&sums[col], (double)((sh1[0][cx] + sh1[1][cx]) + (sh1[2][cx] + sh1[3][cx])));
}

// ---------------------------------------------------------------- BN backward: apply
// dz = gamma*invstd*(da' - s1/N - zhat*s2/N) = A*da' + B + C*z with per-channel A, B, C:
//   A = gamma*inv, C = -gamma*inv^2*(s2/N), B = -gamma*inv*(s1/N) - C*mean.   coef: [A | B | C | scale | shift], 5*c floats
__global__ void bn_bwd_coef_kernel(long rows, int c, const float *__restrict__ scale, const float *__restrict__ shift,
                                   const float *__restrict__ mean, const float *__restrict__ var, float eps,
                                   const float *__restrict__ gamma, const double *__restrict__ sums, float *__restrict__ coef,
                                   float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    const double invn = 1.0 / (double)rows;
    const float inv = 1.0f / sqrtf(var[col] + eps);
    const float m1 = (float)(sums[col] * invn), m2 = (float)(sums[c + col] * invn);
    const float A = gamma[col] * inv;
    const float C = -A * inv * m2;
    coef[col] = A;
    coef[c + col] = -A * m1 - C * mean[col];
    coef[2 * c + col] = C;
    coef[3 * c + col] = scale[col];
    coef[4 * c + col] = shift[col];
    if (dgamma) dgamma[col] += (float)sums[c + col];
    if (dbeta) dbeta[col] += (float)sums[col];
}

// VEC = 4: c % 4 == 0, float4 per thread; VEC = 1: scalar.  k > 0: da is the pooled gout with argmax.
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long rows, int c, int k, const float *__restrict__ da,
                                                           const int *__restrict__ argmax, const float *__restrict__ z,
                                                           const float *__restrict__ coef, int relu, float *__restrict__ dz)
{
    const int cv = c / VEC;
    const long total = rows * cv;
    // the grid stride (gridDim.x * 256) is a multiple of cv whenever cv divides 256: then a thread keeps
    // the same channels for its whole row sweep and the five per-channel coefficients live in registers
    const bool fixed_col = (256 % cv) == 0;
    float cA[VEC], cB[VEC], cC[VEC], cS[VEC], cH[VEC];
    if (fixed_col) {
        const int col0 = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % cv) * VEC;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            cA[q] = coef[col0 + q];
            cB[q] = coef[c + col0 + q];
            cC[q] = coef[2 * c + col0 + q];
            cS[q] = coef[3 * c + col0 + q];
            cH[q] = coef[4 * c + col0 + q];
        }
    }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / cv;
        const int col = (int)(e - r * cv) * VEC;
        float zz[VEC], g[VEC], out[VEC];
        if (!fixed_col) {
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                cA[q] = coef[col + q];
                cB[q] = coef[c + col + q];
                cC[q] = coef[2 * c + col + q];
                cS[q] = coef[3 * c + col + q];
                cH[q] = coef[4 * c + col + q];
            }
        }
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(z + (size_t)r * c + col);
            zz[0] = t.x; zz[VEC > 1 ? 1 : 0] = t.y; zz[VEC > 2 ? 2 : 0] = t.z; zz[VEC > 3 ? 3 : 0] = t.w;
        } else {
            zz[0] = z[(size_t)r * c + col];
        }
        if (k > 0) {
            const long grp = r / k;
            const int rk = (int)(r - grp * k);
#pragma unroll
            for (int q = 0; q < VEC; q++)
                g[q] = (rk == argmax[(size_t)grp * c + col + q]) ? da[(size_t)grp * c + col + q] : 0.0f;
        } else if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(da + (size_t)r * c + col);
            g[0] = t.x; g[VEC > 1 ? 1 : 0] = t.y; g[VEC > 2 ? 2 : 0] = t.z; g[VEC > 3 ? 3 : 0] = t.w;
        } else {
            g[0] = da[(size_t)r * c + col];
        }
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            float gg = g[q];
            if (relu && !(zz[q] * cS[q] + cH[q] > 0.0f)) gg = 0.0f;
            out[q] = cA[q] * gg + cB[q] + cC[q] * zz[q];
        }
        if (VEC == 4)
            *reinterpret_cast<float4 *>(dz + (size_t)r * c + col) = make_float4(out[0], out[VEC > 1 ? 1 : 0], out[VEC > 2 ? 2 : 0], out[VEC > 3 ? 3 : 0]);
        else
            dz[(size_t)r * c + col] = out[0];
    }
}

// dst[i] += (float)src[i]  (bias gradient from colsum)
__global__ void add_f64_to_f32_kernel(int n, const double *__restrict__ src, float *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += (float)src[i];
}

// ---------------------------------------------------------------- weight gradient
constexpr int WG_BR = 16;  // rows per slab (the MFMA contraction index)

template <int MODE>
__device__ __forceinline__ int w_row(int k, int c)
{
    if (MODE == 0) return k;
    return k < c ? k + 3 : k - c;
}

// dW[w_row(i), j] += sum_r A[r,i] * dz[r,j] over this workgroup's row range.
// 4 waves as 2x2, each TI x TJ MFMA tiles of 32x32: the dW tile is (64*TI) x (64*TJ).  LDS images are the
// natural [row][channel] slabs: the A^T operand of lane l is As[k2*2 + (l>>5)][i0 + (l&31)], conflict-free.
// Pipeline as in mlp_linear_kernel: registers hold slab s+1 while slab s computes, they are written to the
// other LDS buffer half way through the MFMAs and re-used at once for slab s+2; in GATHER mode the row
// indices of slab s+3 are fetched at the same point, so the idx -> feature-row dependency is never exposed.
// BSRC selects how the dz operand (rows x cout) is produced (struct BnSrc, mlp_types.h): 0 memory, 1 dense, 2 pooled
template <int MODE, int TI, int TJ, int BSRC>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(MlpIn in, long rows, int cin, int cout, const float *__restrict__ dz,
                                                        BnSrc bs, float *__restrict__ dw, long rows_per_block)
{
    constexpr int BI = 64 * TI, BJ = 64 * TJ;
    constexpr int QA = BI / 4, QB = BJ / 4;          // float4 per slab row
    constexpr int NA = WG_BR * QA / 256, NB = WG_BR * QB / 256; // float4 per thread per slab (1 or 2)
    constexpr int RA = 256 / QA, RB = 256 / QB;      // slab rows covered by one pass of the 256 threads
    __shared__ float As[2][WG_BR][BI + 4];
    __shared__ float Bs[2][WG_BR][BJ + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const int i0 = blockIdx.y * BI, j0 = blockIdx.z * BJ;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    long r_end = r_begin + rows_per_block;
    if (r_end > rows) r_end = rows;
    if (r_begin >= r_end) return;
    const bool a_vec4 = (MODE == 0) ? ((cin & 3) == 0) : ((in.c & 3) == 0 && in.c > 0);
    const bool b_vec4 = (cout & 3) == 0;
    const bool affine = (MODE == 0) && in.in_scale != nullptr;
    const int a_row = tid / QA, a_q = tid % QA;
    const int b_row = tid / QB, b_q = tid % QB;
    const int ka = i0 + a_q * 4; // this thread's A channels (internal order)
    const int nb = j0 + b_q * 4; // this thread's dz channels

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; a++)
#pragma unroll
        for (int b = 0; b < TJ; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[a][b][e] = 0.0f;

    float4 ra[NA], rb[NB];
    float4 rg[NB];  // BSRC 1: da quads, BSRC 2: gout quads
    int4 rm[NB];    // BSRC 2: arg-max quads
    long cur_store_r0 = 0; // first row of the slab currently held in ra / rb
    // this thread's dz channels never change: BatchNorm-backward coefficients in registers
    float kA[4] = {0.f, 0.f, 0.f, 0.f}, kB[4] = {0.f, 0.f, 0.f, 0.f}, kC[4] = {0.f, 0.f, 0.f, 0.f}, kS[4] = {0.f, 0.f, 0.f, 0.f},
          kH[4] = {0.f, 0.f, 0.f, 0.f};
    if (BSRC != 0) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (nb + q < cout) {
                kA[q] = bs.coef[nb + q];
                kB[q] = bs.coef[cout + nb + q];
                kC[q] = bs.coef[2 * cout + nb + q];
                kS[q] = bs.coef[3 * cout + nb + q];
                kH[q] = bs.coef[4 * cout + nb + q];
            }
    }
    int pidx[NA]; // GATHER: idx of the rows of the slab loaded NEXT
    auto load_idx = [&](long r0) {
        if (MODE == 1) {
#pragma unroll
            for (int h = 0; h < NA; h++) {
                const long r = r0 + a_row + h * RA;
                pidx[h] = (r < r_end) ? in.idx[r] : 0;
            }
        }
    };
    auto load_slab = [&](long r0) {
        cur_store_r0 = r0;
#pragma unroll
        for (int h = 0; h < NA; h++) {
            const long r = r0 + a_row + h * RA;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < r_end) {
                if (MODE == 0) {
                    if (a_vec4 && ka + 3 < cin) {
                        va = *reinterpret_cast<const float4 *>(in.x + (size_t)r * cin + ka); // raw; BN+ReLU at store time
                    } else {
                        float t[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            float v = 0.0f;
                            if (ka + q < cin) {
                                v = in.x[(size_t)r * cin + ka + q];
                            }
                            t[q] = v;
                        }
                        va = make_float4(t[0], t[1], t[2], t[3]);
                    }
                } else {
                    const int src = pidx[h];
                    const long scene = r / ((long)in.m * in.nsample);
                    if (a_vec4 && ka + 3 < in.c) {
                        va = *reinterpret_cast<const float4 *>(in.feat + ((size_t)scene * in.n + src) * in.c + ka);
                    } else {
                        float t[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int k = ka + q;
                            float v = 0.0f;
                            if (k < in.c)
                                v = in.feat[((size_t)scene * in.n + src) * in.c + k];
                            else if (k < cin)
                                v = in.xyz[((size_t)scene * in.n + src) * 3 + (k - in.c)] -
                                    in.new_xyz[(size_t)(r / in.nsample) * 3 + (k - in.c)];
                            t[q] = v;
                        }
                        va = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            ra[h] = va;
        }
#pragma unroll
        for (int h = 0; h < NB; h++) {
            const long r = r0 + b_row + h * RB;
            float4 vb = make_float4(0.f, 0.f, 0.f, 0.f), vg = make_float4(0.f, 0.f, 0.f, 0.f);
            int4 vm = make_int4(-1, -1, -1, -1);
            if (r < r_end) {
                const float *zr = (BSRC == 0 ? dz : bs.z) + (size_t)r * cout; // BSRC != 0: the raw layer output z
                // rows < 2^31 (checked by the launcher): 32-bit group arithmetic, a shift for the usual power-of-two k
                const unsigned grp = (BSRC == 2) ? (bs.pool_shift >= 0 ? (unsigned)r >> bs.pool_shift : (unsigned)r / (unsigned)bs.pool_k) : 0u;
                const float *gr = (BSRC == 1) ? bs.da + (size_t)r * cout : (BSRC == 2 ? bs.gout + (size_t)grp * cout : nullptr);
                if (b_vec4 && nb + 3 < cout) {
                    vb = *reinterpret_cast<const float4 *>(zr + nb);
                    if (BSRC != 0) vg = *reinterpret_cast<const float4 *>(gr + nb);
                    if (BSRC == 2) vm = *reinterpret_cast<const int4 *>(bs.argmax + (size_t)grp * cout + nb);
                } else {
                    float tb[4] = {0.f, 0.f, 0.f, 0.f}, tg[4] = {0.f, 0.f, 0.f, 0.f};
                    int tm[4] = {-1, -1, -1, -1};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (nb + q < cout) {
                            tb[q] = zr[nb + q];
                            if (BSRC != 0) tg[q] = gr[nb + q];
                            if (BSRC == 2) tm[q] = bs.argmax[(size_t)grp * cout + nb + q];
                        }
                    vb = make_float4(tb[0], tb[1], tb[2], tb[3]);
                    vg = make_float4(tg[0], tg[1], tg[2], tg[3]);
                    vm = make_int4(tm[0], tm[1], tm[2], tm[3]);
                }
            }
            rb[h] = vb;
            if (BSRC != 0) rg[h] = vg;
            if (BSRC == 2) rm[h] = vm;
        }
    };
    // this thread's A channels never change: the folded BN scale/shift of the previous layer sit in registers and
    // are applied when the slab is written to LDS, so the global loads stay in flight until then
    float csc[4] = {1.f, 1.f, 1.f, 1.f}, csh[4] = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (ka + q < cin) {
                csc[q] = in.in_scale[ka + q];
                csh[q] = in.in_shift[ka + q];
            }
    }
    const bool do_relu = affine && in.in_relu;
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int h = 0; h < NA; h++) {
            float4 v = ra[h];
            if (affine) {
                v.x = v.x * csc[0] + csh[0];
                v.y = v.y * csc[1] + csh[1];
                v.z = v.z * csc[2] + csh[2];
                v.w = v.w * csc[3] + csh[3];
                if (do_relu) {
                    v.x = v.x > 0.f ? v.x : 0.f;
                    v.y = v.y > 0.f ? v.y : 0.f;
                    v.z = v.z > 0.f ? v.z : 0.f;
                    v.w = v.w > 0.f ? v.w : 0.f;
                }
                // rows beyond r_end and channels beyond cin were loaded as 0 and must stay 0
                const long r = cur_store_r0 + a_row + h * RA;
                if (r >= r_end) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ka + 0 >= cin) v.x = 0.f;
                if (ka + 1 >= cin) v.y = 0.f;
                if (ka + 2 >= cin) v.z = 0.f;
                if (ka + 3 >= cin) v.w = 0.f;
            }
            *reinterpret_cast<float4 *>(&As[buf][a_row + h * RA][a_q * 4]) = v;
        }
#pragma unroll
        for (int h = 0; h < NB; h++) {
            float4 v = rb[h];
            if (BSRC != 0) {
                const long r = cur_store_r0 + b_row + h * RB;
                float zz[4] = {v.x, v.y, v.z, v.w}, gg[4] = {rg[h].x, rg[h].y, rg[h].z, rg[h].w}, o[4];
                if (BSRC == 2) {
                    const int ro = bs.pool_shift >= 0 ? (int)((unsigned)r & (unsigned)(bs.pool_k - 1)) : (int)((unsigned)r % (unsigned)bs.pool_k);
                    const int am[4] = {rm[h].x, rm[h].y, rm[h].z, rm[h].w};
#pragma unroll
                    for (int q = 0; q < 4; q++) gg[q] = (am[q] == ro) ? gg[q] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (bs.relu && !(zz[q] * kS[q] + kH[q] > 0.0f)) gg[q] = 0.0f;
                    o[q] = kA[q] * gg[q] + kB[q] + kC[q] * zz[q];
                    if (r >= r_end || nb + q >= cout) o[q] = 0.0f; // padding rows / channels stay zero
                }
                v = make_float4(o[0], o[1], o[2], o[3]);
            }
            *reinterpret_cast<float4 *>(&Bs[buf][b_row + h * RB][b_q * 4]) = v;
        }
    };

    // prologue: slab 0 -> LDS, slab 1 -> registers, idx of slab 2 -> pidx
    load_idx(r_begin);
    load_slab(r_begin);
    store_slab(0);
    load_idx(r_begin + WG_BR);
    if (r_begin + WG_BR < r_end) load_slab(r_begin + WG_BR);
    load_idx(r_begin + 2 * WG_BR);
    __syncthreads();
    int buf = 0;
    const int kh = lane >> 5, l31 = lane & 31;
    for (long r0 = r_begin; r0 < r_end; r0 += WG_BR) {
        const bool have_next = r0 + WG_BR < r_end;
        float fa[2][TI], fb[2][TJ]; // register double-buffered fragments: reads of k2+1 issued before MFMAs of k2
#pragma unroll
        for (int t = 0; t < TI; t++) fa[0][t] = As[buf][kh][(wi * TI + t) * 32 + l31];
#pragma unroll
        for (int t = 0; t < TJ; t++) fb[0][t] = Bs[buf][kh][(wj * TJ + t) * 32 + l31];
#pragma unroll
        for (int k2 = 0; k2 < WG_BR / 2; k2++) {
            if (k2 == WG_BR / 4 && have_next) {
                store_slab(buf ^ 1); // slab s+1: the other buffer was last read one step ago, behind a barrier
                if (r0 + 2 * WG_BR < r_end) load_slab(r0 + 2 * WG_BR); // slab s+2 (GATHER: with the idx fetched a step ago)
                load_idx(r0 + 3 * WG_BR);
            }
            if (k2 + 1 < WG_BR / 2) {
#pragma unroll
                for (int t = 0; t < TI; t++) fa[(k2 + 1) & 1][t] = As[buf][(k2 + 1) * 2 + kh][(wi * TI + t) * 32 + l31];
#pragma unroll
                for (int t = 0; t < TJ; t++) fb[(k2 + 1) & 1][t] = Bs[buf][(k2 + 1) * 2 + kh][(wj * TJ + t) * 32 + l31];
            }
            __builtin_amdgcn_sched_barrier(0); // keep the reads of k2+1 ahead of the MFMAs of k2
#pragma unroll
            for (int s = 0; s < TI; s++)
#pragma unroll
                for (int t = 0; t < TJ; t++)
                    acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k2 & 1][s], fb[k2 & 1][t], acc[s][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier(); // LDS only: the prefetched global loads stay in flight across it
        buf ^= 1;
    }
    // epilogue: atomically add the partial tile.  C/D: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5)
#pragma unroll
    for (int s = 0; s < TI; s++)
#pragma unroll
        for (int t = 0; t < TJ; t++) {
            const int j = j0 + (wj * TJ + t) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = i0 + (wi * TI + s) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (i < cin && j < cout) {
                    const size_t off = (size_t)w_row<MODE>(i, in.c) * cout + j;
                    if (bs.part) bs.part[(size_t)blockIdx.x * bs.pstride + off] = acc[s][t][e];
                    else unsafeAtomicAdd(&dw[off], acc[s][t][e]);
                }
            }
        }
}

// ---------------------------------------------------------------- first-layer input gradient scatter
// Gradient of the sample_and_group concat: the per-row input gradients, given separately for the feature
// columns d_rows_feat (rows x c) and the xyz columns d_rows_xyz (rows x 3), go back to
//   d_feat[s, idx[r], :]  += d_rows_feat[r, :]
//   d_xyz [s, idx[r], :]  += d_rows_xyz[r, :] ,   d_new_xyz[s, j, :] -= sum_k d_rows_xyz[r, :]
// One thread per (group, channel) walks the group's nsample rows.  A ball with fewer than nsample
// neighbours is padded with its FIRST hit (tf_grouping_g.cu:26-29), so rows k >= pts_cnt all target
// idx[g,0]: they are summed in a register and cost ONE atomic instead of nsample - pts_cnt.
__global__ __launch_bounds__(256) void group_concat_grad_kernel(long groups, int n, int cc /* c or 3 */, int groups_per_scene,
                                                                int nsample, const float *__restrict__ d_rows,
                                                                const int *__restrict__ idx, const int *__restrict__ pts_cnt,
                                                                float *__restrict__ d_table, float *__restrict__ d_new_xyz)
{
    const long total = groups * cc;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / cc;
        const int ch = (int)(e - g * cc);
        const long s = g / groups_per_scene;
        const int *__restrict__ gi = idx + (size_t)g * nsample;
        const float *__restrict__ dr = d_rows + (size_t)g * nsample * cc + ch;
        float *__restrict__ tab = d_table + (size_t)s * n * cc + ch;
        int cnt = pts_cnt ? pts_cnt[g] : nsample;
        if (cnt < 1) cnt = 1;
        float pad = 0.0f, all = 0.0f;
        for (int k = 0; k < nsample; k++) {
            const float v = dr[(size_t)k * cc];
            all += v;
            if (k > 0 && k < cnt)
                unsafeAtomicAdd(&tab[(size_t)gi[k] * cc], v);
            else
                pad += v; // row 0 and the padding rows share idx[g,0]
        }
        unsafeAtomicAdd(&tab[(size_t)gi[0] * cc], pad);
        if (d_new_xyz) d_new_xyz[(size_t)g * 3 + ch] -= all; // one thread per (group, axis): no atomic needed
    }
}

// ---------------------------------------------------------------- optimizer
// sum of squares of every tensor's gradient segment: grid (kSumsqSlices slices, ntensors) -> out[tensor * kSumsqSlices + slice]; the
// optimizer adds the partials in slice order (no atomics: every data-parallel replica must compute bit-identical clip factors from
// the same all-reduced gradient, or the replicas drift apart).  32 slices and four loads in flight per thread: the largest tensors
// (512 x 256) bound the launch -- 27 -> see profiles (8 slices, one load at a time)
constexpr int kSumsqSlices = VOTENET_SUMSQ_SLICES;
__global__ __launch_bounds__(256) void seg_sumsq_kernel(const float *__restrict__ g, const long *__restrict__ seg,
                                                        float *__restrict__ out)
{
    __shared__ float sh[256];
    const long a = seg[2 * blockIdx.y], b = seg[2 * blockIdx.y + 1];
    const long step = 256L * gridDim.x;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    long i = a + (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < b; i += 4 * step) {
        const float v0 = g[i], v1 = g[i + step], v2 = g[i + 2 * step], v3 = g[i + 3 * step];
        s0 += v0 * v0;
        s1 += v1 * v1;
        s2 += v2 * v2;
        s3 += v3 * v3;
    }
    for (; i < b; i += step) s0 += g[i] * g[i];
    sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.y * kSumsqSlices + blockIdx.x] = sh[0];
}

// tf.clip_by_average_norm(g, clip): g * clip / max(||g||/numel, clip)   (model.py:249), then tf.train.AdamOptimizer
// (model.py:246) in TensorFlow's form: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  p -= lr_t * m / (sqrt(v) + eps)
// -- epsilon is NOT rescaled by the bias correction there ("epsilon hat" of the Adam paper, section 2)
__global__ void clip_adam_kernel(const long *__restrict__ seg, const float *__restrict__ sumsq, float *__restrict__ p,
                                 const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, float lr, float b1,
                                 float b2, float eps, float bc1, float bc2, float gscale, float clip)
{
    const long a = seg[2 * blockIdx.y], b = seg[2 * blockIdx.y + 1];
    float factor = gscale;
    if (clip > 0.0f) {
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < kSumsqSlices; t++) ss += sumsq[blockIdx.y * kSumsqSlices + t];
        const float avg = sqrtf(ss) * gscale / (float)(b - a);
        factor = gscale * clip / (avg > clip ? avg : clip);
    }
    const float lr_t = lr * sqrtf(bc2) / bc1;
    for (long i = a + (long)blockIdx.x * blockDim.x + threadIdx.x; i < b; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i] * factor;
        const float mm = b1 * m[i] + (1.0f - b1) * gg;
        const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] -= lr_t * mm / (sqrtf(vv) + eps);
    }
}

static inline int grid_for(long total, int block, int cap = 256 * 16)
{
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static MlpIn to_dev(const votenet_mlp_input *in)
{
    MlpIn d = {};
    d.x = in->x;
    d.in_scale = in->in_scale;
    d.in_shift = in->in_shift;
    d.in_relu = in->in_relu;
    d.xyz = in->xyz;
    d.new_xyz = in->new_xyz;
    d.feat = in->feat;
    d.idx = in->idx;
    d.n = in->n;
    d.m = in->m;
    d.nsample = in->nsample;
    d.c = in->feat ? in->c : 0;
    return d;
}

} // namespace votenet

using namespace votenet;

static int g_bn_reduce_passes = 16; // (32: 8-64 workgroups, 14-18 us per launch; 16: 10-12 us; 8: the 2*c atomics of 256-512 workgroups cost more than they buy -- tools/serial_last_step.sh)
extern "C" void votenet_debug_bn_reduce_passes(int n) { VN_DEBUG_GATE(); g_bn_reduce_passes = n > 0 ? n : 16; } // tuning hook
extern "C" int votenet_bn_backward_reduce(long rows, int c, int k, const float *da, const int *argmax, const float *z,
                                          const float *scale, const float *shift, const float *mean, const float *var,
                                          float eps, int relu, double *sums, const votenet_coef_tail *tail_, void *stream)
{
    VN_REQUIRE(!tail_ || (tail_->ticket && tail_->gamma && tail_->coef && tail_->rows > 0), "bn_backward_reduce: incomplete coefficient tail");
    const CoefTail tail = to_tail(tail_);
    VN_REQUIRE(rows > 0 && c > 0 && k >= 0, "bn_backward_reduce expects rows > 0, c > 0, k >= 0");
    VN_REQUIRE(da && z && scale && shift && mean && var && sums, "bn_backward_reduce: null buffer");
    hipStream_t st = as_stream(stream);
    const int ny = (c + 63) / 64;
    if (k > 0) {
        VN_REQUIRE(argmax != nullptr && rows % k == 0, "bn_backward_reduce: pooled mode needs argmax and rows % k == 0");
        const long groups = rows / k;
        hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel, dim3(grid_for(groups, 4, 1024 / ny + 1), ny), dim3(256), 0, st, groups, k, c,
                           da, argmax, z, scale, shift, mean, var, eps, relu, sums, tail);
    } else if (c % 4 == 0 && c <= 1024 && 256 % (c / 4) == 0 && (uintptr_t)da % 16 == 0 && (uintptr_t)z % 16 == 0 &&
               (uintptr_t)scale % 16 == 0 && (uintptr_t)shift % 16 == 0 && (uintptr_t)mean % 16 == 0 && (uintptr_t)var % 16 == 0) {
        const long rpp = 256 / (c / 4);
        long gx = (rows + g_bn_reduce_passes * rpp - 1) / (g_bn_reduce_passes * rpp); // passes per workgroup: its 2*c atomics must amortise
        if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(bn_bwd_reduce_dense_vec_kernel, dim3((unsigned)gx), dim3(256), 0, st, rows, c, da, z, scale, shift, mean,
                           var, eps, relu, sums, tail);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_dense_kernel, dim3(grid_for(rows, 4, 2048 / ny + 1), ny), dim3(256), 0, st, rows, c, da,
                           z, scale, shift, mean, var, eps, relu, sums, tail);
    }
    return check_launch("bn_backward_reduce");
}

extern "C" int votenet_bn_backward_apply(long rows, int c, int k, const float *da, const int *argmax, const float *z,
                                         const float *coef, int relu, float *dz, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0 && k >= 0, "bn_backward_apply expects rows > 0, c > 0, k >= 0");
    VN_REQUIRE(da && z && coef && dz, "bn_backward_apply: null buffer");
    VN_REQUIRE(k == 0 || argmax != nullptr, "bn_backward_apply: pooled mode needs argmax");
    hipStream_t st = as_stream(stream);
    const bool vec = (c % 4 == 0) && ((uintptr_t)z % 16 == 0) && ((uintptr_t)dz % 16 == 0) && (k > 0 || (uintptr_t)da % 16 == 0);
    if (vec)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3(grid_for(rows * (c / 4), 256)), dim3(256), 0, st, rows, c, k, da, argmax, z,
                           coef, relu, dz);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3(grid_for(rows * c, 256)), dim3(256), 0, st, rows, c, k, da, argmax, z,
                           coef, relu, dz);
    return check_launch("bn_backward_apply");
}

extern "C" int votenet_bias_grad(long rows, int c, const float *dz, double *scratch, float *dbias, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0, "bias_grad expects rows > 0, c > 0");
    VN_REQUIRE(dz && scratch && dbias, "bias_grad: null buffer");
    hipStream_t st = as_stream(stream);
    (void)hipMemsetAsync(scratch, 0, sizeof(double) * c, st);
    const int ny = (c + 63) / 64;
    hipLaunchKernelGGL(colsum_kernel, dim3(grid_for(rows, 4, 1024 / ny + 1), ny), dim3(256), 0, st, rows, c, dz, c, scratch);
    hipLaunchKernelGGL(add_f64_to_f32_kernel, dim3((c + 255) / 256), dim3(256), 0, st, c, scratch, dbias);
    return check_launch("bias_grad");
}

// The same for the first c columns of a wider row-major tensor (rows of `pitch` floats: the zero-padded gradient of a ragged
// layer) and a scratch the CALLER has zeroed (a carve-out of the pass's one fill: no memset here).
extern "C" int votenet_bias_grad_strided(long rows, int c, const float *dz, int pitch, double *zeroed_scratch, float *dbias, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0 && pitch >= c, "bias_grad expects rows > 0, 0 < c <= pitch");
    VN_REQUIRE(dz && zeroed_scratch && dbias, "bias_grad: null buffer");
    hipStream_t st = as_stream(stream);
    const int ny = (c + 63) / 64;
    hipLaunchKernelGGL(colsum_kernel, dim3(grid_for(rows, 4, 1024 / ny + 1), ny), dim3(256), 0, st, rows, c, dz, pitch, zeroed_scratch);
    hipLaunchKernelGGL(add_f64_to_f32_kernel, dim3((c + 255) / 256), dim3(256), 0, st, c, zeroed_scratch, dbias);
    return check_launch("bias_grad_strided");
}

// Weight gradient of the few "narrow" input channels of a GATHER layer: the three dxyz columns (W rows 0..2)
// and, when the layer has at most 5 feature channels (sa1: C = 3), those too (W rows 3..).  A 128-wide MFMA
// tile would be ~97 % padding here; this is a streaming reduction over dz instead: every workgroup stages the
// narrow rows of 256 grouped points in LDS, thread (column j, row subset) accumulates nch partial sums.
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(MlpIn in, long rows, int nch, int cout, const float *__restrict__ dz,
                                                           float *__restrict__ dw, long rows_per_block, float *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float As[256][8];
    __shared__ float red[256][8];
    const int tid = threadIdx.x;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    long r_end = r_begin + rows_per_block;
    if (r_end > rows) r_end = rows;
    if (r_begin >= r_end) return;
    const int nsub = 256 / cout > 0 ? 256 / cout : 1; // row subsets when cout < 256
    const int j = tid % cout, sub = tid / cout;        // cout in {64,128,256}: every thread has a column
    const bool active = sub < nsub;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = 0.0f;
    for (long c0 = r_begin; c0 < r_end; c0 += 256) {
        const long r = c0 + tid;
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = 0.0f;
        if (r < r_end) {
            const int src = in.idx[r];
            const long scene = r / ((long)in.m * in.nsample);
#pragma unroll
            for (int d = 0; d < 3; d++)
                a[d] = in.xyz[((size_t)scene * in.n + src) * 3 + d] - in.new_xyz[(size_t)(r / in.nsample) * 3 + d];
            for (int q = 3; q < nch; q++) a[q] = in.feat[((size_t)scene * in.n + src) * in.c + (q - 3)];
        }
        __syncthreads(); // previous chunk fully consumed
        *reinterpret_cast<float4 *>(&As[tid][0]) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4 *>(&As[tid][4]) = make_float4(a[4], a[5], a[6], a[7]);
        __syncthreads();
        const int nrow = (int)((r_end - c0) < 256 ? (r_end - c0) : 256);
        if (active)
            for (int rr = sub; rr < nrow; rr += nsub) {
                const float g = dz[(size_t)(c0 + rr) * cout + j];
                const float4 a0 = *reinterpret_cast<const float4 *>(&As[rr][0]);
                const float4 a1 = *reinterpret_cast<const float4 *>(&As[rr][4]);
                acc[0] += a0.x * g; acc[1] += a0.y * g; acc[2] += a0.z * g; acc[3] += a0.w * g;
                acc[4] += a1.x * g; acc[5] += a1.y * g; acc[6] += a1.z * g; acc[7] += a1.w * g;
            }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) red[tid][q] = acc[q];
    __syncthreads();
    if (tid < cout) {
        for (int q = 0; q < nch; q++) {
            float t = 0.0f;
            for (int u = 0; u < nsub; u++) t += red[u * cout + tid][q];
            // narrow channel q is W row q (dxyz first, utils.py:55); part: this workgroup's slice (see BnSrc::part)
            if (part) part[((size_t)blockIdx.x * nch + q) * cout + tid] = t;
            else unsafeAtomicAdd(&dw[(size_t)q * cout + tid], t);
        }
    }
}

// dw[e] += part[0][e] + part[1][e] + ... in ONE fixed order, e in [e0, e1): the ordered reduction behind BnSrc::part.
// Workgroup = 64 elements x 4 slice quarters: thread (element, quarter q) adds its quarter of the slices in ascending order,
// eight loads in flight; the four quarter sums meet in LDS and are added in quarter order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(int nslice, long pstride, long e0, long e1, const float *__restrict__ part,
                                                           float *__restrict__ dw)
{
    __shared__ float s_q[4][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long e = e0 + (long)blockIdx.x * 64 + el;
    const int per = (nslice + 3) / 4;
    const int t0 = q * per, t1 = (t0 + per) < nslice ? (t0 + per) : nslice;
    float s = 0.0f;
    if (e < e1) {
        int t = t0;
        for (; t + 8 <= t1; t += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = part[(size_t)(t + u) * pstride + e];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; t < t1; t++) s += part[(size_t)t * pstride + e];
    }
    s_q[q][el] = s;
    __syncthreads();
    if (q == 0 && e < e1) dw[e] += ((s_q[0][el] + s_q[1][el]) + s_q[2][el]) + s_q[3][el];
}
namespace votenet {
void wgrad_reduce(int nslice, long pstride, long e0, long e1, const float *part, float *dw, hipStream_t st)
{
    if (e1 > e0) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((e1 - e0 + 63) / 64)), dim3(256), 0, st, nslice, pstride, e0, e1, part, dw);
}
}

// row ranges of the generic kernel: grid.x slices of rpb rows
static void plan_wgrad(long rows, int cin, int cout, int &TIr, int &TJr, int &ti, int &tj, long &rpb, unsigned &gx, bool partials = false)
{
    TIr = cin <= 64 ? 1 : 2;
    TJr = cout <= 64 ? 1 : 2;
    const int BI = 64 * TIr, BJ = 64 * TJr;
    ti = (cin + BI - 1) / BI;
    tj = (cout + BJ - 1) / BJ;
    long splits = (partials ? 256 : 768) / (ti * tj); // partial tiles: every slice is written and read once more
    if (splits < 1) splits = 1;
    rpb = (rows + splits - 1) / splits;
    rpb = (rpb + WG_BR - 1) / WG_BR * WG_BR;
    if (rpb < 8 * WG_BR) rpb = 8 * WG_BR; // short row ranges are dominated by the flush of the dW tile (measured)
    gx = (unsigned)((rows + rpb - 1) / rpb);
}

// wrows: rows of the dw block the launch addresses (MODE 0: cin; MODE 1: 3 + in.c); r0: first of them it writes
template <int MODE, int BSRC>
static void launch_wgrad(const MlpIn &d, long rows, int cin, int cout, const float *dz, BnSrc bs, float *dw, hipStream_t st, float *scratch,
                         int wrows, int r0)
{
    int TIr, TJr, ti, tj;
    long rpb;
    unsigned gx;
    plan_wgrad(rows, cin, cout, TIr, TJr, ti, tj, rpb, gx, scratch != nullptr);
    bs.part = scratch;
    bs.pstride = (long)wrows * cout;
    const dim3 grid(gx, ti, tj);
    if (TIr == 2 && TJr == 2)
        hipLaunchKernelGGL((mlp_wgrad_kernel<MODE, 2, 2, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    else if (TIr == 2)
        hipLaunchKernelGGL((mlp_wgrad_kernel<MODE, 2, 1, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    else if (TJr == 2)
        hipLaunchKernelGGL((mlp_wgrad_kernel<MODE, 1, 2, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    else
        hipLaunchKernelGGL((mlp_wgrad_kernel<MODE, 1, 1, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    if (scratch) votenet::wgrad_reduce((int)gx, bs.pstride, (long)r0 * cout, (long)wrows * cout, scratch, dw, st);
}

namespace votenet {
bool wgrad_fast_launch(int mode, const MlpIn &d, long rows, int cin, int cout, const float *dz, const BnSrc &bs, int bsrc, float *dw,
                       hipStream_t st, float *scratch); // mlp_wgrad_fast.hip
long wgrad_fast_slices(long rows, int cin, int cout); // its grid.x
}

static long narrow_rpb(long rows)
{
    long rpb = (rows + 1023) / 1024;
    return (rpb + 255) / 256 * 256;
}

// floats of scratch that make every launch of wgrad_entry deterministic (an upper bound over the kernels it may choose)
static size_t wgrad_scratch_floats(const votenet_mlp_input *in, long rows, int cin, int cout)
{
    if (rows <= 0 || cin <= 0 || cout <= 0) return 0;
    int TIr, TJr, ti, tj;
    long rpb;
    unsigned gx;
    plan_wgrad(rows, cin, cout, TIr, TJr, ti, tj, rpb, gx);
    size_t need = (size_t)gx * cin * cout;
    const size_t fast = (size_t)wgrad_fast_slices(rows, cin, cout) * cin * cout;
    if (fast > need) need = fast;
    if (in && !in->x) {
        const long rn = narrow_rpb(rows);
        const size_t nar = (size_t)((rows + rn - 1) / rn) * 8 * cout;
        if (nar > need) need = nar;
    }
    return need;
}

static int wgrad_entry(const votenet_mlp_input *in, long rows, int cin, int cout, const float *dz, const BnSrc &bs, int bsrc,
                       float *dw, float *scratch, void *stream)
{
    VN_REQUIRE(in != nullptr, "mlp_wgrad: null input descriptor");
    VN_REQUIRE(rows >= 0 && cin > 0 && cout > 0, "mlp_wgrad expects rows >= 0, cin > 0, cout > 0");
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(dw != nullptr, "mlp_wgrad: null buffer");
    MlpIn d = to_dev(in);
    hipStream_t st = as_stream(stream);
    if (in->x) {
        VN_REQUIRE((in->in_scale == nullptr) == (in->in_shift == nullptr), "mlp_wgrad: in_scale and in_shift go together");
        if (wgrad_fast_launch(0, d, rows, cin, cout, dz, bs, bsrc, dw, st, scratch)) return check_launch("mlp_wgrad");
        if (bsrc == 0) launch_wgrad<0, 0>(d, rows, cin, cout, dz, bs, dw, st, scratch, cin, 0);
        else if (bsrc == 1) launch_wgrad<0, 1>(d, rows, cin, cout, dz, bs, dw, st, scratch, cin, 0);
        else launch_wgrad<0, 2>(d, rows, cin, cout, dz, bs, dw, st, scratch, cin, 0);
    } else {
        VN_REQUIRE(in->xyz && in->new_xyz && in->idx, "mlp_wgrad: GATHER input needs xyz, new_xyz and idx");
        VN_REQUIRE(rows == (long)in->b * in->m * in->nsample, "mlp_wgrad: rows must equal b*m*nsample for a GATHER input");
        VN_REQUIRE(cin == 3 + d.c, "mlp_wgrad: cin must equal 3 + c for a GATHER input");
        const bool narrow_ok = bsrc == 0 && (cout == 64 || cout == 128 || cout == 256);
        if (narrow_ok) {
            // dxyz (and <= 5 feature) columns: streaming reduction; wide feature block: MFMA kernel on W rows 3..
            const int nch = d.c <= 5 ? 3 + d.c : 3;
            const long rpb = narrow_rpb(rows);
            const unsigned gx = (unsigned)((rows + rpb - 1) / rpb);
            hipLaunchKernelGGL(wgrad_narrow_kernel, dim3(gx), dim3(256), 0, st, d, rows, nch, cout, dz, dw, rpb, scratch);
            if (scratch) votenet::wgrad_reduce((int)gx, (long)nch * cout, 0, (long)nch * cout, scratch, dw, st);
            if (d.c > 5 && !wgrad_fast_launch(1, d, rows, d.c, cout, dz, bs, 0, dw, st, scratch))
                launch_wgrad<1, 0>(d, rows, d.c, cout, dz, bs, dw, st, scratch, 3 + d.c, 3); // internal k < c are the feature channels
        } else {
            if (bsrc == 0) launch_wgrad<1, 0>(d, rows, cin, cout, dz, bs, dw, st, scratch, cin, 0);
            else if (bsrc == 1) launch_wgrad<1, 1>(d, rows, cin, cout, dz, bs, dw, st, scratch, cin, 0);
            else launch_wgrad<1, 2>(d, rows, cin, cout, dz, bs, dw, st, scratch, cin, 0);
        }
    }
    return check_launch("mlp_wgrad");
}

extern "C" size_t votenet_mlp_wgrad_scratch_floats(const votenet_mlp_input *in, long rows, int cin, int cout)
{
    return wgrad_scratch_floats(in, rows, cin, cout);
}

extern "C" int votenet_mlp_wgrad(const votenet_mlp_input *in, long rows, int cin, int cout, const float *dz, float *dw,
                                 float *scratch, void *stream)
{
    VN_REQUIRE(dz != nullptr || rows == 0, "mlp_wgrad: null dz");
    BnSrc bs = {};
    return wgrad_entry(in, rows, cin, cout, dz, bs, 0, dw, scratch, stream);
}

// dw += input^T * dz with dz = BatchNorm-backward(da | pooled gout, z, coef) formed inside the dz-operand loader
extern "C" int votenet_mlp_wgrad_bn(const votenet_mlp_input *in, long rows, int cin, int cout, const float *da, const float *gout,
                                    const int *argmax, int pool_k, const float *z, const float *coef, int relu, float *dw,
                                    float *scratch, void *stream)
{
    VN_REQUIRE((da != nullptr) != (gout != nullptr), "mlp_wgrad_bn: exactly one of da / gout");
    VN_REQUIRE(z && coef, "mlp_wgrad_bn: null buffer");
    VN_REQUIRE(gout == nullptr || (argmax != nullptr && pool_k > 0 && rows % pool_k == 0), "mlp_wgrad_bn: pooled source needs argmax and k");
    VN_REQUIRE(rows < (1L << 31), "mlp_wgrad_bn: rows must be below 2^31");
    int shift = -1;
    if (pool_k > 0 && (pool_k & (pool_k - 1)) == 0) shift = __builtin_ctz((unsigned)pool_k);
    BnSrc bs = {da, gout, argmax, pool_k, shift, z, coef, relu, nullptr, 0};
    return wgrad_entry(in, rows, cin, cout, nullptr, bs, da ? 1 : 2, dw, scratch, stream);
}

// Weight gradient of the SECOND layer of a chain whose first layer is NARROW (narrow.hip): dw (c0 x cout) += act(z0)^T dz1 with
// z0 rebuilt from u8 in the loader (act = relu(z0*in_scale+in_shift)) and dz1 from (da, z, coef) as votenet_mlp_wgrad_bn.
extern "C" int votenet_narrow_wgrad_bn(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                                       const float *in_scale, const float *in_shift, int in_relu, const float *da, const float *z,
                                       const float *coef, int relu, float *dw, float *scratch, void *stream)
{
    VN_REQUIRE(rows > 0 && rows < (1L << 31) && k0 >= 3 && k0 <= 8 && c0 > 0 && cout > 0, "narrow_wgrad_bn: bad shape");
    VN_REQUIRE(u8 && w0 && in_scale && in_shift && da && z && coef && dw, "narrow_wgrad_bn: null buffer");
    MlpIn d = {};
    d.in_scale = in_scale;
    d.in_shift = in_shift;
    d.in_relu = in_relu;
    d.u8 = u8;
    d.w0 = w0;
    d.b0 = b0;
    d.k0 = k0;
    BnSrc bs = {da, nullptr, nullptr, 0, -1, z, coef, relu, nullptr, 0};
    if (!wgrad_fast_launch(2, d, rows, c0, cout, nullptr, bs, 1, dw, as_stream(stream), scratch))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "narrow_wgrad_bn: shape not served (c0 %% 64 == 0, cout %% 64 == 0, 16-byte aligned buffers)");
    return check_launch("narrow_wgrad_bn");
}

// votenet_narrow_wgrad_bn on the piece layout (half.hip): da holds totals per compact row, the affine part of dz1 is weighted.
extern "C" int votenet_narrow_wgrad_bn_half(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                                            const float *in_scale, const float *in_shift, int in_relu, const float *da, const float *z,
                                            const float *coef, int relu, const float *wh, float *dw, void *stream)
{
    VN_REQUIRE(rows > 0 && rows % 32 == 0 && rows < (1L << 31) && k0 >= 3 && k0 <= 8 && c0 > 0 && cout > 0, "narrow_wgrad_bn_half: bad shape");
    VN_REQUIRE(u8 && w0 && in_scale && in_shift && da && z && coef && wh && dw, "narrow_wgrad_bn_half: null buffer");
    MlpIn d = {};
    d.in_scale = in_scale;
    d.in_shift = in_shift;
    d.in_relu = in_relu;
    d.u8 = u8;
    d.w0 = w0;
    d.b0 = b0;
    d.k0 = k0;
    BnSrc bs = {da, nullptr, nullptr, 0, -1, z, coef, relu, nullptr, 0, wh};
    if (!wgrad_fast_launch(2, d, rows, c0, cout, nullptr, bs, 4, dw, as_stream(stream), nullptr))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "narrow_wgrad_bn_half: shape not served (as votenet_narrow_wgrad_bn)");
    return check_launch("narrow_wgrad_bn_half");
}

// Weight gradient of the SECOND layer of a chain whose first layer is ASSEMBLED (assemble.hip): dw (c0 x cout) += act(z0)^T dz1 with
// z0[r,:] = P[prow(r),:] + dxyz(r) . wx rebuilt in the loader (act = relu(z0*in_scale+in_shift)), dz1 from (da, z, coef).
extern "C" int votenet_assembled_wgrad_bn(long rows, int c0, int cout, const float *geo, const float *P, const float *wx,
                                          const float *in_scale, const float *in_shift, int in_relu, const float *da, const float *z,
                                          const float *coef, int relu, float *dw, float *scratch, void *stream)
{
    VN_REQUIRE(rows > 0 && rows < (1L << 31) && c0 > 0 && cout > 0, "assembled_wgrad_bn: bad shape");
    VN_REQUIRE(geo && P && wx && in_scale && in_shift && da && z && coef && dw, "assembled_wgrad_bn: null buffer");
    MlpIn d = {};
    d.in_scale = in_scale;
    d.in_shift = in_shift;
    d.in_relu = in_relu;
    d.geo = geo;
    d.ptab = P;
    d.wx = wx;
    BnSrc bs = {da, nullptr, nullptr, 0, -1, z, coef, relu, nullptr, 0};
    if (!wgrad_fast_launch(3, d, rows, c0, cout, nullptr, bs, 1, dw, as_stream(stream), scratch))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "assembled_wgrad_bn: shape not served (c0 %% 64 == 0, cout %% 64 == 0, 16-byte aligned buffers)");
    return check_launch("assembled_wgrad_bn");
}

// The same on the piece layout (half.hip): da holds TOTAL gradients per compact row, the affine part of the rebuilt dz1 counts wh[q]
// times on row 16 q.
extern "C" int votenet_assembled_wgrad_bn_half(long rows, int c0, int cout, const float *geo, const float *P, const float *wx,
                                               const float *in_scale, const float *in_shift, int in_relu, const float *da, const float *z,
                                               const float *coef, int relu, const float *wh, float *dw, const int *nh_dev, void *stream)
{
    VN_REQUIRE(rows > 0 && rows % 32 == 0 && rows < (1L << 31) && c0 > 0 && cout > 0, "assembled_wgrad_bn_half: bad shape");
    VN_REQUIRE(geo && P && wx && in_scale && in_shift && da && z && coef && wh && dw, "assembled_wgrad_bn_half: null buffer");
    MlpIn d = {};
    d.in_scale = in_scale;
    d.in_shift = in_shift;
    d.in_relu = in_relu;
    d.geo = geo;
    d.ptab = P;
    d.wx = wx;
    BnSrc bs = {da, nullptr, nullptr, 0, -1, z, coef, relu, nullptr, 0, wh, nh_dev};
    if (!wgrad_fast_launch(3, d, rows, c0, cout, nullptr, bs, 4, dw, as_stream(stream), nullptr))
        return set_error(VOTENET_E_INVALID_ARGUMENT, "assembled_wgrad_bn_half: shape not served (as votenet_assembled_wgrad_bn)");
    return check_launch("assembled_wgrad_bn_half");
}

// coefficient vector [A | B | C | scale | shift] (5*c floats) of the folded BatchNorm backward, from the reductions
// sums = [sum g', sum g'*zhat]; also dgamma += sums[c:], dbeta += sums[:c] (each may be NULL)
extern "C" int votenet_bn_backward_coef(long rows, int c, const float *scale, const float *shift, const float *mean,
                                        const float *var, float eps, const float *gamma, const double *sums, float *coef,
                                        float *dgamma, float *dbeta, void *stream)
{
    VN_REQUIRE(rows > 0 && c > 0, "bn_backward_coef expects rows > 0, c > 0");
    VN_REQUIRE(scale && shift && mean && var && gamma && sums && coef, "bn_backward_coef: null buffer");
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), rows, c, scale, shift, mean, var,
                       eps, gamma, sums, coef, dgamma, dbeta);
    return check_launch("bn_backward_coef");
}

extern "C" int votenet_group_concat_grad(int b, int n, int c, int m, int nsample, const float *d_rows_feat,
                                         const float *d_rows_xyz, const int *idx, const int *pts_cnt, float *d_feat,
                                         float *d_xyz, float *d_new_xyz, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && c >= 0 && m >= 0 && nsample > 0, "group_concat_grad: bad shape");
    const long groups = (long)b * m;
    if (groups == 0) return VOTENET_OK;
    VN_REQUIRE(idx != nullptr, "group_concat_grad: null idx");
    hipStream_t st = as_stream(stream);
    if (d_rows_feat && d_feat && c > 0)
        hipLaunchKernelGGL(group_concat_grad_kernel, dim3(grid_for(groups * c, 256)), dim3(256), 0, st, groups, n, c, m, nsample,
                           d_rows_feat, idx, pts_cnt, d_feat, (float *)nullptr);
    if (d_rows_xyz && d_xyz)
        hipLaunchKernelGGL(group_concat_grad_kernel, dim3(grid_for(groups * 3, 256)), dim3(256), 0, st, groups, n, 3, m, nsample,
                           d_rows_xyz, idx, pts_cnt, d_xyz, d_new_xyz);
    return check_launch("group_concat_grad");
}

extern "C" int votenet_clip_adam(int ntensors, const long *seg, float *sumsq_scratch, float *p, const float *g, float *m,
                                 float *v, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                                 float clip_avg_norm, void *stream)
{
    VN_REQUIRE(ntensors > 0 && step > 0, "clip_adam expects ntensors > 0 and step >= 1");
    VN_REQUIRE(seg && sumsq_scratch && p && g && m && v, "clip_adam: null buffer");
    hipStream_t st = as_stream(stream);
    if (clip_avg_norm > 0.0f) {
        hipLaunchKernelGGL(seg_sumsq_kernel, dim3(kSumsqSlices, ntensors), dim3(256), 0, st, g, seg, sumsq_scratch);
    }
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(64, ntensors), dim3(256), 0, st, seg, sumsq_scratch, p, g, m, v, lr, beta1, beta2, eps,
                       bc1, bc2, grad_scale, clip_avg_norm);
    return check_launch("clip_adam");
}

// ---------------------------------------------------------------- weight transposes for the input-gradient GEMMs
// da = dz . W^T wants W^T (cout x cin) row-major.  One launch transposes every weight block of the flat parameter bucket
// into a parallel bucket: table[e] = {src offset, dst offset, rows, cols} (elements); one workgroup per entry, 32 x 32
// tiles through LDS (coalesced on both sides).  The weights do not change between the forward and the backward pass of a
// step, so the launch rides on a side stream underneath the sa1 farthest-point sampling.
namespace votenet {
__global__ __launch_bounds__(256) void transpose_segments_kernel(const long *__restrict__ table, const float *__restrict__ src,
                                                                float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const long so = table[6 * blockIdx.x + 0], dof = table[6 * blockIdx.x + 1];
    const int rows = (int)table[6 * blockIdx.x + 2], cols = (int)table[6 * blockIdx.x + 3];
    const int ld = (int)table[6 * blockIdx.x + 4];
    const bool tr = table[6 * blockIdx.x + 5] != 0;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const int tr_ = (rows + 31) / 32, tc = (cols + 31) / 32;
    for (int t = blockIdx.y; t < tr_ * tc; t += gridDim.y) { // the 32 x 32 tiles of a segment are dealt to gridDim.y workgroups
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
        if (!tr) { // plain copy into a buffer with leading dimension ld >= cols (the padding stays as it was: zero)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = r0 + ty + 8 * k, c = c0 + tx;
                if (r < rows && c < cols) dst[dof + (long)r * ld + c] = src[so + (long)r * cols + c];
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[so + (long)r * cols + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = c0 + ty + 8 * k, r = r0 + tx; // dst[c][r], leading dimension ld >= rows
            if (c < cols && r < rows) dst[dof + (long)c * ld + r] = tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}

// out[r, d] = sum_k dz[r, k] * w3[d, k], d < 3: the xyz columns of an input gradient (dz W[0:3]^T) -- a 128-wide MFMA tile would be
// 97 % padding.  32 lanes per row (float4 each, c <= 128 per pass), two rows per wavefront, shuffle reduction.
__global__ __launch_bounds__(256) void rows_dot3_kernel(long rows, int c, const float *__restrict__ dz, const float *__restrict__ w3,
                                                        float *__restrict__ out)
{
    const int l = threadIdx.x & 31;
    for (long r = (long)blockIdx.x * 8 + (threadIdx.x >> 5); r < rows; r += (long)gridDim.x * 8) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int k = l * 4; k < c; k += 128) {
            const float4 v = *reinterpret_cast<const float4 *>(dz + (size_t)r * c + k);
            const float4 x = *reinterpret_cast<const float4 *>(w3 + k), y = *reinterpret_cast<const float4 *>(w3 + c + k);
            const float4 z = *reinterpret_cast<const float4 *>(w3 + 2 * c + k);
            a0 += v.x * x.x + v.y * x.y + v.z * x.z + v.w * x.w;
            a1 += v.x * y.x + v.y * y.y + v.z * y.z + v.w * y.w;
            a2 += v.x * z.x + v.y * z.y + v.z * z.z + v.w * z.w;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            a0 += __shfl_xor(a0, o);
            a1 += __shfl_xor(a1, o);
            a2 += __shfl_xor(a2, o);
        }
        if (l == 0) {
            out[(size_t)r * 3 + 0] = a0;
            out[(size_t)r * 3 + 1] = a1;
            out[(size_t)r * 3 + 2] = a2;
        }
    }
}
} // namespace votenet

extern "C" int votenet_transpose_segments(int nseg, const long *table, const float *src, float *dst, void *stream)
{
    VN_REQUIRE(nseg >= 0, "transpose_segments expects nseg >= 0");
    if (nseg == 0) return VOTENET_OK;
    VN_REQUIRE(table && src && dst, "transpose_segments: null buffer");
    // 16 workgroups per segment: one each took 0.3 ms for the step's ~40 segments (a 512 x 256 block is 128 tiles in a row)
    hipLaunchKernelGGL(votenet::transpose_segments_kernel, dim3(nseg, 16), dim3(256), 0, as_stream(stream), table, src, dst);
    return check_launch("transpose_segments");
}

extern "C" int votenet_rows_dot3(long rows, int c, const float *dz, const float *w3, float *out, void *stream)
{
    VN_REQUIRE(rows >= 0 && c > 0 && c % 4 == 0, "rows_dot3 expects rows >= 0, c a multiple of 4");
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(dz && w3 && out, "rows_dot3: null buffer");
    VN_REQUIRE((uintptr_t)dz % 16 == 0 && (uintptr_t)w3 % 16 == 0, "rows_dot3: dz and w3 must be 16-byte aligned");
    long gx = (rows + 7) / 8;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(votenet::rows_dot3_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), rows, c, dz, w3, out);
    return check_launch("rows_dot3");
}
