// narrow.hip -- a set-abstraction MLP whose grouped input is NARROW (3 + c <= 8 channels: sa1 of VoteNet groups the bare
// coordinates, model.py:39) and whose input points carry no gradient (gfx950).
//
// The first layer's output z0 = [xyz[idx]-new_xyz | feat[idx]] W0 + b0 (utils.py:50-57,125-127) is then a function of EIGHT floats
// per grouped row, so it is never written: every kernel that needs z0 rebuilds it from the row's u = (dx, dy, dz, f0..f4) with
// narrow_z (mlp_types.h) -- 32 bytes of u instead of 4*c0 bytes of z0 per row, for the second layer's GEMM, its weight-gradient
// GEMM and the BatchNorm backward of layer 0 alike.  What the rest of the first layer needs follows from sums over the rows:
//   moments   m[d] = sum_r u[r,d],  M[d,e] = sum_r u[r,d] u[r,e]                          (geometry only: votenet_narrow_rows)
//   BatchNorm statistics of z0:  sum z0[:,c] = m.W0[:,c] + N b0[c],  sum z0[:,c]^2 = W0[:,c]^T M W0[:,c] + 2 b0[c] m.W0[:,c] + N b0[c]^2
//   dW0[d,c] = sum_r u[r,d] dz0[r,c] with dz0 = A g + B + C z0 (BatchNorm + ReLU backward, coef = [A|B|C|S|H]):
//            = A[c] UG[d,c] + B[c] m[d] + C[c] (sum_e M[d,e] W0[e,c] + m[d] b0[c]),   UG[d,c] = sum_r u[r,d] g[r,c]
//   where UG and the BatchNorm-backward sums of layer 0 come out of the epilogue of the second layer's input-gradient GEMM
//   (mlp_fast.hip, EPI 4), which therefore stores nothing: the gradient with respect to z0's activation never reaches HBM,
//   and no GroupPointGrad scatter runs (nothing upstream wants it).
#include "mlp_types.h"

namespace votenet {

__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// thread = grouped row: u (8 floats, zero padded) + this workgroup's share of the moments
__global__ __launch_bounds__(256) void narrow_rows_kernel(long rows, int n, int groups_per_scene, int nsample, int c,
                                                          const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                          const float *__restrict__ feat, const int *__restrict__ idx,
                                                          float *__restrict__ u8, double *__restrict__ moments)
{
    __shared__ double red[4][44];
    const unsigned rows_per_scene = (unsigned)groups_per_scene * (unsigned)nsample;
    double acc[44]; // m[0..8), then the upper triangle of M row by row
#pragma unroll
    for (int i = 0; i < 44; i++) acc[i] = 0.0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const int id = idx[r];
        const size_t prow = (size_t)((unsigned)r / rows_per_scene) * n + id;
        const size_t g = (size_t)((unsigned)r / (unsigned)nsample);
        float u[8];
        u[0] = xyz[prow * 3 + 0] - new_xyz[g * 3 + 0]; // utils.py:55
        u[1] = xyz[prow * 3 + 1] - new_xyz[g * 3 + 1];
        u[2] = xyz[prow * 3 + 2] - new_xyz[g * 3 + 2];
#pragma unroll
        for (int d = 0; d < 5; d++) u[3 + d] = d < c ? feat[prow * c + d] : 0.0f;
        *reinterpret_cast<float4 *>(u8 + (size_t)r * 8) = make_float4(u[0], u[1], u[2], u[3]);
        *reinterpret_cast<float4 *>(u8 + (size_t)r * 8 + 4) = make_float4(u[4], u[5], u[6], u[7]);
        int t = 8;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            acc[d] += (double)u[d];
#pragma unroll
            for (int e = d; e < 8; e++) acc[t++] += (double)u[d] * (double)u[e];
        }
    }
    if (!moments) return;
#pragma unroll
    for (int i = 0; i < 44; i++) {
        double v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 44) {
        const double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        int i = threadIdx.x;
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

// z0 itself, for a caller that does want the layer output (tests; the product path never stores it): thread = (row, channel quad)
__global__ __launch_bounds__(256) void narrow_z0_kernel(long rows, int k0, int c0, const float *__restrict__ u8, const float *__restrict__ w0,
                                                        const float *__restrict__ b0, float *__restrict__ z0)
{
    const int qc = c0 >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * qc) return;
    const long r = t / qc;
    const int c = (int)(t % qc) * 4;
    const float4 ua = *reinterpret_cast<const float4 *>(u8 + (size_t)r * 8), ub = *reinterpret_cast<const float4 *>(u8 + (size_t)r * 8 + 4);
    const float uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float w[8];
#pragma unroll
        for (int d = 0; d < 8; d++) w[d] = d < k0 ? w0[(size_t)d * c0 + c + q] : 0.0f;
        o[q] = narrow_z(uu, w, b0 ? b0[c + q] : 0.0f);
    }
    *reinterpret_cast<float4 *>(z0 + (size_t)r * c0 + c) = make_float4(o[0], o[1], o[2], o[3]);
}

// BatchNorm statistics of z0 from the moments: stats[c] = sum z0[:,c], stats[c0 + c] = sum z0[:,c]^2 (written, not accumulated)
__global__ void narrow_stats_kernel(long rows, int k0, int c0, const double *__restrict__ mom, const float *__restrict__ w0,
                                    const float *__restrict__ b0, double *__restrict__ stats)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= c0) return;
    double w[8];
#pragma unroll
    for (int d = 0; d < 8; d++) w[d] = d < k0 ? (double)w0[(size_t)d * c0 + c] : 0.0;
    const double b = b0 ? (double)b0[c] : 0.0;
    double mw = 0.0, q = 0.0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        mw += mom[d] * w[d];
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < 8; e++) t += mom[8 + d * 8 + e] * w[e];
        q += w[d] * t;
    }
    stats[c] = mw + (double)rows * b;
    stats[c0 + c] = q + 2.0 * b * mw + (double)rows * b * b;
}

// dW0[d,c] += A[c] UG[d,c] + B[c] m[d] + C[c] (sum_e M[d,e] W0[e,c] + m[d] b0[c])
__global__ void narrow_wgrad_first_kernel(int k0, int c0, const double *__restrict__ mom, const double *__restrict__ ug,
                                          const float *__restrict__ coef, const float *__restrict__ w0, const float *__restrict__ b0,
                                          float *__restrict__ dw0)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k0 * c0) return;
    const int d = t / c0, c = t % c0;
    double uz = mom[d] * (b0 ? (double)b0[c] : 0.0);
    for (int e = 0; e < k0; e++) uz += mom[8 + d * 8 + e] * (double)w0[(size_t)e * c0 + c];
    const double g = (double)coef[c] * ug[(size_t)d * c0 + c] + (double)coef[c0 + c] * mom[d] + (double)coef[2 * c0 + c] * uz;
    dw0[(size_t)d * c0 + c] += (float)g;
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_narrow_rows(int b, int n, int m, int nsample, int c, const float *xyz, const float *new_xyz, const float *feat,
                                   const int *idx, float *u8, double *moments, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0 && nsample > 0 && c >= 0 && c <= 5, "narrow_rows expects 0 <= c <= 5 feature channels (3 + c <= 8)");
    const long rows = (long)b * m * nsample;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(rows < (1L << 31), "narrow_rows: b*m*nsample must be below 2^31");
    VN_REQUIRE(xyz && new_xyz && idx && u8 && (c == 0 || feat), "narrow_rows: null buffer");
    VN_REQUIRE((uintptr_t)u8 % 16 == 0, "narrow_rows: u8 must be 16-byte aligned");
    long gx = (rows + 256 * 16 - 1) / (256 * 16);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(narrow_rows_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), rows, n, m, nsample, c, xyz, new_xyz, feat,
                       idx, u8, moments);
    return check_launch("narrow_rows");
}

extern "C" int votenet_narrow_z0(long rows, int k0, int c0, const float *u8, const float *w0, const float *b0, float *z0, void *stream)
{
    VN_REQUIRE(rows > 0 && k0 >= 1 && k0 <= 8 && c0 > 0 && c0 % 4 == 0, "narrow_z0 expects rows > 0, 1 <= k0 <= 8, c0 %% 4 == 0");
    VN_REQUIRE(u8 && w0 && z0 && (uintptr_t)u8 % 16 == 0 && (uintptr_t)z0 % 16 == 0, "narrow_z0: null or misaligned buffer");
    const long work = rows * (c0 / 4);
    hipLaunchKernelGGL(narrow_z0_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, as_stream(stream), rows, k0, c0, u8, w0, b0, z0);
    return check_launch("narrow_z0");
}

extern "C" int votenet_narrow_stats(long rows, int k0, int c0, const double *moments, const float *w0, const float *b0, double *stats,
                                    void *stream)
{
    VN_REQUIRE(rows > 0 && k0 >= 3 && k0 <= 8 && c0 > 0, "narrow_stats expects rows > 0, 3 <= k0 <= 8, c0 > 0");
    VN_REQUIRE(moments && w0 && stats, "narrow_stats: null buffer");
    hipLaunchKernelGGL(narrow_stats_kernel, dim3((c0 + 63) / 64), dim3(64), 0, as_stream(stream), rows, k0, c0, moments, w0, b0, stats);
    return check_launch("narrow_stats");
}

extern "C" int votenet_narrow_wgrad_first(int k0, int c0, const double *moments, const double *ug, const float *coef, const float *w0,
                                          const float *b0, float *dw0, void *stream)
{
    VN_REQUIRE(k0 >= 3 && k0 <= 8 && c0 > 0, "narrow_wgrad_first expects 3 <= k0 <= 8, c0 > 0");
    VN_REQUIRE(moments && ug && coef && w0 && dw0, "narrow_wgrad_first: null buffer");
    hipLaunchKernelGGL(narrow_wgrad_first_kernel, dim3((k0 * c0 + 255) / 256), dim3(256), 0, as_stream(stream), k0, c0, moments, ug, coef,
                       w0, b0, dw0);
    return check_launch("narrow_wgrad_first");
}
