// group_linear.hip -- first layer of a set-abstraction MLP with the linear map applied BEFORE the grouping (gfx950).
//
// The reference convolves the grouped tensor [xyz[idx]-new_xyz | feat[idx]] (utils.py:50-57,125-127).  A gather commutes
// with a per-point linear map:  feat[idx] . W[3:]  ==  (feat . W[3:])[idx],  so the feature block needs ONE GEMM over the
// n points of the scene (P = feat . W[3:], votenet_mlp_linear on b*n rows) instead of b*m*nsample grouped rows -- 32x
// fewer multiply-adds at sa2 -- and the layer output is assembled here:
//      z[b,j,k,:] = P[b, idx[b,j,k], :] + dxyz . W[0:3] + bias ,   dxyz = xyz[b,idx] - new_xyz[b,j]
// This kernel is the HBM-bound part: one 16-byte-vectorised gather of P rows (the P table of a scene is L2 resident),
// three FMAs per output, one streaming write of z, and the per-channel BatchNorm statistics (sum z, sum z^2) of the layer.
// Rounding differs from the k-ordered chain of the fused GEMM by the summation order only (xyz terms added last).
#include "mlp_types.h"

namespace votenet {

// thread = (row lane, channel quad): QC = cout/4 quads, RP = 256/QC rows per pass; 4 passes in flight per loop trip.
__global__ __launch_bounds__(256) void group_linear_kernel(long rows, int n, int groups_per_scene, int nsample, int cout,
                                                           const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const int *__restrict__ idx, const float *__restrict__ P,
                                                           const float *__restrict__ w_xyz, const float *__restrict__ bias,
                                                           float *__restrict__ z, double *__restrict__ stats)
{
    __shared__ float red[2][256][4];
    const int qc = cout >> 2, rpp = 256 / qc;
    const int q = threadIdx.x % qc, rl = threadIdx.x / qc;
    const float4 w0 = *reinterpret_cast<const float4 *>(w_xyz + 4 * q);
    const float4 w1 = *reinterpret_cast<const float4 *>(w_xyz + cout + 4 * q);
    const float4 w2 = *reinterpret_cast<const float4 *>(w_xyz + 2 * cout + 4 * q);
    const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const unsigned rows_per_scene = (unsigned)groups_per_scene * (unsigned)nsample; // rows < 2^31 (launcher): 32-bit divisions
    const long stride = (long)gridDim.x * rpp;
    constexpr int U = 4;
    for (long r0 = (long)blockIdx.x * rpp + rl; r0 < rows; r0 += U * stride) {
        int id[U];
        long sc[U];
        float4 pv[U];
        float dx[U], dy[U], dz[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long r = r0 + u * stride;
            const bool ok = r < rows;
            id[u] = ok ? idx[r] : 0;
            sc[u] = ok ? (long)((unsigned)r / rows_per_scene) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long r = r0 + u * stride;
            const bool ok = r < rows;
            const size_t prow = (size_t)sc[u] * n + id[u];
            pv[u] = *reinterpret_cast<const float4 *>(P + prow * cout + 4 * q);
            const long g = ok ? (long)((unsigned)r / (unsigned)nsample) : 0;
            dx[u] = xyz[prow * 3 + 0] - new_xyz[(size_t)g * 3 + 0]; // utils.py:55
            dy[u] = xyz[prow * 3 + 1] - new_xyz[(size_t)g * 3 + 1];
            dz[u] = xyz[prow * 3 + 2] - new_xyz[(size_t)g * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long r = r0 + u * stride;
            if (r < rows) {
                float4 v;
                v.x = ((pv[u].x + dx[u] * w0.x) + dy[u] * w1.x) + dz[u] * w2.x + bv.x;
                v.y = ((pv[u].y + dx[u] * w0.y) + dy[u] * w1.y) + dz[u] * w2.y + bv.y;
                v.z = ((pv[u].z + dx[u] * w0.z) + dy[u] * w1.z) + dz[u] * w2.z + bv.z;
                v.w = ((pv[u].w + dx[u] * w0.w) + dy[u] * w1.w) + dz[u] * w2.w + bv.w;
                *reinterpret_cast<float4 *>(z + (size_t)r * cout + 4 * q) = v;
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
            }
        }
    }
    if (stats) {
        *reinterpret_cast<float4 *>(&red[0][threadIdx.x][0]) = s1;
        *reinterpret_cast<float4 *>(&red[1][threadIdx.x][0]) = s2;
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * cout; t += 256) {
            const int which = t / cout, ch = t % cout;
            float v = 0.0f;
            for (int i = 0; i < rpp; i++) v += red[which][i * qc + (ch >> 2)][ch & 3];
            unsafeAtomicAdd(&stats[which * cout + ch], (double)v);
        }
    }
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_group_linear(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                    const int *idx, const float *P, const float *w_xyz, const float *bias, float *z,
                                    double *stats, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0 && nsample > 0 && cout > 0, "group_linear: bad shape");
    VN_REQUIRE(cout % 4 == 0 && cout <= 1024 && 256 % (cout / 4) == 0, "group_linear expects cout in {4,8,...,1024}, a power of two");
    const long rows = (long)b * m * nsample;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(rows < (1L << 31), "group_linear: b*m*nsample must be below 2^31");
    VN_REQUIRE(xyz && new_xyz && idx && P && w_xyz && z, "group_linear: null buffer");
    VN_REQUIRE((uintptr_t)P % 16 == 0 && (uintptr_t)w_xyz % 16 == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)bias % 16 == 0,
               "group_linear: P, w_xyz, bias and z must be 16-byte aligned");
    const long rpp = 256 / (cout / 4);
    long gx = (rows + 16 * rpp - 1) / (16 * rpp); // >= 4 loop trips of 4 passes per workgroup
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(group_linear_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), rows, n, m, nsample, cout, xyz,
                       new_xyz, idx, P, w_xyz, bias, z, stats);
    return check_launch("group_linear");
}

// ---------------------------------------------------------------- backward of the assembled first layer
namespace votenet {

// One pass over (z, da) of a BatchNorm'ed first SA layer z = P[idx] + dxyz W[0:3]:
//   dz   = A*g' + B + C*z,  g' = da masked by [z*S+H > 0]                       (BatchNorm + ReLU backward, coef = [A|B|C|S|H])
//   S_pt[scene, idx[row], ch] += dz[row, ch]                                   (GroupPointGrad at the layer output width)
//   dW[d, ch]                 += sum_rows dxyz[row, d] * dz[row, ch], d < 3     (the xyz rows of the weight gradient)
//   dz_out[row, ch]            = dz                                             (optional: the proposal layer also needs dz W[0:3]^T)
// so that neither dz nor the per-row input gradient is ever written.  Thread = (group, channel): it walks the group's
// nsample rows; the group's idx and dxyz rows are staged in LDS once per group.  A ball with fewer than nsample
// neighbours is padded with its first hit (tf_grouping_g.cu:26-29): rows k >= pts_cnt all target idx[g,0] and are summed
// in a register, one atomic instead of nsample - pts_cnt.  Workgroups are persistent over groups so that the 3*cout
// atomics of dW amortise.
template <int GPB /* groups per workgroup pass = 256 / cout */>
__global__ __launch_bounds__(256) void group_linear_bwd_kernel(long groups, int n, int groups_per_scene, int nsample, int cout,
                                                               const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                               const int *__restrict__ idx, const int *__restrict__ pts_cnt,
                                                               const float *__restrict__ z, const float *__restrict__ da,
                                                               const float *__restrict__ coef, int relu, float *__restrict__ spt,
                                                               float *__restrict__ dw_xyz, float *__restrict__ dz_out,
                                                               const float *__restrict__ ptab, const float *__restrict__ wx)
{
    constexpr int KMAX = 128; // nsample <= 128 (launcher)
    __shared__ int s_idx[GPB][KMAX];
    __shared__ float s_dx[GPB][KMAX][3];
    __shared__ float red[256][3];
    const int tid = threadIdx.x;
    const int ch = tid % cout, gl = tid / cout; // channel, group lane inside the pass
    const float kA = coef[ch], kB = coef[cout + ch], kC = coef[2 * cout + ch], kS = coef[3 * cout + ch], kH = coef[4 * cout + ch];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    // z == NULL: the layer output was never stored (assemble.hip): rebuilt per element from the per-point table and the dxyz staged below
    const float wx0 = ptab ? wx[ch] : 0.f, wx1 = ptab ? wx[cout + ch] : 0.f, wx2 = ptab ? wx[2 * cout + ch] : 0.f;
    for (long g0 = (long)blockIdx.x * GPB; g0 < groups; g0 += (long)gridDim.x * GPB) {
        __syncthreads(); // previous pass's LDS fully consumed
        // stage idx and dxyz of the pass's groups: GPB * nsample rows over 256 threads
        for (int t = tid; t < GPB * nsample; t += 256) {
            const int q = t / nsample, k = t - q * nsample;
            const long g = g0 + q;
            if (g < groups) {
                const int id = idx[(size_t)g * nsample + k];
                const size_t p = ((size_t)(g / groups_per_scene) * n + id) * 3;
                s_idx[q][k] = id;
                s_dx[q][k][0] = xyz[p + 0] - new_xyz[(size_t)g * 3 + 0]; // utils.py:55
                s_dx[q][k][1] = xyz[p + 1] - new_xyz[(size_t)g * 3 + 1];
                s_dx[q][k][2] = xyz[p + 2] - new_xyz[(size_t)g * 3 + 2];
            }
        }
        __syncthreads();
        const long g = g0 + gl;
        if (g < groups) {
            const size_t row0 = (size_t)g * nsample;
            float *__restrict__ tab = spt + (size_t)(g / groups_per_scene) * n * cout + ch;
            int cnt = pts_cnt ? pts_cnt[g] : nsample;
            if (cnt < 1) cnt = 1;
            float pad = 0.0f;
            for (int k0 = 0; k0 < nsample; k0 += 8) {
                float zz[8], gg[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + u < nsample ? k0 + u : nsample - 1;
                    if (ptab)
                        zz[u] = ptab[((size_t)(g / groups_per_scene) * n + s_idx[gl][k]) * cout + ch];
                    else
                        zz[u] = z[(row0 + k) * cout + ch];
                    gg[u] = da[(row0 + k) * cout + ch];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + u;
                    if (k < nsample) {
                        if (ptab) zz[u] = assembled_z(zz[u], make_float4(s_dx[gl][k][0], s_dx[gl][k][1], s_dx[gl][k][2], 0.f), wx0, wx1, wx2);
                        float gq = gg[u];
                        if (relu && !(zz[u] * kS + kH > 0.0f)) gq = 0.0f;
                        const float d = kA * gq + kB + kC * zz[u];
                        if (dz_out) dz_out[(row0 + k) * cout + ch] = d;
                        a0 += s_dx[gl][k][0] * d;
                        a1 += s_dx[gl][k][1] * d;
                        a2 += s_dx[gl][k][2] * d;
                        if (k > 0 && k < cnt)
                            unsafeAtomicAdd(&tab[(size_t)s_idx[gl][k] * cout], d);
                        else
                            pad += d; // row 0 and the padding rows share idx[g,0]
                    }
                }
            }
            unsafeAtomicAdd(&tab[(size_t)s_idx[gl][0] * cout], pad);
        }
    }
    red[tid][0] = a0;
    red[tid][1] = a1;
    red[tid][2] = a2;
    __syncthreads();
    if (tid < cout) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float t = 0.0f;
            for (int q = 0; q < GPB; q++) t += red[q * cout + tid][d];
            unsafeAtomicAdd(&dw_xyz[(size_t)d * cout + tid], t);
        }
    }
}

} // namespace votenet

static int group_linear_backward_impl(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                      const int *idx, const int *pts_cnt, const float *z, const float *da, const float *coef,
                                      int relu, float *s_points, float *dw_xyz, float *dz_out, const float *ptab, const float *wx,
                                      void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0 && nsample > 0 && cout > 0, "group_linear_backward: bad shape");
    VN_REQUIRE(nsample <= 128, "group_linear_backward expects nsample <= 128");
    VN_REQUIRE(cout == 32 || cout == 64 || cout == 128 || cout == 256, "group_linear_backward expects cout in {32, 64, 128, 256}");
    const long groups = (long)b * m;
    if (groups == 0) return VOTENET_OK;
    VN_REQUIRE(xyz && new_xyz && idx && (z || (ptab && wx)) && da && coef && s_points && dw_xyz, "group_linear_backward: null buffer");
    hipStream_t st = as_stream(stream);
    const int gpb = 256 / cout;
    long gx = (groups + gpb - 1) / gpb;
    if (gx > 2048) gx = 2048;
#define GLB_LAUNCH(G)                                                                                                             \
    hipLaunchKernelGGL(group_linear_bwd_kernel<G>, dim3((unsigned)gx), dim3(256), 0, st, groups, n, m, nsample, cout, xyz, new_xyz, \
                       idx, pts_cnt, z, da, coef, relu, s_points, dw_xyz, dz_out, ptab, wx)
    if (gpb == 8) GLB_LAUNCH(8);
    else if (gpb == 4) GLB_LAUNCH(4);
    else if (gpb == 2) GLB_LAUNCH(2);
    else GLB_LAUNCH(1);
#undef GLB_LAUNCH
    return check_launch("group_linear_backward");
}

extern "C" int votenet_group_linear_backward(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                             const int *idx, const int *pts_cnt, const float *z, const float *da, const float *coef,
                                             int relu, float *s_points, float *dw_xyz, float *dz_out, void *stream)
{
    return group_linear_backward_impl(b, n, m, nsample, cout, xyz, new_xyz, idx, pts_cnt, z, da, coef, relu, s_points, dw_xyz, dz_out,
                                      nullptr, nullptr, stream);
}

// The same pass for a layer whose output was never stored (assemble.hip): z is rebuilt from P (b*n x cout, bias included) and wx.
extern "C" int votenet_group_linear_backward_assembled(int b, int n, int m, int nsample, int cout, const float *xyz,
                                                       const float *new_xyz, const int *idx, const int *pts_cnt, const float *P,
                                                       const float *wx, const float *da, const float *coef, int relu,
                                                       float *s_points, float *dw_xyz, float *dz_out, void *stream)
{
    VN_REQUIRE(P && wx, "group_linear_backward_assembled: null buffer");
    return group_linear_backward_impl(b, n, m, nsample, cout, xyz, new_xyz, idx, pts_cnt, nullptr, da, coef, relu, s_points, dw_xyz,
                                      dz_out, P, wx, stream);
}
