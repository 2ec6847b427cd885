// assemble.hip -- the first layer of a set-abstraction MLP assembled INSIDE the kernels that consume it (gfx950).
//
// With the linear map applied before the grouping (group_linear.hip) the first layer's output is
//     z0[r,:] = P[prow(r),:] + dxyz(r) . Wx ,      P = feat . W[3:] + b  (one GEMM over the points),  Wx = W[0:3],
// a gather of a per-point row plus three multiply-adds per channel.  votenet_group_linear writes that tensor (rows x c0) once
// and the next layer's GEMM, its weight-gradient GEMM and the BatchNorm backward of layer 0 each read it back.  Here it is never
// written: every consumer rebuilds its elements from
//     geo[r] = (dx, dy, dz, bits(prow))        16 bytes per grouped row (coordinates only: computed with the geometry)
// and the L2-resident table P with assembled_z (mlp_types.h: one fixed fma chain, bit-identical in every kernel).  What the
// BatchNorm of layer 0 needs -- sum z0, sum z0^2 over the rows -- follows from sums over the POINTS:
//     cntv[p] = (number of rows that gather p, sum of their dxyz)        (votenet_assemble_rows: 64-bit INTEGER atomics, the dxyz in
//               fixed point 2^-32 -- integer addition is associative, so the sums and everything derived from them are bit-reproducible)
//     sum_r z0[r,c]   = sum_p cnt_p P[p,c] + (sum_r dxyz) . Wx[:,c]
//     sum_r z0[r,c]^2 = sum_p (cnt_p P[p,c]^2 + 2 P[p,c] V_p . Wx[:,c]) + Wx[:,c]^T (sum_r dxyz dxyz^T) Wx[:,c]
// (votenet_assemble_stats: one pass over P, b*n x c0, instead of one over z0, b*m*nsample x c0).
#include "mlp_types.h"

namespace votenet {

__device__ __forceinline__ double asm_shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// dxyz in fixed point: |dxyz| is a ball radius (metres), 2^-32 resolves 2e-10 -- statistics only, the layer itself uses the fp32 value
__device__ __forceinline__ long long asm_fixed(float v) { return (long long)((double)v * 4294967296.0); }

// thread = grouped row.  moments (9 doubles, accumulated): sum dx, dy, dz, then xx, xy, xz, yy, yz, zz.
__global__ __launch_bounds__(256) void assemble_rows_kernel(long rows, int n, int groups_per_scene, int nsample,
                                                            const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                            const int *__restrict__ idx, const int *__restrict__ pts_cnt,
                                                            float4 *__restrict__ geo, long long *__restrict__ cntv,
                                                            double *__restrict__ moments)
{
    __shared__ double red[4][9];
    const unsigned rows_per_scene = (unsigned)groups_per_scene * (unsigned)nsample;
    double acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = 0.0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const int id = idx[r];
        const unsigned prow = ((unsigned)r / rows_per_scene) * (unsigned)n + (unsigned)id;
        const size_t g = (size_t)((unsigned)r / (unsigned)nsample);
        const float dx = xyz[(size_t)prow * 3 + 0] - new_xyz[g * 3 + 0]; // utils.py:55
        const float dy = xyz[(size_t)prow * 3 + 1] - new_xyz[g * 3 + 1];
        const float dz = xyz[(size_t)prow * 3 + 2] - new_xyz[g * 3 + 2];
        geo[r] = make_float4(dx, dy, dz, __uint_as_float(prow));
        if (cntv) {
            // A ball with fewer than nsample neighbours is padded with its first hit (tf_grouping_g.cu:26-29): slots k >= pts_cnt
            // repeat slot 0 -- same point, same dxyz.  Slot 0 adds them all at once; one atomic per distinct (group, point) instead
            // of up to nsample on one address.
            const int k = (int)((unsigned)r % (unsigned)nsample);
            int cnt = pts_cnt ? pts_cnt[g] : nsample;
            if (cnt < 1) cnt = 1;
            if (k < cnt) {
                const long long mult = (k == 0) ? (long long)(nsample - cnt + 1) : 1LL;
                unsigned long long *cv = reinterpret_cast<unsigned long long *>(cntv + (size_t)prow * 4);
                atomicAdd(cv + 0, (unsigned long long)mult);
                atomicAdd(cv + 1, (unsigned long long)(mult * asm_fixed(dx)));
                atomicAdd(cv + 2, (unsigned long long)(mult * asm_fixed(dy)));
                atomicAdd(cv + 3, (unsigned long long)(mult * asm_fixed(dz)));
            }
        }
        const double x = dx, y = dy, z = dz;
        acc[0] += x; acc[1] += y; acc[2] += z;
        acc[3] += x * x; acc[4] += x * y; acc[5] += x * z; acc[6] += y * y; acc[7] += y * z; acc[8] += z * z;
    }
    if (!moments) return;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        double v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += asm_shfl_xor_f64(v, m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
        unsafeAtomicAdd(&moments[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// BatchNorm statistics of the never-stored z0 from the per-point sums: block = 64 channels x 4 point lanes, grid.x strides the points
template <int U>
__global__ __launch_bounds__(256) void assemble_stats_kernel(long npts, int c0, const float *__restrict__ P, const long long *__restrict__ cntv,
                                                             const float *__restrict__ wx, const double *__restrict__ mom,
                                                             double *__restrict__ stats)
{
    __shared__ double sh1[4][64], sh2[4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < c0) {
        const double w0 = wx[c], w1 = wx[c0 + c], w2 = wx[2 * c0 + c];
        // four points' loads in flight per trip, few workgroups: the pass is a latency chain that ends in fp64 atomics on 2 c0 addresses
        // (256 workgroups a column block made it 15 us for 2 MB: the atomics of one address serialise)
        const long stride = (long)gridDim.x * 4;
        for (long p0 = (long)blockIdx.x * 4 + py; p0 < npts; p0 += U * stride) {
            long long c4[U][4];
            float v4[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long p = p0 + u * stride < npts ? p0 + u * stride : p0; // (clamped: counted once below)
                const long long *cv = cntv + (size_t)p * 4;
                c4[u][0] = cv[0];
                c4[u][1] = cv[1];
                c4[u][2] = cv[2];
                c4[u][3] = cv[3];
                v4[u] = P[(size_t)p * c0 + c];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (p0 + u * stride >= npts || c4[u][0] == 0) continue; // past the end / a point no ball contains
                const double v = v4[u], cn = (double)c4[u][0];
                const double vw = ((double)c4[u][1] * w0 + (double)c4[u][2] * w1 + (double)c4[u][3] * w2) * (1.0 / 4294967296.0);
                s1 += cn * v;
                s2 += cn * v * v + 2.0 * v * vw;
            }
        }
        if (blockIdx.x == 0 && py == 0) { // the coordinate-only terms, once
            s1 += mom[0] * w0 + mom[1] * w1 + mom[2] * w2;
            s2 += w0 * (mom[3] * w0 + mom[4] * w1 + mom[5] * w2) + w1 * (mom[4] * w0 + mom[6] * w1 + mom[7] * w2) +
                  w2 * (mom[5] * w0 + mom[7] * w1 + mom[8] * w2);
        }
    }
    sh1[py][cx] = s1;
    sh2[py][cx] = s2;
    __syncthreads();
    if (py == 0 && c < c0) {
        unsafeAtomicAdd(&stats[c], (sh1[0][cx] + sh1[1][cx]) + (sh1[2][cx] + sh1[3][cx]));
        unsafeAtomicAdd(&stats[c0 + c], (sh2[0][cx] + sh2[1][cx]) + (sh2[2][cx] + sh2[3][cx]));
    }
}

// z0 itself, for a caller that does want the layer output (tests): thread = (row, channel quad)
__global__ __launch_bounds__(256) void assemble_z0_kernel(long rows, int c0, const float4 *__restrict__ geo, const float *__restrict__ P,
                                                          const float *__restrict__ wx, float *__restrict__ z0)
{
    const int qc = c0 >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * qc) return;
    const long r = t / qc;
    const int c = (int)(t % qc) * 4;
    const float4 g = geo[r];
    const float4 p = *reinterpret_cast<const float4 *>(P + (size_t)__float_as_uint(g.w) * c0 + c);
    const float4 w0 = *reinterpret_cast<const float4 *>(wx + c), w1 = *reinterpret_cast<const float4 *>(wx + c0 + c);
    const float4 w2 = *reinterpret_cast<const float4 *>(wx + 2 * c0 + c);
    *reinterpret_cast<float4 *>(z0 + (size_t)r * c0 + c) =
        make_float4(assembled_z(p.x, g, w0.x, w1.x, w2.x), assembled_z(p.y, g, w0.y, w1.y, w2.y), assembled_z(p.z, g, w0.z, w1.z, w2.z),
                    assembled_z(p.w, g, w0.w, w1.w, w2.w));
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_assemble_rows(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const int *idx,
                                     const int *pts_cnt, float *geo, long long *cntv, double *moments, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0 && nsample > 0, "assemble_rows: bad shape");
    const long rows = (long)b * m * nsample;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(rows < (1L << 31) && (long)b * n < (1L << 31), "assemble_rows: b*m*nsample and b*n must be below 2^31");
    VN_REQUIRE(xyz && new_xyz && idx && geo, "assemble_rows: null buffer");
    VN_REQUIRE((uintptr_t)geo % 16 == 0 && (!cntv || (uintptr_t)cntv % 16 == 0), "assemble_rows: geo / cntv must be 16-byte aligned");
    long gx = (rows + 256 * 8 - 1) / (256 * 8);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(assemble_rows_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), rows, n, m, nsample, xyz, new_xyz, idx,
                       pts_cnt, reinterpret_cast<float4 *>(geo), cntv, moments);
    return check_launch("assemble_rows");
}

static int g_asm_stats_cap = 128, g_asm_stats_u = 8; // (8 points in flight: 11.9 / 8.2 / 7.2 -> 10.6 / 7.5 / 6.5 us; 256 workgroups: no better)
extern "C" void votenet_debug_assemble_stats(int cap, int u) // tuning hook: workgroups per column block, points in flight per thread and trip
{
    VN_DEBUG_GATE();
    g_asm_stats_cap = cap > 0 ? cap : 128;
    g_asm_stats_u = u == 4 ? 4 : 8;
}
extern "C" int votenet_assemble_stats(long npts, int c0, const float *P, const long long *cntv, const float *wx, const double *moments,
                                      double *stats, void *stream)
{
    VN_REQUIRE(npts > 0 && c0 > 0, "assemble_stats expects npts > 0, c0 > 0");
    VN_REQUIRE(P && cntv && wx && moments && stats, "assemble_stats: null buffer");
    const int ny = (c0 + 63) / 64;
    long gx = (npts + 4 * 16 - 1) / (4 * 16);
    if (gx > g_asm_stats_cap) gx = g_asm_stats_cap;
    if (g_asm_stats_u == 8)
        hipLaunchKernelGGL(assemble_stats_kernel<8>, dim3((unsigned)gx, ny), dim3(256), 0, as_stream(stream), npts, c0, P, cntv, wx, moments, stats);
    else
        hipLaunchKernelGGL(assemble_stats_kernel<4>, dim3((unsigned)gx, ny), dim3(256), 0, as_stream(stream), npts, c0, P, cntv, wx, moments, stats);
    return check_launch("assemble_stats");
}

extern "C" int votenet_assemble_z0(long rows, int c0, const float *geo, const float *P, const float *wx, float *z0, void *stream)
{
    VN_REQUIRE(rows > 0 && c0 > 0 && c0 % 4 == 0, "assemble_z0 expects rows > 0, c0 %% 4 == 0");
    VN_REQUIRE(geo && P && wx && z0 && (uintptr_t)P % 16 == 0 && (uintptr_t)wx % 16 == 0 && (uintptr_t)z0 % 16 == 0,
               "assemble_z0: null or misaligned buffer");
    const long work = rows * (c0 / 4);
    hipLaunchKernelGGL(assemble_z0_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, as_stream(stream), rows, c0,
                       reinterpret_cast<const float4 *>(geo), P, wx, z0);
    return check_launch("assemble_z0");
}
