// interpolate.hip -- three_nn, inverse-distance weights, three_interpolate (+grad) for gfx950.
//
// Replaces the CPU-only ops of tf_ops/3d_interpolation/tf_interpolate.cpp:60-153 (which cost
// the reference a device->host->device round trip around fp1/fp2 every step) and the four TF
// elementwise ops of utils.py:279-282.
//   three_nn : eight lanes per unknown point over LDS tiles of the known points; the reference's strict '<'
//              insert cascade per lane, lexicographic (distance, index) merge across the octet, so equal
//              distances rank the lower index first.  Distances are SQUARED, fp32, un-fused.
//   three_interpolate : one output row per wave-slice, lanes over channels (float4 when c%4==0),
//              (p1*w1 + p2*w2) + p3*w3 un-fused, as tf_interpolate.cpp:119.
#include "common.h"

namespace votenet {

// Eight lanes per unknown point.  The known points of the scene go through LDS in tiles of TNN_TILE (float4, 16 KB); lane s
// of a point's octet walks the known indices k = s, s + 8, s + 16, ... in ascending order with the reference's strict '<'
// cascade (tf_interpolate.cpp:74-89), so inside a lane equal distances keep the lower index first.  The serial cascade
// over all k returns the three smallest entries under the order (d, then k): the octet's eight sorted triples are merged
// under exactly that lexicographic order with three xor-shuffle steps -- same distances (same fp32 expression, un-fused),
// same indices, same ties as the one-lane-per-point scan it replaces, which was a 512-iteration dependent chain on 32
// workgroups at fp2 (84 us alone); this one is 64 iterations on 256 workgroups.
constexpr int TNN_TILE = 1024;
constexpr int TNN_LPP = 8; // lanes per unknown point

struct Top3 {
    float d1, d2, d3;
    int i1, i2, i3;
    __device__ __forceinline__ void push_ascending(float d, int k) // k larger than every index pushed before: strict '<'
    {
        if (d < d1) {
            d3 = d2; i3 = i2;
            d2 = d1; i2 = i1;
            d1 = d; i1 = k;
        } else if (d < d2) {
            d3 = d2; i3 = i2;
            d2 = d; i2 = k;
        } else if (d < d3) {
            d3 = d; i3 = k;
        }
    }
    __device__ __forceinline__ void push_ordered(float d, int k) // any k: (d, k) lexicographic
    {
        const bool b1 = d < d1 || (d == d1 && k < i1);
        const bool b2 = d < d2 || (d == d2 && k < i2);
        const bool b3 = d < d3 || (d == d3 && k < i3);
        if (b1) {
            d3 = d2; i3 = i2;
            d2 = d1; i2 = i1;
            d1 = d; i1 = k;
        } else if (b2) {
            d3 = d2; i3 = i2;
            d2 = d; i2 = k;
        } else if (b3) {
            d3 = d; i3 = k;
        }
    }
};

__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2, float *__restrict__ dist,
                                                       int *__restrict__ idx)
{
    __shared__ float4 s_known[TNN_TILE];
    const int scene = blockIdx.y;
    const float *__restrict__ unk = xyz1 + (size_t)scene * n * 3;
    const float *__restrict__ known = xyz2 + (size_t)scene * m * 3;
    const int tid = threadIdx.x;
    const int sub = tid & (TNN_LPP - 1);
    const int j = blockIdx.x * (256 / TNN_LPP) + (tid >> 3);
    const int jj = j < n ? j : n - 1;
    const float x1 = unk[(size_t)jj * 3 + 0], y1 = unk[(size_t)jj * 3 + 1], z1 = unk[(size_t)jj * 3 + 2];
    // tf_interpolate.cpp:66-67: best* = 1e40 (double) -> +inf once stored as float, indices 0
    Top3 t = {INFINITY, INFINITY, INFINITY, 0, 0, 0};
    for (int base = 0; base < m; base += TNN_TILE) {
        const int cnt = (m - base) < TNN_TILE ? (m - base) : TNN_TILE;
        __syncthreads();
        for (int k = tid; k < cnt; k += 256)
            s_known[k] = make_float4(known[(size_t)(base + k) * 3 + 0], known[(size_t)(base + k) * 3 + 1],
                                     known[(size_t)(base + k) * 3 + 2], 0.0f);
        __syncthreads();
        for (int k = sub; k < cnt; k += TNN_LPP) {
            const float4 q = s_known[k];
            const float d = (q.x - x1) * (q.x - x1) + (q.y - y1) * (q.y - y1) + (q.z - z1) * (q.z - z1); // :73, un-fused
            t.push_ascending(d, base + k);
        }
    }
#pragma unroll
    for (int sh = 1; sh < TNN_LPP; sh <<= 1) { // both partners end with the same merged triple
        const float od1 = __shfl_xor(t.d1, sh), od2 = __shfl_xor(t.d2, sh), od3 = __shfl_xor(t.d3, sh);
        const int oi1 = __shfl_xor(t.i1, sh), oi2 = __shfl_xor(t.i2, sh), oi3 = __shfl_xor(t.i3, sh);
        // an untouched slot (inf, 0) must not displace anything: real entries never carry inf (strict '<' against inf)
        if (od1 < INFINITY) t.push_ordered(od1, oi1);
        if (od2 < INFINITY) t.push_ordered(od2, oi2);
        if (od3 < INFINITY) t.push_ordered(od3, oi3);
    }
    if (j < n && sub == 0) {
        float *__restrict__ od = dist + ((size_t)scene * n + j) * 3;
        int *__restrict__ oi = idx + ((size_t)scene * n + j) * 3;
        od[0] = t.d1; od[1] = t.d2; od[2] = t.d3;
        oi[0] = t.i1; oi[1] = t.i2; oi[2] = t.i3;
    }
}

__global__ void three_nn_weights_kernel(long rows, const float *__restrict__ dist, float *__restrict__ weight)
{
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        float d0 = dist[r * 3 + 0], d1 = dist[r * 3 + 1], d2 = dist[r * 3 + 2];
        d0 = d0 > 1e-10f ? d0 : 1e-10f; // utils.py:279
        d1 = d1 > 1e-10f ? d1 : 1e-10f;
        d2 = d2 > 1e-10f ? d2 : 1e-10f;
        const float r0 = 1.0f / d0, r1 = 1.0f / d1, r2 = 1.0f / d2;
        const float norm = (r0 + r1) + r2; // utils.py:280
        weight[r * 3 + 0] = r0 / norm;     // utils.py:282
        weight[r * 3 + 1] = r1 / norm;
        weight[r * 3 + 2] = r2 / norm;
    }
}

template <typename V>
__device__ __forceinline__ V lerp3(const V &a, const V &b, const V &c, float w1, float w2, float w3);
template <>
__device__ __forceinline__ float lerp3<float>(const float &a, const float &b, const float &c, float w1, float w2, float w3)
{
    return a * w1 + b * w2 + c * w3;
}
template <>
__device__ __forceinline__ float4 lerp3<float4>(const float4 &a, const float4 &b, const float4 &c, float w1, float w2,
                                                float w3)
{
    float4 r;
    r.x = a.x * w1 + b.x * w2 + c.x * w3;
    r.y = a.y * w1 + b.y * w2 + c.y * w3;
    r.z = a.z * w1 + b.z * w2 + c.z * w3;
    r.w = a.w * w1 + b.w * w2 + c.w * w3;
    return r;
}

// rows = b*n output rows, cv = channels per row in units of V
template <typename V>
__global__ void three_interpolate_kernel(long rows, int n, int m, int cv, const V *__restrict__ points,
                                         const int *__restrict__ idx, const float *__restrict__ weight, V *__restrict__ out)
{
    const long total = rows * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / cv;
        const int l = (int)(e - row * cv);
        const long s = row / n;
        const int i1 = idx[row * 3], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
        const float w1 = weight[row * 3], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
        const V *__restrict__ p = points + (size_t)s * m * cv;
        out[e] = lerp3<V>(p[(size_t)i1 * cv + l], p[(size_t)i2 * cv + l], p[(size_t)i3 * cv + l], w1, w2, w3);
    }
}

// [three_interpolate(points2) | skip]: the FP layer's concat (utils.py:283-286) written by the interpolation itself, rows of
// cv + sv vectors; the skip features (the unknown points' own, sv vectors per row) are copied next to the interpolated ones
template <typename V>
__global__ void three_interpolate_concat_kernel(long rows, int n, int m, int cv, int sv, const V *__restrict__ points,
                                                const int *__restrict__ idx, const float *__restrict__ weight,
                                                const V *__restrict__ skip, V *__restrict__ out)
{
    const int pv = cv + sv;
    const long total = rows * pv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / pv;
        const int l = (int)(e - row * pv);
        if (l >= cv) {
            out[e] = skip[(size_t)row * sv + (l - cv)];
            continue;
        }
        const long s = row / n;
        const int i1 = idx[row * 3], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
        const float w1 = weight[row * 3], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
        const V *__restrict__ p = points + (size_t)s * m * cv;
        out[e] = lerp3<V>(p[(size_t)i1 * cv + l], p[(size_t)i2 * cv + l], p[(size_t)i3 * cv + l], w1, w2, w3);
    }
}

// grad_out: rows of go_pitch floats, the c channels of interest at go_off (a slice of the concat's gradient, read in place)
__global__ void three_interpolate_grad_kernel(long rows, int n, int m, int c, const float *__restrict__ grad_out, int go_pitch,
                                              int go_off, const int *__restrict__ idx, const float *__restrict__ weight,
                                              float *__restrict__ grad_points)
{
    const long total = rows * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / c;
        const int l = (int)(e - row * c);
        const long s = row / n;
        const float g = grad_out[(size_t)row * go_pitch + go_off + l];
        float *__restrict__ gp = grad_points + (size_t)s * m * c;
        unsafeAtomicAdd(&gp[(size_t)idx[row * 3 + 0] * c + l], g * weight[row * 3 + 0]); // tf_interpolate.cpp:144-146
        unsafeAtomicAdd(&gp[(size_t)idx[row * 3 + 1] * c + l], g * weight[row * 3 + 1]);
        unsafeAtomicAdd(&gp[(size_t)idx[row * 3 + 2] * c + l], g * weight[row * 3 + 2]);
    }
}

static inline int grid_for(long total, int block)
{
    long g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                                void *stream)
{
    VN_REQUIRE(b >= 0 && n >= 0, "ThreeNN expects (b,n,3) xyz1 shape."); // tf_interpolate.cpp:163
    VN_REQUIRE(m >= 0, "ThreeNN expects (b,m,3) xyz2 shape.");           // :168
    if (b == 0 || n == 0) return VOTENET_OK;
    VN_REQUIRE(xyz1 && (xyz2 || m == 0) && dist && idx, "ThreeNN: null buffer");
    hipLaunchKernelGGL(three_nn_kernel, dim3((n + 31) / 32, b), dim3(256), 0, as_stream(stream), n, m, xyz1, xyz2, dist,
                       idx);
    return check_launch("three_nn");
}

extern "C" int votenet_three_nn_weights(int b, int n, const float *dist, float *weight, void *stream)
{
    const long rows = (long)b * n;
    VN_REQUIRE(b >= 0 && n >= 0, "three_nn_weights expects (b,n,3) dist shape");
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(dist && weight, "three_nn_weights: null buffer");
    hipLaunchKernelGGL(three_nn_weights_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, as_stream(stream), rows, dist,
                       weight);
    return check_launch("three_nn_weights");
}

extern "C" int votenet_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                                         const float *weight, float *out, void *stream)
{
    VN_REQUIRE(b >= 0 && m > 0 && c > 0, "ThreeInterpolate expects (b,m,c) points shape"); // tf_interpolate.cpp:197
    VN_REQUIRE(n >= 0, "ThreeInterpolate expects (b,n,3) idx shape");                     // :203
    const long rows = (long)b * n;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(points && idx && weight && out, "ThreeInterpolate: null buffer");
    hipStream_t st = as_stream(stream);
    if (c % 4 == 0 && ((uintptr_t)points % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
        hipLaunchKernelGGL((three_interpolate_kernel<float4>), dim3(grid_for(rows * (c / 4), 256)), dim3(256), 0, st, rows, n,
                           m, c / 4, (const float4 *)points, idx, weight, (float4 *)out);
    } else {
        hipLaunchKernelGGL((three_interpolate_kernel<float>), dim3(grid_for(rows * c, 256)), dim3(256), 0, st, rows, n, m, c,
                           points, idx, weight, out);
    }
    return check_launch("three_interpolate");
}

extern "C" int votenet_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                              const float *weight, float *grad_points, void *stream)
{
    VN_REQUIRE(b >= 0 && m > 0 && c > 0, "ThreeInterpolateGrad expects (b,m,c) points shape"); // tf_interpolate.cpp:231
    VN_REQUIRE(n >= 0, "ThreeInterpolateGrad expects (b,n,3) idx shape");                     // :237
    const long rows = (long)b * n;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(grad_out && idx && weight && grad_points, "ThreeInterpolateGrad: null buffer");
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(rows * c, 256)), dim3(256), 0, as_stream(stream), rows, n,
                       m, c, grad_out, c, 0, idx, weight, grad_points);
    return check_launch("three_interpolate_grad");
}

// ThreeInterpolateGrad reading its upstream gradient in place from a wider row-major tensor (rows of go_pitch floats, this
// op's c channels at go_off): the FP layer's input gradient is [d interpolated | d skip], utils.py:286
extern "C" int votenet_three_interpolate_grad_strided(int b, int n, int c, int m, const float *grad_out, int go_pitch, int go_off,
                                                      const int *idx, const float *weight, float *grad_points, void *stream)
{
    VN_REQUIRE(b >= 0 && m > 0 && c > 0, "ThreeInterpolateGrad expects (b,m,c) points shape");
    VN_REQUIRE(n >= 0, "ThreeInterpolateGrad expects (b,n,3) idx shape");
    VN_REQUIRE(go_off >= 0 && go_off + c <= go_pitch, "ThreeInterpolateGrad: the slice [go_off, go_off + c) exceeds the row pitch");
    const long rows = (long)b * n;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(grad_out && idx && weight && grad_points, "ThreeInterpolateGrad: null buffer");
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(rows * c, 256)), dim3(256), 0, as_stream(stream), rows, n,
                       m, c, grad_out, go_pitch, go_off, idx, weight, grad_points);
    return check_launch("three_interpolate_grad_strided");
}

// ThreeInterpolate + the concat with the unknown points' own features (utils.py:283-286): out (b,n,c + c1) = [interpolated | skip]
extern "C" int votenet_three_interpolate_concat(int b, int m, int c, int n, const float *points, const int *idx,
                                                const float *weight, const float *skip, int c1, float *out, void *stream)
{
    VN_REQUIRE(b >= 0 && m > 0 && c > 0 && c1 > 0, "ThreeInterpolate expects (b,m,c) points shape (and c1 > 0 skip channels)");
    VN_REQUIRE(n >= 0, "ThreeInterpolate expects (b,n,3) idx shape");
    const long rows = (long)b * n;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(points && idx && weight && skip && out, "ThreeInterpolate: null buffer");
    hipStream_t st = as_stream(stream);
    if (c % 4 == 0 && c1 % 4 == 0 && (((uintptr_t)points | (uintptr_t)out | (uintptr_t)skip) % 16 == 0)) {
        hipLaunchKernelGGL((three_interpolate_concat_kernel<float4>), dim3(grid_for(rows * ((c + c1) / 4), 256)), dim3(256), 0, st, rows,
                           n, m, c / 4, c1 / 4, (const float4 *)points, idx, weight, (const float4 *)skip, (float4 *)out);
    } else {
        hipLaunchKernelGGL((three_interpolate_concat_kernel<float>), dim3(grid_for(rows * (c + c1), 256)), dim3(256), 0, st, rows, n, m,
                           c, c1, points, idx, weight, skip, out);
    }
    return check_launch("three_interpolate_concat");
}
