// sampling.hip -- farthest point sampling, gather_point and its gradient for gfx950.
//
// Replaces tf_ops/sampling/tf_sampling_g.cu:105-211 of the reference.  Design (MI355X-first):
//
//  * FPS is a chain of m-1 dependent arg-max rounds per scene.  The reference keeps the
//    running distances in global memory and re-reads points past the first 3072 every round
//    (tf_sampling_g.cu:111-141).  Here a scene of up to 24576 points lives entirely in the
//    register file of ONE compute unit (xyz + running distance = 4 VGPRs per point, up to 24
//    points per lane, 16 waves): a round touches no memory except one 12-byte centre fetch.
//    The per-round arg-max is a DPP wave reduction (no LDS) + one 16-entry LDS exchange and
//    ONE workgroup barrier (the reference: 9 tree levels, 10 barriers).
//  * The reference's tie rule is part of the result (512-thread stride + left-biased tree):
//    winner = max d2, then smallest (k mod 512), then smallest k.  It is reproduced with a
//    32-bit tie key ((k & 511) << 23 | k >> 9) minimised among the lanes that hold the max.
//  * Distances are evaluated un-fused, left to right (the library is built with
//    -ffp-contract=off), matching oracle/oracle_sampling.c.
//  * Scenes that do not fit the register file fall back to a streaming kernel with the
//    running distances in the caller's temp buffer (same layout as the reference).
#include "common.h"

namespace votenet {

__device__ __forceinline__ unsigned fps_tiekey(unsigned k) { return ((k & 511u) << 23) | (k >> 9); }
__device__ __forceinline__ unsigned fps_key_to_index(unsigned key) { return ((key & 0x7FFFFFu) << 9) | (key >> 23); }

// Block-wide arg-max under the reference order.  best/key are this lane's candidates.
// Returns the winning point index, uniform over the block.  NW = waves per block.
template <int NW>
__device__ __forceinline__ unsigned fps_block_argmax(float best, unsigned key, float *s_best, unsigned *s_key, int round)
{
    const float wmax = wave_max_f32(best);
    const unsigned wkey = wave_min_u32(best == wmax ? key : 0xFFFFFFFFu);
    if (NW == 1) return fps_key_to_index(wkey);
    const int buf = (round & 1) * 16;
    const int w = wave_id_uniform();
    if (lane_id() == 0) {
        s_best[buf + w] = wmax;
        s_key[buf + w] = wkey;
    }
    __syncthreads();
    const int e = lane_id() & 15;
    float b2 = (e < NW) ? s_best[buf + e] : -2.0f;
    unsigned k2 = (e < NW) ? s_key[buf + e] : 0xFFFFFFFFu;
    const float bmax = row16_max_f32(b2);
    const unsigned bkey = row16_min_u32(b2 == bmax ? k2 : 0xFFFFFFFFu);
    return fps_key_to_index((unsigned)__builtin_amdgcn_readfirstlane((int)bkey));
}

// Slot -> point index.  The per-lane arg-max keeps the LOWEST slot on ties, so slots must be
// ordered by the reference tie key (k mod 512, then k).  With T = 64*NW threads:
//   T >= 512 : k = tid + i*T              (k mod 512 is the same for every slot of a lane)
//   T <  512 : a lane owns R = 512/T residues; slot i = a*Q + b  ->  k = b*512 + a*T + tid
//              (all points of residue a*T+tid first, in ascending k, then the next residue)
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

// Register-resident FPS: one workgroup (NW waves) per scene, P points per lane.
template <int NW, int P>
__global__ __launch_bounds__(NW * 64) void fps_reg_kernel(int n, int m, const float *__restrict__ xyz,
                                                           int *__restrict__ out)
{
    __shared__ float s_best[32];
    __shared__ unsigned s_key[32];
    const float *__restrict__ pts = xyz + (size_t)blockIdx.x * n * 3;
    int *__restrict__ o = out + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x;

    float x[P], y[P], z[P], td[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        const int k = fps_slot_to_k<NW, P>(tid, i);
        if (k < n) {
            x[i] = pts[(size_t)k * 3 + 0];
            y[i] = pts[(size_t)k * 3 + 1];
            z[i] = pts[(size_t)k * 3 + 2];
            td[i] = 1e38f; // tf_sampling_g.cu:118
        } else {
            x[i] = y[i] = z[i] = 0.0f;
            td[i] = -1.0f; // padding: min(d,-1) = -1 never beats best = -1 (strict >)
        }
    }
    int old = 0;
    if (tid == 0) o[0] = 0; // tf_sampling_g.cu:114-116
    for (int j = 1; j < m; j++) {
        // centre from the original cloud (tf_sampling_g.cu:127-129); uniform address -> scalar load
        const float cx = pts[(size_t)old * 3 + 0];
        const float cy = pts[(size_t)old * 3 + 1];
        const float cz = pts[(size_t)old * 3 + 2];
        float best = -1.0f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            const float dx = x[i] - cx, dy = y[i] - cy, dz = z[i] - cz;
            const float d = dx * dx + dy * dy + dz * dz; // tf_sampling_g.cu:142, un-fused
            const float d2 = (d < td[i]) ? d : td[i];    // :143
            td[i] = d2;
            if (d2 > best) { // :146 strict: the lowest slot (smallest tie key of this lane) wins ties
                best = d2;
                bi = i;
            }
        }
        const unsigned k = (unsigned)fps_slot_to_k<NW, P>(tid, bi);
        old = (int)fps_block_argmax<NW>(best, fps_tiekey(k), s_best, s_key, j);
        if (tid == 0) o[j] = old;
    }
}

// Streaming fallback for scenes larger than the register file: running distances in `temp`
// (one row of n floats per resident block, as tf_sampling.cpp:115), points re-read from
// L2/HBM each round.  Same selection rule.
template <int NW>
__global__ __launch_bounds__(NW * 64) void fps_stream_kernel(int b, int n, int m, const float *__restrict__ xyz,
                                                              float *__restrict__ temp, int *__restrict__ out)
{
    constexpr int T = NW * 64;
    __shared__ float s_best[32];
    __shared__ unsigned s_key[32];
    const int tid = threadIdx.x;
    float *__restrict__ td = temp + (size_t)blockIdx.x * n;
    for (int scene = blockIdx.x; scene < b; scene += gridDim.x) {
        const float *__restrict__ pts = xyz + (size_t)scene * n * 3;
        int *__restrict__ o = out + (size_t)scene * m;
        for (int k = tid; k < n; k += T) td[k] = 1e38f;
        int old = 0;
        if (tid == 0) o[0] = 0;
        for (int j = 1; j < m; j++) {
            const float cx = pts[(size_t)old * 3 + 0];
            const float cy = pts[(size_t)old * 3 + 1];
            const float cz = pts[(size_t)old * 3 + 2];
            float best = -1.0f;
            unsigned bk = 0;
            for (int k = tid; k < n; k += T) {
                const float dx = pts[(size_t)k * 3 + 0] - cx, dy = pts[(size_t)k * 3 + 1] - cy,
                            dz = pts[(size_t)k * 3 + 2] - cz;
                const float d = dx * dx + dy * dy + dz * dz;
                const float t0 = td[k];
                const float d2 = (d < t0) ? d : t0;
                if (d2 != t0) td[k] = d2;
                if (d2 > best) {
                    best = d2;
                    bk = (unsigned)k;
                }
            }
            old = (int)fps_block_argmax<NW>(best, fps_tiekey(bk), s_best, s_key, j);
            if (tid == 0) o[j] = old;
        }
        __syncthreads(); // td[] is re-initialised by other lanes for the next scene
    }
}

// gather: out[s,j,:] = inp[s,idx[s,j],:]   (tf_sampling_g.cu:172-181)
__global__ void gather_point_kernel(int n, int m, long total, const float *__restrict__ inp,
                                    const int *__restrict__ idx, float *__restrict__ out)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / 3;
        const int ch = (int)(e - row * 3);
        const long s = row / m;
        const int a = idx[row];
        out[e] = inp[((size_t)s * n + a) * 3 + ch];
    }
}

// scatter-add: inp_g[s,idx[s,j],:] += out_g[s,j,:]   (tf_sampling_g.cu:183-192)
__global__ void gather_point_grad_kernel(int n, int m, long total, const float *__restrict__ out_g,
                                         const int *__restrict__ idx, float *__restrict__ inp_g)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / 3;
        const int ch = (int)(e - row * 3);
        const long s = row / m;
        const int a = idx[row];
        unsafeAtomicAdd(&inp_g[((size_t)s * n + a) * 3 + ch], out_g[e]);
    }
}

static const int kFpsRegMax = 1024 * 24;

} // namespace votenet

using namespace votenet;

extern "C" size_t votenet_fps_temp_floats(int b, int n)
{
    if (n <= kFpsRegMax) return 0;
    return (size_t)(b < 32 ? b : 32) * (size_t)n;
}

#define FPS_LAUNCH(NW, P) \
    hipLaunchKernelGGL((fps_reg_kernel<NW, P>), dim3(b), dim3(NW * 64), 0, st, n, m, inp, out)

extern "C" int votenet_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream)
{
    VN_REQUIRE(m > 0, "FarthestPointSample expects positive npoint");                  // tf_sampling.cpp:99
    VN_REQUIRE(b >= 0 && n > 0, "FarthestPointSample expects (batch_size,num_points,3) inp shape"); // :105
    VN_REQUIRE(inp && out, "FarthestPointSample: null buffer");
    if (b == 0) return VOTENET_OK;
    hipStream_t st = as_stream(stream);
    if (n <= 64 * 8) {
        FPS_LAUNCH(1, 8);
    } else if (n <= 256 * 4) {
        FPS_LAUNCH(4, 4);
    } else if (n <= 256 * 8) {
        FPS_LAUNCH(4, 8);
    } else if (n <= 256 * 16) {
        FPS_LAUNCH(4, 16);
    } else if (n <= 1024 * 8) {
        FPS_LAUNCH(16, 8);
    } else if (n <= 1024 * 12) {
        FPS_LAUNCH(16, 12);
    } else if (n <= 1024 * 16) {
        FPS_LAUNCH(16, 16);
    } else if (n <= 1024 * 20) {
        FPS_LAUNCH(16, 20);
    } else if (n <= 1024 * 24) {
        FPS_LAUNCH(16, 24);
    } else {
        VN_REQUIRE(temp != nullptr, "FarthestPointSample: temp scratch of %zu floats required for n=%d",
                   votenet_fps_temp_floats(b, n), n);
        const int grid = b < 32 ? b : 32; // tf_sampling_g.cu:204
        hipLaunchKernelGGL((fps_stream_kernel<16>), dim3(grid), dim3(1024), 0, st, b, n, m, inp, temp, out);
    }
    return check_launch("farthest_point_sample");
}

static inline int grid_for(long total, int block)
{
    long g = (total + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int votenet_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0, "GatherPoint expects (batch_size,num_points,3) inp shape"); // tf_sampling.cpp:131
    const long total = (long)b * m * 3;
    if (total == 0) return VOTENET_OK;
    VN_REQUIRE(inp && idx && out, "GatherPoint: null buffer");
    hipLaunchKernelGGL(gather_point_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), n, m, total, inp,
                       idx, out);
    return check_launch("gather_point");
}

extern "C" int votenet_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                         void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0, "GatherPointGradGpuOp expects (batch_size,num_points,3) inp"); // :156
    const long total = (long)b * m * 3;
    if (total == 0) return VOTENET_OK;
    VN_REQUIRE(out_g && idx && inp_g, "GatherPointGrad: null buffer");
    hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), n, m, total,
                       out_g, idx, inp_g);
    return check_launch("gather_point_grad");
}

// ---- the reference's own launcher names, C++ linkage, exact signatures (tf_sampling.cpp:94,125,150)
// so that tf_sampling.cpp links against this library unchanged.  Null stream, as the reference.
void farthestpointsamplingLauncher(int b, int n, int m, const float *inp, float *temp, int *out)
{
    votenet_farthest_point_sample(b, n, m, inp, temp, out, nullptr);
}
void gatherpointLauncher(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    votenet_gather_point(b, n, m, inp, idx, out, nullptr);
}
void scatteraddpointLauncher(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    votenet_gather_point_grad(b, n, m, out_g, idx, inp_g, nullptr);
}
