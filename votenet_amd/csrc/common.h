// common.h -- shared host/device helpers of libvotenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
// the library is built -fvisibility=hidden: only what the two public headers declare (and the reference's launcher names, VN_EXPORT) is exported
#pragma GCC visibility push(default)
#include "../../include/votenet_hip.h"
#include "../../include/votenet_hip_debug.h"
#pragma GCC visibility pop
#define VN_EXPORT __attribute__((visibility("default")))

namespace votenet {

// ---- error plumbing (thread-local text behind votenet_last_error()) ----
int set_error(int code, const char *fmt, ...);
int check_launch(const char *what);
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// The votenet_debug_* / votenet_fps_debug_* switches (include/votenet_hip_debug.h) are process-global measurement / tuning hooks.  They do
// nothing until votenet_debug_enable(1) has been called (or VOTENET_DEBUG=1 was in the environment when the first one was called): a host that
// never opts in gets a library whose launches depend on their arguments only.
bool debug_gate(const char *name);
#define VN_DEBUG_GATE()                                      \
    do {                                                     \
        if (!::votenet::debug_gate(__func__)) return;        \
    } while (0)

#define VN_REQUIRE(cond, ...)                                                   \
    do {                                                                        \
        if (!(cond)) return ::votenet::set_error(VOTENET_E_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

// ---- BatchNorm from raw sums in the consumer's prologue (struct votenet_bn_raw) ----
struct BnRaw {
    const double *stats;
    const float *gamma, *beta;
    long rows;
    float eps;
    float *out;
};
inline BnRaw to_raw(const votenet_bn_raw *b)
{
    BnRaw r = {};
    if (b) r = BnRaw{b->stats, b->gamma, b->beta, b->rows, b->eps, b->out};
    return r;
}
// scale / shift of channel o (bn_finalize_kernel's arithmetic); `writer`: this thread also records the four vectors
__device__ __forceinline__ void bn_raw_channel(const BnRaw &r, int c, int o, bool writer, float &sc, float &sh)
{
    const double mu = r.stats[o] / (double)r.rows;
    double v = r.stats[c + o] / (double)r.rows - mu * mu;
    if (v < 0) v = 0;
    const float muf = (float)mu, vf = (float)v;
    sc = r.gamma[o] / sqrtf(vf + r.eps);
    sh = r.beta[o] - muf * sc;
    if (writer && r.out) {
        r.out[o] = sc;
        r.out[c + o] = sh;
        r.out[2 * c + o] = muf;
        r.out[3 * c + o] = vf;
    }
}

// ---- BatchNorm-backward coefficients in the tail of the kernel that reduced their sums (struct votenet_coef_tail) ----
struct CoefTail {
    unsigned *ticket;
    long rows;
    const float *gamma;
    float *coef, *dgamma, *dbeta;
};
inline CoefTail to_tail(const votenet_coef_tail *t)
{
    CoefTail r = {};
    if (t) r = CoefTail{t->ticket, t->rows, t->gamma, t->coef, t->dgamma, t->dbeta};
    return r;
}
#ifdef __HIPCC__
// Call at the very end of a kernel, by ALL threads of every workgroup, after the workgroup's atomics on sums.  The last
// workgroup to take a ticket computes bn_bwd_coef_kernel's result (mlp_bwd.hip) from device-scope loads of the sums.
// Ordering without a fence: the sums are updated by device-scope atomics only, which execute at the memory side; a thread waits
// for the acknowledgement of its own atomics (vmcnt) before the workgroup takes its ticket, so whoever draws the last ticket
// finds every update in place.  __threadfence() would be correct too and costs 2 ms per train step: a device-scope release
// writes the XCD's L2 back, i.e. the whole output tile the kernel has just stored.
__device__ __forceinline__ void coef_tail(const CoefTail &t, unsigned nwg, int c, const double *sums, const float *scale,
                                          const float *shift, const float *mean, const float *var, float eps)
{
    if (!t.ticket) return;
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(t.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const double invn = 1.0 / (double)t.rows;
    for (int col = threadIdx.x; col < c; col += blockDim.x) {
        const double q1 = __hip_atomic_load(&sums[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double q2 = __hip_atomic_load(&sums[c + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float inv = 1.0f / sqrtf(var[col] + eps);
        const float m1 = (float)(q1 * invn), m2 = (float)(q2 * invn);
        const float A = t.gamma[col] * inv;
        const float C = -A * inv * m2;
        t.coef[col] = A;
        t.coef[c + col] = -A * m1 - C * mean[col];
        t.coef[2 * c + col] = C;
        t.coef[3 * c + col] = scale[col];
        t.coef[4 * c + col] = shift[col];
        if (t.dgamma) t.dgamma[col] += (float)q2;
        if (t.dbeta) t.dbeta[col] += (float)q1;
    }
    if (threadIdx.x == 0) *t.ticket = 0u;
}
#endif

// ---- spatial index of a batch of clouds (fps.hip builds it, grouping.hip reads it; include/votenet_hip.h) ----
// Points sorted by Morton cell into buckets of 64 consecutive sorted points.  One float scratch, four regions:
//   perm    b * n ints          original index of the p-th sorted point
//   bbox    b * nb * 6 floats   (xmin, ymin, zmin, xmax, ymax, zmax) of every bucket, nb = ceil(n / 64)
//   sorted  b * nb * 64 float4  16-byte aligned; (x, y, z, w): w = bits of the original index until a sampling kernel
//                               replaces it by its running distance -- readers take indices from perm
//   work    b * kSidxWork ints  per-workgroup cell histograms / offsets + partial bounds while the index is built
struct SpatialIndex {
    int *perm;
    float *bbox;
    float4 *sorted;
    int *work;
};
constexpr int kSidxCells = 4096;
constexpr int kSidxParts = 16;  // workgroups per scene: each counts / places its own slice of the points through LDS
constexpr int kSidxWork = kSidxCells * kSidxParts + 6 * kSidxParts + 8;
inline size_t spatial_index_floats(int b, int n)
{
    const size_t nb = (size_t)(n + 63) / 64;
    return (size_t)b * ((size_t)n + 6 * nb + 4 * 64 * nb + kSidxWork) + 4;
}
inline SpatialIndex spatial_index_view(float *base, int b, int n)
{
    const size_t nb = (size_t)(n + 63) / 64;
    SpatialIndex v;
    v.perm = reinterpret_cast<int *>(base);
    v.bbox = base + (size_t)b * n;
    v.sorted = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(v.bbox + (size_t)b * nb * 6) + 15) & ~(uintptr_t)15);
    v.work = reinterpret_cast<int *>(v.sorted + (size_t)b * nb * 64);
    return v;
}
int build_spatial_index(int b, int n, const float *xyz, float *index, hipStream_t st); // fps.hip

// ---- wave64 cross-lane helpers (DPP; no LDS traffic) ----
// dpp_ctrl encodings (gfx9): quad_perm 0x00-0xFF, row_shr:n 0x110+n, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast:15 0x142, row_bcast:31 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __uint_as_float(dpp_u32<CTRL, ROW_MASK>(__float_as_uint(v)));
}

// max over the 16 lanes of each DPP row; every lane of the row ends with the row result
__device__ __forceinline__ float row16_max_f32(float v)
{
    v = fmaxf(v, dpp_f32<0xB1>(v));  // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f32<0x4E>(v));  // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f32<0x141>(v)); // row_half_mirror
    v = fmaxf(v, dpp_f32<0x140>(v)); // row_mirror
    return v;
}
__device__ __forceinline__ unsigned row16_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    return v;
}
// full wave64 reductions; the result is returned wave-uniform (SGPR)
__device__ __forceinline__ float wave_max_f32(float v)
{
    v = row16_max_f32(v);
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v)); // row_bcast:15 into rows 1,3
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v)); // row_bcast:31 into rows 2,3
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = row16_min_u32(v);
    v = min(v, dpp_u32<0x142, 0xA>(v));
    v = min(v, dpp_u32<0x143, 0xC>(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_min_f32(float v)
{
    v = fminf(v, dpp_f32<0xB1>(v));
    v = fminf(v, dpp_f32<0x4E>(v));
    v = fminf(v, dpp_f32<0x141>(v));
    v = fminf(v, dpp_f32<0x140>(v));
    v = fminf(v, dpp_f32<0x142, 0xA>(v));
    v = fminf(v, dpp_f32<0x143, 0xC>(v));
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}
__device__ __forceinline__ float readlane_f32(float v, int lane)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0), i.e. it
// drains every global load / store in flight -- fatal for a software pipeline whose prefetched global
// loads are meant to stay outstanding across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

} // namespace votenet
