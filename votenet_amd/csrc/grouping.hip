// grouping.hip -- ball query, group_point and its gradient for gfx950.
//
// Replaces tf_ops/grouping/tf_grouping_g.cu:3-78,125-141 of the reference.
//
// Ball query.  The reference runs one 256-thread block per scene, each thread scanning all
// n candidates serially for its queries (tf_grouping_g.cu:13-35).  The result is order
// dependent (first nsample hits in ascending candidate index), so a parallel scan has to emit
// hits in candidate order.  Here a workgroup owns 64 queries (one per lane) and NW waves split
// each candidate "super-chunk" between them; every wave computes, for its slice, a hit
// BITMASK per query (32 candidates -> one dword, candidates broadcast from scalar loads, no
// divergence, no early-exit bookkeeping in the hot loop) into LDS.  After one barrier the
// waves switch roles: each takes 64/NW queries, pops the mask words in candidate order with
// a wave prefix sum of popcounts and writes the first nsample indices.  The scan stops as soon
// as all 64 queries of the workgroup are full (the reference's early exit, per workgroup).
//
// The hit test max(sqrtf(s),1e-20f) < r is evaluated as s < T(r), where T(r) is the smallest
// fp32 with sqrtf(T) >= r, computed on the host with correctly rounded sqrtf.  This is
// exactly equivalent for r > 1e-20 (sqrtf is monotone) and is NOT the same as s < r*r
// (SURVEY.md appendix A.3).
#include "common.h"
#include <cmath>

namespace votenet {


// G = 32-candidate groups per wave per super-chunk: NW * G * 32 candidates between barriers.  Small clouds (the levels below sa1,
// the proposal module's votes: n <= 2048) take 16 waves and a G that makes the cloud ONE super-chunk -- a wave's share of the scan is
// a serial chain of scalar loads (4 waves x 8 groups at n = 1024: 26 us for 8 x 256 queries; 16 x 2: 13 us, tools/probe/bq_small_time.py)
template <int NW, int G = 8>
__global__ __launch_bounds__(NW * 64) void ball_query_kernel(int n, int m, float thr, int nsample,
                                                              const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2, int *__restrict__ idx,
                                                              int *__restrict__ pts_cnt)
{
    constexpr int WPQ = NW * G;          // mask words per query per super-chunk
    constexpr int CHUNK = WPQ * 32;      // candidates per super-chunk
    constexpr int QPW = 64 / NW;         // queries finalised by each wave
    constexpr int LPQ = WPQ / 64 > 0 ? WPQ / 64 : 1; // mask words per lane in the pop phase (NW=16: 2, NW=4: 1 with half the lanes idle)
    __shared__ unsigned masks[64][WPQ + 1];

    const int scene = blockIdx.y;
    const float *__restrict__ cand = xyz1 + (size_t)scene * n * 3;
    const float *__restrict__ qry = xyz2 + (size_t)scene * m * 3;
    int *__restrict__ oidx = idx + (size_t)scene * m * nsample;
    int *__restrict__ ocnt = pts_cnt + (size_t)scene * m;

    const int lane = lane_id();
    const int w = wave_id_uniform();
    const int q0 = blockIdx.x * 64;
    const int q = q0 + lane;
    const bool qvalid = q < m;
    const int qq = qvalid ? q : (m - 1);
    const float qx = qry[(size_t)qq * 3 + 0], qy = qry[(size_t)qq * 3 + 1], qz = qry[(size_t)qq * 3 + 2];

    int cnt[QPW];   // hits found so far for the queries this wave finalises (uniform)
    int first[QPW]; // first hit (uniform)
#pragma unroll
    for (int i = 0; i < QPW; i++) {
        cnt[i] = 0;
        first[i] = 0;
    }

    for (int base = 0; base < n; base += CHUNK) {
        // ---- phase 1: hit masks, lane = query, candidates uniform across the wave
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            const int wbase = __builtin_amdgcn_readfirstlane(base + (w * G + g) * 32);
            unsigned mask = 0;
            if (wbase < n) {
                const int valid = (n - wbase) < 32 ? (n - wbase) : 32;
                if (valid == 32) {
#pragma unroll
                    for (int t = 0; t < 32; t++) {
                        const float cx = cand[(size_t)(wbase + t) * 3 + 0];
                        const float cy = cand[(size_t)(wbase + t) * 3 + 1];
                        const float cz = cand[(size_t)(wbase + t) * 3 + 2];
                        const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
                        const float s = dx * dx + dy * dy + dz * dz; // tf_grouping_g.cu:24, un-fused
                        mask |= (s < thr ? 1u : 0u) << t;
                    }
                } else {
                    for (int t = 0; t < valid; t++) {
                        const float cx = cand[(size_t)(wbase + t) * 3 + 0];
                        const float cy = cand[(size_t)(wbase + t) * 3 + 1];
                        const float cz = cand[(size_t)(wbase + t) * 3 + 2];
                        const float dx = qx - cx, dy = qy - cy, dz = qz - cz;
                        const float s = dx * dx + dy * dy + dz * dz;
                        mask |= (s < thr ? 1u : 0u) << t;
                    }
                }
            }
            masks[lane][w * G + g] = mask;
        }
        __syncthreads();
        // ---- phase 2: pop masks in candidate order, wave w finalises queries w*QPW .. +QPW-1
        bool wave_full = true;
#pragma unroll
        for (int i = 0; i < QPW; i++) {
            const int ql = w * QPW + i;
            if (cnt[i] < nsample && q0 + ql < m) {
                unsigned long long bits = 0;
                int cbase = 0;
                if (WPQ >= 64) {
                    bits = (unsigned long long)masks[ql][lane * LPQ] |
                           ((unsigned long long)masks[ql][lane * LPQ + (LPQ > 1 ? 1 : 0)] << 32);
                    if (LPQ == 1) bits &= 0xFFFFFFFFull;
                    cbase = base + lane * LPQ * 32;
                } else {
                    bits = lane < WPQ ? (unsigned long long)masks[ql][lane] : 0ull;
                    cbase = base + lane * 32;
                }
                const int pc = __popcll(bits);
                int incl = pc; // inclusive prefix sum over lanes
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(incl, d);
                    if (lane >= d) incl += t;
                }
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (total > 0) {
                    if (cnt[i] == 0) {
                        const unsigned long long have = __ballot(pc > 0);
                        const int fl = __ffsll((long long)have) - 1;
                        const int mine = cbase + (bits ? __ffsll((long long)bits) - 1 : 0);
                        first[i] = __builtin_amdgcn_readlane(mine, fl);
                    }
                    int pos = cnt[i] + incl - pc;
                    int *__restrict__ row = oidx + (size_t)(q0 + ql) * nsample;
                    while (bits && pos < nsample) {
                        const int t = __ffsll((long long)bits) - 1;
                        row[pos] = cbase + t;
                        pos++;
                        bits &= bits - 1;
                    }
                    cnt[i] += total;
                }
            }
            if (cnt[i] < nsample && q0 + ql < m) wave_full = false;
        }
        // barrier: masks are rewritten by the next super-chunk; also the block-wide early exit
        if (__syncthreads_and(wave_full ? 1 : 0)) break;
    }
    // ---- epilogue: pad with the first hit (tf_grouping_g.cu:26-29), write pts_cnt (:34)
#pragma unroll
    for (int i = 0; i < QPW; i++) {
        const int ql = w * QPW + i;
        if (q0 + ql < m) {
            const int c = cnt[i] < nsample ? cnt[i] : nsample;
            int *__restrict__ row = oidx + (size_t)(q0 + ql) * nsample;
            for (int l = c + lane; l < nsample; l += 64) row[l] = first[i];
            if (lane == 0) ocnt[q0 + ql] = c;
        }
    }
}

// group_point, vectorised: c % 4 == 0.  One float4 per thread, rows = b*m*nsample.
__global__ void group_point_vec4_kernel(long rows, int n, int c4, long rows_per_scene, const float4 *__restrict__ points,
                                        const int *__restrict__ idx, float4 *__restrict__ out)
{
    const long total = rows * c4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / c4;
        const int v = (int)(e - row * c4);
        const long s = row / rows_per_scene;
        const int ii = idx[row];
        out[e] = points[((size_t)s * n + ii) * c4 + v];
    }
}

__global__ void group_point_kernel(long rows, int n, int c, long rows_per_scene, const float *__restrict__ points,
                                   const int *__restrict__ idx, float *__restrict__ out)
{
    const long total = rows * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / c;
        const int l = (int)(e - row * c);
        const long s = row / rows_per_scene;
        const int ii = idx[row];
        out[e] = points[((size_t)s * n + ii) * c + l];
    }
}

// group_point_grad: grad_points[s, idx[row], :] += grad_out[row, :]  (tf_grouping_g.cu:61-78).
// fp32 hardware atomics (global_atomic_add_f32); summation order is unspecified, as in the
// reference.
__global__ void group_point_grad_kernel(long rows, int n, int c, long rows_per_scene, const float *__restrict__ grad_out,
                                        const int *__restrict__ idx, float *__restrict__ grad_points)
{
    const long total = rows * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / c;
        const int l = (int)(e - row * c);
        const long s = row / rows_per_scene;
        const int ii = idx[row];
        unsafeAtomicAdd(&grad_points[((size_t)s * n + ii) * c + l], grad_out[e]);
    }
}

// smallest fp32 T with sqrtf(T) >= r (host, correctly rounded sqrtf)
float ball_threshold(float r)
{
    float t = r * r;
    while (t > 0.0f && sqrtf(t) >= r) t = nextafterf(t, 0.0f);
    while (sqrtf(t) < r) t = nextafterf(t, INFINITY);
    return t;
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

extern "C" float votenet_ball_threshold(float radius) { return ball_threshold(radius); }

static int g_bq_small = 0; // votenet_debug_ball_query_small: 0 = by cloud size, 4 = the four-wave kernel, 16 = sixteen waves x eight groups
extern "C" void votenet_debug_ball_query_small(int form) { VN_DEBUG_GATE(); g_bq_small = form; }

extern "C" int votenet_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                        const float *xyz2, int *idx, int *pts_cnt, void *stream)
{
    VN_REQUIRE(radius > 0, "QueryBallPoint expects positive radius");   // tf_grouping.cpp:71
    VN_REQUIRE(nsample > 0, "QueryBallPoint expects positive nsample"); // tf_grouping.cpp:74
    VN_REQUIRE(b >= 0 && n > 0, "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape."); // :79
    VN_REQUIRE(m >= 0, "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.");            // :84
    if (b == 0 || m == 0) return VOTENET_OK;
    VN_REQUIRE(xyz1 && xyz2 && idx && pts_cnt, "QueryBallPoint: null buffer");
    // d = max(sqrtf(s),1e-20f) >= 1e-20 : nothing can be closer than a radius <= 1e-20
    const float thr = (radius <= 1e-20f) ? -1.0f : ball_threshold(radius);
    hipStream_t st = as_stream(stream);
    dim3 grid((m + 63) / 64, b);
    if (n > 2048 || g_bq_small == 16)
        hipLaunchKernelGGL((ball_query_kernel<16>), grid, dim3(1024), 0, st, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt);
    else if (g_bq_small == 4)
        hipLaunchKernelGGL((ball_query_kernel<4>), grid, dim3(256), 0, st, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt);
    else if (n > 1024)
        hipLaunchKernelGGL((ball_query_kernel<16, 4>), grid, dim3(1024), 0, st, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt);
    else if (n > 512)
        hipLaunchKernelGGL((ball_query_kernel<16, 2>), grid, dim3(1024), 0, st, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt);
    else
        hipLaunchKernelGGL((ball_query_kernel<16, 1>), grid, dim3(1024), 0, st, n, m, thr, nsample, xyz1, xyz2, idx, pts_cnt);
    return check_launch("query_ball_point");
}

namespace votenet {
// ---------------------------------------------------------------- ball query over the spatial index
// One wave per query.  The candidates are the index's buckets of 64 Morton-sorted points (common.h: SpatialIndex, built
// by the farthest-point sampling of the same cloud or by votenet_spatial_index): the wave tests the query against all
// bucket boxes (lane = bucket, conservative: a bucket is skipped only if its box is farther than the threshold even
// after shrinking the bound by 1e-5), then evaluates the reference's hit test -- the same un-fused fp32 expression on
// the same coordinates -- only on the points of the surviving buckets (r = 0.2 in a 5 m room: ~15 of 320).  The result
// must be the first nsample hits in ORIGINAL index order (tf_grouping_g.cu:13-35) while the buckets arrive in Morton
// order: hits set bits in a per-wave LDS bitmap over the original indices (n bits), which is then read in index order
// with a wave prefix sum of popcounts -- the brute-force kernel's pop phase, on one n-bit row.  Neighbour lists and
// pts_cnt are identical to ball_query_kernel's; the pair tests drop from m * n to about m * n / 20.
constexpr int BQI_CH = 6; // 64-bucket chunks whose boxes are loaded together (sa1: 320 buckets = one pass)
constexpr int BQI_U = 6;  // buckets fetched together
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ball_query_indexed_kernel(int n, int m, float thr, int nsample,
                                                                        const float *__restrict__ xyz2, const int *__restrict__ perm,
                                                                        const float *__restrict__ bbox,
                                                                        const float4 *__restrict__ sorted, int *__restrict__ idx,
                                                                        int *__restrict__ pts_cnt)
{
    extern __shared__ unsigned s_bits[]; // WAVES x nw32 dwords
    const int scene = blockIdx.y;
    const int lane = lane_id();
    const int w = wave_id_uniform();
    const int q = blockIdx.x * WAVES + w;
    if (q >= m) return; // wave-uniform; no workgroup barrier below
    const int nb = (n + 63) / 64;
    const int nw32 = (n + 31) / 32;
    const int wpl = (nw32 + 63) / 64; // bitmap dwords per lane in the read-out
    unsigned *__restrict__ bits = s_bits + (size_t)w * wpl * 64;
    const float *__restrict__ qp = xyz2 + ((size_t)scene * m + q) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const int *__restrict__ pm = perm + (size_t)scene * n;
    const float *__restrict__ bb = bbox + (size_t)scene * nb * 6;
    const float4 *__restrict__ sp = sorted + (size_t)scene * nb * 64;
    for (int i = 0; i < wpl; i++) bits[i * 64 + lane] = 0u;
    for (int g00 = 0; g00 < nb; g00 += 64 * BQI_CH) {
        // box tests of up to BQI_CH x 64 buckets: all box loads of the chunk are issued before the first test
        float bl[BQI_CH][6];
#pragma unroll
        for (int h = 0; h < BQI_CH; h++) {
            const int g = g00 + h * 64 + lane;
            const int gg = g < nb ? g : nb - 1;
#pragma unroll
            for (int t = 0; t < 6; t++) bl[h][t] = bb[gg * 6 + t];
        }
        unsigned long long acts[BQI_CH];
#pragma unroll
        for (int h = 0; h < BQI_CH; h++) {
            const int g = g00 + h * 64 + lane;
            const float ex = fmaxf(fmaxf(bl[h][0] - qx, qx - bl[h][3]), 0.0f);
            const float ey = fmaxf(fmaxf(bl[h][1] - qy, qy - bl[h][4]), 0.0f);
            const float ez = fmaxf(fmaxf(bl[h][2] - qz, qz - bl[h][5]), 0.0f);
            acts[h] = __ballot(g < nb && !((ex * ex + ey * ey + ez * ez) * 0.99999f >= thr));
        }
#pragma unroll
        for (int h = 0; h < BQI_CH; h++) {
            unsigned long long act = acts[h];
            const int g0 = g00 + h * 64;
            while (act) { // up to BQI_U buckets per pass: their loads are in flight together
                int gi[BQI_U];
                float4 c[BQI_U];
                int k[BQI_U];
#pragma unroll
                for (int u = 0; u < BQI_U; u++) {
                    gi[u] = -1;
                    if (act) {
                        gi[u] = g0 + __ffsll((long long)act) - 1;
                        act &= act - 1;
                    }
                }
#pragma unroll
                for (int u = 0; u < BQI_U; u++) {
                    const int p = (gi[u] >= 0 ? gi[u] : 0) * 64 + lane;
                    const bool ok = gi[u] >= 0 && p < n;
                    c[u] = ok ? sp[p] : make_float4(0.f, 0.f, 0.f, 0.f);
                    k[u] = ok ? pm[p] : -1;
                }
#pragma unroll
                for (int u = 0; u < BQI_U; u++) {
                    const float dx = qx - c[u].x, dy = qy - c[u].y, dz = qz - c[u].z;
                    const float s = dx * dx + dy * dy + dz * dz; // tf_grouping_g.cu:24, un-fused
                    if (k[u] >= 0 && s < thr) atomicOr(&bits[k[u] >> 5], 1u << (k[u] & 31));
                }
            }
        }
    }
    // read-out in index order: lane l owns bitmap dwords [l * wpl, (l + 1) * wpl)
    int pc = 0;
    for (int i = 0; i < wpl; i++) pc += __popc(bits[lane * wpl + i]);
    int incl = pc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    int *__restrict__ row = idx + ((size_t)scene * m + q) * nsample;
    int first = 0;
    if (total > 0) {
        int pos = incl - pc;
        int myfirst = -1;
        for (int i = 0; i < wpl && (pos < nsample || myfirst < 0); i++) {
            unsigned v = bits[lane * wpl + i];
            const int base = (lane * wpl + i) * 32;
            if (v && myfirst < 0) myfirst = base + __ffs((int)v) - 1;
            while (v && pos < nsample) {
                row[pos++] = base + __ffs((int)v) - 1;
                v &= v - 1;
            }
        }
        const unsigned long long have = __ballot(pc > 0);
        first = __builtin_amdgcn_readlane(myfirst, __ffsll((long long)have) - 1);
    }
    const int cnt = total < nsample ? total : nsample;
    for (int l = cnt + lane; l < nsample; l += 64) row[l] = first; // tf_grouping_g.cu:26-29 (all-zero row without a hit)
    if (lane == 0) pts_cnt[(size_t)scene * m + q] = cnt;
}
} // namespace votenet

// The ball query of a cloud whose spatial index exists (votenet_spatial_index, or the temp scratch of a
// votenet_farthest_point_sample call on the same cloud with 4096 < n <= 262144): same results as votenet_query_ball_point.
extern "C" int votenet_query_ball_point_indexed(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                                const float *xyz2, const float *index, int *idx, int *pts_cnt, void *stream)
{
    VN_REQUIRE(radius > 0, "QueryBallPoint expects positive radius");   // tf_grouping.cpp:71
    VN_REQUIRE(nsample > 0, "QueryBallPoint expects positive nsample"); // tf_grouping.cpp:74
    VN_REQUIRE(b >= 0 && n > 0, "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape."); // :79
    VN_REQUIRE(m >= 0, "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.");            // :84
    if (b == 0 || m == 0) return VOTENET_OK;
    VN_REQUIRE(xyz1 && xyz2 && idx && pts_cnt, "QueryBallPoint: null buffer");
    if (!index || n > 131072) // no index (or a bitmap row that would not fit the LDS): the brute-force scan
        return votenet_query_ball_point(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, stream);
    const float thr = (radius <= 1e-20f) ? -1.0f : ball_threshold(radius);
    const SpatialIndex v = spatial_index_view(const_cast<float *>(index), b, n);
    const int wpl = ((n + 31) / 32 + 63) / 64;
    const size_t lds = (size_t)4 * wpl * 64 * 4;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ball_query_indexed_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL((ball_query_indexed_kernel<4>), dim3((m + 3) / 4, b), dim3(256), lds, as_stream(stream), n, m, thr, nsample,
                       xyz2, (const int *)v.perm, (const float *)v.bbox, (const float4 *)v.sorted, idx, pts_cnt);
    return check_launch("query_ball_point_indexed");
}

extern "C" int votenet_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                                   float *out, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && c > 0, "GroupPoint expects (batch_size, num_points, channel) points shape"); // tf_grouping.cpp:149
    VN_REQUIRE(m >= 0 && nsample >= 0, "GroupPoint expects (batch_size, npoints, nsample) idx shape");        // :155
    const long rows = (long)b * m * nsample;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(points && idx && out, "GroupPoint: null buffer");
    hipStream_t st = as_stream(stream);
    const long rps = (long)m * nsample;
    if (c % 4 == 0 && ((uintptr_t)points % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
        const long total = rows * (c / 4);
        hipLaunchKernelGGL(group_point_vec4_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, rows, n, c / 4, rps,
                           (const float4 *)points, idx, (float4 *)out);
    } else {
        hipLaunchKernelGGL(group_point_kernel, dim3(grid_for(rows * c, 256)), dim3(256), 0, st, rows, n, c, rps, points, idx,
                           out);
    }
    return check_launch("group_point");
}

extern "C" int votenet_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                        float *grad_points, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && c > 0, "GroupPointGrad expects (batch_size, num_points, channel) points shape"); // :180
    VN_REQUIRE(m >= 0 && nsample >= 0, "GroupPointGrad expects (batch_size, npoints, nsample) idx shape");        // :186
    const long rows = (long)b * m * nsample;
    if (rows == 0) return VOTENET_OK;
    VN_REQUIRE(grad_out && idx && grad_points, "GroupPointGrad: null buffer");
    hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid_for(rows * c, 256)), dim3(256), 0, as_stream(stream), rows, n, c,
                       (long)m * nsample, grad_out, idx, grad_points);
    return check_launch("group_point_grad");
}

// ---- reference launcher names, C++ linkage, exact signatures (tf_grouping.cpp:66,142,173)
VN_EXPORT void queryBallPointLauncher(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2, int *idx,
                            int *pts_cnt)
{
    votenet_query_ball_point(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt, nullptr);
}
VN_EXPORT void groupPointLauncher(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out)
{
    votenet_group_point(b, n, c, m, nsample, points, idx, out, nullptr);
}
VN_EXPORT void groupPointGradLauncher(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx, float *grad_points)
{
    votenet_group_point_grad(b, n, c, m, nsample, grad_out, idx, grad_points, nullptr);
}
