// csr.hip -- deterministic "scatter-add" of the backward pass: gather-sums over an inverse index (gfx950).
//
// GroupPointGrad / ThreeInterpolateGrad (tf_grouping_g.cu:61-78, tf_interpolate.cpp:131-153) add every grouped row's
// gradient to the point it was gathered from; on a GPU that is fp32 atomics -- summation order unspecified, results not
// reproducible run to run.  The grouping (ball query / three_nn) depends on coordinates only, so its INVERSE is known
// before the backward pass (built with the geometry, on the geometry stream): for every point the list of the slots that
// reference it, in ascending slot order (CSR: offsets (points + 1), order (slots); votenet_amd/pointnet2.py builds it with a
// stable sort).  The gradients of a point are then summed by ONE thread per channel in that fixed order: no atomics, no
// zero-fill of the target, every source row still read exactly once.
//   csr_gather_sum_kernel        out[p, :] = sum_t w[order[t]] * src[order[t] / div, :]      (generic: xyz / feature gradients
//                                of a grouping, three_interpolate's gradient with its weights, div = 3)
//   group_linear_bwd_gather      votenet_group_linear_backward on the inverse index: dz formed from (z, da, coef) on the fly,
//                                S[p, :] = sum of the dz rows of point p; the xyz rows of the weight gradient as
//                                per-workgroup partials + ordered reduction; optional dz output
#include "mlp_types.h"

namespace votenet {

void wgrad_reduce(int nslice, long pstride, long e0, long e1, const float *part, float *dw, hipStream_t st); // mlp_bwd.hip

// thread = (point lane, channel): TPP = threads per point (>= c, a divisor of 256)
__global__ __launch_bounds__(256) void csr_gather_sum_kernel(long npts, int c, int tpp, const float *__restrict__ src, long pitch,
                                                             const int *__restrict__ order, const int *__restrict__ offsets,
                                                             const float *__restrict__ weight, int div, float *__restrict__ out)
{
    const int ppb = 256 / tpp;
    const int ch = threadIdx.x % tpp, pl = threadIdx.x / tpp;
    for (long p = (long)blockIdx.x * ppb + pl; p < npts; p += (long)gridDim.x * ppb) {
        const int t0 = offsets[p], t1 = offsets[p + 1];
        float acc = 0.0f;
        int t = t0;
        for (; t + 4 <= t1; t += 4) { // four source rows in flight, added in list order
            const int s0 = order[t], s1 = order[t + 1], s2 = order[t + 2], s3 = order[t + 3];
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            if (ch < c) {
                v0 = src[(size_t)(s0 / div) * pitch + ch];
                v1 = src[(size_t)(s1 / div) * pitch + ch];
                v2 = src[(size_t)(s2 / div) * pitch + ch];
                v3 = src[(size_t)(s3 / div) * pitch + ch];
            }
            if (weight) {
                v0 *= weight[s0];
                v1 *= weight[s1];
                v2 *= weight[s2];
                v3 *= weight[s3];
            }
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; t < t1; t++) {
            const int s = order[t];
            float v = ch < c ? src[(size_t)(s / div) * pitch + ch] : 0.0f;
            if (weight) v *= weight[s];
            acc += v;
        }
        if (ch < c) out[(size_t)p * c + ch] = acc;
    }
}

// thread = (point lane, channel QUAD): cout / 4 threads per point, 1024 / cout points per workgroup pass -- four points per
// wavefront at cout = 64, so that a wavefront has the row loads of four independent points in flight (one point per
// wavefront was a chain of dependent round trips: offsets -> order -> rows, 0.34 ms at sa1 against 0.20 ms for the atomics)
__global__ __launch_bounds__(256) void group_linear_bwd_gather_kernel(long npts, int n, int groups_per_scene, int nsample, int cout,
                                                                      const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                      const int *__restrict__ order, const int *__restrict__ offsets,
                                                                      const int *__restrict__ visit, const float *__restrict__ z,
                                                                      const float *__restrict__ da, const float *__restrict__ coef, int relu,
                                                                      float *__restrict__ spt, float *__restrict__ part,
                                                                      float *__restrict__ dz_out)
{
    __shared__ float red[256][12];
    const int tid = threadIdx.x;
    const int tpp = cout >> 2, ppb = 256 / tpp;
    const int cq = tid % tpp, pl = tid / tpp;
    const float4 kA = *reinterpret_cast<const float4 *>(coef + 4 * cq), kB = *reinterpret_cast<const float4 *>(coef + cout + 4 * cq);
    const float4 kC = *reinterpret_cast<const float4 *>(coef + 2 * cout + 4 * cq), kS = *reinterpret_cast<const float4 *>(coef + 3 * cout + 4 * cq);
    const float4 kH = *reinterpret_cast<const float4 *>(coef + 4 * cout + 4 * cq);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    for (long pi = (long)blockIdx.x * ppb + pl; pi < npts; pi += (long)gridDim.x * ppb) {
        const long p = visit ? visit[pi] : pi; // visiting order of the points (any fixed permutation)
        const int t0 = offsets[p], t1 = offsets[p + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t1 > t0) {
            const float px = xyz[(size_t)p * 3 + 0], py = xyz[(size_t)p * 3 + 1], pz = xyz[(size_t)p * 3 + 2];
            auto one = [&](int s, const float4 &zz, float4 g) {
                if (relu) {
                    if (!(zz.x * kS.x + kH.x > 0.0f)) g.x = 0.0f;
                    if (!(zz.y * kS.y + kH.y > 0.0f)) g.y = 0.0f;
                    if (!(zz.z * kS.z + kH.z > 0.0f)) g.z = 0.0f;
                    if (!(zz.w * kS.w + kH.w > 0.0f)) g.w = 0.0f;
                }
                const float4 d = make_float4(kA.x * g.x + kB.x + kC.x * zz.x, kA.y * g.y + kB.y + kC.y * zz.y,
                                             kA.z * g.z + kB.z + kC.z * zz.z, kA.w * g.w + kB.w + kC.w * zz.w);
                if (dz_out) *reinterpret_cast<float4 *>(dz_out + (size_t)s * cout + 4 * cq) = d;
                const size_t g3 = (size_t)(s / nsample) * 3;
                const float ex = px - new_xyz[g3 + 0], ey = py - new_xyz[g3 + 1], ez = pz - new_xyz[g3 + 2]; // utils.py:55
                a0.x += ex * d.x; a0.y += ex * d.y; a0.z += ex * d.z; a0.w += ex * d.w;
                a1.x += ey * d.x; a1.y += ey * d.y; a1.z += ey * d.z; a1.w += ey * d.w;
                a2.x += ez * d.x; a2.y += ez * d.y; a2.z += ez * d.z; a2.w += ez * d.w;
                acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
            };
            int t = t0;
            for (; t + 4 <= t1; t += 4) { // four rows in flight per point, consumed in list order
                int sl[4];
                float4 zz[4], gg[4];
#pragma unroll
                for (int u = 0; u < 4; u++) sl[u] = order[t + u];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    zz[u] = *reinterpret_cast<const float4 *>(z + (size_t)sl[u] * cout + 4 * cq);
                    gg[u] = *reinterpret_cast<const float4 *>(da + (size_t)sl[u] * cout + 4 * cq);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) one(sl[u], zz[u], gg[u]);
            }
            for (; t < t1; t++) {
                const int s = order[t];
                one(s, *reinterpret_cast<const float4 *>(z + (size_t)s * cout + 4 * cq),
                    *reinterpret_cast<const float4 *>(da + (size_t)s * cout + 4 * cq));
            }
        }
        *reinterpret_cast<float4 *>(spt + (size_t)p * cout + 4 * cq) = acc; // every point is written: no zero fill of S
    }
    float *r = red[tid];
    r[0] = a0.x; r[1] = a0.y; r[2] = a0.z; r[3] = a0.w;
    r[4] = a1.x; r[5] = a1.y; r[6] = a1.z; r[7] = a1.w;
    r[8] = a2.x; r[9] = a2.y; r[10] = a2.z; r[11] = a2.w;
    __syncthreads();
    if (tid < cout) { // channel tid: quad tid / 4, component tid % 4; point lanes added in ascending order
        const int q = tid >> 2, e = tid & 3;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float t = 0.0f;
            for (int l = 0; l < ppb; l++) t += red[l * tpp + q][d * 4 + e];
            part[((size_t)blockIdx.x * 3 + d) * cout + tid] = t;
        }
    }
}

// The inverse of a small grouping in ONE launch, one workgroup per scene: idx (b, slots) with values in [0, m) -> offsets (b*m + 1),
// order (b*slots) -- for every target the slots (flat positions of idx) that reference it, ASCENDING (one fixed summation order for
// the gather-sums above).  Counts and cursors live in LDS (m <= kInvMaxTargets); the fill uses LDS atomics, so a target's list comes
// out in arrival order and is sorted in place afterwards by the thread that owns the target (the lists are short: three_nn's taps,
// ~6 per target).  Replaces a stable sort + searchsorted of the tensor library (10 launches) for the taps of three_interpolate.
constexpr int kInvMaxTargets = 8192;
constexpr int kInvShortList = 24;   // lists up to this length: insertion sort by the thread that owns the target
constexpr int kInvLongLists = 1024; // longer ones queue up for the wavefronts (a queue overflow falls back to the owner's insertion sort)
constexpr int kInvRankPerLane = 16; // a wavefront rank-sorts lists of up to 64 * 16 entries in registers
__global__ __launch_bounds__(1024) void inverse_index_kernel(int slots, int m, const int *__restrict__ idx, int *__restrict__ order,
                                                             int *__restrict__ offsets)
{
    __shared__ int cnt[kInvMaxTargets];
    __shared__ int wsum[16];
    __shared__ int longs[kInvLongLists];
    __shared__ int nlong;
    const int scene = blockIdx.x, tid = threadIdx.x;
    const int *__restrict__ si = idx + (size_t)scene * slots;
    const int base = scene * slots;
    for (int j = tid; j < m; j += 1024) cnt[j] = 0;
    __syncthreads();
    // an index outside [0, m) (a caller's bug, garbage memory) counts for target 0: the LDS tables are never indexed out of range
    auto target = [&](int t) { const unsigned v = (unsigned)si[t]; return v < (unsigned)m ? (int)v : 0; };
    for (int t = tid; t < slots; t += 1024) atomicAdd(&cnt[target(t)], 1);
    __syncthreads();
    // exclusive scan of cnt[0..m) in place: every thread owns a contiguous run of PER targets
    const int PER = (m + 1023) / 1024;
    const int j0 = tid * PER;
    int mine = 0;
    for (int q = 0; q < PER; q++) mine += (j0 + q < m) ? cnt[j0 + q] : 0;
    int incl = mine;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; w++) before += wsum[w];
    int run = before + incl - mine;
    for (int q = 0; q < PER; q++) {
        if (j0 + q < m) {
            const int c = cnt[j0 + q];
            offsets[(size_t)scene * m + j0 + q] = base + run;
            cnt[j0 + q] = run; // the cursor of the fill
            run += c;
        }
    }
    if (scene == (int)gridDim.x - 1 && tid == 0) offsets[(size_t)gridDim.x * m] = (int)gridDim.x * slots;
    __syncthreads();
    for (int t = tid; t < slots; t += 1024) order[base + atomicAdd(&cnt[target(t)], 1)] = base + t;
    __syncthreads(); // the lists are complete (global writes of this workgroup, read back by this workgroup below)
    __threadfence_block();
    if (tid == 0) nlong = 0;
    __syncthreads();
    int *__restrict__ l = order + base;
    auto insertion = [&](int lo, int hi) { // ascending, by one thread: O(L^2), for the short lists (three_nn's taps: ~6 per target)
        for (int a = lo + 1; a < hi; a++) {
            const int v = l[a];
            int k = a - 1;
            while (k >= lo && l[k] > v) {
                l[k + 1] = l[k];
                k--;
            }
            l[k + 1] = v;
        }
    };
    for (int j = tid; j < m; j += 1024) {
        const int hi = cnt[j]; // the cursor now stands at the end of the list
        const int lo = (j == 0) ? 0 : cnt[j - 1];
        // cnt[j - 1] is the END of list j - 1 = the start of list j (lists are contiguous in target order)
        if (hi - lo <= kInvShortList) {
            insertion(lo, hi);
        } else { // a long list (duplicate points, holes mapped to point 0: one target referenced by hundreds of slots): a wavefront's job
            const int q = atomicAdd(&nlong, 1);
            if (q < kInvLongLists)
                longs[q] = j;
            else
                insertion(lo, hi);
        }
    }
    __syncthreads();
    // long lists: rank sort by one wavefront -- the values are distinct slots, so rank(v) = #{u in list : u < v} is v's final position;
    // up to 16 values per lane stay in registers with their ranks until every read of the list is done, then they are written in place
    const int nl = min(nlong, kInvLongLists);
    for (int q = wv; q < nl; q += 16) {
        const int j = longs[q];
        const int hi = cnt[j], lo = (j == 0) ? 0 : cnt[j - 1];
        const int len = hi - lo;
        if (len > 64 * kInvRankPerLane) { // beyond the register budget: the slow way, still correct
            if (lane == 0) insertion(lo, hi);
            continue;
        }
        int val[kInvRankPerLane], rank[kInvRankPerLane];
#pragma unroll
        for (int e = 0; e < kInvRankPerLane; e++) {
            const int a = lane + 64 * e;
            val[e] = a < len ? l[lo + a] : 0x7fffffff;
            rank[e] = 0;
        }
        for (int c0 = 0; c0 < len; c0 += 64) {
            const int mine = (c0 + lane < len) ? l[lo + c0 + lane] : 0x7fffffff;
            const int lim = min(64, len - c0);
            for (int u = 0; u < lim; u++) {
                const int other = __shfl(mine, u);
#pragma unroll
                for (int e = 0; e < kInvRankPerLane; e++) rank[e] += other < val[e] ? 1 : 0;
            }
        }
        // (every lane of the wavefront has finished reading the list: the loop above is wave-synchronous)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int e = 0; e < kInvRankPerLane; e++)
            if (lane + 64 * e < len) l[lo + rank[e]] = val[e];
    }
}

} // namespace votenet
using namespace votenet;

extern "C" int votenet_inverse_index(int b, int slots, int m, const int *idx, int *order, int *offsets, void *stream)
{
    VN_REQUIRE(b >= 0 && slots > 0 && m > 0 && m <= kInvMaxTargets, "inverse_index expects b >= 0, slots > 0, 0 < m <= 8192 targets per scene");
    VN_REQUIRE((long)b * slots < (1L << 31), "inverse_index: more than 2^31 slots");
    if (b == 0) return VOTENET_OK;
    VN_REQUIRE(idx && order && offsets, "inverse_index: null buffer");
    hipLaunchKernelGGL(inverse_index_kernel, dim3(b), dim3(1024), 0, as_stream(stream), slots, m, idx, order, offsets);
    return check_launch("inverse_index");
}

static int csr_gather_sum_launch(long npts, int c, const float *src, long pitch, const int *order, const int *offsets, const float *weight,
                                 int div, float *out, void *stream)
{
    VN_REQUIRE(npts >= 0 && c > 0 && c <= 256 && div > 0 && pitch >= c, "csr_gather_sum expects npts >= 0, 0 < c <= 256, div > 0, pitch >= c");
    if (npts == 0) return VOTENET_OK;
    VN_REQUIRE(src && order && offsets && out, "csr_gather_sum: null buffer");
    int tpp = 1;
    while (tpp < c) tpp <<= 1;
    const int ppb = 256 / tpp;
    long gx = (npts + ppb - 1) / ppb;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(csr_gather_sum_kernel, dim3((unsigned)gx), dim3(256), 0, as_stream(stream), npts, c, tpp, src, pitch, order, offsets,
                       weight, div, out);
    return check_launch("csr_gather_sum");
}

extern "C" int votenet_csr_gather_sum(long npts, int c, const float *src, const int *order, const int *offsets, const float *weight,
                                      int div, float *out, void *stream)
{
    return csr_gather_sum_launch(npts, c, src, c, order, offsets, weight, div, out, stream);
}

extern "C" int votenet_csr_gather_sum_pitched(long npts, int c, const float *src, long src_pitch, const int *order, const int *offsets,
                                              const float *weight, int div, float *out, void *stream)
{
    return csr_gather_sum_launch(npts, c, src, src_pitch, order, offsets, weight, div, out, stream);
}

static long glbg_grid(long npts, int ppb)
{
    long gx = (npts + ppb - 1) / ppb;
    return gx > 2048 ? 2048 : gx;
}

extern "C" size_t votenet_group_linear_backward_scratch_floats(int b, int n, int cout)
{
    if (cout <= 0 || 256 % cout != 0) return 0;
    return (size_t)glbg_grid((long)b * n, 1024 / cout) * 3 * cout;
}

// votenet_group_linear_backward over the grouping's inverse index (order: b*m*nsample slots sorted by (scene, point), ascending
// slot inside a point; offsets: b*n + 1): s_points need NOT be zeroed, dw_xyz += the ordered sum of the workgroups' partials.
extern "C" int votenet_group_linear_backward_csr(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                                 const int *order, const int *offsets, const int *visit, const float *z,
                                                 const float *da, const float *coef, int relu, float *s_points, float *dw_xyz,
                                                 float *dz_out, float *scratch, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0 && nsample > 0 && cout > 0, "group_linear_backward_csr: bad shape");
    VN_REQUIRE(cout == 32 || cout == 64 || cout == 128 || cout == 256, "group_linear_backward_csr expects cout in {32, 64, 128, 256}");
    const long npts = (long)b * n;
    if (npts == 0) return VOTENET_OK;
    VN_REQUIRE((long)b * m * nsample < (1L << 31), "group_linear_backward_csr: b*m*nsample must be below 2^31");
    VN_REQUIRE(xyz && new_xyz && order && offsets && z && da && coef && s_points && dw_xyz && scratch, "group_linear_backward_csr: null buffer");
    hipStream_t st = as_stream(stream);
    VN_REQUIRE((uintptr_t)z % 16 == 0 && (uintptr_t)da % 16 == 0 && (uintptr_t)coef % 16 == 0 && (uintptr_t)s_points % 16 == 0 &&
                   (uintptr_t)dz_out % 16 == 0, "group_linear_backward_csr: z, da, coef, s_points, dz_out must be 16-byte aligned");
    const long gx = glbg_grid(npts, 1024 / cout);
    hipLaunchKernelGGL(group_linear_bwd_gather_kernel, dim3((unsigned)gx), dim3(256), 0, st, npts, n, m, nsample, cout, xyz, new_xyz,
                       order, offsets, visit, z, da, coef, relu, s_points, scratch, dz_out);
    wgrad_reduce((int)gx, (long)3 * cout, 0, (long)3 * cout, scratch, dw_xyz, st);
    return check_launch("group_linear_backward_csr");
}
