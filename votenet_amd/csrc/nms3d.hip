// nms3d.hip -- rotated-box 3D IoU matrix and greedy batch NMS for gfx950.
//
// Replaces the CPU-only op of tf_ops/3d_nms/tf_nms3d.cpp (geometry :43-192, greedy loop
// :202-273), which the reference reaches through a device->host copy in the predict tower
// (model.py:133).  The reference evaluates IoUs lazily, pair by pair, inside a serial loop with
// heap allocations per pair.  Here:
//   1. iou3d_matrix_kernel : one thread per ordered pair (i,j) of a scene, fixed-size vertex
//      buffers in registers/scratch, the reference's arithmetic kept step for step (fp32 ray
//      test, DOUBLE line intersection with the |det| < 1e-7 parallel test, atan2f vertex sort,
//      fabsf triangle fan);
//   2. nms_rank_kernel     : global visit order = descending score over the whole batch
//      (rank by counting; equal scores -> ascending flat index);
//   3. nms_greedy_kernel   : one wave per scene walks that scene's candidates in visit order,
//      lanes test the candidate against the scene's kept boxes in parallel;
//   4. nms_emit_kernel     : compacts kept boxes in global visit order into (count, [batch,box]).
// Everything stays on the device; the data-dependent output length is returned through
// *out_count.
#include "common.h"

namespace votenet {

struct P2 {
    float x, z;
};

// tf_nms3d.cpp:43-46
__device__ __forceinline__ float box_area2d(const float *bb)
{
    return sqrtf((bb[0] - bb[3]) * (bb[0] - bb[3]) + (bb[2] - bb[5]) * (bb[2] - bb[5])) *
           sqrtf((bb[3] - bb[6]) * (bb[3] - bb[6]) + (bb[5] - bb[8]) * (bb[5] - bb[8]));
}
// tf_nms3d.cpp:48-50
__device__ __forceinline__ float box_area3d(const float *bb) { return box_area2d(bb) * (bb[1] - bb[13]); }

// tf_nms3d.cpp:53-67 : even-odd ray test against the first four corners
__device__ __forceinline__ bool point_in_quad(float px, float pz, const float *poly)
{
    bool result = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = (i + 3) & 3;
        const float xi = poly[i * 3], zi = poly[i * 3 + 2], xj = poly[j * 3], zj = poly[j * 3 + 2];
        if ((zi > pz) != (zj > pz) && (px < (xj - xi) * (pz - zi) / (zj - zi) + xi)) result = !result;
    }
    return result;
}

// tf_nms3d.cpp:69-100
__device__ __forceinline__ bool seg_intersect(float ax, float az, float bx, float bz, float cx, float cz, float dx,
                                              float dz, P2 *out)
{
    const double A1 = (double)(bz - az);
    const double B1 = (double)(ax - bx);
    const double C1 = A1 * (double)ax + B1 * (double)az;
    const double A2 = (double)(dz - cz);
    const double B2 = (double)(cx - dx);
    const double C2 = A2 * (double)cx + B2 * (double)cz;
    const double det = A1 * B2 - A2 * B1;
    if (fabs(det) < 1e-7) return false;
    const double x = (B2 * C1 - B1 * C2) / det;
    const double z = (A1 * C2 - A2 * C1) / det;
    const bool on1 = ((double)fminf(ax, bx) <= x) && ((double)fmaxf(ax, bx) >= x) && ((double)fminf(az, bz) <= z) &&
                     ((double)fmaxf(az, bz) >= z);
    const bool on2 = ((double)fminf(cx, dx) <= x) && ((double)fmaxf(cx, dx) >= x) && ((double)fminf(cz, dz) <= z) &&
                     ((double)fmaxf(cz, dz) >= z);
    if (on1 && on2) {
        out->x = (float)x;
        out->z = (float)z;
        return true;
    }
    return false;
}

// tf_nms3d.cpp:122-175 : at most 4 + 4 + 16 vertices
__device__ float bev_intersection(const float *b1, const float *b2)
{
    P2 cc[24];
    float ang[24];
    int nc = 0;
    for (int i = 0; i < 4; i++)
        if (point_in_quad(b1[i * 3], b1[i * 3 + 2], b2)) {
            cc[nc].x = b1[i * 3];
            cc[nc].z = b1[i * 3 + 2];
            nc++;
        }
    for (int i = 0; i < 4; i++)
        if (point_in_quad(b2[i * 3], b2[i * 3 + 2], b1)) {
            cc[nc].x = b2[i * 3];
            cc[nc].z = b2[i * 3 + 2];
            nc++;
        }
    for (int i = 0; i < 4; i++) {
        const int nx = (i + 1) & 3;
        for (int e = 0; e < 4; e++) {
            const int en = (e + 1) & 3;
            P2 ip;
            if (seg_intersect(b1[i * 3], b1[i * 3 + 2], b1[nx * 3], b1[nx * 3 + 2], b2[e * 3], b2[e * 3 + 2], b2[en * 3],
                              b2[en * 3 + 2], &ip)) {
                cc[nc] = ip;
                nc++;
            }
        }
    }
    if (nc == 0) return 0.0f; // reference: 0/0 centroid, both loops skipped, area 0
    float mx = 0, mz = 0;
    for (int i = 0; i < nc; i++) {
        mx += cc[i].x;
        mz += cc[i].z;
    }
    mx /= (float)nc;
    mz /= (float)nc;
    for (int i = 0; i < nc; i++) ang[i] = atan2f(cc[i].z - mz, cc[i].x - mx);
    // insertion sort by angle (what std::sort does for <= 16 elements; stable)
    for (int i = 1; i < nc; i++) {
        const P2 p = cc[i];
        const float a = ang[i];
        int j = i - 1;
        while (j >= 0 && a < ang[j]) {
            cc[j + 1] = cc[j];
            ang[j + 1] = ang[j];
            j--;
        }
        cc[j + 1] = p;
        ang[j + 1] = a;
    }
    float area = 0;
    for (int i = 0, j = nc - 1; i < nc; j = i++)
        area += fabsf((mx * (cc[i].z - cc[j].z) + cc[i].x * (cc[j].z - mz) + cc[j].x * (mz - cc[i].z)) / 2);
    return area;
}

// tf_nms3d.cpp:178-192
__device__ float iou3d_pair(const float *bi, const float *bj)
{
    const float inter2d = bev_intersection(bi, bj);
    const float top = fminf(bi[1], bj[1]);
    const float bot = fmaxf(bi[13], bj[13]);
    const float h = (top - bot) > 0.0f ? (top - bot) : 0.0f;
    const float inter3d = h * inter2d;
    return inter3d / (box_area3d(bi) + box_area3d(bj) - inter3d);
}

__global__ __launch_bounds__(64) void iou3d_matrix_kernel(int n, const float *__restrict__ bboxes, float *__restrict__ iou)
{
    const int scene = blockIdx.z;
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float *__restrict__ base = bboxes + (size_t)scene * n * 24;
    float bi[24], bj[24];
#pragma unroll
    for (int t = 0; t < 24; t++) {
        bi[t] = base[(size_t)i * 24 + t];
        bj[t] = base[(size_t)j * 24 + t];
    }
    iou[((size_t)scene * n + i) * n + j] = iou3d_pair(bi, bj);
}

// IoU of every box of set A against every box of set B of the same scene (detections x ground truth: evaluator.py:26-39)
__global__ __launch_bounds__(64) void iou3d_cross_kernel(int n, int m, const float *__restrict__ a, const float *__restrict__ bset,
                                                         float *__restrict__ iou)
{
    const int scene = blockIdx.z;
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    float bi[24], bj[24];
#pragma unroll
    for (int t = 0; t < 24; t++) {
        bi[t] = a[((size_t)scene * n + i) * 24 + t];
        bj[t] = bset[((size_t)scene * m + j) * 24 + t];
    }
    iou[((size_t)scene * n + i) * m + j] = iou3d_pair(bi, bj);
}

// visit order: rank[e] = number of candidates visited before flat element e; -1 if not a candidate.
// The scores and candidate flags of ALL boxes of the batch are walked by every candidate: they are staged through LDS in chunks
// (one coalesced pass per workgroup) and read back as broadcasts -- the loop over global memory took 213 us for 8 x 256 boxes.
constexpr int NMS_RANK_CHUNK = 2048;
__global__ __launch_bounds__(256) void nms_rank_kernel(int total, const float *__restrict__ scores, const float *__restrict__ obj,
                                                       int *__restrict__ order /* rank -> flat index */, int *__restrict__ ncand)
{
    __shared__ float2 s_sc[NMS_RANK_CHUNK]; // (score with NaN -> -inf, 1.0 if candidate else 0.0)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < total;
    const bool cand = live && obj[e * 2 + 1] > obj[e * 2]; // tf_nms3d.cpp:230
    // a TOTAL order even for the scores of a diverged model: NaN ranks with -inf (visited last), ties by flat index --
    // every candidate gets its own rank, so order[0 .. ncand) is a permutation of the candidates
    const float s0 = live ? scores[e] : 0.0f;
    const float s = (s0 != s0) ? -__builtin_inff() : s0;
    int rank = 0;
    for (int base = 0; base < total; base += NMS_RANK_CHUNK) {
        const int cnt = total - base < NMS_RANK_CHUNK ? total - base : NMS_RANK_CHUNK;
        __syncthreads();
        for (int f = threadIdx.x; f < cnt; f += blockDim.x) {
            const float t0 = scores[base + f];
            s_sc[f] = make_float2((t0 != t0) ? -__builtin_inff() : t0, obj[(base + f) * 2 + 1] > obj[(base + f) * 2] ? 1.0f : 0.0f);
        }
        __syncthreads();
        if (cand)
            for (int f = 0; f < cnt; f++) {
                const float2 tc = s_sc[f];
                if (tc.y != 0.0f && (tc.x > s || (tc.x == s && base + f < e))) rank++;
            }
    }
    if (!cand) return;
    order[rank] = e;
    atomicAdd(ncand, 1);
}

// Greedy suppression of one scene by bit masks (n <= NMS_MASK_MAX boxes per scene): the scene's candidates in visit order go to
// LDS (an ordered compaction of `order`), every candidate i gets the bit row {j later in the order : IoU(i, j) > thr} -- all rows at
// once, the IoU reads independent of each other -- and ONE pass over the rows keeps a candidate iff no kept candidate has
// removed it: the same decisions as the loop of tf_nms3d.cpp:237-262 (a box is dropped iff its IoU with an earlier KEPT box
// exceeds thr, strictly), without a memory round trip per candidate (the wave-per-scene loop took 159 us for 8 x 256 boxes).
constexpr int NMS_MASK_MAX = 512;
__global__ __launch_bounds__(256) void nms_greedy_mask_kernel(int n, float thr, const float *__restrict__ iou, const int *__restrict__ order,
                                                              const int *__restrict__ ncand, int *__restrict__ keep_flag)
{
    extern __shared__ unsigned long long s_mask[]; // n rows x W words, then the list of boxes (n ints), then per-wave counts
    const int W = (n + 63) / 64;
    int *s_list = reinterpret_cast<int *>(s_mask + (size_t)n * W);
    int *s_wcnt = s_list + n;
    __shared__ int s_len;
    const int scene = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nc = *ncand;
    const float *__restrict__ miou = iou + (size_t)scene * n * n;
    if (tid == 0) s_len = 0;
    __syncthreads();
    // (a) this scene's candidates, in visit order
    for (int start = 0; start < nc; start += 256) {
        const int p = start + tid;
        const int e = p < nc ? order[p] : -1;
        const bool mine = e >= 0 && e / n == scene;
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) s_wcnt[w] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < 4; i++) {
            if (i < w) woff += s_wcnt[i];
            tot += s_wcnt[i];
        }
        const int base = s_len;
        if (mine) s_list[base + woff + __popcll(bal & ((1ull << lane) - 1ull))] = e - scene * n;
        __syncthreads();
        if (tid == 0) s_len = base + tot;
        __syncthreads();
    }
    const int L = s_len;
    // (b) suppression rows: bit j of row i = candidate j comes later and overlaps candidate i by more than thr.  A wave per row,
    //     a lane per candidate j of the word: the word is the ballot of the 64 comparisons
    //     The matrix is read as iou[later candidate][earlier box] -- suppress_check(candidate, selected) of tf_nms3d.cpp:250, the
    //     orientation nms_greedy_kernel below uses too: iou3d_pair clips the first box by the second, so the two orientations
    //     of a pair can differ in the last bit, and a pair within an ulp of the threshold must fall the reference's way
    for (int i = w; i < L; i += 4) {
        const int col = s_list[i];
        for (int wd = 0; wd < W; wd++) {
            const int j = wd * 64 + lane;
            const bool hit = j > i && j < L && miou[(size_t)s_list[j < L ? j : 0] * n + col] > thr; // strict >
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_mask[(size_t)i * W + wd] = m;
        }
    }
    __syncthreads();
    // (c) one pass: lane wd of wave 0 owns word wd of the removed set
    if (w == 0) {
        unsigned long long removed = 0ull;
        for (int i = 0; i < L; i++) {
            const unsigned long long cur = __shfl(removed, i >> 6);
            if (!((cur >> (i & 63)) & 1ull)) { // uniform
                if (lane < W) removed |= s_mask[(size_t)i * W + lane];
                if (lane == 0) keep_flag[scene * n + s_list[i]] = 1;
            }
        }
    }
}

// one wave per scene; the scene's kept boxes (visit order) live in LDS
__global__ __launch_bounds__(64) void nms_greedy_kernel(int n, float thr, const float *__restrict__ iou,
                                                        const int *__restrict__ order, const int *__restrict__ ncand,
                                                        int *__restrict__ keep_flag)
{
    extern __shared__ int s_kept[]; // n ints
    const int scene = blockIdx.x;
    const int lane = threadIdx.x;
    const int nc = *ncand;
    const float *__restrict__ miou = iou + (size_t)scene * n * n;
    int nk = 0;
    for (int p = 0; p < nc; p++) {
        const int e = order[p];
        if (e / n != scene) continue; // uniform
        const int box = e - scene * n;
        bool sup = false;
        for (int t = lane; t < nk; t += 64)
            if (miou[(size_t)box * n + s_kept[t]] > thr) sup = true; // tf_nms3d.cpp:250 (strict >)
        if (!__any(sup)) {
            if (lane == 0) {
                s_kept[nk] = box;
                keep_flag[e] = 1;
            }
            nk++;
            __syncthreads(); // single wave: orders the LDS store before the next round's loads
        }
    }
}

// compact kept boxes in global visit order (single block)
__global__ __launch_bounds__(1024) void nms_emit_kernel(int n, const int *__restrict__ order, const int *__restrict__ ncand,
                                                        const int *__restrict__ keep_flag, int *__restrict__ out,
                                                        int *__restrict__ out_count)
{
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int nc = *ncand;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int start = 0; start < nc; start += 1024) {
        const int p = start + tid;
        int e = -1, k = 0;
        if (p < nc) {
            e = order[p];
            k = keep_flag[e];
        }
        const unsigned long long bal = __ballot(k != 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[w] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int i = 0; i < 16; i++) {
            if (i < w) woff += s_wsum[i];
            tot += s_wsum[i];
        }
        const int base = s_base;
        if (k) {
            const int r = base + woff + before;
            out[r * 2 + 0] = e / n;
            out[r * 2 + 1] = e % n;
        }
        __syncthreads();
        if (tid == 0) s_base = base + tot;
        __syncthreads();
    }
    if (tid == 0) *out_count = s_base;
}

struct NmsWorkspace {
    float *iou;
    int *order, *keep_flag, *ncand;
    size_t bytes;
};
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static NmsWorkspace nms_layout(int b, int n, void *base)
{
    NmsWorkspace w;
    char *p = (char *)base;
    size_t off = 0;
    const size_t total = (size_t)b * n;
    w.iou = (float *)(p + off);
    off += align256(total * n * sizeof(float));
    w.order = (int *)(p + off);
    off += align256(total * sizeof(int));
    w.keep_flag = (int *)(p + off);
    off += align256(total * sizeof(int));
    w.ncand = (int *)(p + off);
    off += 256;
    w.bytes = off;
    return w;
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_iou3d_matrix(int b, int n, const float *bboxes, float *iou, void *stream)
{
    VN_REQUIRE(b >= 0 && n >= 0, "3D NMS expects (batch_size, nbbox, 8, 3) bbox shape."); // tf_nms3d.cpp:287
    if (b == 0 || n == 0) return VOTENET_OK;
    VN_REQUIRE(bboxes && iou, "iou3d_matrix: null buffer");
    VN_REQUIRE(n <= 65535 && b <= 65535, "iou3d_matrix: n and b must be <= 65535");
    hipLaunchKernelGGL(iou3d_matrix_kernel, dim3((n + 63) / 64, n, b), dim3(64), 0, as_stream(stream), n, bboxes, iou);
    return check_launch("iou3d_matrix");
}

extern "C" int votenet_iou3d_cross(int b, int n, int m, const float *boxes_a, const float *boxes_b, float *iou, void *stream)
{
    VN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "iou3d_cross expects (batch, n, 8, 3) and (batch, m, 8, 3) boxes");
    if (b == 0 || n == 0 || m == 0) return VOTENET_OK;
    VN_REQUIRE(boxes_a && boxes_b && iou, "iou3d_cross: null buffer");
    VN_REQUIRE(n <= 65535 && b <= 65535, "iou3d_cross: n and b must be <= 65535");
    hipLaunchKernelGGL(iou3d_cross_kernel, dim3((m + 63) / 64, n, b), dim3(64), 0, as_stream(stream), n, m, boxes_a, boxes_b, iou);
    return check_launch("iou3d_cross");
}

extern "C" size_t votenet_nms3d_workspace_bytes(int b, int n)
{
    if (b <= 0 || n <= 0) return 256;
    return nms_layout(b, n, nullptr).bytes;
}

extern "C" int votenet_nms3d(int b, int n, const float *bboxes, const float *scores, const float *objectiveness,
                             float iou_threshold, int *out, int *out_count, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    VN_REQUIRE(b >= 0 && n >= 0, "3D NMS expects (batch_size, nbbox, 8, 3) bbox shape.");          // tf_nms3d.cpp:287
    VN_REQUIRE(iou_threshold >= 0 && iou_threshold <= 1, "iou_threshold must be in [0, 1]");       // :300
    VN_REQUIRE(out_count != nullptr, "3D NMS: out_count is required");
    hipStream_t st = as_stream(stream);
    if (b == 0 || n == 0) {
        (void)hipMemsetAsync(out_count, 0, sizeof(int), st);
        return check_launch("nms3d");
    }
    VN_REQUIRE(bboxes && scores && objectiveness && out, "3D NMS: null buffer");
    VN_REQUIRE(n <= 16384, "3D NMS: at most 16384 boxes per scene");
    if (workspace == nullptr || workspace_bytes < votenet_nms3d_workspace_bytes(b, n))
        return set_error(VOTENET_E_WORKSPACE, "3D NMS: workspace of %zu bytes required", votenet_nms3d_workspace_bytes(b, n));
    NmsWorkspace w = nms_layout(b, n, workspace);
    const int total = b * n;
    int rc = votenet_iou3d_matrix(b, n, bboxes, w.iou, stream);
    if (rc) return rc;
    (void)hipMemsetAsync(w.keep_flag, 0, (size_t)total * sizeof(int), st);
    (void)hipMemsetAsync(w.ncand, 0, sizeof(int), st);
    hipLaunchKernelGGL(nms_rank_kernel, dim3((total + 255) / 256), dim3(256), 0, st, total, scores, objectiveness, w.order,
                       w.ncand);
    if (n <= NMS_MASK_MAX) {
        const size_t smem = (size_t)n * ((n + 63) / 64) * 8 + (size_t)n * 4 + 16;
        hipLaunchKernelGGL(nms_greedy_mask_kernel, dim3(b), dim3(256), smem, st, n, iou_threshold, w.iou, w.order, w.ncand, w.keep_flag);
    } else {
        hipLaunchKernelGGL(nms_greedy_kernel, dim3(b), dim3(64), (size_t)n * sizeof(int), st, n, iou_threshold, w.iou, w.order,
                           w.ncand, w.keep_flag);
    }
    hipLaunchKernelGGL(nms_emit_kernel, dim3(1), dim3(1024), 0, st, n, w.order, w.ncand, w.keep_flag, out, out_count);
    return check_launch("nms3d");
}
