// variants.hip -- the tf_ops the reference ships but VoteNet's model never reaches: SelectionSort /
// kNN (tf_grouping_g.cu:83-123, tf_grouping.py:47-73) and ProbSample (tf_sampling_g.cu:7-104).
//
// SelectionSort in the reference: one THREAD per row walks n elements k times in global memory.  Here one WAVEFRONT owns a
// row held in LDS; a selection step is a strided scan (each lane keeps the first minimum of its positions) and one 64-bit
// min across the wave on the key (ordered value | position), which is the sequential "first position of the minimum".
// kNN fuses the distance row into the same kernel.  ProbSample keeps the reference's float summation tree (see the
// header) but walks it without the down-sweep: after the up-sweep a prefix is at most log2 additions of block sums.
#include "common.h"

namespace votenet {

constexpr int SEL_LDS_MAX = 16384; // rows of up to this many elements live in LDS (8 bytes per element: 128 KB of the CU's 160)
constexpr int SEL_WIDE = 2048;     // longer rows get four wavefronts

__device__ __forceinline__ unsigned long long sel_key(float v, int pos)
{
    unsigned u = __float_as_uint(v);
    if (v == 0.0f) u = 0u;                      // -0 == +0 for '<'
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u); // monotone in the float order
    if (v != v) u = 0xFFFFFFFFu;                // NaN is never '<' anything
    return ((unsigned long long)u << 32) | (unsigned)pos;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)k, off), hi = __shfl_xor((unsigned)(k >> 32), off);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        k = o < k ? o : k;
    }
    return k;
}

// k selection steps on the row (v, id) of n elements by the NW wavefronts of the workgroup
template <bool GLOBAL, int NW>
__device__ __forceinline__ void select_steps(float *v, int *id, int n, int k)
{
    __shared__ unsigned long long slot[2]; // cross-wave minimum, double-buffered by step parity
    const int tid = threadIdx.x;
    if (NW > 1) {
        if (tid < 2) slot[tid] = ~0ull;
        __syncthreads();
    }
    for (int s = 0; s < k; s++) {
        unsigned long long best = ~0ull;
        for (int t = s + tid; t < n; t += 64 * NW) {
            const unsigned long long key = sel_key(v[t], t);
            best = key < best ? key : best;
        }
        best = wave_min_u64(best);
        if (NW > 1) {
            if ((tid & 63) == 0) atomicMin(&slot[s & 1], best);
            __syncthreads();
            best = slot[s & 1];
        }
        const float vs = v[s];
        const int pos = (vs != vs) ? s : (int)(unsigned)best;
        if (NW == 1) __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
            if (pos != s) {
                const float tv = v[pos];
                v[pos] = vs;
                v[s] = tv;
                const int ti = id[pos];
                id[pos] = id[s];
                id[s] = ti;
            }
            if (NW > 1) slot[(s + 1) & 1] = ~0ull;
        }
        if (GLOBAL) __threadfence_block();
        if (NW > 1)
            __syncthreads();
        else
            __builtin_amdgcn_wave_barrier();
    }
}

template <bool LDS, int NW>
__global__ __launch_bounds__(64 * NW) void selection_sort_kernel(long rows, int n, int k, const float *__restrict__ dist, int *outi,
                                                                 float *out)
{
    extern __shared__ unsigned char sel_smem[];
    const int tid = threadIdx.x;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float *d = dist + row * n;
        float *v = LDS ? reinterpret_cast<float *>(sel_smem) : out + row * n;
        int *id = LDS ? reinterpret_cast<int *>(sel_smem) + n : outi + row * n;
        for (int t = tid; t < n; t += 64 * NW) {
            v[t] = d[t];
            id[t] = t;
        }
        if (!LDS) __threadfence_block();
        __syncthreads();
        select_steps<!LDS, NW>(v, id, n, k);
        __syncthreads();
        if (LDS)
            for (int t = tid; t < n; t += 64 * NW) {
                out[row * n + t] = v[t];
                outi[row * n + t] = id[t];
            }
        __syncthreads();
    }
}

template <bool LDS, int NW>
__global__ __launch_bounds__(64 * NW) void knn_point_kernel(int n, int m, int c, int k, long rows, const float *__restrict__ xyz1,
                                                            const float *__restrict__ xyz2, float *__restrict__ val,
                                                            int *__restrict__ idx, unsigned char *work)
{
    extern __shared__ unsigned char sel_smem[];
    const int tid = threadIdx.x;
    unsigned char *base = LDS ? sel_smem : work + (size_t)blockIdx.x * n * 8;
    float *v = reinterpret_cast<float *>(base);
    int *id = reinterpret_cast<int *>(base) + n;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long bi = row / m;
        const float *q = xyz2 + row * c;
        const float *p = xyz1 + bi * n * c;
        for (int t = tid; t < n; t += 64 * NW) { // tf_grouping.py:61-63: reduce_sum((xyz1 - xyz2)**2, -1), left to right
            float s = 0.0f;
            for (int a = 0; a < c; a++) {
                const float d = p[(long)t * c + a] - q[a];
                const float sq = d * d;
                s = a == 0 ? sq : s + sq;
            }
            v[t] = s;
            id[t] = t;
        }
        if (!LDS) __threadfence_block();
        __syncthreads();
        select_steps<!LDS, NW>(v, id, n, k);
        __syncthreads();
        for (int t = tid; t < k; t += 64 * NW) { // tf.slice(..., [-1,-1,k]), tf_grouping.py:66-67
            val[row * k + t] = v[t];
            idx[row * k + t] = id[t];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- ProbSample
constexpr int CS_T = 1024;     // threads
constexpr int CS_CHUNK = 8192; // elements per scan (tf_sampling_g.cu:8,14: BlockSize*4)

// inclusive sum of the first q >= 1 groups from the up-swept block sums: the top power-of-two block, then each lower
// block added onto the accumulated prefix -- the association the reference's down-sweep produces
__device__ __forceinline__ float prefix_groups(const float *B, int q)
{
    float acc = 0.0f;
    bool first = true;
    int pos = 0;
    for (int bit = 31 - __clz(q); bit >= 0; bit--)
        if ((q >> bit) & 1) {
            pos += 1 << bit;
            const float t = B[pos - 1];
            acc = first ? t : t + acc;
            first = false;
        }
    return acc;
}

__global__ __launch_bounds__(CS_T) void cumsum_kernel(int n, const float *__restrict__ inp, float *__restrict__ out)
{
    __shared__ float P[CS_CHUNK];
    __shared__ float B[CS_CHUNK / 4];
    const int tid = threadIdx.x;
    const float *x0 = inp + (long)blockIdx.x * n;
    float *o0 = out + (long)blockIdx.x * n;
    float running = 0.0f, comp = 0.0f;
    for (int j = 0; j < n; j += CS_CHUNK) {
        const int len = n - j < CS_CHUNK ? n - j : CS_CHUNK;
        const int ng = (len + 3) >> 2;
        const float *x = x0 + j;
        for (int g = tid; g < ng; g += CS_T) { // in-group prefixes, tf_sampling_g.cu:19-42
            const int e = g * 4;
            if (e + 3 < len) {
                const float v1 = x[e], v2 = x[e + 1], v3 = x[e + 2], v4 = x[e + 3];
                const float p2 = v2 + v1;
                const float p3 = v3 + p2;
                const float p4 = (v4 + v3) + p2;
                P[e] = v1;
                P[e + 1] = p2;
                P[e + 2] = p3;
                P[e + 3] = p4;
                B[g] = p4;
            } else {
                float v = 0.0f;
                for (int t = e; t < len; t++) {
                    v += x[t];
                    P[t] = v;
                }
                B[g] = v;
            }
        }
        for (int u = 0; (2 << u) <= ng; u++) { // block sums: T(block) = T(left) + T(right), :45-55
            __syncthreads();
            for (int kk = tid; kk < (ng >> (u + 1)); kk += CS_T) {
                const int i1 = ((2 * kk + 2) << u) - 1, i2 = ((2 * kk + 1) << u) - 1;
                B[i1] = B[i1] + B[i2];
            }
        }
        __syncthreads();
        for (int t = tid; t < len; t += CS_T) {
            const int g = t >> 2;
            const float pre = g == 0 ? P[t] : P[t] + prefix_groups(B, g);
            o0[j + t] = pre + running;
        }
        const float total = prefix_groups(B, ng) + comp; // compensated carry, :78-84
        const float r2 = running + total;
        comp = total - (r2 - running);
        running = r2;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void prob_search_kernel(int n, int m, const float *__restrict__ cums,
                                                          const float *__restrict__ query, int *__restrict__ result)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const float *d = cums + (long)blockIdx.y * n;
    int base = 1;
    while (base < n) base <<= 1;
    const float q = query[(long)blockIdx.y * m + j] * d[n - 1]; // tf_sampling_g.cu:94
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (r >= k && d[r - k] >= q) r -= k;
    result[(long)blockIdx.y * m + j] = r;
}

static int sel_grid(long rows) { return (int)(rows < 8192 ? rows : 8192); }
constexpr long KNN_WORK_ROWS = 8192;

} // namespace votenet

using namespace votenet;

template <bool LDS, int NW>
static void launch_selection_sort(long rows, int n, int k, const float *dist, int *outi, float *out, hipStream_t st)
{
    const size_t smem = LDS ? (size_t)n * 8 : 0;
    if (LDS)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&selection_sort_kernel<LDS, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((selection_sort_kernel<LDS, NW>), dim3(sel_grid(rows)), dim3(64 * NW), smem, st, rows, n, k, dist, outi, out);
}

extern "C" int votenet_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream)
{
    VN_REQUIRE(b > 0 && n > 0 && m > 0, "SelectionSort expects a non-empty (b,m,n) distance matrix, got (%d,%d,%d)", b, m, n);
    VN_REQUIRE(k > 0 && k <= n, "SelectionSort expects 0 < k <= n, got k = %d, n = %d", k, n);
    VN_REQUIRE(dist && outi && out, "SelectionSort: null pointer");
    const long rows = (long)b * m;
    if (n <= SEL_WIDE)
        launch_selection_sort<true, 1>(rows, n, k, dist, outi, out, as_stream(stream));
    else if (n <= SEL_LDS_MAX)
        launch_selection_sort<true, 4>(rows, n, k, dist, outi, out, as_stream(stream));
    else
        launch_selection_sort<false, 4>(rows, n, k, dist, outi, out, as_stream(stream));
    return check_launch("selection_sort");
}

extern "C" size_t votenet_knn_workspace_bytes(int b, int n, int m)
{
    if (n <= SEL_LDS_MAX) return 0;
    const long rows = (long)b * m;
    return (size_t)(rows < KNN_WORK_ROWS ? rows : KNN_WORK_ROWS) * (size_t)n * 8;
}

extern "C" int votenet_knn_point(int b, int n, int m, int c, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                                 void *workspace, void *stream)
{
    VN_REQUIRE(b > 0 && n > 0 && m > 0 && c > 0, "knn_point expects (b,n,c) and (b,m,c) clouds, got b=%d n=%d m=%d c=%d", b, n, m, c);
    VN_REQUIRE(k > 0 && k <= n, "knn_point expects 0 < k <= n, got k = %d, n = %d", k, n);
    VN_REQUIRE(xyz1 && xyz2 && val && idx, "knn_point: null pointer");
    const long rows = (long)b * m;
    if (n <= SEL_LDS_MAX) {
        const size_t smem = (size_t)n * 8;
        if (n <= SEL_WIDE) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_point_kernel<true, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL((knn_point_kernel<true, 1>), dim3(sel_grid(rows)), dim3(64), smem, as_stream(stream), n, m, c, k, rows,
                               xyz1, xyz2, val, idx, (unsigned char *)nullptr);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_point_kernel<true, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL((knn_point_kernel<true, 4>), dim3(sel_grid(rows)), dim3(256), smem, as_stream(stream), n, m, c, k, rows,
                               xyz1, xyz2, val, idx, (unsigned char *)nullptr);
        }
    } else {
        VN_REQUIRE(workspace, "knn_point: n = %d needs a workspace of votenet_knn_workspace_bytes() bytes", n);
        const int grid = (int)(rows < KNN_WORK_ROWS ? rows : KNN_WORK_ROWS);
        hipLaunchKernelGGL((knn_point_kernel<false, 4>), dim3(grid), dim3(256), 0, as_stream(stream), n, m, c, k, rows, xyz1, xyz2,
                           val, idx, (unsigned char *)workspace);
    }
    return check_launch("knn_point");
}

extern "C" int votenet_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                                   void *stream)
{
    VN_REQUIRE(b > 0 && n > 0 && m > 0, "ProbSample expects (batch_size,num_choices) and (batch_size,num_points) inputs, got b=%d n=%d m=%d",
               b, n, m);
    VN_REQUIRE(inp_p && inp_r && temp && out, "ProbSample: null pointer");
    hipLaunchKernelGGL(cumsum_kernel, dim3(b), dim3(CS_T), 0, as_stream(stream), n, inp_p, temp);
    hipLaunchKernelGGL(prob_search_kernel, dim3((m + 255) / 256, b), dim3(256), 0, as_stream(stream), n, m, temp, inp_r, out);
    return check_launch("prob_sample");
}

// ---- the reference's own launcher names, C++ linkage, exact signatures (tf_sampling.cpp:65 called :89;
// tf_grouping.cpp:108 called :134), so that tf_sampling.cpp / tf_grouping.cpp link against this library
// unchanged.  Null stream, as the reference; `temp` is the caller's (b,n) float scratch of tf_sampling.cpp:86.
VN_EXPORT void probsampleLauncher(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out)
{
    votenet_prob_sample(b, n, m, inp_p, inp_r, temp, out, nullptr);
}
VN_EXPORT void selectionSortLauncher(int b, int n, int m, int k, const float *dist, int *outi, float *out)
{
    votenet_selection_sort(b, n, m, k, dist, outi, out, nullptr);
}
