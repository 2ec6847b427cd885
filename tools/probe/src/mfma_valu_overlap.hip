// probe: how many vector instructions does an MFMA hide, in its own wave and in the other wave of its SIMD?
// Workgroups of 256 (one wave per SIMD) or 512 threads (two), one per CU.
//   split: waves 0-3 issue v_mfma_f32_32x32x16_bf16 back to back (4 independent accumulators), waves 4-7 independent v_fma_f32
//          chains -- alone (mfma only / valu only) and together
//   mix V: every wave alternates 1 MFMA : V independent v_fma_f32, with one or two waves per SIMD
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap ; GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int V>
__device__ __forceinline__ void mix(f32x16 (&acc)[4], float (&f)[8], bf16x8 va, bf16x8 vb, int iters)
{
    const float c = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[a], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < V; q++) f[q] = __builtin_fmaf(f[q], c, d);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out)
{
    const int w = threadIdx.x >> 6;
    const bool mf = w < 4;
    f32x16 acc[4];
    for (int a = 0; a < 4; a++)
        for (int e = 0; e < 16; e++) acc[a][e] = 0.f;
    uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(5, threadIdx.x, 7, 8);
    bf16x8 va = __builtin_bit_cast(bf16x8, ua), vb = __builtin_bit_cast(bf16x8, ub);
    float f[8];
    for (int q = 0; q < 8; q++) f[q] = threadIdx.x * 0.001f + q;
    const float c = 1.0001f, d = 0.5f;
    if (mode >= 10) {
        switch (mode - 10) {
        case 0: mix<0>(acc, f, va, vb, iters); break;
        case 2: mix<2>(acc, f, va, vb, iters); break;
        case 3: mix<3>(acc, f, va, vb, iters); break;
        case 4: mix<4>(acc, f, va, vb, iters); break;
        case 5: mix<5>(acc, f, va, vb, iters); break;
        case 6: mix<6>(acc, f, va, vb, iters); break;
        case 8: mix<8>(acc, f, va, vb, iters); break;
        }
    } else if (mf && (mode & 1)) {
        if (mode & 4) __builtin_amdgcn_s_setprio(3); // split + priority: the MFMA waves above the VALU waves
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int a = 0; a < 4; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[a], 0, 0, 0);
        }
    } else if (!mf && (mode & 2)) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int q = 0; q < 8; q++) f[q] = __builtin_fmaf(f[q], c, d);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; a++) s += acc[a][0] + acc[a][7];
    for (int q = 0; q < 8; q++) s += f[q];
    if (s == 12345.678f) out[0] = s;
}
static float run(int mode, int threads, int iters, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, mode, 100, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, mode, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 20000;
    const float base = run(1, 512, iters, out); // MFMA-only waves: 4 MFMAs per iteration back to back = 128 cycles at the real clock
    const double cyc = 128.0 / base;            // cycles per ms-unit, calibrated on that
    printf("split  mfma only %.1f cyc/iter | valu only (24 fma) %.1f | both on one SIMD %.1f\n", base * cyc, run(2, 512, iters, out) * cyc,
           run(3, 512, iters, out) * cyc);
    printf("split, MFMA waves at s_setprio 3: both on one SIMD %.1f cyc/iter\n", run(7, 512, iters, out) * cyc);
    const int vs[] = {0, 2, 3, 4, 5, 6, 8};
    for (int v : vs) {
        const float t1 = run(10 + v, 256, iters, out), t2 = run(10 + v, 512, iters, out);
        printf("mix 1 MFMA : %d VALU   one wave per SIMD %.1f cycles per MFMA | two waves per SIMD %.1f cycles per MFMA of the SIMD\n", v,
               t1 * cyc / 4, t2 * cyc / 8);
    }
    return 0;
}
