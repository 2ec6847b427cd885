// probe (round 5): can the OTHER wave of a SIMD issue vector instructions while a wave's next MFMA waits for the matrix pipe?
// mfma_valu_overlap.hip found that an MFMA-only wave and a VALU-only wave on one SIMD ADD UP (128 + 86 -> 208 cycles per iteration).
// Hypothesis: the MFMA wave's next MFMA is picked by the arbiter and waits AT THE VECTOR ISSUE PORT for the pipe (28 of every 32
// cycles), so nobody else's vector instruction gets through.  If so, an MFMA wave that does not PRESENT its next MFMA early -- it
// executes s_nop (scalar side, no vector port) for most of the 32 cycles -- lets the other wave's vector instructions through.
//   mode bits: 1 MFMA waves run, 2 VALU waves run; pad = cycles of s_nop behind every MFMA; prio: 0 none, 1 VALU waves at s_setprio 3,
//   2 MFMA waves at s_setprio 3
// Workgroups of 512 threads, one per CU: waves 0-3 MFMA, waves 4-7 VALU (wave w and w + 4 share a SIMD).
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_yield.hip -o mfma_valu_yield ; GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int PAD>
__device__ __forceinline__ void pad_nops()
{
    // s_nop N waits N + 1 cycles (N <= 15)
    if constexpr (PAD >= 16) {
        asm volatile("s_nop 15");
        pad_nops<PAD - 16>();
    } else if constexpr (PAD > 0) {
        asm volatile("s_nop %0" ::"n"(PAD - 1));
    }
}
template <int PAD>
__device__ __forceinline__ void mfma_loop(f32x16 (&acc)[4], bf16x8 va, bf16x8 vb, int iters)
{
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            pad_nops<PAD>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__global__ __launch_bounds__(512) void k(int mode, int pad, int prio, int iters, float *out)
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
    if (mf && (mode & 1)) {
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        switch (pad) {
        case 0: mfma_loop<0>(acc, va, vb, iters); break;
        case 8: mfma_loop<8>(acc, va, vb, iters); break;
        case 16: mfma_loop<16>(acc, va, vb, iters); break;
        case 20: mfma_loop<20>(acc, va, vb, iters); break;
        case 24: mfma_loop<24>(acc, va, vb, iters); break;
        case 28: mfma_loop<28>(acc, va, vb, iters); break;
        case 32: mfma_loop<32>(acc, va, vb, iters); break;
        }
    } else if (!mf && (mode & 2)) {
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
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
static float run(int mode, int pad, int prio, int iters, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, pad, prio, 100, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, pad, prio, iters, out);
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
    const float base = run(1, 0, 0, iters, out); // MFMA-only waves, no padding: 4 MFMAs back to back = 128 cycles at the real clock
    const double cyc = 128.0 / base;
    printf("valu only (24 fma per iteration): %.1f cycles per iteration\n", run(2, 0, 0, iters, out) * cyc);
    const int pads[] = {0, 8, 16, 20, 24, 28, 32};
    for (int pad : pads)
        for (int prio = 0; prio < 3; prio++)
            printf("pad %2d prio %d: mfma only %.1f | both on one SIMD %.1f cycles per iteration (4 MFMAs + 24 fma)\n", pad, prio,
                   run(1, pad, prio, iters, out) * cyc, run(3, pad, prio, iters, out) * cyc);
    return 0;
}
