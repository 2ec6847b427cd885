// probe 2: what stops vector instructions from hiding behind MFMAs in the BF3 slab loop?  Two waves per SIMD, every wave runs
// "MFMA, V vector instructions" groups, with one ingredient of the real loop added at a time:
//   dep     the V instructions form ONE dependent chain (as split3 does) instead of V independent ones
//   ldsop   the MFMA operands come from ds_read_b128 (one per MFMA, prefetched one group ahead)
//   ldsw    one ds_write_b64 per group
//   cvt     the chain is the real split: v_cvt_pk_bf16_f32 / shift / and / sub
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_mix2.hip -o mfma_valu_mix2 ; GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int V, int MODE>
__global__ __launch_bounds__(512) void k(int iters, float *out)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[4096 + 2048 * 2 + 64];
    f32x16 acc[4];
    for (int a = 0; a < 4; a++)
        for (int e = 0; e < 16; e++) acc[a][e] = 0.f;
    for (int i = threadIdx.x; i < 4096 + 2048 * 2; i += 512) lds[i] = i * 2654435761u;
    __syncthreads();
    uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(5, threadIdx.x, 7, 8);
    float f[8];
    for (int q = 0; q < 8; q++) f[q] = threadIdx.x * 0.001f + q;
    const float c = 1.0001f, d = 0.5f;
    uint4 nxt = ua;
    const uint4 *rbase = reinterpret_cast<const uint4 *>(&lds[(threadIdx.x & 63) * 4]); // loop-invariant addresses: immediate offsets only
    uint2 *wbase = reinterpret_cast<uint2 *>(&lds[4096 + (threadIdx.x & 255) * 2]);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if (MODE & 2) { // operands from LDS, read one group ahead
                ua = nxt;
                nxt = rbase[a * 64];
            }
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[a], 0, 0, 0);
            if (MODE & 8) { // the real split of two values: 11 instructions, one dependent chain per value
                float x = f[0], y = f[1];
                const bf16x2 hv = {(__bf16)x, (__bf16)y};
                const unsigned h = __builtin_bit_cast(unsigned, hv);
                const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xffff0000u);
                const bf16x2 mv = {(__bf16)rx, (__bf16)ry};
                const unsigned m = __builtin_bit_cast(unsigned, mv);
                f[0] = rx - __uint_as_float(m << 16) + 1.0f;
                f[1] = ry - __uint_as_float(m & 0xffff0000u) + 2.0f;
            } else if (MODE & 1) {
#pragma unroll
                for (int q = 0; q < V; q++) f[0] = __builtin_fmaf(f[0], c, d); // one chain
            } else {
#pragma unroll
                for (int q = 0; q < V; q++) f[q] = __builtin_fmaf(f[q], c, d);
            }
            if (MODE & 4) wbase[a * 256] = make_uint2(__float_as_uint(f[0]), __float_as_uint(f[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; a++) s += acc[a][0] + acc[a][7];
    for (int q = 0; q < 8; q++) s += f[q];
    if (s == 12345.678f) out[0] = s + lds[threadIdx.x];
}
template <int V, int MODE>
static float run(int iters, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, MODE>), dim3(256), dim3(512), 0, 0, 100, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, MODE>), dim3(256), dim3(512), 0, 0, iters, out);
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
    const float base = run<0, 0>(iters, out);
    const double cyc = 8 * 32.0 / base / 8; // two waves x 4 MFMAs back to back = 8 x 32 cycles of the SIMD per iteration
    printf("cycles of the SIMD per MFMA, two waves per SIMD (bare MFMAs = 32 by calibration)\n");
    printf("V=4 independent            %.1f\n", run<4, 0>(iters, out) * cyc);
    printf("V=4 one dependent chain    %.1f\n", run<4, 1>(iters, out) * cyc);
    printf("V=2 one dependent chain    %.1f\n", run<2, 1>(iters, out) * cyc);
    printf("V=4 indep + LDS operands   %.1f\n", run<4, 2>(iters, out) * cyc);
    printf("V=0 + LDS operands         %.1f\n", run<0, 2>(iters, out) * cyc);
    printf("V=4 indep + ds_write_b64   %.1f\n", run<4, 4>(iters, out) * cyc);
    printf("V=4 indep + both LDS       %.1f\n", run<4, 6>(iters, out) * cyc);
    printf("real split of 2 values     %.1f   (11 instructions)\n", run<0, 8>(iters, out) * cyc);
    printf("real split + both LDS      %.1f\n", run<0, 14>(iters, out) * cyc);
    return 0;
}
