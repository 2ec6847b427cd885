// hold.hip -- a kernel that does nothing but OCCUPY: `grid` workgroups of `block` threads with `lds` bytes each stay resident for
// `cycles` shader clocks (s_sleep in a loop, or a busy ALU loop with spin != 0).  Probe: what costs the GEMMs beside the FPS kernel,
// the CUs it holds or the work it does?
#include <hip/hip_runtime.h>
extern "C" __global__ void hold_kernel(long cycles, int spin, float *sink)
{
    extern __shared__ float smem[];
    const long t0 = __builtin_readcyclecounter();
    float a = threadIdx.x;
    while ((long)__builtin_readcyclecounter() - t0 < cycles) {
        if (spin) {
#pragma unroll
            for (int i = 0; i < 64; i++) a = a * 1.0001f + 0.5f;
            smem[threadIdx.x] = a;
        } else {
            __builtin_amdgcn_s_sleep(127);
        }
    }
    if (a == 12345.678f) sink[0] = a + smem[0];
}
extern "C" int hold_launch(int grid, int block, int lds, long cycles, int spin, float *sink, void *stream)
{
    static int set = 0;
    if (set != lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        set = lds;
    }
    hipLaunchKernelGGL(hold_kernel, dim3(grid), dim3(block), lds, reinterpret_cast<hipStream_t>(stream), cycles, spin, sink);
    return (int)hipGetLastError();
}
