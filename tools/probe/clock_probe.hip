// Micro-probe: effective shader clock and dependent-VALU latency when only a few CUs are busy.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void dep_chain(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  long long t0 = __builtin_readcyclecounter();
  long long m0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) a = a * b + 0.5f;   // 16 dependent FMAs (2 flops) per iteration
  }
  long long t1 = __builtin_readcyclecounter();
  long long m1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = m1 - m0; }
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 2048 * 1024 * 4); hipMalloc(&cyc, 2048 * 16);
  for (int blocks : {1, 8, 256, 2048}) for (int threads : {64, 256, 1024}) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dep_chain<<<blocks, threads>>>(out, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    dep_chain<<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    double instr = (double)iters * 16;
    printf("blocks=%4d threads=%4d: %.3f ms, shader-cycles %lld (%.2f cyc/dep-instr), wallclk ticks %lld -> shader clock %.0f MHz (if wall_clock=100MHz), ns/instr %.2f\n",
           blocks, threads, ms, h[0], h[0] / instr, h[1], h[0] / (h[1] / 100.0), ms * 1e6 / instr);
  }
  return 0;
}
