// Probe: throughput of float atomic adds to an 8 MB table (the shape of GroupPointGrad's scatter at sa2: 16384 points x 128 channels)
// at agent scope (what unsafeAtomicAdd issues) and at workgroup scope (served by the XCD's own L2; NOT coherent across XCDs -- timing only).
// A wavefront adds 64 consecutive floats of one random row, as the scatter kernels do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE>
__global__ __launch_bounds__(256) void k(float *tab, const int *rows, long nrow_adds, int c)
{
    const int ch = threadIdx.x % c, rl = threadIdx.x / c, rpb = 256 / c;
    for (long i = (long)blockIdx.x * rpb + rl; i < nrow_adds; i += (long)gridDim.x * rpb) {
        float *p = tab + (size_t)rows[i] * c + ch;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 3) { // half of the lanes (pseudo-random) have nothing to add: does a wavefront's atomic cost its ACTIVE lanes?
            unsigned hsh = (unsigned)(i * 131 + ch) * 2654435761u;
            if (hsh & 0x10000u) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (SCOPE == 4) { // a quarter of the lanes
            unsigned hsh = (unsigned)(i * 131 + ch) * 2654435761u;
            if ((hsh & 0x30000u) == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else *p = 1.0f; // plain store: the traffic without the read-modify-write
    }
}
int main()
{
    const int c = 128, npts = 16384;
    const long adds = 131072; // row-adds (x 128 channels = 16.8 M float atomics, sa2's count)
    float *tab; int *rows;
    hipMalloc(&tab, (size_t)npts * c * 4); hipMalloc(&rows, adds * 4);
    std::vector<int> h(adds);
    unsigned s = 12345;
    for (long i = 0; i < adds; i++) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % npts; }
    hipMemcpy(rows, h.data(), adds * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int scope = 0; scope < 5; scope++)
        for (int grid : {512, 2048, 8192}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipMemset(tab, 0, (size_t)npts * c * 4);
                hipEventRecord(e0);
                if (scope == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, tab, rows, adds, c);
                else if (scope == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, tab, rows, adds, c);
                else if (scope == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, tab, rows, adds, c);
                else if (scope == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, tab, rows, adds, c);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, tab, rows, adds, c);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s grid %5d: %.1f us  (%.1f G float-adds/s)\n", scope == 0 ? "agent    " : scope == 1 ? "workgroup" : scope == 2 ? "store    " : scope == 3 ? "agent 1/2" : "agent 1/4", grid,
                   best * 1e3, adds * c / best / 1e6);
        }
    return 0;
}
