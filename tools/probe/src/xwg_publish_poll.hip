// probe (round 5): the exchange of fps_bucket_split_kernel in isolation.  W workgroups per group; every round each publishes five 64-bit
// words [round | payload] with relaxed device-scope stores and every one of its NWV wavefronts polls the 5 W words of the round until
// all carry the round's number.  Reported: microseconds per round -- for W = 2, 4; 1 / 12 polling wavefronts per workgroup; partners
// on one XCD (blocks x, x + 8, ...) or on different XCDs (blocks 4 g .. 4 g + 3); 1 or 4 groups at a time; with s_sleep between polls.
// build: hipcc --offload-arch=gfx950 -O3 xwg_publish_poll.hip -o xwg_publish_poll ; GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int W>
__global__ __launch_bounds__(768) void xch(unsigned long long *slots, int rounds, int same_xcd, int groups, int sleep, int scope_sys,
                                           unsigned long long *sink)
{
    int grp, q;
    if (same_xcd) {
        const int xcd = blockIdx.x & 7, turn = blockIdx.x >> 3;
        grp = xcd + 8 * (turn / W);
        q = turn % W;
    } else {
        grp = blockIdx.x / W;
        q = blockIdx.x % W;
    }
    if (grp >= groups) return;
    unsigned long long *xs = slots + (size_t)grp * 2 * W * 5;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long acc = 0;
    for (int j = 1; j <= rounds; j++) {
        unsigned long long *cur = xs + (size_t)(j & 1) * W * 5;
        if (w == 0 && lane < 5) {
            const unsigned long long v = ((unsigned long long)(unsigned)j << 32) | (unsigned)(q * 131 + lane + j);
            if (scope_sys) __hip_atomic_store(&cur[q * 5 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else __hip_atomic_store(&cur[q * 5 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long got = 0;
        for (int spin = 0; spin < (1 << 22); spin++) {
            if (lane < 5 * W) {
                if (scope_sys) got = __hip_atomic_load(&cur[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else got = __hip_atomic_load(&cur[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (__ballot(lane < 5 * W && (unsigned)(got >> 32) != (unsigned)j) == 0ull) break;
            if (sleep) __builtin_amdgcn_s_sleep(1);
        }
        acc += got;
        // the in-CU barrier of the real kernel (its waves stay in step)
        __syncthreads();
    }
    if (acc == 42ull) sink[0] = acc;
}

template <int W>
static double run(int nwv, int same_xcd, int groups, int sleep, int scope_sys, int rounds)
{
    unsigned long long *slots, *sink;
    hipMalloc(&slots, sizeof(unsigned long long) * 64 * 2 * W * 5);
    hipMalloc(&sink, 8);
    const int grid = same_xcd ? 8 * W * ((groups + 7) / 8) : groups * W;
    double best = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(slots, 0, sizeof(unsigned long long) * 64 * 2 * W * 5);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(xch<W>, dim3(grid), dim3(nwv * 64), 0, 0, slots, rounds, same_xcd, groups, sleep, scope_sys, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipFree(slots);
    hipFree(sink);
    return best * 1e3 / rounds;
}

int main()
{
    const int rounds = 2047;
    printf("publish + poll exchange, us per round (2047 rounds, nothing else on the GPU)\n");
    for (int nwv : {1, 12})
        for (int same : {1, 0})
            for (int groups : {1, 4}) {
                printf("  W=2 %2d polling waves, %s, %d group(s): %.3f   W=4: %.3f   W=4 with s_sleep: %.3f   W=4 system scope: %.3f\n", nwv,
                       same ? "one XCD      " : "different XCDs", groups, run<2>(nwv, same, groups, 0, 0, rounds), run<4>(nwv, same, groups, 0, 0, rounds),
                       run<4>(nwv, same, groups, 1, 0, rounds), run<4>(nwv, same, groups, 0, 1, rounds));
            }
    return 0;
}
