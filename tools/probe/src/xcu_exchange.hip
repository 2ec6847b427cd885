// probe: what would a farthest-point-sampling round cost if ONE scene were split over TWO compute units?  Every round the two halves
// must agree on the block-wide arg-max before the next distance update can start: each publishes its local best into a shared word
// (64-bit atomic max in L2: running distance | tie key), announces it, waits for the other and reads the winner -- the minimal
// exchange, with nothing else on the GPU.  Reported: microseconds per round for partner workgroups on the same XCD (blocks b, b + 8)
// and on different XCDs (blocks b, b + 1), 1 and 8 pairs at a time (8 scenes).  A round of the one-CU kernel is 0.82 us in all
// (profiles/r02_fps_round_trace.txt), of which the in-CU exchange is 0.07.
// build: hipcc --offload-arch=gfx950 -O3 xcu_exchange.hip -o xcu_exchange ; GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void exch(unsigned long long *slots, unsigned *cnt, int rounds, int stride, int pairs, unsigned long long *sink)
{
    // pair p = blocks (p, p + stride) for same-XCD placement (stride 8) or (2p, 2p + 1) when stride == 1
    int pair, side;
    if (stride == 1) {
        pair = blockIdx.x >> 1;
        side = blockIdx.x & 1;
    } else {
        pair = blockIdx.x % stride;
        side = blockIdx.x / stride;
        if (side > 1) return;
    }
    if (pair >= pairs) return;
    unsigned long long *my = slots + (size_t)pair * rounds;
    unsigned *mc = cnt + (size_t)pair * rounds;
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; r++) {
        if (threadIdx.x == 0) {
            const unsigned long long mine = ((unsigned long long)(unsigned)(r * 2654435761u + side * 97u) << 32) | (unsigned)(side + 1);
            atomicMax(&my[r], mine);                                                              // publish (device scope, L2)
            __hip_atomic_fetch_add(&mc[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // announce
            while (__hip_atomic_load(&mc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) {  // wait for the partner
            }
            acc ^= __hip_atomic_load(&my[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // the winner
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && acc == 42ull) sink[0] = acc;
}

static double run(int stride, int pairs, int rounds)
{
    unsigned long long *slots, *sink;
    unsigned *cnt;
    hipMalloc(&slots, sizeof(unsigned long long) * pairs * rounds);
    hipMalloc(&cnt, sizeof(unsigned) * pairs * rounds);
    hipMalloc(&sink, 8);
    const int grid = stride == 1 ? 2 * pairs : 2 * stride;
    double best = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(slots, 0, sizeof(unsigned long long) * pairs * rounds);
        hipMemset(cnt, 0, sizeof(unsigned) * pairs * rounds);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(exch, dim3(grid), dim3(64), 0, 0, slots, cnt, rounds, stride, pairs, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    hipFree(slots);
    hipFree(cnt);
    hipFree(sink);
    return best * 1e3 / rounds; // us per round
}

int main()
{
    const int rounds = 2047;
    printf("two-CU exchange per FPS round (2047 rounds, 64-bit atomic max + arrival counter + poll, nothing else on the GPU):\n");
    printf("  same XCD (blocks b, b+8), 1 pair : %.3f us per round\n", run(8, 1, rounds));
    printf("  same XCD (blocks b, b+8), 8 pairs: %.3f us per round\n", run(8, 8, rounds));
    printf("  cross XCD (blocks 2p, 2p+1), 1 pair : %.3f us per round\n", run(1, 1, rounds));
    printf("  cross XCD (blocks 2p, 2p+1), 8 pairs: %.3f us per round\n", run(1, 8, rounds));
    return 0;
}
