// probe (round 5): what does handing work to a second stream cost the MAIN queue, by mechanism?  A chain of small dependent kernels
// on stream A; after every kernel a second stream B is told "A has got this far" and runs a small kernel of its own:
//   0  nothing between the links (the chain alone)
//   1  hipEventRecord(A) + hipStreamWaitEvent(B) + kernel on B            (what pointnet2._hand_over does, torch events)
//   2  the same with events created hipEventDisableTiming
//   3  hipStreamWriteValue32(A, flag, k) + hipStreamWaitValue32(B, flag, k, >=) + kernel on B   (signal memory, no event)
//   4  no synchronisation at all, kernel on B every link                  (what B's own work costs A: contention only)
// Reported: GPU microseconds per link of chain A (HIP events around the whole chain), host microseconds per link.
// build: hipcc --offload-arch=gfx950 -O3 handover_cost.hip -o handover_cost ; GPU box only.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

__global__ void link(float *x, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += 1.0f;
}

int main(int argc, char **argv)
{
    const int links = argc > 1 ? atoi(argv[1]) : 400, n = argc > 2 ? atoi(argv[2]) : (1 << 23); // 2^23 floats: a ~20 us kernel, the GPU chain (not the host) is what is timed
    float *x, *y;
    CHECK(hipMalloc(&x, n * 4));
    CHECK(hipMalloc(&y, n * 4));
    CHECK(hipMemset(x, 0, n * 4));
    CHECK(hipMemset(y, 0, n * 4));
    unsigned *flag = nullptr;
    const bool have_signal = hipExtMallocWithFlags(reinterpret_cast<void **>(&flag), 64, hipMallocSignalMemory) == hipSuccess;
    if (!have_signal) {
        (void)hipGetLastError();
        CHECK(hipMalloc(&flag, 64));
    }
    CHECK(hipMemset(flag, 0, 64));
    hipStream_t A, B;
    CHECK(hipStreamCreate(&A));
    CHECK(hipStreamCreate(&B));
    const int NE = 64;
    hipEvent_t evt[NE], evn[NE], e0, e1;
    for (int i = 0; i < NE; i++) {
        CHECK(hipEventCreate(&evt[i]));
        CHECK(hipEventCreateWithFlags(&evn[i], hipEventDisableTiming));
    }
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    static const char *names[] = {"chain alone", "event (timing) + wait + side kernel", "event (no timing) + wait + side kernel",
                                  "write-value + wait-value + side kernel", "no synchronisation, side kernel"};
    unsigned epoch = 0;
    for (int mode = 0; mode < 5; mode++) {
        if (mode == 3 && !have_signal) printf("(signal memory not available: mode 3 on ordinary device memory)\n");
        double best_gpu = 1e30, best_host = 1e30;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            CHECK(hipEventRecord(e0, A));
            for (int k = 0; k < links; k++) {
                hipLaunchKernelGGL(link, dim3(n / 256), dim3(256), 0, A, x, n);
                if (mode == 1 || mode == 2) {
                    hipEvent_t ev = (mode == 1 ? evt : evn)[k % NE];
                    CHECK(hipEventRecord(ev, A));
                    CHECK(hipStreamWaitEvent(B, ev, 0));
                } else if (mode == 3) {
                    ++epoch;
                    CHECK(hipStreamWriteValue32(A, flag, epoch, 0));
                    CHECK(hipStreamWaitValue32(B, flag, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
                }
                if (mode >= 1) hipLaunchKernelGGL(link, dim3(n / 256), dim3(256), 0, B, y, n);
            }
            CHECK(hipEventRecord(e1, A));
            const auto t1 = std::chrono::steady_clock::now();
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double g = ms * 1e3 / links, h = std::chrono::duration<double, std::micro>(t1 - t0).count() / links;
            if (rep > 0 && g < best_gpu) best_gpu = g;
            if (rep > 0 && h < best_host) best_host = h;
        }
        printf("%-44s chain A: %.2f us per link on the GPU, host %.2f us per link\n", names[mode], best_gpu, best_host);
    }
    return 0;
}
