// Probe: which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  For single-bit masks (and a few others) launch many
// workgroups and report the set of (XCC id, SE id, CU id) they saw -- the map from mask bit to hardware CU on this GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ void where(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11));
        unsigned hw = __builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | ((32 - 1) << 11));
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
        // keep the CU busy a little so that other CUs of the mask get work too
        for (volatile int i = 0; i < 2000; i++) {}
    }
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    const int nwg = 4096;
    unsigned *d; hipMalloc(&d, nwg * 8);
    std::vector<unsigned> h(nwg * 2);
    auto run = [&](const char *name, std::vector<uint32_t> mask) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: create failed %d\n", name, (int)e); return; }
        hipMemsetAsync(d, 0xff, nwg * 8, s);
        hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per; // xcc -> set of (se, cu) codes
        for (int i = 0; i < nwg; i++) {
            unsigned hw = h[2 * i + 1];
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7; // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
            per[h[2 * i] & 0xf].insert(se * 100 + sh * 16 + cu);
        }
        printf("%-28s:", name);
        int total = 0;
        for (auto &kv : per) { printf(" xcc%u{", kv.first); for (unsigned c : kv.second) printf("%u.%u ", c / 100, c % 100); printf("}"); total += (int)kv.second.size(); }
        printf("  -> %d CUs\n", total);
        hipStreamDestroy(s);
    };
    for (int b : {0, 1, 2, 7, 8, 9, 16, 31, 32, 33, 63, 64, 128, 255}) {
        std::vector<uint32_t> m(8, 0u);
        m[b / 32] = 1u << (b % 32);
        char nm[64]; snprintf(nm, sizeof nm, "bit %d", b);
        run(nm, m);
    }
    { std::vector<uint32_t> m(8, 0u); m[0] = 0xff; run("bits 0-7", m); }
    { std::vector<uint32_t> m(8, 0xffffffffu); m[0] = 0xffffff00u; run("all but bits 0-7", m); }
    { std::vector<uint32_t> m(8, 0xffffffffu); run("all", m); }
    return 0;
}
