// probe: ds_read_b64_tr_b16 on gfx950 -- (1) which LDS element lands in which lane / register half, (2) the cost of the read for the
// row-major [row][channel] bf16 images a weight-gradient GEMM on split operands would stage (contraction over ROWS: an MFMA fragment
// is 8 consecutive rows of one channel), by row pitch, against a plain ds_read_b64 of the same bytes.
// build: hipcc --offload-arch=gfx950 -O3 tr_b16_probe.hip -o tr_b16_probe ; GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint2 tr_read(unsigned addr)
{
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// (1) semantics: LDS halfword i holds the value i; lane l reads at byte address addr[l]; out[l][0..3] = the four halfwords it gets
__global__ void sem_kernel(const unsigned *addr, unsigned short *out)
{
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    asm volatile("" : : "v"(lds) : "memory"); // the array escapes: its stores are not dead
    __syncthreads();
    const uint2 v = tr_read((unsigned)(uintptr_t)lds + addr[threadIdx.x]);
    out[threadIdx.x * 4 + 0] = v.x & 0xffff;
    out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff;
    out[threadIdx.x * 4 + 3] = v.y >> 16;
}

// (2) cost: every wave of a 256-thread workgroup issues `iters` x 16 reads at lane addresses base[l] + step * (read number);
// cycles per read from s_memtime of wave 0 lane 0.  tr = 1: ds_read_b64_tr_b16, 0: ds_read_b64.
template <int TR>
__global__ __launch_bounds__(256) void cost_kernel(const unsigned *base, unsigned step, int iters, unsigned long long *cyc, unsigned *sink)
{
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const unsigned a0 = base[threadIdx.x & 63];
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint2 v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned a = a0 + step * q;
            if (TR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[q]) : "v"(a) : "memory");
            else asm volatile("ds_read_b64 %0, %1" : "=v"(v[q]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 16; q++) acc ^= v[q].x ^ v[q].y;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

static void run_sem(const char *what, const std::vector<unsigned> &addr)
{
    unsigned *d_a;
    unsigned short *d_o;
    hipMalloc(&d_a, 64 * 4);
    hipMalloc(&d_o, 64 * 4 * 2);
    hipMemcpy(d_a, addr.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, d_a, d_o);
    std::vector<unsigned short> o(256);
    hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
    printf("== semantics: %s (lane: byte address -> halfword indices received)\n", what);
    for (int l = 0; l < 64; l++) {
        printf("  lane %2d addr %5u -> %5u %5u %5u %5u", l, addr[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
        if (l % 2 == 1) printf("\n");
    }
    hipFree(d_a);
    hipFree(d_o);
}

static double run_cost(int tr, const std::vector<unsigned> &base, unsigned step)
{
    unsigned *d_b, *d_s;
    unsigned long long *d_c;
    const int blocks = 256, iters = 2000;
    hipMalloc(&d_b, 64 * 4);
    hipMalloc(&d_s, 4);
    hipMalloc(&d_c, blocks * 8);
    hipMemcpy(d_b, base.data(), 64 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        if (tr) hipLaunchKernelGGL(cost_kernel<1>, dim3(blocks), dim3(256), 65536, 0, d_b, step, iters, d_c, d_s);
        else hipLaunchKernelGGL(cost_kernel<0>, dim3(blocks), dim3(256), 65536, 0, d_b, step, iters, d_c, d_s);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_c, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    hipFree(d_b);
    hipFree(d_s);
    hipFree(d_c);
    return s / blocks / (iters * 16.0); // s_memtime ticks (100 MHz constant clock on this part? reported as is) per read per wave
}

int main()
{
    // (1) semantics with lane l -> 8 * l (contiguous), and with a 4-row x 16-column block per 16-lane group at a row pitch of 64 bytes
    std::vector<unsigned> a(64);
    for (int l = 0; l < 64; l++) a[l] = 8 * l;
    run_sem("lane l reads bytes [8l, 8l+8)", a);
    for (int l = 0; l < 64; l++) {
        const int g = l >> 4, s = l & 15;
        a[l] = (unsigned)(g * 1024 + (s >> 2) * 64 + (s & 3) * 8); // group g: rows 0..3 of its own 1 KB block, pitch 64 B, 4 halfwords per lane
    }
    run_sem("group g = l/16: row s/4 (pitch 64 B) column quad s%4 of block g (1 KB apart)", a);

    // (2) cost.  Fragment read of an MFMA 32x32x16 operand whose k runs over ROWS from a [row][channel] bf16 image of pitch P bytes:
    // group 0: channels 0-15 rows 0-3, group 1: channels 16-31 rows 0-3, group 2: channels 0-15 rows 8-11, group 3: channels 16-31 rows 8-11;
    // successive reads (step) walk 32 channels to the right (64 bytes).
    const unsigned pitches[] = {256, 264, 272, 288, 320, 512 + 16, 64, 32};
    printf("== cost: ticks per read per wave (4 waves per CU, 16 reads per wait); plain = ds_read_b64 at the same addresses\n");
    for (unsigned P : pitches) {
        std::vector<unsigned> b(64);
        for (int l = 0; l < 64; l++) {
            const int g = l >> 4, s = l & 15;
            const int row = (g >> 1) * 8 + (s >> 2), ch = (g & 1) * 16 + (s & 3) * 4;
            b[l] = (unsigned)(row * P + ch * 2);
        }
        const unsigned step = (P >= 256) ? 64 : P * 16; // next 32 channels, or (narrow images) the next 16 rows
        printf("  pitch %4u B: tr %.3f   plain b64 %.3f\n", P, run_cost(1, b, step % 4096), run_cost(0, b, step % 4096));
    }
    // reference: fully contiguous 8 bytes per lane
    {
        std::vector<unsigned> b(64);
        for (int l = 0; l < 64; l++) b[l] = 8 * l;
        printf("  contiguous 8 B / lane: tr %.3f   plain b64 %.3f\n", run_cost(1, b, 512), run_cost(0, b, 512));
    }
    // sub-tiled image [row/4][channel/16][4 rows][16 channels] (128-byte blocks): group g reads block (rowblock, chblock)
    {
        std::vector<unsigned> b(64);
        for (int l = 0; l < 64; l++) {
            const int g = l >> 4, s = l & 15;
            const int rb = (g >> 1) * 2, cb = g & 1;             // row block (rows 0-3 | 8-11), channel block
            b[l] = (unsigned)(((rb * 8 + cb) * 128) + s * 8);    // 8 channel blocks (128 channels) per row block
        }
        printf("  sub-tiled [row/4][ch/16][4][16]: tr %.3f   plain b64 %.3f\n", run_cost(1, b, 256), run_cost(0, b, 256));
    }
    return 0;
}
