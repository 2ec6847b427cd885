// probe (round 5): does v_pk_fma_f32 return a wrong LOW half when the low half takes its multiplier from the HIGH register of SRC1
// (op_sel:[0,1,0]) -- sporadically, for the last 16 lanes, when the GPU is shared between processes?
// Found in pool_dgrad_scatter_wave_kernel: with the broadcast operand second (src1 = [v_t, v_t+1] + op_sel) rows came out with ONE list
// entry's contribution missing in lanes 48-63 of the low register (a few hundred rows of 131072 per launch, 10-25 % of the launches,
// only with two other processes on the GPU); with the operands swapped (broadcast operand first) never (tools/probe/scatter_repeat.py).
// This kernel repeats the instruction pattern in isolation and checks itself: every trip reads four list entries (value pairs by
// ds_read2_b32, rows of a table in LDS by ds_read_b64) and accumulates them TWICE from the same registers -- once with the broadcast
// operand in src1, once in src0 -- into two accumulators that must stay bit-equal.
//   build: hipcc --offload-arch=gfx950 -O3 pk_opsel_hazard.hip -o pk_opsel_hazard      run: ./pk_opsel_hazard [launches] [side]
//   side = 1: a second stream runs MFMA-only workgroups beside it; side = 2: vector-only workgroups.  Run three at once to share the GPU (tools/probe/pk3.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int NC = 256, CIN = 128, NWV = 8, LCAP = 448;

__global__ __launch_bounds__(NWV * 64) void pk_kernel(const float *__restrict__ table, const float *__restrict__ vals,
                                                      const unsigned short *__restrict__ codes, int trips, int rounds,
                                                      unsigned *__restrict__ bad, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;                                                                 // [NC][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float *lv = Wl + NC * CIN + wv * (LCAP + LCAP / 2);                               // [LCAP] values
    unsigned short *lc = reinterpret_cast<unsigned short *>(lv + LCAP);               // [LCAP] channel numbers
    for (int e = tid; e < NC * CIN / 4; e += NWV * 64) reinterpret_cast<float4 *>(Wl)[e] = reinterpret_cast<const float4 *>(table)[e];
    __syncthreads();
    unsigned mism = 0;
    f32x2 keep = {0.0f, 0.0f};
    for (int r = 0; r < rounds; r++) {
        // a fresh list every round (written by the lanes, read back by everybody: the write -> read order of the real kernel)
        const int src = ((blockIdx.x * NWV + wv) * 131 + r * 17) % 4096;
        for (int e = lane; e < LCAP; e += 64) {
            lv[e] = vals[(src + e) % 4096];
            lc[e] = codes[(src + e) % 4096];
        }
        f32x2 a = {1.0f, -1.0f}, b = a;
        for (int i = 0; i < trips * 4; i += 4) {
            const int o = i % LCAP;
            const unsigned short *pc = static_cast<const unsigned short *>(__builtin_assume_aligned(&lc[o], 8));
            const float *pv = static_cast<const float *>(__builtin_assume_aligned(&lv[o], 16));
            const unsigned c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3];
            f32x2 v01 = {pv[0], pv[1]}, v23 = {pv[2], pv[3]};
            const f32x2 w0 = *reinterpret_cast<const f32x2 *>(&Wl[c0 * CIN + lane * 2]);
            const f32x2 w1 = *reinterpret_cast<const f32x2 *>(&Wl[c1 * CIN + lane * 2]);
            const f32x2 w2 = *reinterpret_cast<const f32x2 *>(&Wl[c2 * CIN + lane * 2]);
            const f32x2 w3 = *reinterpret_cast<const f32x2 *>(&Wl[c3 * CIN + lane * 2]);
            // the broadcast operand second: even entries op_sel_hi:[1,0,1] (high half x low register), odd entries op_sel:[0,1,0]
            // (low half x HIGH register)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"
                         "v_pk_fma_f32 %0, %3, %2, %0 op_sel:[0,1,0]\n\t"
                         "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n\t"
                         "v_pk_fma_f32 %0, %6, %5, %0 op_sel:[0,1,0]"
                         : "+v"(a)
                         : "v"(w0), "v"(v01), "v"(w1), "v"(w2), "v"(v23), "v"(w3));
            // the broadcast operand first
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]\n\t"
                         "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,0,0]\n\t"
                         "v_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n\t"
                         "v_pk_fma_f32 %0, %5, %6, %0 op_sel:[1,0,0]"
                         : "+v"(b)
                         : "v"(w0), "v"(v01), "v"(w1), "v"(w2), "v"(v23), "v"(w3));
        }
        if (__float_as_uint(a.x) != __float_as_uint(b.x)) mism |= 1u << ((lane >> 4) * 2);
        if (__float_as_uint(a.y) != __float_as_uint(b.y)) mism |= 2u << ((lane >> 4) * 2);
        keep += a;
    }
    if (mism) {
        atomicAdd(&bad[0], 1u);          // lanes with a mismatch
        atomicOr(&bad[1], mism);         // which half (bit 2 q + h: lanes 16 q .. 16 q + 15, half h)
    }
    out[(size_t)blockIdx.x * NWV * 64 + tid] = keep.x + keep.y;
}

// the real kernel's form only (no second chain beside it): results compared between launches on the host
template <bool SRC1>
__global__ __launch_bounds__(NWV * 64) void pk_kernel_one(const float *__restrict__ table, const float *__restrict__ vals,
                                                          const unsigned short *__restrict__ codes, int trips, int rounds,
                                                          float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Wl = smem;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float *lv = Wl + NC * CIN + wv * (LCAP + LCAP / 2);
    unsigned short *lc = reinterpret_cast<unsigned short *>(lv + LCAP);
    for (int e = tid; e < NC * CIN / 4; e += NWV * 64) reinterpret_cast<float4 *>(Wl)[e] = reinterpret_cast<const float4 *>(table)[e];
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        const int src = ((blockIdx.x * NWV + wv) * 131 + r * 17) % 4096;
        for (int e = lane; e < LCAP; e += 64) {
            lv[e] = vals[(src + e) % 4096];
            lc[e] = codes[(src + e) % 4096];
        }
        f32x2 a = {1.0f, -1.0f};
        for (int i = 0; i < trips * 4; i += 4) {
            const int o = i % LCAP;
            const unsigned short *pc = static_cast<const unsigned short *>(__builtin_assume_aligned(&lc[o], 8));
            const float *pv = static_cast<const float *>(__builtin_assume_aligned(&lv[o], 16));
            const unsigned c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3];
            f32x2 v01 = {pv[0], pv[1]}, v23 = {pv[2], pv[3]};
            const f32x2 w0 = *reinterpret_cast<const f32x2 *>(&Wl[c0 * CIN + lane * 2]);
            const f32x2 w1 = *reinterpret_cast<const f32x2 *>(&Wl[c1 * CIN + lane * 2]);
            const f32x2 w2 = *reinterpret_cast<const f32x2 *>(&Wl[c2 * CIN + lane * 2]);
            const f32x2 w3 = *reinterpret_cast<const f32x2 *>(&Wl[c3 * CIN + lane * 2]);
            if (SRC1)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %3, %2, %0 op_sel:[0,1,0]\n\t"
                             "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n\t"
                             "v_pk_fma_f32 %0, %6, %5, %0 op_sel:[0,1,0]"
                             : "+v"(a)
                             : "v"(w0), "v"(v01), "v"(w1), "v"(w2), "v"(v23), "v"(w3));
            else
                asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]\n\t"
                             "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,0,0]\n\t"
                             "v_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n\t"
                             "v_pk_fma_f32 %0, %5, %6, %0 op_sel:[1,0,0]"
                             : "+v"(a)
                             : "v"(w0), "v"(v01), "v"(w1), "v"(w2), "v"(v23), "v"(w3));
        }
        float *o = out + (((size_t)blockIdx.x * rounds + r) * NWV * 64 + tid) * 2;
        o[0] = a.x;
        o[1] = a.y;
    }
}


// the same self-check WITHOUT any LDS traffic in the loop (operands made in registers): one packed form against plain v_fma_f32 /
// v_mul_f32 / v_add_f32 / v_mov_b32 on the halves -- is it the instruction alone, and which operand selections are affected?
//   FORM 1  v_pk_fma_f32 op_sel:[0,1,0]      low half = src0.lo * src1.HI + src2.lo
//   FORM 2  v_pk_fma_f32 op_sel:[1,0,0]      low half = src0.HI * src1.lo + src2.lo
//   FORM 3  v_pk_fma_f32 op_sel_hi:[1,0,1]   high half = src0.hi * src1.LO + src2.hi
//   FORM 4  v_pk_fma_f32 op_sel_hi:[0,1,1]   high half = src0.LO * src1.hi + src2.hi
//   FORM 5  v_pk_mul_f32 op_sel:[0,1]        FORM 6  v_pk_add_f32 op_sel:[0,1]      FORM 7  v_pk_fma_f32 op_sel:[0,0,1] (src2.HI into the low half)
//   FORM 8  v_pk_mov_b32 op_sel:[0,1]        (low = src0.lo, high = src1.HI: the default selection of the high half)
//   FORM 9  v_pk_mov_b32 op_sel:[1,0]        (low = src0.HI, high = src1.lo)
//   FORM 10 v_pk_fma_f32, no operand selection (the plain packed form)
template <int FORM>
__global__ __launch_bounds__(NWV * 64) void pk_kernel_regs(int trips, int rounds, unsigned *__restrict__ bad, float *__restrict__ out)
{
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned mism = 0;
    f32x2 keep = {0.0f, 0.0f};
    for (int r = 0; r < rounds; r++) {
        f32x2 a = {1.0f, -1.0f}, b = a;
        f32x2 w = {0.001f * (float)(lane + 1), -0.002f * (float)(lane + 3)}, v = {0.5f + 0.01f * (float)r, -0.25f};
        for (int i = 0; i < trips * 4; i++) {
            float bx = b.x, by = b.y;
            if (FORM == 1) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a) : "v"(w), "v"(v));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(bx) : "v"(w.x), "v"(v.y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(by) : "v"(w.y), "v"(v.y));
            } else if (FORM == 2) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a) : "v"(w), "v"(v));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(bx) : "v"(w.y), "v"(v.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(by) : "v"(w.y), "v"(v.y));
            } else if (FORM == 3) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "v"(w), "v"(v));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(bx) : "v"(w.x), "v"(v.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(by) : "v"(w.y), "v"(v.x));
            } else if (FORM == 4) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "v"(w), "v"(v));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(bx) : "v"(w.x), "v"(v.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(by) : "v"(w.x), "v"(v.y));
            } else if (FORM == 5) {
                f32x2 t;
                float tx, ty;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(w), "v"(v));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tx) : "v"(w.x), "v"(v.y));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ty) : "v"(w.y), "v"(v.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.x) : "v"(t.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.y) : "v"(t.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bx) : "v"(tx));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(by) : "v"(ty));
            } else if (FORM == 6) {
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(a) : "v"(w));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bx) : "v"(w.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(by) : "v"(w.y));
            } else if (FORM == 7) {
                f32x2 t = a;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(a) : "v"(w), "v"(v), "v"(t));
                const float ty = by;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(bx) : "v"(w.x), "v"(v.x), "v"(ty));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(by) : "v"(w.y), "v"(v.y), "v"(ty));
            } else if (FORM == 8 || FORM == 9) {
                f32x2 t;
                float tx, ty;
                if (FORM == 8) {
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(w), "v"(v));
                    tx = w.x;
                    ty = v.y;
                } else {
                    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(w), "v"(v));
                    tx = w.y;
                    ty = v.x;
                }
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.x) : "v"(t.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a.y) : "v"(t.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(bx) : "v"(tx));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(by) : "v"(ty));
            } else {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(w), "v"(v));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(bx) : "v"(w.x), "v"(v.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(by) : "v"(w.y), "v"(v.y));
            }
            b.x = bx;
            b.y = by;
            w.x += 0.0001f;
            w.y -= 0.0001f;
            v.y += 0.001f;
            v.x -= 0.0005f;
        }
        if (__float_as_uint(a.x) != __float_as_uint(b.x)) mism |= 1u << ((lane >> 4) * 2);
        if (__float_as_uint(a.y) != __float_as_uint(b.y)) mism |= 2u << ((lane >> 4) * 2);
        keep += a;
    }
    if (mism) {
        atomicAdd(&bad[0], 1u);
        atomicOr(&bad[1], mism);
    }
    out[(size_t)blockIdx.x * NWV * 64 + tid] = keep.x + keep.y;
}

__global__ __launch_bounds__(256) void mfma_side(int iters, float *out)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; a++)
        for (int i = 0; i < 16; i++) acc[a][i] = 0.0f;
    bf16x8 va, vb;
    for (int i = 0; i < 8; i++) {
        va[i] = (__bf16)(0.001f * (threadIdx.x & 7));
        vb[i] = (__bf16)(0.002f * (threadIdx.x & 3));
    }
    for (int it = 0; it < iters; it++)
        for (int a = 0; a < 4; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[a], 0, 0, 0);
    float s = 0.0f;
    for (int a = 0; a < 4; a++)
        for (int i = 0; i < 16; i++) s += acc[a][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// side = 2: the neighbour runs plain vector multiply-adds instead of MFMAs (is it the matrix pipe, or any neighbour?)
__global__ __launch_bounds__(256) void valu_side(int iters, float *out)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = 0.001f * (float)(threadIdx.x + i);
    for (int it = 0; it < iters * 8; it++)
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(0.999f));
    float s = 0.0f;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int side = argc > 2 ? atoi(argv[2]) : 0;
    // argv[4] = workgroups of the checked kernels (default 256: one per CU), argv[5] = dynamic LDS bytes the MFMA neighbour asks for (with
    // 32768 it cannot share a CU with a checked workgroup of the LDS-fed kernel: is the hazard local to a compute unit?)
    const int trips = 64, rounds = 48, grid = argc > 4 ? atoi(argv[4]) : 256;
    const int side_lds = argc > 5 ? atoi(argv[5]) : 0;
    const int regs_lds = argc > 6 ? atoi(argv[6]) : 0; // unused dynamic LDS of the register-fed kernels (131072 + a neighbour asking for 32768: never on one CU)
    std::vector<float> table(NC * CIN), vals(4096);
    std::vector<unsigned short> codes(4096);
    unsigned s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
    };
    for (auto &v : table) v = rnd();
    for (auto &v : vals) v = rnd() * 0.05f;
    for (auto &c : codes) {
        s = s * 1664525u + 1013904223u;
        c = (unsigned short)((s >> 12) % NC);
    }
    float *d_table, *d_vals, *d_out, *d_out1, *d_ref, *d_side;
    unsigned short *d_codes;
    unsigned *d_bad;
    const size_t per = (size_t)grid * rounds * NWV * 64 * 2;
    CHECK(hipMalloc(&d_table, table.size() * 4));
    CHECK(hipMalloc(&d_vals, vals.size() * 4));
    CHECK(hipMalloc(&d_codes, codes.size() * 2));
    CHECK(hipMalloc(&d_out, (size_t)grid * NWV * 64 * 4));
    CHECK(hipMalloc(&d_out1, per * 4));
    CHECK(hipMalloc(&d_ref, per * 4));
    CHECK(hipMalloc(&d_side, 2048 * 256 * 4));
    CHECK(hipMalloc(&d_bad, 8));
    CHECK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_codes, codes.data(), codes.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_bad, 0, 8));
    const size_t smem = (size_t)NC * CIN * 4 + (size_t)NWV * (LCAP * 6);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pk_kernel_one<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pk_kernel_one<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipStream_t st, st2;
    CHECK(hipStreamCreate(&st));
    CHECK(hipStreamCreate(&st2));
    // 1. the self-checking kernel
    for (int l = 0; l < launches; l++) {
        if (side == 1) hipLaunchKernelGGL(mfma_side, dim3(1024), dim3(256), side_lds, st2, 4000, d_side);
            if (side == 2) hipLaunchKernelGGL(valu_side, dim3(1024), dim3(256), 0, st2, 4000, d_side);
        hipLaunchKernelGGL(pk_kernel, dim3(grid), dim3(NWV * 64), smem, st, d_table, d_vals, d_codes, trips, rounds, d_bad, d_out);
    }
    CHECK(hipDeviceSynchronize());
    unsigned bad[2];
    CHECK(hipMemcpy(bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("both forms in one kernel, %d launches: %u lanes saw the two accumulators differ (half mask 0x%02x: bit 2q+h = lanes 16q.., half h)\n",
           launches, bad[0], bad[1]);
    static const char *names[] = {"", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
                                  "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]",
                                  "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_fma_f32 (no selection)"};
    for (int form = 1; form <= 10; form++) {
        CHECK(hipMemset(d_bad, 0, 8));
        for (int l = 0; l < launches; l++) {
            if (side == 1) hipLaunchKernelGGL(mfma_side, dim3(1024), dim3(256), side_lds, st2, 4000, d_side);
            if (side == 2) hipLaunchKernelGGL(valu_side, dim3(1024), dim3(256), 0, st2, 4000, d_side);
#define GO(F) case F: if (l == 0 && regs_lds) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(pk_kernel_regs<F>), hipFuncAttributeMaxDynamicSharedMemorySize, regs_lds)); \
    hipLaunchKernelGGL(pk_kernel_regs<F>, dim3(grid), dim3(NWV * 64), regs_lds, st, trips * 8, rounds, d_bad, d_out); break;
            switch (form) { GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8) GO(9) GO(10) }
#undef GO
        }
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(bad, d_bad, 8, hipMemcpyDeviceToHost));
        printf("operands in registers, %-32s against the unpacked instructions: %6u lanes differ (half mask 0x%02x)\n", names[form], bad[0], bad[1]);
    }
    if (argc > 3) return 0;
    // 2. one form per kernel, launch against launch
    std::vector<float> ref(per), got(per);
    for (int form = 1; form >= 0; form--) {
        auto launch = [&](float *dst) {
            if (side == 1) hipLaunchKernelGGL(mfma_side, dim3(1024), dim3(256), side_lds, st2, 4000, d_side);
            if (side == 2) hipLaunchKernelGGL(valu_side, dim3(1024), dim3(256), 0, st2, 4000, d_side);
            if (form)
                hipLaunchKernelGGL(pk_kernel_one<true>, dim3(grid), dim3(NWV * 64), smem, st, d_table, d_vals, d_codes, trips, rounds, dst);
            else
                hipLaunchKernelGGL(pk_kernel_one<false>, dim3(grid), dim3(NWV * 64), smem, st, d_table, d_vals, d_codes, trips, rounds, dst);
        };
        launch(d_ref);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ref.data(), d_ref, per * 4, hipMemcpyDeviceToHost));
        long diff_launches = 0, diff_vals = 0, lo48 = 0;
        for (int l = 0; l < launches; l++) {
            launch(d_out1);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(got.data(), d_out1, per * 4, hipMemcpyDeviceToHost));
            long d = 0;
            for (size_t i = 0; i < per; i++)
                if (got[i] != ref[i]) {
                    d++;
                    const int lane = (int)((i / 2) % 64);
                    if (lane >= 48 && i % 2 == 0) lo48++;
                }
            if (d) diff_launches++;
            diff_vals += d;
        }
        printf("broadcast operand in %s: %ld of %d launches differ from the first (%ld values, %ld of them low half of lanes 48-63)\n",
               form ? "src1" : "src0", diff_launches, launches, diff_vals, lo48);
    }
    return 0;
}
