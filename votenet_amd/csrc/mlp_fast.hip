// mlp_fast.hip -- lean fp32 MFMA GEMM for the dense layers of the grouped-point MLP (gfx950).
//
// Same contract as mlp_linear_kernel (mlp.hip) for the case every dense VoteNet layer is in:
//   DENSE input, cin % 16 == 0, cout % BN == 0, rows % 128 == 0, 16-byte aligned operands, cin <= 512.
// Written separately so that the hot loop carries no bounds checks, no mode branches and few live
// registers (the generic kernel needs > 220 VGPRs and spills its prefetch registers, which turns the
// "asynchronous" global loads into synchronous ones):
//   * operand addresses are two pointers per thread that advance by constants;
//   * a (tile, k-slab) step loads the NEXT step's A / W quads raw into registers right after the LDS
//     barrier, writes them to the other LDS buffer half way through the step's MFMAs -- applying the
//     previous layer's folded BN scale/shift + ReLU at that point, so the global loads stay in flight for
//     half a step of matrix work -- and the barrier waits for LDS only (lgkmcnt), never for vmcnt;
//   * the pipeline runs across row tiles of a persistent workgroup (no drain at tile boundaries);
//   * MFMA operand fragments are double-buffered in registers (ds_reads of sub-step k2+1 before the MFMAs
//     of k2); LDS images are [k][row] / [k][col], conflict-free for both operand reads.
#include "common.h"

namespace votenet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FG_BM = 128;
constexpr int FG_BK = 16;
constexpr int FG_LDA = FG_BM + 2;

// WM x WN waves (WM*WN = 4), each MT x NT tiles of 32x32: BM = WM*MT*32 = 128, BN = WN*NT*32.
// amdgpu_waves_per_eu caps the occupancy the register allocator aims for: at 4 waves/SIMD (128 VGPRs) the
// 2x2 variant spills exactly its prefetch registers, which makes the prefetch synchronous.
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void mlp_linear_fast_kernel(
    const float *__restrict__ x, const float *__restrict__ in_scale, const float *__restrict__ in_shift, int in_relu,
    long rows, int cin, int cout, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ z,
    double *__restrict__ stats)
{
    static_assert(WM * WN == 4 && WM * MT * 32 == FG_BM, "tile shape");
    constexpr int BN = WN * NT * 32;
    constexpr int LDB = BN + 4;
    constexpr int NB4 = FG_BK * BN / 4 / 256; // W float4 per thread per slab (1 or 2)
    __shared__ float As[2][FG_BK][FG_LDA];
    __shared__ float Bs[2][FG_BK][LDB];
    __shared__ __attribute__((aligned(16))) float Ssc[512], Ssh[512];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const int n0 = blockIdx.y * BN;
    const int nk = cin / FG_BK;
    const long ntiles = rows / FG_BM;
    const bool affine = in_scale != nullptr;
    if (affine) {
        for (int k = tid; k < cin; k += 256) {
            Ssc[k] = in_scale[k];
            Ssh[k] = in_shift[k];
        }
    }
    long my_tiles = 0;
    if ((long)blockIdx.x < ntiles) my_tiles = (ntiles - 1 - blockIdx.x) / gridDim.x + 1;
    long steps_to_load = my_tiles * nk; // steps whose operands still have to be fetched

    // A staging: thread t -> tile rows (t>>2) and (t>>2)+64, k-quad (t&3); W staging: float4 #t (+256)
    const int a_row = tid >> 2, a_kq = tid & 3;
    const float *pa0 = x + ((size_t)blockIdx.x * FG_BM + a_row) * cin + a_kq * 4;
    const float *pa1 = pa0 + (size_t)64 * cin;
    const size_t a_tile_jump = (size_t)gridDim.x * FG_BM * cin - cin; // after the last slab of a tile
    const float *pb[NB4];
#pragma unroll
    for (int u = 0; u < NB4; u++) {
        const int f = tid + u * 256;
        pb[u] = w + (size_t)(f / (BN / 4)) * cout + n0 + (f % (BN / 4)) * 4;
    }
    const size_t b_step = (size_t)FG_BK * cout, b_wrap = (size_t)cin * cout;
    int lkt = 0; // k-slab index of the step being loaded
    float4 ra0, ra1, rb[NB4];
    int rk = 0;
    auto issue_loads = [&]() {
        ra0 = *reinterpret_cast<const float4 *>(pa0);
        ra1 = *reinterpret_cast<const float4 *>(pa1);
#pragma unroll
        for (int u = 0; u < NB4; u++) rb[u] = *reinterpret_cast<const float4 *>(pb[u]);
        rk = lkt * FG_BK + a_kq * 4;
        pa0 += FG_BK;
        pa1 += FG_BK;
#pragma unroll
        for (int u = 0; u < NB4; u++) pb[u] += b_step;
        if (++lkt == nk) {
            lkt = 0;
            pa0 += a_tile_jump;
            pa1 += a_tile_jump;
#pragma unroll
            for (int u = 0; u < NB4; u++) pb[u] -= b_wrap;
        }
        --steps_to_load;
    };
    auto act4 = [&](float4 v) {
        if (affine) {
            const float4 sc = *reinterpret_cast<const float4 *>(&Ssc[rk]);
            const float4 sh = *reinterpret_cast<const float4 *>(&Ssh[rk]);
            v.x = v.x * sc.x + sh.x;
            v.y = v.y * sc.y + sh.y;
            v.z = v.z * sc.z + sh.z;
            v.w = v.w * sc.w + sh.w;
            if (in_relu) {
                v.x = v.x > 0.f ? v.x : 0.f;
                v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f;
                v.w = v.w > 0.f ? v.w : 0.f;
            }
        }
        return v;
    };
    auto store_regs = [&](int buf) {
        const float4 v0 = act4(ra0), v1 = act4(ra1);
        As[buf][a_kq * 4 + 0][a_row] = v0.x;
        As[buf][a_kq * 4 + 1][a_row] = v0.y;
        As[buf][a_kq * 4 + 2][a_row] = v0.z;
        As[buf][a_kq * 4 + 3][a_row] = v0.w;
        As[buf][a_kq * 4 + 0][a_row + 64] = v1.x;
        As[buf][a_kq * 4 + 1][a_row + 64] = v1.y;
        As[buf][a_kq * 4 + 2][a_row + 64] = v1.z;
        As[buf][a_kq * 4 + 3][a_row + 64] = v1.w;
#pragma unroll
        for (int u = 0; u < NB4; u++) {
            const int f = tid + u * 256;
            *reinterpret_cast<float4 *>(&Bs[buf][f / (BN / 4)][(f % (BN / 4)) * 4]) = rb[u];
        }
    };

    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) s1[j] = s2[j] = 0.0f;
    if (my_tiles == 0) return;
    __syncthreads(); // Ssc / Ssh
    issue_loads();   // step 0
    store_regs(0);
    if (steps_to_load > 0) issue_loads(); // step 1 in flight
    __syncthreads();

    const int kh = lane >> 5, l31 = lane & 31;
    int buf = 0;
    long steps_left = my_tiles * nk; // steps still to compute, including the current one
    for (long t = 0; t < my_tiles; t++) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
        for (int kt = 0; kt < nk; kt++) {
            const bool have_next = steps_left > 1; // registers hold the next step
            float fa[2][MT], fb[2][NT];
#pragma unroll
            for (int i = 0; i < MT; i++) fa[0][i] = As[buf][kh][(wm * MT + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; j++) fb[0][j] = Bs[buf][kh][(wn * NT + j) * 32 + l31];
#pragma unroll
            for (int k2 = 0; k2 < FG_BK / 2; k2++) {
                if (k2 == FG_BK / 4 && have_next) {
                    store_regs(buf ^ 1); // the other buffer was last read one step ago, behind a barrier
                    if (steps_to_load > 0) issue_loads();
                }
                if (k2 + 1 < FG_BK / 2) {
#pragma unroll
                    for (int i = 0; i < MT; i++) fa[(k2 + 1) & 1][i] = As[buf][(k2 + 1) * 2 + kh][(wm * MT + i) * 32 + l31];
#pragma unroll
                    for (int j = 0; j < NT; j++) fb[(k2 + 1) & 1][j] = Bs[buf][(k2 + 1) * 2 + kh][(wn * NT + j) * 32 + l31];
                }
                __builtin_amdgcn_sched_barrier(0); // keep the reads of k2+1 ahead of the MFMAs of k2
#pragma unroll
                for (int i = 0; i < MT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k2 & 1][i], fb[k2 & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier(); // LDS only: the prefetched global loads stay in flight across it
            buf ^= 1;
            --steps_left;
        }
        // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
        const long m0 = ((long)blockIdx.x + t * gridDim.x) * FG_BM;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int col = n0 + (wn * NT + j) * 32 + l31;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < MT; i++) {
                float *zr = z + (size_t)(m0 + (wm * MT + i) * 32 + 4 * kh) * cout + col;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const float v = acc[i][j][e] + bv;
                    zr[(size_t)((e & 3) + 8 * (e >> 2)) * cout] = v;
                    s1[j] += v;
                    s2[j] += v * v;
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32);
            const float t2 = s2[j] + __shfl_xor(s2[j], 32);
            const int col = n0 + (wn * NT + j) * 32 + l31;
            if (lane < 32) {
                unsafeAtomicAdd(&stats[col], (double)t1);
                unsafeAtomicAdd(&stats[cout + col], (double)t2);
            }
        }
    }
}

// returns true when the fast kernel took the launch
bool mlp_linear_fast_launch(const float *x, const float *in_scale, const float *in_shift, int in_relu, long rows, int cin,
                            int cout, const float *w, const float *bias, float *z, double *stats, hipStream_t st)
{
    const bool aligned = ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)z % 16 == 0);
    if (!aligned || cin % FG_BK != 0 || cin > 512 || rows % FG_BM != 0 || rows == 0) return false;
    const long ntiles = rows / FG_BM;
    if (cout % 128 == 0) {
        const int ny = cout / 128;
        long gx = ntiles < 1024 / ny ? ntiles : 1024 / ny;
        hipLaunchKernelGGL((mlp_linear_fast_kernel<2, 2, 2, 2>), dim3((unsigned)gx, ny), dim3(256), 0, st, x, in_scale, in_shift,
                           in_relu, rows, cin, cout, w, bias, z, stats);
        return true;
    }
    if (cout == 64) {
        long gx = ntiles < 2048 ? ntiles : 2048;
        hipLaunchKernelGGL((mlp_linear_fast_kernel<4, 1, 1, 2>), dim3((unsigned)gx, 1), dim3(256), 0, st, x, in_scale, in_shift,
                           in_relu, rows, cin, cout, w, bias, z, stats);
        return true;
    }
    return false;
}

} // namespace votenet
