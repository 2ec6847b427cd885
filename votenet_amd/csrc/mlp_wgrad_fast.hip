// mlp_wgrad_fast.hip -- lean weight-gradient GEMM  dW (cin x cout) += X^T * dZ  for the aligned shapes every wide
// VoteNet layer has (gfx950).  Same contract as mlp_wgrad_kernel (mlp_bwd.hip), which keeps serving the ragged shapes:
//   cin % (64*TI) == 0, cout % (64*TJ) == 0, 16-byte aligned operands; GATHER input: the feature block only
//   (X = feat[scene, idx[row], :], dW rows offset by 3 -- the xyz columns go through wgrad_narrow_kernel).
// What the lean form buys, as in mlp_fast.hip:
//   * the loop body has NO memory operation under a branch (rows past the end re-read the last row and are zeroed on
//     the dZ side when the slab is written to LDS), so the compiler's s_waitcnt bookkeeping stays exact;
//   * two register sets hold the raw quads of the next two 16-row slabs: half way through a slab's MFMAs the older set
//     is written to the other LDS buffer (folded BN+ReLU of the layer below on X; BatchNorm backward rebuilt from
//     (da | pooled gout, z, coef) on dZ -- struct BnSrc; BSRC 3: dZ is itself a folded activation, for the Gram matrix of
//     votenet_mlp_gram) and refilled with the slab three steps ahead.  Loads lead by
//     two slabs of matrix work; one slab is shorter than the loaded HBM latency;
//   * NARROW (MODE 2): X = z0 of a narrow first layer, rebuilt from the row's eight floats u8[r] with narrow_z (narrow.hip): the
//     thread's four channels of W0 / b0 sit in registers, a slab row costs 32 bytes instead of 4*cin;
//   * ASSEMBLED (MODE 3): X = z0 of a first SA layer rebuilt from geo[r] and a gather of the per-point table P (assemble.hip); the geo
//     of a slab travels like the idx of a GATHER slab, one refill ahead;
//   * GATHER: the idx of a slab is loaded one refill BEFORE the feature rows that need it, and ahead of that refill's
//     other loads in program order, so neither the dependency nor vmcnt's in-order retirement exposes it.
// LDS images are the natural [row][channel] slabs; lane l reads As[k2*2 + (l>>5)][i0 + (l&31)]: conflict-free.
//
// BF3 = true (the default for every shape once votenet_debug_wgrad_bf3 is on): the products on bf16 x 3 split operands as in
// mlp_fast.hip -- six v_mfma_f32_32x32x16_bf16 per 32 x 32 sub-tile and 16-row slab instead of eight v_mfma_f32_32x32x2_f32 at a
// sixteenth of the rate.  The contraction runs over the ROWS here, so an MFMA fragment is 8 consecutive rows of ONE channel, while the
// loaders hold 4 consecutive channels of one row: the slabs stay row-major in LDS -- three bf16 images [piece][row][channel], split
// where they are staged (3 x ds_write_b64 per float4) -- and the fragments come out through the transposing LDS read
// ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-channel block (lane s: row s / 4, channels 4 (s % 4) ..) and lane t receives
// channel t of the four rows (tools/probe/src/tr_b16_probe.hip: semantics, and the row pitch that keeps it conflict-free -- the read
// banks like ds_read_b64 at the same addresses: a pitch of 64 bytes mod 256 puts the four rows of a half-wave on disjoint banks).
// Both operands use the same row -> k assignment (lanes 0-31: rows 0-3 | 4-7, lanes 32-63: rows 8-11 | 12-15), which is all a
// contraction needs.  Loaders, channel constants, the two-set prefetch and the epilogue are shared with the fp32 form.
#include "mlp_types.h"

namespace votenet {

constexpr int WF_BR = 16; // rows per slab (the MFMA contraction index)

typedef short tr_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_v4s *tr_lds_ptr;
// 8 consecutive rows of one channel (two transposing reads 4 rows apart) as one bf16 x 8 MFMA operand; lds_off: byte offset in LDS
__device__ __forceinline__ uint4 tr_fragment(unsigned lds_off, unsigned four_rows)
{
    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_ptr)(uintptr_t)lds_off);
    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_ptr)(uintptr_t)(lds_off + four_rows));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return make_uint4(a.x, a.y, b.x, b.y);
}

template <int MODE, int TI, int TJ, int BSRC, bool BF3 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void mlp_wgrad_fast_kernel(
    MlpIn in, long rows, int cin, int cout, const float *__restrict__ dz, BnSrc bs, float *__restrict__ dw, long rows_per_block)
{
    if (bs.nh_dev != nullptr) {
        const long lim = (long)bs.nh_dev[0] * kPiece;
        rows = lim < rows ? lim : rows;
    }
    constexpr int BI = 64 * TI, BJ = 64 * TJ;
    constexpr int QA = BI / 4, QB = BJ / 4;                       // float4 per slab row
    constexpr int NA = WF_BR * QA / 256, NB = WF_BR * QB / 256;   // float4 per thread per slab (1 or 2)
    constexpr int RA = 256 / QA, RB = 256 / QB;                   // slab rows covered by one pass of the 256 threads
    __shared__ float As[BF3 ? 1 : 2][BF3 ? 1 : WF_BR][BI + 4];
    __shared__ float Bs[BF3 ? 1 : 2][BF3 ? 1 : WF_BR][BJ + 4];
    // BF3: [buffer][piece hi, mid, lo][row][channel] bf16, row pitch = the channels' bytes + 64 (see the file header)
    constexpr int PA = BI * 2 + 64, PB = BJ * 2 + 64;
    __shared__ __attribute__((aligned(16))) unsigned char A3[BF3 ? 2 * 3 * WF_BR * PA : 16];
    __shared__ __attribute__((aligned(16))) unsigned char B3[BF3 ? 2 * 3 * WF_BR * PB : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const int i0 = blockIdx.y * BI, j0 = blockIdx.z * BJ;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    if (r_begin >= rows) return;
    const int nrow = (int)((r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows) - r_begin);
    const int nslab = (nrow + WF_BR - 1) / WF_BR;
    const int a_row = tid / QA, a_q = tid % QA;
    const int b_row = tid / QB, b_q = tid % QB;
    const int ka = i0 + a_q * 4; // this thread's X channels
    const int nb = j0 + b_q * 4; // this thread's dZ channels
    const int xc = (MODE == 0) ? cin : in.c;

    // per-thread channel constants in registers
    const bool affine = (MODE == 0 || MODE == 2 || MODE == 3) && in.in_scale != nullptr;
    float4 wx0 = make_float4(0.f, 0.f, 0.f, 0.f), wx1 = wx0, wx2 = wx0; // MODE 3: W[0:3][ka..ka+3]
    if (MODE == 3) {
        wx0 = *reinterpret_cast<const float4 *>(in.wx + ka);
        wx1 = *reinterpret_cast<const float4 *>(in.wx + cin + ka);
        wx2 = *reinterpret_cast<const float4 *>(in.wx + 2 * cin + ka);
    }
    float w0r[4][8], b0r[4]; // MODE 2: W0[:, ka..ka+3] (zero padded to 8 rows) and b0[ka..ka+3]
    if (MODE == 2) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const float4 w4 = d < in.k0 ? *reinterpret_cast<const float4 *>(in.w0 + (size_t)d * cin + ka) : make_float4(0.f, 0.f, 0.f, 0.f);
            w0r[0][d] = w4.x;
            w0r[1][d] = w4.y;
            w0r[2][d] = w4.z;
            w0r[3][d] = w4.w;
        }
        const float4 b4 = in.b0 ? *reinterpret_cast<const float4 *>(in.b0 + ka) : make_float4(0.f, 0.f, 0.f, 0.f);
        b0r[0] = b4.x;
        b0r[1] = b4.y;
        b0r[2] = b4.z;
        b0r[3] = b4.w;
    }
    float4 csc = make_float4(1.f, 1.f, 1.f, 1.f), csh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (affine) {
        csc = *reinterpret_cast<const float4 *>(in.in_scale + ka);
        csh = *reinterpret_cast<const float4 *>(in.in_shift + ka);
    }
    const float x_floor = (affine && in.in_relu) ? 0.0f : -__builtin_inff(); // ReLU as a floor: branch-free
    float4 kA, kB, kC, kS, kH;
    kA = kB = kC = kS = kH = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BSRC == 1 || BSRC == 2 || BSRC == 4) {
        kA = *reinterpret_cast<const float4 *>(bs.coef + nb);
        kB = *reinterpret_cast<const float4 *>(bs.coef + cout + nb);
        kC = *reinterpret_cast<const float4 *>(bs.coef + 2 * cout + nb);
        kS = *reinterpret_cast<const float4 *>(bs.coef + 3 * cout + nb);
        kH = *reinterpret_cast<const float4 *>(bs.coef + 4 * cout + nb);
    }
    if (BSRC == 3) { // the right operand is an ACTIVATION: act(z * scale + shift); coef = [scale | shift]
        kS = *reinterpret_cast<const float4 *>(bs.coef + nb);
        kH = *reinterpret_cast<const float4 *>(bs.coef + cout + nb);
    }
    const float b_floor = (BSRC == 3 && bs.relu) ? 0.0f : -__builtin_inff();
    // wave-uniform bases at the workgroup's first row; threads carry 32-bit element offsets (checked by the launcher)
    const float *xb = (MODE == 0) ? in.x + (size_t)r_begin * cin + ka : (MODE == 2) ? in.u8 + (size_t)r_begin * 8 : (MODE == 3) ? in.ptab + ka : in.feat + ka;
    const float4 *geob = (MODE == 3) ? reinterpret_cast<const float4 *>(in.geo) + r_begin : nullptr;
    const float *zb = (BSRC == 0 ? dz : bs.z) + (size_t)r_begin * cout + nb;
    const float *gb = (BSRC == 1 || BSRC == 4) ? bs.da + (size_t)r_begin * cout + nb : nullptr;
    const int *idxb = (MODE == 1) ? in.idx + r_begin : nullptr;
    const unsigned grows = (MODE == 1) ? (unsigned)in.m * (unsigned)in.nsample : 1u; // rows per scene

    struct Regs {
        float4 a[NA], b[NB], g[NB];
        float4 a2[(MODE == 2 || MODE == 3) ? NA : 1]; // MODE 2: the second half of the rows' u; MODE 3: the rows' geo
        int4 m[NB];
        float mu[NB]; // BSRC 4: the row's weight
        int s; // slab index (local)
    };
    Regs R[2];
    struct Pix { // what a GATHER / ASSEMBLED row needs one refill ahead: its idx, or its geo record
        int i;
        float4 g;
    };
    Pix pidx[NA]; // of this thread's rows of the slab loaded by the NEXT refill
    auto clampr = [&](int lr) { return lr < nrow ? lr : nrow - 1; };
    auto load_idx = [&](int s) {
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int h = 0; h < NA; h++) {
                const int lr = clampr(s * WF_BR + a_row + h * RA);
                if (MODE == 1) pidx[h].i = idxb[lr];
                else pidx[h].g = geob[lr];
            }
        }
    };
    auto load_slab = [&](Regs &r, int s, const Pix (&pi)[NA]) {
        r.s = s;
#pragma unroll
        for (int h = 0; h < NA; h++) {
            const int lr = clampr(s * WF_BR + a_row + h * RA);
            if (MODE == 0) {
                r.a[h] = *reinterpret_cast<const float4 *>(xb + (size_t)((unsigned)lr * (unsigned)cin));
            } else if (MODE == 2) {
                r.a[h] = *reinterpret_cast<const float4 *>(xb + (size_t)((unsigned)lr * 8u));
                r.a2[h] = *reinterpret_cast<const float4 *>(xb + (size_t)((unsigned)lr * 8u) + 4);
            } else if (MODE == 3) {
                r.a[h] = *reinterpret_cast<const float4 *>(xb + (size_t)__float_as_uint(pi[h].g.w) * cin);
                r.a2[h] = pi[h].g;
            } else {
                const unsigned scene = (unsigned)(r_begin + lr) / grows;
                r.a[h] = *reinterpret_cast<const float4 *>(xb + ((size_t)scene * in.n + pi[h].i) * xc);
            }
        }
#pragma unroll
        for (int h = 0; h < NB; h++) {
            const int lr = clampr(s * WF_BR + b_row + h * RB);
            const unsigned off = (unsigned)lr * (unsigned)cout;
            r.b[h] = *reinterpret_cast<const float4 *>(zb + (size_t)off);
            if (BSRC == 1 || BSRC == 4) r.g[h] = *reinterpret_cast<const float4 *>(gb + (size_t)off);
            if (BSRC == 4) {
                const unsigned gr = (unsigned)(r_begin + lr);
                r.mu[h] = (gr % (unsigned)kPiece) == 0u ? bs.wh[gr / (unsigned)kPiece] : 1.0f; // row 0 of a piece (half.hip)
            }
            if (BSRC == 2) {
                const unsigned gr = (unsigned)(r_begin + lr);
                const unsigned grp = bs.pool_shift >= 0 ? gr >> bs.pool_shift : gr / (unsigned)bs.pool_k;
                r.g[h] = *reinterpret_cast<const float4 *>(bs.gout + (size_t)grp * cout + nb);
                r.m[h] = *reinterpret_cast<const int4 *>(bs.argmax + (size_t)grp * cout + nb);
            }
        }
    };
    auto store_slab = [&](int buf, const Regs &r) {
#pragma unroll
        for (int h = 0; h < NA; h++) {
            float4 v = r.a[h];
            if (MODE == 2) {
                const float4 v2 = r.a2[h];
                const float uu[8] = {v.x, v.y, v.z, v.w, v2.x, v2.y, v2.z, v2.w};
                const f32x2 z01 = narrow_z2(uu, w0r[0], w0r[1], b0r[0], b0r[1]), z23 = narrow_z2(uu, w0r[2], w0r[3], b0r[2], b0r[3]);
                v.x = z01.x;
                v.y = z01.y;
                v.z = z23.x;
                v.w = z23.y;
            }
            if (MODE == 3) {
                const float4 g = r.a2[h];
                const f32x2 z01 = assembled_z2(v.x, v.y, g, wx0.x, wx0.y, wx1.x, wx1.y, wx2.x, wx2.y);
                const f32x2 z23 = assembled_z2(v.z, v.w, g, wx0.z, wx0.w, wx1.z, wx1.w, wx2.z, wx2.w);
                v.x = z01.x;
                v.y = z01.y;
                v.z = z23.x;
                v.w = z23.y;
            }
            v.x = fmaxf(v.x * csc.x + csh.x, x_floor);
            v.y = fmaxf(v.y * csc.y + csh.y, x_floor);
            v.z = fmaxf(v.z * csc.z + csh.z, x_floor);
            v.w = fmaxf(v.w * csc.w + csh.w, x_floor);
            if constexpr (BF3) {
                unsigned h0, m0, l0, h1, m1, l1;
                split3(v.x, v.y, h0, m0, l0);
                split3(v.z, v.w, h1, m1, l1);
                unsigned char *d = A3 + ((size_t)(buf * 3) * WF_BR + a_row + h * RA) * PA + a_q * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(d + WF_BR * PA) = make_uint2(m0, m1);
                *reinterpret_cast<uint2 *>(d + 2 * WF_BR * PA) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<float4 *>(&As[buf][a_row + h * RA][a_q * 4]) = v;
            }
        }
#pragma unroll
        for (int h = 0; h < NB; h++) {
            const int lr = r.s * WF_BR + b_row + h * RB;
            float4 v = r.b[h];
            if (BSRC == 3) {
                v.x = fmaxf(v.x * kS.x + kH.x, b_floor);
                v.y = fmaxf(v.y * kS.y + kH.y, b_floor);
                v.z = fmaxf(v.z * kS.z + kH.z, b_floor);
                v.w = fmaxf(v.w * kS.w + kH.w, b_floor);
            }
            if (BSRC == 1 || BSRC == 2 || BSRC == 4) {
                float4 g = r.g[h];
                if (BSRC == 2) {
                    const unsigned gr = (unsigned)(r_begin + lr);
                    const int ro = bs.pool_shift >= 0 ? (int)(gr & (unsigned)(bs.pool_k - 1)) : (int)(gr % (unsigned)bs.pool_k);
                    g.x = (r.m[h].x == ro) ? g.x : 0.0f;
                    g.y = (r.m[h].y == ro) ? g.y : 0.0f;
                    g.z = (r.m[h].z == ro) ? g.z : 0.0f;
                    g.w = (r.m[h].w == ro) ? g.w : 0.0f;
                }
                if (bs.relu) {
                    if (!(v.x * kS.x + kH.x > 0.0f)) g.x = 0.0f;
                    if (!(v.y * kS.y + kH.y > 0.0f)) g.y = 0.0f;
                    if (!(v.z * kS.z + kH.z > 0.0f)) g.z = 0.0f;
                    if (!(v.w * kS.w + kH.w > 0.0f)) g.w = 0.0f;
                }
                if (BSRC == 4) {
                    const float mu = r.mu[h];
                    v.x = kA.x * g.x + mu * (kB.x + kC.x * v.x);
                    v.y = kA.y * g.y + mu * (kB.y + kC.y * v.y);
                    v.z = kA.z * g.z + mu * (kB.z + kC.z * v.z);
                    v.w = kA.w * g.w + mu * (kB.w + kC.w * v.w);
                } else {
                    v.x = kA.x * g.x + kB.x + kC.x * v.x;
                    v.y = kA.y * g.y + kB.y + kC.y * v.y;
                    v.z = kA.z * g.z + kB.z + kC.z * v.z;
                    v.w = kA.w * g.w + kB.w + kC.w * v.w;
                }
            }
            if (lr >= nrow) v = make_float4(0.f, 0.f, 0.f, 0.f); // padding rows contribute nothing
            if constexpr (BF3) {
                unsigned h0, m0, l0, h1, m1, l1;
                split3(v.x, v.y, h0, m0, l0);
                split3(v.z, v.w, h1, m1, l1);
                unsigned char *d = B3 + ((size_t)(buf * 3) * WF_BR + b_row + h * RB) * PB + b_q * 8;
                *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(d + WF_BR * PB) = make_uint2(m0, m1);
                *reinterpret_cast<uint2 *>(d + 2 * WF_BR * PB) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<float4 *>(&Bs[buf][b_row + h * RB][b_q * 4]) = v;
            }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; a++)
#pragma unroll
        for (int b = 0; b < TJ; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[a][b][e] = 0.0f;

    // prologue: slab 0 -> LDS buffer 0; slabs 1 and 2 in flight in sets 1 and 0; idx of slab 3 in pidx.
    // Slab numbers past the end are clamped row by row (they re-read the last row and are stored as zeros).
    Pix pcur[NA];
    load_idx(0);
#pragma unroll
    for (int h = 0; h < NA; h++) pcur[h] = pidx[h];
    load_idx(1);
    load_slab(R[0], 0, pcur);
    store_slab(0, R[0]);
#pragma unroll
    for (int h = 0; h < NA; h++) pcur[h] = pidx[h];
    load_idx(2);
    load_slab(R[1], 1, pcur);
    __builtin_amdgcn_sched_barrier(0); // same issue order as in the loop: idx, set 1, idx, set 0
#pragma unroll
    for (int h = 0; h < NA; h++) pcur[h] = pidx[h];
    load_idx(3);
    load_slab(R[0], 2, pcur);
    __syncthreads();

    int buf = 0;
    const int kh = lane >> 5, l31 = lane & 31;
    const int nslab2 = (nslab + 1) & ~1; // the loop runs slab pairs; a padding slab multiplies zeros
    for (int s = 0; s < nslab2; s += 2) {
#pragma unroll
        for (int par = 0; par < 2; par++) {
            Regs &rs = R[par ^ 1]; // holds slab s+par+1
            if constexpr (BF3) {
                // one slab = ONE k-step of v_mfma_f32_32x32x16_bf16 per piece pair.  Lane l of a 16-lane group g = l / 16 addresses
                // row 8 (g / 2) + (l % 16) / 4 (and + 4), channels 16 (g % 2) + 4 (l % 4) .. of the 32-channel block: it receives
                // channel l % 32, rows 8 (l / 32) .. + 7
                const unsigned g = (unsigned)lane >> 4, sl = (unsigned)lane & 15u;
                const unsigned la = ((g >> 1) * 8u + (sl >> 2)) * PA + ((g & 1u) * 16u + (sl & 3u) * 4u) * 2u + (unsigned)(wi * TI) * 64u;
                const unsigned lb = ((g >> 1) * 8u + (sl >> 2)) * PB + ((g & 1u) * 16u + (sl & 3u) * 4u) * 2u + (unsigned)(wj * TJ) * 64u;
                const unsigned a0 = (unsigned)(uintptr_t)A3 + (unsigned)buf * (3u * WF_BR * PA) + la;
                const unsigned b0 = (unsigned)(uintptr_t)B3 + (unsigned)buf * (3u * WF_BR * PB) + lb;
                uint4 fa3[3][TI], fb3[3][TJ];
#pragma unroll
                for (int pc = 0; pc < 3; pc++) {
#pragma unroll
                    for (int t = 0; t < TI; t++) fa3[pc][t] = tr_fragment(a0 + (unsigned)pc * (WF_BR * PA) + (unsigned)t * 64u, 4u * PA);
#pragma unroll
                    for (int t = 0; t < TJ; t++) fb3[pc][t] = tr_fragment(b0 + (unsigned)pc * (WF_BR * PB) + (unsigned)t * 64u, 4u * PB);
                }
                auto mm = [&](int pa, int pb) {
#pragma unroll
                    for (int a = 0; a < TI; a++)
#pragma unroll
                        for (int b = 0; b < TJ; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa3[pa][a]),
                                                                                __builtin_bit_cast(bf16x8, fb3[pb][b]), acc[a][b], 0, 0, 0);
                };
#ifdef WG_ABL_HALF // probe builds only (tools/probe/ablate_h2_bwd.sh): three of the six products -- results wrong by construction, only the time is read
                mm(2, 0);
#else
                mm(2, 0); // lo * hi
                mm(0, 2); // hi * lo: the smallest terms first
                mm(1, 1);
#endif
                store_slab(buf ^ 1, rs); // the other buffer was last read one step ago, behind a barrier
#pragma unroll
                for (int h = 0; h < NA; h++) pcur[h] = pidx[h];
                load_idx(s + par + 4);
                load_slab(rs, s + par + 3, pcur);
#ifdef WG_ABL_HALF
                mm(0, 1);
                mm(0, 0);
#else
                mm(1, 0);
                mm(0, 1);
                mm(0, 0);
#endif
                lds_barrier();
                buf ^= 1;
                continue;
            }
            float fa[2][TI], fb[2][TJ]; // register double-buffered fragments: reads of k2+1 issued before MFMAs of k2
#pragma unroll
            for (int t = 0; t < TI; t++) fa[0][t] = As[buf][kh][(wi * TI + t) * 32 + l31];
#pragma unroll
            for (int t = 0; t < TJ; t++) fb[0][t] = Bs[buf][kh][(wj * TJ + t) * 32 + l31];
#pragma unroll
            for (int k2 = 0; k2 < WF_BR / 2; k2++) {
                if (k2 == WF_BR / 4) {
                    store_slab(buf ^ 1, rs); // the other buffer was last read one step ago, behind a barrier
#pragma unroll
                    for (int h = 0; h < NA; h++) pcur[h] = pidx[h];
                    load_idx(s + par + 4);           // idx first: it must be OLDER than the loads that follow
                    load_slab(rs, s + par + 3, pcur); // refill with the slab three steps ahead
                }
                if (k2 + 1 < WF_BR / 2) {
#pragma unroll
                    for (int t = 0; t < TI; t++) fa[(k2 + 1) & 1][t] = As[buf][(k2 + 1) * 2 + kh][(wi * TI + t) * 32 + l31];
#pragma unroll
                    for (int t = 0; t < TJ; t++) fb[(k2 + 1) & 1][t] = Bs[buf][(k2 + 1) * 2 + kh][(wj * TJ + t) * 32 + l31];
                }
                __builtin_amdgcn_sched_barrier(0); // keep the reads of k2+1 ahead of the MFMAs of k2
#pragma unroll
                for (int a = 0; a < TI; a++)
#pragma unroll
                    for (int b = 0; b < TJ; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k2 & 1][a], fb[k2 & 1][b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier(); // LDS only: the prefetched global loads stay in flight across it
            buf ^= 1;
        }
    }
    // epilogue: atomically add the partial tile.  C/D: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5)
    const int wrow0 = (MODE == 1) ? 3 : 0;
#pragma unroll
    for (int a = 0; a < TI; a++)
#pragma unroll
        for (int b = 0; b < TJ; b++) {
            const int j = j0 + (wj * TJ + b) * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = i0 + (wi * TI + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const size_t off = (size_t)(wrow0 + i) * cout + j;
                if (bs.part) bs.part[(size_t)blockIdx.x * bs.pstride + off] = acc[a][b][e]; // ordered reduction follows (BnSrc::part)
                else unsafeAtomicAdd(&dw[off], acc[a][b][e]);
            }
        }
}

int g_wgrad_fast_wgs = 0; // votenet_debug_wgrad_workgroups (tuning hook): 0 = the measured defaults below
int g_wgrad_bf3 = 1;      // votenet_debug_wgrad_bf3: 1 = products on bf16 x 3 split operands (transposing LDS reads), 0 = fp32 MFMA
static void plan_fast(long rows, int cin, int cout, int &TIr, int &TJr, int &ti, int &tj, long &rpb, unsigned &gx, bool partials)
{
    TIr = cin % 128 == 0 ? 2 : 1;
    TJr = cout % 128 == 0 ? 2 : 1;
    ti = cin / (64 * TIr);
    tj = cout / (64 * TJr);
    // 384 workgroups, not the 768 that are fastest when the kernel has the GPU to itself (+12 % there): in a train step it runs
    // on its own stream beside the input-gradient chain, which is the critical one (8.00 -> 7.90 ms per step, measured)
    // with partial tiles + ordered reduction (scratch given) 256: every extra workgroup is another 64 KB slice to write and read
    // (same-box A/B of the train step: 384 -> 8.03 ms, 256 -> 7.91 ms, 192 -> 8.04 ms, 128 -> 8.65 ms; atomics, 384: 7.83 ms)
    const int wgs = g_wgrad_fast_wgs > 0 ? g_wgrad_fast_wgs : (partials ? 256 : 384);
    long splits = wgs / (ti * tj);
    if (splits < 1) splits = 1;
    rpb = (rows + splits - 1) / splits;
    rpb = (rpb + 2 * WF_BR - 1) / (2 * WF_BR) * (2 * WF_BR);
    if (rpb < 8 * WF_BR) rpb = 8 * WF_BR; // short row ranges are dominated by the flush of the dW tile (measured)
    gx = (unsigned)((rows + rpb - 1) / rpb);
}

long wgrad_fast_slices(long rows, int cin, int cout)
{
    if (cin % 64 != 0 || cout % 64 != 0 || rows <= 0) return 0;
    int TIr, TJr, ti, tj;
    long rpb;
    unsigned gx;
    plan_fast(rows, cin, cout, TIr, TJr, ti, tj, rpb, gx, false); // the larger of the two plans: an upper bound
    return gx;
}

void wgrad_reduce(int nslice, long pstride, long e0, long e1, const float *part, float *dw, hipStream_t st); // mlp_bwd.hip

template <int MODE, int BSRC>
static bool launch(const MlpIn &d, long rows, int cin, int cout, const float *dz, BnSrc bs, float *dw, hipStream_t st, float *scratch)
{
    int TIr, TJr, ti, tj;
    long rpb;
    unsigned gx;
    plan_fast(rows, cin, cout, TIr, TJr, ti, tj, rpb, gx, scratch != nullptr);
    const int wide = cin > cout ? cin : cout;
    if (rpb * wide >= (1L << 31)) return false; // 32-bit element offsets inside a workgroup's row range
    const int wrow0 = (MODE == 1) ? 3 : 0;
    bs.part = scratch;
    bs.pstride = (long)(wrow0 + cin) * cout;
    const dim3 grid(gx, ti, tj);
    if (g_wgrad_bf3 && BSRC != 3) { // (BSRC 3, the Gram matrix, has its own split-operand kernel: pool_bwd.hip)
        if (TIr == 2 && TJr == 2)
            hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 2, 2, BSRC, true>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
        else if (TIr == 2)
            hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 2, 1, BSRC, true>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
        else if (TJr == 2)
            hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 1, 2, BSRC, true>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
        else
            hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 1, 1, BSRC, true>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    } else if (TIr == 2 && TJr == 2)
        hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 2, 2, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    else if (TIr == 2)
        hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 2, 1, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    else if (TJr == 2)
        hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 1, 2, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    else
        hipLaunchKernelGGL((mlp_wgrad_fast_kernel<MODE, 1, 1, BSRC>), grid, dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, rpb);
    if (scratch) wgrad_reduce((int)gx, bs.pstride, (long)wrow0 * cout, (long)(wrow0 + cin) * cout, scratch, dw, st);
    return true;
}

// Takes the launch when the shape fits; mode 1: cin = the feature channels of the GATHER input (d.c).
bool wgrad_fast_launch(int mode, const MlpIn &d, long rows, int cin, int cout, const float *dz, const BnSrc &bs, int bsrc, float *dw,
                       hipStream_t st, float *scratch)
{
    if (cin % 64 != 0 || cout % 64 != 0 || rows <= 0 || rows >= (1L << 31)) return false;
    auto al = [](const void *p) { return ((uintptr_t)p % 16) == 0; };
    if (!al(dw) || (bsrc == 0 && !al(dz)) || (bsrc != 0 && (!al(bs.z) || !al(bs.coef))) || ((bsrc == 1 || bsrc == 4) && !al(bs.da)) ||
        (bsrc == 2 && (!al(bs.gout) || !al(bs.argmax))))
        return false;
    if (mode == 0) {
        if (!al(d.x) || (d.in_scale && (!al(d.in_scale) || !al(d.in_shift)))) return false;
        if (bsrc == 0) return launch<0, 0>(d, rows, cin, cout, dz, bs, dw, st, scratch);
        if (bsrc == 1) return launch<0, 1>(d, rows, cin, cout, dz, bs, dw, st, scratch);
        if (bsrc == 3) return launch<0, 3>(d, rows, cin, cout, dz, bs, dw, st, scratch);
        return launch<0, 2>(d, rows, cin, cout, dz, bs, dw, st, scratch);
    }
    if (mode == 3) { // ASSEMBLED first layer below: x rebuilt from geo + P (votenet_assembled_wgrad_bn)
        if (!al(d.geo) || !al(d.ptab) || !al(d.wx) || (bsrc != 1 && bsrc != 4)) return false;
        if (d.in_scale && (!al(d.in_scale) || !al(d.in_shift))) return false;
        if (bsrc == 4) return bs.wh != nullptr && rows % 32 == 0 && launch<3, 4>(d, rows, cin, cout, dz, bs, dw, st, scratch);
        return launch<3, 1>(d, rows, cin, cout, dz, bs, dw, st, scratch);
    }
    if (mode == 2) { // NARROW first layer below: x rebuilt from u8 (votenet_narrow_wgrad_bn)
        if (!al(d.u8) || !al(d.w0) || (d.b0 && !al(d.b0)) || d.k0 < 1 || d.k0 > 8 || (bsrc != 1 && bsrc != 4)) return false;
        if (d.in_scale && (!al(d.in_scale) || !al(d.in_shift))) return false;
        if (bsrc == 4) return bs.wh != nullptr && rows % 32 == 0 && launch<2, 4>(d, rows, cin, cout, dz, bs, dw, st, scratch);
        return launch<2, 1>(d, rows, cin, cout, dz, bs, dw, st, scratch);
    }
    if (d.c != cin || !al(d.feat) || bsrc != 0) return false;
    return launch<1, 0>(d, rows, cin, cout, dz, bs, dw, st, scratch);
}

} // namespace votenet

extern "C" void votenet_debug_wgrad_workgroups(int n) { VN_DEBUG_GATE(); votenet::g_wgrad_fast_wgs = n > 0 ? n : 0; } // tuning hook
extern "C" void votenet_debug_wgrad_bf3(int on) { VN_DEBUG_GATE(); votenet::g_wgrad_bf3 = on ? 1 : 0; } // 0: the fp32 MFMA form of every weight gradient
