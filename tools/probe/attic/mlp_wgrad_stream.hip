// mlp_wgrad.hip -- weight gradient dW (cin x cout) += X^T (cin x rows) * dZ (rows x cout) of the grouped-point MLP,
// streaming form for gfx950: NO LDS, NO barriers.
//
// v_mfma_f32_32x32x2_f32 wants A[i][k] in lane (i = lane&31, k = lane>>5) and B[k][j] in lane (j = lane&31,
// k = lane>>5).  With the contraction index k = row and i / j = input / output channel, a half-wave reads 32
// consecutive channels of ONE row: both operands come straight from global memory as perfectly coalesced 128-byte
// dword loads, already in MFMA layout.  A lane's channels never change, so everything per-channel lives in
// registers: the previous layer's folded BN scale/shift (+ReLU) on X, and the BatchNorm-backward coefficients that
// rebuild dz from (da | pooled gout, z) (struct BnSrc) -- the dz tensor need not exist in memory.
//
// A wave owns a 64 x 64 block of dW (2 x 2 MFMA tiles, 64 accumulator registers) over a contiguous range of rows;
// the four waves of a workgroup tile WI x WJ such blocks over the SAME rows (operand re-reads hit the CU's L1) and
// split the rows WR = 4/(WI*WJ) ways.  Loads are software-pipelined one 2*STEPS-row block ahead in registers.
// Partial blocks are added to dW with fp32 atomics (summation order unspecified, as in the reference's gradients).
#include "mlp_types.h"
#include <cstdlib>

namespace votenet {

// MODE 0: X dense (rows x cin), optional relu(x*scale+shift).  MODE 1: X = feat[b, idx[row], :] (the feature block of
// the sample_and_group concat; its 3 xyz columns go through wgrad_narrow_kernel), dW rows offset by 3.
//
// Lane l of a wave owns input channels i0 + 2*(l&31) + {0,1} and output channels j0 + 2*(l&31) + {0,1}: one 8-byte
// load per operand, row and lane (a half-wave reads 256 contiguous bytes); MFMA tile (a, b) pairs channel parity a
// of X with parity b of dZ, i.e. tile row mi <-> input channel i0 + 2*mi + a, tile column <-> j0 + 2*(l&31) + b.
// Pipeline: a ring of P register slots, one per MFMA k-step (2 rows); the loads of step s + P are issued right
// after step s has been consumed, so P steps (P * 4 MFMAs = P * 256 matrix-pipe cycles) of loads are in flight.
template <int MODE, int BSRC>
__global__ __launch_bounds__(256) void wgrad_stream_kernel(MlpIn in, long rows, int cin, int cout, const float *__restrict__ dz,
                                                           BnSrc bs, float *__restrict__ dw, int wi, int wj, long rows_per_wave)
{
    constexpr int P = (BSRC == 2) ? 8 : (BSRC == 1 ? 10 : 16);
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = 4 / (wi * wj);
    const int tile = wv % (wi * wj), rsplit = wv / (wi * wj);
    const int i0 = (blockIdx.y * wi + tile / wj) * 64; // input-channel base of this wave's block
    const int j0 = (blockIdx.z * wj + tile % wj) * 64; // output-channel base
    const long r0 = ((long)blockIdx.x * wr + rsplit) * rows_per_wave;
    const long r1 = r0 + rows_per_wave < rows ? r0 + rows_per_wave : rows;
    if (r0 >= rows || i0 >= cin || j0 >= cout) return; // wave-uniform
    const int nrow = (int)(r1 - r0);                  // rows of this wave (< 2^31)

    const int ic = i0 + 2 * l31, jc = j0 + 2 * l31;
    const bool affine = (MODE == 0) && in.in_scale != nullptr;
    float2 isc = make_float2(1.f, 1.f), ish = make_float2(0.f, 0.f);
    if (affine) {
        isc = *reinterpret_cast<const float2 *>(in.in_scale + ic);
        ish = *reinterpret_cast<const float2 *>(in.in_shift + ic);
    }
    const float floor_x = (affine && in.in_relu) ? 0.0f : -__builtin_inff();
    float2 kA, kB, kC, kS, kH;
    kA = kB = kC = kS = kH = make_float2(0.f, 0.f);
    if (BSRC != 0) {
        kA = *reinterpret_cast<const float2 *>(bs.coef + jc);
        kB = *reinterpret_cast<const float2 *>(bs.coef + cout + jc);
        kC = *reinterpret_cast<const float2 *>(bs.coef + 2 * cout + jc);
        kS = *reinterpret_cast<const float2 *>(bs.coef + 3 * cout + jc);
        kH = *reinterpret_cast<const float2 *>(bs.coef + 4 * cout + jc);
    }
    // wave-uniform bases at the wave's first row; lanes carry 32-bit element offsets
    const float *xb = (MODE == 0) ? in.x + (size_t)r0 * cin + ic : nullptr;
    const float *zb = (BSRC == 0 ? dz : bs.z) + (size_t)r0 * cout + jc;
    const float *gb = (BSRC == 1) ? bs.da + (size_t)r0 * cout + jc : nullptr;
    const float *featb = nullptr;
    const int *idxb = nullptr;
    if (MODE == 1) {
        const unsigned scene = (unsigned)r0 / ((unsigned)in.m * (unsigned)in.nsample); // the wave's rows lie in one scene
        featb = in.feat + (size_t)scene * in.n * in.c + ic;
        idxb = in.idx + r0;
    }

    float2 sx[P], sz[P], sg[P];
    int2 sm[P];
    int sid[P]; // MODE 1: idx of the row that slot s will load NEXT (one ring turn ahead of the x load)
    // local row of (ring turn t, slot s) for this half-wave
    auto lrow = [&](int t, int s) { return (t * P + s) * 2 + kh; };
    auto clampr = [&](int lr) { return lr < nrow ? lr : nrow - 1; };
    auto load_idx = [&](int t, int s) {
        if (MODE == 1) sid[s] = idxb[clampr(lrow(t, s))];
    };
    auto load_slot = [&](int t, int s) {
        const int lr = clampr(lrow(t, s)); // rows past the end re-read the last row; their dz is zeroed at use
        if (MODE == 0) sx[s] = *reinterpret_cast<const float2 *>(xb + (size_t)(unsigned)lr * cin);
        else sx[s] = *reinterpret_cast<const float2 *>(featb + (size_t)(unsigned)sid[s] * in.c);
        sz[s] = *reinterpret_cast<const float2 *>(zb + (size_t)(unsigned)lr * cout);
        if (BSRC == 1) sg[s] = *reinterpret_cast<const float2 *>(gb + (size_t)(unsigned)lr * cout);
        if (BSRC == 2) {
            const unsigned r = (unsigned)(r0 + lr);
            const unsigned grp = bs.pool_shift >= 0 ? r >> bs.pool_shift : r / (unsigned)bs.pool_k;
            sg[s] = *reinterpret_cast<const float2 *>(bs.gout + (size_t)grp * cout + jc);
            sm[s] = *reinterpret_cast<const int2 *>(bs.argmax + (size_t)grp * cout + jc);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[a][b][e] = 0.0f;

#pragma unroll
    for (int s = 0; s < P; s++) load_idx(0, s);
#pragma unroll
    for (int s = 0; s < P; s++) {
        load_slot(0, s);
        load_idx(1, s);
        // same issue order as inside the loop: the compiler's s_waitcnt insertion merges the loop header's two
        // predecessors conservatively, and a prologue that issued slot 0 last would cost vmcnt(0) on every turn
        __builtin_amdgcn_sched_barrier(0);
    }
    const int nturn = (nrow + 2 * P - 1) / (2 * P);
    for (int t = 0; t < nturn; t++) {
#pragma unroll
        for (int s = 0; s < P; s++) {
            const int lr = lrow(t, s);
            const bool ok = lr < nrow;
            float fa[2] = {sx[s].x, sx[s].y};
            float fb[2] = {sz[s].x, sz[s].y};
            // folded BN (+ReLU) of the layer below, branch-free: identity is x*1+0 with a floor of -inf
            fa[0] = fmaxf(fa[0] * isc.x + ish.x, floor_x);
            fa[1] = fmaxf(fa[1] * isc.y + ish.y, floor_x);
            if (BSRC != 0) {
                float g0 = sg[s].x, g1 = sg[s].y;
                if (BSRC == 2) {
                    const unsigned r = (unsigned)(r0 + lr);
                    const int ro = bs.pool_shift >= 0 ? (int)(r & (unsigned)(bs.pool_k - 1)) : (int)(r % (unsigned)bs.pool_k);
                    g0 = (sm[s].x == ro) ? g0 : 0.0f;
                    g1 = (sm[s].y == ro) ? g1 : 0.0f;
                }
                if (bs.relu) {
                    if (!(fb[0] * kS.x + kH.x > 0.0f)) g0 = 0.0f;
                    if (!(fb[1] * kS.y + kH.y > 0.0f)) g1 = 0.0f;
                }
                fb[0] = kA.x * g0 + kB.x + kC.x * fb[0];
                fb[1] = kA.y * g1 + kB.y + kC.y * fb[1];
            }
            fb[0] = ok ? fb[0] : 0.0f; // padding rows contribute nothing
            fb[1] = ok ? fb[1] : 0.0f;
            // refill this slot for the next ring turn (its values now live in fa / fb)
            load_slot(t + 1, s);
            load_idx(t + 2, s);
            __builtin_amdgcn_sched_barrier(0); // keep the refill here: the scheduler must not sink it behind later steps
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C/D layout: column = lane&31, row mi = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int wrow0 = (MODE == 1) ? 3 : 0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int mi = (e & 3) + 8 * (e >> 2) + 4 * kh;
                unsafeAtomicAdd(&dw[(size_t)(wrow0 + i0 + 2 * mi + a) * cout + jc + b], acc[a][b][e]);
            }
}

template <int MODE, int BSRC>
static bool launch(const MlpIn &d, long rows, int cin, int cout, const float *dz, const BnSrc &bs, float *dw, hipStream_t st)
{
    const int ti = cin / 64, tj = cout / 64; // 64 x 64 blocks of dW
    const int wi = ti >= 2 ? 2 : 1, wj = tj >= 2 ? 2 : 1;
    const int wr = 4 / (wi * wj);
    const int gy = (ti + wi - 1) / wi, gz = (tj + wj - 1) / wj;
    constexpr int RB = 2 * ((BSRC == 2) ? 8 : (BSRC == 1 ? 10 : 16)); // rows per ring turn
    // about 3 waves per SIMD over the whole launch, at least 4 ring turns per wave
    static const long tune_waves = getenv("VOTENET_WGRAD_WAVES") ? atol(getenv("VOTENET_WGRAD_WAVES")) : 3072;
    long want_waves = tune_waves / ((long)gy * gz * wi * wj);
    if (want_waves < 1) want_waves = 1;
    long rpw = (rows + want_waves - 1) / want_waves;
    if (rpw < 4 * RB) rpw = 4 * RB;
    rpw = (rpw + RB - 1) / RB * RB;
    if (MODE == 1) {
        // a wave's rows must lie in one scene: rows per wave = (rows per scene) / q
        const long g = (long)d.m * d.nsample;
        long q = (g + rpw - 1) / rpw, best = 0;
        for (; q <= g / RB; q++)
            if (g % q == 0 && (g / q) % RB == 0) {
                best = g / q;
                break;
            }
        if (best == 0) return false;
        rpw = best;
    }
    const long nsplit = (rows + rpw - 1) / rpw;
    const unsigned gx = (unsigned)((nsplit + wr - 1) / wr);
    hipLaunchKernelGGL((wgrad_stream_kernel<MODE, BSRC>), dim3(gx, gy, gz), dim3(256), 0, st, d, rows, cin, cout, dz, bs, dw, wi, wj,
                       rpw);
    return true;
}

// Takes the launch when the shape fits: cin, cout multiples of 64 (MODE 1: cin = feature channels), rows < 2^31,
// MODE 1: m*nsample a multiple of 16 so that a pipeline block never straddles two scenes.
bool wgrad_stream_launch(int mode, const MlpIn &d, long rows, int cin, int cout, const float *dz, const BnSrc &bs, int bsrc,
                         float *dw, hipStream_t st)
{
    if (cin % 64 != 0 || cout % 64 != 0 || rows >= (1L << 31) || rows <= 0) return false;
    if (mode == 1 && d.c != cin) return false;
    if (mode == 0) {
        if (bsrc == 0) return launch<0, 0>(d, rows, cin, cout, dz, bs, dw, st);
        if (bsrc == 1) return launch<0, 1>(d, rows, cin, cout, dz, bs, dw, st);
        return launch<0, 2>(d, rows, cin, cout, dz, bs, dw, st);
    }
    if (bsrc == 0) return launch<1, 0>(d, rows, cin, cout, dz, bs, dw, st);
    if (bsrc == 1) return launch<1, 1>(d, rows, cin, cout, dz, bs, dw, st);
    return launch<1, 2>(d, rows, cin, cout, dz, bs, dw, st);
}

} // namespace votenet
