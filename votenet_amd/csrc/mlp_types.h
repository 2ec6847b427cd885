// mlp_types.h -- device-side descriptors shared by the grouped-point MLP kernels (mlp*.hip).
#pragma once
#include "common.h"

namespace votenet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The implicit input matrix of a layer (mirror of struct votenet_mlp_input, include/votenet_hip.h)
struct MlpIn {
    // DENSE
    const float *x;
    const float *in_scale;
    const float *in_shift;
    int in_relu;
    // GATHER
    const float *xyz;
    const float *new_xyz;
    const float *feat;
    const int *idx;
    int n, m, nsample, c;
    // NARROW first layer (narrow.hip): x[r,:] = z0[r,:] = narrow_z(u8[r], W0, b0), rows of 8 floats; then in_scale / in_shift / in_relu
    const float *u8, *w0, *b0;
    int k0; // 3..8 rows of W0 (k0 x cin), the rest of u is zero
    // ASSEMBLED first layer (assemble.hip): x[r,:] = z0[r,:] = assembled_z(P[prow(r),:], geo[r], wx); then in_scale / in_shift / in_relu
    const float *geo, *ptab, *wx;
};

// z0[r,c] of a narrow first layer: ONE fma chain in a fixed order.  Every kernel that needs z0 rebuilds it with this function, so the
// forward and the backward pass see bit-identical values (identical ReLU masks); zero-padded u / W0 entries add exactly nothing.
__device__ __forceinline__ float narrow_z(const float (&u)[8], const float (&w)[8], float b)
{
    float z = b;
#pragma unroll
    for (int d = 0; d < 8; d++) z = __builtin_fmaf(u[d], w[d], z);
    return z;
}

// Two channels at once: the same fma chain per channel as ONE v_pk_fma_f32 per u[d] (bit-identical to narrow_z: a packed fma is the same
// IEEE fma in each half).  The broadcast operand is written FIRST (op_sel lands on src0: the packed-f32 hazard of DESIGN_HISTORY, round 5,
// concerns src1).  The operand loaders are bound by their vector-instruction count: this halves the 64 fmas per 16-deep slab and thread.
__device__ __forceinline__ f32x2 narrow_z2(const float (&u)[8], const float (&wa)[8], const float (&wb)[8], float ba, float bb)
{
    f32x2 z = {ba, bb};
#pragma unroll
    for (int d = 0; d < 8; d++) z = __builtin_elementwise_fma(f32x2{u[d], u[d]}, f32x2{wa[d], wb[d]}, z);
    return z;
}

// How a backward GEMM obtains its dz operand (rows x cout):
//   da == gout == NULL: dz read from memory;
//   da   : dz = A*g + B + C*z with g = da masked by [z*S+H > 0]                    (BatchNorm backward folded in)
//   gout : the same with g = gout[row/k] where row%k == argmax[row/k]              (max-pooled upstream)
// coef = [A|B|C|S|H], 5*cout floats (votenet_bn_backward_coef)
struct BnSrc {
    const float *da, *gout;
    const int *argmax;
    int pool_k, pool_shift; // pool_shift = log2(pool_k) when pool_k is a power of two, else -1
    const float *z, *coef;
    int relu;
    // deterministic weight gradients: when part != NULL a workgroup STORES its partial dW tile at
    // part[blockIdx.x * pstride + (the element's offset inside dw)] instead of adding it to dw with atomics;
    // wgrad_reduce_kernel then adds the blockIdx.x slices to dw in ascending order (one summation order -> bit-reproducible)
    float *part;
    long pstride;
    const float *wh; // BSRC 4 (piece layout, half.hip): da holds totals; the affine part B + C z of row 16 q counts wh[q] times
    const int *nh_dev; // piece layout with the count on the device: `rows` is an upper bound, the kernel stops at 16 * nh_dev[0] rows
};

// ---- the PIECE layout of a set-abstraction level (half.hip): a ball's 64 slots in pieces of kPiece rows, the all-copy pieces dropped ----
constexpr int kPiece = 16;              // rows per piece (the MFMA tiles pool per 16 rows: elements e < 8 / e >= 8 of a 32 x 32 tile)
constexpr int kBallPieces = 64 / kPiece; // pieces of a full ball
constexpr int kTilePieces = 128 / kPiece; // pieces per 128-row GEMM tile: the number of pieces of a level is a multiple of this

// z0[r,c] of a first SA layer assembled where it is consumed (assemble.hip): p = P[prow(r), c], g = geo[r] = (dx, dy, dz, bits(prow)),
// w0..w2 = W[0:3][c].  ONE fma chain, the same in every kernel, so all consumers see bit-identical values (identical ReLU masks).
__device__ __forceinline__ float assembled_z(float p, const float4 &g, float w0, float w1, float w2)
{
    return __builtin_fmaf(g.z, w2, __builtin_fmaf(g.y, w1, __builtin_fmaf(g.x, w0, p)));
}
// Two channels at once (bit-identical per channel; broadcast operand first: see narrow_z2)
__device__ __forceinline__ f32x2 assembled_z2(float pa, float pb, const float4 &g, float w0a, float w0b, float w1a, float w1b, float w2a, float w2b)
{
    f32x2 z = __builtin_elementwise_fma(f32x2{g.x, g.x}, f32x2{w0a, w0b}, f32x2{pa, pb});
    z = __builtin_elementwise_fma(f32x2{g.y, g.y}, f32x2{w1a, w1b}, z);
    return __builtin_elementwise_fma(f32x2{g.z, g.z}, f32x2{w2a, w2b}, z);
}

// ---- bf16 x 3 split operands (BF3: mlp_fast.hip, the Gram kernel of pool_bwd.hip) ---------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 -> three bf16 pieces, each rounded to nearest even: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid).  Both subtractions
// are exact (a float minus its own rounding), and the second remainder has at most 8 significant bits, so lo is exact as well:
// x = hi + mid + lo EXACTLY, with |mid| <= 2^-8 |x| and |lo| <= 2^-16 |x|, signs as the roundings fall.  The pieces of x go to the
// low halves of h / m / l and those of y to the high halves (v_cvt_pk_bf16_f32 converts and packs two values in one instruction).
// A product x*w is then the six bf16 MFMA terms hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi accumulated in fp32: what is left
// out (mid*lo, lo*mid, lo*lo) is at most 2^-23 of |x*w| -- the size of the fp32 rounding of the product itself -- and unbiased.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rne(float x, float y)
{
    const bf16x2 v = {(__bf16)x, (__bf16)y};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split3(float x, float y, unsigned &h, unsigned &m, unsigned &l)
{
    h = pack_bf16_rne(x, y);
    const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16_rne(rx, ry);
    const float sx = rx - __uint_as_float(m << 16), sy = ry - __uint_as_float(m & 0xffff0000u);
    l = pack_bf16_rne(sx, sy);
}

// H2 (round 6): the same idea on fp16.  x = hi + lo with hi = rne16(x), lo = rne16(x - hi): 22 significant bits, a product is the THREE
// fp16 MFMA terms hi*hi + hi*lo + lo*hi (v_mfma_f32_32x32x16_f16, fp32 accumulate; gfx950 honours fp16 subnormals in MFMA inputs:
// tools/probe/src/mfma_f16_denorm.hip, profiles/r06_mfma_f16_denorm.txt -- the lo piece of an O(0.1) value IS a subnormal).  What is
// left out (lo*lo and the 2^-22 the split itself drops) is below the fp32 rounding of a 128-term dot product: measured error against
// float64 equal to bf16 x 3's and the fp32 chain's (same probe).  Half the MFMAs and a third of the staging instructions of split3 --
// but fp16's RANGE: |x| must stay below 65504 and values below 2^-24 vanish, so this form serves FORWARD operands only (activations
// behind a BatchNorm, coordinates, weights); gradients stay on bf16 x 3.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2(float x, float y, unsigned &h, unsigned &l)
{
    const f16x2 hv = {(_Float16)x, (_Float16)y};
    h = __builtin_bit_cast(unsigned, hv);
    // the residual x - hi as ONE v_fma_mix_f32 (-hi * 1 + x, exact: hi is read as fp16 straight out of the packed register) instead of
    // v_cvt_f32_f16 + v_sub_f32: the staging arithmetic is what bounds these kernels.  (The compiler folds fmaf(ext(hi), -1, x) back
    // into the subtraction, hence the asm.)
    float rx, ry;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(h), "v"(x));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(h), "v"(y));
    const f16x2 lv = {(_Float16)rx, (_Float16)ry};
    l = __builtin_bit_cast(unsigned, lv);
}

} // namespace votenet
