/*
 * oracle_mlp.c -- CPU statement of the grouped-point MLP (utils.py:125-132,149-155,286-293).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED: the arithmetic of Conv2D / BatchNorm / BNReLU / FullyConnected lives in
 * Tensorpack + TensorFlow 1.x, which are not part of the reference tree (README.md:21
 * names them without versions) and are absent from this image.  The call sites
 * (utils.py:126-127,152-154,291-292, model.py:56) fix only the structure:
 *     1x1 conv (== row-wise linear layer, NHWC) -> BatchNorm (training: batch statistics)
 *     -> ReLU, three times, then max over the K neighbours of each group.
 * This file defines the numbers the HIP path is checked against:
 *   linear : z[r,o] = bias[o] + sum_k x[r,k]*w[k,o], accumulated in ascending k with one
 *            rounding per product-add (fmaf) -- the numerics of v_mfma_f32_32x32x2_f32
 *   stats  : mean and biased variance over rows, accumulated in double, rounded to fp32
 *   bn+relu: y = max(0, gamma*(z-mean)*rsqrt(var+eps)+beta) evaluated in fp32
 */
#include "oracle.h"
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

void oracle_linear(long rows, int cin, int cout, const float *x, const float *w, const float *bias, float *z)
{
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp) */
    for (long r = 0; r < rows; r++) {
        const float *xr = x + (size_t)r * cin;
        float *zr = z + (size_t)r * cout;
        for (int o = 0; o < cout; o++) zr[o] = 0.0f;
        for (int k = 0; k < cin; k++) {
            float a = xr[k];
            const float *wk = w + (size_t)k * cout;
            for (int o = 0; o < cout; o++) zr[o] = fmaf(a, wk[o], zr[o]);
        }
        if (bias)
            for (int o = 0; o < cout; o++) zr[o] = zr[o] + bias[o];
    }
}

/* Sums in double over fixed blocks of 4096 rows, the block sums added in block order: the same numbers whatever the number of
 * threads (liboracle_omp.so gives each block to one thread), row-contiguous reads. */
#define BN_BLOCK 4096
void oracle_bn_stats(long rows, int c, const float *z, float *mean, float *var)
{
    const long nblk = (rows + BN_BLOCK - 1) / BN_BLOCK;
    double *part = (double *)malloc(sizeof(double) * (size_t)(nblk > 0 ? nblk : 1) * (size_t)c);
    double *mu = (double *)malloc(sizeof(double) * (size_t)c);
    for (int pass = 0; pass < 2; pass++) {
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp) */
        for (long blk = 0; blk < nblk; blk++) {
            double *p = part + (size_t)blk * c;
            for (int o = 0; o < c; o++) p[o] = 0;
            const long r1 = (blk + 1) * BN_BLOCK < rows ? (blk + 1) * BN_BLOCK : rows;
            for (long r = blk * BN_BLOCK; r < r1; r++) {
                const float *zr = z + (size_t)r * c;
                if (pass == 0)
                    for (int o = 0; o < c; o++) p[o] += zr[o];
                else
                    for (int o = 0; o < c; o++) {
                        double d = zr[o] - mu[o];
                        p[o] += d * d;
                    }
            }
        }
        for (int o = 0; o < c; o++) {
            double s = 0;
            for (long blk = 0; blk < nblk; blk++) s += part[(size_t)blk * c + o];
            if (pass == 0) {
                mu[o] = s / (double)rows;
                mean[o] = (float)mu[o];
            } else
                var[o] = (float)(s / (double)rows);
        }
    }
    free(part);
    free(mu);
}

void oracle_bn_relu(long rows, int c, const float *z, const float *mean, const float *var,
                    const float *gamma, const float *beta, float eps, int relu, float *y)
{
    float *scale = (float *)malloc(sizeof(float) * (size_t)c * 2), *shift = scale + c;
    for (int o = 0; o < c; o++) {
        scale[o] = gamma[o] / sqrtf(var[o] + eps);
        shift[o] = beta[o] - mean[o] * scale[o];
    }
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp) */
    for (long r = 0; r < rows; r++)
        for (int o = 0; o < c; o++) {
            float v = z[(size_t)r * c + o] * scale[o] + shift[o];
            if (relu && !(v > 0.0f)) v = 0.0f;
            y[(size_t)r * c + o] = v;
        }
    free(scale);
}

/* utils.py:132 reduce_max over the nsample axis */
void oracle_max_over_k(long groups, int k, int c, const float *y, float *out)
{
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp) */
    for (long g = 0; g < groups; g++)
        for (int o = 0; o < c; o++) {
            float m = y[((size_t)g * k) * c + o];
            for (int j = 1; j < k; j++) {
                float v = y[((size_t)g * k + j) * c + o];
                if (v > m) m = v;
            }
            out[(size_t)g * c + o] = m;
        }
}
