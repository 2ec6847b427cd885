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

void oracle_bn_stats(long rows, int c, const float *z, float *mean, float *var)
{
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp) */
    for (int o = 0; o < c; o++) {
        double s = 0;
        for (long r = 0; r < rows; r++) s += z[(size_t)r * c + o];
        double mu = s / (double)rows;
        double v = 0;
        for (long r = 0; r < rows; r++) {
            double d = z[(size_t)r * c + o] - mu;
            v += d * d;
        }
        mean[o] = (float)mu;
        var[o] = (float)(v / (double)rows);
    }
}

void oracle_bn_relu(long rows, int c, const float *z, const float *mean, const float *var,
                    const float *gamma, const float *beta, float eps, int relu, float *y)
{
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp) */
    for (int o = 0; o < c; o++) {
        float scale = gamma[o] / sqrtf(var[o] + eps);
        float shift = beta[o] - mean[o] * scale;
        for (long r = 0; r < rows; r++) {
            float v = z[(size_t)r * c + o] * scale + shift;
            if (relu && !(v > 0.0f)) v = 0.0f;
            y[(size_t)r * c + o] = v;
        }
    }
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
