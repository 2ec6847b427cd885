/*
 * oracle_interpolate.c -- CPU restatement of tf_ops/3d_interpolation.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Pinned against oracle/_ref/libref_interpolate.so
 * (the reference's stand-alone interpolate.cpp) by tests/test_oracle_vs_ref.py.
 */
#include "oracle.h"
#include <stddef.h>

/*
 * threenn_cpu, tf_interpolate.cpp:60-103.  Squared distance evaluated in fp32 exactly as
 * written (:73), widened to double for the comparisons (exact widening, so the order is
 * the fp32 order); strict '<' cascade (:74-89) so equal distances keep the lower index
 * first; best* start at 1e40 (-> +inf when stored to float) with index 0 (:66-67).
 */
void oracle_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                     float *dist, int *idx)
{
    for (int i = 0; i < b; ++i) {
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp): independent queries */
        for (int j = 0; j < n; ++j) {
            float x1 = xyz1[j * 3 + 0];
            float y1 = xyz1[j * 3 + 1];
            float z1 = xyz1[j * 3 + 2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                float x2 = xyz2[k * 3 + 0];
                float y2 = xyz2[k * 3 + 1];
                float z2 = xyz2[k * 3 + 2];
                float df = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                double d = df;
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            dist[j * 3] = (float)best1; idx[j * 3] = besti1;
            dist[j * 3 + 1] = (float)best2; idx[j * 3 + 1] = besti2;
            dist[j * 3 + 2] = (float)best3; idx[j * 3 + 2] = besti3;
        }
        xyz1 += (size_t)n * 3;
        xyz2 += (size_t)m * 3;
        dist += (size_t)n * 3;
        idx += (size_t)n * 3;
    }
}

/* utils.py:279-282: dist=max(dist,1e-10); norm=sum(1/dist); weight=(1/dist)/norm.
 * The 3-element sum is taken left to right (TF's reduce_sum order is not verifiable
 * here; last-ulp differences are inside the 1e-5 feature tolerance). */
void oracle_three_nn_weights(int b, int n, const float *dist, float *weight)
{
    size_t rows = (size_t)b * n;
    for (size_t r = 0; r < rows; ++r) {
        float d0 = dist[r * 3 + 0], d1 = dist[r * 3 + 1], d2 = dist[r * 3 + 2];
        d0 = d0 > 1e-10f ? d0 : 1e-10f;
        d1 = d1 > 1e-10f ? d1 : 1e-10f;
        d2 = d2 > 1e-10f ? d2 : 1e-10f;
        float r0 = 1.0f / d0, r1 = 1.0f / d1, r2 = 1.0f / d2;
        float norm = (r0 + r1) + r2;
        weight[r * 3 + 0] = r0 / norm;
        weight[r * 3 + 1] = r1 / norm;
        weight[r * 3 + 2] = r2 / norm;
    }
}

/* threeinterpolate_cpu, tf_interpolate.cpp:107-127: (p1*w1 + p2*w2) + p3*w3 */
void oracle_three_interpolate(int b, int m, int c, int n, const float *points,
                              const int *idx, const float *weight, float *out)
{
    for (int i = 0; i < b; ++i) {
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp): independent queries */
        for (int j = 0; j < n; ++j) {
            float w1 = weight[j * 3], w2 = weight[j * 3 + 1], w3 = weight[j * 3 + 2];
            int i1 = idx[j * 3], i2 = idx[j * 3 + 1], i3 = idx[j * 3 + 2];
            for (int l = 0; l < c; ++l)
                out[(size_t)j * c + l] = points[(size_t)i1 * c + l] * w1 + points[(size_t)i2 * c + l] * w2 +
                                         points[(size_t)i3 * c + l] * w3;
        }
        points += (size_t)m * c;
        idx += (size_t)n * 3;
        weight += (size_t)n * 3;
        out += (size_t)n * c;
    }
}

/* threeinterpolate_grad_cpu, tf_interpolate.cpp:131-153 (ascending j, taps 1,2,3) */
void oracle_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out,
                                   const int *idx, const float *weight, float *grad_points)
{
    for (int i = 0; i < b; ++i) {
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp): independent queries */
        for (int j = 0; j < n; ++j) {
            float w1 = weight[j * 3], w2 = weight[j * 3 + 1], w3 = weight[j * 3 + 2];
            int i1 = idx[j * 3], i2 = idx[j * 3 + 1], i3 = idx[j * 3 + 2];
            for (int l = 0; l < c; ++l) {
                grad_points[(size_t)i1 * c + l] += grad_out[(size_t)j * c + l] * w1;
                grad_points[(size_t)i2 * c + l] += grad_out[(size_t)j * c + l] * w2;
                grad_points[(size_t)i3 * c + l] += grad_out[(size_t)j * c + l] * w3;
            }
        }
        grad_out += (size_t)n * c;
        idx += (size_t)n * 3;
        weight += (size_t)n * 3;
        grad_points += (size_t)m * c;
    }
}
