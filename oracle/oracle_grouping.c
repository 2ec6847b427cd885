/*
 * oracle_grouping.c -- CPU restatement of tf_ops/grouping (ball query, group, group-grad).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Pinned against the reference's compiled CPU
 * twins (oracle/_ref/libref_grouping.so) by tests/test_oracle_vs_ref.py and the golden
 * fixtures under tests/golden/.
 */
#include "oracle.h"
#include <math.h>
#include <stddef.h>

/*
 * query_ball_point_gpu, tf_grouping_g.cu:3-36 (CPU twin: test/query_ball_point.cpp:19-47,
 * which lacks pts_cnt).  Candidates visited in ascending k; hit iff
 * max(sqrtf(s),1e-20f) < radius (:24-25); first hit fills all nsample slots (:26-29);
 * stop once nsample hits were found (:16-17); pts_cnt = #hits found (:34).
 * A query with no hit leaves its idx row untouched in the reference (uninitialised
 * memory); the oracle writes 0 there and the build never relies on it.
 */
void oracle_query_ball_point(int b, int n, int m, float radius, int nsample,
                             const float *xyz1, const float *xyz2, int *idx, int *pts_cnt)
{
    for (int i = 0; i < b; ++i) {
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp): independent queries */
        for (int j = 0; j < m; ++j) {
            int cnt = 0;
            for (int l = 0; l < nsample; ++l) idx[j * nsample + l] = 0;
            for (int k = 0; k < n; ++k) {
                if (cnt == nsample) break;
                float x2 = xyz2[j * 3 + 0];
                float y2 = xyz2[j * 3 + 1];
                float z2 = xyz2[j * 3 + 2];
                float x1 = xyz1[k * 3 + 0];
                float y1 = xyz1[k * 3 + 1];
                float z1 = xyz1[k * 3 + 2];
                float s = sqrtf((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1));
                float d = s > 1e-20f ? s : 1e-20f;
                /* NOT a restatement: CUDA's max(NaN, 1e-20f) is 1e-20f, so a point (or centre) with a NaN coordinate is a "hit"
                 * of every pair in the reference -- an artefact of fmaxf.  The build defines a non-finite distance as no hit
                 * (the device kernels decide s < T(r), false for NaN; DESIGN_HISTORY.md 2); finite clouds are untouched. */
                if (s != s) continue;
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) idx[j * nsample + l] = k;
                    idx[j * nsample + cnt] = k;
                    cnt += 1;
                }
            }
            if (pts_cnt) pts_cnt[j] = cnt;
        }
        xyz1 += (size_t)n * 3;
        xyz2 += (size_t)m * 3;
        idx += (size_t)m * nsample;
        if (pts_cnt) pts_cnt += m;
    }
}

/* group_point_gpu, tf_grouping_g.cu:40-57 */
void oracle_group_point(int b, int n, int c, int m, int nsample,
                        const float *points, const int *idx, float *out)
{
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                int ii = idx[j * nsample + k];
                for (int l = 0; l < c; ++l)
                    out[(size_t)j * nsample * c + (size_t)k * c + l] = points[(size_t)ii * c + l];
            }
        points += (size_t)n * c;
        idx += (size_t)m * nsample;
        out += (size_t)m * nsample * c;
    }
}

/* group_point_grad_gpu, tf_grouping_g.cu:61-78 (atomics: order unspecified there;
 * the oracle sums in ascending (j,k), as test/query_ball_point.cpp:70-84 does) */
void oracle_group_point_grad(int b, int n, int c, int m, int nsample,
                             const float *grad_out, const int *idx, float *grad_points)
{
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                int ii = idx[j * nsample + k];
                for (int l = 0; l < c; ++l)
                    grad_points[(size_t)ii * c + l] += grad_out[(size_t)j * nsample * c + (size_t)k * c + l];
            }
        idx += (size_t)m * nsample;
        grad_out += (size_t)m * nsample * c;
        grad_points += (size_t)n * c;
    }
}

/* sample_and_group, utils.py:50-57: grouped_xyz - centre, then concat [dxyz, feats] */
void oracle_group_concat(int b, int n, int c, int m, int nsample, const float *xyz,
                         const float *new_xyz, const float *points, const int *idx, float *out)
{
    int co = 3 + c;
    for (int i = 0; i < b; ++i) {
#pragma omp parallel for schedule(static) /* only in liboracle_omp.so (-fopenmp): independent queries */
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                int ii = idx[j * nsample + k];
                float *o = out + ((size_t)j * nsample + k) * co;
                o[0] = xyz[(size_t)ii * 3 + 0] - new_xyz[j * 3 + 0];
                o[1] = xyz[(size_t)ii * 3 + 1] - new_xyz[j * 3 + 1];
                o[2] = xyz[(size_t)ii * 3 + 2] - new_xyz[j * 3 + 2];
                for (int l = 0; l < c; ++l) o[3 + l] = points[(size_t)ii * c + l];
            }
        xyz += (size_t)n * 3;
        new_xyz += (size_t)m * 3;
        if (points) points += (size_t)n * c;
        idx += (size_t)m * nsample;
        out += (size_t)m * nsample * co;
    }
}
