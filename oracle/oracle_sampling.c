/*
 * oracle_sampling.c -- CPU restatement of tf_ops/sampling (FPS, gather, scatter-add).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  The reference has no CPU implementation and no test of these ops; this follows
 * the device kernel line by line and is PINNED against that kernel itself: tf_sampling_g.cu includes nothing and calls no
 * CUDA runtime function, so hipcc compiles it for gfx950 where it lies (oracle/_ref/libref_sampling_gpu.so, oracle/Makefile)
 * and its outputs on MI355X are the fixtures tests/golden/ref_gpu_{fps,gather}.npz (every pick of the 8 x 20480 -> 2048
 * headline clouds among them: tests/test_oracle_ref_gpu_golden.py; live three-way comparison on the GPU box:
 * tests/test_gpu_reference_kernels.py).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define FPS_BLOCK 512 /* tf_sampling_g.cu:108 BlockSize; launcher :204 uses 512 threads */

/*
 * Literal simulation of farthestpointsamplingKernel (tf_sampling_g.cu:105-170):
 * 512 "threads", thread t scans k = t, t+512, ... with a strict '>' (:146), then the
 * 9-level shared-memory tree keeps the LEFT slot on ties (:153-163).
 *   - temp[] (running min distance) starts at 1e38 (:117-119)
 *   - idxs[0] = 0 (:114-116); centre coordinates are read from the original cloud (:127-129)
 *   - d = (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 in fp32, no FMA (:142); d2 = min(d,td) (:143)
 *   - m <= 0: nothing written (:106-107)
 */

/*
 * Non-finite coordinates.  NOT a restatement: with a NaN point the CUDA kernel's fminf / '>' leave it at its initial distance
 * 1e38, pick it at once and, with a NaN centre, freeze every running distance -- an artefact, not a behaviour anyone relies on.
 * The build DEFINES the case (DESIGN_HISTORY.md 2 "Non-finite coordinates", csrc/fps.hip fps_get): a point with a non-finite coordinate is read as a copy of
 * point 0 (so it is never sampled), point 0's own non-finite components are read as 0.  Finite clouds are untouched.
 * -> a sanitized copy of the batch, or NULL when every coordinate is finite.
 */
static int nonfinite_bits(float v)
{
    unsigned u;
    memcpy(&u, &v, 4);
    return (u & 0x7f800000u) == 0x7f800000u;
}
static float *fps_defined_cloud(int b, int n, const float *dataset)
{
    const size_t tot = (size_t)(b > 0 ? b : 0) * (size_t)(n > 0 ? n : 0) * 3;
    size_t i;
    for (i = 0; i < tot; i++)
        if (nonfinite_bits(dataset[i])) break;
    if (i == tot) return NULL;
    float *c = (float *)malloc(sizeof(float) * (tot ? tot : 1));
    memcpy(c, dataset, sizeof(float) * tot);
    for (int s = 0; s < b; s++) {
        float *p = c + (size_t)s * n * 3;
        for (int a = 0; a < 3 && n > 0; a++)
            if (nonfinite_bits(p[a])) p[a] = 0.0f;
        for (int k = 1; k < n; k++)
            if (nonfinite_bits(p[k * 3]) || nonfinite_bits(p[k * 3 + 1]) || nonfinite_bits(p[k * 3 + 2])) {
                p[k * 3] = p[0];
                p[k * 3 + 1] = p[1];
                p[k * 3 + 2] = p[2];
            }
    }
    return c;
}

void oracle_farthest_point_sample(int b, int n, int m, const float *dataset, int *idxs)
{
    if (m <= 0) return;
    float *defined = fps_defined_cloud(b, n, dataset);
    if (defined) dataset = defined;
    float *temp_all = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * (size_t)(b > 0 ? b : 1));
    /* only in liboracle_omp.so: the scenes are independent (the rounds of a scene are a dependent chain; a fork/join per
     * round costs more than the round) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b; i++) {
        float *temp = temp_all + (size_t)i * n;
        float dists[FPS_BLOCK];
        int dists_i[FPS_BLOCK];
        const float *pts = dataset + (size_t)i * n * 3;
        int old = 0;
        idxs[(size_t)i * m + 0] = old;
        for (int j = 0; j < n; j++) temp[j] = 1e38f;
        for (int j = 1; j < m; j++) {
            float x1 = pts[old * 3 + 0];
            float y1 = pts[old * 3 + 1];
            float z1 = pts[old * 3 + 2];
            for (int t = 0; t < FPS_BLOCK; t++) {
                int besti = 0;
                float best = -1;
                for (int k = t; k < n; k += FPS_BLOCK) {
                    float td = temp[k];
                    float x2 = pts[k * 3 + 0];
                    float y2 = pts[k * 3 + 1];
                    float z2 = pts[k * 3 + 2];
                    float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                    float d2 = (d < td) ? d : td; /* CUDA min(float,float) == fminf */
                    if (d2 != td) temp[k] = d2;
                    if (d2 > best) {
                        best = d2;
                        besti = k;
                    }
                }
                dists[t] = best;
                dists_i[t] = besti;
            }
            for (int u = 0; (1 << u) < FPS_BLOCK; u++) {
                for (int t = 0; t < (FPS_BLOCK >> (u + 1)); t++) {
                    int i1 = (t * 2) << u;
                    int i2 = (t * 2 + 1) << u;
                    if (dists[i1] < dists[i2]) {
                        dists[i1] = dists[i2];
                        dists_i[i1] = dists_i[i2];
                    }
                }
            }
            old = dists_i[0];
            idxs[(size_t)i * m + j] = old;
        }
    }
    free(temp_all);
    free(defined);
}

/*
 * Closed form of the same selection rule: winner = max d2; ties -> smallest (k mod 512),
 * then smallest k.  Lanes with no point (t >= n) hold (best=-1, besti=0) and can only
 * win when every real d2 is < -1, i.e. never for finite input.
 */
void oracle_farthest_point_sample_closed(int b, int n, int m, const float *dataset, int *idxs)
{
    if (m <= 0) return;
    float *defined = fps_defined_cloud(b, n, dataset);
    if (defined) dataset = defined;
    float *temp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < b; i++) {
        const float *pts = dataset + (size_t)i * n * 3;
        int old = 0;
        idxs[(size_t)i * m + 0] = old;
        for (int j = 0; j < n; j++) temp[j] = 1e38f;
        for (int j = 1; j < m; j++) {
            float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
            float best = -1;
            int besti = 0;
            int bestlane = 0;
            for (int k = 0; k < n; k++) {
                float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
                float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
                float td = temp[k];
                float d2 = (d < td) ? d : td;
                temp[k] = d2;
                int lane = k % FPS_BLOCK;
                if (d2 > best || (d2 == best && lane < bestlane)) {
                    best = d2;
                    besti = k;
                    bestlane = lane;
                }
            }
            old = besti;
            idxs[(size_t)i * m + j] = old;
        }
    }
    free(temp);
    free(defined);
}

/* gatherpointKernel, tf_sampling_g.cu:172-181 */
void oracle_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++) {
            int a = idx[(size_t)i * m + j];
            out[((size_t)i * m + j) * 3 + 0] = inp[((size_t)i * n + a) * 3 + 0];
            out[((size_t)i * m + j) * 3 + 1] = inp[((size_t)i * n + a) * 3 + 1];
            out[((size_t)i * m + j) * 3 + 2] = inp[((size_t)i * n + a) * 3 + 2];
        }
}

/* scatteraddpointKernel, tf_sampling_g.cu:183-192.  The reference adds with atomics
 * (order unspecified); the oracle adds in ascending j. */
void oracle_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++) {
            int a = idx[(size_t)i * m + j];
            inp_g[((size_t)i * n + a) * 3 + 0] += out_g[((size_t)i * m + j) * 3 + 0];
            inp_g[((size_t)i * n + a) * 3 + 1] += out_g[((size_t)i * m + j) * 3 + 1];
            inp_g[((size_t)i * n + a) * 3 + 2] += out_g[((size_t)i * m + j) * 3 + 2];
        }
}

/* liboracle_omp.so only: number of OpenMP threads for the `parallel for` loops (a no-op in liboracle.so) */
#ifdef _OPENMP
#include <omp.h>
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
#else
void oracle_set_threads(int n) { (void)n; }
#endif
