/*
 * oracle_variants.c -- CPU restatement of the PointNet++ ops the reference ships but VoteNet's model.py never reaches
 * (SURVEY.md section 8f rank 4): SelectionSort / kNN and ProbSample.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "oracle.h"
#include <stdlib.h>

/* tf_ops/grouping/tf_grouping_g.cu:83-123 (CPU twin: tf_ops/grouping/test/selection_sort.cpp:19-62).
 * out / outi start as a copy of the row / 0..n-1; k steps of selection sort: the FIRST position holding the minimum of
 * [s, n) (strict '<') is swapped into s.  The tail [k, n) is whatever the swaps left there. */
void oracle_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out)
{
    for (long row = 0; row < (long)b * m; row++) {
        const float *d = dist + row * n;
        float *v = out + row * n;
        int *id = outi + row * n;
        for (int s = 0; s < n; s++) {
            v[s] = d[s];
            id[s] = s;
        }
        for (int s = 0; s < k; s++) {
            int best = s;
            for (int t = s + 1; t < n; t++)
                if (v[t] < v[best]) best = t;
            if (best != s) {
                const float tv = v[best];
                v[best] = v[s];
                v[s] = tv;
                const int ti = id[best];
                id[best] = id[s];
                id[s] = ti;
            }
        }
    }
}

/* tf_ops/grouping/tf_grouping.py:61-63 (knn_point): dist[b,j,i] = reduce_sum((xyz1[b,i,:] - xyz2[b,j,:])**2, -1).
 * The channel sum is taken left to right (TensorFlow's reduction order for a 3-wide inner axis is not pinned by the
 * reference; this oracle defines it). */
void oracle_knn_dist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist)
{
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++)
            for (int p = 0; p < n; p++) {
                float s = 0.0f;
                for (int a = 0; a < c; a++) {
                    const float d = xyz1[((long)i * n + p) * c + a] - xyz2[((long)i * m + j) * c + a];
                    const float sq = d * d;
                    s = a == 0 ? sq : s + sq;
                }
                dist[((long)i * m + j) * n + p] = s;
            }
}

/* ---- ProbSample: tf_ops/sampling/tf_sampling_g.cu:7-104 (cumsumKernel, binarysearchKernel), launcher :197-200 ----
 * The running sum is a float sum whose ASSOCIATION is fixed by the kernel's scan tree; restated here as a recurrence
 * instead of the kernel's index loops:
 *   chunk  = 8192 consecutive elements (BlockSize*4, :8,14)
 *   group  = 4 consecutive elements; in-group prefixes p1=v1, p2=v1+v2, p3=p2+v3, p4=(v3+v4)+p2 (:19-32); a last, partial
 *            group is summed serially (:33-42).  G[g] = the group's total.
 *   T(block) for an aligned block of 2^u groups = T(left half) + T(right half)            (up-sweep, :45-55)
 *   E(q) = inclusive sum of the first q groups: q a power of two -> T([0,q)); otherwise, with low = lowest set bit of q,
 *          E(q) = T([q-low, q)) + E(q-low)                                                (down-sweep, :56-66)
 *   element prefix inside the chunk = p_i (first group) or p_i + E(g) (group g >= 1)      (:68-76)
 *   out = prefix + runningsum; the chunk total is carried with a compensated (Kahan) update (:78-84). */
static float tree_sum(const float *G, int lo, int len)
{
    if (len == 1) return G[lo];
    return tree_sum(G, lo, len / 2) + tree_sum(G, lo + len / 2, len / 2);
}
static float prefix_groups(const float *G, int q)
{
    const int low = q & -q;
    if (low == q) return tree_sum(G, 0, q);
    return tree_sum(G, q - low, low) + prefix_groups(G, q - low);
}

void oracle_cumsum(int b, int n, const float *inp, float *out)
{
    float *P = (float *)malloc(sizeof(float) * 8192);
    float *G = (float *)malloc(sizeof(float) * 2048);
    for (int i = 0; i < b; i++) {
        float running = 0.0f, comp = 0.0f;
        for (int j = 0; j < n; j += 8192) {
            const int len = n - j < 8192 ? n - j : 8192;
            const int ng = (len + 3) / 4;
            const float *x = inp + (long)i * n + j;
            for (int g = 0; g < ng; g++) {
                const int e = g * 4;
                if (e + 3 < len) {
                    const float p2 = x[e] + x[e + 1];
                    const float p3 = x[e + 2] + p2;
                    const float p4 = (x[e + 3] + x[e + 2]) + p2;
                    P[e] = x[e];
                    P[e + 1] = p2;
                    P[e + 2] = p3;
                    P[e + 3] = p4;
                    G[g] = p4;
                } else {
                    float v = 0.0f;
                    for (int t = e; t < len; t++) {
                        v += x[t];
                        P[t] = v;
                    }
                    G[g] = v;
                }
            }
            for (int t = 0; t < len; t++) {
                const int g = t / 4;
                const float pre = g == 0 ? P[t] : P[t] + prefix_groups(G, g);
                out[(long)i * n + j + t] = pre + running;
            }
            const float total = prefix_groups(G, ng) + comp;
            const float r2 = running + total;
            comp = total - (r2 - running);
            running = r2;
        }
    }
    free(P);
    free(G);
}

/* binarysearchKernel, tf_sampling_g.cu:88-104: q = r * total; the first index whose running sum is >= q, found by the
 * kernel's power-of-two descent from n-1 (equal to that only while the running sums are non-decreasing). */
void oracle_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out)
{
    oracle_cumsum(b, n, inp_p, temp);
    int base = 1;
    while (base < n) base <<= 1;
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++) {
            const float q = inp_r[(long)i * m + j] * temp[(long)i * n + n - 1];
            int r = n - 1;
            for (int k = base; k >= 1; k >>= 1)
                if (r >= k && temp[(long)i * n + r - k] >= q) r -= k;
            out[(long)i * m + j] = r;
        }
}
