/*
 * ref_interpolate_driver.cpp -- builds the REFERENCE's stand-alone
 * tf_ops/3d_interpolation/interpolate.cpp into oracle/_ref/libref_interpolate.so.
 * TEST INFRASTRUCTURE ONLY.  Compiled from where it lies (-DREF_SRC=...), nothing copied.
 *
 * interpolate_cpu / interpolate_grad_cpu there are the same loops as
 * tf_interpolate.cpp:107-153.  Its threenn_cpu (interpolate.cpp:21-61) ignores xyz1
 * (d = x2*x2+y2*y2+z2*z2) -- i.e. it is tf_interpolate.cpp:60-103 evaluated for a query at
 * the origin.  Because fp32 (x2 - x1) is one rounded subtraction in both files, calling it
 * on the pre-translated cloud fl(xyz2 - q) reproduces tf_interpolate.cpp:73 bit for bit:
 * that is what ref_three_nn() below does, one query at a time, so the reference's own
 * compare/insert cascade and its double-widening decide the result.
 */
#define main votenet_ref_interpolate_main
#include REF_SRC
#undef main
#include <vector>

extern "C" {
void ref_threenn_origin(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx)
{
    threenn_cpu(b, n, m, xyz1, xyz2, dist, idx);
}
void ref_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx)
{
    std::vector<float> shifted((size_t)m * 3);
    for (int i = 0; i < b; i++)
        for (int j = 0; j < n; j++) {
            const float *q = xyz1 + ((size_t)i * n + j) * 3;
            const float *src = xyz2 + (size_t)i * m * 3;
            for (int k = 0; k < m; k++) {
                shifted[k * 3 + 0] = src[k * 3 + 0] - q[0];
                shifted[k * 3 + 1] = src[k * 3 + 1] - q[1];
                shifted[k * 3 + 2] = src[k * 3 + 2] - q[2];
            }
            threenn_cpu(1, 1, m, q, shifted.data(), dist + ((size_t)i * n + j) * 3, idx + ((size_t)i * n + j) * 3);
        }
}
void ref_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                           float *out)
{
    interpolate_cpu(b, m, c, n, points, idx, weight, out);
}
void ref_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points)
{
    interpolate_grad_cpu(b, n, c, m, grad_out, idx, weight, grad_points);
}
}
