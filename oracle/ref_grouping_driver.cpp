/*
 * ref_grouping_driver.cpp -- builds the REFERENCE's own CPU twins of ball query / group /
 * group-grad into oracle/_ref/libref_grouping.so.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference file tf_ops/grouping/test/query_ball_point.cpp is a stand-alone program
 * (libc only).  It is compiled from where it lies under the reference tree (path given by
 * -DREF_SRC=...); nothing of it is copied into this repository.  Its main() is renamed so
 * the functions can be called, and thin extern "C" shims expose them to ctypes.
 * Only available where the reference tree is mounted (this container, not the GPU box).
 */
#define main votenet_ref_grouping_main
#include REF_SRC
#undef main

extern "C" {
void ref_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                          const float *xyz2, int *idx)
{
    query_ball_point_cpu(b, n, m, radius, nsample, xyz1, xyz2, idx);
}
void ref_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out)
{
    group_point_cpu(b, n, c, m, nsample, points, idx, out);
}
void ref_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                          float *grad_points)
{
    group_point_grad_cpu(b, n, c, m, nsample, grad_out, idx, grad_points);
}
}
