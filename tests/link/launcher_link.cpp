// Link-level proof of the drop-in seam.
// The reference's op wrappers only DECLARE their launchers and leave the definitions to the .cu object:
//   tf_ops/sampling/tf_sampling.cpp:65 (probsample), :94 (farthestpointsampling), :125 (gatherpoint), :150 (scatteraddpoint)
//   tf_ops/grouping/tf_grouping.cpp:66 (queryBallPoint), :108 (selectionSort), :142 (groupPoint), :173 (groupPointGrad)
// This TU declares the same eight names with the same C++ signatures and is linked with
//   g++ -shared -Wl,-z,defs ... -lvotenet_hip
// so one missing or mis-typed export is an undefined symbol and the link fails -- the failure
// tf.load_op_library (dlopen RTLD_NOW) would hit with the reference's real wrappers.
// The link_* functions let the tests drive each launcher through this object.
void probsampleLauncher(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out);
void farthestpointsamplingLauncher(int b, int n, int m, const float *inp, float *temp, int *out);
void gatherpointLauncher(int b, int n, int m, const float *inp, const int *idx, float *out);
void scatteraddpointLauncher(int b, int n, int m, const float *out_g, const int *idx, float *inp_g);
void queryBallPointLauncher(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2, int *idx,
                            int *pts_cnt);
void selectionSortLauncher(int b, int n, int m, int k, const float *dist, int *outi, float *out);
void groupPointLauncher(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out);
void groupPointGradLauncher(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx, float *grad_points);

extern "C" {
void link_prob_sample(int b, int n, int m, const float *p, const float *r, float *temp, int *out)
{
    probsampleLauncher(b, n, m, p, r, temp, out);
}
void link_fps(int b, int n, int m, const float *inp, float *temp, int *out) { farthestpointsamplingLauncher(b, n, m, inp, temp, out); }
void link_gather(int b, int n, int m, const float *inp, const int *idx, float *out) { gatherpointLauncher(b, n, m, inp, idx, out); }
void link_scatter_add(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    scatteraddpointLauncher(b, n, m, out_g, idx, inp_g);
}
void link_query_ball(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2, int *idx, int *cnt)
{
    queryBallPointLauncher(b, n, m, radius, nsample, xyz1, xyz2, idx, cnt);
}
void link_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out)
{
    selectionSortLauncher(b, n, m, k, dist, outi, out);
}
void link_group(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out)
{
    groupPointLauncher(b, n, c, m, nsample, points, idx, out);
}
void link_group_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx, float *grad_points)
{
    groupPointGradLauncher(b, n, c, m, nsample, grad_out, idx, grad_points);
}
}
