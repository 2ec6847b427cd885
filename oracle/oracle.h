/*
 * oracle.h -- CPU restatement of the VoteNet point-cloud hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the timed CPU baseline.  The product path is
 * votenet_amd/csrc (HIP) behind include/votenet_hip.h and fails loudly without it.
 *
 * Every function cites the reference file:line (relative to the reference tree of
 * qq456cvb/VoteNet) whose arithmetic it follows.  Build flags are pinned in
 * oracle/Makefile: -O2 -ffp-contract=off, no -march (fp32, no FMA contraction,
 * left-to-right evaluation exactly as the reference expressions are written).
 *
 * Pinning status (see oracle/README.md):
 *   ball query / group / group-grad ... pinned against the reference's own compiled
 *                                       CPU twins (oracle/_ref, built from
 *                                       tf_ops/grouping/test/query_ball_point.cpp) and against
 *                                       its device kernels (tf_grouping_g.cu compiled for gfx950:
 *                                       oracle/_ref/libref_grouping_gpu.so, ref_gpu_grouping.npz)
 *   three_nn / interpolate / grad ..... pinned against oracle/_ref built from
 *                                       tf_ops/3d_interpolation/interpolate.cpp
 *   FPS / gather / scatter-add ........ restatement of a device-only kernel, pinned against
 *                                       that kernel itself: tf_sampling_g.cu compiled for
 *                                       gfx950 where it lies (oracle/_ref/libref_sampling_gpu.so;
 *                                       fixtures tests/golden/ref_gpu_fps.npz, ref_gpu_gather.npz)
 *   3D IoU / NMS ...................... tf_nms3d.cpp needs TensorFlow headers, which
 *                                       this image lacks: unbuildable here.  Pinned only
 *                                       by the known answer of the reference's own smoke
 *                                       input and an independent
 *                                       polygon-clipping cross-check: PARITY PARTIAL
 *   SelectionSort ..................... pinned against oracle/_ref built from
 *                                       tf_ops/grouping/test/selection_sort.cpp
 *   ProbSample ........................ restatement of a device-only kernel, pinned against
 *                                       that kernel itself (same library; fixture
 *                                       tests/golden/ref_gpu_prob_sample.npz: cumsum and picks)
 *   grouped MLP ....................... arithmetic lives in Tensorpack/TensorFlow 1.x
 *                                       (not in the reference tree, versions unpinned):
 *                                       PARITY UNPINNED, semantics defined here
 */
#ifndef VOTENET_ORACLE_H
#define VOTENET_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* tf_ops/sampling/tf_sampling_g.cu:105-170 (kernel), :203-205 (launcher <<<32,512>>>) */
void oracle_farthest_point_sample(int b, int n, int m, const float *dataset, int *idxs);
/* closed form of the same tie rule (max d2, then min k%512, then min k); used to
 * cross-check the literal 512-lane simulation above */
void oracle_farthest_point_sample_closed(int b, int n, int m, const float *dataset, int *idxs);
/* tf_sampling_g.cu:172-181 */
void oracle_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out);
/* tf_sampling_g.cu:183-192 ; inp_g must be pre-zeroed (tf_sampling.cpp:174) */
void oracle_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g);

/* tf_ops/grouping/tf_grouping_g.cu:3-36 (adds pts_cnt to the CPU twin
 * tf_ops/grouping/test/query_ball_point.cpp:19-47) */
void oracle_query_ball_point(int b, int n, int m, float radius, int nsample,
                             const float *xyz1, const float *xyz2, int *idx, int *pts_cnt);
/* tf_grouping_g.cu:40-57 / test/query_ball_point.cpp:52-66 */
void oracle_group_point(int b, int n, int c, int m, int nsample,
                        const float *points, const int *idx, float *out);
/* tf_grouping_g.cu:61-78 / test/query_ball_point.cpp:70-84 ; grad_points pre-zeroed */
void oracle_group_point_grad(int b, int n, int c, int m, int nsample,
                             const float *grad_out, const int *idx, float *grad_points);

/* tf_ops/3d_interpolation/tf_interpolate.cpp:60-103 */
void oracle_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                     float *dist, int *idx);
/* utils.py:279-282 : d=max(d,1e-10); w=(1/d)/sum(1/d)  (sum taken left to right) */
void oracle_three_nn_weights(int b, int n, const float *dist, float *weight);
/* tf_interpolate.cpp:107-127 */
void oracle_three_interpolate(int b, int m, int c, int n, const float *points,
                              const int *idx, const float *weight, float *out);
/* tf_interpolate.cpp:131-153 ; grad_points pre-zeroed */
void oracle_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out,
                                   const int *idx, const float *weight, float *grad_points);

/* tf_ops/3d_nms/tf_nms3d.cpp:43-50,53-175 : BEV quad/quad intersection area */
float oracle_bev_intersection(const float *bbox1, const float *bbox2);
/* tf_nms3d.cpp:178-192 : 3D IoU of two (8,3) corner boxes */
float oracle_iou3d(const float *bbox1, const float *bbox2);
/* full (nboxes x nboxes) IoU matrix of one scene, row-major */
void oracle_iou3d_matrix(int nboxes, const float *bboxes, float *iou);
/* tf_nms3d.cpp:202-273 : returns number of selected rows written to out (cap b*n rows of 2) */
int oracle_nms3d(int b, int n, const float *bboxes, const float *scores,
                 const float *objectiveness, float iou_threshold, int *out);

/* Grouped-point MLP (utils.py:50-57,125-132).  Semantics defined by this oracle:
 *   x[r, :]   = [xyz[idx]-center (3), feat[idx] (c)]            (sample_and_group)
 *   z         = x W + bias                                      (Conv2D 1x1, NHWC)
 *   y         = relu(gamma*(z-mean)/sqrt(var+eps)+beta)         (BNReLU, batch stats,
 *                                                                biased variance)
 * all fp32, accumulation in k order with fmaf (matches v_mfma_f32 numerics, see
 * oracle_mlp.c).  */
void oracle_group_concat(int b, int n, int c, int m, int nsample, const float *xyz,
                         const float *new_xyz, const float *points, const int *idx,
                         float *out /* (b,m,nsample,3+c) */);
void oracle_linear(long rows, int cin, int cout, const float *x, const float *w /*cin x cout*/,
                   const float *bias /* cout or NULL */, float *z);
void oracle_bn_stats(long rows, int c, const float *z, float *mean, float *var);
void oracle_bn_relu(long rows, int c, const float *z, const float *mean, const float *var,
                    const float *gamma, const float *beta, float eps, int relu, float *y);
void oracle_max_over_k(long groups, int k, int c, const float *y, float *out);

/* ---- ops the reference ships but model.py never reaches (oracle_variants.c) ---- */
/* tf_ops/grouping/tf_grouping_g.cu:83-123 / test/selection_sort.cpp:19-62 (pinned against oracle/_ref) */
void oracle_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out);
/* tf_ops/grouping/tf_grouping.py:61-63 : squared distances of knn_point, (b,m,n) */
void oracle_knn_dist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist);
/* tf_ops/sampling/tf_sampling_g.cu:7-86 : float running sum with the kernel's scan-tree association */
void oracle_cumsum(int b, int n, const float *inp, float *out);
/* tf_sampling_g.cu:88-104,197-200 ; temp = b*n floats (the running sums) */
void oracle_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out);

/* liboracle_omp.so (the same sources with -fopenmp, oracle/Makefile): OpenMP thread count of the parallel-for loops over
 * independent queries / rows / FPS lanes; a no-op in liboracle.so.  bench.py's all-core cpu_baseline only. */
void oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
