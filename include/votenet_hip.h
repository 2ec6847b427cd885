/*
 * votenet_hip.h -- C ABI of libvotenet_hip.so: the MI355X (gfx950) implementation of the
 * VoteNet / PointNet++ point-cloud hot path.
 *
 * Boundary.  The reference (qq456cvb/VoteNet) puts this path behind a "launcher seam": free
 * functions declared in tf_ops/<op>/tf_<op>.cpp and defined in tf_<op>_g.cu (GPU ops) or in
 * the same .cpp (CPU ops).  Each entry point below replaces one of them; the citation says
 * which.  Differences from the reference seam, all deliberate:
 *   - extern "C", plain pointers and sizes, no TensorFlow / torch types;
 *   - every pointer is DEVICE memory (the reference's interpolate / NMS ops take host
 *     memory and force a device<->host round trip every step);
 *   - an explicit stream (a hipStream_t passed as void*; NULL = the null stream) -- the
 *     reference launches on the legacy default stream (tf_sampling_g.cu:204);
 *   - an int status (0 = ok, VOTENET_E_* otherwise; text via votenet_last_error()) -- the
 *     reference launchers return void and never check an error.  Argument validation
 *     mirrors the OP_REQUIRES checks of the TF wrappers (cited per function).
 * Ownership is the reference's: the caller owns every buffer, launchers never allocate,
 * gradient buffers must be zeroed by the caller (tf_sampling.cpp:174, tf_grouping.cpp:204,
 * tf_interpolate.cpp:258).  All tensors are dense, row-major, fp32 / int32.
 * Calls are asynchronous with respect to the host and re-entrant.  What a call does depends on its arguments only, with two
 * documented pieces of per-thread / registered state: the split-K arming (votenet_mlp_split_k_*: per calling thread, consumed by the
 * next launch) and the bf16 x 3 weight images a host registers (votenet_register_split_weights: keyed by the weight pointer).  The
 * process-global measurement / tuning switches live in votenet_hip_debug.h and are INERT unless the host opts in with
 * votenet_debug_enable(1) (or VOTENET_DEBUG=1 in the environment): a drop-in consumer never sees them.  The shared object exports
 * exactly the functions declared in these two headers plus the reference's eight launcher names (csrc/exports.map, checked by
 * tests/test_abi.py); nothing else is visible.  No launcher synchronises with the device, allocates
 * device memory or copies from pageable host memory, so a sequence of calls on one stream can be captured into a HIP graph
 * (hipStreamBeginCapture ... hipStreamEndCapture) and replayed over the same buffers: the host side does that with the whole
 * coordinate-only chain of a batch (votenet_farthest_point_sample, votenet_gather_point, votenet_query_ball_point*, votenet_three_nn,
 * votenet_half_groups, votenet_assemble_rows_half / votenet_narrow_rows_half, votenet_half_sort_rows: votenet_amd/model.py,
 * GeometryGraph).  votenet_half_groups writes its count into a mapped pinned host int (nh_host): that works from a graph too.
 *
 * libvotenet_hip.so additionally exports the reference's EIGHT launcher names with their exact
 * C++ signatures (probsampleLauncher, farthestpointsamplingLauncher, gatherpointLauncher,
 * scatteraddpointLauncher -- tf_sampling.cpp:65,94,125,150; queryBallPointLauncher,
 * selectionSortLauncher, groupPointLauncher, groupPointGradLauncher -- tf_grouping.cpp:66,108,142,173),
 * so tf_sampling.cpp / tf_grouping.cpp link against it unchanged in place of tf_*_g.cu.o -- see INTEGRATION.md.
 */
#ifndef VOTENET_HIP_H
#define VOTENET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOTENET_OK 0
#define VOTENET_E_INVALID_ARGUMENT 1 /* the reference's errors::InvalidArgument */
#define VOTENET_E_HIP 2              /* a HIP runtime / launch error (hipGetLastError) */
#define VOTENET_E_WORKSPACE 3        /* caller-provided workspace too small */

/* Text of the last error raised on the calling thread ("" if none). */
const char *votenet_last_error(void);
/* Library / build identification, e.g. "votenet_hip 0.1 gfx950". */
const char *votenet_version(void);

/* ---------------------------------------------------------------- tf_ops/sampling */

/* Replaces farthestpointsamplingLauncher (decl tf_sampling.cpp:94, def tf_sampling_g.cu:203-205,
 * kernel :105-170).  inp (b,n,3) -> out (b,m) int32.  temp: scratch of at least
 * votenet_fps_temp_floats(b,n) floats (the reference allocates 32*n, tf_sampling.cpp:115);
 * may be NULL when that function returns 0 (n <= 4096: the cloud lives in registers).  Above that the
 * scratch holds the cloud's SPATIAL INDEX (see votenet_spatial_index: about 5.3*n floats per scene), which the call
 * builds and leaves behind for votenet_query_ball_point_indexed on the same cloud (was: 1.1*n floats per scene, n <= 24 576;
 * about 5.1*n floats per scene,
 * n <= 262 144); beyond, the reference's running-distance rows.  Requires m > 0 (tf_sampling.cpp:99).
 * Bit-exact with the reference rule: start at 0, running distance 1e38, arg-max of
 * min(d, running) with ties -> smallest (k mod 512), then smallest k. */
int votenet_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out, void *stream);
size_t votenet_fps_temp_floats(int b, int n);

/* Replaces gatherpointLauncher (tf_sampling.cpp:125, tf_sampling_g.cu:172-181,206-208).
 * inp (b,n,3), idx (b,m) -> out (b,m,3). */
int votenet_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream);

/* Replaces scatteraddpointLauncher (tf_sampling.cpp:150, tf_sampling_g.cu:183-192,209-211).
 * out_g (b,m,3), idx (b,m) -> inp_g (b,n,3) += ; inp_g pre-zeroed by the caller. */
int votenet_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g, void *stream);

/* ---------------------------------------------------------------- tf_ops/grouping */

/* Spatial index of a batch of clouds (no reference counterpart: the reference scans every candidate for every query,
 * tf_grouping_g.cu:13-35 / tf_sampling_g.cu:130-147).  Points sorted by Morton cell (16^3 grid over the cloud's bounding
 * box) into buckets of 64 consecutive sorted points with their bounding boxes.  index: scratch of
 * votenet_spatial_index_floats(b, n) floats, laid out as [perm b*n int | boxes b*nb*6 | sorted float4 b*nb*64 (16-byte
 * aligned) | work], nb = ceil(n/64).  votenet_farthest_point_sample builds the same structure in its temp scratch for
 * 4096 < n <= 262144, so a ball query on the cloud that was just sampled needs no second build. */
size_t votenet_spatial_index_floats(int b, int n);
int votenet_spatial_index(int b, int n, const float *xyz, float *index, void *stream);

/* votenet_query_ball_point with the candidates' spatial index: identical idx / pts_cnt, but only the buckets whose box
 * reaches into the query ball are tested (first-nsample-in-index-order through an n-bit LDS bitmap per query).
 * index == NULL or n > 131072: falls back to votenet_query_ball_point. */
int votenet_query_ball_point_indexed(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                     const float *xyz2, const float *index, int *idx, int *pts_cnt, void *stream);

/* Replaces queryBallPointLauncher (tf_grouping.cpp:66, tf_grouping_g.cu:3-36,125-128).
 * xyz1 (b,n,3) candidates, xyz2 (b,m,3) queries -> idx (b,m,nsample), pts_cnt (b,m).
 * First nsample candidates in ascending index with max(sqrtf(d2),1e-20f) < radius, remaining
 * slots = first hit; a query with no hit gets idx row 0 and pts_cnt 0 (the reference leaves
 * the row uninitialised).  Requires radius > 0, nsample > 0 (tf_grouping.cpp:71,74). */
int votenet_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                             const float *xyz2, int *idx, int *pts_cnt, void *stream);

/* Host helper (no GPU work): the squared-distance threshold the ball-query kernel compares
 * against, T(r) = smallest fp32 with sqrtf(T) >= r, so that  s < T(r)  <=>  sqrtf(s) < r
 * (tf_grouping_g.cu:24-25).  Not equal to r*r in general. */
float votenet_ball_threshold(float radius);

/* Replaces groupPointLauncher (tf_grouping.cpp:142, tf_grouping_g.cu:40-57,133-136).
 * points (b,n,c), idx (b,m,nsample) -> out (b,m,nsample,c). */
int votenet_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                        float *out, void *stream);

/* Replaces groupPointGradLauncher (tf_grouping.cpp:173, tf_grouping_g.cu:61-78,137-141).
 * grad_out (b,m,nsample,c), idx -> grad_points (b,n,c) += ; pre-zeroed by the caller. */
int votenet_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                             float *grad_points, void *stream);

/* ---------------------------------------------------------------- tf_ops/3d_interpolation */

/* Replaces threenn_cpu (tf_interpolate.cpp:60-103).  xyz1 (b,n,3) unknown, xyz2 (b,m,3) known
 * -> dist (b,n,3) SQUARED distances, idx (b,n,3); ties keep the lower index first; fewer
 * than three known points -> (inf, 0). */
int votenet_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                     void *stream);

/* Replaces the four TF elementwise ops of utils.py:279-282: d=max(d,1e-10),
 * w_i=(1/d_i)/((1/d_1+1/d_2)+1/d_3).  dist (b,n,3) -> weight (b,n,3). */
int votenet_three_nn_weights(int b, int n, const float *dist, float *weight, void *stream);

/* Replaces threeinterpolate_cpu (tf_interpolate.cpp:107-127).  points (b,m,c), idx/weight
 * (b,n,3) -> out (b,n,c) = (p1*w1 + p2*w2) + p3*w3. */
int votenet_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                              const float *weight, float *out, void *stream);

/* Replaces threeinterpolate_grad_cpu (tf_interpolate.cpp:131-153).  grad_out (b,n,c) ->
 * grad_points (b,m,c) += ; pre-zeroed by the caller. */
int votenet_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points, void *stream);
/* The FP layer's concat written by the interpolation (utils.py:283-286): out (b,n,c+c1) = [three_interpolate(points) | skip],
 * skip (b,n,c1) = the unknown points' own features. */
int votenet_three_interpolate_concat(int b, int m, int c, int n, const float *points, const int *idx, const float *weight,
                                     const float *skip, int c1, float *out, void *stream);
/* votenet_three_interpolate_grad with grad_out read in place from a wider tensor: rows of go_pitch floats, the c channels at go_off
 * (the gradient of the concat above is [d interpolated | d skip]). */
int votenet_three_interpolate_grad_strided(int b, int n, int c, int m, const float *grad_out, int go_pitch, int go_off,
                                           const int *idx, const float *weight, float *grad_points, void *stream);

/* ---------------------------------------------------------------- tf_ops/3d_nms */

/* 3D IoU of every ordered pair of boxes of a scene (tf_nms3d.cpp:43-192, evaluated lazily
 * pair by pair there).  bboxes (b,n,8,3) corner boxes in the order of model.py:108-110 ->
 * iou (b,n,n), iou[s,i,j] = IoU(box i, box j) with box i as the reference's first argument. */
int votenet_iou3d_matrix(int b, int n, const float *bboxes, float *iou, void *stream);

/* Replaces NonMaxSuppression3DOp::Compute / DoNonMaxSuppressionOp (tf_nms3d.cpp:202-308).
 * bboxes (b,n,8,3), scores (b,n), objectiveness (b,n,2), 0 <= iou_threshold <= 1
 * (tf_nms3d.cpp:300) -> out (capacity b*n rows of [batch, box], int32) in descending-score
 * visit order over the whole batch, *out_count (device int) = rows written.  The output
 * length is data dependent, hence caller-provided capacity + device-side count.
 * workspace: votenet_nms3d_workspace_bytes(b,n) bytes of device scratch.
 * Equal scores are visited in ascending flat index (the reference's heap order for ties is
 * unspecified). */
int votenet_nms3d(int b, int n, const float *bboxes, const float *scores, const float *objectiveness,
                  float iou_threshold, int *out, int *out_count, void *workspace, size_t workspace_bytes,
                  void *stream);
size_t votenet_nms3d_workspace_bytes(int b, int n);

/* ---------------------------------------------------------------- grouped-point MLP
 * (utils.py:50-57,125-132,149-155,286-293; the reference runs it as Tensorpack Conv2D 1x1 +
 * BNReLU graph nodes on a materialised (B,m,K,3+C) tensor).  fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32.  Row r of the implicit input matrix is either
 *   GATHER : [xyz[idx[r]] - new_xyz[r / nsample] (3), feat[idx[r]] (c)]      (sample_and_group)
 *   DENSE  : x[r, :] optionally passed through y = max(0, x*scale + shift)    (BNReLU of the
 *            previous layer folded into the load)
 * and the output is z = row * W + bias, plus per-channel sum / sum of squares of z
 * (the BatchNorm batch statistics of this layer) accumulated into stats[2*cout].
 */
/* The BatchNorm of a tensor, given by the RAW column sums of the launch that produced it (votenet_mlp_linear stats):
 * the consumer derives mean / var / scale / shift itself (votenet_bn_finalize's arithmetic) in its prologue, so no
 * separate finalize launch exists, and one of its workgroups writes them to out for the backward pass. */
typedef struct votenet_bn_raw {
    const double *stats;       /* 2*c: column sums of z, of z*z */
    const float *gamma, *beta; /* c each */
    long rows;                 /* rows the sums run over */
    float eps;
    float *out;                /* 4*c floats: scale | shift | mean | var (may be NULL) */
} votenet_bn_raw;

/* Optional tail of a kernel that reduces the BatchNorm-backward sums of a layer (votenet_bn_backward_reduce(_pool),
 * votenet_pool_dgrad_scatter, votenet_mlp_dgrad_bn_reduce, votenet_narrow_dgrad_bn_reduce): the LAST workgroup to finish
 * (an atomic ticket) turns the completed sums into the coefficient vector, i.e. does votenet_bn_backward_coef's work --
 * coef = [A|B|C|scale|shift] (5*c floats), dgamma += sums[c:], dbeta += sums[:c] -- so that no separate launch sits between
 * the reduction and its consumers on the step's dependent chain.  ticket: one zero-initialised unsigned the kernel resets
 * to zero when it is done; kernels that may run concurrently need distinct tickets.  scale / shift / mean / var / eps are
 * the ones the reducing call already takes.  NULL: no tail (the caller launches votenet_bn_backward_coef). */
typedef struct votenet_coef_tail {
    unsigned *ticket;
    long rows;          /* rows the sums run over (N of the BatchNorm) */
    const float *gamma; /* c */
    float *coef;        /* 5*c, written */
    float *dgamma;      /* c, accumulated; may be NULL */
    float *dbeta;       /* c, accumulated; may be NULL */
} votenet_coef_tail;

typedef struct votenet_mlp_input {
    /* DENSE source (rows x cin); NULL for GATHER */
    const float *x;
    const float *in_scale; /* cin, or NULL: no affine+relu on load */
    const float *in_shift; /* cin */
    int in_relu;           /* apply max(0,.) after the affine */
    const votenet_bn_raw *in_bn; /* alternative to in_scale / in_shift: the affine comes from raw statistics, or NULL */
    /* GATHER source */
    const float *xyz;     /* (b,n,3) */
    const float *new_xyz; /* (b,m,3) */
    const float *feat;    /* (b,n,c) or NULL */
    const int *idx;       /* (b,m,nsample) */
    int b, n, m, nsample, c;
} votenet_mlp_input;

/* z (rows x cout) = input(rows x cin) * w (cin x cout, row-major) + bias (cout, may be NULL).
 * stats: 2*cout DOUBLES, [0,cout) += column sums of z, [cout,2cout) += column sums of z*z
 * (fp32 partial sums per workgroup, fp64 accumulation across workgroups); pre-zeroed by the
 * caller; may be NULL.  rows = b*m*nsample for GATHER. */
int votenet_mlp_linear(const votenet_mlp_input *in, long rows, int cin, int cout, const float *w,
                       const float *bias, float *z, double *stats, void *stream);

/* BatchNorm scale/shift from accumulated statistics: mean = sum/rows, var = sumsq/rows - mean^2
 * (biased), scale = gamma*rsqrt(var+eps), shift = beta - mean*scale.  Also writes mean and
 * var (each may be NULL). */
int votenet_bn_finalize(long rows, int c, const double *stats, const float *gamma, const float *beta, float eps,
                        float *scale, float *shift, float *mean, float *var, void *stream);

/* First SA layer with the linear map applied before the grouping (a gather commutes with a per-point
 * linear map): with P (b*n x cout) = feat . W[3:] from votenet_mlp_linear over the b*n points,
 *   z[b,j,k,:] = P[b, idx[b,j,k], :] + (xyz[b,idx[b,j,k]] - new_xyz[b,j]) . w_xyz + bias
 * is the layer output over the b*m*nsample grouped rows (the conv2d of utils.py:125-127 over the
 * sample_and_group concat, utils.py:50-57) -- nsample-times fewer multiply-adds than the grouped GEMM.
 * w_xyz (3 x cout) = W[0:3].  stats as for votenet_mlp_linear (may be NULL).  cout: a power of two
 * in [4, 1024].  The summation order differs from the fused GATHER GEMM (xyz terms last): results
 * agree to fp32 rounding, not bit for bit. */
int votenet_group_linear(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                         const int *idx, const float *P, const float *w_xyz, const float *bias, float *z,
                         double *stats /* 2*cout, pre-zeroed, may be NULL */, void *stream);

/* Backward of votenet_group_linear for a BatchNorm'ed layer, one pass over (z, da) (rows = b*m*nsample):
 *   dz = A*g' + B + C*z with g' = da masked by [z*S+H > 0] when relu   (coef = [A|B|C|S|H] from votenet_bn_backward_coef)
 *   s_points[b, idx[b,j,k], :] += dz[b,j,k,:]      (b x n x cout, pre-zeroed: GroupPointGrad, tf_grouping_g.cu:61-78,
 *                                                   at the layer OUTPUT width; then dW[3:] += feat^T S, d_feat = S W[3:]^T)
 *   dw_xyz[d, :] += sum_rows (xyz[b,idx]-new_xyz[b,j])[d] * dz       (3 x cout: the xyz rows of the weight gradient)
 *   dz_out = dz (rows x cout) if not NULL.
 * pts_cnt as for votenet_group_concat_grad (may be NULL).  cout in {32,64,128,256}, nsample <= 128. */
int votenet_group_linear_backward(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                  const int *idx, const int *pts_cnt, const float *z, const float *da, const float *coef,
                                  int relu, float *s_points, float *dw_xyz, float *dz_out, void *stream);

/* out (groups x c) = max over the k rows of each group of max(0?, z*scale+shift);
 * argmax (groups x c, int32 row offset inside the group, may be NULL) for the backward pass. */
int votenet_bn_relu_max(long groups, int k, int c, const float *z, const float *scale, const float *shift,
                        int relu, float *out, int *argmax, void *stream);

/* votenet_mlp_linear with the max-pool of utils.py:132 started in its epilogue: besides z and stats it writes, per group
 * of pool_k consecutive rows and channel, the RAW maximum and minimum of z and the row offsets where they are attained
 * (zmax / zmin / amax / amin, each rows/pool_k x cout).  The layer's BatchNorm scale needs the statistics of the whole
 * launch, but max_k act(s*z+h) = act(s * max_k z + h) for s >= 0 and act(s * min_k z + h) for s < 0, so
 * votenet_bn_pool_finalize completes the pool without another pass over z (same values as votenet_bn_relu_max; among
 * rows that tie after BatchNorm the arg-max may name a different one; with a zero scale -- every row ties -- it names the
 * row of the raw maximum).  zsel (groups x c, may be NULL) receives the raw z at the arg-max row, which is all the
 * backward pass of the layer needs from z (votenet_bn_backward_reduce_pool ...).  z may be NULL when nothing else reads it
 * (inference; training through the Gram-form backward below).  Served: DENSE input, pool_k == 64, rows % 128 == 0, cin % 32 == 0, cin <= 512, cout % 128 == 0,
 * 16-byte aligned buffers; anything else returns VOTENET_E_INVALID_ARGUMENT. */
int votenet_mlp_linear_pool(const votenet_mlp_input *in, long rows, int cin, int cout, const float *w, const float *bias,
                            float *z /* may be NULL */, double *stats, int pool_k, float *zmax, float *zmin, int *amax,
                            int *amin, void *stream);
int votenet_bn_pool_finalize(long groups, int c, const float *zmax, const float *zmin, const int *amax, const int *amin,
                             const float *scale, const float *shift, const votenet_bn_raw *bn /* instead of scale / shift, or NULL */,
                             int relu, float *out, int *argmax /* may be NULL */, float *zsel /* may be NULL */, void *stream);

/* ---- backward of the pooled (last) layer of an SA chain in "Gram form" (pool_bwd.hip) ----
 * Layer: z = x W + b (x = the layer's input activation, rows x cin, given as raw xz + folded BatchNorm/ReLU), BatchNorm,
 * ReLU, max over the k rows of each group.  The folded BatchNorm backward is dz = A g' + B + C z with g' non-zero only at
 * the arg-max rows (coef = [A|B|C|scale|shift], votenet_bn_backward_coef).  Substituting z = x W + b:
 *     da = dz W^T = x (W diag(C) W^T) + (B + C.b) W^T  +  scatter: da[g*k + argmax[g,c], :] += A[c] g'[g,c] W[:,c]^T
 *     dW = x^T dz = ((x^T x) W) . C + (sum_r x)^T (B + C.b)  +  gather: dW[:,c] += x[g*k + argmax[g,c], :]^T A[c] g'[g,c]
 * so both big GEMMs are rows x cin x cin instead of rows x cin x cout, z is never read, and x^T x depends on the forward
 * pass only.  Call order for one layer:
 *   reduce_pool -> votenet_bn_backward_coef -> dgrad_prepare -> votenet_mlp_linear(x, mmat, bias = cvec) -> dgrad_scatter
 *   mlp_gram (any time after the forward pass) ; wgrad_sparse -> wgrad_finish (after coef)
 * Served shapes: votenet_pool_backward_supported (k = 64; cin x cout = 128x256, 128x128, 64x128: every pooled layer of
 * VoteNet); other layers keep votenet_mlp_wgrad_bn / votenet_mlp_dgrad_bn. */
int votenet_pool_backward_supported(int cin, int cout, int k);
/* sums (2c doubles, pre-zeroed) += [sum g', sum g' zhat] over the arg-max entries: g' = gout masked by the ReLU */
int votenet_bn_backward_reduce_pool(long groups, int c, const float *gout, const float *zsel, const float *scale,
                                    const float *shift, const float *mean, const float *var, float eps, int relu,
                                    double *sums, const votenet_coef_tail *tail /* may be NULL */, void *stream);
/* mmat (cin x cin) = W diag(C) W^T, cvec (cin) = (B + C.b) W^T; w (cin x cout), bias may be NULL */
int votenet_pool_dgrad_prepare(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat,
                               float *cvec, void *stream);
/* the same launch also writes the bf16 x 3 image of mmat (cin * cin * 6 bytes, 16-byte aligned, cin % 16 == 0; the layout of
 * votenet_split_weights) for the one forward-type GEMM that multiplies by it (register it around that launch) */
int votenet_pool_dgrad_prepare_split(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat,
                                     float *cvec, void *image, void *stream);
/* Round 6: the image as two fp16 pieces scaled by powers of two (the matrix is gradient-sized: below fp16's range unscaled).  image:
 * cin * cin * 4 bytes; ascale / unscale: cin floats each, written by the launch -- the power-of-two factors of the GEMM's input
 * channels and output columns.  Register the three around the ONE GEMM that multiplies by mmat:
 *   votenet_register_split_weights_scaled(mmat, cin, cin, image, ascale, unscale); <the GEMM>; votenet_register_split_weights(mmat, cin, cin, NULL). */
int votenet_pool_dgrad_prepare_h2(int cin, int cout, const float *w, const float *bias, const float *coef, float *mmat, float *cvec,
                                  void *image, float *ascale, float *unscale, void *stream);
int votenet_register_split_weights_scaled(const float *w, int cin, int cout, const void *w3, const float *ascale, const float *unscale);
/* da (groups*k x cin) += the scattered rows; wT = W^T (cout x cin).  With below_z != NULL (the raw output, rows x cin, of
 * the layer BELOW, whose output gradient da is) the same pass also performs that layer's votenet_bn_backward_reduce:
 * below_sums (2*cin doubles, pre-zeroed) += [sum g', sum g' zhat] with g' = the final da masked by the layer's ReLU. */
int votenet_pool_dgrad_scatter(long groups, int k, int cin, int cout, const float *gout, const int *argmax,
                               const float *zsel, const float *coef, int relu, const float *wT, float *da,
                               const float *below_z, const float *below_scale, const float *below_shift,
                               const float *below_mean, const float *below_var, float eps, int below_relu,
                               double *below_sums, const votenet_coef_tail *tail /* may be NULL */, void *stream);
/* gram (c x c, pre-zeroed or accumulating) += a^T a with a = act(z * scale + shift); scale_shift = [scale | shift] (2c) */
/* scratch: votenet_mlp_wgrad_scratch_floats(NULL, rows, c, c) floats or NULL, as for votenet_mlp_wgrad */
int votenet_mlp_gram(long rows, int c, const float *z, const float *scale_shift, int relu, float *gram, float *scratch,
                     void *stream);
/* dw (cin x cout) += the gathered rows; colsum (cin, pre-zeroed) += sum_r x[r,:]; xz / in_scale / in_shift / in_relu
 * describe x as in votenet_mlp_input */
int votenet_pool_wgrad_sparse(long groups, int k, int cin, int cout, const float *xz, const float *in_scale,
                              const float *in_shift, int in_relu, const float *gout, const int *argmax, const float *zsel,
                              const float *coef, int relu, float *dw, float *colsum, float *scratch, void *stream);
/* scratch of votenet_pool_wgrad_sparse (NULL: fp32 atomics, summation order unspecified) */
size_t votenet_pool_wgrad_scratch_floats(long groups, int cin, int cout);
/* dw[j,c] += C[c] (gram[j,:] . W[:,c]) + colsum[j] (B[c] + C[c] b[c]) */
int votenet_pool_wgrad_finish(int cin, int cout, const float *gram, const float *colsum, const float *w, const float *bias,
                              const float *coef, float *dw, void *stream);

/* y = max(0?, z*scale+shift) materialised (rows x c); used where the next consumer is not a
 * votenet_mlp_linear (e.g. the FP-layer output that feeds the voting head). */
int votenet_bn_relu(long rows, int c, const float *z, const float *scale, const float *shift,
                    const votenet_bn_raw *bn /* instead of scale / shift, or NULL */, int relu, float *y, void *stream);


/* ---------------------------------------------------------------- grouped-point MLP, backward
 * (the reference obtains these from TensorFlow autodiff over Conv2D / BatchNorm / ReLU /
 * reduce_max nodes plus GroupPointGrad, tf_grouping.py:42-46).  Training-mode BatchNorm:
 *   da' = da * [z*scale+shift > 0]        (ReLU mask; skipped when relu == 0)
 *   s1 = sum_r da', s2 = sum_r da' * zhat, zhat = (z-mean)*rsqrt(var+eps)
 *   dz = gamma*rsqrt(var+eps) * (da' - s1/N - zhat*s2/N),  dgamma += s2,  dbeta += s1
 * da is dense (rows x c) when k == 0; when k > 0 it is the max-pool scatter of gout (rows/k x c)
 * through argmax (rows/k x c): da[g*k+argmax[g,ch], ch] = gout[g,ch], zero elsewhere (utils.py:132). */
int votenet_bn_backward_reduce(long rows, int c, int k, const float *da, const int *argmax, const float *z,
                               const float *scale, const float *shift, const float *mean, const float *var, float eps,
                               int relu, double *sums /* 2*c, pre-zeroed */, const votenet_coef_tail *tail /* may be NULL */, void *stream);
int votenet_bn_backward_apply(long rows, int c, int k, const float *da, const int *argmax, const float *z,
                              const float *coef /* 5*c, from votenet_bn_backward_coef */, int relu,
                              float *dz /* rows x c */, void *stream);

/* dbias[c] += column sums of dz (rows x c); scratch: c doubles. */
int votenet_bias_grad(long rows, int c, const float *dz, double *scratch, float *dbias, void *stream);
/* the same over the first c columns of rows of `pitch` floats (the zero-padded gradient of a ragged layer); zeroed_scratch:
 * c doubles the CALLER has cleared */
int votenet_bias_grad_strided(long rows, int c, const float *dz, int pitch, double *zeroed_scratch, float *dbias, void *stream);

/* dw (cin x cout, the caller's row order) += input(rows x cin)^T * dz (rows x cout), the input
 * described exactly as for votenet_mlp_linear (same fused GATHER / DENSE+BNReLU loaders).
 * The contraction over rows is split across workgroups.  scratch != NULL (votenet_mlp_wgrad_scratch_floats floats): every
 * workgroup stores its partial tile there and a second launch adds the partials to dw in workgroup order -- one summation
 * order, bit-reproducible gradients.  scratch == NULL: the partial tiles are added with fp32 atomics (summation order
 * unspecified, like the reference's cuDNN / atomics-based gradients). */
size_t votenet_mlp_wgrad_scratch_floats(const votenet_mlp_input *in, long rows, int cin, int cout);
int votenet_mlp_wgrad(const votenet_mlp_input *in, long rows, int cin, int cout, const float *dz, float *dw,
                      float *scratch, void *stream);

/* ---- BatchNorm backward folded into the two backward GEMMs (no dz tensor in memory) -----------------
 * With sums = [sum g', sum g'*zhat] (votenet_bn_backward_reduce) the
 * gradient of a BatchNorm'ed layer is per element  dz = A*g' + B + C*z,  g' = g masked by the ReLU
 * [z*scale+shift > 0] (and, for a max-pooled output, by row%k == argmax[row/k]).
 * votenet_bn_backward_coef writes coef = [A | B | C | scale | shift] (5*c floats) and accumulates
 * dgamma += sums[c:], dbeta += sums[:c] (either may be NULL). */
int votenet_bn_backward_coef(long rows, int c, const float *scale, const float *shift, const float *mean,
                             const float *var, float eps, const float *gamma, const double *sums,
                             float *coef /* 5*c */, float *dgamma, float *dbeta, void *stream);

/* votenet_mlp_wgrad with dz formed inside the operand loader.  Exactly one of da (rows x cout, dense
 * upstream gradient) / gout (rows/pool_k x cout, gradient of the max over pool_k rows, with argmax). */
int votenet_mlp_wgrad_bn(const votenet_mlp_input *in, long rows, int cin, int cout, const float *da,
                         const float *gout, const int *argmax, int pool_k, const float *z, const float *coef,
                         int relu, float *dw, float *scratch, void *stream);

/* da_prev (rows x cout) = dz (rows x c) * wT (c x cout), dz formed as above from (da | gout+argmax, zsrc,
 * coef) inside the operand loader.
 * Shapes served: rows % 128 == 0, c % 32 == 0, c <= 512, cout % 64 == 0, 16-byte aligned
 * buffers; anything else returns VOTENET_E_INVALID_ARGUMENT (use votenet_bn_backward_apply +
 * votenet_mlp_linear instead). */
int votenet_mlp_dgrad_bn(long rows, int c, int cout, const float *da, const float *gout, const int *argmax,
                         int pool_k, const float *zsrc, const float *coef, int relu, const float *wT,
                         float *da_prev, void *stream);

/* votenet_mlp_dgrad_bn (dense upstream gradient da only) whose store epilogue also reduces the BatchNorm backward of the
 * layer BELOW -- the layer whose activation da_prev is the gradient of, z_prev (rows x cout) its pre-BatchNorm output:
 *   sums[0:cout] += sum_rows da_prev', sums[cout:2cout] += sum_rows da_prev' * (z_prev - mean_prev) / sqrt(var_prev + eps),
 *   da_prev' = da_prev * [z_prev*scale_prev + shift_prev > 0] (relu_prev != 0) or da_prev (relu_prev == 0),
 * i.e. votenet_bn_backward_reduce(rows, cout, 0, da_prev, NULL, z_prev, ...) without a pass of its own over da_prev: the
 * tile is still in the accumulators.  sums: 2*cout doubles, zeroed by the caller.  Same shapes as votenet_mlp_dgrad_bn. */
int votenet_mlp_dgrad_bn_reduce(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                const float *wT, float *da_prev, const float *z_prev, const float *scale_prev,
                                const float *shift_prev, const float *mean_prev, const float *var_prev, float eps,
                                int relu_prev, double *sums, const votenet_coef_tail *tail /* may be NULL */, void *stream);

/* ---- first SA layer ASSEMBLED inside its consumers (assemble.hip): z0 = P[idx] + dxyz . W[0:3] is never stored ----
 * With the linear map before the grouping (votenet_group_linear) z0[r,:] = P[prow(r),:] + dxyz(r) . wx is a gather of a per-point
 * row (P = feat . W[3:] + b, points x c0, L2 resident) plus three multiply-adds per channel; instead of writing it (rows x c0) and
 * reading it back, the consumers rebuild its elements from geo[r] = (dx, dy, dz, bits(prow)), 16 bytes per grouped row.
 * votenet_assemble_rows writes geo (rows x 4 floats; prow = scene*n + idx) and ADDS, when given, cntv (b*n x 4 64-bit integers,
 * pre-zeroed: per point the number of rows that gather it and the sum of their dxyz in fixed point 2^-32 -- integer atomics, so the
 * sums do not depend on the order of the additions) and moments (9 doubles, pre-zeroed:
 * sum dx, dy, dz, then xx, xy, xz, yy, yz, zz).  Coordinates only.
 * votenet_assemble_stats ADDS the BatchNorm statistics of z0 (2*c0 doubles, pre-zeroed: sum z0, sum z0^2) from one pass over
 * the points (see assemble.hip); exact sums of exact products -- they agree with sums over the fp32 z0 to fp32 rounding.
 * votenet_assemble_z0 writes z0 after all (tests: the device's own arithmetic). */
int votenet_assemble_rows(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const int *idx,
                          const int *pts_cnt /* (b,m) or NULL: the padding slots k >= pts_cnt repeat slot 0 and are added by it */,
                          float *geo, long long *cntv /* may be NULL */, double *moments /* may be NULL */, void *stream);
int votenet_assemble_stats(long npts, int c0, const float *P, const long long *cntv, const float *wx, const double *moments,
                           double *stats, void *stream);
int votenet_assemble_z0(long rows, int c0, const float *geo, const float *P, const float *wx, float *z0, void *stream);
/* Second layer: z (rows x cout) = relu(bn0(z0)) w + bias with z0 rebuilt in the GEMM's operand loader, bn0 from in_bn (raw
 * statistics) or in_scale / in_shift; stats as votenet_mlp_linear.  Served: rows % 128 == 0, c0 % 32 == 0, c0 <= 512,
 * cout % 64 == 0, 16-byte aligned buffers. */
int votenet_assembled_linear(long rows, int c0, int cout, const float *geo, const float *P, const float *wx, const float *in_scale,
                             const float *in_shift, const votenet_bn_raw *in_bn, int in_relu, const float *w, const float *bias,
                             float *z, double *stats, void *stream);

/* Backward of the second layer over an ASSEMBLED first layer, and of the first layer itself (z0 rebuilt, never read):
 * votenet_assembled_wgrad_bn = votenet_mlp_wgrad_bn with x = relu(z0*in_scale+in_shift); votenet_assembled_dgrad_bn_reduce =
 * votenet_mlp_dgrad_bn_reduce with z_prev = z0 (P: points x cout with points*cout*4 < 2^32); votenet_group_linear_backward_assembled
 * = votenet_group_linear_backward with z = z0. */
int votenet_assembled_wgrad_bn(long rows, int c0, int cout, const float *geo, const float *P, const float *wx, const float *in_scale,
                               const float *in_shift, int in_relu, const float *da, const float *z, const float *coef, int relu,
                               float *dw, float *scratch, void *stream);
int votenet_assembled_dgrad_bn_reduce(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                      const float *wT, float *da_prev, const float *geo, const float *P, const float *wx,
                                      const float *scale_prev, const float *shift_prev, const float *mean_prev, const float *var_prev,
                                      float eps, int relu_prev, double *sums, const votenet_coef_tail *tail /* may be NULL */, void *stream);
int votenet_group_linear_backward_assembled(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                            const int *idx, const int *pts_cnt, const float *P, const float *wx, const float *da,
                                            const float *coef, int relu, float *s_points, float *dw_xyz, float *dz_out, void *stream);

/* ---- NARROW first layer of a set-abstraction MLP (narrow.hip): 3 + c <= 8 grouped input channels, no input gradient ----
 * sa1 of VoteNet groups the bare coordinates (model.py:39: l0_points = xyz, so [xyz[idx]-new_xyz | xyz[idx]] has 6 channels).
 * The first layer's output z0 = u W0 + b0 is then a function of eight floats per grouped row and is never stored: the kernels
 * below rebuild it where they need it (one fixed fma chain: bit-identical in every kernel), from
 *   u8 (rows x 8 floats, 16-byte aligned) = (dx, dy, dz, feat[idx][0..c), 0...),  rows = b*m*nsample, row order of idx.
 * votenet_narrow_rows writes u8 and ADDS the moments (72 doubles, pre-zeroed, may be NULL): m[d] = sum_r u[r,d] at [0,8),
 * M[d,e] = sum_r u[r,d] u[r,e] at [8 + 8 d + e].  Coordinates and input features only: it can run ahead of the step. */
int votenet_narrow_rows(int b, int n, int m, int nsample, int c, const float *xyz, const float *new_xyz,
                        const float *feat /* (b,n,c) or NULL when c == 0 */, const int *idx, float *u8, double *moments,
                        void *stream);
/* z0 (rows x c0) = u8 W0 + b0 with the kernels' own fma chain, for a caller that wants the layer output after all (tests). */
int votenet_narrow_z0(long rows, int k0, int c0, const float *u8, const float *w0, const float *b0, float *z0, void *stream);
/* BatchNorm statistics of z0 from the moments (W0: k0 x c0 row-major, rows [xyz(3) | feat(c)], k0 = 3 + c; b0: c0 or NULL):
 * stats[c] = sum_r z0[r,c], stats[c0 + c] = sum_r z0[r,c]^2 (2*c0 doubles, WRITTEN; the layout votenet_bn_finalize and
 * votenet_bn_raw take).  Exact sums of the exact products, not of the fp32-rounded z0: they agree to fp32 rounding. */
int votenet_narrow_stats(long rows, int k0, int c0, const double *moments, const float *w0, const float *b0, double *stats,
                         void *stream);
/* Second layer: z (rows x cout) = relu(bn0(z0)) w + bias, bn0 from in_bn (raw statistics, finalized in the prologue) or
 * in_scale / in_shift.  stats as votenet_mlp_linear.  Served: rows % 128 == 0, c0 % 32 == 0, c0 <= 128, cout == 64 or
 * cout % 128 == 0, 16-byte aligned buffers. */
int votenet_narrow_linear(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                          const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu,
                          const float *w, const float *bias, float *z, double *stats, void *stream);
/* Backward of the second layer.  votenet_narrow_wgrad_bn: dw (c0 x cout) += act(z0)^T dz1, dz1 from (da, z, coef) as
 * votenet_mlp_wgrad_bn (c0 % 64 == 0, cout % 64 == 0).  votenet_narrow_dgrad_bn_reduce: the input-gradient GEMM
 * da0 = dz1 wT (c x c0) whose result is NOT stored: its epilogue reduces the first layer's BatchNorm backward
 * (sums, 2*c0 doubles, as votenet_mlp_dgrad_bn_reduce) and ug[d*c0 + c] += sum_r u8[r,d] da0'[r,c] (8*c0 doubles); both
 * pre-zeroed.  Served: rows % 128 == 0, c % 32 == 0, c <= 512, c0 % 64 == 0, c0 <= 128 per launch column block. */
int votenet_narrow_wgrad_bn(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                            const float *in_scale, const float *in_shift, int in_relu, const float *da, const float *z,
                            const float *coef, int relu, float *dw, float *scratch, void *stream);
int votenet_narrow_dgrad_bn_reduce(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef,
                                   int relu, const float *wT, const float *u8, const float *w0, const float *b0,
                                   const float *scale0, const float *shift0, const float *mean0, const float *var0, float eps,
                                   int relu0, double *sums, double *ug, const votenet_coef_tail *tail /* may be NULL */, void *stream);
/* Weight gradient of the first layer from the sums alone (dz0 = A g + B + C z0, coef = [A|B|C|S|H] of votenet_bn_backward_coef):
 * dw0[d,c] += A[c] ug[d,c] + B[c] m[d] + C[c] (sum_e M[d,e] W0[e,c] + m[d] b0[c]),  d < k0. */
int votenet_narrow_wgrad_first(int k0, int c0, const double *moments, const double *ug, const float *coef, const float *w0,
                               const float *b0, float *dw0, void *stream);

/* One launch for every re-laid-out copy of a weight block the GEMMs want (element offsets, device array of 6*nseg longs):
 * table[e] = {src_off, dst_off, rows, cols, ld, transpose}.  transpose != 0: dst[dst_off + c*ld + r] = src[src_off + r*cols + c]
 * (W^T for the input-gradient GEMMs; ld >= rows);  transpose == 0: dst[dst_off + r*ld + c] = src[...] (a copy with a padded
 * leading dimension ld >= cols: 16-byte aligned rows for the 259- and 79-wide layers).  Padding elements are not written. */
int votenet_transpose_segments(int nseg, const long *table, const float *src, float *dst, void *stream);

/* BF3 GEMMs (mlp_fast.hip): the fused forward / input-gradient GEMMs multiply fp32 operands as three bf16 pieces each
 * (x = hi + mid + lo exactly; six v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulate: the error of a product is below the fp32
 * rounding of the product itself) when the weight matrix they are given has a registered IMAGE: the matrix pre-split into the
 * kernel's LDS order, cin * cout * 6 bytes.  No reference counterpart (Tensorpack Conv2D / FullyConnected, utils.py:125-155).
 *   votenet_split_weights: one launch that (re)builds the images of nseg matrices; table (device, 4 longs per segment) = source
 *     address (cin x cout floats, row-major), image address (16-byte aligned), cin (% 16 == 0), cout.  Call after every change of
 *     the weights.
 *   votenet_register_split_weights: from now on a GEMM entry point that receives `w` (with these cin, cout) reads `w3` instead
 *     (w3 == NULL: forget the registration).  Matrices without a registration run on the fp32 MFMA kernels. */
int votenet_split_weights(int nseg, const long *table, void *stream);
int votenet_split_weights_one(const float *w, int cin, int cout, void *image, void *stream); /* one matrix, arguments by value */
int votenet_register_split_weights(const float *w, int cin, int cout, const void *w3);
/* Round 6: the image as TWO fp16 pieces per weight (x = hi + lo, hi = rne16(x), lo = rne16(x - hi); [k/16][hi, lo][k-half][column][8 fp16],
 * cin * cout * 4 bytes) for matrices that multiply FORWARD operands -- activations behind a BatchNorm, coordinates: a product is three
 * v_mfma_f32_32x32x16_f16 instead of six bf16 ones, error against float64 equal to the bf16 x 3 form's (profiles/r06_mfma_f16_denorm.txt).
 * fp16's range applies to both operands: |value| < 65504, values below 2^-24 vanish (a diverged network shows as inf / NaN, loudly).
 * votenet_split_weights_h2 builds such images from the same table; votenet_register_split_weights_pieces(w, ..., w3, 2) registers one
 * (pieces = 3: votenet_register_split_weights).  Only forward-type entry points (votenet_mlp_linear*, votenet_assembled_linear*,
 * votenet_narrow_linear*) have a two-piece kernel; any other entry point that receives `w` ignores a two-piece image and multiplies w
 * itself on the fp32 MFMA kernel. */
int votenet_split_weights_h2(int nseg, const long *table, void *stream);
int votenet_register_split_weights_pieces(const float *w, int cin, int cout, const void *w3, int pieces);

/* ---- split-K for the fused GEMMs of few row tiles (round 5) -------------------------------------------------------------------
 * The GEMMs of the model's static stretch (feature propagation, voting, proposal head: utils.py:286-293, model.py:53-57,89-93) have
 * 2048-8192 rows: 16-64 row tiles of 128 rows, one wavefront per SIMD on half of the CUs with a serial chain of cin / 16 slabs.
 * Split, the contraction of an output tile is shared by 2-4 workgroups (gridDim.z); each stores its partial tile in a workspace and
 * takes the tile's ticket, the last arriver adds the parts in a FIXED order (results do not depend on the arrival order) and runs the
 * usual epilogue -- BatchNorm statistics included -- on the complete tile.  Launchers never allocate: the caller
 *   votenet_mlp_split_k_tickets(t, n)   registers n ZEROED unsigned ints of device memory once (kept alive and untouched while split
 *                                       launches may run; NULL, 0: split-K off -- the default);
 *   votenet_mlp_split_k_floats(rows, cin, cout)   -> floats of workspace a votenet_mlp_linear / votenet_mlp_dgrad_bn /
 *                                       votenet_mlp_dgrad_bn_reduce launch of these sizes would use split (0: it would not split);
 *   votenet_mlp_split_k_arm(ws, floats) hands the workspace (any contents, alive until the launch has run) to the NEXT such launch
 *                                       of the calling thread; the launch consumes it whether it splits or not (NULL, 0: disarm).
 * An unarmed launch never splits (tuning hook: votenet_debug_split_k, votenet_hip_debug.h). */
long votenet_mlp_split_k_floats(long rows, int cin, int cout);
int votenet_mlp_split_k_arm(void *ws, long floats);
int votenet_mlp_split_k_tickets(void *tickets, long n);

/* out (rows x 3) = dz (rows x c) * w3 (3 x c)^T: the xyz columns of an input gradient (dz W[0:3]^T), c % 4 == 0. */
int votenet_rows_dot3(long rows, int c, const float *dz, const float *w3, float *out, void *stream);

/* Gradient of the sample_and_group concat (utils.py:50-57) = GroupPointGrad (tf_grouping_g.cu:61-78) on the
 * feature columns + the gradients of grouped_xyz - tile(new_xyz) on the xyz columns.  The per-row input
 * gradients are given separately: d_rows_feat (b*m*nsample x c) and d_rows_xyz (b*m*nsample x 3):
 *   d_feat[b,idx,:] += d_rows_feat ; d_xyz[b,idx,:] += d_rows_xyz ; d_new_xyz[b,j,:] -= sum_k d_rows_xyz
 * Either pair may be NULL (not needed); outputs are accumulated into (pre-zeroed by the caller).
 * pts_cnt (b,m) from votenet_query_ball_point (may be NULL): rows k >= pts_cnt[b,j] are padding that
 * repeats idx[b,j,0]; they are pre-summed so each group issues pts_cnt instead of nsample atomics. */
int votenet_group_concat_grad(int b, int n, int c, int m, int nsample, const float *d_rows_feat,
                              const float *d_rows_xyz, const int *idx, const int *pts_cnt, float *d_feat, float *d_xyz,
                              float *d_new_xyz, void *stream);

/* ---- deterministic "scatter-add" of the backward pass: gather-sums over the inverse of a grouping (csr.hip) --------------
 * GroupPointGrad / ThreeInterpolateGrad add every gathered row's gradient back to its source point (tf_grouping_g.cu:61-78,
 * tf_interpolate.cpp:131-153): fp32 atomics on a GPU, summation order unspecified.  The inverse of the grouping -- for every
 * point the slots that reference it, ascending (CSR: offsets[npts + 1], order[slots]) -- depends on coordinates only and is
 * built ahead (stable sort); one thread per (point, channel) then sums in that fixed order.
 * out[p, :] = sum over t in [offsets[p], offsets[p+1]) of weight[order[t]] * src[order[t] / div, :]   (weight may be NULL;
 * div = slots per source row: 1 for a grouped tensor (b*m*k rows), 3 for three_interpolate's (b*n, 3) taps). */
/* The inverse itself for a small grouping, in one launch: idx (b, slots) int32 with values in [0, m), m <= 8192 targets per scene ->
 * offsets (b*m + 1), order (b*slots): the slots (flat positions of idx) that reference target p are order[offsets[p]:offsets[p+1]],
 * ascending.  (three_nn's taps: slots = 3 n.  Larger groupings are inverted by the caller, e.g. with a stable sort.) */
int votenet_inverse_index(int b, int slots, int m, const int *idx, int *order, int *offsets, void *stream);
int votenet_csr_gather_sum(long npts, int c, const float *src, const int *order, const int *offsets, const float *weight,
                           int div, float *out, void *stream);
/* The same with the source rows src_pitch floats apart (a column slice of a wider row-major tensor, read in place: the gradient of an
 * FP layer's concat is [d interpolated | d skip], utils.py:286). */
int votenet_csr_gather_sum_pitched(long npts, int c, const float *src, long src_pitch, const int *order, const int *offsets,
                                   const float *weight, int div, float *out, void *stream);
/* votenet_group_linear_backward over the inverse index: s_points is written (not accumulated: no zero fill), the xyz rows of
 * the weight gradient go through per-workgroup partials in scratch (votenet_group_linear_backward_scratch_floats) + an
 * ordered reduction.  Bit-reproducible. */
size_t votenet_group_linear_backward_scratch_floats(int b, int n, int cout);
int votenet_group_linear_backward_csr(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz,
                                      const int *order, const int *offsets, const int *visit /* b*n point ids: the order in which
                                      the points are visited (any fixed permutation; NULL = 0, 1, 2, ...) */, const float *z,
                                      const float *da, const float *coef, int relu, float *s_points, float *dw_xyz, float *dz_out,
                                      float *scratch, void *stream);

/* Optimizer of model.py:240-250 over one flat parameter bucket: per-tensor
 * tf.clip_by_average_norm(g, clip) = g*clip/max(||g||_2/numel, clip) (skipped when clip <= 0), then
 * Adam(lr, beta1, beta2, eps) with bias correction for `step` (1-based).  seg: 2*ntensors element
 * offsets (device, int64), [start, end) of each tensor inside the bucket; g is pre-multiplied by grad_scale
 * (1/world_size after a summing all-reduce).  sumsq_scratch: VOTENET_SUMSQ_SLICES * ntensors floats (that many partial sums per
 * tensor, added in order: the result is bit-reproducible, so data-parallel replicas that hold the same all-reduced gradient stay identical).
 * Adam in TensorFlow's form: lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), p -= lr_t m / (sqrt(v) + eps). */
#define VOTENET_SUMSQ_SLICES 32
int votenet_clip_adam(int ntensors, const long *seg, float *sumsq_scratch, float *p, const float *g, float *m, float *v,
                      float lr, float beta1, float beta2, float eps, int step, float grad_scale, float clip_avg_norm,
                      void *stream);

/* ---------------------------------------------------------------- loss graph (caller of the hot path: SURVEY 8f-1)
 * The reference's total cost (model.py:61-84 vote targets + vote regression, :147-212 proposal assignment,
 * objectness, centre + Chamfer centre, heading, size, semantic losses, :205,228 weights) and its cotangents
 * with respect to votes_xyz (b,n_seeds,3), proposals_xyz (b,n_prop,3) and proposals_output
 * (b,n_prop,5+2*nh+4*ns+nc) -- the three tensors through which the cost reaches the hot path.  Ground
 * truth in the reference's input layout (model.py:22-32): bboxes_xyz / bboxes_lwh (b,n_box,3), bboxes_roty,
 * semantic / heading / size labels (b,n_box), heading_residuals (b,n_box), size_residuals (b,n_box,3); ragged
 * scenes are padded by repeating a box (run.py:14-24; at most 256 boxes per scene).  The three d_* buffers and the
 * workspace (votenet_loss_workspace_floats(b) floats) must be zero on entry.
 * losses (12 floats): total_cost, vote_reg_loss, obj_cls_loss, center_loss (incl. the dual term),
 * heading_cls_loss, heading_residual_loss, size_cls_loss, size_residual_loss, sem_cls_loss, box_loss,
 * #positive, #negative proposals.  No positive (or no negative) proposal: the affected means are NaN, as
 * tf.reduce_mean of an empty tensor. */
int votenet_loss(int b, int n_seeds, int n_prop, int n_box, int nh, int ns, int nc, const float *seeds_xyz,
                 const float *votes_xyz, const float *proposals_xyz, const float *proposals_output,
                 const float *bboxes_xyz, const float *bboxes_lwh, const float *bboxes_roty,
                 const int *semantic_labels, const int *heading_labels, const float *heading_residuals,
                 const int *size_labels, const float *size_residuals, float pos_thr, float neg_thr, float *losses,
                 float *d_votes_xyz, float *d_proposals_xyz, float *d_proposals_output, float *workspace, void *stream);
/* The same with the rows of proposals_output output_pitch floats apart (>= its width): the output of the proposal module's last layer
 * is a column slice of a padded (rows x 128) GEMM result, read in place.  d_proposals_output stays dense (b, n_prop, width). */
int votenet_loss_pitched(int b, int n_seeds, int n_prop, int n_box, int nh, int ns, int nc, const float *seeds_xyz,
                         const float *votes_xyz, const float *proposals_xyz, const float *proposals_output, long output_pitch,
                         const float *bboxes_xyz, const float *bboxes_lwh, const float *bboxes_roty,
                         const int *semantic_labels, const int *heading_labels, const float *heading_residuals,
                         const int *size_labels, const float *size_residuals, float pos_thr, float neg_thr, float *losses,
                         float *d_votes_xyz, float *d_proposals_xyz, float *d_proposals_output, float *workspace, void *stream);
size_t votenet_loss_workspace_floats(int b);

/* Box decode of the predict tower (model.py:100-129): proposals_xyz (b,n_prop,3), proposals_output
 * (b,n_prop,5+2*nh+4*ns+nc), class_mean_size (ns,3; dataset.py:47-49) -> bboxes (b,n_prop,8,3) in the corner
 * order of get_3d_bbox (first four = top face: what votenet_nms3d expects) and scores (b,n_prop) = max class logit. */
int votenet_decode_boxes(int b, int n_prop, int nh, int ns, int nc, const float *proposals_xyz,
                         const float *proposals_output, const float *class_mean_size, float *bboxes, float *scores,
                         void *stream);

/* ---- PointNet++ ops the reference ships but model.py never reaches (kNN grouping, ProbSample) ----
 *
 * Replaces selectionSortLauncher (tf_grouping.cpp:108, kernel tf_grouping_g.cu:83-123; op SelectionSort, Python
 * select_top_k(k, dist), tf_grouping.py:22-32).  dist (b,m,n) -> outi (b,m,n) int, out (b,m,n): each row starts as a copy
 * of the distances / 0..n-1 and goes through k steps of selection sort (the FIRST position of the minimum of [s,n), strict
 * '<', is swapped into s): the first k columns are the k smallest, ascending; the tail is the swapped remainder, exactly
 * as the reference leaves it.  One wavefront per row (row in LDS up to n = 16384, in place in `out` above; four wavefronts per row beyond n = 2048). */
int votenet_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream);

/* knn_point(k, xyz1, xyz2) (tf_grouping.py:47-73) as ONE kernel: the reference tiles both clouds to (b,m,n,c), forms the
 * (b,m,n) squared distances and runs SelectionSort on them, then slices the first k columns.  Here a wavefront builds its
 * row of distances in LDS (channel sum left to right), does the same k selection steps and writes only val (b,m,k) and
 * idx (b,m,k): the (b,m,n) tensors never exist.  xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries.  n > 16384: the rows live
 * in `workspace` (votenet_knn_workspace_bytes(b, n, m) bytes of device memory; may be NULL below that size). */
int votenet_knn_point(int b, int n, int m, int c, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                      void *workspace, void *stream);
size_t votenet_knn_workspace_bytes(int b, int n, int m);

/* Replaces probsampleLauncher (tf_sampling.cpp:65, kernels tf_sampling_g.cu:7-104,197-200; op ProbSample, Python
 * prob_sample(inp, inpr), tf_sampling.py:13-21).  inp_p (b,n) non-negative category weights, inp_r (b,m) uniform draws in
 * [0,1) -> out (b,m) int: the category whose running-sum interval holds r * total.  temp: b*n floats (the running sums,
 * tf_sampling.cpp:83).  The running sum is the reference's float scan with the same association (4-element groups, the
 * scan tree over group totals, compensated carry between 8192-element chunks), so the category boundaries -- and with
 * them every result -- are bit-identical. */
int votenet_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out, void *stream);

/* ---- input pipeline (the step before the path): random subsample + augmentation + ragged ground-truth padding ----
 * Replaces the per-scene numpy code of MyDataFlow.__iter__ (dataset.py:183-189 subsample + depth->camera axes,
 * :219-231 the draws, :262-276 boxes, :302-308 points) and the batch padding of run.py:14-24,60-64.  The DRAWS stay on the
 * host (votenet_amd/input_pipeline.py mirrors the reference's draw order); the kernels apply them.  Per-scene parameters
 * are HOST arrays (they travel as kernel arguments: no staging copies); the bulk data is device memory.
 *
 * Points: scene s owns the raw rows [raw_offset[s], raw_offset[s+1]) of `raw` (raw_stride elements per row, xyz first;
 * doubles when raw_f64 -- np.loadtxt gives float64, sunutils.py:178-180 -- else floats).  Output row j of scene s is raw
 * row choice[s*n_out+j] (the rng.choice(n, POINT_NUM, replace=False) of dataset.py:185-186, drawn by the caller), or, with
 * choice == NULL, row perm_s(j) of a keyed pseudo-random PERMUTATION of the scene's rows (6-round Feistel network with
 * cycle walking; key from `seed` and the scene number s + scene0): a sample without replacement in random order, drawn on
 * the device.  Then, in double precision and in the reference's order: (x,y,z) -> (x,-z,y) when depth_to_camera
 * (sunutils.py:70-77); x = -x when flip[s]&1; z = -z when flip[s]&2; rotation about y by the angle whose cosine / sine
 * are rot_cos[s] / rot_sin[s] (sunutils.py:133-139: x' = c x + s z, z' = -s x + c z); times scale[s]; rounded once to
 * float.  flip == NULL: evaluation, none of the four is applied (dataset.py:302 `if self.training`).
 * Every scene needs at least n_out raw rows (numpy raises for replace=False otherwise). */
int votenet_subsample_augment(int b, int n_out, const void *raw, int raw_f64, int raw_stride, const long *raw_offset,
                              const int *choice, unsigned long long seed, long scene0, int depth_to_camera, const int *flip,
                              const double *rot_cos, const double *rot_sin, const double *scale, float *out, void *stream);

/* Ground truth: scene s owns boxes [box_offset[s], box_offset[s+1]) of center (nbox,3), size (nbox,3; full l,w,h as
 * dataset.py:258), heading (nbox), cls (nbox), all device memory, doubles / ints.  Applies dataset.py:262-276 (flip_x:
 * x = -x, heading = pi - heading; flip_z: z = -z, heading = -heading; rotation of the centre, heading += angle[s]; centre
 * and size times scale[s]), size2class (dataset.py:80-84), angle2class (dataset.py:52-67, python float modulo), the
 * residual normalisations of dataset.py:294-298, and pads every scene to n_box_out rows by repeating its last box
 * (run.py:14-24, np.pad mode='edge').  mean_size: HOST (nc,3) doubles (dataset.py:36-45), nc <= 32.  A scene without
 * boxes is an error (the reference skips such scenes, dataset.py:300).  Outputs are the eight model inputs of
 * model.py:22-32, float / int. */
int votenet_augment_boxes(int b, int n_box_out, const long *box_offset, const double *center, const double *size,
                          const double *heading, const int *cls, const int *flip, const double *angle, const double *rot_cos,
                          const double *rot_sin, const double *scale, const double *mean_size, int nc, int nh,
                          float *bboxes_xyz, float *bboxes_lwh, float *bboxes_roty, int *semantic_labels,
                          int *heading_labels, float *heading_residuals, int *size_labels, float *size_residuals,
                          void *stream);

/* 3D IoU (the arithmetic of tf_nms3d.cpp:178-192) of every box of set A (b,n,8,3) against every box of set B
 * (b,m,8,3) of the same scene -> iou (b,n,m): the detections-vs-ground-truth overlaps of evaluator.py:26-39,122-132. */
int votenet_iou3d_cross(int b, int n, int m, const float *boxes_a, const float *boxes_b, float *iou, void *stream);

/* Plumbing between the path's kernels in one launch: concat (utils.py:286, model.py:53), slices of an input gradient, zero
 * padding of a ragged layer, residual sums (votes = x + offset, model.py:57-61; gradients from two consumers) of row-major
 * (rows x width) tensors.  For every segment s < nseg <= 8:
 *     dst[r*dst_pitch + dst_off + c] = a[r*a_pitch + a_off + c] (+ b[r*b_pitch + b_off + c]),  c < width
 * a == NULL writes zeros; b may be NULL.  Pitches in elements; the segments of one call must not overlap in memory. */
typedef struct votenet_row_segment {
    float *dst;
    int dst_pitch, dst_off, width;
    const float *a;
    int a_pitch, a_off;
    const float *b;
    int b_pitch, b_off;
} votenet_row_segment;
int votenet_row_segments(long rows, int nseg, const votenet_row_segment *seg, void *stream);
/* Up to 32 plain device-to-device copies in ONE launch: dst_s[0:bytes_s) = src_s[0:bytes_s) for s < nseg.  bytes % 4 == 0; 16-byte
 * vectors where both ends of a segment are 16-byte aligned.  Segments must not overlap.  (The inputs of a captured stretch of the
 * train step -- the level outputs, the feature-propagation geometry, the ground truth -- go into the graph's fixed buffers this way:
 * one launch instead of one memcpy node per tensor.) */
typedef struct votenet_copy_segment {
    void *dst;
    const void *src;
    long bytes;
} votenet_copy_segment;
int votenet_copy_segments(int nseg, const votenet_copy_segment *seg, void *stream);
/* Moving averages of every BatchNorm layer in one launch (Tensorpack BatchNorm, momentum 0.9): ema[i] = momentum * ema[i] +
 * factor[i] * batch[i] over flat buffers holding all layers' (scale | shift | mean | var) blocks. */
int votenet_ema_update(long n, float momentum, float *ema, const float *batch, const float *factor, void *stream);

/* ---- PIECE layout of a set-abstraction level (half.hip): the grouped MLP without (most of) the rows that are copies ----
 * A ball with fewer than nsample = 64 neighbours repeats its first hit in the remaining slots (tf_grouping_g.cu:26-29); a repeated
 * slot is an identical row through every layer of the grouped MLP (utils.py:125-132).  The rows of a level are laid out in pieces of
 * 16 (votenet_half_piece_rows()): piece q < G (G = b*m centres) = slots 0..15 of centre q; piece q >= G = slots 16j..16j+15 of centre
 * c, hc[q] = 4c + j, kept for j < ceil(pts_cnt / 16) (plus up to seven all-copy ones so that nh % 8 == 0: a GEMM tile is 128 rows).
 * The dropped pieces hold copies of slot 0 only; slot 0 -- row 0 of a ball's first piece -- stands for them: wh[q] = 1 + 16 * (dropped
 * pieces) (1 for every other piece) is its weight in every sum over the true rows -- the BatchNorm statistics and the affine part
 * B + C z of every BatchNorm backward.  Gradients per compact row are TOTALS over the rows it stands for.  Same results as the full
 * layout up to the association of those sums.
 * votenet_half_groups: pos (G x 3) = index (from G) of piece j = 1..3 of centre c or -1, hc (4G), wh (4G), nh (1 int, device) = number
 * of pieces.  One workgroup, a prefix scan: the layout is the same in every run. */
int votenet_half_groups(int G, const int *pts_cnt, int *pos, int *hc, float *wh, int *nh,
                        int *nh_host /* may be NULL: pinned host memory (mapped into the device's address space) that receives the count too */,
                        void *stream);
int votenet_half_piece_rows(void);
/* votenet_assemble_rows on the piece layout: geo (up to 64G x 4 floats; rows past 16*nh[0] are not written), cntv and moments
 * exactly as votenet_assemble_rows (they run over the true rows).  nh is read on the device: no host synchronisation. */
int votenet_assemble_rows_half(int b, int n, int m, const int *nh, const float *xyz, const float *new_xyz, const int *idx,
                               const int *pts_cnt, const int *hc, float *geo, long long *cntv, double *moments,
                               int *count /* may be NULL; b*n ints, pre-zeroed: += the compact rows that gather every point -- the first
                                             pass of votenet_half_sort_rows, which is then called with counted = 1 */,
                               void *stream);
/* Forward GEMMs on rows = 16*nh compact rows: votenet_assembled_linear / votenet_mlp_linear_pool with the statistics weighted by wh;
 * the pool variant leaves ONE candidate per 16-row piece and channel (zbest / abest, nh x cout): the raw max of z where gamma -- the
 * pooled layer's BatchNorm weight, whose sign is the sign of the scale the pool applies -- is >= 0, the raw min where it is negative,
 * first occurrence; votenet_bn_pool_finalize_half joins a centre's pieces (ties -> the earlier piece, the first occurrence as in the
 * 64-row epilogue; argmax = slot 0..63). */
/* nh_dev (every entry below that has it; may be NULL): the piece count when only the device knows it -- a level whose geometry is made
 * inside the step (the proposal module groups the votes) cannot wait for the count on the host without draining the queue.  The caller
 * then passes its upper bound as rows / nh (buffers sized for it) and the kernels stop at 16 * nh_dev[0] rows. */
int votenet_assembled_linear_half(long rows, int c0, int cout, const float *geo, const float *P, const float *wx, const float *in_scale,
                                  const float *in_shift, const votenet_bn_raw *in_bn, int in_relu, const float *w, const float *bias,
                                  float *z, double *stats, const float *wh, const int *nh_dev, void *stream);
int votenet_mlp_linear_half(const float *x, const float *in_scale, const float *in_shift, int in_relu, long rows, int cin, int cout,
                            const float *w, const float *bias, float *z, const int *nh_dev, void *stream);
int votenet_mlp_linear_pool_half(const votenet_mlp_input *in, long rows, int cin, int cout, const float *w, const float *bias, float *z,
                                 double *stats, const float *wh, const float *gamma, float *zbest, int *abest, const int *nh_dev,
                                 void *stream);
int votenet_bn_pool_finalize_half(long G, int c, const float *zbest, const int *abest, const int *pos, const float *scale, const float *shift,
                                  const votenet_bn_raw *bn, int relu, float *out, int *argmax, float *zsel, void *stream);
/* Backward on the compact rows (gout / argmax / zsel stay per centre): the scatter of the pooled layer's input gradient (also scales
 * the dense part of a piece's row 0 by wh), its Gram matrix a^T diag(w) a, its sparse weight-gradient part with weighted column sums, the
 * second layer's weight / input gradient over the assembled first layer, and the first layer's scatter to the points. */
int votenet_pool_dgrad_scatter_half(long nh, int G, int cin, int cout, const float *gout, const int *argmax, const float *zsel,
                                    const float *coef, int relu, const float *wT, float *da, const int *hc, const float *wh,
                                    const float *below_z, const float *below_scale, const float *below_shift, const float *below_mean,
                                    const float *below_var, float eps, int below_relu, double *below_sums,
                                    const votenet_coef_tail *below_tail, const int *nh_dev, void *stream);
int votenet_mlp_gram_half(long rows, int c, const float *z, const float *scale_shift, int relu, const float *wh, float *gram,
                          const int *nh_dev, void *stream);
int votenet_pool_wgrad_sparse_half(long nh, int G, int cin, int cout, const float *xz, const float *in_scale, const float *in_shift,
                                   int in_relu, const float *gout, const int *argmax, const float *zsel, const float *coef, int relu,
                                   float *dw, float *colsum, const int *hc, const float *wh, const int *nh_dev, void *stream);
/* The same walking CENTRES (round 5): pos = the layout's (G, 3) table of every centre's pieces j >= 1 (votenet_half_groups); all kept
 * pieces of a ball are staged together and every channel reads its arg-max row once -- the piece form sends every wavefront through
 * the row loop once per piece with 1 / pieces of its lanes live.  Same sums in another association.
 * Limit: nh * 16 * cin * 4 < 2^31 bytes (32-bit buffer offsets; VOTENET_E_INVALID above it -- call the piece form there). */
int votenet_pool_wgrad_sparse_half_centres(long nh, int G, int cin, int cout, const float *xz, const float *in_scale, const float *in_shift,
                                           int in_relu, const float *gout, const int *argmax, const float *zsel, const float *coef,
                                           int relu, float *dw, float *colsum, const int *pos, const float *wh, const int *nh_dev,
                                           void *stream);
int votenet_assembled_wgrad_bn_half(long rows, int c0, int cout, const float *geo, const float *P, const float *wx, const float *in_scale,
                                    const float *in_shift, int in_relu, const float *da, const float *z, const float *coef, int relu,
                                    const float *wh, float *dw, const int *nh_dev, void *stream);
int votenet_assembled_dgrad_bn_reduce_half(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                                           const float *wT, float *da_prev, const float *geo, const float *P, const float *wx,
                                           const float *scale_prev, const float *shift_prev, const float *mean_prev,
                                           const float *var_prev, float eps, int relu_prev, double *sums,
                                           const votenet_coef_tail *tail /* may be NULL */, const float *wh, const int *nh_dev, void *stream);
/* The scatter to the points without one atomic per row (an fp32 atomic costs an L2 channel ~14 cycles per line): with the geometry,
 * votenet_half_sort_rows buckets the compact rows by the point they gather (order: 64G ints, 16*nh[0] written; work: npts ints);
 * votenet_group_linear_backward_sorted then sums a point's consecutive rows in a register and stores S point by point
 * (s_points pre-zeroed; atomics only where a chunk of 64 entries shares a point with its neighbour): the piece layout's form of
 * votenet_group_linear_backward_assembled (da = total gradients per compact row; no xyz gradient). */
int votenet_half_sort_rows(int npts, int G, const int *nh, const float *geo, int *work, int counted, int *order, void *stream);
int votenet_group_linear_backward_sorted(long nh, int cout, const int *order, const float *geo, const float *wh, const float *P,
                                         const float *wx, const float *da, const float *coef, int relu, float *s_points, float *dw_xyz,
                                         const int *nh_dev, void *stream);
/* The xyz gradient of such a first layer (the proposal module, model.py:89: its input coordinates are the votes).  With
 * dz0 W[0:3]^T linear in dz0, the per-point part is S W[0:3]^T (votenet_rows_dot3 on s_points) and the per-centre part -(sum of the
 * centre's dz0 rows) W[0:3]^T: votenet_half_centre_sums leaves T (G x cout) = MINUS the sum of the total gradients dz0 of every centre's
 * compact rows (rebuilt from da, geo, P as in votenet_group_linear_backward_sorted); votenet_rows_dot3(T, W[0:3]) is d new_xyz. */
/* The first layer's backward DECOMPOSED over the points (round 4; utils.py:125-127 is the loop whose backward this is, tf_grouping_g.cu:61-78
 * the scatter).  dz0 = A g + B + C z0 (g = da0 . relu'(bn0(z0))) is linear in everything the points and weights receive, so the
 * input-gradient GEMM of the layer above is a PLAIN one (votenet_mlp_dgrad_bn_half: no epilogue gathers) and
 *   votenet_group_linear_backward_masked  one pass over the compact rows bucketed by point (votenet_half_sort_rows): sg (points x cout,
 *       pre-zeroed) = scatter of the MASKED total gradients; ug (3 x cout) = sum_r dxyz(r)^T g[r] and sums (2 cout doubles) = (sum g,
 *       sum g zhat0) -- the first layer's BatchNorm-backward sums -- are WRITTEN by the kernel's last workgroup from `part`
 *       (votenet_group_linear_backward_masked_slots() x 5 x cout doubles, pre-zeroed: the workgroups' atomics spread over that many
 *       address sets), together with the coefficient vector of the REQUIRED tail (votenet_coef_tail)
 *   votenet_assembled_point_grad          over the POINTS: sg -> S = A sg + cnt_p B + C (cnt_p P[p] + V_p wx) in place (cntv: votenet_assemble_rows);
 *       vp (votenet_group_linear_backward_masked_slots() x 3 x cout, pre-zeroed) += partial sums of sum_p V_p^T P[p]
 *   votenet_assembled_wx_finish           dw_xyz (3 x cout) += A ug + B (sum dxyz) + C (sum of the nparts slots of vp + (sum dxyz dxyz^T) wx)
 * give the same S and dW[0:3] as votenet_group_linear_backward_sorted up to the association of the sums. */
int votenet_mlp_dgrad_bn_half(long rows, int c, int cout, const float *da, const float *zsrc, const float *coef, int relu,
                              const float *wT, float *da_prev, const float *wh, const int *nh_dev, void *stream);
int votenet_group_linear_backward_masked(long nh, int cout, const int *order, const float *geo, const float *P, const float *wx,
                                         const float *da, const float *scale, const float *shift, const float *mean, const float *var,
                                         float eps, int relu, float *sg, float *ug, double *sums, double *part, const votenet_coef_tail *tail,
                                         const int *nh_dev, void *stream);
int votenet_group_linear_backward_masked_slots(void);
int votenet_assembled_point_grad(long npts, int cout, const float *P, const long long *cntv, const float *wx, const float *coef, float *s,
                                 float *vp, void *stream);
int votenet_assembled_wx_finish(int cout, const float *coef, const float *ug, const float *vp, int nparts, const double *moments,
                                const float *wx, float *dw_xyz, void *stream);
int votenet_half_centre_sums(long G, int cout, const int *pos, const float *geo, const float *wh, const float *P, const float *wx,
                             const float *da, const float *coef, int relu, float *T, void *stream);
/* The narrow first layer (sa1) on the same layout: u8 (up to 64G x 8 floats; moments over the true rows), the second layer's GEMMs. */
int votenet_narrow_rows_half(int b, int n, int m, int c, const int *nh, const float *xyz, const float *new_xyz, const float *feat,
                             const int *idx, const int *pts_cnt, const int *hc, float *u8, double *moments, void *stream);
int votenet_narrow_linear_half(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0, const float *in_scale,
                               const float *in_shift, const votenet_bn_raw *in_bn, int in_relu, const float *w, const float *bias, float *z,
                               double *stats, const float *wh, void *stream);
int votenet_narrow_wgrad_bn_half(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0, const float *in_scale,
                                 const float *in_shift, int in_relu, const float *da, const float *z, const float *coef, int relu,
                                 const float *wh, float *dw, void *stream);
int votenet_narrow_dgrad_bn_reduce_half(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef, int relu,
                                        const float *wT, const float *u8, const float *w0, const float *b0, const float *scale0,
                                        const float *shift0, const float *mean0, const float *var0, float eps, int relu0, double *sums,
                                        double *ug, const votenet_coef_tail *tail /* may be NULL */, const float *wh, void *stream);
/* The narrow first layer's ReLU mask recorded by the FORWARD pass (round 4): votenet_narrow_linear_masked is votenet_narrow_linear(_half)
 * (wh may be NULL) that also writes mask (rows x c0 / 16 16-bit words, 8-byte aligned, c0 % 64 == 0): bit k % 16 of word [row][k / 16] =
 * [relu(bn0(z0[row, k])) > 0]; votenet_narrow_dgrad_bn_reduce_masked is votenet_narrow_dgrad_bn_reduce_half reading that mask instead of
 * rebuilding z0 per accumulator element, with sums[c0:2 c0] derived in the (required) coefficient tail from ug and sums[0:c0] -- z0 is
 * linear in u: sum g z0[:, c] = sum_d W0[d, c] ug[d, c] + b0[c] sum g.  (utils.py:125-127 is the loop whose backward this is.) */
int votenet_narrow_linear_masked(long rows, int k0, int c0, int cout, const float *u8, const float *w0, const float *b0,
                                 const float *in_scale, const float *in_shift, const votenet_bn_raw *in_bn, int in_relu, const float *w,
                                 const float *bias, float *z, double *stats, const float *wh, unsigned short *mask, void *stream);
int votenet_narrow_dgrad_bn_reduce_masked(long rows, int c, int c0, int k0, const float *da, const float *zsrc, const float *coef, int relu,
                                          const float *wT, const float *u8, const float *w0, const float *b0, const float *scale0,
                                          const float *shift0, const float *mean0, const float *var0, float eps, int relu0, double *sums,
                                          double *ug, const votenet_coef_tail *tail, const float *wh, const void *mask, void *stream);
#ifdef __cplusplus
}
#endif
#endif /* VOTENET_HIP_H */
