/*
 * ref_sampling_gpu_driver.cpp -- builds the REFERENCE's own device kernels of tf_ops/sampling into
 * oracle/_ref/libref_sampling_gpu.so, for gfx950.  TEST INFRASTRUCTURE ONLY: loaded by tests/ (GPU marker) and by
 * tests/golden/make_ref_gpu_golden.py, never by votenet_amd/.
 *
 * tf_ops/sampling/tf_sampling_g.cu includes nothing and calls no CUDA runtime function: its kernels and `<<< >>>` launchers are
 * the common subset of CUDA and HIP, so `hipcc -x hip` compiles the file where it lies under the reference tree (-DREF_SRC=...);
 * nothing of it is copied here and nothing is written in its stead.  Built with -ffp-contract=off: the arithmetic model of SURVEY
 * appendix A.1 (the distance expression evaluated as written) -- contraction is a property of the compiler flags, not of the source.
 * The launchers use the null stream; every entry point here waits for the device and returns the hipError_t.
 *
 * What the TF op wrappers do before the launch is left to the caller (tests): temp of 32*n floats for the sampling
 * (tf_sampling.cpp:115), b*n for ProbSample (:86), inp_g zeroed before the scatter-add (:174).
 */
#include <hip/hip_runtime.h>
#include REF_SRC

static int done()
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    return (int)e;
}

extern "C" int ref_gpu_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out)
{
    farthestpointsamplingLauncher(b, n, m, inp, temp, out); /* tf_sampling_g.cu:203-205 */
    return done();
}
extern "C" int ref_gpu_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    gatherpointLauncher(b, n, m, inp, idx, out); /* :206-208 */
    return done();
}
extern "C" int ref_gpu_scatter_add_point(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    scatteraddpointLauncher(b, n, m, out_g, idx, inp_g); /* :209-211 */
    return done();
}
extern "C" int ref_gpu_cumsum(int b, int n, const float *inp, float *out)
{
    cumsumLauncher(b, n, inp, out); /* :194-196 */
    return done();
}
extern "C" int ref_gpu_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out)
{
    probsampleLauncher(b, n, m, inp_p, inp_r, temp, out); /* :198-201 */
    return done();
}
