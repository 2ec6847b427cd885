// sampling.hip -- gather_point and its gradient for gfx950 (FPS lives in fps.hip).
//
// Replaces gatherpointKernel / scatteraddpointKernel (tf_ops/sampling/tf_sampling_g.cu:172-192,206-211).
#include "common.h"

namespace votenet {

// gather: out[s,j,:] = inp[s,idx[s,j],:]   (tf_sampling_g.cu:172-181)
__global__ void gather_point_kernel(int n, int m, long total, const float *__restrict__ inp,
                                    const int *__restrict__ idx, float *__restrict__ out)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / 3;
        const int ch = (int)(e - row * 3);
        const long s = row / m;
        const int a = idx[row];
        out[e] = inp[((size_t)s * n + a) * 3 + ch];
    }
}

// scatter-add: inp_g[s,idx[s,j],:] += out_g[s,j,:]   (tf_sampling_g.cu:183-192)
__global__ void gather_point_grad_kernel(int n, int m, long total, const float *__restrict__ out_g,
                                         const int *__restrict__ idx, float *__restrict__ inp_g)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long row = e / 3;
        const int ch = (int)(e - row * 3);
        const long s = row / m;
        const int a = idx[row];
        unsafeAtomicAdd(&inp_g[((size_t)s * n + a) * 3 + ch], out_g[e]);
    }
}

} // namespace votenet

using namespace votenet;

static inline int grid_for(long total, int block)
{
    long g = (total + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int votenet_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0, "GatherPoint expects (batch_size,num_points,3) inp shape"); // tf_sampling.cpp:131
    const long total = (long)b * m * 3;
    if (total == 0) return VOTENET_OK;
    VN_REQUIRE(inp && idx && out, "GatherPoint: null buffer");
    hipLaunchKernelGGL(gather_point_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), n, m, total, inp,
                       idx, out);
    return check_launch("gather_point");
}

extern "C" int votenet_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                         void *stream)
{
    VN_REQUIRE(b >= 0 && n > 0 && m >= 0, "GatherPointGradGpuOp expects (batch_size,num_points,3) inp"); // :156
    const long total = (long)b * m * 3;
    if (total == 0) return VOTENET_OK;
    VN_REQUIRE(out_g && idx && inp_g, "GatherPointGrad: null buffer");
    hipLaunchKernelGGL(gather_point_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), n, m, total,
                       out_g, idx, inp_g);
    return check_launch("gather_point_grad");
}

// ---- the reference's own launcher names, C++ linkage, exact signatures (tf_sampling.cpp:94,125,150)
// so that tf_sampling.cpp links against this library unchanged.  Null stream, as the reference.
VN_EXPORT void gatherpointLauncher(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    votenet_gather_point(b, n, m, inp, idx, out, nullptr);
}
VN_EXPORT void scatteraddpointLauncher(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    votenet_gather_point_grad(b, n, m, out_g, idx, inp_g, nullptr);
}
