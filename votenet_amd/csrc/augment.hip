// augment.hip -- the step before the hot path: random subsample of the raw depth cloud to n_out points, flip / y-rotation /
// scale augmentation, depth->camera axes, and the same augmentation + label encoding + ragged padding of the ground-truth
// boxes (dataset.py:183-189, 219-231, 262-308; run.py:14-24).  The reference runs this per scene in numpy inside N/2 ZMQ
// worker processes; here one launch serves a batch: a row gather (random 12-24 byte reads, coalesced 12-byte writes), pure
// HBM traffic.  Arithmetic is double precision, un-fused, in the reference's order, rounded once to float -- what numpy
// float64 followed by the float32 feed does.
#include "common.h"

namespace votenet {

constexpr int AUG_CHUNK = 16; // scenes per launch: their parameters travel as kernel arguments

struct AugScenes {
    long off[AUG_CHUNK + 1];
    double c[AUG_CHUNK], s[AUG_CHUNK], scale[AUG_CHUNK], angle[AUG_CHUNK];
    int flip[AUG_CHUNK];
    unsigned key[AUG_CHUNK];
};

__host__ __device__ inline unsigned lowbias32(unsigned x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// keyed permutation of [0, n): balanced Feistel network over the next even number of bits, cycle walking back into range
__device__ __forceinline__ long feistel_perm(long j, long n, unsigned key, int half)
{
    const unsigned mask = (1u << half) - 1u;
    unsigned long long x = (unsigned long long)j;
    do {
        unsigned l = (unsigned)(x >> half) & mask, r = (unsigned)x & mask;
#pragma unroll
        for (int round = 0; round < 6; round++) {
            const unsigned f = lowbias32(r + key + 0x9E3779B9u * (unsigned)(round + 1)) & mask;
            const unsigned nl = r;
            r = l ^ f;
            l = nl;
        }
        x = ((unsigned long long)l << half) | r;
    } while (x >= (unsigned long long)n);
    return (long)x;
}

template <typename T>
__global__ __launch_bounds__(256) void subsample_augment_kernel(AugScenes P, int n_out, const T *__restrict__ raw, int stride,
                                                                const int *__restrict__ choice, int to_camera, int train,
                                                                float *__restrict__ out)
{
    const int sc = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_out) return;
    const long n = P.off[sc + 1] - P.off[sc];
    long i;
    if (choice) {
        i = choice[(long)sc * n_out + j];
        i = i < 0 ? 0 : (i >= n ? n - 1 : i); // validated on the host side of the Python mirror; never read out of range
    } else {
        int bits = 2;
        while ((1ll << bits) < n) bits += 2;
        i = feistel_perm(j, n, P.key[sc], bits >> 1);
    }
    const T *p = raw + (P.off[sc] + i) * stride;
    double x = (double)p[0], y = (double)p[1], z = (double)p[2];
    if (to_camera) { // sunutils.py:70-77: (x, y, z) -> (x, -z, y)
        const double t = y;
        y = -z;
        z = t;
    }
    if (train) {
        if (P.flip[sc] & 1) x = -x; // dataset.py:303-306
        if (P.flip[sc] & 2) z = -z;
        const double c = P.c[sc], s = P.s[sc];
        const double xr = c * x + s * z; // roty(a) @ p, sunutils.py:133-139
        const double zr = -s * x + c * z;
        x = xr * P.scale[sc]; // dataset.py:308
        y = y * P.scale[sc];
        z = zr * P.scale[sc];
    }
    float *o = out + ((long)sc * n_out + j) * 3;
    o[0] = (float)x;
    o[1] = (float)y;
    o[2] = (float)z;
}

__device__ __forceinline__ double py_mod(double a, double m) // CPython float %: fmod, then the sign of the divisor
{
    double r = fmod(a, m);
    if (r != 0.0) {
        if ((m < 0.0) != (r < 0.0)) r += m;
    } else {
        r = copysign(0.0, m);
    }
    return r;
}

struct BoxOut {
    float *xyz, *lwh, *roty, *hres, *sres;
    int *sem, *hlab, *slab;
};
struct MeanSizes {
    double v[32][3];
};

__global__ void augment_boxes_kernel(AugScenes P, int bb, const double *__restrict__ center, const double *__restrict__ size,
                                     const double *__restrict__ heading, const int *__restrict__ cls, int train, MeanSizes M,
                                     int nc, int nh, BoxOut O)
{
    const int sc = blockIdx.x;
    for (int slot = threadIdx.x; slot < bb; slot += blockDim.x) {
        const long cnt = P.off[sc + 1] - P.off[sc];
        const long g = P.off[sc] + (slot < cnt ? slot : cnt - 1); // np.pad mode='edge': repeat the last box
        double cx = center[g * 3], cy = center[g * 3 + 1], cz = center[g * 3 + 2];
        double l = size[g * 3], w = size[g * 3 + 1], h = size[g * 3 + 2];
        double ang = heading[g];
        const int k = cls[g];
        const int km = k < 0 ? 0 : (k >= nc ? nc - 1 : k); // mean-size row; labels are written as given
        const double pi = 3.141592653589793;
        if (train) { // dataset.py:262-276
            if (P.flip[sc] & 1) {
                cx = -cx;
                ang = pi - ang;
            }
            if (P.flip[sc] & 2) {
                cz = -cz;
                ang = -ang;
            }
            const double c = P.c[sc], s = P.s[sc];
            const double xr = c * cx + s * cz, zr = -s * cx + c * cz;
            ang += P.angle[sc];
            cx = xr * P.scale[sc];
            cy = cy * P.scale[sc];
            cz = zr * P.scale[sc];
            l *= P.scale[sc];
            w *= P.scale[sc];
            h *= P.scale[sc];
        }
        // angle2class, dataset.py:61-67
        const double two_pi = 2 * pi;
        const double a = py_mod(ang, two_pi);
        const double per = two_pi / (double)nh;
        const double shifted = py_mod(a + per / 2, two_pi);
        const int cid = (int)(shifted / per);
        const double res = shifted - (cid * per + per / 2);
        const long o = (long)sc * bb + slot;
        O.xyz[o * 3] = (float)cx;
        O.xyz[o * 3 + 1] = (float)cy;
        O.xyz[o * 3 + 2] = (float)cz;
        O.lwh[o * 3] = (float)l;
        O.lwh[o * 3 + 1] = (float)w;
        O.lwh[o * 3 + 2] = (float)h;
        O.roty[o] = (float)ang;
        O.sem[o] = k;
        O.hlab[o] = cid;
        O.hres[o] = (float)(res / (pi / nh)); // dataset.py:296
        O.slab[o] = k;
        O.sres[o * 3] = (float)((l - M.v[km][0]) / M.v[km][0]); // dataset.py:83,298
        O.sres[o * 3 + 1] = (float)((w - M.v[km][1]) / M.v[km][1]);
        O.sres[o * 3 + 2] = (float)((h - M.v[km][2]) / M.v[km][2]);
    }
}

static unsigned scene_key(unsigned long long seed, long scene)
{
    const unsigned hi = lowbias32((unsigned)(seed >> 32) + 0x632BE5ABu * (unsigned)(scene + 1));
    return lowbias32((unsigned)seed ^ hi ^ (0x85EBCA6Bu * (unsigned)(scene + 1)));
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_subsample_augment(int b, int n_out, const void *raw, int raw_f64, int raw_stride, const long *raw_offset,
                                         const int *choice, unsigned long long seed, long scene0, int depth_to_camera,
                                         const int *flip, const double *rot_cos, const double *rot_sin, const double *scale,
                                         float *out, void *stream)
{
    VN_REQUIRE(b > 0 && n_out > 0, "subsample_augment: b and n_out must be positive, got %d, %d", b, n_out);
    VN_REQUIRE(raw && raw_offset && out, "subsample_augment: null pointer");
    VN_REQUIRE(raw_stride >= 3, "subsample_augment: raw rows need at least 3 elements, got %d", raw_stride);
    VN_REQUIRE(!flip || (rot_cos && rot_sin && scale), "subsample_augment: flip given without rot_cos / rot_sin / scale");
    for (int s = 0; s < b; s++) {
        const long n = raw_offset[s + 1] - raw_offset[s];
        VN_REQUIRE(n >= n_out, "subsample_augment: scene %d has %ld points, cannot take %d without replacement", s, n, n_out);
        VN_REQUIRE(n < (1l << 31), "subsample_augment: scene %d has %ld points (limit 2^31)", s, n);
    }
    for (int s0 = 0; s0 < b; s0 += AUG_CHUNK) {
        const int ns = b - s0 < AUG_CHUNK ? b - s0 : AUG_CHUNK;
        AugScenes P = {};
        for (int s = 0; s < ns; s++) {
            P.off[s] = raw_offset[s0 + s];
            P.off[s + 1] = raw_offset[s0 + s + 1];
            P.key[s] = scene_key(seed, scene0 + s0 + s);
            if (flip) {
                P.flip[s] = flip[s0 + s];
                P.c[s] = rot_cos[s0 + s];
                P.s[s] = rot_sin[s0 + s];
                P.scale[s] = scale[s0 + s];
            }
        }
        const dim3 grid((n_out + 255) / 256, ns);
        const int *ch = choice ? choice + (long)s0 * n_out : nullptr;
        float *o = out + (long)s0 * n_out * 3;
        if (raw_f64)
            hipLaunchKernelGGL(subsample_augment_kernel<double>, grid, dim3(256), 0, as_stream(stream), P, n_out,
                               (const double *)raw, raw_stride, ch, depth_to_camera, flip ? 1 : 0, o);
        else
            hipLaunchKernelGGL(subsample_augment_kernel<float>, grid, dim3(256), 0, as_stream(stream), P, n_out,
                               (const float *)raw, raw_stride, ch, depth_to_camera, flip ? 1 : 0, o);
    }
    return check_launch("subsample_augment");
}

extern "C" int votenet_augment_boxes(int b, int n_box_out, const long *box_offset, const double *center, const double *size,
                                     const double *heading, const int *cls, const int *flip, const double *angle,
                                     const double *rot_cos, const double *rot_sin, const double *scale, const double *mean_size,
                                     int nc, int nh, float *bboxes_xyz, float *bboxes_lwh, float *bboxes_roty,
                                     int *semantic_labels, int *heading_labels, float *heading_residuals, int *size_labels,
                                     float *size_residuals, void *stream)
{
    VN_REQUIRE(b > 0 && n_box_out > 0, "augment_boxes: b and n_box_out must be positive, got %d, %d", b, n_box_out);
    VN_REQUIRE(box_offset && center && size && heading && cls && mean_size, "augment_boxes: null input pointer");
    VN_REQUIRE(bboxes_xyz && bboxes_lwh && bboxes_roty && semantic_labels && heading_labels && heading_residuals &&
                   size_labels && size_residuals,
               "augment_boxes: null output pointer");
    VN_REQUIRE(nc > 0 && nc <= 32 && nh > 0, "augment_boxes: nc must be in [1, 32] and nh positive, got %d, %d", nc, nh);
    VN_REQUIRE(!flip || (angle && rot_cos && rot_sin && scale), "augment_boxes: flip given without angle / rot_cos / rot_sin / scale");
    for (int s = 0; s < b; s++) {
        const long n = box_offset[s + 1] - box_offset[s];
        VN_REQUIRE(n >= 1, "augment_boxes: scene %d has no boxes", s);
        VN_REQUIRE(n <= n_box_out, "augment_boxes: scene %d has %ld boxes, more than n_box_out = %d", s, n, n_box_out);
    }
    MeanSizes M = {};
    for (int k = 0; k < nc; k++)
        for (int a = 0; a < 3; a++) M.v[k][a] = mean_size[k * 3 + a];
    for (int s0 = 0; s0 < b; s0 += AUG_CHUNK) {
        const int ns = b - s0 < AUG_CHUNK ? b - s0 : AUG_CHUNK;
        AugScenes P = {};
        for (int s = 0; s < ns; s++) {
            P.off[s] = box_offset[s0 + s];
            P.off[s + 1] = box_offset[s0 + s + 1];
            if (flip) {
                P.flip[s] = flip[s0 + s];
                P.c[s] = rot_cos[s0 + s];
                P.s[s] = rot_sin[s0 + s];
                P.scale[s] = scale[s0 + s];
                P.angle[s] = angle[s0 + s];
            }
        }
        const long o = (long)s0 * n_box_out;
        BoxOut O = {bboxes_xyz + o * 3, bboxes_lwh + o * 3, bboxes_roty + o, heading_residuals + o, size_residuals + o * 3,
                    semantic_labels + o, heading_labels + o, size_labels + o};
        hipLaunchKernelGGL(augment_boxes_kernel, dim3(ns), dim3(64), 0, as_stream(stream), P, n_box_out, center, size, heading,
                           cls, flip ? 1 : 0, M, nc, nh, O);
    }
    return check_launch("augment_boxes");
}
