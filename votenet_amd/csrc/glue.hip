// glue.hip -- the plumbing between the path's kernels as ONE small kernel instead of framework element-wise launches (gfx950).
//
// Between two GEMMs the reference's graph concatenates, slices, pads and adds small per-point tensors: the FP layers' concat
// (utils.py:286), the voting input [seeds_xyz | seeds_points] and votes = x + offset (model.py:53-61), the split of a layer's
// input gradient back into its sources, residual sums of gradients arriving from two consumers.  Done with tensor-library ops
// each is its own launch (torch.cat, .contiguous(), F.pad = fill + copy, +): ~30 launches of 3-8 us per train step, all on the
// dependent chain.  votenet_row_segments does any such regrouping of row-major (rows x width) tensors in one launch:
//
//     for every segment s (up to 8):   dst_s[r, dst_off_s + c] = a_s[r, a_off_s + c] (+ b_s[r, b_off_s + c])   c < width_s
//                                      width_s columns of zeros when a_s == NULL (padding of a ragged layer)
//
// Every tensor has its own row pitch (elements), so a slice of a wider tensor is read or written in place.  The tensors are small
// (<= 8192 x 320 floats): one thread per element of the widest segment layout, no vectorisation games -- the launch latency is the cost.
#include "common.h"

namespace votenet {

constexpr int kMaxSeg = 8;
struct RowSegs {
    int nseg;
    float *dst[kMaxSeg];
    const float *a[kMaxSeg], *b[kMaxSeg];
    int dst_pitch[kMaxSeg], dst_off[kMaxSeg], width[kMaxSeg], a_pitch[kMaxSeg], a_off[kMaxSeg], b_pitch[kMaxSeg], b_off[kMaxSeg];
    int cum[kMaxSeg + 1]; // prefix sums of the widths: a thread's column -> its segment
};

// tpr threads walk the columns of a row (a power of two <= 256 chosen by the launcher: the smallest that covers the row), 256 / tpr rows
// per workgroup.  A thread finds its column's segment, pointers and pitches ONCE and then walks its rows: no division, no per-element
// search (the first version did both per element and ran a 24 MB add at 1-2 TB/s, 11-23 us on the train step's critical path).
__global__ __launch_bounds__(256) void row_segments_kernel(long rows, RowSegs S, int tpr_log2)
{
    const int total = S.cum[S.nseg];
    const int tpr = 1 << tpr_log2, rpw = 256 >> tpr_log2;
    const int tc = threadIdx.x & (tpr - 1);
    const long r0 = (long)blockIdx.x * rpw + (threadIdx.x >> tpr_log2), rstep = (long)gridDim.x * rpw;
    for (int col = tc; col < total; col += tpr) {
        int s = 0;
#pragma unroll
        for (int t = 1; t < kMaxSeg; t++) s += (t < S.nseg && col >= S.cum[t]) ? 1 : 0;
        const int c = col - S.cum[s];
        const float *__restrict__ a = S.a[s];
        const float *__restrict__ b = S.b[s];
        float *__restrict__ d = S.dst[s];
        const long ap = S.a_pitch[s], bp = S.b_pitch[s], dp = S.dst_pitch[s];
        const int ao = S.a_off[s] + c, bo = S.b_off[s] + c, doff = S.dst_off[s] + c;
        if (a && b) {
#pragma unroll 4
            for (long r = r0; r < rows; r += rstep) d[r * dp + doff] = a[r * ap + ao] + b[r * bp + bo];
        } else if (a) {
#pragma unroll 4
            for (long r = r0; r < rows; r += rstep) d[r * dp + doff] = a[r * ap + ao];
        } else {
            for (long r = r0; r < rows; r += rstep) d[r * dp + doff] = 0.0f;
        }
    }
}

} // namespace votenet

using namespace votenet;

extern "C" int votenet_row_segments(long rows, int nseg, const votenet_row_segment *seg, void *stream)
{
    VN_REQUIRE(rows >= 0 && nseg > 0 && nseg <= kMaxSeg && seg, "row_segments expects rows >= 0 and 1..8 segments");
    if (rows == 0) return VOTENET_OK;
    RowSegs S = {};
    S.nseg = nseg;
    for (int i = 0; i < nseg; i++) {
        const votenet_row_segment &g = seg[i];
        VN_REQUIRE(g.dst && g.width > 0 && g.dst_off >= 0 && g.dst_off + g.width <= g.dst_pitch, "row_segments: segment %d does not fit its destination row", i);
        VN_REQUIRE(!g.a || (g.a_off >= 0 && g.a_off + g.width <= g.a_pitch), "row_segments: segment %d reads past its source row", i);
        VN_REQUIRE(!g.b || (g.a && g.b_off >= 0 && g.b_off + g.width <= g.b_pitch), "row_segments: segment %d: second source without a first / past its row", i);
        S.dst[i] = g.dst;
        S.a[i] = g.a;
        S.b[i] = g.b;
        S.dst_pitch[i] = g.dst_pitch;
        S.dst_off[i] = g.dst_off;
        S.width[i] = g.width;
        S.a_pitch[i] = g.a_pitch;
        S.a_off[i] = g.a_off;
        S.b_pitch[i] = g.b_pitch;
        S.b_off[i] = g.b_off;
        S.cum[i + 1] = S.cum[i] + g.width;
    }
    int tpr_log2 = 0;
    while (tpr_log2 < 8 && (1 << tpr_log2) < S.cum[nseg]) tpr_log2++;
    const int rpw = 256 >> tpr_log2;
    long grid = (rows + rpw - 1) / rpw;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(row_segments_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), rows, S, tpr_log2);
    return check_launch("row_segments");
}

// Plain copies, many per launch (struct by value: no table in device memory, so a caller may refill the arguments per launch).
namespace votenet {
constexpr int kMaxCopy = 32;
struct CopySegs {
    int nseg;
    char *dst[kMaxCopy];
    const char *src[kMaxCopy];
    long bytes[kMaxCopy];
};
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs S)
{
    const int s = blockIdx.y;
    if (s >= S.nseg) return;
    char *__restrict__ d = S.dst[s];
    const char *__restrict__ a = S.src[s];
    const long nb = S.bytes[s];
    if ((((uintptr_t)d | (uintptr_t)a) & 15) == 0) {
        const long n16 = nb >> 4;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256)
            reinterpret_cast<float4 *>(d)[i] = reinterpret_cast<const float4 *>(a)[i];
        const long done = n16 << 4; // the tail (< 16 bytes, a multiple of 4)
        if (blockIdx.x == 0 && threadIdx.x < (nb - done) / 4)
            reinterpret_cast<float *>(d + done)[threadIdx.x] = reinterpret_cast<const float *>(a + done)[threadIdx.x];
    } else {
        const long n4 = nb >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
            reinterpret_cast<float *>(d)[i] = reinterpret_cast<const float *>(a)[i];
    }
}
} // namespace votenet

extern "C" int votenet_copy_segments(int nseg, const votenet_copy_segment *seg, void *stream)
{
    VN_REQUIRE(nseg >= 0 && nseg <= kMaxCopy && (nseg == 0 || seg), "copy_segments expects 0..32 segments");
    if (nseg == 0) return VOTENET_OK;
    CopySegs S = {};
    S.nseg = nseg;
    long most = 0;
    for (int i = 0; i < nseg; i++) {
        VN_REQUIRE(seg[i].bytes >= 0 && seg[i].bytes % 4 == 0 && (seg[i].bytes == 0 || (seg[i].dst && seg[i].src)) &&
                       (uintptr_t)seg[i].dst % 4 == 0 && (uintptr_t)seg[i].src % 4 == 0,
                   "copy_segments: segment %d: null buffer, or bytes / addresses not multiples of 4", i);
        S.dst[i] = static_cast<char *>(seg[i].dst);
        S.src[i] = static_cast<const char *>(seg[i].src);
        S.bytes[i] = seg[i].bytes;
        most = seg[i].bytes > most ? seg[i].bytes : most;
    }
    long gx = (most / 16 + 256 * 4 - 1) / (256 * 4); // four 16-byte vectors per thread on the largest segment
    gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)gx, nseg), dim3(256), 0, as_stream(stream), S);
    return check_launch("copy_segments");
}

// BatchNorm moving averages of every layer in one launch: ema = momentum * ema + factor .* batch over the flat buffers that hold all
// layers' (scale | shift | mean | var) blocks (factor = 1 - momentum, times rows / (rows - 1) on the variance rows: the unbiased batch
// variance tf.nn.fused_batch_norm hands to the update).
namespace votenet {
__global__ __launch_bounds__(256) void ema_update_kernel(long n, float momentum, float *__restrict__ ema, const float *__restrict__ batch,
                                                         const float *__restrict__ factor)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) ema[i] = momentum * ema[i] + factor[i] * batch[i];
}
} // namespace votenet

extern "C" int votenet_ema_update(long n, float momentum, float *ema, const float *batch, const float *factor, void *stream)
{
    VN_REQUIRE(n >= 0 && (n == 0 || (ema && batch && factor)), "ema_update: bad arguments");
    if (n == 0) return VOTENET_OK;
    long grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(votenet::ema_update_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), n, momentum, ema, batch, factor);
    return check_launch("ema_update");
}
