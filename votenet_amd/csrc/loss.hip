// loss.hip -- the reference's loss graph (model.py:61-84, 141-231) as ONE kernel: vote targets (rotated-box membership,
// nearest ground-truth centre), proposal assignment (nearest centre, positive / negative by distance), objectness,
// centre + Chamfer ("dual") centre, heading, size and semantic losses, the total cost -- and the cotangents of that cost
// with respect to votes_xyz, proposals_xyz and proposals_output, which is what the backward pass of the hot path needs.
// The reference spends ~120 TensorFlow ops (gather_nd / where / one_hot / reduce_mean ...) on a few thousand elements; here
// one workgroup walks them (B*1024 seeds, B*256 proposals, <= 16 boxes per scene): the step is launch-latency, not work.
// Reductions are wave shuffles + a fixed-order combine: the loss values are reproducible bit for bit.
#include "common.h"

namespace votenet {

#ifndef LOSS_ABL
#define LOSS_ABL 0 // probe builds (tools/probe/loss_ablate.sh): 1 no proposal terms, 2 no seed terms, 4 no dual term -- wrong results, time only
#endif
constexpr int LOSS_T = 1024;
constexpr int LOSS_MAXC = 32;  // nh, ns, nc <= 32
constexpr int LOSS_NACC = 12;

struct LossArgs {
    int b, n, p, bb, nh, ns, nc;
    const float *seeds, *votes, *pxyz, *pout;
    const float *gxyz, *glwh, *groty;
    const int *sem, *hlab, *slab;
    const float *hres, *sres;
    float pos_thr, neg_thr;
    float *losses, *d_votes, *d_pxyz, *d_pout;
    long pout_pitch; // floats between the rows of proposals_output (>= its width: a column slice of a wider tensor is read in place)
};

__device__ __forceinline__ float huber(float e, float &grad) // tf.losses.huber_loss, delta = 1: e = prediction - label
{
    const float a = fabsf(e);
    grad = a <= 1.0f ? e : (e > 0.0f ? 1.0f : -1.0f);
    return a <= 1.0f ? 0.5f * e * e : a - 0.5f;
}

// softmax cross entropy of `c` logits (stride 1) against `label`; probs[] receives softmax - onehot
__device__ __forceinline__ float softmax_ce(const float *lg, int c, int label, float *probs)
{
    float m = lg[0];
    for (int i = 1; i < c; i++) m = fmaxf(m, lg[i]);
    float s = 0.0f;
    for (int i = 0; i < c; i++) {
        probs[i] = expf(lg[i] - m);
        s += probs[i];
    }
    const float inv = 1.0f / s;
    for (int i = 0; i < c; i++) probs[i] = probs[i] * inv - (i == label ? 1.0f : 0.0f);
    return logf(s) + m - lg[label];
}

// nearest ground-truth centre of proposal (px,py,pz) among the scene's boxes staged in LDS: -> distance, box index
__device__ __forceinline__ float nearest_box(const float (*s_box)[8], int BB, float px, float py, float pz, int &g)
{
    float best = 0.0f;
    g = 0;
    for (int j = 0; j < BB; j++) {
        const float dx = px - s_box[j][0], dy = py - s_box[j][1], dz = pz - s_box[j][2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        if (j == 0 || d < best) { // tf.argmin: first minimum
            best = d;
            g = j;
        }
    }
    return best;
}

constexpr int LOSS_MAXBOX = 256; // boxes per scene

// pass 1: one workgroup per scene counts its positive / negative proposals (model.py:147-153); the means of the loss are
// over the counts of the WHOLE batch, so they have to exist before any cotangent can be written.
__global__ __launch_bounds__(256) void votenet_loss_count_kernel(LossArgs A, int *counts)
{
    __shared__ float s_box[LOSS_MAXBOX][8];
    const int b = blockIdx.x, tid = threadIdx.x, BB = A.bb, P = A.p;
    for (int j = tid; j < BB; j += 256)
        for (int k = 0; k < 3; k++) s_box[j][k] = A.gxyz[(b * BB + j) * 3 + k];
    __syncthreads();
    int np_local = 0, nn_local = 0;
    for (int pq = tid; pq < P; pq += 256) {
        const int q = b * P + pq;
        int g;
        const float best = nearest_box(s_box, BB, A.pxyz[q * 3 + 0], A.pxyz[q * 3 + 1], A.pxyz[q * 3 + 2], g);
        np_local += best < A.pos_thr;
        nn_local += best > A.neg_thr;
    }
    for (int off = 32; off > 0; off >>= 1) {
        np_local += __shfl_down(np_local, off);
        nn_local += __shfl_down(nn_local, off);
    }
    if ((tid & 63) == 0) {
        if (np_local) atomicAdd(&counts[0], np_local);
        if (nn_local) atomicAdd(&counts[1], nn_local);
    }
}

// pass 2: one workgroup per scene: every loss term and cotangent of its proposals, boxes and seeds; the per-scene partial
// sums are combined by the last workgroup to finish, in scene order (bit-reproducible).
__global__ __launch_bounds__(LOSS_T) void votenet_loss_kernel(LossArgs A, int *counts, float *partial /* b x LOSS_NACC */)
{
    __shared__ float s_box[LOSS_MAXBOX][8];
    __shared__ float s_red[LOSS_T / 64][LOSS_NACC];
    __shared__ float s_dual[LOSS_MAXBOX][3];
    __shared__ int s_dualp[LOSS_MAXBOX];
    __shared__ int s_last;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int B = A.b, N = A.n, P = A.p, BB = A.bb, NH = A.nh, NS = A.ns, NC = A.nc;
    const int W = 5 + 2 * NH + 4 * NS + NC; // width of proposals_output (79)
    // per-box constants once, in LDS: centre, half extents, cos / sin of -roty (the loops below touch them (N+P)*BB times)
    for (int j = tid; j < BB; j += LOSS_T) {
        const int e = b * BB + j;
        s_box[j][0] = A.gxyz[e * 3 + 0];
        s_box[j][1] = A.gxyz[e * 3 + 1];
        s_box[j][2] = A.gxyz[e * 3 + 2];
        s_box[j][3] = A.glwh[e * 3 + 0] * 0.5f;
        s_box[j][4] = A.glwh[e * 3 + 1] * 0.5f;
        s_box[j][5] = A.glwh[e * 3 + 2] * 0.5f;
        s_box[j][6] = cosf(-A.groty[e]);
        s_box[j][7] = sinf(-A.groty[e]);
    }
    __syncthreads();
    const int n_pos = counts[0], n_neg = counts[1];
    const float inv_np = 1.0f / (float)n_pos, inv_nn = 1.0f / (float)n_neg; // empty set -> inf -> NaN loss, as reduce_mean of []
    // accumulators: 0 vote, 1 obj_pos, 2 obj_neg, 3 center, 4 center_dual, 5 hcls, 6 hres, 7 scls, 8 sres, 9 sem
    float acc[LOSS_NACC];
#pragma unroll
    for (int i = 0; i < LOSS_NACC; i++) acc[i] = 0.0f;
    // Roles.  With P proposals on LOSS_T threads only the first ceil(P / 64) waves own a proposal (P = 256: four of sixteen), and a
    // proposal is the longest chain of the kernel (79 strided floats read, ~70 written, three softmaxes).  The other waves take the dual
    // term (one wave per box) and ALL the seeds meanwhile instead of idling at the barrier and doing both afterwards
    // (tools/probe/loss_ablate.sh: proposals 14.6 us, dual 5.2, seeds 4.3 of the launch).  P >= LOSS_T: every wave owns proposals, the
    // phases run one after the other as before.
    const int wave = tid >> 6, nwave = LOSS_T / 64;
    const int nbusy = (P + 63) / 64 < nwave ? (P + 63) / 64 : nwave, nfree = nwave - nbusy;
    const bool early = nfree > 0;
    // ---- seeds: vote targets and vote regression loss (model.py:61-84), as a loop over this thread's seeds first, first + step, ...
    const float inv_bn = 1.0f / (float)(B * N);
    auto seed_terms = [&](int first, int step) {
        for (int i = first; i < ((LOSS_ABL & 2) ? 0 : N); i += step) {
            const int e = b * N + i;
            const float sx = A.seeds[e * 3 + 0], sy = A.seeds[e * 3 + 1], sz = A.seeds[e * 3 + 2];
            float best = 0.0f;
            int g = 0;
            bool surface = false;
            for (int j = 0; j < BB; j++) {
                // |seed - centre| first, THEN the rotation by -roty (the reference's order, model.py:61,74)
                const float dx = fabsf(sx - s_box[j][0]), dy = fabsf(sy - s_box[j][1]), dz = fabsf(sz - s_box[j][2]);
                const float c = s_box[j][6], s = s_box[j][7];
                const float rx = c * dx + s * dz, ry = dy, rz = -s * dx + c * dz;
                surface = surface || (rx < s_box[j][3] && ry < s_box[j][4] && rz < s_box[j][5]);
                const float d = sqrtf(rx * rx + ry * ry + rz * rz);
                if (j == 0 || d < best) {
                    best = d;
                    g = j;
                }
            }
            if (surface) {
                for (int k = 0; k < 3; k++) {
                    const float df = A.votes[e * 3 + k] - s_box[g][k];
                    acc[0] += fabsf(df);
                    A.d_votes[e * 3 + k] = (df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f)) * inv_bn;
                }
            }
        }
    };
    float pr[LOSS_MAXC];
    // ---- proposals: losses and cotangents (model.py:156-212); every weight of model.py:205,228 folded in
    for (int pq = tid; pq < ((LOSS_ABL & 1) ? 0 : P); pq += LOSS_T) {
        const int q = b * P + pq;
        const float px = A.pxyz[q * 3 + 0], py = A.pxyz[q * 3 + 1], pz = A.pxyz[q * 3 + 2];
        int g;
        const float best = nearest_box(s_box, BB, px, py, pz, g);
        const float *o = A.pout + (size_t)q * A.pout_pitch;
        float *go = A.d_pout + (size_t)q * W;
        const bool pos = best < A.pos_thr, neg = best > A.neg_thr;
        if (pos || neg) { // objectness, weight 0.5 (a proposal can only be one of the two: pos_thr < neg_thr)
            const float l = softmax_ce(o, 2, pos ? 1 : 0, pr);
            const float w = 0.5f * (pos ? inv_np : inv_nn);
            acc[pos ? 1 : 2] += l;
            go[0] = pr[0] * w;
            go[1] = pr[1] * w;
        }
        if (pos) {
            const int gi = b * BB + g;
            // centre (weight 1): error = prediction - (gt centre - proposal centre)
            const float cg[3] = {s_box[g][0] - px, s_box[g][1] - py, s_box[g][2] - pz};
            for (int k = 0; k < 3; k++) {
                float gr;
                acc[3] += huber(o[2 + k] - cg[k], gr);
                go[2 + k] += gr * inv_np;                  // this thread owns the proposal here; the dual term adds below, behind a barrier
                A.d_pxyz[q * 3 + k] += gr * inv_np;        // d(error)/d(proposal centre) = +1
            }
            // heading class (0.1) and residual (1)
            const int hl = A.hlab[gi];
            acc[5] += softmax_ce(o + 5, NH, hl, pr);
            for (int i = 0; i < NH; i++) go[5 + i] = pr[i] * (0.1f * inv_np);
            {
                float gr;
                acc[6] += huber(o[5 + NH + hl] - A.hres[gi], gr);
                go[5 + NH + hl] = gr * inv_np;
            }
            // size class (0.1) and residual (1)
            const int sl = A.slab[gi], so = 5 + 2 * NH;
            acc[7] += softmax_ce(o + so, NS, sl, pr);
            for (int i = 0; i < NS; i++) go[so + i] = pr[i] * (0.1f * inv_np);
            for (int k = 0; k < 3; k++) {
                float gr;
                acc[8] += huber(o[so + NS + sl * 3 + k] - A.sres[gi * 3 + k], gr);
                go[so + NS + sl * 3 + k] = gr * inv_np;
            }
            // semantic class (0.1)
            const int co = W - NC;
            acc[9] += softmax_ce(o + co, NC, A.sem[gi], pr);
            for (int i = 0; i < NC; i++) go[co + i] = pr[i] * (0.1f * inv_np);
        }
    }
    // ---- Chamfer / dual centre term (model.py:172-177): every ground-truth box pulls its nearest proposal
    const float inv_bbb = 1.0f / (float)(B * BB);
    const int dw = early ? wave - nbusy : wave, dstep = early ? nfree : nwave; // (a wave that owns proposals: dw < 0 when the others take the boxes)
    for (int j = dw < 0 ? BB : dw; j < ((LOSS_ABL & 4) ? 0 : BB); j += dstep) { // one wave per box: lanes scan the proposals, wave arg-min
        const int lane = tid & 63;
        const float gx = s_box[j][0], gy = s_box[j][1], gz = s_box[j][2];
        float best = INFINITY;
        int bp = 0x7FFFFFFF;
        for (int p = lane; p < P; p += 64) {
            const float dx = A.pxyz[(b * P + p) * 3 + 0] - gx, dy = A.pxyz[(b * P + p) * 3 + 1] - gy,
                        dz = A.pxyz[(b * P + p) * 3 + 2] - gz;
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            if (d < best) { // ascending p inside a lane: first minimum of the lane
                best = d;
                bp = p;
            }
        }
        for (int off = 32; off > 0; off >>= 1) { // smallest distance, then smallest index: tf.argmin's first minimum
            const float ob = __shfl_xor(best, off);
            const int op = __shfl_xor(bp, off);
            if (ob < best || (ob == best && op < bp)) {
                best = ob;
                bp = op;
            }
        }
        if (lane < 3) {
            const int k = lane, q = b * P + bp;
            const float cgk = (k == 0 ? gx : (k == 1 ? gy : gz)) - A.pxyz[q * 3 + k];
            float gr;
            acc[4] += huber(A.pout[(size_t)q * A.pout_pitch + 2 + k] - cgk, gr);
            s_dual[j][k] = gr * inv_bbb; // added to the proposal's cotangents below, box by box in ascending order
            if (k == 0) s_dualp[j] = bp;
        }
    }
    if (early && wave >= nbusy) seed_terms(tid - nbusy * 64, nfree * 64);
    __syncthreads(); // the proposals' own terms (above) and every box's pull are in place
    // several boxes may pull the same proposal: one thread per proposal adds them in box order (no atomics: one summation
    // order, bit-reproducible cotangents)
    for (int pq = tid; pq < P; pq += LOSS_T) {
        const int q = b * P + pq;
        for (int j = 0; j < BB; j++)
            if (s_dualp[j] == pq) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    A.d_pout[(size_t)q * W + 2 + k] += s_dual[j][k];
                    A.d_pxyz[q * 3 + k] += s_dual[j][k];
                }
            }
    }
    if (!early) seed_terms(tid, LOSS_T);
    // ---- fixed-order reduction: lanes -> waves -> scene partial -> (last workgroup) scenes in order
#pragma unroll
    for (int i = 0; i < LOSS_NACC; i++) {
        float v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((tid & 63) == 0) s_red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < LOSS_NACC) {
        float v = 0.0f;
        for (int w = 0; w < LOSS_T / 64; w++) v += s_red[w][tid];
        partial[b * LOSS_NACC + tid] = v;
    }
    __threadfence(); // the partial sums are visible device-wide before the ticket is taken
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(&counts[2], 1) == B - 1);
    __syncthreads();
    if (!s_last) return; // (uniform over the workgroup)
    __threadfence();
    if (tid < LOSS_NACC) { // one thread per accumulator walks the scenes in order: twelve chains of B loads side by side instead of one of 12 B
        float v = 0.0f;
        for (int sc = 0; sc < B; sc++) v += __builtin_nontemporal_load(&partial[sc * LOSS_NACC + tid]);
        s_red[0][tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float t[LOSS_NACC];
        for (int i = 0; i < LOSS_NACC; i++) t[i] = s_red[0][i];
        const float vote = t[0] * inv_bn;
        const float obj = t[1] * inv_np + t[2] * inv_nn;
        const float center = t[3] * inv_np + t[4] * inv_bbb;
        const float hcls = t[5] * inv_np, hres = t[6] * inv_np, scls = t[7] * inv_np, sres = t[8] * inv_np, sem = t[9] * inv_np;
        const float box = center + 0.1f * hcls + hres + 0.1f * scls + sres; // model.py:205
        float *L = A.losses;
        L[0] = vote + 0.5f * obj + box + 0.1f * sem;                        // model.py:228
        L[1] = vote;
        L[2] = obj;
        L[3] = center;
        L[4] = hcls;
        L[5] = hres;
        L[6] = scls;
        L[7] = sres;
        L[8] = sem;
        L[9] = box;
        L[10] = (float)n_pos;
        L[11] = (float)n_neg;
    }
}

} // namespace votenet

using namespace votenet;

extern "C" size_t votenet_loss_workspace_floats(int b) { return 4 + (size_t)(b > 0 ? b : 0) * LOSS_NACC; }

extern "C" int votenet_loss_pitched(int b, int n_seeds, int n_prop, int n_box, int nh, int ns, int nc, const float *seeds_xyz,
                            const float *votes_xyz, const float *proposals_xyz, const float *proposals_output, long output_pitch,
                            const float *bboxes_xyz,
                            const float *bboxes_lwh, const float *bboxes_roty, const int *semantic_labels, const int *heading_labels,
                            const float *heading_residuals, const int *size_labels, const float *size_residuals, float pos_thr,
                            float neg_thr, float *losses, float *d_votes_xyz, float *d_proposals_xyz, float *d_proposals_output,
                            float *workspace, void *stream)
{
    VN_REQUIRE(b > 0 && n_seeds > 0 && n_prop > 0 && n_box > 0, "votenet_loss expects b, n_seeds, n_prop, n_box > 0");
    VN_REQUIRE(n_box <= LOSS_MAXBOX, "votenet_loss expects at most 256 boxes per scene");
    VN_REQUIRE(nh > 0 && ns > 0 && nc > 0 && nh <= LOSS_MAXC && ns <= LOSS_MAXC && nc <= LOSS_MAXC, "votenet_loss expects 0 < nh, ns, nc <= 32");
    VN_REQUIRE(pos_thr < neg_thr, "votenet_loss expects pos_thr < neg_thr (config.py)");
    VN_REQUIRE(seeds_xyz && votes_xyz && proposals_xyz && proposals_output && bboxes_xyz && bboxes_lwh && bboxes_roty &&
                   semantic_labels && heading_labels && heading_residuals && size_labels && size_residuals && losses &&
                   d_votes_xyz && d_proposals_xyz && d_proposals_output && workspace,
               "votenet_loss: null buffer");
    VN_REQUIRE(output_pitch >= 5 + 2 * nh + 4 * ns + nc, "votenet_loss: the pitch of proposals_output is smaller than its width");
    LossArgs a = {b, n_seeds, n_prop, n_box, nh, ns, nc, seeds_xyz, votes_xyz, proposals_xyz, proposals_output, bboxes_xyz, bboxes_lwh,
                  bboxes_roty, semantic_labels, heading_labels, size_labels, heading_residuals, size_residuals, pos_thr, neg_thr, losses,
                  d_votes_xyz, d_proposals_xyz, d_proposals_output, output_pitch};
    int *counts = reinterpret_cast<int *>(workspace); // [positives, negatives, finished workgroups, pad], zero on entry
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(votenet_loss_count_kernel, dim3(b), dim3(256), 0, st, a, counts);
    hipLaunchKernelGGL(votenet_loss_kernel, dim3(b), dim3(LOSS_T), 0, st, a, counts, workspace + 4);
    return check_launch("votenet_loss");
}

extern "C" int votenet_loss(int b, int n_seeds, int n_prop, int n_box, int nh, int ns, int nc, const float *seeds_xyz,
                            const float *votes_xyz, const float *proposals_xyz, const float *proposals_output, const float *bboxes_xyz,
                            const float *bboxes_lwh, const float *bboxes_roty, const int *semantic_labels, const int *heading_labels,
                            const float *heading_residuals, const int *size_labels, const float *size_residuals, float pos_thr,
                            float neg_thr, float *losses, float *d_votes_xyz, float *d_proposals_xyz, float *d_proposals_output,
                            float *workspace, void *stream)
{
    return votenet_loss_pitched(b, n_seeds, n_prop, n_box, nh, ns, nc, seeds_xyz, votes_xyz, proposals_xyz, proposals_output,
                                5 + 2 * nh + 4 * ns + nc, bboxes_xyz, bboxes_lwh, bboxes_roty, semantic_labels, heading_labels,
                                heading_residuals, size_labels, size_residuals, pos_thr, neg_thr, losses, d_votes_xyz, d_proposals_xyz,
                                d_proposals_output, workspace, stream);
}

// ---------------------------------------------------------------- box decode of the predict tower (model.py:100-129)
namespace votenet {

// One thread per proposal: size class arg-max -> class mean size * max(1 + residual, 1e-6); centre = proposal + offset;
// heading bin arg-max + residual -> angle = floormod((2*bin + residual) * pi/NH, 2*pi); the 8 corners of get_3d_bbox
// (model.py:100-112: x = +-l/2, y = +-h/2 (first four = top face), z = +-w/2, rotated about y) and the NMS score
// (max class logit) -- what NonMaxSuppression3D consumes, without the reference's one_hot / gather_nd / einsum nodes.
__global__ void decode_boxes_kernel(int total, int nh, int ns, int nc, const float *__restrict__ pxyz, const float *__restrict__ pout,
                                    const float *__restrict__ mean_size, float *__restrict__ boxes, float *__restrict__ scores)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int W = 5 + 2 * nh + 4 * ns + nc;
    const float *o = pout + (size_t)q * W;
    int sc = 0;
    for (int i = 1; i < ns; i++)
        if (o[5 + 2 * nh + i] > o[5 + 2 * nh + sc]) sc = i; // tf.argmax: first maximum
    int hc = 0;
    for (int i = 1; i < nh; i++)
        if (o[5 + i] > o[5 + hc]) hc = i;
    float size[3];
    for (int k = 0; k < 3; k++) size[k] = mean_size[sc * 3 + k] * fmaxf(1.0f + o[5 + 2 * nh + ns + sc * 3 + k], 1e-6f);
    const float cx = pxyz[q * 3 + 0] + o[2], cy = pxyz[q * 3 + 1] + o[3], cz = pxyz[q * 3 + 2] + o[4];
    const float PI_F = 3.14159265358979323846f;
    const float t = ((float)hc * 2.0f + o[5 + nh + hc]) * (PI_F / (float)nh);
    float ang = fmodf(t, 2.0f * PI_F); // tf.floormod: the result takes the sign of the divisor
    if (ang < 0.0f) ang += 2.0f * PI_F;
    const float c = cosf(ang), s = sinf(ang);
    const float l = size[0], w = size[1], h = size[2]; // lwh = (x, z, y) extents
    const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {1, 1, 1, 1, -1, -1, -1, -1}, sz[8] = {1, -1, -1, 1, 1, -1, -1, 1};
    float *bx = boxes + (size_t)q * 24;
    for (int m = 0; m < 8; m++) {
        const float x0 = sx[m] * l * 0.5f, y0 = sy[m] * h * 0.5f, z0 = sz[m] * w * 0.5f;
        bx[m * 3 + 0] = c * x0 + s * z0 + cx;
        bx[m * 3 + 1] = y0 + cy;
        bx[m * 3 + 2] = -s * x0 + c * z0 + cz;
    }
    float best = o[W - nc];
    for (int i = 1; i < nc; i++) best = fmaxf(best, o[W - nc + i]);
    scores[q] = best;
}

} // namespace votenet

extern "C" int votenet_decode_boxes(int b, int n_prop, int nh, int ns, int nc, const float *proposals_xyz, const float *proposals_output,
                                    const float *class_mean_size, float *bboxes, float *scores, void *stream)
{
    VN_REQUIRE(b >= 0 && n_prop >= 0 && nh > 0 && ns > 0 && nc > 0, "decode_boxes: bad shape");
    const int total = b * n_prop;
    if (total == 0) return VOTENET_OK;
    VN_REQUIRE(proposals_xyz && proposals_output && class_mean_size && bboxes && scores, "decode_boxes: null buffer");
    hipLaunchKernelGGL(votenet::decode_boxes_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), total, nh, ns, nc,
                       proposals_xyz, proposals_output, class_mean_size, bboxes, scores);
    return check_launch("decode_boxes");
}
