/*
 * oracle_nms3d.cpp -- CPU restatement of tf_ops/3d_nms/tf_nms3d.cpp (3D IoU + greedy NMS).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY PARTIAL: tf_nms3d.cpp includes TensorFlow headers that this image lacks, so the
 * reference cannot be compiled here (and no stand-ins are written).  The restatement is
 * pinned by (i) closed-form geometry: the reference's own smoke input (tf_ops/3d_nms/tf_nms3d.py:21-46)
 * has BEV intersection 0.64 - 4 (0.4 sqrt 2 - 0.5)^2 = 0.6227418, hence IoU 0.49143 and keep lists
 * [[0,1],[0,0]] @ 0.5 / [[0,1]] @ 0.25; shifted unit cubes (1-t)/(1+t); the pi/4 octagon 2 (sqrt 2 - 1)
 * (tests/test_oracle_properties.py::test_iou_and_nms_closed_form_known_answers), and (ii) an independent
 * Sutherland-Hodgman clipping cross-check over random boxes (test_iou_against_independent_clipping,
 * test_nms_semantics).
 *
 * C++ (not C) on purpose: the reference's vertex ordering goes through std::sort and its
 * visit order through std::priority_queue; using the same libstdc++ containers keeps the
 * unspecified parts (unstable sort, heap order of equal scores) identical.
 */
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <deque>
#include <queue>
#include <vector>

namespace {

struct P2 {
    float x, z;
    P2(float xx, float zz) : x(xx), z(zz) {}
};

#define O_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define O_MAX(a, b) (((a) > (b)) ? (a) : (b))

/* tf_nms3d.cpp:43-46 */
inline float area2d(const float *bbox)
{
    return sqrtf((bbox[0 * 3] - bbox[1 * 3]) * (bbox[0 * 3] - bbox[1 * 3]) +
                 (bbox[0 * 3 + 2] - bbox[1 * 3 + 2]) * (bbox[0 * 3 + 2] - bbox[1 * 3 + 2])) *
           sqrtf((bbox[1 * 3] - bbox[2 * 3]) * (bbox[1 * 3] - bbox[2 * 3]) +
                 (bbox[1 * 3 + 2] - bbox[2 * 3 + 2]) * (bbox[1 * 3 + 2] - bbox[2 * 3 + 2]));
}

/* tf_nms3d.cpp:48-50 */
inline float area3d(const float *bbox) { return area2d(bbox) * (bbox[0 * 3 + 1] - bbox[4 * 3 + 1]); }

/* tf_nms3d.cpp:53-67: even-odd ray test against the first four corners (x,z) */
inline bool point_in_polygon(const P2 &p, const float *poly)
{
    bool result = false;
    for (int i = 0, j = 3; i < 4; j = i++) {
        if ((poly[i * 3 + 2] > p.z) != (poly[j * 3 + 2] > p.z) &&
            (p.x < (poly[j * 3] - poly[i * 3]) * (p.z - poly[i * 3 + 2]) / (poly[j * 3 + 2] - poly[i * 3 + 2]) +
                       poly[i * 3]))
            result = !result;
    }
    return result;
}

/* tf_nms3d.cpp:69-100: double math, |det|<1e-7 is "parallel", inclusive on-segment tests */
inline bool segment_intersection(const P2 &l1p1, const P2 &l1p2, const P2 &l2p1, const P2 &l2p2, P2 *out)
{
    double A1 = l1p2.z - l1p1.z;
    double B1 = l1p1.x - l1p2.x;
    double C1 = A1 * l1p1.x + B1 * l1p1.z;
    double A2 = l2p2.z - l2p1.z;
    double B2 = l2p1.x - l2p2.x;
    double C2 = A2 * l2p1.x + B2 * l2p1.z;
    double det = A1 * B2 - A2 * B1;
    if (std::abs(det) < 1e-7) return false;
    double x = (B2 * C1 - B1 * C2) / det;
    double z = (A1 * C2 - A2 * C1) / det;
    bool online1 = ((O_MIN(l1p1.x, l1p2.x) <= x) && (O_MAX(l1p1.x, l1p2.x) >= x) &&
                    (O_MIN(l1p1.z, l1p2.z) <= z) && (O_MAX(l1p1.z, l1p2.z) >= z));
    bool online2 = ((O_MIN(l2p1.x, l2p2.x) <= x) && (O_MAX(l2p1.x, l2p2.x) >= x) &&
                    (O_MIN(l2p1.z, l2p2.z) <= z) && (O_MAX(l2p1.z, l2p2.z) >= z));
    if (online1 && online2) {
        *out = P2((float)x, (float)z);
        return true;
    }
    return false;
}

/* tf_nms3d.cpp:122-175 */
float intersection(const float *bbox1, const float *bbox2)
{
    std::vector<P2> cc;
    for (int i = 0; i < 4; i++)
        if (point_in_polygon(P2(bbox1[i * 3], bbox1[i * 3 + 2]), bbox2)) cc.emplace_back(bbox1[i * 3], bbox1[i * 3 + 2]);
    for (int i = 0; i < 4; i++)
        if (point_in_polygon(P2(bbox2[i * 3], bbox2[i * 3 + 2]), bbox1)) cc.emplace_back(bbox2[i * 3], bbox2[i * 3 + 2]);
    for (int i = 0; i < 4; i++) {
        int next = (i + 1 == 4) ? 0 : i + 1;
        P2 a(bbox1[i * 3], bbox1[i * 3 + 2]), bq(bbox1[next * 3], bbox1[next * 3 + 2]);
        for (int e = 0; e < 4; e++) {
            int en = (e + 1 == 4) ? 0 : e + 1;
            P2 ip(0, 0);
            if (segment_intersection(a, bq, P2(bbox2[e * 3], bbox2[e * 3 + 2]), P2(bbox2[en * 3], bbox2[en * 3 + 2]), &ip))
                cc.emplace_back(ip.x, ip.z);
        }
    }
    float mx = 0, mz = 0;
    for (const auto &p : cc) {
        mx += p.x;
        mz += p.z;
    }
    mx /= cc.size();
    mz /= cc.size();
    std::sort(cc.begin(), cc.end(), [&](const P2 &p1, const P2 &p2) {
        return atan2f(p1.z - mz, p1.x - mx) < atan2f(p2.z - mz, p2.x - mx);
    });
    float area = 0;
    int i, j;
    for (i = 0, j = (int)cc.size() - 1; i < (int)cc.size(); j = i++)
        area += fabsf((mx * (cc[i].z - cc[j].z) + cc[i].x * (cc[j].z - mz) + cc[j].x * (mz - cc[i].z)) / 2);
    return area;
}

/* tf_nms3d.cpp:178-192 (iou2d at :186 is computed and unused there) */
float iou3d(const float *bi, const float *bj)
{
    float intersection2d = intersection(bi, bj);
    float intersection3d = O_MAX(O_MIN(bi[1], bj[1]) - O_MAX(bi[4 * 3 + 1], bj[4 * 3 + 1]), 0) * intersection2d;
    return intersection3d / (area3d(bi) + area3d(bj) - intersection3d);
}

} // namespace

extern "C" float oracle_bev_intersection(const float *bbox1, const float *bbox2) { return intersection(bbox1, bbox2); }
extern "C" float oracle_iou3d(const float *bbox1, const float *bbox2) { return iou3d(bbox1, bbox2); }

extern "C" void oracle_iou3d_matrix(int nboxes, const float *bboxes, float *iou)
{
    for (int i = 0; i < nboxes; i++)
        for (int j = 0; j < nboxes; j++) iou[(size_t)i * nboxes + j] = iou3d(bboxes + (size_t)i * 24, bboxes + (size_t)j * 24);
}

/*
 * DoNonMaxSuppressionOp, tf_nms3d.cpp:202-273: candidates = objectness[...,1] > [...,0]
 * (:230); one global max-heap over all scenes keyed on score (:222-234); a candidate is
 * dropped iff an already-selected box OF THE SAME SCENE has iou > threshold, scanning the
 * selected list backwards (:248-255); output rows [batch, box] in visit order (:266-272).
 */
extern "C" int oracle_nms3d(int b, int n, const float *bboxes, const float *scores,
                            const float *objectiveness, float iou_threshold, int *out)
{
    struct Candidate {
        int batch_index;
        int box_index;
        float score;
    };
    auto cmp = [](const Candidate bs_i, const Candidate bs_j) { return bs_i.score < bs_j.score; };
    std::priority_queue<Candidate, std::deque<Candidate>, decltype(cmp)> pq(cmp);
    for (int i = 0; i < b * n; ++i)
        if (objectiveness[i * 2 + 1] > objectiveness[i * 2]) pq.emplace(Candidate({i / n, i % n, scores[i]}));
    std::vector<int> selected;
    while (!pq.empty()) {
        Candidate c = pq.top();
        bool should_select = true;
        for (int j = (int)selected.size() - 2; j >= 0; j -= 2) {
            if (selected[j] == c.batch_index) {
                const float *bi = bboxes + ((size_t)c.batch_index * n + c.box_index) * 24;
                const float *bj = bboxes + ((size_t)c.batch_index * n + selected[j + 1]) * 24;
                if (iou3d(bi, bj) > iou_threshold) {
                    should_select = false;
                    break;
                }
            }
        }
        if (should_select) {
            selected.push_back(c.batch_index);
            selected.push_back(c.box_index);
        }
        pq.pop();
    }
    for (size_t i = 0; i < selected.size(); i++) out[i] = selected[i];
    return (int)(selected.size() / 2);
}
