"""Generate tests/golden/ref_gpu_*.npz from the REFERENCE's own device kernels (oracle/_ref/libref_{sampling,grouping}_gpu.so:
tf_sampling_g.cu / tf_grouping_g.cu compiled for gfx950 where they lie, oracle/Makefile).  Needs a GPU, so it runs on the GPU box:

    gpurun -- python tests/golden/make_ref_gpu_golden.py            # writes gpurun_out/ref_gpu_golden/*.npz + report.txt
    cp gpurun_out/ref_gpu_golden/*.npz tests/golden/                # in the build container

The inputs are the seeded generators of cases.py / votenet_amd/synth.py; the expected outputs are what the reference kernels
computed on MI355X (source="ref-gpu").  Large index tensors are stored whole when they are what a CPU test recomputes (FPS picks),
as sha256 digests otherwise.  The script also compares every output with the CPU oracle and writes the verdicts to report.txt --
the fixtures are written whatever the verdict is, so a disagreement shows up as a failing CPU test, not as a missing file.
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import ref_gpu as R  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert R.available(), "needs oracle/_ref/libref_*_gpu.so (built in the container) and a GPU"
    out_dir = os.path.join(ROOT, "gpurun_out", "ref_gpu_golden")
    os.makedirs(out_dir, exist_ok=True)
    report = []

    def verdict(name, ok, extra=""):
        report.append("%-44s %s %s" % (name, "oracle == ref-gpu" if ok else "ORACLE DIFFERS", extra))
        print(report[-1], flush=True)

    # ---- FPS: the small cases of cases.fps_cases() and the full-size clouds
    fps = {}
    for name, (xyz, m) in list(cases.fps_cases().items()) + list(cases.full_size_cases().items()):
        t0 = time.time()
        ref = R.farthest_point_sample(m, xyz)
        t1 = time.time()
        fps[name] = ref
        ora = O.farthest_point_sample(m, xyz)
        verdict("fps " + name, (ora == ref).all(), "(ref kernel %.1f ms incl. copies, %d picks differ)" % ((t1 - t0) * 1e3, int((ora != ref).sum())))
    np.savez_compressed(os.path.join(out_dir, "ref_gpu_fps.npz"), source="ref-gpu (tf_sampling_g.cu:105-170 on gfx950)", **fps)

    # ---- gather_point and its gradient (integer-valued cotangents: the atomic adds are exact in any order)
    xyz, m = cases.fps_cases()["small_n300"]
    idx = fps["small_n300"]
    g = R.gather_point(xyz, idx)
    verdict("gather_point small_n300", (g == O.gather_point(xyz, idx)).all())
    rng = np.random.default_rng(3)
    cot = rng.integers(-8, 9, size=g.shape).astype(np.float32)
    gg = R.gather_point_grad(xyz.shape[1], idx, cot)
    verdict("gather_point_grad small_n300", (gg == O.gather_point_grad(xyz, idx, cot)).all())
    np.savez_compressed(os.path.join(out_dir, "ref_gpu_gather.npz"), out=g, cot=cot, grad=gg, source="ref-gpu (tf_sampling_g.cu:172-192)")

    # ---- ProbSample / cumsum
    ps = {}
    for name, (p, r) in cases.prob_sample_cases().items():
        ps[name] = R.prob_sample(p, r)
        cs = R.cumsum(p)
        ps[name + "_cumsum_sha"] = sha(cs)
        verdict("prob_sample " + name, (ps[name] == O.prob_sample(p, r)).all() and (cs == O.cumsum(p)).all())
    np.savez_compressed(os.path.join(out_dir, "ref_gpu_prob_sample.npz"), source="ref-gpu (tf_sampling_g.cu:7-104)", **ps)

    # ---- ball query / group / group-grad on the device kernels: the reference's test shapes and the full-size sa1 level
    bq = {}
    for name, c in (("optest", cases.grouping_optest()), ("demo", cases.grouping_demo())):
        idx, cnt = R.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
        oi, oc = O.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
        grouped = R.group_point(c["points"], idx)
        verdict("ball query + group " + name, (idx == oi).all() and (cnt == oc).all() and (grouped == O.group_point(c["points"], oi)).all())
        bq[name + "_idx_sha"], bq[name + "_cnt"], bq[name + "_out_sha"] = sha(idx), cnt, sha(grouped)
    c = cases.grouping_optest()
    idx, _ = R.query_ball_point(c["radius"], c["nsample"], c["xyz1"], c["xyz2"])
    cot = np.random.default_rng(4).integers(-8, 9, size=(1, 8, 32, 16)).astype(np.float32)
    gg = R.group_point_grad(128, idx, cot)
    verdict("group_point_grad optest", (gg == O.group_point_grad(c["points"], idx, cot)).all())
    bq["optest_grad_cot"], bq["optest_grad"] = cot, gg
    room = cases.full_size_cases()["room_8x20480"][0]
    centres = O.gather_point(room, fps["room_8x20480"])
    for r, k in ((0.2, 64),):
        idx, cnt = R.query_ball_point(r, k, room, centres)
        oi, oc = O.query_ball_point(r, k, room, centres)
        verdict("ball query sa1 8x20480x2048 r=%.1f K=%d" % (r, k), (idx == oi).all() and (cnt == oc).all(), "(mean pts_cnt %.2f)" % cnt.mean())
        bq["sa1_idx_sha"], bq["sa1_cnt_sha"], bq["sa1_idx_head"] = sha(idx), sha(cnt), idx[0, :4]
    np.savez_compressed(os.path.join(out_dir, "ref_gpu_grouping.npz"), source="ref-gpu (tf_grouping_g.cu:3-78)", **bq)

    # ---- SelectionSort on the device kernel
    ss = {}
    for name, (dist, k) in cases.selection_sort_cases().items():
        outi, val = R.selection_sort(k, dist)
        oi, ov = O.select_top_k(k, dist)
        ok = (outi[..., :k] == oi[..., :k]).all() and (val[..., :k] == ov[..., :k]).all()
        verdict("selection_sort " + name, ok)
        ss[name + "_idx_sha"], ss[name + "_val_sha"] = sha(outi[..., :k]), sha(val[..., :k])
    np.savez_compressed(os.path.join(out_dir, "ref_gpu_selection_sort.npz"), source="ref-gpu (tf_grouping_g.cu:83-123)", **ss)

    with open(os.path.join(out_dir, "report.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    print("written to", out_dir)
    return 0 if all("DIFFERS" not in r for r in report) else 1


if __name__ == "__main__":
    sys.exit(main())
