"""Compose the judged summaries under profiles/ from a tools/collect_profiles.sh run:  python tools/make_profile_summaries.py r02a r02
copies the kernel-trace tables and bench lines and writes rNN_pmc_fps.txt / rNN_pmc_mlp.txt / pmc_latest.json from the PMC tables
(FETCH_SIZE doubled: gfx950 counts 128-byte requests at 64 B, MI355X_MICROARCH.md)."""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = os.path.join(R, "gpurun_out", sys.argv[1]), sys.argv[2]
P = os.path.join(R, "profiles")


def table(name):
    rows = {}
    for ln in open(os.path.join(src, name)):
        if ln.startswith("#") or ln.startswith("kernel") or not ln.strip():
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+(.*)$", ln.rstrip())
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), [float(v) for v in m.group(4).split()])
    return rows


for f in ("train_kernel_stats.txt", "fwd_kernel_stats.txt", "train_bench_line.json", "fwd_bench_line.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(P, "%s_%s" % (tag, f)))
# ---- FPS / spatial index / ball query
fe, wr = table("fps_fetch.txt"), table("fps_write.txt")
B, n, m, K = 8, 20480, 2048, 64
alg_f, alg_b = B * (m - 1) * n * 16 + B * n * 12 + B * m * 4, B * m * n * 12 + B * m * (K + 1) * 4
lines = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_fps.py: sa1 geometry, 8 x 20480 -> 2048, room scenes",
         "# KB per dispatch, mean of 5; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as reported",
         "%-34s %9s %14s %12s" % ("kernel", "avg_us", "FETCH_KB(x2)", "WRITE_KB")]
tot_f = tot_w = 0.0
grp = {"fps": [0.0, 0.0, 0.0], "bq": [0.0, 0.0, 0.0]}
for k, (c, us, v) in fe.items():
    w = wr.get(k, (0, 0, [0.0]))[2][0]
    lines.append("%-34s %9.2f %14.1f %12.1f" % (k[:34], us, 2 * v[0], w))
    g = "bq" if "ball_query" in k else ("fps" if ("fps_" in k or "sidx_" in k) else None)
    if g:
        grp[g][0] += us
        grp[g][1] += 2 * v[0] * 1024
        grp[g][2] += w * 1024
fb = grp["fps"]
lines += ["# sa1 FPS launch = spatial index (5 kernels) + sampling kernel: %.1f us of kernels, HBM read %d B (corrected), written %d B, total %d B"
          % (fb[0], fb[1], fb[2], fb[1] + fb[2]),
          "#   algorithmic bytes of the reference access pattern (SURVEY 8d): %d -> traffic / algorithmic = %.4f: the kernel is register / LDS resident"
          % (alg_f, (fb[1] + fb[2]) / alg_f),
          "# sa1 ball query over the index: %.1f us, HBM read %d B, written %d B; algorithmic (all pairs) %d -> %.4f"
          % (grp["bq"][0], grp["bq"][1], grp["bq"][2], alg_b, (grp["bq"][1] + grp["bq"][2]) / alg_b)]
open(os.path.join(P, "%s_pmc_fps.txt" % tag), "w").write("\n".join(lines) + "\n")
_pj_path = os.path.join(P, "pmc_latest.json")
_pj = json.load(open(_pj_path)) if os.path.exists(_pj_path) else {}  # other blocks (mlp_families: tools/pmc_mlp_bwd_summary.py) stay
_pj.update({"fps_sa1": {"hbm_bytes_per_launch": int(fb[1] + fb[2]), "fetch_bytes_corrected": int(fb[1]), "write_bytes": int(fb[2]),
                        "source": "profiles/%s_pmc_fps.txt" % tag},
            "ball_query_sa1": {"hbm_bytes_per_launch": int(grp["bq"][1] + grp["bq"][2]), "source": "profiles/%s_pmc_fps.txt" % tag},
            "source": "profiles/%s_pmc_fps.txt" % tag})
json.dump(_pj, open(_pj_path, "w"), indent=1)
# ---- MLP (skipped when the run collected the sampling counters only: tools/pmc_fps.sh)
if not os.path.exists(os.path.join(src, "mlp_sq.txt")):
    print(open(os.path.join(P, "%s_pmc_fps.txt" % tag)).read())
    sys.exit(0)
sq, mf, mw = table("mlp_sq.txt"), table("mlp_fetch.txt"), table("mlp_write.txt")
hdr = open(os.path.join(src, "mlp_sq.txt")).read().splitlines()[1].split()
ci = {name: i for i, name in enumerate(hdr[3:])}
lines = ["# rocprofv3 --pmc on tools/pmc_mlp.py (separate passes: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES | FETCH_SIZE | WRITE_SIZE)",
         "# two layers, 17.18 GFLOP each: sa1 L2 = 1048576 x 64 -> 128 (HBM-bound), sa2 L2 = 262144 x 128 -> 256 (MFMA-bound)",
         "# SQ/GRBM counters per XCD (means).  SQ_VALU_MFMA_BUSY_CYCLES in quad-cycles: MfmaUtil = 4 * MFMA_BUSY / (128 SIMDs per XCD * GRBM_GUI_ACTIVE);",
         "# clock = GRBM_GUI_ACTIVE / duration.  FETCH_SIZE doubled (gfx950), KB -> MB.",
         "# Last template argument true = BF3: the products on six v_mfma_f32_32x32x16_bf16 of exactly split operands (32 busy cycles each)",
         "# instead of eight v_mfma_f32_32x32x2_f32 (64 each) per 16-deep slab -- MfmaUtil is the matrix pipe's busy share either way,",
         "# TFLOP/s the fp32 multiply-adds of the GEMM per second, of_157.3 that against the fp32 MFMA peak (what an fp32-MFMA kernel",
         "# could reach at most is 1.0).  The first pass of tools/pmc_mlp.py runs the same launches on the fp32 kernels (false).",
         "%-52s %8s %9s %9s %8s %9s %10s %10s" % ("kernel", "avg_us", "MfmaUtil", "GHz", "TFLOP/s", "of_157.3", "HBM_rd_MB", "HBM_wr_MB")]
for k, (c, us, v) in sq.items():
    if "mlp_" not in k:
        continue
    gui, mfma = v[ci["GRBM_GUI_ACTIVE"]], v[ci["SQ_VALU_MFMA_BUSY_CYCLES"]]
    tf = 17.18e9 / (us * 1e-6) / 1e12
    rd = 2 * mf.get(k, (0, 0, [0.0]))[2][0] / 1024
    w = mw.get(k, (0, 0, [0.0]))[2][0] / 1024
    lines.append("%-52s %8.1f %9.3f %9.2f %8.1f %9.3f %10.1f %10.1f" % (k[:52], us, 4 * mfma / (128 * gui), gui / us / 1e3, tf, tf / 157.3, rd, w))
open(os.path.join(P, "%s_pmc_mlp.txt" % tag), "w").write("\n".join(lines) + "\n")
print(open(os.path.join(P, "%s_pmc_fps.txt" % tag)).read())
print(open(os.path.join(P, "%s_pmc_mlp.txt" % tag)).read())
