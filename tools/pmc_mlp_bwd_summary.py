"""Compose profiles/rNN_pmc_mlp_bwd.txt (and the mlp_families block of profiles/pmc_latest.json that bench.py quotes) from a
tools/pmc_mlp_bwd.sh run:   python tools/pmc_mlp_bwd_summary.py r03
Counters are per-XCD samples (rocprofv3 --pmc, one pass per counter group); MfmaUtil = 4 * SQ_VALU_MFMA_BUSY_CYCLES / (128 SIMDs
per XCD * GRBM_GUI_ACTIVE) as in profiles/r02_pmc_mlp.txt; FETCH_SIZE doubled (gfx950 counts 128-byte requests at 64 B)."""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = os.path.join(R, "gpurun_out", "pmc_bwd"), sys.argv[1]
P = os.path.join(R, "profiles")


def table(name):
    rows, hdr = {}, None
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return rows, []
    for ln in open(path):
        if ln.startswith("#") or not ln.strip():
            continue
        if ln.startswith("kernel"):
            hdr = ln.split()[3:]
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+(.*)$", ln.rstrip())
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), [float(v) if v != "-" else 0.0 for v in m.group(4).split()])
    return rows, hdr or []


FAM = [("mlp_linear_fast_kernel<2, 2, 2, 2, 1, 6, true>", "dgrad_bn_reduce assembled", 524288, 128, 128),
       ("mlp_wgrad_fast_kernel<3, 2, 2, 1, true>", "wgrad_bn assembled", 524288, 128, 128),
       ("mlp_wgrad_fast_kernel<3, 2, 2, 1>", "wgrad_bn assembled (fp32 MFMA)", 524288, 128, 128),
       ("gram_bf3_kernel<128", "gram", 524288, 128, 128),
       ("mlp_linear_fast_kernel<2, 2, 2, 2, 0, 1, true>", "fwd+bn (the dense part of the Gram-form dgrad)", 524288, 128, 128),
       ("mlp_linear_fast_kernel<2, 2, 2, 2, 1, 1, true>", "dgrad_bn", 524288, 128, 128),
       ("mlp_linear_fast_kernel<2, 2, 2, 2, 0, 2, true>", "fwd+pool", 524288, 128, 256),
       ("mlp_linear_fast_kernel<2, 2, 2, 2, 4, 0, true>", "fwd+bn assembled", 524288, 128, 128),
       ("mlp_linear_fast_kernel<4, 1, 1, 2, 1, 4, true>", "dgrad_bn_reduce narrow", 1048576, 64, 64),
       ("mlp_wgrad_fast_kernel<2, 1, 1, 1, true>", "wgrad_bn narrow", 1048576, 64, 64),
       ("mlp_wgrad_fast_kernel<2, 1, 1, 1>", "wgrad_bn narrow (fp32 MFMA)", 1048576, 64, 64),
       ("gram_bf3_kernel<64", "gram (64)", 1048576, 64, 64)]
a, ha = table("a.txt")
b, hb = table("b.txt")
c, hc = table("c.txt")
d, _ = table("d.txt")
e, _ = table("e.txt")
ia, ib, ic = {n: i for i, n in enumerate(ha)}, {n: i for i, n in enumerate(hb)}, {n: i for i, n in enumerate(hc)}
lines = ["# rocprofv3 --pmc over tools/pmc_mlp_bwd.py alone (tools/pmc_mlp_bwd.sh: five passes -- SQ cycles | SQ instruction mix + MFMA busy | instruction counts + LDS |",
         "# FETCH_SIZE | WRITE_SIZE), the backward GEMM families at sa2's shape (524 288 x 128 x 128, real sa2 geometry of room scenes) and sa1's",
         "# (1 048 576 x 64 x 64), each kernel alone on the GPU.  Counters are per-XCD samples; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles,",
         "# SQ_VALU_MFMA_BUSY_CYCLES in cycles.  MfmaUtil = 4 * MFMA_BUSY / (128 SIMDs * GRBM_GUI_ACTIVE) (the formula of r02_pmc_mlp.txt); wait_any / wait_inst /",
         "# active = share of SQ_WAVE_CYCLES (parked at s_waitcnt or a barrier | issue stalls: MFMA dependencies, full pipes | issuing); valu/mfma = vector",
         "# instructions per MFMA; HBM MB per launch (FETCH_SIZE x 2: gfx950 counts 128-byte requests at 64 B; WRITE_SIZE as reported); TF/s = 2 rows cin cout",
         "# (fp32 multiply-adds of the GEMM) / the traced duration; of_157.3 against the fp32 MFMA peak, of_419.5 against six bf16 MFMAs per product (2516.8 / 6).",
         "%-52s %8s %8s %6s %8s %9s %9s %9s %8s %9s %8s %8s %9s %9s %8s" % ("family (kernel alone)", "avg_us", "TF/s", "of_157", "of_419.5", "MfmaUtil", "wait_any", "wait_inst", "active",
                                                                            "valu/mfma", "ldsconfl", "GHz", "HBM_rd_MB", "HBM_wr_MB", "waves")]
fam_json = {}
for kern, fam, rows, ci, co in FAM:
    if kern not in a:
        continue
    _, us, va = a[kern]
    vb = b.get(kern, (0, 0, []))[2]
    vc = c.get(kern, (0, 0, []))[2]
    g = lambda v, idx, k: (v[idx[k]] if (k in idx and idx[k] < len(v)) else 0.0)
    gui, wc = g(va, ia, "GRBM_GUI_ACTIVE"), max(g(va, ia, "SQ_WAVE_CYCLES"), 1.0)
    mf = g(vb, ib, "SQ_VALU_MFMA_BUSY_CYCLES")
    util = 4.0 * mf / (128.0 * gui) if gui else 0.0
    tf = 2.0 * rows * ci * co / (us * 1e-6) / 1e12
    nm, nv = g(vc, ic, "SQ_INSTS_MFMA"), g(vc, ic, "SQ_INSTS_VALU")
    rd = 2.0 * d.get(kern, (0, 0, [0.0]))[2][0] / 1024.0
    wr = e.get(kern, (0, 0, [0.0]))[2][0] / 1024.0
    lines.append("%-52s %8.1f %8.1f %6.3f %8.3f %9.3f %9.3f %9.3f %8.3f %9.1f %8.3f %8.2f %9.1f %9.1f %8.0f" % (
        fam[:52], us, tf, tf / 157.3, tf / 419.5, util, g(va, ia, "SQ_WAIT_ANY") / wc, g(va, ia, "SQ_WAIT_INST_ANY") / wc,
        g(va, ia, "SQ_ACTIVE_INST_ANY") / wc, (nv - nm) / nm if nm else 0.0,
        g(vc, ic, "SQ_LDS_BANK_CONFLICT") / max(g(vc, ic, "SQ_LDS_IDX_ACTIVE"), 1.0), gui / us / 1e3 if us else 0.0, rd, wr, g(vc, ic, "SQ_WAVES")))
    fam_json[fam] = {"mfma_util": round(util, 3), "alone_us": round(us, 1), "alone_tflops": round(tf, 1), "kernel": kern}
for nm in ("time.txt", "alone_stats.txt", "beside_stats.txt"):
    path = os.path.join(src, nm)
    if os.path.exists(path):
        lines.append("")
        lines.append("# ---- %s (%s)" % (nm, {"time.txt": "HIP-event timing, un-profiled: every family alone, then the main-stream kernel with its weight-gradient-stream partner launched beside it",
                                              "alone_stats.txt": "rocprofv3 --kernel-trace, every kernel alone", "beside_stats.txt":
                                              "rocprofv3 --kernel-trace, the pairs on two streams as the train step runs them"}[nm]))
        keep = [ln.rstrip()[:200] for ln in open(path) if ("mlp_" in ln or "gram" in ln or ln.startswith("sa") or ln.startswith("kernel") or ln.startswith("#"))]
        lines += keep[:24]
open(os.path.join(P, "%s_pmc_mlp_bwd.txt" % tag), "w").write("\n".join(lines) + "\n")
pj_path = os.path.join(P, "pmc_latest.json")
pj = json.load(open(pj_path)) if os.path.exists(pj_path) else {}
pj["mlp_families"] = {"source": "profiles/%s_pmc_mlp_bwd.txt" % tag, "what": "rocprofv3 --pmc, each family's kernel alone at sa2's / sa1's shape", "families": fam_json}
json.dump(pj, open(pj_path, "w"), indent=1)
print("\n".join(lines[:40]))
