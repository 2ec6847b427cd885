"""Compose profiles/rNN_pmc_pieces.txt from a tools/pmc_pieces.sh run:  python tools/pmc_pieces_summary.py r03
Per kernel of one SA level on the piece layout, alone on the GPU: duration (kernel trace), HBM bytes per launch (FETCH_SIZE x 2: gfx950
counts 128-byte requests at 64 B; WRITE_SIZE as reported), the HBM rate that is and its share of 8 TB/s, MfmaUtil = 4 * SQ_VALU_MFMA_BUSY_CYCLES
/ (128 SIMDs per XCD * GRBM_GUI_ACTIVE) (per-XCD samples)."""
import os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = os.path.join(R, "gpurun_out", "pmc_pieces"), sys.argv[1]


def table(name):
    rows, hdr = {}, []
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return rows, hdr
    for ln in open(path):
        if ln.startswith("#") or not ln.strip():
            continue
        if ln.startswith("kernel"):
            hdr = ln.split()[3:]
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+(.*)$", ln.rstrip())
        if m:
            rows[m.group(1).strip()[:64]] = (int(m.group(2)), float(m.group(3)), [float(v) if v != "-" else 0.0 for v in m.group(4).split()])
    return rows, hdr


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("votenet::", "")[:64]


# the kernels of the level itself (its GEMMs carry level-specific template arguments or run REP times more often than the others)
LEVEL_KERNELS = {
    "sa2": ["mlp_linear_fast_kernel<2, 2, 2, 2, 0, 8, true", "mlp_linear_fast_kernel<2, 2, 2, 2, 4, 0, true", "mlp_linear_fast_kernel<2, 2, 2, 2, 5, 6", "mlp_linear_fast_kernel<2, 2, 2, 2, 5, 1",
            "group_linear_bwd_masked_kernel", "assembled_point_grad_kernel", "assembled_wx_finish",
            "mlp_linear_fast_kernel<2, 2, 2, 2, 0, 1, false", "mlp_linear_fast_kernel<2, 2, 2, 2, 0, 1, true", "mlp_wgrad_fast_kernel<3, 2, 2, 4", "pool_wgrad_sparse_kernel<128, 16", "pool_wgrad_sparse_centre_kernel<128",
            "pool_dgrad_scatter_wave_kernel<128, 256, 16", "gram_bf3_kernel<128", "group_linear_bwd_sorted_kernel", "bn_pool_finalize_half",
            "bn_bwd_reduce_zsel", "assemble_rows_half", "half_sort", "half_groups", "pool_wgrad_finish", "pool_dgrad_prepare_kernel<256>"],
    "sa1": ["mlp_linear_fast_kernel<2, 2, 2, 2, 0, 8, true", "mlp_linear_fast_kernel<4, 1, 1, 2, 3, 0, true", "mlp_linear_fast_kernel<4, 1, 1, 2, 5, 4", "mlp_linear_fast_kernel<4, 1, 1, 2, 5, 7",
            "mlp_linear_fast_kernel<4, 1, 1, 2, 0, 1, false", "mlp_linear_fast_kernel<4, 1, 1, 2, 0, 1, true", "mlp_wgrad_fast_kernel<2, 1, 1, 4", "pool_wgrad_sparse_kernel<64, 16", "pool_wgrad_sparse_centre_kernel<64",
            "pool_dgrad_scatter_wave_kernel<64, 128, 16", "gram_bf3_kernel<64", "bn_pool_finalize_half", "bn_bwd_reduce_zsel", "narrow_rows_half",
            "half_groups", "pool_wgrad_finish", "pool_dgrad_prepare_kernel<128>", "narrow_wgrad_first"],
}
# bench.py's family names (mlp._Timed shapes) -> (level, kernel, cin, cout) for the mlp_families block of profiles/pmc_latest.json
FAMILIES = {
    "fwd+pool half": ("sa2", "mlp_linear_fast_kernel<2, 2, 2, 2, 0, 8, true", 128, 256),
    "fwd+bn assembled half": ("sa2", "mlp_linear_fast_kernel<2, 2, 2, 2, 4, 0, true", 128, 128),
    "dgrad_bn_reduce assembled half": ("sa2", "mlp_linear_fast_kernel<2, 2, 2, 2, 5, 6, true", 128, 128),
    "dgrad_bn half": ("sa2", "mlp_linear_fast_kernel<2, 2, 2, 2, 5, 1, true", 128, 128),
    "wgrad_bn assembled half": ("sa2", "mlp_wgrad_fast_kernel<3, 2, 2, 4, true>", 128, 128),
    "gram half": ("sa2", "gram_bf3_kernel<128", 128, 128),
    "gram-form dense dgrad half": ("sa2", "mlp_linear_fast_kernel<2, 2, 2, 2, 0, 1, true", 128, 128),
    "fwd+bn narrow half": ("sa1", "mlp_linear_fast_kernel<4, 1, 1, 2, 3, 0, true", 64, 64),
    "dgrad_bn_reduce narrow half": ("sa1", "mlp_linear_fast_kernel<4, 1, 1, 2, 5, 7, true", 64, 64),
    "wgrad_bn narrow half": ("sa1", "mlp_wgrad_fast_kernel<2, 1, 1, 4, true>", 64, 64),
}
fam_out = {}
out = ["# rocprofv3 over tools/pmc_pieces.py (tools/pmc_pieces.sh): every kernel of ONE set-abstraction level's forward + backward on the piece",
       "# layout (csrc/half.hip), alone on one stream, real geometry / activations of room scenes (8 x 20480 points).  Durations: --kernel-trace;",
       "# HBM MB per launch: --pmc FETCH_SIZE (x 2: gfx950 counts 128-byte requests at 64 B) and WRITE_SIZE, separate passes; GB/s = (rd + wr) / duration,",
       "# of_8TB/s against the HBM3E peak; MfmaUtil = 4 * SQ_VALU_MFMA_BUSY_CYCLES / (128 SIMDs * GRBM_GUI_ACTIVE) (per-XCD samples).",
       "# The backward passes of a level are HBM-bound at these sizes: what bounds them is the row traffic, not the matrix pipe."]
for lv in ("sa2", "sa1"):
    tr = {}
    hdrline = None
    p = os.path.join(src, lv + "_trace.txt")
    if not os.path.exists(p):
        continue
    for ln in open(p):
        if ln.startswith("#") or ln.startswith("kernel") or not ln.strip():
            continue
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s+(\d+)", ln.rstrip())
        if m:
            tr[short(m.group(1))] = (int(m.group(2)), float(m.group(4)), int(m.group(8)), int(m.group(10)))
    fe, _ = table(lv + "_fetch.txt")
    wr, _ = table(lv + "_write.txt")
    sq, sqh = table(lv + "_sq.txt")
    head = open(os.path.join(src, lv + "_trace.log")).read()
    m = re.search(r"%s: .*" % lv, head)
    out.append("")
    out.append("== " + (m.group(0) if m else lv))
    out.append("%-62s %6s %9s %10s %10s %9s %9s %9s %6s" % ("kernel (alone)", "calls", "avg_us", "HBM_rd_MB", "HBM_wr_MB", "GB/s", "of_8TB/s", "MfmaUtil", "vgpr"))
    for k, (calls, us, vgpr, lds) in sorted(tr.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if not any(t in k for t in LEVEL_KERNELS[lv]):
            continue  # (the warm-up forward pass of the whole network and the geometry chain run in the same process)
        rd = 2 * fe.get(k, (0, 0, [0.0]))[2][0] / 1024.0  # KB -> MB
        w = wr.get(k, (0, 0, [0.0]))[2][0] / 1024.0
        util = ""
        if k in sq and sqh:
            d = dict(zip(sqh, sq[k][2]))
            if d.get("GRBM_GUI_ACTIVE", 0) > 0:
                util = "%.3f" % (4.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * d["GRBM_GUI_ACTIVE"]))
        gbs = (rd + w) * 1e6 / (us * 1e-6) / 1e9 if us > 0 else 0.0
        for fname, (flv, fk, cin, cout) in FAMILIES.items():
            if flv == lv and k.startswith(fk) and not (fname == "fwd+pool half" and lv != "sa2"):
                rows_c = int(re.search(r"= (\d+) compact rows", m.group(0)).group(1)) if m else 0
                fam_out[fname] = {"mfma_util": float(util) if util else None, "alone_us": round(us, 1),
                                  "alone_tflops": round(2.0 * rows_c * cin * cout / (us * 1e-6) / 1e12, 1), "hbm_gbs_alone": round(gbs),
                                  "hbm_frac_alone": round(gbs / 8000.0, 3), "shape": [rows_c, cin, cout]}
        out.append("%-62s %6d %9.1f %10.1f %10.1f %9.0f %9.3f %9s %6d" % (k[:62], calls, us, rd, w, gbs, gbs / 8000.0, util, vgpr))
open(os.path.join(R, "profiles", "%s_pmc_pieces.txt" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))

import json
pj = os.path.join(R, "profiles", "pmc_latest.json")
d = json.load(open(pj))
mf = d.setdefault("mlp_families", {"families": {}})
mf["families"].update(fam_out)
mf["source_pieces"] = "profiles/%s_pmc_pieces.txt" % tag
json.dump(d, open(pj, "w"), indent=1)
print(json.dumps(fam_out, indent=1))
