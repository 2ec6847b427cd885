"""profiles/rNN_pmc_mlp.txt from the three rocpd_pmc.py summaries of tools/pmc_mlp.py (SQ pass, FETCH_SIZE pass, WRITE_SIZE pass):
    python tools/pmc_mlp_table.py gpurun_out/r01c > profiles/r01_pmc_mlp.txt"""
import re
import sys


def parse(fn):
    rows = {}
    lines = open(fn).read().splitlines()
    hdr = lines[1].split()
    for l in lines[2:]:
        m = re.match(r"(.{64})\s+(\d+)\s+([\d.]+)\s+(.*)", l)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), [float(v) if v != "-" else None for v in m.group(4).split()])
    return hdr[3:], rows


def main(d):
    h, sq = parse(d + "/mlp_sq.txt")
    _, fe = parse(d + "/mlp_fetch.txt")
    _, wr = parse(d + "/mlp_write.txt")
    desc = {"mlp_linear_fast_kernel<2, 2, 2, 2, 0, 0>": "forward (folded BN+ReLU in, stats out), mean of both shapes",
            "mlp_linear_fast_kernel<4, 1, 1, 2, 2, 0>": "dgrad + folded BN backward (pooled), sa1 L2 128->64",
            "mlp_linear_fast_kernel<2, 2, 2, 2, 2, 0>": "dgrad + folded BN backward (pooled), sa2 L2 256->128",
            "mlp_wgrad_fast_kernel<0, 2, 2, 2>": "wgrad + folded BN backward (pooled), sa2 L2",
            "mlp_wgrad_fast_kernel<0, 1, 2, 2>": "wgrad + folded BN backward (pooled), sa1 L2"}
    print("# rocprofv3 --pmc on tools/pmc_mlp.py (separate passes: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES | FETCH_SIZE | WRITE_SIZE)")
    print("# two layers, 17.18 GFLOP each: sa1 L2 = 1048576 x 64 -> 128 (HBM-bound, 21 flop/B), sa2 L2 = 262144 x 128 -> 256 (MFMA-bound, 43 flop/B)")
    print("# SQ/GRBM counters are reported per XCD (8 rows per dispatch; means below are per XCD).  SQ_VALU_MFMA_BUSY_CYCLES counts quad-cycles:")
    print("#   16 per v_mfma_f32_32x32x2_f32 (= 64 cycles), so  MfmaUtil = 4 * MFMA_BUSY / (128 SIMDs per XCD * GRBM_GUI_ACTIVE).")
    print("# FETCH_SIZE / WRITE_SIZE in KB per dispatch; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md).")
    print("%-46s %8s %12s %12s %9s %9s %11s %11s  %s" % ("kernel", "avg_us", "GUI_ACTIVE", "MFMA_BUSY", "MfmaUtil", "TFLOP/s", "HBM_rd_MB",
                                                         "HBM_wr_MB", "what"))
    for k, d_ in desc.items():
        if k not in sq:
            continue
        _, us, v = sq[k]
        gui, mf = v[h.index("GRBM_GUI_ACTIVE")], v[h.index("SQ_VALU_MFMA_BUSY_CYCLES")]
        print("%-46s %8.1f %12.0f %12.0f %9.3f %9.1f %11.1f %11.1f  %s" % (k, us, gui, mf, 4 * mf / (128 * gui), 17.18e9 / us / 1e6,
                                                                           fe[k][2][0] * 2 / 1024, wr[k][2][0] / 1024, d_))
    print("# algorithmic bytes per launch: forward sa1 L2 805 MB (x 268 + z 537), sa2 L2 403 MB; dgrad sa1 L2 805 MB (z 537 + da 268), sa2 L2 403 MB (+ gout/argmax / k);")
    print("# wgrad reads its dz source once per 64/128-row block of W: sa1 L2 805 MB, sa2 L2 403 MB (+ atomically added partial dW tiles).")


if __name__ == "__main__":
    main(sys.argv[1])
