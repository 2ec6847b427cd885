"""Summarise a rocprofv3 rocpd (.db) kernel trace into the same table `rocprofv3 --stats` prints:
per-kernel calls, total / average / min / max duration and share of GPU kernel time.

    python tools/rocpd_stats.py gpurun_out/prof/xyz_results.db > profiles/rNN_xyz_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("# source: %s" % path)
    print("# total GPU kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    print("%-100s %7s %12s %11s %11s %11s %6s %5s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct",
                                                                "vgpr", "agpr", "lds"))
    for r in rows[:top]:
        name = r[0] if len(r[0]) <= 100 else r[0][:97] + "..."
        print("%-100s %7d %12.1f %11.2f %11.2f %11.2f %6.2f %5d %5d %7d" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                          r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
