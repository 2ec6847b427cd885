"""Busy / idle accounting of a rocprofv3 kernel trace (rocpd .db): union of kernel intervals vs wall, per-queue busy time,
and the largest gaps with the kernels on either side.   python tools/rocpd_timeline.py trace.db [skip_first_n_kernels]"""
import sqlite3
import sys


def main(path, skip=0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute("select start, end, name%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
    rows = rows[skip:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    last_name = rows[0][2]
    for r in rows[1:]:
        if r[0] > cur_e:
            busy += cur_e - cur_s
            gaps.append((r[0] - cur_e, last_name[:60], r[2][:60]))
            cur_s, cur_e = r[0], r[1]
        else:
            cur_e = max(cur_e, r[1])
        if r[1] >= cur_e:
            last_name = r[2]
    busy += cur_e - cur_s
    print("wall %.3f ms, busy (union) %.3f ms, idle %.3f ms (%.1f%%), kernel-time sum %.3f ms" % (
        (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / (t1 - t0), sum(r[1] - r[0] for r in rows) / 1e6))
    if qcol:
        per = {}
        for r in rows:
            per[r[3]] = per.get(r[3], 0) + r[1] - r[0]
        print("per %s busy ms:" % qcol, {k: round(v / 1e6, 3) for k, v in per.items()})
    gaps.sort(reverse=True)
    print("gap histogram: >100us %d, 20-100us %d, 5-20us %d, <5us %d; total gap in <20us gaps %.3f ms" % (
        sum(g[0] > 1e5 for g in gaps), sum(2e4 < g[0] <= 1e5 for g in gaps), sum(5e3 < g[0] <= 2e4 for g in gaps),
        sum(g[0] <= 5e3 for g in gaps), sum(g[0] for g in gaps if g[0] <= 2e4) / 1e6))
    for g in gaps[:12]:
        print("  gap %8.1f us  after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
