"""Per-queue (HIP stream) kernel time of a rocprofv3 kernel trace: which kernels occupy the main stream's critical chain.
    python tools/rocpd_queue_stats.py trace.db [skip_first_n_kernels] [top]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("votenet::", "")[:58]


def main(path, skip=0, top=22):
    c = sqlite3.connect(path)
    rows = c.execute("select start, end, name, queue_id from kernels order by start").fetchall()[skip:]
    wall = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    per = {}
    for s, e, n, q in rows:
        d = per.setdefault(q, {})
        t = d.setdefault(short(n), [0, 0])
        t[0] += e - s
        t[1] += 1
    print("wall %.3f ms" % wall)
    for q, d in sorted(per.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        tot = sum(v[0] for v in d.values()) / 1e6
        print("queue %s: busy %.3f ms (%.0f%% of wall), %d kernels" % (q, tot, 100 * tot / wall, sum(v[1] for v in d.values())))
        for n, (t, k) in sorted(d.items(), key=lambda kv: -kv[1][0])[:top]:
            print("    %-58s %6d %9.3f ms %5.1f%%" % (n, k, t / 1e6, 100 * t / 1e6 / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 22)
