"""Idle gaps on the busiest queue (the main stream) of a rocprofv3 kernel trace: histogram + the largest ones with the
kernels on either side.   python tools/rocpd_gaps.py trace.db [skip_first_n_kernels]"""
import re, sqlite3, sys


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("votenet::", "")[:48]


def main(path, skip=0):
    c = sqlite3.connect(path)
    rows = c.execute("select start, end, name, queue_id from kernels order by start").fetchall()[skip:]
    per = {}
    for r in rows:
        per.setdefault(r[3], []).append(r)
    q = max(per, key=lambda k: sum(e - s for s, e, _, _ in per[k]))
    ks = per[q]
    gaps = [(ks[i + 1][0] - ks[i][1], short(ks[i][2]), short(ks[i + 1][2])) for i in range(len(ks) - 1)]
    tot = sum(g for g, _, _ in gaps if g > 0)
    wall = ks[-1][1] - ks[0][0]
    print("queue %s: wall %.2f ms, idle %.2f ms (%.1f%%), %d kernels" % (q, wall / 1e6, tot / 1e6, 100.0 * tot / wall, len(ks)))
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
        sel = [g for g, _, _ in gaps if lo * 1e3 <= g < hi * 1e3]
        print("  gaps %3g-%-4g us: %5d  total %.3f ms" % (lo, hi if hi < 1e9 else float("inf"), len(sel), sum(sel) / 1e6))
    agg = {}
    for g, a, b in gaps:
        if g > 0:
            k = a + "  ->  " + b
            t = agg.setdefault(k, [0, 0])
            t[0] += g
            t[1] += 1
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
        print("  %8.3f ms %4d x %6.1f us  %s" % (t / 1e6, n, t / n / 1e3, k))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
