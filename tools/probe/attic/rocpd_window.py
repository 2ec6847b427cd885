"""Per-queue busy time and kernel totals inside a window of a rocprofv3 kernel trace (rocpd .db): the window runs between the
FIRST launches of a marker kernel in call number a and call number b (default marker: the first kernel of sa1's MLP).
    python tools/rocpd_window.py trace.db [marker_substring] [a] [b]"""
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "narrow_stats_kernel"
a, b = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (20, 36)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute("select start, end, name, %s from kernels order by start" % qcol).fetchall()
marks = [r[0] for r in rows if marker in r[2]]
t0, t1 = marks[a], marks[b]
n = b - a
print("window: %d calls, %.3f ms per call" % (n, (t1 - t0) / 1e6 / n))
per, agg = collections.defaultdict(float), collections.defaultdict(lambda: collections.defaultdict(float))
for s, e, name, q in rows:
    if t0 <= s < t1:
        per[q] += e - s
        k = re.sub(r"\(.*", "", name).replace("void ", "").replace("votenet::", "").replace("at::native::", "")[:70]
        agg[q][k] += e - s
for q, t in sorted(per.items(), key=lambda kv: -kv[1]):
    print("== queue %s: busy %.3f ms per call" % (q, t / 1e6 / n))
    for k, v in sorted(agg[q].items(), key=lambda kv: -kv[1])[:14]:
        print("   %8.1f us  %s" % (v / 1e3 / n, k))
