"""One steady-state train step out of a rocprofv3 kernel trace (rocpd .db): the kernels of every queue in launch order with
start (relative to the step's clip_adam-to-clip_adam window), duration and the gap to the previous kernel of the same queue.
    python tools/rocpd_step.py trace.db [which_step]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute("select start, end, name, %s from kernels order by start" % qcol).fetchall()
adam = [r[1] for r in rows if "clip_adam_kernel" in r[2]]
t0, t1 = adam[which - 1], adam[which]
print("step window %.3f ms" % ((t1 - t0) / 1e6))
per = {}
for s, e, n, q in rows:
    if s >= t0 and s < t1:
        per.setdefault(q, []).append((s, e, n))
for q, ks in sorted(per.items(), key=lambda t: -sum(e - s for s, e, _ in t[1])):
    busy = sum(e - s for s, e, _ in ks)
    print("== queue %s: %d kernels, busy %.3f ms" % (q, len(ks), busy / 1e6))
    prev = None
    for s, e, n in ks:
        n = n.replace("votenet::", "").replace("void ", "")
        print("  %8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3 if prev else 0), n[:90]))
        prev = e
