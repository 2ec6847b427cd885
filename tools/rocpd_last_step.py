"""Every kernel dispatch of the LAST train step of a rocprofv3 rocpd (.db) trace of tools/serial_step.py, in launch order: name (short),
grid (workgroups), duration -- the per-launch view behind the per-kernel table of tools/rocpd_stats.py.
    python tools/rocpd_last_step.py gpurun_out/serial/t/..._results.db [marker kernel substring = clip_adam]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "clip_adam"
rows = c.execute("select name, start, duration, grid_x, workgroup_x from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if marker in r[0]]
lo, hi = ends[-2] + 1, ends[-1] + 1
tot = 0.0
for name, start, dur, gx, wx in rows[lo:hi]:
    short = re.sub(r"\(.*", "", name).replace("votenet::", "").replace("void ", "")
    print("%-70s %6d wg  %8.1f us" % (short[:70], gx // max(wx, 1), dur / 1e3))
    tot += dur
print("# %d dispatches, %.3f ms" % (hi - lo, tot / 1e6))
