"""Per-kernel averages of the counters in a rocprofv3 --pmc run (rocpd .db):
    python tools/rocpd_pmc.py gpurun_out/pmc_x/p_results.db
prints kernel (short name), dispatches, mean duration (us) and the mean of each counter per dispatch."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("votenet::", "")


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                     "group by name, counter_name").fetchall()
    table, counters = {}, []
    for name, cn, cnt, val, dur in rows:
        table.setdefault(name, {})[cn] = (cnt, val, dur)
        if cn not in counters:
            counters.append(cn)
    print("# source: %s" % path)
    print("%-64s %6s %10s " % ("kernel", "calls", "avg_us") + " ".join("%22s" % cn for cn in counters))
    for name, d in sorted(table.items(), key=lambda kv: -max(v[0] * v[2] for v in kv[1].values())):
        cnt, _, dur = next(iter(d.values()))
        print("%-64s %6d %10.2f " % (short(name)[:64], cnt, dur / 1e3) + " ".join("%22.1f" % d[cn][1] if cn in d else "%22s" % "-"
                                                                                  for cn in counters))


if __name__ == "__main__":
    main(sys.argv[1])
