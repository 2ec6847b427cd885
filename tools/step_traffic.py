"""HBM bytes of a bench run by kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd .db each):
    python tools/step_traffic.py fetch.db write.db <steps in the run>
FETCH_SIZE counts 128-byte requests at 64 B on gfx950 -> x 2 (MI355X_MICROARCH.md); both counters in KB.  Per kernel: launches per step,
MB read / written per step, share of the step's traffic."""
import re, sqlite3, sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("votenet::", "")


def load(path, counter):
    c = sqlite3.connect(path)
    out = {}
    for name, cnt, total, dur in c.execute("select name, count(*), sum(counter_value), sum(duration) from pmc_events where counter_name = ? group by name", (counter,)):
        out[short(name)] = (cnt, total, dur)
    return out


def main(fetch_db, write_db, steps):
    rd, wr = load(fetch_db, "FETCH_SIZE"), load(write_db, "WRITE_SIZE")
    rows = []
    for k in sorted(set(rd) | set(wr)):
        cnt = (rd.get(k) or wr.get(k))[0]
        r = (rd.get(k, (0, 0, 0))[1] or 0) * 2 * 1024 / 1e6   # KB -> MB, x 2
        w = (wr.get(k, (0, 0, 0))[1] or 0) * 1024 / 1e6
        dur = (rd.get(k) or wr.get(k))[2] / 1e3
        rows.append((r + w, k, cnt, r, w, dur))
    tot = sum(x[0] for x in rows)
    print("# all dispatches of the run (%d steps incl. 8 set-up and 2 warm-up steps): %.1f GB read + written = %.2f GB per step" % (steps, tot / 1e3, tot / 1e3 / steps))
    print("%-66s %9s %11s %11s %7s %10s" % ("kernel", "per step", "rd MB/step", "wr MB/step", "share", "GB/s (pmc)"))
    for t, k, cnt, r, w, dur in sorted(rows, reverse=True)[:45]:
        print("%-66s %9.1f %11.1f %11.1f %6.1f%% %10.0f" % (k[:66], cnt / steps, r / steps, w / steps, 100 * t / tot, t / 1e3 / (dur * 1e-6) if dur else 0))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
