"""Text summary of a rocprofv3 run kept as a rocpd SQLite database (the default output of rocprofv3 7.x: <dir>/<host>/<pid>_results.db):
per-kernel statistics (the table `rocprofv3 --stats` prints) and the kernel timeline of the last iterations with the HIP
queue each kernel ran on -- what shows a collective on the side stream running UNDER a GEMV launch on the main stream.
    python tools/rocpd_timeline.py <results.db> [n_last_kernels]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    print("== per-kernel statistics (name, calls, total us, average us, % of kernel time)")
    for name, calls, tot, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-58s %6d %12.1f %10.2f %6.2f" % (short(name), calls, tot, avg, pct))
    rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    sel = rows[-nlast:]
    t0 = sel[0][1]
    print("\n== timeline of the last %d kernels (queue, start us, duration us, kernel); gaps on a queue are idle time" % len(sel))
    for name, st, en, q in sel:
        print("q%-3d %10.1f %9.1f  %s" % (q, (st - t0) / 1e3, (en - st) / 1e3, short(name)))


if __name__ == "__main__":
    main()
