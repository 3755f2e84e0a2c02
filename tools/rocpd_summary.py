#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs: per-kernel time stats and PMC counter averages."""
import sqlite3, sys, re

def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("scship::", "")
    return name[:90]

def kernel_stats(db):
    con = sqlite3.connect(db)
    q = "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by sum(end-start) desc"
    rows = con.execute(q).fetchall()
    tot = sum(r[2] for r in rows) or 1
    # "exec" columns ignore launches shorter than 20 us: CG-step kernels are enqueued in chunks and
    # return immediately once the device-side convergence flag is set (no host sync per CG step)
    ex = {r[0]: (r[1], r[2]) for r in con.execute(
        "select name, count(*), avg(end-start) from kernels where (end-start) > 20000 group by name")}
    print("%-90s %7s %11s %9s %9s %9s %6s %8s %10s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "exec_n", "exec_avg"))
    for r in rows[:40]:
        e = ex.get(r[0], (0, 0.0))
        print("%-90s %7d %11.1f %9.2f %9.2f %9.2f %6.1f %8d %10.2f" % (short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, e[0], e[1] / 1e3))

def counter_stats(db, executed=False):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, counter_name, count(*), avg(value) from counters_collection group by %s, counter_name order by %s" % (namecol, namecol, namecol)
    if executed:  # only launches that did their work (CG-step kernels return at once when the convergence flag is set)
        q = ("select c.%s, c.counter_name, count(*), avg(c.value) from counters_collection c join "
             "(select %s as k, counter_name as cn, max(value) as mx from counters_collection group by %s, counter_name) m "
             "on c.%s = m.k and c.counter_name = m.cn where c.value > 0.5 * m.mx group by c.%s, c.counter_name order by c.%s"
             % (namecol, namecol, namecol, namecol, namecol, namecol))
    print("%-90s %-28s %8s %16s" % ("kernel", "counter", "calls", "avg_value"))
    for r in con.execute(q):
        print("%-90s %-28s %8d %16.1f" % (short(r[0]), r[1], r[2], r[3]))

if __name__ == "__main__":
    executed = "--executed" in sys.argv
    for db in [a for a in sys.argv[1:] if not a.startswith("--")]:
        print("==", db)
        con = sqlite3.connect(db)
        n = con.execute("select count(*) from counters_collection").fetchone()[0]
        if n:
            counter_stats(db, executed)
        else:
            kernel_stats(db)
