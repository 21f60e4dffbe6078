#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel time stats and PMC counter sums.
usage: rocpd_summary.py <results.db> [...]"""
import sqlite3, sys, re

def short(n):
    n = re.sub(r"\(.*", "", n)
    return n[-70:]

for f in sys.argv[1:]:
    con = sqlite3.connect(f)
    print("==", f)
    try:
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':70s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
        for r in rows[:25]:
            print(f"{short(r[0]):70s} {r[1]:6d} {r[2]/1e3:10.1f} {r[3]/1e3:9.1f} {r[4]/1e3:9.1f} {r[5]/1e3:9.1f} {100*r[2]/tot:6.1f}")
    except Exception as e:
        print("kernels view:", e)
    try:
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        q = "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name"
        rows = con.execute(q).fetchall()
        if rows:
            print(f"{'kernel':60s} {'counter':28s} {'n':>4s} {'avg/dispatch':>16s}")
            for r in rows:
                if any(k in r[0] for k in ("nn_", "gn_", "compact", "pt2pl")):
                    print(f"{short(r[0])[:60]:60s} {r[1]:28s} {r[2]:4d} {r[4]:16.1f}")
    except Exception as e:
        print("counters:", e, cols if 'cols' in dir() else '')
