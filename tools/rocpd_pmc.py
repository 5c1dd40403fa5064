#!/usr/bin/env python
"""Per-kernel PMC sums from a rocprofv3 --pmc rocpd database:  python tools/rocpd_pmc.py db [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
                  "group by name, counter_name").fetchall()
for name, cname, cnt, val, dur in rows:
    if pat in name:
        short = name.replace("(anonymous namespace)::", "")[:90]
        print("%-90s %-14s n=%4d avg=%14.1f avg_dur_us=%9.1f" % (short, cname, cnt, val, (dur or 0) / 1e3))
