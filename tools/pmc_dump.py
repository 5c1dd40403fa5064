#!/usr/bin/env python
"""Dump the pmc_events of a rocprofv3 --pmc output directory as TSV: launch order, kernel, grid, duration, counter, value."""
import glob
import os
import re
import sqlite3
import sys

for db in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(pmc_events)")]
    want = [c for c in ("dispatch_id", "name", "grid_size", "grid_size_x", "workgroup_size", "duration", "counter_name", "counter_value") if c in cols]
    order = "dispatch_id" if "dispatch_id" in cols else "rowid"
    print("\t".join(want))
    for row in con.execute("select %s from pmc_events order by %s" % (", ".join(want), order)):
        row = list(row)
        if "name" in want:
            i = want.index("name")
            row[i] = re.sub(r"\(anonymous namespace\)::|void ", "", str(row[i]))[:60]
        print("\t".join(str(v) for v in row))
