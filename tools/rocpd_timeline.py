#!/usr/bin/env python
"""Idle gaps of the GPU timeline from a rocprofv3 --kernel-trace rocpd database: union of the kernel intervals, the largest
gaps and the kernels on both sides (a gap of milliseconds = a host synchronisation or an allocation, not a launch gap).

    python tools/rocpd_timeline.py results.db [n_gaps]
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = [c for c in cols if "queue" in c or "stream" in c]
    rows = db.execute("select start, end, name%s from kernels order by start" % ("".join(", " + c for c in qcol[:1]))).fetchall()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_end, gaps = 0, rows[0][0], []
    last_name = ""
    for r in rows:
        s, e, name = r[0], r[1], r[2]
        if s > cur_end:
            gaps.append((s - cur_end, cur_end - t0, last_name, name))
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
        if e >= cur_end:
            last_name = name
    short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:60]
    print("span %.1f ms, busy (union) %.1f ms, idle %.1f ms, kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(rows)))
    for g, at, a, b in sorted(gaps, reverse=True)[:top]:
        print("gap %8.3f ms at %9.1f ms  after %-60s before %s" % (g / 1e6, at / 1e6, short(a), short(b)))


if __name__ == "__main__":
    main()
