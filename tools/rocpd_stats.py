#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel stats table (like --stats CSV).

    python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [out.csv]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["name,calls,total_ms,avg_us,min_us,max_us,pct"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%.3f,%.1f,%.1f,%.1f,%.2f' % (short(n), c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    lines.append('"TOTAL",%d,%.3f,,,,100' % (sum(r[1] for r in rows), tot / 1e6))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
