#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel stats table (like --stats CSV).

    python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [out.csv] [--by-grid]

--by-grid keeps launches of one kernel with different grid sizes apart (e.g. the relation-attention kernel runs on
6464 workgroups in the graph encoder and on 3200 in the decoder): the name gets a " grid=N" suffix.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110] if " grid=" not in name else name.split(" grid=")[0][:96] + " grid=" + name.split(" grid=")[1]


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    by_grid = "--by-grid" in sys.argv
    db = sqlite3.connect(argv[0])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    key = name_col
    if by_grid:
        gcols = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
        wcols = [c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
        if gcols:
            g = gcols[0]
            # rocpd stores the grid in work-items: divide by the workgroup size when it is there
            gexpr = "(%s / %s)" % (g, wcols[0]) if wcols else g
            key = "%s || ' grid=' || %s" % (name_col, gexpr)
        else:
            print("# no grid column in %s" % cols, file=sys.stderr)
    rows = cur.execute("select %s as k, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by k order by 3 desc" % key).fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["name,calls,total_ms,avg_us,min_us,max_us,pct"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%.3f,%.1f,%.1f,%.1f,%.2f' % (short(n), c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    lines.append('"TOTAL",%d,%.3f,,,,100' % (sum(r[1] for r in rows), tot / 1e6))
    text = "\n".join(lines)
    if len(argv) > 1:
        open(argv[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
