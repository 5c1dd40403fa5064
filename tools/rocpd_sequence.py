#!/usr/bin/env python
"""One training step of a rocprofv3 --kernel-trace rocpd database as an ordered kernel list: start offset, duration, queue,
grid (workgroups), short name.  The step is delimited by the launches of a marker kernel that runs once per step
(default: the flat gradient square-norm, ``sqnorm_kernel``): the k-th interval between two markers is printed.

    python tools/rocpd_sequence.py results.db [--step K] [--marker sqnorm_kernel] [--min-us 0] > step.txt
"""
import re
import sqlite3
import sys


def main():
    argv = sys.argv[1:]
    def opt(name, default):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        return default
    step = int(opt("--step", "2"))
    marker = opt("--marker", "sqnorm_kernel")
    min_us = float(opt("--min-us", "0"))
    db = sqlite3.connect(argv[0])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = ([c for c in cols if "queue" in c] + [c for c in cols if "stream" in c] + ["0"])[0]
    gcol = ([c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols] + ["0"])[0]
    wcol = ([c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols] + ["1"])[0]
    rows = db.execute("select start, end, name, %s, %s, %s from kernels order by start" % (qcol, gcol, wcol)).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < step + 2:
        raise SystemExit("only %d marker launches (%s)" % (len(marks), marker))
    lo, hi = marks[step] + 1, marks[step + 1] + 1
    t0 = rows[lo][0]
    short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:90]
    queues = {}
    print("# step %d: %d kernels, %.3f ms wall" % (step, hi - lo, (max(r[1] for r in rows[lo:hi]) - t0) / 1e6))
    print("#  start_us   dur_us  q   groups  kernel")
    for s, e, name, q, g, w in rows[lo:hi]:
        if (e - s) / 1e3 < min_us:
            continue
        qi = queues.setdefault(q, len(queues))
        print("%10.1f %8.1f %2d %8d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, qi, (g // w) if w else g, short(name)))


if __name__ == "__main__":
    main()
