#!/usr/bin/env python
"""Condensed phase view of one training step from a tools/rocpd_sequence.py listing: consecutive launches of the same kernel on the same
queue merged into one line (start, queue, kernel, launches, span, busy time); lines below --min-us of busy time are dropped.
    python tools/step_phases.py profiles/r5z_step_sequence.txt [--min-us 300] > profiles/r5z_step_phases.txt"""
import sys

path = sys.argv[1]
min_us = float(sys.argv[sys.argv.index("--min-us") + 1]) if "--min-us" in sys.argv else 300.0
rows, head = [], ""
for line in open(path):
    if line.startswith("#"):
        head = head or line.strip()
        continue
    p = line.split(None, 4)
    if len(p) == 5:
        rows.append((float(p[0]), float(p[1]), int(p[2]), int(p[3]), p[4].split("(")[0][:44]))
out = []
for st, du, q, g, name in rows:
    key = (q, name)
    if out and out[-1][0] == key and st - (out[-1][1] + out[-1][2]) < 50:
        out[-1][2] = st + du - out[-1][1]
        out[-1][3] += 1
        out[-1][4] += du
    else:
        out.append([key, st, du, 1, du])
print(head)
print("# start_ms queue kernel launches span_us busy_us   (runs of one kernel on one queue; >= %.0f us of busy time)" % min_us)
for key, st, span, n, busy in out:
    if busy >= min_us:
        print("%8.2f  q%d  %-46s x%-3d %9.1f %9.1f" % (st / 1e3, key[0], key[1], n, span, busy))
