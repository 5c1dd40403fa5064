#!/usr/bin/env python
"""Summarise the four rocprofv3 --pmc passes of tools/pmc_rel_attn.sh (dense|factored x FETCH_SIZE|WRITE_SIZE) into
profiles/rN_rel_attn_pmc.json: HBM bytes per launch of every relation-attention kernel, per operand mode.
Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are KiB; on gfx950
FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads -- these kernels read everything that way
(8 bf16 channels per lane) -- so it is doubled; WRITE_SIZE is taken as is."""
import glob
import json
import os
import re
import sqlite3
import sys


def counters(dbdir):
    dbs = glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True)
    out = {}
    for db in dbs:
        con = sqlite3.connect(db)
        for name, cname, cnt, val, dur in con.execute(
                "select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name"):
            short = re.sub(r"\(anonymous namespace\)::|void ", "", name)
            out[(short.split("<")[0], cname)] = (cnt, val, (dur or 0) / 1e3)
    return out


def main():
    root, dst = sys.argv[1], sys.argv[2]
    res = {"config": "C2", "dtype": "bf16",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, one pair of passes per "
                     "operand mode over `tools/bench_rel_attn.py --mode dense|factored` (tools/pmc_rel_attn.sh); FETCH_SIZE KiB x 1024 "
                     "x 2 (gfx950: wide coalesced reads are tallied at half their bytes, MI355X_MICROARCH.md), WRITE_SIZE KiB x 1024"}
    for mode in ("dense", "factored"):
        f = counters(os.path.join(root, mode + "_FETCH_SIZE"))
        w = counters(os.path.join(root, mode + "_WRITE_SIZE"))
        res[mode] = {}
        for (kern, cname), (cnt, val, dur) in sorted(f.items()):
            if cname != "FETCH_SIZE" or not kern.startswith("rel_attn"):
                continue
            wv = w.get((kern, "WRITE_SIZE"), (0, 0.0, 0.0))
            fetch, write = val * 1024 * 2, wv[1] * 1024
            res[mode][kern] = {"launches_sampled": cnt, "avg_us_under_pmc": round(dur, 1), "fetch_bytes_per_launch": fetch,
                               "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write}
            print("%-9s %-28s n=%3d  FETCH %.3f GB (x2 corrected)  WRITE %.3f GB  traffic %.3f GB  %.1f us" % (
                mode, kern, cnt, fetch / 1e9, write / 1e9, (fetch + write) / 1e9, dur))
    json.dump(res, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
