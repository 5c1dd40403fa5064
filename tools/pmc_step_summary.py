#!/usr/bin/env python
"""Per-kernel HBM traffic of the training step from the two rocprofv3 --pmc passes of tools/pmc_step.sh.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies wide coalesced reads at half their bytes,
/opt/skills/guides/MI355X_MICROARCH.md "HBM") -- every kernel listed reads with 16-byte lanes or LDS-DMA."""
import glob
import json
import os
import re
import sqlite3
import sys

KEEP = ("rel_attn", "gru_", "seg_sum", "gemm", "copy_nll", "ln_", "adam", "embed_rows")


def counters(dbdir):
    out = {}
    for db in glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for name, cname, cnt, val, dur, tot in con.execute(
                "select name, counter_name, count(*), avg(counter_value), avg(duration), sum(counter_value) from pmc_events "
                "group by name, counter_name"):
            short = re.sub(r"\(anonymous namespace\)::|void ", "", name)
            short = re.sub(r"\(.*$", "", short)
            out[(short, cname)] = (cnt, val, (dur or 0) / 1e3, tot)
    return out


def main():
    root, dst = sys.argv[1], sys.argv[2]
    f = counters(os.path.join(root, "FETCH_SIZE"))
    w = counters(os.path.join(root, "WRITE_SIZE"))
    res = {"config": "C2", "dtype": "bf16", "steps_traced": 3,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `python bench.py --steps 2 "
                     "--warmup 1` (tools/pmc_step.sh); FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction), WRITE_SIZE KiB x 1024; "
                     "averages over all launches of a kernel name (all shapes)", "kernels": {}}
    rows = []
    for (kern, cname), (cnt, val, dur, tot) in f.items():
        if cname != "FETCH_SIZE" or not kern.startswith(KEEP):
            continue
        wv = w.get((kern, "WRITE_SIZE"), (0, 0.0, 0.0, 0.0))
        fetch, write = val * 1024 * 2, wv[1] * 1024
        rows.append((tot * 2048 + wv[3] * 1024, kern, cnt, fetch, write, dur))
    for tot, kern, cnt, fetch, write, dur in sorted(rows, reverse=True)[:24]:
        res["kernels"][kern] = {"launches_sampled": cnt, "avg_us_under_pmc": round(dur, 1), "fetch_bytes_per_launch": fetch,
                                "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
                                "GBps_under_pmc": round((fetch + write) / max(dur, 1e-9) / 1e3, 1)}
        print("%-70s n=%4d  FETCH %8.3f GB  WRITE %8.3f GB  traffic %8.3f GB  %8.1f us  %7.1f GB/s" % (
            kern[:70], cnt, fetch / 1e9, write / 1e9, (fetch + write) / 1e9, dur, (fetch + write) / max(dur, 1e-9) / 1e3))
    json.dump(res, open(dst, "w"), indent=1)


if __name__ == "__main__":
    main()
