#!/usr/bin/env python
"""Summarise gpurun_out/pmc_gemm/g*.tsv (tools/pmc_gemm.sh): per shape of the K sweep, the average over the 3 timed
launches of every counter (instances summed; *_avr counters averaged) and derived figures:
  clock      = GRBM_GUI_ACTIVE / 8 XCDs / duration
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM cycles per XCD)     (16 cycles per v_mfma_f32_16x16x32_bf16)
  wait_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
    python tools/pmc_gemm_summary.py gpurun_out/pmc_gemm > profiles/r2z_gemm_pmc_ksweep.txt"""
import collections
import csv
import glob
import os
import sys

SHAPES = ["K=32", "K=128", "K=256", "K=512", "K=1024", "K=2016", "K=32 N=128 (M x8)", "K=32 N=256 (M x4)", "K=32 N=4096 (M /4)"]


def main():
    root = sys.argv[1]
    res = collections.OrderedDict()
    for path in sorted(glob.glob(os.path.join(root, "g*.tsv"))):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(path), delimiter="\t"):
            if "gemm" not in r["name"]:
                continue
            key = (int(r["dispatch_id"]), r["name"].split("<")[0].split("(")[0], int(r["duration"]))
            per.setdefault(key, collections.defaultdict(list))[r["counter_name"]].append(float(r["counter_value"]))
        disp = list(per.items())
        for si in range(len(disp) // 4):                       # 1 warm-up + 3 timed launches per shape
            grp = disp[si * 4 + 1: si * 4 + 4]
            d = res.setdefault(si, {"kernel": grp[0][0][1], "us": []})
            d["us"].append(sum(g[0][2] for g in grp) / 3e3)
            for c in grp[0][1]:
                vals = [(sum(g[1][c]) / len(g[1][c])) if c.endswith("_avr") else sum(g[1][c]) for g in grp]
                d[c] = sum(vals) / 3
    print("M = 434624, N = 1024 unless noted; bf16 NT; counters are per launch, summed over all instances")
    for si, d in res.items():
        us = sum(d["us"]) / len(d["us"])
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8
        print("\n%s  %s  %.1f us under the counters" % (SHAPES[si] if si < len(SHAPES) else si, d["kernel"], us))
        if cyc:
            print("  clock %.2f GHz   mfma_util %.3f   wait_share %.3f   TA_BUSY %.3f" % (
                cyc / us / 1e3, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc),
                d.get("SQ_WAIT_INST_ANY", 0) / max(1.0, d.get("SQ_WAVE_CYCLES", 1)), d.get("TA_BUSY_avr", 0) / cyc))
        for k in sorted(d):
            if k not in ("kernel", "us"):
                print("    %-32s %.4g" % (k, d[k]))


if __name__ == "__main__":
    main()
