#!/usr/bin/env python
"""gemm_w4_nt_kernel keeps its accumulators in a0..a255 behind the compiler's back (gtos_amd/csrc/gemm_w4_gen.h).  That is sound only
while hipcc itself never touches an accumulation register in that kernel and never spills (a scratch access would also break the
kernel's hand-counted vmcnt waits).  This script compiles gemm.hip to assembly and checks both on the kernel's text.

    python tools/check_gemm_w4_isa.py      # exit code 0 = sound
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_text(asm, name="gemm_w4_nt_kernel"):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % name, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def check(lines):
    bad, in_asm, n_mfma, done = [], False, 0, False
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if in_asm:
            n_mfma += t.startswith("v_mfma")
            done = done or t.startswith("s_nop 15")          # the epilogue's marker: the counted region is over
            continue
        body = t.split(";")[0]
        if re.search(r"\ba\[?\d+", body) or "accvgpr" in body:
            bad.append("compiler-issued accumulation-register use: " + t)
        if "scratch_" in body:
            bad.append("scratch access: " + t)
        if body.startswith("v_mfma"):
            bad.append("compiler-issued MFMA: " + t)
        if re.match(r"(global|buffer|flat)_(load|store|atomic)|ds_", body) and n_mfma and not done:
            bad.append("compiler-issued memory instruction inside the hand-counted region: " + t)
        if "s_endpgm" in body or (n_mfma and body.startswith("s_nop 15")):
            done = True
    return bad, n_mfma


def main():
    src = os.path.join(ROOT, "gtos_amd", "csrc", "gemm.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only",
                        "-S", "--cuda-device-only", "-o", out, src], check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
        lines = kernel_text(open(out).read())
    bad, n_mfma = check(lines)
    print("gemm_w4_nt_kernel: %d lines, %d MFMAs in inline assembly, %d findings" % (len(lines), n_mfma, len(bad)))
    for b in bad[:20]:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
