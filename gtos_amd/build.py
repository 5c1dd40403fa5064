"""Builds libgtos_hip.so (gfx950) in-tree with hipcc.  No torch in the build: the library is plain C ABI."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libgtos_hip.so")
SOURCES = ["gemm.hip", "rel_attn.hip", "rowops.hip", "gru_step.hip", "copy_nll.hip", "tokenenc.hip", "pathtrie_dev.hip", "relbatch_dev.hip", "relindex_dev.hip"]
EXACT_FP = {"relbatch_dev.hip"}      # double comparisons that must branch like the host builder: no fast-math, no contraction
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(obj, fallback):
    """Headers the object was compiled from, read from the depfile its last compilation wrote (-MMD); without one, the header the
    source is known to include.  Only headers inside the repository are tracked (the ROCm headers do not change under us)."""
    dfile = obj[:-2] + ".d"
    root = os.path.dirname(os.path.dirname(CSRC))
    try:
        words = open(dfile).read().replace("\\\n", " ").split()
    except OSError:
        return [fallback]
    deps = [w for w in words[1:] if w.endswith((".h", ".hpp")) and os.path.abspath(w).startswith(root) and os.path.exists(w)]
    return deps or [fallback]


HOST_SRCS = [os.path.join(os.path.dirname(CSRC), "csrc_host", f) for f in ("relbatch.cpp", "pathtrie.cpp", "relindex.cpp")]
HOST_LIB = os.path.join(os.path.dirname(CSRC), "csrc_host", "libgtos_host.so")
CXX = os.environ.get("CXX", "g++")


def build_host(force=False, verbose=True):
    """libgtos_host.so: the C++ graph -> relation-tensor path (include/gtos_host.h); plain g++, no GPU code."""
    hdr = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "gtos_host.h")
    shared = [os.path.join(CSRC, h) for h in ("relbatch_kernels.h", "relindex_kernels.h", "trie_kernels.h")]   # stage code the host and GPU builders share
    if force or _stale(HOST_LIB, HOST_SRCS + [hdr] + shared):
        cmd = [CXX, "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread"] + HOST_SRCS + ["-o", HOST_LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_LIB


def build(force=False, verbose=True):
    build_host(force, verbose)
    own_hdr = {"pathtrie_dev.hip": "trie_kernels.h", "relbatch_dev.hip": "relbatch_kernels.h",
               "relindex_dev.hip": "relindex_kernels.h"}     # the others include common.h
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + _deps(o, os.path.join(CSRC, own_hdr.get(src, "common.h")))):
            flags = [f for f in FLAGS if f not in ("-ffast-math", "-fno-finite-math-only")] + ["-ffp-contract=off"] if src in EXACT_FP else FLAGS
            cmd = [HIPCC] + flags + ["-MMD", "-MF", o[:-2] + ".d", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
