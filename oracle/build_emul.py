"""TEST INFRASTRUCTURE: builds the host emulations of the GPU index builders -- oracle/trie_emul.cpp (csrc/trie_kernels.h) and
oracle/relbatch_emul.cpp (csrc/relbatch_kernels.h), oracle/relindex_emul.cpp (csrc/relindex_kernels.h): the per-thread stage code of the HIP kernels as serial host loops -- into
oracle/_build/.  Called by the tests and by __graft_entry__.build(); nothing under gtos_amd/ uses it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TARGETS = {"trie": ("trie_emul.cpp", "trie_kernels.h", "libtrie_emul.so"),
           "relbatch": ("relbatch_emul.cpp", "relbatch_kernels.h", "librelbatch_emul.so"),
           "relindex": ("relindex_emul.cpp", "relindex_kernels.h", "librelindex_emul.so")}


def build(which="trie", force=False):
    src, hdr, out = TARGETS[which]
    src, hdr, out = os.path.join(HERE, src), os.path.join(ROOT, "gtos_amd", "csrc", hdr), os.path.join(HERE, "_build", out)
    deps = [src, hdr, os.path.join(HERE, "emul_order.h")]
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = out + ".%d.tmp" % os.getpid()                  # xdist workers may build at once: write aside, then rename
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", tmp])
        os.replace(tmp, out)
    return out


def build_all(force=False):
    return [build(k, force) for k in TARGETS]
