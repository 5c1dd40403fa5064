"""TEST INFRASTRUCTURE: builds oracle/trie_emul.cpp (the GPU trie builder's per-thread stages as serial host loops) into
oracle/_build/libtrie_emul.so.  Called by tests/test_pathtrie.py and by __graft_entry__.build(); nothing under gtos_amd/ uses it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "trie_emul.cpp")
HDR = os.path.join(ROOT, "gtos_amd", "csrc", "trie_kernels.h")
OUT = os.path.join(HERE, "_build", "libtrie_emul.so")


def build(force=False):
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        tmp = OUT + ".%d.tmp" % os.getpid()                  # xdist workers may build at once: write aside, then rename
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", tmp])
        os.replace(tmp, OUT)
    return OUT
