"""Fixture for the dropout-semantics A/B (tools/dropout_ab.py): the 2,169 dependency-parsed sentence pairs of the reference's
shipped development set, /root/reference/translator_data/dev.txt (4 lines per example: labels, heads, source tokens, target
tokens -- translator/extract.py:11-45), as JSON rows [dep, head, tok, tgt], gzip-compressed.  DATA only; run in the build
container (the reference never travels to the GPU box):  python tests/golden/make_golden_trees.py"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gtos_amd.data import read_dependency_file  # noqa: E402

trees = read_dependency_file("/root/reference/translator_data/dev.txt")
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dep_dev_trees.json.gz")
with gzip.open(out, "wt", encoding="utf8", compresslevel=9) as f:
    json.dump([[d, h, t, g] for d, h, t, g in trees], f, ensure_ascii=False, separators=(",", ":"))
print(len(trees), "trees ->", out, os.path.getsize(out), "bytes")
