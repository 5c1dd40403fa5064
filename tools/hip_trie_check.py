"""The staged HIP trie builder (gtos_amd.pathtrie_hip / csrc/pathtrie_dev.hip) on the GPU against the host builder (csrc_host/
pathtrie.cpp): every array of both tries on a small random bank and on the C2 bank, then the time of a build.
python tools/hip_trie_check.py [out.json]   (a few seconds of GPU time)"""
import json
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import synth                                            # noqa: E402
from gtos_amd.pathtrie import build_path_trie                         # noqa: E402
from gtos_amd.pathtrie_hip import HipBackend, build_path_trie_staged  # noqa: E402


def differing(a, b, path=""):
    """names of the public attributes (recursively) that differ between two index objects"""
    bad = []
    for k in sorted(set(vars(a)) | set(vars(b))):
        if k.startswith("_"):
            continue
        x, y = vars(a).get(k), vars(b).get(k)
        if isinstance(x, torch.Tensor):
            if not (isinstance(y, torch.Tensor) and x.dtype == y.dtype and x.shape == y.shape and torch.equal(x.cpu(), y.cpu())):
                bad.append(path + k)
        elif hasattr(x, "__dict__") and not isinstance(x, (int, float, list, tuple)):
            bad += differing(x, y, path + k + ".")
        elif x != y:
            bad.append(path + k)
    return bad


def main():
    out = {"tool": "tools/hip_trie_check.py"}
    dev = torch.device("cuda", 0)
    try:
        g = torch.Generator().manual_seed(5)
        length = torch.randint(1, 9, (3000,), generator=g)
        bank = torch.randint(1, 7, (8, 3000), generator=g) * (torch.arange(8)[:, None] < length[None, :])
        host = build_path_trie(bank, length)
        hip = build_path_trie_staged(bank.to(dev), length.to(dev), HipBackend.shared())
        out["small_bank_differing"] = differing(host, hip)
        batch, st = synth.make_config_batch("C2", rank=0, B=64)
        bank, length = batch["relation_bank"], batch["relation_length"]
        t0 = time.perf_counter()
        host = build_path_trie(bank, length)
        out["host_build_s"] = round(time.perf_counter() - t0, 4)
        bank_d, length_d = bank.to(dev), length.to(dev)
        hip = build_path_trie_staged(bank_d, length_d, HipBackend.shared(), n_rows=host.N)
        out["c2"] = {"R": host.R, "N": host.N, "nodes_pf": host.pf.n_nodes, "nodes_sf": host.sf.n_nodes}
        out["c2_differing"] = differing(host, hip)
        times = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            build_path_trie_staged(bank_d, length_d, HipBackend.shared(), n_rows=host.N)
            torch.cuda.synchronize()
            times.append(round(1e3 * (time.perf_counter() - t0), 3))
        out["hip_build_ms"] = times
        out["ok"] = not out["small_bank_differing"] and not out["c2_differing"]
    except Exception:
        out["ok"] = False
        out["error"] = traceback.format_exc()
    line = json.dumps(out)
    print(line, flush=True)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        with open(sys.argv[1], "w") as f:
            f.write(line + "\n")
    return 0 if out.get("ok") else 1


if __name__ == "__main__":
    sys.exit(main())
