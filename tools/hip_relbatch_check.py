"""The staged HIP builders of a batch's relation tensors and relation index (gtos_amd.relbatch_hip / relindex_hip) on the GPU against the
host builders at C2 size: every array, then the time of a build and of the whole device-side preparation (relations + index + tries).
python tools/hip_relbatch_check.py [out.json]   (a few seconds of GPU time)"""
import json
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import data, relbatch, synth                                # noqa: E402
from gtos_amd.pathtrie import build_path_trie                             # noqa: E402
from gtos_amd.relindex import build_relation_index                        # noqa: E402


def differing(a, b, path=""):
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


def timed(fn, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append(round(1e3 * (time.perf_counter() - t0), 3))
    return out


def main():
    from gtos_amd import pathtrie_hip, relbatch_hip, relindex_hip
    out = {"tool": "tools/hip_relbatch_check.py"}
    dev = torch.device("cuda", 0)
    try:
        vocabs = synth.synth_vocabs()
        items, graphs = synth.make_amr_items("C2", 64, first_graph=0, vocabs=vocabs)
        ids = data.relation_special_ids(vocabs['relation'])
        t0 = time.perf_counter()
        host = relbatch.build_relation_batch(graphs, ids, path_mode=relbatch.PATH_UNIFORM, seed=99, n_threads=2)
        out["host_relbatch_s"] = round(time.perf_counter() - t0, 4)
        t0 = time.perf_counter()
        csr = relbatch_hip.graphs_csr(graphs)
        out["host_csr_s"] = round(time.perf_counter() - t0, 4)

        def rel():
            return relbatch_hip.build_relation_batch_staged(None, ids, relbatch_hip.HipBackend.shared(), path_mode=relbatch.PATH_UNIFORM, seed=99,
                                                            device=dev, csr=csr)
        hip = rel()
        out["relbatch_differing"] = [k for k in ("relation", "relation_bank", "relation_length") if not torch.equal(host[k], hip[k].cpu())]
        R = host["relation_bank"].shape[1]
        t0 = time.perf_counter()
        host_idx = build_relation_index(host["relation"], R)
        out["host_relindex_s"] = round(time.perf_counter() - t0, 4)

        def idx():
            return relindex_hip.build_relation_index_staged(hip["relation"], R, relindex_hip.HipBackend.shared())
        out["relindex_differing"] = differing(host_idx, idx())
        t0 = time.perf_counter()
        host_trie = build_path_trie(host["relation_bank"], host["relation_length"])
        out["host_tries_s"] = round(time.perf_counter() - t0, 4)

        def tries():
            return pathtrie_hip.build_path_trie_staged(hip["relation_bank"], hip["relation_length"], pathtrie_hip.HipBackend.shared(),
                                                       n_rows=hip["relation_rows"])
        out["tries_differing"] = differing(host_trie, tries())
        # the eval-mode batch of the same graphs: every shortest path of a pair, relation [n,n,B,K]
        sub = graphs[:16]
        csr16 = relbatch_hip.graphs_csr(sub)
        t0 = time.perf_counter()
        host_all = relbatch.build_relation_batch(sub, ids, path_mode=relbatch.PATH_ALL, n_threads=2)
        out["host_relbatch_all_s_16_graphs"] = round(time.perf_counter() - t0, 4)

        def rel_all():
            return relbatch_hip.build_relation_batch_all_staged(None, ids, relbatch_hip.HipBackend.shared(), device=dev, csr=csr16)
        hip_all = rel_all()
        out["relbatch_all_differing"] = [k for k in ("relation", "relation_bank", "relation_length") if not torch.equal(host_all[k], hip_all[k].cpu())]
        out["relbatch_all_K"] = int(host_all["relation"].shape[3])
        out["hip_relbatch_all_ms"] = timed(rel_all, 3)
        out["c2"] = {"R": R, "N": hip["relation_rows"], "nchunks": host_idx.nchunks}
        out["hip_relbatch_ms"], out["hip_relindex_ms"], out["hip_tries_ms"] = timed(rel), timed(idx), timed(tries)
        out["hip_all_ms"] = timed(lambda: (rel(), idx(), tries()))
        out["ok"] = not (out["relbatch_differing"] or out["relindex_differing"] or out["tries_differing"] or out["relbatch_all_differing"])
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
