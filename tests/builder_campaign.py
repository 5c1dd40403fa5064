"""Randomised equality campaign of the three staged index builders (stage code run by the emulation libraries, under ascending /
descending / random thread orders) against the host builders: python tests/builder_campaign.py <seed> <seconds>.  Not collected by
pytest (long-running); the closing session of round 3 ran 2 x 900 s = 327,359 cases and, with the
every-shortest-path mode on 30 % of them, 2 x 500 s = 143,534 more, then 2 x 1,500 s = 222,031
with graphs of up to 150 nodes mixed in: all equal (693 k cases in total)."""
import sys, time
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, torch
from gtos_amd import relbatch
from gtos_amd.pathtrie import build_path_trie
from gtos_amd.pathtrie_hip import build_path_trie_staged
from gtos_amd.relbatch_hip import build_relation_batch_all_staged, build_relation_batch_staged
from gtos_amd.relindex import build_relation_index
from gtos_amd.relindex_hip import build_relation_index_staged
from test_pathtrie import _EmulBackend as TrieEmul, _same_object
from test_relbatch_dev import EmulBackend as RelEmul, IDS, _same, _random_graphs
from test_relindex_dev import EmulBackend as IdxEmul
t0=time.time(); n_ok=0
rng=np.random.RandomState(int(sys.argv[1]))
while time.time()-t0 < float(sys.argv[2]):
    order=int(rng.choice([0,1,int(rng.randint(2,10**6))]))
    rel_e, idx_e, trie_e = RelEmul(order), IdxEmul(order), TrieEmul(order)
    big = len(sys.argv) > 3 and rng.rand() < 0.2           # (a third argument: a fifth of the cases with graphs of up to 150 nodes)
    B=int(rng.randint(1,7)); nlo=int(rng.randint(1,20)); nhi=nlo+int(rng.randint(0,130 if big else 40)); extra=float(rng.choice([0.0,0.1,0.5,1.5]))
    labels=int(rng.choice([2,5,40,100]))
    graphs=_random_graphs(int(rng.randint(0,2**31-1)), B, nlo, nhi, extra, labels=labels, tree_only=bool(rng.rand()<0.2))
    mode=int(rng.choice([relbatch.PATH_FIRST, relbatch.PATH_UNIFORM])); seed=int(rng.randint(0,2**63-1))*2+int(rng.randint(0,2)); max_len=int(rng.choice([8,8,8,5,2,1]))
    host=relbatch.build_relation_batch(graphs, IDS, path_mode=mode, seed=seed, max_len=max_len, n_threads=int(rng.choice([1,2,3])))
    st=build_relation_batch_staged(graphs, IDS, rel_e, path_mode=mode, seed=seed, max_len=max_len)
    bad=_same(host, st)
    if rng.rand() < 0.3:                                   # the every-shortest-path mode (eval batches) on the same graphs
        host_all = relbatch.build_relation_batch(graphs, IDS, path_mode=relbatch.PATH_ALL, max_len=max_len, n_threads=1)
        bad += ["all:" + str(b) for b in _same(host_all, build_relation_batch_all_staged(graphs, IDS, rel_e, max_len=max_len))]
    R=host["relation_bank"].shape[1]; chunk=int(rng.choice([32,32,7,1,64]))
    ichunk=int(rng.choice([32,32,5,1,128]))
    bad+=_same_object(build_relation_index(host["relation"],R,chunk=ichunk), build_relation_index_staged(st["relation"],R,idx_e,chunk=ichunk))
    bad+=_same_object(build_path_trie(host["relation_bank"],host["relation_length"],chunk=chunk), build_path_trie_staged(st["relation_bank"],st["relation_length"],trie_e,chunk=chunk,n_rows=st["relation_rows"]))
    if bad:
        print("MISMATCH", bad, dict(order=order,B=B,nlo=nlo,nhi=nhi,extra=extra,labels=labels,mode=mode,seed=seed,max_len=max_len,chunk=chunk,ichunk=ichunk)); sys.exit(1)
    n_ok+=1
print("campaign ok:", n_ok, "cases in", round(time.time()-t0), "s")
