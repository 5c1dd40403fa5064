"""The staged index builders' stage code under AddressSanitizer: the emulation libraries compiled with -fsanitize=address, the whole
Python process under libasan (LD_PRELOAD), so that a stage reading or writing past a buffer the Python glue allocated (torch CPU
tensors: malloc'd, red-zoned by ASan) is reported -- the class of bug an equality test cannot see and a GPU would turn into silent
corruption.  Not collected by pytest.  Usage (from the repo root):

    mkdir -p /tmp/asan && for k in trie relbatch relindex; do g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=address \
        -fno-omit-frame-pointer oracle/${k}_emul.cpp -o /tmp/asan/lib${k}_emul.so; done
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/asan_campaign.py 150 big

The closing session of round 3 ran it (150 random cases in all three path modes, two chunk sizes, then C2-sized banks of 16 graphs): no
report.  The PRODUCT's host library passed the same way: libgtos_host.so compiled with -fsanitize=address,undefined (same three
sources, -O1 -g), ``relbatch.LIB_PATH`` pointed at it, the host test files (test_host_relbatch / test_pathtrie / test_relindex_dev /
test_relbatch_dev, 90 tests incl. the reference goldens and the C2-size banks) under libasan + libubsan: no report; and with -fsanitize=thread under libtsan (relation batches of all three modes with 4 threads, the
two-sided trie build, the relation index): no report.  (torch aligns CPU allocations to 64 bytes: an overrun of fewer bytes than the padding behind a buffer goes unseen.)"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, torch
import oracle.build_emul as be
# point the emulation backends at the ASan-instrumented libraries
be.build = lambda which="trie", force=False: "/tmp/asan/lib%s_emul.so" % which
from gtos_amd import relbatch, synth, data
from gtos_amd.pathtrie import build_path_trie
from gtos_amd.pathtrie_hip import build_path_trie_staged
from gtos_amd.relbatch_hip import build_relation_batch_all_staged, build_relation_batch_staged
from gtos_amd.relindex import build_relation_index
from gtos_amd.relindex_hip import build_relation_index_staged
from test_pathtrie import _EmulBackend as TrieEmul, _same_object
from test_relbatch_dev import EmulBackend as RelEmul, IDS, _same, _random_graphs
from test_relindex_dev import EmulBackend as IdxEmul
rel_e, idx_e, trie_e = RelEmul(), IdxEmul(), TrieEmul()
rng=np.random.RandomState(5)
n=0
for it in range(int(sys.argv[1])):
    B=int(rng.randint(1,6)); nlo=int(rng.randint(1,15)); nhi=nlo+int(rng.randint(0,30)); extra=float(rng.choice([0.0,0.1,0.5,1.5]))
    graphs=_random_graphs(int(rng.randint(0,2**31-1)), B, nlo, nhi, extra, labels=int(rng.choice([2,5,40])))
    mode=int(rng.choice([0,1])); max_len=int(rng.choice([8,5,2,1]))
    host=relbatch.build_relation_batch(graphs, IDS, path_mode=mode, seed=it, max_len=max_len, n_threads=1)
    st=build_relation_batch_staged(graphs, IDS, rel_e, path_mode=mode, seed=it, max_len=max_len)
    assert _same(host, st)==[]
    ha=relbatch.build_relation_batch(graphs, IDS, path_mode=relbatch.PATH_ALL, max_len=max_len, n_threads=1)
    assert _same(ha, build_relation_batch_all_staged(graphs, IDS, rel_e, max_len=max_len))==[]
    R=host["relation_bank"].shape[1]
    for chunk in (32, 3):
        assert _same_object(build_relation_index(host["relation"],R,chunk=chunk), build_relation_index_staged(st["relation"],R,idx_e,chunk=chunk))==[]
        assert _same_object(build_path_trie(host["relation_bank"],host["relation_length"],chunk=chunk), build_path_trie_staged(st["relation_bank"],st["relation_length"],trie_e,chunk=chunk,n_rows=st["relation_rows"]))==[]
    n+=1
print("asan campaign ok", n)
if len(sys.argv) > 2:
    batch,_=synth.make_config_batch("C2",rank=0,B=16)
    R=batch["relation_bank"].shape[1]
    assert _same_object(build_relation_index(batch["relation"],R), build_relation_index_staged(batch["relation"],R,idx_e))==[]
    assert _same_object(build_path_trie(batch["relation_bank"],batch["relation_length"]), build_path_trie_staged(batch["relation_bank"],batch["relation_length"],trie_e))==[]
    vocabs = synth.synth_vocabs()
    items, graphs = synth.make_amr_items("C2", 16, first_graph=0, vocabs=vocabs)
    ids = data.relation_special_ids(vocabs['relation'])
    host=relbatch.build_relation_batch(graphs, ids, path_mode=1, seed=3, n_threads=2)
    assert _same(host, build_relation_batch_staged(graphs, ids, rel_e, path_mode=1, seed=3))==[]
    print("asan C2-size (16 graphs) ok")
