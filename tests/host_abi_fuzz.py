"""Invalid-input fuzz of the host C ABI (libgtos_host.so: relation batch in all path modes, flattened graphs, path tries, relation
index) meant to run under AddressSanitizer / UBSan (see tests/asan_campaign.py for the recipe; relbatch.LIB_PATH is pointed at the
instrumented build): every call either returns arrays or is refused with ValueError -- never a stray write.  It found one: the
flattened-graphs export sized its outputs from node counts it had not validated yet (fixed).  Not collected by pytest."""
import sys
import os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from gtos_amd import relbatch
if os.path.exists("/tmp/asan/libgtos_host.so"):
    relbatch.LIB_PATH = "/tmp/asan/libgtos_host.so"
import numpy as np, torch
from gtos_amd.pathtrie import build_path_trie
from gtos_amd.relindex import build_relation_index
from gtos_amd.relbatch_hip import graphs_csr
rng=np.random.RandomState(1)
ok=rej=0
for it in range(3000):
    B=int(rng.randint(1,4)); graphs=[]
    for _ in range(B):
        n=int(rng.randint(-1,8)); E=int(rng.randint(0,12))
        edges=np.stack([rng.randint(-2,n+3,size=E), rng.randint(-2,n+3,size=E), rng.randint(-3,300,size=E)],1).astype(np.int32) if E else np.zeros((0,3),np.int32)
        graphs.append((n, int(rng.randint(-1,n+2)) if n>0 else 0, edges))
    for mode in (0,1,2):
        try:
            relbatch.build_relation_batch(graphs, (0,2,3,4,5), path_mode=mode, seed=it, max_len=int(rng.choice([8,1,0,9])), n_threads=int(rng.choice([0,1,3])))
            ok+=1
        except (ValueError, AssertionError):
            rej+=1
    try:
        graphs_csr(graphs); ok+=1
    except ValueError:
        rej+=1
    # tries / index with invalid input
    L=int(rng.randint(1,10)); R=int(rng.randint(1,20))
    bank=torch.from_numpy(rng.randint(-1,300,size=(L,R)).astype(np.int64)); length=torch.from_numpy(rng.randint(-1,L+3,size=R).astype(np.int64))
    try:
        build_path_trie(bank,length,chunk=int(rng.choice([1,8,64,65,0]))); ok+=1
    except ValueError:
        rej+=1
    n=int(rng.randint(1,5)); Bb=int(rng.randint(1,4)); RR=int(rng.randint(1,6))
    rel=torch.from_numpy(rng.randint(-1,RR+2,size=(n,n,Bb)).astype(np.int64))
    try:
        build_relation_index(rel, RR, chunk=int(rng.choice([1,4,32]))); ok+=1
    except ValueError:
        rej+=1
print("host ABI fuzz under ASan/UBSan: accepted", ok, "rejected", rej)
