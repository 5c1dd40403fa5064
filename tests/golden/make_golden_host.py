"""Golden vectors for the host graph->relation-tensor path (libgtos_host.so), produced by the REFERENCE's own code:

  host_dep_dev.npz     translator flavour on the real data shipped with the reference (translator_data/dev.txt, first 48
                       trees): translator/extract.py builds the vocabularies, translator/dependencyGraph.py +
                       translator/data.py:batchify build relation / relation_bank / relation_length / concept_depth.
  host_amr_smatch.npz  generator flavour, eval mode (all shortest paths), on the six AMRs of generator/smatch/test_input{1,2}.txt plus two
                       hand-written re-entrant graphs (pairs with several shortest paths):
                       generator/AMRGraph.py + generator/data.py:batchify(train=False).

Run in the build container only:  python tests/golden/make_golden_host.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
np.int = int


def dep_case():
    tdir = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(REF, "translator", "extract.py"), "--train_data",
                           os.path.join(REF, "translator_data", "dev.txt")], cwd=tdir, stdout=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(REF, "translator"))
    import data as rdata
    from extract import IO, LexicalMap
    from dependencyGraph import dependencyGraph
    V = rdata.Vocab
    vocabs = {'concept': V(tdir + '/concept_vocab', 5, [rdata.CLS]), 'token': V(tdir + '/token_vocab', 5, [rdata.STR, rdata.END]),
              'predictable_token': V(tdir + '/predictable_token_vocab', 5, [rdata.END]),
              'token_char': V(tdir + '/token_char_vocab', 100, [rdata.STR, rdata.END]),
              'concept_char': V(tdir + '/concept_char_vocab', 100, [rdata.STR, rdata.END]),
              'relation': V(tdir + '/relation_vocab', 5, [rdata.CLS, rdata.rCLS, rdata.SEL, rdata.TL])}
    rv = vocabs['relation']
    recs = []
    for k, rec in enumerate(IO.read1(os.path.join(REF, "translator_data", "dev.txt"))):
        if k >= 48:
            break
        recs.append(rec)
    lex = LexicalMap()
    out = {}
    for bi, chunk in enumerate([recs[:24], recs[24:]]):
        items, heads_all, dep_all, rev_all, off = [], [], [], [], [0]
        for dep, head, tok, tgt in chunk:
            g = dependencyGraph(dep, head, tok, tgt)
            concept, depth, relation, ok = g.collect_concepts_and_relations()
            assert ok
            cp_seq, t2i, i2t = lex.get(concept, vocabs['predictable_token'])
            items.append({'concept': concept, 'depth': depth, 'relation': relation, 'token': tgt, 'cp_seq': cp_seq,
                          'token2idx': t2i, 'idx2token': i2t})
            heads_all += head
            dep_all += rv.token2idx(dep)
            rev_all += rv.token2idx([r + '_r_' for r in dep])
            off.append(len(heads_all))
        batch = rdata.batchify(items, vocabs)
        p = "b%d/" % bi
        out[p + "heads"] = np.array(heads_all)
        out[p + "dep_ids"] = np.array(dep_all)
        out[p + "rev_ids"] = np.array(rev_all)
        out[p + "off"] = np.array(off)
        out[p + "relation"] = np.asarray(batch['relation'])
        out[p + "relation_bank"] = np.asarray(batch['relation_bank'])
        out[p + "relation_length"] = np.asarray(batch['relation_length'])
        out[p + "concept_depth"] = np.asarray(batch['concept_depth'])
    out["special_ids"] = np.array([rv.padding_idx, rv.token2idx(rdata.CLS), rv.token2idx(rdata.rCLS), rv.token2idx(rdata.SEL),
                                   rv.token2idx(rdata.TL)])
    np.savez_compressed(os.path.join(HERE, "host_dep_dev.npz"), **out)
    print("host_dep_dev.npz", {k: v.shape for k, v in out.items() if k.startswith("b0/")})
    for m in ("data", "extract", "dependencyGraph"):
        sys.modules.pop(m, None)
    sys.path.pop(0)


def amr_case():
    sys.path.insert(0, os.path.join(REF, "generator"))
    import data as rdata
    from smatch import AMR
    from AMRGraph import AMRGraph
    graphs = []
    for fn in ("test_input1.txt", "test_input2.txt"):
        with open(os.path.join(REF, "generator", "smatch", fn), encoding="utf8") as f:
            while True:
                line = AMR.get_amr_line(f)
                if not line:
                    break
                graphs.append(AMRGraph(AMR.parse_AMR_line(line)))
    # two hand-written graphs with re-entrancies, so that pairs with SEVERAL shortest paths occur (K > 1 in eval batches)
    for line in ("(a / alpha :ARG0 (b / beta :ARG0 (d / delta)) :ARG1 (c / gamma :ARG0 d))",
                 "(r / root-01 :ARG0 (x / xx :mod (y / yy :ARG1 (z / zz))) :ARG1 (u / uu :mod y :ARG2 z) :ARG2 (v / vv :ARG0 x :ARG1 u))"):
        graphs.append(AMRGraph(AMR.parse_AMR_line(line)))
    labels = sorted({d['label'] for g in graphs for _, _, d in g.graph.edges(data=True)})
    tdir = tempfile.mkdtemp()

    def wv(name, toks):
        with open(os.path.join(tdir, name), "w") as fo:
            for t in toks:
                fo.write("%s\t1000\n" % t)
    concepts = sorted({c for g in graphs for c in g.name2concept.values()})
    wv("relation_vocab", labels)
    wv("concept_vocab", concepts)
    wv("token_vocab", ["a", "b"])
    wv("predictable_token_vocab", ["a", "b"])
    wv("concept_char_vocab", sorted({ch for c in concepts for ch in c}))
    wv("token_char_vocab", ["a", "b"])
    V = rdata.Vocab
    vocabs = {'concept': V(tdir + '/concept_vocab', 5, [rdata.CLS]), 'token': V(tdir + '/token_vocab', 5, [rdata.STR, rdata.END]),
              'predictable_token': V(tdir + '/predictable_token_vocab', 5, [rdata.END]),
              'token_char': V(tdir + '/token_char_vocab', 100, [rdata.STR, rdata.END]),
              'concept_char': V(tdir + '/concept_char_vocab', 100, [rdata.STR, rdata.END]),
              'relation': V(tdir + '/relation_vocab', 5, [rdata.CLS, rdata.rCLS, rdata.SEL, rdata.TL])}
    rv = vocabs['relation']
    items, n_nodes, roots, edges, eoff = [], [], [], [], [0]
    for g in graphs:
        concept, depth, relation, ok = g.collect_concepts_and_relations()
        assert ok
        item = {'concept': concept, 'depth': depth, 'relation': relation, 'token': ["a"], 'cp_seq': concept,
                'token2idx': {}, 'idx2token': {}, 'abstract': {}}
        items.append(json.loads(json.dumps(item)))        # the loader indexes relation[str(i)][str(j)] (data.py:149)
        names = list(g.graph.nodes)
        nid = {nm: k for k, nm in enumerate(names)}
        n_nodes.append(len(names))
        roots.append(nid[g.root])
        for u in names:                                    # adjacency in networkx (insertion) order
            for v, d in g.graph[u].items():
                edges.append((nid[u], nid[v], rv.token2idx(d['label'])))
        eoff.append(len(edges))
    batch = rdata.batchify(items, vocabs, train=False)
    out = dict(n_nodes=np.array(n_nodes), roots=np.array(roots), edges=np.array(edges), edge_off=np.array(eoff),
               relation=batch['relation'].numpy(), relation_bank=batch['relation_bank'].numpy(),
               relation_length=batch['relation_length'].numpy(), concept_depth=batch['concept_depth'].numpy(),
               special_ids=np.array([rv.padding_idx, rv.token2idx(rdata.CLS), rv.token2idx(rdata.rCLS), rv.token2idx(rdata.SEL),
                                     rv.token2idx(rdata.TL)]))
    np.savez_compressed(os.path.join(HERE, "host_amr_smatch.npz"), **out)
    print("host_amr_smatch.npz", {k: v.shape for k, v in out.items()})
    # the items exactly as the generator's DataLoader hands them to batchify (concept / depth / relation path lists, as the
    # preprocessed JSON stores them) + the relation vocabulary file: input of gtos_amd.data.batchify_amr
    with open(os.path.join(HERE, "host_amr_smatch_items.json"), "w") as fo:
        json.dump({"items": [{k: it[k] for k in ("concept", "depth", "relation")} for it in items],
                   "relation_vocab": open(os.path.join(tdir, "relation_vocab")).read()}, fo)
    print("host_amr_smatch_items.json %.1f KB" % (os.path.getsize(os.path.join(HERE, "host_amr_smatch_items.json")) / 1024))


if __name__ == "__main__":
    dep_case()
    amr_case()
