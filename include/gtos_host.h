/* gtos_host.h -- C ABI of libgtos_host.so: the host-side graph -> relation-tensor path.
 *
 * Native (C++17, multi-threaded) counterpart of the reference's Python/networkx preprocessing that feeds the hot path:
 *   - bidirectional labelled graph + BFS node order/depth     generator/AMRGraph.py:76-98, translator/dependencyGraph.py:30-52
 *   - all-pairs shortest label paths                          generator/AMRGraph.py:100-115 (nx.all_shortest_paths),
 *                                                             translator/dependencyGraph.py:54-74 (nx.single_source_shortest_path)
 *   - relation type ids, relation bank, <CLS> row/column      generator/data.py:134-176 (train), :178-232 (eval),
 *                                                             translator/data.py:132-176
 * Integer work only: results are bit-exact with the reference wherever the reference is deterministic (the translator
 * flavour always; the generator flavour in eval mode); the generator's train-mode random.choice among alternative
 * shortest paths is replaced by an exactly-uniform choice driven by a splitmix64 stream.
 */
#ifndef GTOS_HOST_H
#define GTOS_HOST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gtos_relbatch gtos_relbatch;

#define GTOS_PATH_FIRST 0    /* one path per pair: BFS first discovery == nx.single_source_shortest_path (translator)        */
#define GTOS_PATH_UNIFORM 1  /* one path per pair, uniform among all shortest paths (generator train, data.py:150)         */
#define GTOS_PATH_ALL 2      /* every shortest path, networkx enumeration order (generator eval, data.py:199-213)          */

/* Builds the relation tensors of one batch.
 *   B graphs; graph g has n_nodes[g] nodes (ids 0..n-1), root roots[g] and the directed labelled edges
 *   e in [edge_off[g], edge_off[g+1]) given in INSERTION order and already doubled with their reverse-labelled twins
 *   (the reference inserts (src,des,rel) then (des,src,rel+'_reverse_'/'_r_')).  Labels are relation-vocabulary ids in
 *   [1,255].  A repeated (src,dst) overwrites the earlier label in place (networkx DiGraph semantics).
 *   ids: {pad, cls, rcls, self, tl} relation-vocabulary ids; max_len: paths longer than this collapse to <TL> (8).
 * Returns NULL on invalid input (disconnected graph, label out of range, ...); in the one-path-per-pair modes (GTOS_PATH_FIRST /
 * GTOS_PATH_UNIFORM) also for a graph of more than 32,767 nodes or more than 32,767 adjacency entries (its shortest-path DAG has at
 * most that many edges): the flat search shared with the GPU builder keeps node and DAG-edge ids in 16 bits.  GTOS_PATH_ALL has no
 * such limit.  AMR / dependency graphs have tens to a few hundred nodes. */
gtos_relbatch* gtos_relbatch_build(int B, const int* n_nodes, const int* roots, const int64_t* edge_off,
                                   const int* e_src, const int* e_dst, const int* e_label,
                                   int path_mode, uint64_t seed, const int* special_ids5, int max_len, int n_threads);

/* n = 1 + max nodes (the <CLS> node is index 0), R = number of distinct paths (bank rows), L = longest bank path,
 * K = alternatives per pair (1 unless GTOS_PATH_ALL). */
int gtos_relbatch_dims(const gtos_relbatch* h, int* n, int* R, int* L, int* K);

/* relation: int64 [n,n,B] (K==1) or [n,n,B,K]; relation[a][c][b] = type id of the path from node c to node a
 * (generator/data.py:164-165), 0-padded.  bank: int64 [L,R] 0-padded label ids; length: int64 [R];
 * order: int32 [B, n-1] node id at each BFS position (-1 padded); depth: int32 [B, n-1] BFS depth. */
int gtos_relbatch_export(const gtos_relbatch* h, int64_t* relation, int64_t* bank, int64_t* length,
                         int32_t* order, int32_t* depth);

void gtos_relbatch_free(gtos_relbatch* h);

/* The batch's graphs flattened for the GPU relation-batch builder (include/gtos_hip.h gtos_relbatch_dev_*): the ordered adjacency
 * (networkx insertion order, a repeated (src,dst) overwriting the label in place) and the BFS order / depth from the root that
 * gtos_relbatch_build uses itself (generator/AMRGraph.py:76-98, translator/dependencyGraph.py:30-52).  Inputs as gtos_relbatch_build.
 *   node_off [B+1]  prefix sums of n_nodes               adj_base [B+1]  first adjacency entry of graph g
 *   adj_off  [S+B]  per graph n+1 LOCAL offsets, graph g's block at node_off[g] + g      (S = sum of n_nodes)
 *   adj_dst, adj_lab [<= edge_off[B]]                     order, depth [S]  node id / BFS depth at each BFS position
 * Returns the number of adjacency entries, -1 on invalid input (as gtos_relbatch_build; also a graph of more than 32,767 nodes or
 * adjacency entries). */
int64_t gtos_relbatch_csr(int B, const int* n_nodes, const int* roots, const int64_t* edge_off, const int* e_src, const int* e_dst,
                          const int* e_label, int32_t* node_off, int32_t* adj_base, int32_t* adj_off, int32_t* adj_dst,
                          int32_t* adj_lab, int32_t* order, int32_t* depth);

/* ---- Prefix / suffix tries of a relation bank (gtos_amd/csrc_host/pathtrie.cpp).
 * Index preparation for the RelationEncoder of generator/encoder.py:66-119 on MI355X: the first bi-GRU layer runs once
 * per trie node instead of once per (sequence, position), the second layer's input-gate product splits into a
 * prefix-node and a suffix-node term.  bank: int64 [L,R] 0-padded label ids (relation_bank of generator/data.py:166-176,
 * time-major), length: int64 [R] (1..L).  chunk: rows per reduction chunk, 1..64 (gtos_segment_sum_rows
 * reads one row id per lane of a 64-lane wave).  NULL on invalid input (L > 64, a length outside 1..L, chunk outside 1..64). */
typedef struct gtos_pathtrie gtos_pathtrie;
gtos_pathtrie* gtos_pathtrie_build(int L, int64_t R, const int64_t* bank, const int64_t* length, int chunk);

/* sizes[9] = {L (longest sequence), R, N = sum(length), nodes of the prefix trie, of the suffix trie,
 *             prefix-trie row chunks, prefix-trie multi-chunk ("heavy") nodes, suffix-trie row chunks, heavy nodes}. */
int gtos_pathtrie_sizes(const gtos_pathtrie* h, int64_t* sizes);

/* Fills 25 caller-allocated int32 arrays (a NULL entry is skipped), in this order:
 *   batch_sizes[L]      sequences longer than t (rows of packed step t)
 *   seq_order[R]        packed position -> sequence id (length descending, then lexicographic)
 *   seq_pos[R]          sequence id -> packed position
 *   row_pf[N], row_sf[N] packed row (step t of the sequence at packed position m = offs[t] + m) -> prefix-trie node of
 *                       tokens 0..t / suffix-trie node of tokens t..len-1
 * then for the prefix trie and again for the suffix trie (10 arrays each):
 *   level_off[L+1]      nodes of level k (k+1 tokens) are level_off[k] .. level_off[k+1]-1, lexicographic within a level
 *   tok[n]              label id the node appends (prefix trie: last token; suffix trie: first token of the suffix)
 *   par[n]              parent node, n (one past the last node: the all-zero state row) at level 0
 *   child_off[2n]       [start, end) of the node's children, a contiguous node range
 *   rows[N]             packed rows sorted by node
 *   chunk_node/start/cnt/slot[chunks]  reduction chunks over `rows`: node, first entry, entries (<= chunk), and the heavy
 *                       slot of a node with several chunks (-1: single chunk)
 *   heavy_node[heavy]   node of every heavy slot.
 * Returns the number of arrays (25). */
int gtos_pathtrie_export(const gtos_pathtrie* h, int32_t** out);

/* What the backward walk of the trie GRU and the streaming segmented sum read besides the arrays above, per trie:
 *   sum_idx[n]            row of the [n + 1 + n_multi, .] gradient buffers holding the SUM over node u's children: the child
 *                         itself when there is one, row n (all zero) for a leaf, row n + 1 + j for the j-th node with several
 *   multi_ranges[2*n_multi]  the child ranges [start, end) of those nodes, in node order
 *   multi_level_off[L+1]  how many of them precede each level
 *   wave_off[n_waves+1]   chunk indices cutting the chunk list into ranges of about rows_per_wave rows (gtos_segment_sum_stream);
 *                         n_waves = max(1, ceil(N / rows_per_wave))
 * _derived_sizes: sizes[4] = {n_multi (prefix trie), n_waves, n_multi (suffix trie), n_waves}.  _export_derived fills the 8
 * caller-allocated int32 arrays (prefix trie's four, then the suffix trie's), returns 8. */
int gtos_pathtrie_derived_sizes(const gtos_pathtrie* h, int rows_per_wave, int64_t* sizes);
int gtos_pathtrie_export_derived(const gtos_pathtrie* h, int rows_per_wave, int32_t** out);
void gtos_pathtrie_free(gtos_pathtrie* h);

/* ---- Index preparation of the factored relation operand (gtos_amd/csrc_host/relindex.cpp): what the attention kernels
 * read instead of the dense relation.index_select(...).view(n,n,B,d) of generator/generator.py:79.
 * relation: int64 [n,n,B] type ids in [0,R) (relation[j][i][b] pairs query i with key j).  chunk: pairs per bank-gradient
 * chunk (32).  NULL on invalid input. */
typedef struct gtos_relindex gtos_relindex;
gtos_relindex* gtos_relindex_build(int n, int B, int64_t R, const int64_t* relation, int chunk);
/* sizes[3] = {P = n*n*B, chunks, heavy types} */
int gtos_relindex_sizes(const gtos_relindex* h, int64_t* sizes);
/* Fills 9 caller-allocated int32 arrays (NULL entries are skipped):
 *   idx_q[P]  ids in [i,b,j] order      idx_k[P]  ids in [j,b,i] order; bit 31 set on the ids of types that occur once in
 *             the batch (no chunk below: gtos_rel_attn_bwd writes their bank-gradient row itself)
 *   pair_sorted[P]  flat pair indices (j*n+i)*B+b grouped by type, graph-major inside a type
 *   chunk_type/start/count/slot[chunks]  gradient chunks over pair_sorted (slot = heavy slot, -1 for single-chunk types),
 *                   grouped by the XCD that walks them: the XCD of the pairs' graphs when they share one, else the XCD with the
 *                   fewest load rounds so far; inside an XCD the chunks of more than 8 pairs first (longest first), then
 *                   (graph, key row) order
 *   xcd_off[9]      chunk ranges per XCD      heavy_types[heavy]  type id of every heavy slot.  Returns 9. */
int gtos_relindex_export(const gtos_relindex* h, int32_t** out);
void gtos_relindex_free(gtos_relindex* h);

#ifdef __cplusplus
}
#endif
#endif
