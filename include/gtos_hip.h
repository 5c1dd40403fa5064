/* gtos_hip.h -- C ABI of libgtos_hip.so, the MI355X (gfx950) kernels behind the gtos graph-transformer hot path.
 *
 * The reference (jcyk/gtos) has no FFI: its hot path is plain PyTorch, every "kernel" being whatever ATen
 * dispatches for an op sequence.  The entry points below are therefore what a binding for THIS path would
 * bind: one entry per reference op sequence, taking raw device pointers, sizes, leading dimensions and a
 * hipStream_t (as void*).  No torch types cross this boundary.  Each declaration cites the reference lines
 * it replaces (paths relative to the reference repository root).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   dtype       GTOS_F32 (0) or GTOS_BF16 (1): storage type of activations; accumulation is always fp32.
 *   layouts     time-major [T,B,C] activations like the reference; a "row" is one (t,b) pair and rows are
 *               addressed as base + (t*B+b)*ld, so q/k/v may be views into one packed [T,B,3d] projection.
 *   stream      the HIP stream to enqueue on (the caller's current stream); nothing here synchronises.
 *   return      0 on success; >0 a hipError_t from the launch; <0 an argument error (shape the kernels do
 *               not support).  Nothing falls back to a CPU path.
 */
#ifndef GTOS_HIP_H
#define GTOS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTOS_F32 0
#define GTOS_BF16 1

int gtos_abi_version(void);

/* C[M,N] (+)= act(opA(A)[M,K] . opB(B)[K,N] + bias[N]), row-major, MFMA (bf16: 16x16x32, fp32: 16x16x4).
 * Supported (transA,transB): (0,1) Linear forward Y = X W^T; (0,0) dX = dY W; (1,0) dW = dY^T X.
 * relu / p_drop>0 fuse ReLU and dropout(seed) into the epilogue; accumulate adds into C; splitk>1 splits K
 * (needs out_dtype F32, accumulate=1, no bias/relu/dropout): with a workspace of >= splitk*M*N*4 bytes (16-byte
 * aligned, N % 4 == 0) the splits write partial tiles that a second kernel adds into C in a fixed order
 * (deterministic); without one they accumulate into C with fp32 atomics.
 * The shape picks the kernel (csrc/gemm.hip): 128x128 tiles by default; bf16 (0,1) with >= 512 macro tiles, K >= 1024,
 * K % 32 == 0, 256 <= N <= 2048 and bf16 (1,0) split-K with M, N >= 256 run on 256x256 tiles with four LDS stages.
 * Rows of A / B past M / N may be re-read (never stored): the operands must be readable up to their last valid row only.
 * Replaces F.linear / nn.Linear and their autograd mm's: generator/graph_transformer.py:61-63 (fc1, relu,
 * dropout, fc2), :106-122 (in_proj, relation_in_proj), :166 (out_proj), :176-197; generator/transformer.py:66-69,
 * :109-119,:162,:175-196; generator/encoder.py:117 and the nn.GRU gate products (:76-82). */
int gtos_gemm(int in_dtype, int out_dtype, int transA, int transB, int M, int N, int K,
              const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
              const float* bias, int relu, float p_drop, uint64_t seed, int accumulate, int splitk,
              void* workspace, int64_t workspace_bytes, void* stream);

/* Fused relation-aware attention forward: scores, both masks, softmax over keys, weight dropout, P.V.
 *   mode 0: no relation (MultiheadAttention, generator/transformer.py:120-162; q is NOT pre-scaled: scale is
 *           applied to the dot product, which equals the reference's q *= scaling);
 *   mode 1: rel = rarb[S,T,B,2d] = relation_in_proj(relation) (generator/graph_transformer.py:122-133, note the
 *           [key j][query i] order created by the transposes at :123-124);
 *   mode 2: rel = bank[R,2d] = relation_in_proj(relation_encoder output), idx_q[T,B,S] int32 type ids
 *           (idx_q[i,b,j] = relation[j,i,b]); fuses the index_select of generator/generator.py:79.
 * key_pad[S,B] / attn_mask[T,S] are uint8 (non-zero = masked, generator/graph_transformer.py:136-149).
 * Outputs o[T,B,d] (ldo), lse[T,B,H]; w (optional, [T,S,B,H] fp32) receives the post-dropout weights that
 * the reference returns with need_weights (generator/graph_transformer.py:168-172).
 * SHAPES (-10 outside): d % H == 0 like the reference (generator/graph_transformer.py:75), head width d/H a multiple of 8
 * (a lane owns 8 channels) and at most 512.  d and d/H powers of two with d <= 512 -- every configuration the reference ships
 * or BASELINE.json names (d = 256 / 512, H = 8) -- run on the fast lane map: a row of d channels over d/8 lanes of ONE 64-lane
 * wave, per-head reductions as DPP butterflies.  Any other shape runs on the generic map: a head takes pow2ceil(d/H/8) lanes
 * (the surplus lanes idle), a key row as many heads as fit 64 lanes, the remaining heads further slices on blockIdx.y.  The
 * host modules refuse shapes outside at construction (gtos_amd.ops.check_attention_shape).
 * Modes 1,2 need T == S (-12). */
int gtos_rel_attn_fwd(int dtype, int mode, int T, int S, int B, int H, int d,
                      const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const void* rel, const int* idx_q, const uint8_t* key_pad, const uint8_t* attn_mask,
                      float scale, float p_drop, uint64_t seed,
                      void* o, int64_t ldo, float* lse, float* w, void* stream);

/* Backward of the above (two launches: query-major then key-major).  Recomputes the probabilities from lse.
 * dw (optional) is an upstream gradient on the returned weights w (TokenGenerator's copy attention,
 * generator/decoder.py:32-34,55); then w must be the forward's output.  Mode 1 writes d_rel = d(rarb) [S,T,B,2d];
 * mode 2 leaves the bank gradient to gtos_rel_attn_bwd_bank -- except for types whose id carries bit 31 in idx_q / idx_k
 * (set by the host index for types that occur exactly once in the batch): their single pair's term IS their bank gradient
 * row and is written straight into d_rel[type] (row stride ld_drel; d_rel may be NULL when no id is flagged).
 * pd/gs are caller-provided fp32 scratch [T,S,B,H] (post-dropout probabilities, scale*dS). */
int gtos_rel_attn_bwd(int dtype, int mode, int T, int S, int B, int H, int d,
                      const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                      const void* rel, const int* idx_q, const int* idx_k,
                      const uint8_t* key_pad, const uint8_t* attn_mask,
                      float scale, float p_drop, uint64_t seed,
                      const void* o, int64_t ldo, const float* lse, const float* w,
                      const void* d_o, int64_t lddo, const float* dw,
                      void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                      void* d_rel, int64_t ld_drel, float* pd, float* gs, void* stream);

/* Mode-2 bank gradient: d_bank[R,2d] (dtype) = scatter-add over pairs of [gs*(k_j + RB[t]) | gs*(q_i + RA[t])],
 * i.e. the backward of generator/generator.py:79's index_select composed with the relation term of
 * graph_transformer.py:126-127.  Pairs (id = (j*n+i)*B+b) arrive sorted by type and cut into chunks; a type that
 * fits one chunk (chunk_slot = -1) stores its row directly, a type spanning several chunks accumulates with fp32
 * atomics into heavy[chunk_slot] ([n_heavy,2d], zeroed by the caller, who folds it back into d_bank).
 * xcd_off (optional, int[9]): the chunk list is grouped so that chunks [xcd_off[x], xcd_off[x+1]) gather q/k rows of
 * the graphs XCD x owns (graph b -> XCD b / (B/8), as in the attention kernels); workgroup ids == x (mod 8) walk them.
 * ld_dbank (>= 2d): row stride of d_bank -- the layers' gradients are written side by side into one [R, L*2d] slab so
 * that the bank's input gradient is ONE deep-K product over all layers (gtos_amd/ops.py GradAccumGroup). */
int gtos_rel_attn_bwd_bank(int dtype, int n, int B, int H, int d,
                           const void* q, int64_t ldq, const void* k, int64_t ldk,
                           const void* bank, const float* gs,
                           const int* pair_sorted, const int* chunk_type, const int* chunk_start,
                           const int* chunk_count, const int* chunk_slot, const int* xcd_off, int nchunks,
                           void* d_bank, int64_t ld_dbank, float* heavy, void* stream);

/* ---- Path tries of a relation bank built on the GPU (csrc/pathtrie_dev.hip; stages in csrc/trie_kernels.h): the device-side
 * counterpart of gtos_pathtrie_build in include/gtos_host.h (generator/encoder.py:66-119's index preparation), same arrays.
 * Two phases with one host read between them; gtos_amd/pathtrie_hip.py drives them.  common / pf / sf are tables of device
 * pointers in the order of the C_* / T_* enums of trie_kernels.h; sizes is int32[65] on the device (SZ_* / S_* there);
 * workspace = rocPRIM temporary storage of at least _workspace bytes.  Covered: paths of 1..8 labels with ids in [0, 255)
 * (sizes[0] != 0 after phase A otherwise).  Opt-in (see the file's header for the measured status). */
int gtos_pathtrie_dev_workspace(int64_t R, int64_t N, int64_t* bytes_out);
int gtos_pathtrie_dev_phase_a(int L, int64_t R, const int64_t* bank, const int64_t* length, void** common, void** pf, void** sf,
                              int32_t* sizes, void* workspace, size_t workspace_bytes, void* stream);
int gtos_pathtrie_dev_phase_b(int64_t R, int64_t N, int n_pf, int n_sf, int chunk, int rows_per_wave, void** common, void** pf,
                              void** sf, int32_t* sizes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- The relation tensors of a batch built on the GPU (csrc/relbatch_dev.hip; stages in csrc/relbatch_kernels.h): the device-side
 * counterpart of gtos_relbatch_build in include/gtos_host.h for the one-path-per-pair modes (GTOS_PATH_FIRST / GTOS_PATH_UNIFORM), i.e.
 * the relation section of batchify (generator/data.py:134-176, translator/data.py:132-176) over the all-pairs shortest label paths
 * (generator/AMRGraph.py:100-115, translator/dependencyGraph.py:54-74).  The graphs arrive flattened by gtos_relbatch_csr
 * (include/gtos_host.h).  Two phases with one host read between them; gtos_amd/relbatch_hip.py drives them.  geom = int64[16] host
 * integers (enum GE_*), tab = host table of device pointers (enum T_*), both in csrc/relbatch_kernels.h; workspace = rocPRIM temporary
 * storage of at least _workspace(pairs + 3) bytes.  After phase A sizes[0] = R (distinct paths), sizes[1] = L (the longest); phase B
 * takes R back and fills relation int64 [n,n,B] (zero-filled by the caller), bank int64 [8,R] (zero-filled; rows >= L stay zero),
 * length int64 [R].  Same type numbering (first-seen order) and the same uniform choice among alternative shortest paths (splitmix64
 * stream keyed by seed, graph, source, target) as the host builder.  Opt-in (see the file's header for the measured status). */
int gtos_relbatch_dev_workspace(int64_t total, int64_t* bytes_out);
int gtos_relbatch_dev_phase_a(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream);
int gtos_relbatch_dev_phase_b(const int64_t* geom, int64_t R, void** tab, void* workspace, size_t workspace_bytes, void* stream);
/* The eval-mode batches (GTOS_PATH_ALL: EVERY shortest path of every pair in networkx's enumeration order, relation int64 [n,n,B,K], type
 * 0 = <PAD> behind a pair's last alternative; generator/data.py:178-232): three phases, a host read after each of the first two.
 * _all_count: the searches and the number of paths per pair -> sizes[3] = paths in total T, sizes[4] = most of one pair K; geom[13], [14]
 * carry T and K into _all_keys (keys, key sort, distinct keys -> sizes[0..2] = R, L, N) and _all_fill (numbering, relation, bank). */
int gtos_relbatch_dev_all_count(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream);
int gtos_relbatch_dev_all_keys(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream);
int gtos_relbatch_dev_all_fill(const int64_t* geom, int64_t R, void** tab, void* workspace, size_t workspace_bytes, void* stream);

/* ---- The relation index of the factored attention operand built on the GPU (csrc/relindex_dev.hip; stages in
 * csrc/relindex_kernels.h): the device-side counterpart of gtos_relindex_build in include/gtos_host.h, same arrays (the index
 * preparation that stands in for the dense expansion of generator/generator.py:79).  Two phases with one host read between them;
 * gtos_amd/relindex_hip.py drives them.  geom = int64[6] host integers (enum GE_*), tab = host table of device pointers (enum T_*), both
 * in csrc/relindex_kernels.h; workspace = rocPRIM temporary storage of at least _workspace(max(P, R)) bytes.  After phase A sizes =
 * {error flag (a type id outside [0,R)), chunks, heavy types, -}; phase B takes the chunk count back.  Default chunk order of the host
 * builder only (GTOS_BANK_BALANCE on, GTOS_HEAVY_FIRST off).  Opt-in (see the file's header for the measured status). */
int gtos_relindex_dev_workspace(int64_t n, int64_t* bytes_out);
int gtos_relindex_dev_phase_a(const int64_t* geom, void** tab, void* workspace, size_t workspace_bytes, void* stream);
int gtos_relindex_dev_phase_b(const int64_t* geom, int64_t nchunks, void** tab, void* workspace, size_t workspace_bytes, void* stream);

/* y = LayerNorm(x + dropout(r)) * gamma + beta (r may be NULL), saving mean/rstd per row.
 * Replaces F.dropout + nn.LayerNorm(residual + x): generator/graph_transformer.py:57-58,64-65;
 * generator/transformer.py:57-58,63-64,70-71; generator/decoder.py:35-36; generator/generator.py:73,172. */
int gtos_ln_residual_fwd(int dtype, int rows, int d, const void* x, const void* r, float p_drop, uint64_t seed,
                         const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd, void* stream);
/* dx (residual branch), dr (dropout branch; pass NULL to skip), dgamma/dbeta accumulated with fp32 atomics. */
int gtos_ln_residual_bwd(int dtype, int rows, int d, const void* dy, const void* x, const void* r, float p_drop,
                         uint64_t seed, const float* gamma, const float* mean, const float* rstd,
                         void* dx, void* dr, float* dgamma, float* dbeta, void* stream);

/* The same pair with MIXED storage types (round 4): x_dtype = the residual stream (x, and dx), r_dtype = the sub-layer output (r, and
 * dr), y_dtype = the output (y, and dy).  Supported triples: all equal; (F32, BF16, F32) and (BF16, BF16, F32) = bf16 activations on
 * an fp32 residual stream -- the post-LN outputs of the reference's layers (same lines as above) reach |3-4|, where a bf16 store per
 * layer alone costs 1.6e-2 absolute; the stream is [rows, d], a few MB per layer.  y2_bf16 (optional): a bf16 copy of y, the next
 * projection's MFMA operand; dy2_bf16 (optional) its gradient, added to dy in registers (dy itself may then be NULL).  -21: triple not
 * supported. */
int gtos_ln_residual_fwd2(int x_dtype, int r_dtype, int y_dtype, int rows, int d, const void* x, const void* r, float p_drop,
                          uint64_t seed, const float* gamma, const float* beta, float eps, void* y, void* y2_bf16,
                          float* mean, float* rstd, void* stream);
int gtos_ln_residual_bwd2(int x_dtype, int r_dtype, int y_dtype, int rows, int d, const void* dy, const void* dy2_bf16,
                          const void* x, const void* r, float p_drop, uint64_t seed, const float* gamma, const float* mean,
                          const float* rstd, void* dx, void* dr, float* dgamma, float* dbeta, void* stream);

/* In place dh *= (h > 0)/(1-p): backward of relu + dropout (generator/graph_transformer.py:61-62) given the saved output. */
int gtos_relu_dropout_bwd(int dtype, int64_t n, void* dh, const void* h, float p_drop, void* stream);

/* out[N] += column sums of dy[rows,N] (bias gradients of every Linear above). */
int gtos_colsum(int dtype, int rows, int N, int64_t ld, const void* dy, float* out, void* stream);

/* One GRU step over the active (length-sorted) prefix of the relation sequences; gate order r,z,n as in
 * torch.nn.GRU, which RelationEncoder wraps (generator/encoder.py:76-82,101-105).  xg = x W_ih^T + b_ih and
 * hg = h W_hh^T + b_hh come from gtos_gemm.  Updates h in place, writes the layer output y (with the
 * inter-layer dropout of nn.GRU(dropout=...) when p_drop>0), saves h_prev and the gates [rows,4*hs]. */
int gtos_gru_cell_fwd(int dtype, int rows, int hs, const void* xg, const void* hg, void* h, void* y, int64_t ldy,
                      void* hprev_save, void* gates, float p_drop, uint64_t seed, int64_t drop_base, void* stream);
/* Fused forward GRU step, bf16 only, hs % 64 == 0 (gru_step.hip): the gate products run on MFMA and the cell is the
 * epilogue, so no [rows,3*hs] gate tensor round-trips through HBM.  The input gates come from ONE of
 *   x [rows,in_dim] (row stride ldx, in_dim % 8 == 0) with w_ih [3hs,in_dim], b_ih -- the input product is fused too;
 *   xg [rows,3hs] already holding x W_ih^T + b_ih (x == NULL, gf == NULL);
 *   gf / gb: two bf16 tables [*,3hs] gathered per row, xg[m] = gf[gf_idx[m]] + gb[gb_idx[m]] + b_ih -- the second GRU layer
 *   on the path tries, whose input product splits into a prefix-node and a suffix-node term (gtos_amd/gru.py).
 * h_in [*,hs] is read-only; row m enters with h_in[h_idx[m]] (h_idx == NULL: h_in[m]; the trie's parent state).  The new
 * state of row m is written to h_out[m] ([*,hs]) when m < n_out (the next step's h_in slot); otherwise the sequence is
 * finished and its state goes to h_fin + (fin_idx ? fin_idx[m] : m) * ld_fin -- a column block of the [R, 2hs] "final states"
 * matrix in BANK order, which replaces the reference's torch.cat of the two directions' h_n and the unsort of the packed
 * order (generator/encoder.py:104-110).  gates [rows,4hs] = r, z, n, hn as gtos_gru_cell_fwd saves them; y (optional, row stride ldy) receives
 * dropout(h_new) with the same counter layout. */
int gtos_gru_step_fwd(int rows, int hs, const void* x, int64_t ldx, int in_dim, const void* w_ih, const float* b_ih,
                      const void* xg, const void* gf, const int* gf_idx, const void* gb, const int* gb_idx,
                      const void* h_in, const int* h_idx, const void* w_hh, const float* b_hh,
                      void* h_out, int n_out, void* h_fin, int64_t ld_fin, const int* fin_idx, void* gates, void* y, int64_t ldy,
                      float p_drop, uint64_t seed, int64_t drop_base, void* stream);
/* Which kernel runs for x != NULL (both give the same bits): gru_step_fwd_a2w3_kernel (64-k stages, three slots of activation rows + two of
 * weight rows in flight, 256-row panels on eight waves) for launches of >= 8192 rows with in_dim % 64 == 0 and (in_dim + hs) / 64 a multiple of 6
 * -- both layers of every BASELINE config --, else the single-stage gru_step_fwd_kernel<1>; GTOS_GRU_FWD_A2W3=0: always the latter.
 * Round 6 removed the other forms of round 5 (three-slot ring of 32-k stages, two whole-stage slots, the hn-recompute flag `save_hn`). */

/* Fused backward GRU step, bf16 only, hs % 64 == 0 (gru_step.hip).  d4 [rows,4hs] = d r | d z | d n_x | d n_h in ONE
 * buffer (d(xg) = columns 0..3hs, d(hg) = columns 0..2hs and 3hs..4hs).  First adds d(hg) W_hh of the step processed
 * just before (d4_prev [rows_prev,4hs], may be NULL; w_hh_t = W_hh^T [hs,3hs]) to the running state gradient dh
 * [rows,hs] (dh_dtype: GTOS_F32 or GTOS_BF16; row stride ld_dh, so a column block of the [R,2hs] gradient of the final states
 * serves as the buffer) -- this replaces the per-step GEMM -- then runs the cell backward: dh += dropout-masked dy, writes d4,
 * leaves dh = dh_total * z.  The state row m entered the step with is hprev[hprev_idx[m]] (hprev_idx == NULL: hprev[m]).
 * bias_partials [n_partials,4hs] fp32 (optional, zeroed by the caller) accumulates the
 * column sums of d4 with atomics (sum over dim 0 = GRU bias gradients, as in gtos_gru_cell_bwd).  hprev_out (optional,
 * [rows,hs]): the entering state of every row written compactly -- the gathered operand of the recurrent weight gradient on
 * the tries.  sum_idx (optional, [rows]) with dh_src ([*,hs], dh's dtype): row m takes its recurrent operand row from
 * d4_prev[sum_idx[m]] and its incoming state gradient from dh_src[sum_idx[m]] instead of d4_prev[m] / dh[m] -- on the tries a parent
 * with ONE child reads the child's rows in place, a leaf (sum_idx[m] == zero_row) takes zeros without a fetch, and only parents
 * with several children need summed rows (gtos_segment_sum_ranges over those parents alone); dh[m] still receives dh_total * z. */
int gtos_gru_step_bwd(int rows, int hs, const void* d4_prev, int rows_prev, const void* w_hh_t,
                      const void* gates, const void* hprev, const int* hprev_idx, const void* dy, int64_t ldy, void* dh, int dh_dtype,
                      int64_t ld_dh, void* d4, float p_drop, uint64_t seed, int64_t drop_base, float* bias_partials, int n_partials,
                      void* hprev_out, const int* sum_idx, const void* dh_src, int zero_row, void* stream);

/* gtos_gru_step_bwd with a second role in the same launch (round 5; gtos_gru_step_bwd forwards here with w_ih_t = dinp = NULL).  With dinp,
 * additional workgroups turn the d4 rows of the step processed just before into THAT step's input gradient,
 *   dinp[rows_prev, n_in] (=/+=) mask * (d4_prev[:, 0:3hs] x W_ih)        w_ih_t = W_ih^T [n_in, 3hs], n_in % 64 == 0, row stride ld_dinp
 * -- the product nn.GRU's autograd runs as one [N,3hs] x [3hs,in] GEMM per layer and direction after BPTT
 * (generator/encoder.py:104 -> torch's GRU backward).  The tiles of a 128-row panel (hs/64 cell tiles, ceil(n_in/128) input-gradient tiles)
 * run back to back on one XCD, so the previous step's d4 rows are fetched from HBM once for both products.  dinp_accumulate: add to what
 * the other direction wrote.  p_in > 0: mask dinp with the dropout the forward applied to the layer's INPUT (counter in_drop_base + m * n_in +
 * column, m = row inside rows_prev): the label-embedding rows of layer 0.  rows == 0 with dinp: only the input gradient (the step
 * processed last has no successor launch).  sum_idx (the trie indirection) and dinp exclude each other. */
int gtos_gru_step_bwd_fused(int rows, int hs, const void* d4_prev, int rows_prev, const void* w_hh_t,
                            const void* gates, const void* hprev, const int* hprev_idx, const void* dy, int64_t ldy, void* dh, int dh_dtype,
                            int64_t ld_dh, void* d4, float p_drop, uint64_t seed, int64_t drop_base, float* bias_partials, int n_partials,
                            void* hprev_out, const int* sum_idx, const void* dh_src, int zero_row,
                            const void* w_ih_t, void* dinp, int64_t ld_dinp, int n_in, int dinp_accumulate, float p_in, uint64_t seed_in,
                            int64_t in_drop_base, void* stream);

/* Both weight gradients of one GRU layer and direction over all its packed rows in ONE grouped product (round 5; torch's GRU backward runs
 * dW_ih = d(xg)^T x and dW_hh = d(hg)^T h_prev as separate GEMMs): with d4 [rows,4hs] = d r | d z | d n_x | d n_h,
 *   dwih[3hs, in_valid] += d4[:, 0:3hs]^T x[:, 0:in_valid]          x [rows, in_dim] (row stride ldx; columns in_valid.. are zero padding)
 *   dwhh[3hs, hs]       += d4[:, {0:2hs, 3hs:4hs}]^T hprev           hprev [rows, hs] (row stride ldh)
 * as 256x256 tiles of d4^T [x | hprev] (only the tiles holding a needed element) on the split-K ping-pong kernel; d4, x and hprev are
 * each read from HBM once.  fp32 accumulation; partial tiles go through `workspace` (>= tiles * 256 KB; more = more K splits) and are added
 * in a fixed order (deterministic).  bf16, hs % 64 == 0, in_dim % 8 == 0, in_valid % 4 == 0. */
int gtos_gru_weight_grads(int rows, int hs, int in_dim, int in_valid, const void* d4, const void* x, int64_t ldx, const void* hprev, int64_t ldh,
                          float* dwih, int64_t ld_dwih, float* dwhh, int64_t ld_dwhh, void* workspace, int64_t workspace_bytes, void* stream);

/* Weight gradients of n independent SMALL linear layers in one launch (round 6; the reference's autograd runs one mm per nn.Linear:
 * generator/graph_transformer.py:61-63,106-122,166, transformer.py: the same per decoder layer).  All ten arguments after n are HOST arrays of n
 * entries.  For job j:  C[j] [M[j], N[j]] (fp32, row stride ldc[j]) += A[j]^T B[j]  with A[j] [K[j], M[j]] = dY and B[j] [K[j], N[j]] = X (bf16,
 * row strides lda[j], ldb[j]), and, where bias != NULL and bias[j] != NULL,  bias[j][M[j]] (fp32) += column sums of A[j].  One workgroup per
 * 256x256 tile of one job over the job's whole K on the ping-pong TN kernel (no split-K: the tile has one writer, so the result is
 * deterministic), then one batched column-sum launch (fp32 atomics across row blocks).  M, N % 8 == 0, leading dimensions % 8 (ldc % 4),
 * A / B / C 16-byte aligned; anything else returns -22 / -25 and launches nothing. */
int gtos_gemm_tn_batch(int n, const void* const* A, const int64_t* lda, const int* M, const void* const* B, const int64_t* ldb, const int* N,
                       const int* K, float* const* C, const int64_t* ldc, float* const* bias, void* stream);

/* Segmented row sums for the trie-evaluated RelationEncoder's backward (generator/encoder.py:93-111 runs every path
 * separately; here the gradient of a shared trie node is the sum over the rows that share it).  bf16 rows, fp32
 * accumulation, bf16 result.  _rows: chunk c adds the rows rows[chunk_start[c] .. +chunk_cnt[c]) of src into node
 * chunk_node[c]: straight into dst[node] when chunk_slot[c] < 0 (the node's only chunk), else with fp32 atomics into
 * heavy[chunk_slot[c], 0:width] (zeroed by the caller), which _finish rounds into dst[heavy_node[s]].  _ranges: dst[s] =
 * sum of the consecutive src rows ranges[2s] .. ranges[2s+1]-1 (zeros for an empty range).  width % 8 == 0, <= 1536.
 * _rows takes an optional second source / destination / heavy triple (src2, dst2, heavy2; NULL = none) reduced over the
 * same row lists in the same pass: the two GRU directions' gradients share them. */
int gtos_segment_sum_rows(int n_chunks, const int* rows, const int* chunk_node, const int* chunk_start, const int* chunk_cnt,
                          const int* chunk_slot, const void* src, const void* src2, int64_t ld_src, int width,
                          void* dst, void* dst2, int64_t ld_dst, float* heavy, float* heavy2, void* stream);
/* _stream: the same reduction for a chunk list that is a CSR (chunks in node order, their rows consecutive in `rows`, as
 * gtos_pathtrie_build emits them): wave_off[n_waves + 1] are chunk indices that cut the list into ranges of about equal row
 * counts, one range per 64-lane wave (gtos_amd.pathtrie.TrieSide.wave_off); total_rows = the length of `rows`.
 * width % 256 == 0, <= 1536. */
int gtos_segment_sum_stream(int n_chunks, int total_rows, const int* rows, const int* chunk_node, const int* chunk_start,
                            const int* chunk_cnt, const int* chunk_slot, const int* wave_off, int n_waves, const void* src,
                            int64_t ld_src, int width, void* dst, int64_t ld_dst, float* heavy, void* stream);
int gtos_segment_sum_finish(int n_heavy, const int* heavy_node, const float* heavy, int width, void* dst, int64_t ld_dst,
                            void* stream);
int gtos_segment_sum_ranges(int n_seg, const int* ranges, const void* src, int64_t ld_src, int width, void* dst, int64_t ld_dst,
                            void* stream);

/* Backward of one step: dh (fp32, in/out) carries the state gradient; writes d(xg), d(hg) [rows,3*hs].
 * bias_partials (optional, fp32 [n_partials, 4*hs], zeroed by the caller once per layer/direction): running column sums
 * of d(r), d(z), d(n_x), d(n_h) per launch block -- summed over dim 0 they are the GRU bias gradients
 * (b_ih: r,z,n_x; b_hh: r,z,n_h), so no extra pass over d(xg)/d(hg) is needed.  Requires 256 % (hs/8) == 0. */
int gtos_gru_cell_bwd(int dtype, int rows, int hs, const void* gates, const void* hprev, const void* dy, int64_t ldy,
                      float* dh, void* dxg, void* dhg, float p_drop, uint64_t seed, int64_t drop_base,
                      float* bias_partials, int n_partials, void* stream);

/* Relation lookup/aggregation: out[P,d] = mean over the K path ids of bank rows, row 0 zeroed, divisor
 * clamp(#non-zero ids, 1) (generator/generator.py:83-88, :60-65); zero_row0=0,K=1 is the train lookup (:79). */
int gtos_relation_gather_mean(int dtype, int64_t P, int K, int d, const void* bank, const int64_t* idx, int zero_row0,
                              void* out, void* stream);

/* Relation-label embedding rows for the packed GRU input: out[n, 0:dim_pad] = dropout(table[tok[n]]) zero-padded to
 * dim_pad (multiple of 8) in `dtype` (nn.Embedding + F.dropout, generator/encoder.py:99-100), and its backward
 * dtable[V,dim] += scatter of dout (the embedding's index_add backward); small tables (V*dim_pad*4 <= 60 KB) are accumulated per
 * block in LDS and, when the caller lends a workspace (optional; any size >= one table, more = more blocks), summed over the
 * blocks by a second launch instead of flushed with global atomics onto the same few thousand addresses. */
int gtos_embed_rows_fwd(int dtype, int64_t n, int dim, int dim_pad, const int64_t* tok, const float* table, void* out,
                        float p_drop, uint64_t seed, void* stream);
int gtos_embed_rows_bwd(int dtype, int64_t n, int V, int dim, int dim_pad, const int64_t* tok, const void* dout,
                        float* dtable, float p_drop, uint64_t seed, float* workspace, int64_t workspace_bytes, void* stream);

/* The RelationEncoder's packed, embedded input in one pass and without a host read (round 5): the reference sorts the paths by length,
 * packs them time-major and embeds the packed tokens (generator/encoder.py:93-100).  Here the sorted order and the step offsets arrive
 * with the batch: packed row n = (step t, sorted path m), offs[t] <= n < offs[t+1] (offs: device int32 [L+1], L <= 64), m = n - offs[t],
 * token = bank[t, order[m]] (bank int64 [L,R], order int32: sorted position -> bank column).  x[n, 0:dim_pad] = dropout(table[token])
 * zero-padded, counter n * dim_pad + column (as gtos_embed_rows_fwd); onehot (optional, bf16 [n_rows, vp], vp % 8 == 0, vp >= V):
 * row n = e_token, the operand that turns the embedding's index_add backward into a product; tokens (optional, int64 [n_rows]). */
int gtos_embed_packed_paths(int dtype, int L, int R, int64_t n_rows, const int64_t* bank, const int* order, const int* offs,
                            const float* table, int dim, int dim_pad, void* x, float p_drop, uint64_t seed, void* onehot, int vp,
                            int64_t* tokens, void* stream);

/* Elementwise pieces of TokenEncoder / CNNEncoder / Highway (generator/encoder.py:123-201), each one kernel per direction in place
 * of a dozen small ATen kernels; `dtype` rows are contiguous.
 *   highway: y [N,2D] = layer(x) (columns [0,D) = new_x, [D,2D) = gate), out = sigmoid(gate) * x + (1 - sigmoid(gate)) * relu(new_x)
 *            (encoder.py:141-149); _bwd writes dy [N,2D] and dx [N,D] (the direct gate * x path only).  D % 8 == 0.
 *   max_relu: y [N,L,F] -> out [N,F] = relu(max over L) (the char CNN's max over time, encoder.py:169-172), arg [N,F] uint8 = first
 *            maximising position; _bwd writes all of dy [N,L,F].  F % 8 == 0, L <= 255.
 *   token_row: out [N,Cp] = dropout([feat [N,Cc] | table[tok[n], 0:Ct] | zeros]) with Cp = Cc + Ct rounded up to a multiple of 8
 *            (torch.cat + nn.Embedding + F.dropout of encoder.py:196-199; table fp32 [V,Ct], tok int64 [N], Cc % 8 == 0; feat may be
 *            NULL with Cc == 0); _bwd: dfeat [N,Cc] = mask * dout[:, 0:Cc], dtable[tok[n]] += mask * dout[n, Cc:Cc+Ct] (fp32
 *            atomics; either output may be NULL). */
int gtos_highway_fwd(int dtype, int64_t N, int D, const void* y, const void* x, void* out, void* stream);
int gtos_highway_bwd(int dtype, int64_t N, int D, const void* y, const void* x, const void* dout, void* dy, void* dx, void* stream);
int gtos_max_relu_fwd(int dtype, int64_t N, int L, int F, const void* y, void* out, uint8_t* arg, void* stream);
int gtos_max_relu_bwd(int dtype, int64_t N, int L, int F, const void* out, const uint8_t* arg, const void* dout, void* dy, void* stream);
int gtos_token_row_fwd(int dtype, int64_t N, int Cc, int Ct, int Cp, const void* feat, const int64_t* tok, const float* table,
                       void* out, float p_drop, uint64_t seed, void* stream);
int gtos_token_row_bwd(int dtype, int64_t N, int Cc, int Ct, int Cp, const void* dout, const int64_t* tok, void* dfeat,
                       float* dtable, float p_drop, uint64_t seed, void* stream);

/* Flat-buffer optimizer: sum of squares (clip_grad_norm_, generator/train.py:151) and the Adam variant of
 * generator/adam.py:66-87 (no bias correction, decoupled weight decay), with the gradient averaging of
 * generator/train.py:74-79 (gscale = 1/world_size) and the clip coefficient folded in; optionally refreshes a
 * bf16 mirror of the parameters. */
int gtos_sqnorm(int64_t n, const float* g, float* out, void* stream);
int gtos_adam_step(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                   float eps, float weight_decay, float gscale, const float* sqnorm, float max_norm,
                   void* bf16_mirror, void* stream);
int gtos_cast_f32_to_bf16(int64_t n, const float* src, void* dst, void* stream);
/* Transposes of all 2-D bf16 weights of a flat buffer in one launch (the operands of dX = dY W as NT products; the reference gets
 * them from autograd's mm backward, generator/graph_transformer.py:106-122 etc.).  desc (device, int64 [n_mat,3]) = {offset in
 * elements, rows, cols} of every matrix in src; tile_start (device, int32 [n_mat]) = 32x32 tiles before matrix m; the [cols,rows]
 * transpose of matrix m is written at the same offset of dst. */
int gtos_transpose_batch_bf16(int n_mat, const int64_t* desc, const int* tile_start, int total_tiles, const void* src, void* dst,
                              void* stream);

/* The abnormal-loss rule and the learning-rate schedule of the training loop (generator/train.py:81-83,142-148) decided ON
 * THE DEVICE, so no host read of the loss sits between forward and backward.  loss: fp32 scalar; state: double[3] =
 * {loss_acm, batches_acm, discarded}; flag: fp32 scalar; ctl: fp32[2] = {learning rate of this step, skip}.
 *   phase 0: flag = batches_acm > warmup_steps && loss > 5 * loss_acm / batches_acm
 *   (data parallel: the caller MAX-all-reduces flag between the phases)
 *   phase 1: flag != 0 -> discarded += 1, ctl[1] = 1; else loss_acm += loss, batches_acm += 1, ctl = {lr(batches_acm), 0}
 *            with lr(s) = embed_dim^-0.5 * min(s^-0.5, s * warmup_steps^-1.5).
 * gtos_adam_step_ctl = gtos_adam_step with the learning rate read from ctl[0] and no update at all when ctl[1] != 0. */
int gtos_step_control(int phase, const float* loss, double* state, float* flag, int warmup_steps, int embed_dim,
                      float* ctl, void* stream);

/* hipGraph replay of a training step (gtos_amd.train.GraphedStep; no reference counterpart -- the reference launches every step from
 * Python): a replay freezes every kernel argument, dropout seeds included.  ``epoch`` is a device uint64 the captured step increments
 * once per replay; while it is registered, every kernel of this library that draws a dropout mask uses
 * seed + *epoch * 0x9E3779B97F4A7C15 instead of seed (forward and backward of an op read the same value within one replay).
 * NULL (the default) restores the plain seeds.  Synchronous (hipMemcpyToSymbol): call it outside a capture. */
int gtos_set_seed_epoch(const void* epoch);
int gtos_adam_step_ctl(int64_t n, float* p, const float* g, float* m, float* v, const float* ctl, float beta1, float beta2,
                       float eps, float weight_decay, float gscale, const float* sqnorm, float max_norm,
                       void* bf16_mirror, void* stream);

/* Fused generate/copy mixture of TokenGenerator (generator/decoder.py:40-63): vocabulary softmax, 2-way diverter softmax,
 * gen_gate * p_vocab extended by the per-graph copy ids, scatter_add of copy_gate * alignment weights at cp_seq, log(p + 1e-12).
 * logits [T,B,V] (row stride ld_logits) and div [T,B,2] share `dtype`; align fp32 [T,B,S] (head-max alignment weights),
 * cp_seq int64 [S,B], target int64 [T,B].
 * _nll_fwd (training): nll[t,b] = -log p(target) (0 where target == pad_idx) without materialising the distribution; saves
 * lse and p(target).  _nll_bwd: d_logits [T,B,V] (dtype), d_div [T,B,2] (dtype), d_align fp32 [T,B,S] from d_nll [T,B].
 * _ll_fwd (inference, work=True): the whole row ll[t,b, 0:tot_ext] fp32, tot_ext = 1 + max(cp_seq) >= V. */
int gtos_copy_nll_fwd(int dtype, int T, int B, int V, int S, const void* logits, int64_t ld_logits, const void* div,
                      const float* align, const int64_t* cp_seq, const int64_t* target, int64_t pad_idx,
                      float* nll, float* lse, float* p_tgt, void* stream);
int gtos_copy_nll_bwd(int dtype, int T, int B, int V, int S, const void* logits, int64_t ld_logits, const void* div,
                      const float* align, const int64_t* cp_seq, const int64_t* target, int64_t pad_idx,
                      const float* lse, const float* p_tgt, const float* d_nll, void* d_logits, void* d_div,
                      float* d_align, void* stream);
int gtos_copy_ll_fwd(int dtype, int T, int B, int V, int S, int tot_ext, const void* logits, int64_t ld_logits,
                     const void* div, const float* align, const int64_t* cp_seq, float* ll, void* stream);

#ifdef __cplusplus
}
#endif
#endif
