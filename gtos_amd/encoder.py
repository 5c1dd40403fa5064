"""RelationEncoder / TokenEncoder / CNNEncoder / Highway on the gfx950 kernels.

Drop-in for /root/reference/generator/encoder.py (same signatures and state_dict keys, including nn.GRU's
native parameter names ``rnn.weight_ih_l0`` ...).  The pretrained-embedding file loader is out of scope.
"""
import os

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from . import gru as _gru
from .gru import bigru_final, trie_bigru_final, packed_path_gru, PackPlan
from .pathtrie import build_path_trie
from .transformer import Embedding

# A bank that arrives WITHOUT its tries (a caller feeding the reference's own batches): 1 = build them with torch ops on the device
# (gtos_amd.pathtrie_device: ~12 ms of GPU time at C2) instead of the host round trip below (bank to the host, ~0.1-0.2 s of C++,
# tries back).  Off by default: round 3 ran that builder on the GPU only through the loader's Prefetcher at C2.
# "hip": the staged HIP builder (gtos_amd.pathtrie_hip) instead of the torch ops.
TRIE_DEVICE = {"1": "torch", "torch": "torch", "hip": "hip"}.get(os.environ.get("GTOS_TRIE_DEVICE", "0"), "")
# Training-mode dropout of the RelationEncoder (round 4).  The reference draws its two masks -- on the label embeddings and between the
# GRU layers (encoder.py:91-92,105) -- independently per (path, position, channel).  "path" = exactly that distribution (counter-based
# hash of (row, channel) instead of ATen's Philox stream, like every other dropout here): nothing is shared between paths, so the
# encoder runs one row per (path, position) like the reference's packed sequence.  "node" = the masks are drawn per TRIE NODE (every
# path still sees independent Bernoulli(1-p) masks at each of its positions -- the reference's per-path marginals -- but two paths with
# a common prefix / suffix share them on the common part), which is what lets layer 0 run once per trie node and layer 1 gather its
# input gates from per-node tables under dropout: RelationEncoder forward 23.9 -> 10 ms, backward 39.6 -> 20 ms at C2.  It is a
# different regulariser (same function at p = 0 and in eval mode, where the trie evaluation is always used): opt-in.
# Default "path" = the reference's function; GTOS_RELENC_MASKS=node or ``set_relation_mask_sharing(model, "node")``.
MASK_SHARING = os.environ.get("GTOS_RELENC_MASKS", "path")
# Round 5: the per-(path, position) evaluation on gtos_amd.gru.PackedPathGRUFn (bf16, two layers); False = round 4's BiGRUFinalFn behind
# sort / cat / index_select (86.3-86.6 vs 81.3-81.8 ms per step; kept: the fp32 parity path is that function).
PACKED = True        # (module constant; the parity tests compare with round 4's per-row function through monkeypatch)
assert MASK_SHARING in ("path", "node"), MASK_SHARING


def set_relation_mask_sharing(module, mode):
    """"path" (reference semantics) or "node" (trie-shared masks, faster) for every RelationEncoder inside ``module``."""
    assert mode in ("path", "node"), mode
    for m in module.modules():
        if isinstance(m, RelationEncoder):
            m.mask_sharing = mode
    return module


def AMREmbedding(vocab, embedding_dim, pretrained_file=None, amr=False, dump_file=None):
    if pretrained_file is not None:
        raise NotImplementedError("pretrained embedding files are outside the hot path (encoder.py:13-64)")
    return Embedding(vocab.size, embedding_dim, vocab.padding_idx)


class RelationEncoder(nn.Module):
    """Label paths -> relation vectors: embedding -> dropout -> 2-layer bi-GRU (final states) -> Linear
    (encoder.py:66-119).  Sequences are sorted by length and packed (the reference does the same for cuDNN);
    the GRU itself is gtos_amd.gru (MFMA GEMMs + fused cell kernel)."""

    def __init__(self, vocab, rel_dim, embed_dim, hidden_size, num_layers, dropout, bidirectional=True):
        super().__init__()
        assert bidirectional, "only the bidirectional encoder is ever built (generator.py:29)"
        self.vocab, self.embed_dim, self.hidden_size = vocab, embed_dim, hidden_size
        self.num_layers, self.dropout, self.bidirectional = num_layers, dropout, bidirectional
        self.rel_embed = AMREmbedding(vocab, rel_dim)
        self.rnn = nn.GRU(input_size=rel_dim, hidden_size=hidden_size, num_layers=num_layers,
                          dropout=self.dropout if num_layers > 1 else 0., bidirectional=True)   # parameter container
        self.out_proj = nn.Linear(2 * hidden_size, embed_dim)      # keeps torch's default init (reset never called)
        self.compute_dtype = torch.float32
        self.mask_sharing = MASK_SHARING          # see MASK_SHARING above

    def reset_parameters(self):
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.constant_(self.out_proj.bias, 0.)

    def _weights(self, pad):
        ws = []
        for l in range(self.num_layers):
            for suf in ("", "_reverse"):
                w_ih = getattr(self.rnn, "weight_ih_l%d%s" % (l, suf))
                if l == 0 and pad:
                    w_ih = F.pad(w_ih, (0, pad))
                ws += [w_ih, getattr(self.rnn, "weight_hh_l%d%s" % (l, suf)),
                       getattr(self.rnn, "bias_ih_l%d%s" % (l, suf)), getattr(self.rnn, "bias_hh_l%d%s" % (l, suf))]
        return ws

    def _trie_ok(self, src_tokens):
        # with dropout active the trie evaluation shares masks between paths: only when that was asked for
        masks_shared_ok = self.mask_sharing == "node" or not (self.training and self.dropout > 0)
        return (_gru.TRIE and masks_shared_ok and self.compute_dtype == torch.bfloat16 and self.num_layers == 2
                and self.hidden_size % 64 == 0 and src_tokens.is_cuda)

    def forward(self, src_tokens, src_lengths, trie=None):
        """``trie``: the batch's gtos_amd.pathtrie.PathTrie (``batch['relation_trie']``, built by the loader on the host
        with the bank); when the trie path applies and none is given it is built here (a host round trip)."""
        rel_dim = self.rel_embed.weight.shape[1]
        pad = (-rel_dim) % 8                                                       # 16-byte rows for the GEMM
        if self._trie_ok(src_tokens):
            if trie is None or not trie.matches(src_tokens, src_lengths):
                trie = None
                if TRIE_DEVICE:
                    try:
                        if TRIE_DEVICE == "hip":
                            from .pathtrie_hip import HipBackend, build_path_trie_staged
                            trie = build_path_trie_staged(src_tokens, src_lengths, HipBackend.shared())
                        else:
                            from .pathtrie_device import build_path_trie_device
                            trie = build_path_trie_device(src_tokens, src_lengths)
                    except ValueError:        # more than 8 labels per path / label ids >= 255: the host builder's business
                        trie = None
                if trie is None:
                    try:
                        trie = build_path_trie(src_tokens, src_lengths).to(src_tokens.device)
                    except ValueError:        # a bank the trie builder rejects (paths longer than 64 labels): one row per
                        trie = None           # (path, position) below, like the reference's packed sequence
        if trie is not None and trie.matches(src_tokens, src_lengths) and self._trie_ok(src_tokens):
            p_e = self.dropout if self.training else 0.0
            # final states [R, 2h], already in bank order (the step kernels scatter them through trie.seq_order)
            fin = trie_bigru_final(trie, self.rel_embed.weight, rel_dim + pad, p_e, self.hidden_size, p_e, self._weights(pad))
            return ops.linear(fin, self.out_proj.weight, self.out_proj.bias)
        if PACKED and src_tokens.is_cuda and self.compute_dtype == torch.bfloat16 and self.num_layers == 2 and self.hidden_size % 64 == 0 \
                and src_tokens.size(0) <= 64:
            # One row per (path, position) -- the reference's training-mode function -- on the fused kernels (gtos_amd.gru.PackedPathGRUFn).
            # With the batch's trie the sorted order and the step sizes are already known (the loader built them with the bank): no sort,
            # no host read, no packing copies; a bare bank costs one sort and one read of the step sizes.
            plan = PackPlan.of_trie(trie) if (trie is not None and trie.matches(src_tokens, src_lengths)) \
                else PackPlan.of_lengths(src_lengths, src_tokens.size(0))
            p = self.dropout if self.training else 0.0
            fin = packed_path_gru(src_tokens, plan, self.rel_embed.weight, rel_dim + (-rel_dim) % 64, p, self.hidden_size, p, self._weights(0))
            return ops.linear(fin, self.out_proj.weight, self.out_proj.bias)
        seq_len, bsz = src_tokens.size()
        sorted_len, indices = torch.sort(src_lengths, descending=True, stable=True)
        toks = src_tokens.index_select(1, indices)                                  # [L, R] sorted
        # packed layout: step t holds the sequences with length > t (a prefix of the sorted order)
        batch_sizes = (sorted_len.unsqueeze(0) > torch.arange(seq_len, device=src_tokens.device).unsqueeze(1)).sum(1).tolist()
        while batch_sizes and batch_sizes[-1] == 0:
            batch_sizes.pop()
        packed = torch.cat([toks[t, :a] for t, a in enumerate(batch_sizes)])       # [N]
        x = ops.embed_rows(packed, self.rel_embed.weight, rel_dim + pad, self.dropout if self.training else 0.0,
                           self.compute_dtype)
        p = self.dropout if (self.training and self.num_layers > 1) else 0.0
        fin = bigru_final(x, batch_sizes, self.hidden_size, self.num_layers, p, self._weights(pad))   # [R, 2h] sorted
        positions = torch.sort(indices)[1]
        fin = ops.permute_rows(fin, positions, indices)                            # unsort (inverse of `indices`)
        return ops.linear(fin, self.out_proj.weight, self.out_proj.bias)


class Highway(nn.Module):
    def __init__(self, input_dim, layers):
        super().__init__()
        self.input_dim = input_dim
        self.layers = nn.ModuleList([nn.Linear(input_dim, input_dim * 2) for _ in range(layers)])
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.layers:
            nn.init.normal_(layer.weight, std=0.02)
            nn.init.constant_(layer.bias[self.input_dim:], 1)
            nn.init.constant_(layer.bias[:self.input_dim], 0)

    def forward(self, x):
        for layer in self.layers:
            y = ops.linear(x, layer.weight, layer.bias)                  # [.., 2D] = [new_x | gate]
            if x.is_cuda and self.input_dim % 8 == 0:
                x = ops.highway_gate(y, x)                               # sigmoid(gate) * x + (1 - sigmoid(gate)) * relu(new_x), one kernel
            else:
                new_x, gate = y.chunk(2, dim=-1)
                gate = torch.sigmoid(gate)
                x = gate * x + (1 - gate) * F.relu(new_x)
        return x


class CNNEncoder(nn.Module):
    """char CNN: Conv1d(k) -> max over time -> ReLU -> highway -> Linear (encoder.py:151-178).  The convolution is
    evaluated as unfold + MFMA GEMM on the Conv1d weights (no MIOpen)."""

    def __init__(self, filters, input_dim, output_dim, highway_layers=1):
        super().__init__()
        self.convolutions = nn.ModuleList()
        for width, out_c in filters:
            self.convolutions.append(nn.Conv1d(input_dim, out_c, kernel_size=width))
        final_dim = sum(f[1] for f in filters)
        self.highway = Highway(final_dim, highway_layers)
        self.out_proj = nn.Linear(final_dim, output_dim)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, input):                     # [N, chars, dim]
        N, L, C = input.shape
        feats = []
        for conv in self.convolutions:
            k = conv.kernel_size[0]
            win = input.unfold(1, k, 1)                                   # [N, L-k+1, C, k]
            y = ops.linear(win.reshape(N * (L - k + 1), C * k), conv.weight.view(conv.out_channels, C * k), conv.bias)
            y = y.view(N, L - k + 1, -1)
            if y.is_cuda and conv.out_channels % 8 == 0 and L - k + 1 <= 255:
                feats.append(ops.max_relu(y))                            # max over time + ReLU, one kernel (argmax kept for backward)
            else:
                feats.append(F.relu(y.max(1)[0]))
        x = self.highway(feats[0] if len(feats) == 1 else torch.cat(feats, dim=-1))
        return ops.linear(x, self.out_proj.weight, self.out_proj.bias)


class TokenEncoder(nn.Module):
    def __init__(self, token_vocab, char_vocab, char_dim, token_dim, embed_dim, filters, char2token_dim, dropout,
                 pretrained_file=None):
        super().__init__()
        self.char_embed = AMREmbedding(char_vocab, char_dim)
        self.token_embed = AMREmbedding(token_vocab, token_dim, pretrained_file)
        self.char2token = CNNEncoder(filters, char_dim, char2token_dim)
        self.out_proj = nn.Linear(char2token_dim + token_dim, embed_dim)
        self.char_dim, self.token_dim, self.dropout = char_dim, token_dim, dropout
        self.compute_dtype = torch.float32
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, token_input, char_input):
        seq_len, bsz, _ = char_input.size()
        cd = self.compute_dtype
        ce = self.char_embed
        cdim = ce.weight.shape[1]
        if cdim % 8 == 0 and ce.weight.shape[0] * cdim * 4 <= 48 * 1024:
            # small table: gather kernel forward, LDS-accumulated scatter backward (torch's sort-based embedding
            # backward costs more than the whole character CNN here)
            char_repr = ops.embed_rows(char_input.reshape(-1), ce.weight, cdim, 0.0, cd, pad_idx=ce.padding_idx)
            char_repr = char_repr.view(seq_len * bsz, -1, cdim)
        else:
            char_repr = ce(char_input.view(seq_len * bsz, -1)).to(cd)
        char_repr = self.char2token(char_repr).view(seq_len, bsz, -1)
        te = self.token_embed
        if char_repr.is_cuda and char_repr.shape[-1] % 8 == 0 and te.weight.dtype == torch.float32:
            # cat([char_repr, token_embed(token)]) -> dropout -> zero-pad to a multiple of 8 columns, one kernel; the embedding's
            # backward is the kernel's scatter into the fp32 table (no sort-based embedding_dense_backward)
            token = ops.token_row(char_repr, token_input, te.weight, self.dropout if self.training else 0.0, te.padding_idx)
        else:
            token_repr = te(token_input).to(cd)
            token = F.dropout(torch.cat([char_repr, token_repr], -1), p=self.dropout, training=self.training)
            token = F.pad(token, (0, (-token.shape[-1]) % 8))
        w = self.out_proj.weight
        pad = token.shape[-1] - w.shape[1]
        if pad:                                   # 428 = 128+300 is not a multiple of 8: keep GEMM rows 16-byte aligned
            w = self._padded_weight(w, pad)
        return ops.linear(token, w, self.out_proj.bias)

    def _padded_weight(self, w, pad):
        return F.pad(w, (0, pad))
