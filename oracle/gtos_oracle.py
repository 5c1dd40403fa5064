"""CPU oracle for the gtos graph-transformer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gtos_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and only as the checker / reported baseline -- never as the thing shipped or measured.

It is a from-scratch restatement, in plain PyTorch CPU ops, of the math of the
reference's hot-path modules (citations are relative to /root/reference):

  * RelationMultiheadAttention / GraphTransformerLayer / GraphTransformer
        generator/graph_transformer.py:6-197
  * MultiheadAttention / TransformerLayer / Transformer / sinusoid / causal mask
        generator/transformer.py:7-281
  * RelationEncoder (embedding -> 2-layer bi-GRU -> final state -> Linear)
        generator/encoder.py:66-119
  * TokenEncoder / CNNEncoder / Highway        generator/encoder.py:123-201
  * TokenGenerator / DecodeLayer               generator/decoder.py:10-94
  * Generator.encode_step / forward            generator/generator.py:71-95,169-182
  * AdamWeightDecayOptimizer.step / update_lr  generator/adam.py:28-87, generator/train.py:81-83

Parity is PINNED: tests/test_oracle_golden.py checks every function here against the golden
vectors in tests/golden/*.npz, which tests/golden/make_golden.py produced by importing the
reference itself in the build container (the reference never travels to the GPU box).

The modules keep the reference's constructor signatures and state_dict keys so one state
dict loads into the reference, this oracle and the HIP-backed product modules alike.
"""
import math
from collections import namedtuple

import torch
from torch import nn
import torch.nn.functional as F

VocabSpec = namedtuple("VocabSpec", ["size", "padding_idx"])

NEG_INF = float("-inf")


# Mask injection (tests only): MASK_HOOK(tag, x) -> a keep mask (bool / 0-1 tensor broadcastable to x) or None.  With a hook, a tagged
# dropout site applies x * keep / (1 - p) instead of drawing from torch's generator, so that a test can hand the oracle exactly the masks
# the product's counter-based hash draws (tests/test_hip_parity.py: the RelationEncoder's training-mode function, reference masks per
# (path, position) and the opt-in trie-shared ones).  Without a hook nothing changes.
MASK_HOOK = None


def _drop(x, p, training, tag=None):
    if not (training and p > 0):
        return x
    if MASK_HOOK is not None and tag is not None:
        keep = MASK_HOOK(tag, x)
        if keep is not None:
            return x * keep.to(x.dtype) / (1.0 - p)
    return F.dropout(x, p=p, training=True)


# --------------------------------------------------------------------------------------
# relation-aware attention   (generator/graph_transformer.py:93-174)
# --------------------------------------------------------------------------------------
def relation_attention_scores(q, k, ra, rb, scaling):
    """s[i,j,b,h] = scaling * sum_e (q[i,b,h,e] + ra[j,i,b,h,e]) * (k[j,b,h,e] + rb[j,i,b,h,e]).

    q:[T,B,H,E] k:[S,B,H,E] ra,rb:[S(j),T(i),B,H,E] (memory order of the projected relation;
    the reference swaps the first two axes at graph_transformer.py:123-124)."""
    qa = q.unsqueeze(0) + ra          # [j,i,b,h,e]
    kb = k.unsqueeze(1) + rb          # [j,i,b,h,e]
    return scaling * (qa * kb).sum(-1).transpose(0, 1)   # -> [i,j,b,h]


def relation_attention_core(q, k, v, ra, rb, scaling, key_padding_mask=None, attn_mask=None,
                            p_drop=0.0, training=False):
    """Returns (o[T,B,H,E], p[T,S,B,H]) -- p is post-dropout, like the reference's weights."""
    s = relation_attention_scores(q, k, ra, rb, scaling)
    if attn_mask is not None:
        s = s.masked_fill(attn_mask.bool()[:, :, None, None], NEG_INF)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.bool()[None, :, :, None], NEG_INF)
    p = torch.softmax(s, dim=1)
    p = _drop(p, p_drop, training)
    o = torch.einsum("ijbh,jbhe->ibhe", p, v)
    return o, p


class RelationMultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., weights_dropout=True):
        super().__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.weights_dropout = weights_dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.relation_in_proj = nn.Linear(embed_dim, 2 * embed_dim, bias=False)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        for w in (self.in_proj_weight, self.out_proj.weight, self.relation_in_proj.weight):
            nn.init.normal_(w, std=0.02)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, query, key, value, relation, key_padding_mask=None, attn_mask=None,
                need_weights=False):
        T, B, D = query.shape
        S = key.shape[0]
        assert key.shape == value.shape
        H, E = self.num_heads, self.head_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        q = F.linear(query, W[:D], b[:D]).view(T, B, H, E)
        k = F.linear(key, W[D:2 * D], b[D:2 * D]).view(S, B, H, E)
        v = F.linear(value, W[2 * D:], b[2 * D:]).view(S, B, H, E)
        rr = self.relation_in_proj(relation)                   # [a,b,B,2D], a pairs with keys
        assert rr.shape[0] == S and rr.shape[1] == T
        ra = rr[..., :D].reshape(S, T, B, H, E)
        rb = rr[..., D:].reshape(S, T, B, H, E)
        o, p = relation_attention_core(q, k, v, ra, rb, self.scaling, key_padding_mask, attn_mask,
                                       self.dropout if self.weights_dropout else 0.0, self.training)
        if not self.weights_dropout:
            o = _drop(o, self.dropout, self.training)
        out = self.out_proj(o.reshape(T, B, D))
        return out, (p if need_weights else None)


class GraphTransformerLayer(nn.Module):
    def __init__(self, embed_dim, ff_embed_dim, num_heads, dropout, weights_dropout=True):
        super().__init__()
        self.self_attn = RelationMultiheadAttention(embed_dim, num_heads, dropout, weights_dropout)
        self.fc1 = nn.Linear(embed_dim, ff_embed_dim)
        self.fc2 = nn.Linear(ff_embed_dim, embed_dim)
        self.attn_layer_norm = nn.LayerNorm(embed_dim)
        self.ff_layer_norm = nn.LayerNorm(embed_dim)
        self.dropout = dropout
        for fc in (self.fc1, self.fc2):
            nn.init.normal_(fc.weight, std=0.02)
            nn.init.zeros_(fc.bias)

    def forward(self, x, relation, kv=None, self_padding_mask=None, self_attn_mask=None,
                need_weights=False):
        src = x if kv is None else kv
        a, w = self.self_attn(x, src, src, relation, self_padding_mask, self_attn_mask, need_weights)
        x = self.attn_layer_norm(x + _drop(a, self.dropout, self.training))
        h = _drop(torch.relu(self.fc1(x)), self.dropout, self.training)
        x = self.ff_layer_norm(x + _drop(self.fc2(h), self.dropout, self.training))
        return x, w


class GraphTransformer(nn.Module):
    def __init__(self, layers, embed_dim, ff_embed_dim, num_heads, dropout, weights_dropout=True):
        super().__init__()
        self.layers = nn.ModuleList(
            GraphTransformerLayer(embed_dim, ff_embed_dim, num_heads, dropout, weights_dropout)
            for _ in range(layers))

    def forward(self, x, relation, kv=None, self_padding_mask=None, self_attn_mask=None):
        for layer in self.layers:
            x, _ = layer(x, relation, kv, self_padding_mask, self_attn_mask)
        return x

    def get_attn_weights(self, x, relation, kv=None, self_padding_mask=None, self_attn_mask=None):
        ws = []
        for layer in self.layers:
            x, w = layer(x, relation, kv, self_padding_mask, self_attn_mask, need_weights=True)
            ws.append(w)
        return torch.stack(ws)


# --------------------------------------------------------------------------------------
# relation lookup / aggregation   (generator/generator.py:76-90)
# --------------------------------------------------------------------------------------
def relation_lookup_train(bank_repr, rel_idx):
    """bank_repr [R,d], rel_idx [n,n,B] -> dense relation [n,n,B,d] (generator.py:79)."""
    return bank_repr.index_select(0, rel_idx.reshape(-1)).view(*rel_idx.shape, -1)


def relation_lookup_eval(bank_repr, rel_idx):
    """bank_repr [R,d], rel_idx [n,n,B,K]: zero row 0, mean over the non-zero ids
    (generator.py:83-88).  Does not mutate bank_repr (the reference writes row 0 in place)."""
    bank = torch.cat([torch.zeros_like(bank_repr[:1]), bank_repr[1:]], 0)
    g = bank[rel_idx]                                    # [n,n,B,K,d]
    cnt = rel_idx.ne(0).sum(-1).clamp(min=1)
    return g.sum(3) / cnt.unsqueeze(-1).to(g.dtype)


# --------------------------------------------------------------------------------------
# vanilla attention blocks   (generator/transformer.py)
# --------------------------------------------------------------------------------------
class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., weights_dropout=True):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.weights_dropout = weights_dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        nn.init.normal_(self.in_proj_weight, std=0.02)
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, query, key, value, key_padding_mask=None, attn_mask=None, need_weights=False):
        T, B, D = query.shape
        S = key.shape[0]
        H, E = self.num_heads, self.head_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        q = (F.linear(query, W[:D], b[:D]) * self.scaling).view(T, B, H, E)
        k = F.linear(key, W[D:2 * D], b[D:2 * D]).view(S, B, H, E)
        v = F.linear(value, W[2 * D:], b[2 * D:]).view(S, B, H, E)
        s = torch.einsum("tbhe,sbhe->tsbh", q, k)
        if attn_mask is not None:
            s = s.masked_fill(attn_mask.bool()[:, :, None, None], NEG_INF)
        if key_padding_mask is not None:
            s = s.masked_fill(key_padding_mask.bool()[None, :, :, None], NEG_INF)
        p = torch.softmax(s, dim=1)
        if self.weights_dropout:
            p = _drop(p, self.dropout, self.training)
        o = torch.einsum("tsbh,sbhe->tbhe", p, v)
        if not self.weights_dropout:
            o = _drop(o, self.dropout, self.training)
        out = self.out_proj(o.reshape(T, B, D))
        w = p.max(dim=3)[0].transpose(1, 2) if need_weights else None    # [T,B,S] head-max
        return out, w


class TransformerLayer(nn.Module):
    def __init__(self, embed_dim, ff_embed_dim, num_heads, dropout, with_external=False,
                 weights_dropout=True):
        super().__init__()
        self.self_attn = MultiheadAttention(embed_dim, num_heads, dropout, weights_dropout)
        self.fc1 = nn.Linear(embed_dim, ff_embed_dim)
        self.fc2 = nn.Linear(ff_embed_dim, embed_dim)
        self.attn_layer_norm = nn.LayerNorm(embed_dim)
        self.ff_layer_norm = nn.LayerNorm(embed_dim)
        self.with_external = with_external
        self.dropout = dropout
        if with_external:
            self.external_attn = MultiheadAttention(embed_dim, num_heads, dropout, weights_dropout)
            self.external_layer_norm = nn.LayerNorm(embed_dim)
        for fc in (self.fc1, self.fc2):
            nn.init.normal_(fc.weight, std=0.02)
            nn.init.zeros_(fc.bias)

    def forward(self, x, kv=None, self_padding_mask=None, self_attn_mask=None,
                external_memories=None, external_padding_mask=None, need_weights=False):
        src = x if kv is None else kv
        a, sw = self.self_attn(x, src, src, self_padding_mask, self_attn_mask, need_weights)
        x = self.attn_layer_norm(x + _drop(a, self.dropout, self.training))
        ew = None
        if self.with_external:
            a, ew = self.external_attn(x, external_memories, external_memories,
                                       external_padding_mask, None, need_weights)
            x = self.external_layer_norm(x + _drop(a, self.dropout, self.training))
        h = _drop(torch.relu(self.fc1(x)), self.dropout, self.training)
        x = self.ff_layer_norm(x + _drop(self.fc2(h), self.dropout, self.training))
        return x, sw, ew


class Transformer(nn.Module):
    def __init__(self, layers, embed_dim, ff_embed_dim, num_heads, dropout, with_external=False,
                 weights_dropout=True):
        super().__init__()
        self.layers = nn.ModuleList(
            TransformerLayer(embed_dim, ff_embed_dim, num_heads, dropout, with_external, weights_dropout)
            for _ in range(layers))

    def forward(self, x, kv=None, self_padding_mask=None, self_attn_mask=None,
                external_memories=None, external_padding_mask=None):
        for layer in self.layers:
            x, _, _ = layer(x, kv, self_padding_mask, self_attn_mask, external_memories,
                            external_padding_mask)
        return x


def causal_mask(size):
    """True above the diagonal (generator/transformer.py:204-219, as bool)."""
    return torch.ones(size, size, dtype=torch.bool).triu_(1)


def sinusoid_table(num, dim):
    """[sin block | cos block], tensor2tensor style (generator/transformer.py:252-266)."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], 1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num, 1)], 1)
    return emb


def make_embedding(num, dim, padding_idx):
    m = nn.Embedding(num, dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, std=0.02)
    nn.init.zeros_(m.weight[padding_idx])
    return m


# --------------------------------------------------------------------------------------
# relation encoder: bi-GRU over label paths   (generator/encoder.py:66-119)
# --------------------------------------------------------------------------------------
def gru_cell(x_gates, h, w_hh, b_hh):
    """PyTorch GRU cell, gate order [r; z; n].  x_gates = W_ih x + b_ih, shape [R,3h]."""
    hg = F.linear(h, w_hh, b_hh)
    hs = h.shape[1]
    r = torch.sigmoid(x_gates[:, :hs] + hg[:, :hs])
    z = torch.sigmoid(x_gates[:, hs:2 * hs] + hg[:, hs:2 * hs])
    n = torch.tanh(x_gates[:, 2 * hs:] + r * hg[:, 2 * hs:])
    return (1 - z) * n + z * h


def gru_direction(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [L,R,in], lengths [R] -> (outputs [L,R,h] (zeros past the end), final h [R,h]).
    Each sequence r runs over its own len_r steps with h0 = 0."""
    L, R, _ = x.shape
    hs = w_hh.shape[1]
    h = x.new_zeros(R, hs)
    outs = [None] * L
    steps = range(L - 1, -1, -1) if reverse else range(L)
    for t in steps:
        active = (lengths > t).unsqueeze(1)
        hn = gru_cell(F.linear(x[t], w_ih, b_ih), h, w_hh, b_hh)
        h = torch.where(active, hn, h)
        outs[t] = torch.where(active, hn, torch.zeros_like(hn))
    return torch.stack(outs), h


class RelationEncoder(nn.Module):
    def __init__(self, vocab, rel_dim, embed_dim, hidden_size, num_layers, dropout, bidirectional=True):
        super().__init__()
        assert bidirectional, "the reference only ever builds the bidirectional encoder"
        self.vocab, self.embed_dim, self.hidden_size = vocab, embed_dim, hidden_size
        self.num_layers, self.dropout = num_layers, dropout
        self.rel_embed = make_embedding(vocab.size, rel_dim, vocab.padding_idx)
        # nn.GRU only as the parameter container (reference key names rnn.weight_ih_l0 ...)
        self.rnn = nn.GRU(rel_dim, hidden_size, num_layers=num_layers,
                          dropout=dropout if num_layers > 1 else 0., bidirectional=True)
        self.out_proj = nn.Linear(2 * hidden_size, embed_dim)   # default init: reset never called

    def forward(self, src_tokens, src_lengths):
        emb = self.rel_embed(src_tokens)                                      # [L,R,rel_dim]
        x = _drop(emb, self.dropout, self.training, "relenc.embed")
        xr = x                                                                # the reverse direction reads the same dropped embeddings ...
        if MASK_HOOK is not None and self.training and self.dropout > 0:
            keep = MASK_HOOK("relenc.embed.reverse", emb)                     # ... unless a test injects a separate mask for it (the trie-
            if keep is not None:                                              # shared variant draws one per direction)
                xr = emb * keep.to(emb.dtype) / (1.0 - self.dropout)
        fin = None
        for l in range(self.num_layers):
            p = lambda n, s="": getattr(self.rnn, "%s_l%d%s" % (n, l, s))
            of, hf = gru_direction(x, src_lengths, p("weight_ih"), p("weight_hh"),
                                   p("bias_ih"), p("bias_hh"), False)
            ob, hb = gru_direction(xr, src_lengths, p("weight_ih", "_reverse"), p("weight_hh", "_reverse"),
                                   p("bias_ih", "_reverse"), p("bias_hh", "_reverse"), True)
            fin = torch.cat([hf, hb], 1)
            x = torch.cat([of, ob], 2)
            if l + 1 < self.num_layers:
                x = _drop(x, self.dropout, self.training, "relenc.layer%d" % l)
            xr = x
        return self.out_proj(fin)


# --------------------------------------------------------------------------------------
# token / concept encoder   (generator/encoder.py:123-201)
# --------------------------------------------------------------------------------------
class Highway(nn.Module):
    def __init__(self, input_dim, layers):
        super().__init__()
        self.input_dim = input_dim
        self.layers = nn.ModuleList(nn.Linear(input_dim, 2 * input_dim) for _ in range(layers))
        for l in self.layers:
            nn.init.normal_(l.weight, std=0.02)
            nn.init.zeros_(l.bias[:input_dim])
            nn.init.ones_(l.bias[input_dim:])

    def forward(self, x):
        for l in self.layers:
            y, g = l(x).chunk(2, -1)
            g = torch.sigmoid(g)
            x = g * x + (1 - g) * torch.relu(y)
        return x


class CNNEncoder(nn.Module):
    def __init__(self, filters, input_dim, output_dim, highway_layers=1):
        super().__init__()
        self.convolutions = nn.ModuleList(nn.Conv1d(input_dim, c, kernel_size=w) for w, c in filters)
        tot = sum(c for _, c in filters)
        self.highway = Highway(tot, highway_layers)
        self.out_proj = nn.Linear(tot, output_dim)
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x):                                   # [N, chars, dim]
        x = x.transpose(1, 2)
        feats = [torch.relu(conv(x).max(-1)[0]) for conv in self.convolutions]
        return self.out_proj(self.highway(torch.cat(feats, -1)))


class TokenEncoder(nn.Module):
    def __init__(self, token_vocab, char_vocab, char_dim, token_dim, embed_dim, filters,
                 char2token_dim, dropout, pretrained_file=None):
        super().__init__()
        assert pretrained_file is None, "pretrained-embedding file loading is out of scope"
        self.char_embed = make_embedding(char_vocab.size, char_dim, char_vocab.padding_idx)
        self.token_embed = make_embedding(token_vocab.size, token_dim, token_vocab.padding_idx)
        self.char2token = CNNEncoder(filters, char_dim, char2token_dim)
        self.out_proj = nn.Linear(char2token_dim + token_dim, embed_dim)
        self.dropout = dropout
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, token_input, char_input):
        L, B, C = char_input.shape
        ch = self.char2token(self.char_embed(char_input.view(L * B, C))).view(L, B, -1)
        tok = torch.cat([ch, self.token_embed(token_input)], -1)
        return self.out_proj(_drop(tok, self.dropout, self.training))


# --------------------------------------------------------------------------------------
# decoder   (generator/decoder.py)
# --------------------------------------------------------------------------------------
class TokenGenerator(nn.Module):
    def __init__(self, vocabs, embed_dim, token_size, dropout):
        super().__init__()
        self.alignment_layer = MultiheadAttention(embed_dim, 1, dropout, weights_dropout=False)
        self.alignment_layer_norm = nn.LayerNorm(embed_dim)
        self.transfer = nn.Linear(embed_dim, token_size)
        self.generator = nn.Linear(token_size, vocabs['predictable_token'].size)
        self.diverter = nn.Linear(token_size, 2)
        self.vocabs, self.dropout = vocabs, dropout
        for l in (self.transfer, self.generator, self.diverter):
            nn.init.normal_(l.weight, std=0.02)
            nn.init.zeros_(l.bias)

    def forward(self, outs, graph_state, graph_padding_mask, copy_seq, target=None, work=False):
        a, align = self.alignment_layer(outs, graph_state, graph_state, graph_padding_mask,
                                        need_weights=True)               # align [T,B,S]
        outs = self.alignment_layer_norm(outs + _drop(a, self.dropout, self.training))
        T, B, _ = outs.shape
        u = _drop(torch.tanh(self.transfer(outs)), self.dropout, self.training)
        gate = torch.softmax(self.diverter(u), -1)
        probs = gate[..., :1] * torch.softmax(self.generator(u), -1)
        V = probs.shape[-1]
        ext = 1 + int(copy_seq.max()) - V
        if ext > 0:
            probs = torch.cat([probs, probs.new_zeros(T, B, ext)], -1)
        index = copy_seq.transpose(0, 1).reshape(1, B, -1).expand(T, -1, -1)
        probs = probs.scatter_add(-1, index, (gate[..., 1:] * align).reshape(T, B, -1))
        ll = torch.log(probs + 1e-12)
        if work:
            return ll
        nll = -ll.gather(-1, target.unsqueeze(-1)).squeeze(-1)
        pad = target.eq(self.vocabs['predictable_token'].padding_idx)
        return nll.masked_fill(pad, 0.).sum(0)


class DecodeLayer(nn.Module):
    def __init__(self, vocabs, inference_layers, embed_dim, ff_embed_dim, num_heads, token_size,
                 rel_size, dropout):
        super().__init__()
        self.inference_core = Transformer(inference_layers, embed_dim, ff_embed_dim, num_heads,
                                          dropout, with_external=True)
        self.token_generator = TokenGenerator(vocabs, embed_dim, token_size, dropout)
        self.dropout, self.vocabs = dropout, vocabs

    def forward(self, probe, graph_state, snt_state, graph_padding_mask, snt_padding_mask, attn_mask,
                copy_seq, target=None, work=False):
        outs = self.inference_core(_drop(probe, self.dropout, self.training), kv=snt_state,
                                   self_padding_mask=snt_padding_mask, self_attn_mask=attn_mask,
                                   external_memories=graph_state,
                                   external_padding_mask=graph_padding_mask)
        if work:
            return self.token_generator(outs, graph_state, graph_padding_mask, copy_seq, work=True)
        loss = self.token_generator(outs, graph_state, graph_padding_mask, copy_seq, target=target)
        ntok = snt_padding_mask.shape[0] - snt_padding_mask.float().sum(0)
        return (loss / ntok).mean()


# --------------------------------------------------------------------------------------
# model assembly   (generator/generator.py:14-95,169-182; translator: depth table of 256)
# --------------------------------------------------------------------------------------
class Generator(nn.Module):
    def __init__(self, vocabs, word_char_dim, word_dim, concept_char_dim, concept_dim, cnn_filters,
                 char2word_dim, char2concept_dim, rel_dim, rnn_hidden_size, rnn_num_layers,
                 embed_dim, ff_embed_dim, num_heads, dropout, snt_layers, graph_layers,
                 inference_layers, pretrained_file=None, device=None, depth_size=32):
        super().__init__()
        self.vocabs = vocabs
        self.concept_encoder = TokenEncoder(vocabs['concept'], vocabs['concept_char'], concept_char_dim,
                                            concept_dim, embed_dim, cnn_filters, char2concept_dim, dropout)
        self.relation_encoder = RelationEncoder(vocabs['relation'], rel_dim, embed_dim,
                                                rnn_hidden_size, rnn_num_layers, dropout)
        self.token_encoder = TokenEncoder(vocabs['token'], vocabs['token_char'], word_char_dim,
                                          word_dim, embed_dim, cnn_filters, char2word_dim, dropout)
        self.graph_encoder = GraphTransformer(graph_layers, embed_dim, ff_embed_dim, num_heads, dropout)
        self.snt_encoder = Transformer(snt_layers, embed_dim, ff_embed_dim, num_heads, dropout,
                                       with_external=True)
        self.embed_dim, self.embed_scale, self.dropout = embed_dim, math.sqrt(embed_dim), dropout
        self.concept_depth = nn.Embedding(depth_size, embed_dim)
        self.token_embed_layer_norm = nn.LayerNorm(embed_dim)
        self.concept_embed_layer_norm = nn.LayerNorm(embed_dim)
        self.decoder = DecodeLayer(vocabs, inference_layers, embed_dim, ff_embed_dim, num_heads,
                                   concept_dim, rel_dim, dropout)
        self.probe_generator = nn.Linear(embed_dim, embed_dim)
        nn.init.normal_(self.probe_generator.weight, std=0.02)
        nn.init.zeros_(self.probe_generator.bias)
        nn.init.zeros_(self.concept_depth.weight)

    def encode_step(self, inp, train=True):
        c = self.embed_scale * self.concept_encoder(inp['concept'], inp['concept_char']) \
            + self.concept_depth(inp['concept_depth'])
        c = self.concept_embed_layer_norm(c)
        cmask = inp['concept'].eq(self.vocabs['concept'].padding_idx)
        bank = self.relation_encoder(inp['relation_bank'], inp['relation_length'])
        # eval batches of the generator carry K alternative paths per pair; the translator's carry one (its encode_step
        # has no train flag, translator/generator.py:67-80)
        rel = relation_lookup_train(bank, inp['relation']) if (train or inp['relation'].dim() == 3) else \
            relation_lookup_eval(bank, inp['relation'])
        c = self.graph_encoder(c, rel, self_padding_mask=cmask)
        probe = torch.tanh(self.probe_generator(c[:1]))
        return c[1:], cmask[1:], probe

    def forward(self, data):
        graph, gmask, probe = self.encode_step(data)
        T = data['token_in'].shape[0]
        pos = sinusoid_table(max(T, 2), self.embed_dim)[:T].unsqueeze(1)
        tok = self.embed_scale * self.token_encoder(data['token_in'], data['token_char_in']) + pos
        tok = _drop(self.token_embed_layer_norm(tok), self.dropout, self.training)
        tmask = data['token_in'].eq(self.vocabs['token'].padding_idx)
        amask = causal_mask(T)
        tok = self.snt_encoder(tok, self_padding_mask=tmask, self_attn_mask=amask,
                               external_memories=graph, external_padding_mask=gmask)
        return self.decoder(probe.expand_as(tok), graph, tok, gmask, tmask, amask, data['cp_seq'],
                            target=data['token_out'])


# --------------------------------------------------------------------------------------
# inference: beam search   (generator/generator.py:96-167, generator/search.py)
# --------------------------------------------------------------------------------------
# The reference decodes incrementally and interleaves all sentences of a batch in one loop.  Sentences never interact
# and decoding is causal, so this restatement does the plainest equivalent thing: one sentence at a time, and at every
# step the whole prefix of every live hypothesis is pushed through the decoder again (no state is carried).
def next_token_ll(model, graph, gmask, probe, cp_seq, prefixes, vocabs, max_string_len=20):
    """graph [S,1,d], gmask [S,1], probe [1,1,d], cp_seq [S,1] of ONE sentence; prefixes: n token-string lists of equal
    length t (each starts with <STR>).  Returns log-likelihoods [n, V+ext] of the next token (decoder.py:85-87)."""
    n, t = len(prefixes), len(prefixes[0])
    tv, cv = vocabs['token'], vocabs['token_char']
    ids = torch.tensor([[tv.token2idx(w) for w in p] for p in prefixes], dtype=torch.int64).t().contiguous()      # [t,n]
    chars = torch.tensor([[cv.token2idx(['<STR>'] + list(w[:max_string_len]) + ['<END>'])
                           + [cv.padding_idx] * (max_string_len - len(w[:max_string_len])) for w in p] for p in prefixes],
                         dtype=torch.int64).transpose(0, 1).contiguous()                                           # [t,n,22]
    g, gm, cs = graph.expand(-1, n, -1), gmask.expand(-1, n), cp_seq.expand(-1, n)
    pos = sinusoid_table(max(t, 2), model.embed_dim)[:t].unsqueeze(1)
    tok = model.token_embed_layer_norm(model.embed_scale * model.token_encoder(ids, chars) + pos)
    tok = model.snt_encoder(tok, self_padding_mask=None, self_attn_mask=causal_mask(t),
                            external_memories=g, external_padding_mask=gm)
    # the query is the probe alone: one row that sees the whole (already causal) prefix (generator.py:149)
    return model.decoder(probe.expand(-1, n, -1), g, tok, gm, None, None, cs, work=True)[0]


def beam_search_sentence(model, graph, gmask, probe, cp_seq, local_idx2token, vocabs, beam_size, max_time_step,
                         min_time_step=1):
    """-> (finished, alive): lists of (token strings incl. <STR>/<END>, accumulated log-likelihood) in the order the
    reference's Beam holds them (search.py:57-101)."""
    pv = vocabs['predictable_token']
    alive, finished, steps = [(['<STR>'], 0.0)], [], 0
    while len(finished) < beam_size and steps < max_time_step and alive:
        ll = next_token_ll(model, graph, gmask, probe, cp_seq, [seq for seq, _ in alive], vocabs)
        top_s, top_i = torch.topk(ll, beam_size, 1)
        pool = []
        for h, (seq, score) in enumerate(alive):
            for s, i in zip(top_s[h].tolist(), top_i[h].tolist()):
                word = local_idx2token[i] if i in local_idx2token else pv.idx2token(i)
                pool.append((seq + [word], float('-inf') if word == '<UNK>' else score + s))
        pool = sorted(pool, key=lambda c: -c[1])[:beam_size - len(finished)]       # stable, like list.sort(reverse=True)
        alive = []
        for seq, score in pool:
            if seq[-1] == '<END>':
                if len(seq) - 2 >= min_time_step:
                    finished.append((seq, score))
            else:
                alive.append((seq, score))
        steps += 1
    return finished, alive


def k_best(finished, alive, k, alpha):
    """Final ranking with length normalisation (search.py:97-101)."""
    cands = finished if finished else alive
    return sorted(cands, key=lambda c: -(c[1] / ((1 + len(c[0])) ** alpha)))[:k]


def generator_work(model, data, vocabs, beam_size, max_time_step, min_time_step=1):
    """All sentences of a batch (generator.py:96-110): list of (finished, alive)."""
    with torch.no_grad():
        graph, gmask, probe = model.encode_step(data, train=False)
        out = []
        for b in range(graph.shape[1]):
            out.append(beam_search_sentence(model, graph[:, b:b + 1], gmask[:, b:b + 1], probe[:, b:b + 1],
                                            data['cp_seq'][:, b:b + 1], data['local_idx2token'][b], vocabs, beam_size,
                                            max_time_step, min_time_step))
    return out


# --------------------------------------------------------------------------------------
# optimizer step   (generator/adam.py:28-87, generator/train.py:81-83,123-132,136-148,151)
# --------------------------------------------------------------------------------------
def inverse_sqrt_lr(embed_size, step, warmup_steps):
    return embed_size ** -0.5 * min(step ** -0.5, step * (warmup_steps ** -1.5))


class LoopCounters(object):
    """The bookkeeping of the training loop around the optimizer (generator/train.py:136-148): a batch whose loss exceeds
    5x the running mean after the warm-up is discarded (``continue`` before backward); otherwise loss_acm / batches_acm
    advance and the learning rate of the update is inverse_sqrt_lr(embed, batches_acm, warmup)."""

    def __init__(self, embed_size, warmup_steps):
        self.embed_size, self.warmup_steps = embed_size, warmup_steps
        self.loss_acm, self.batches_acm, self.discarded = 0.0, 0, 0

    def abnormal(self, loss_value):
        return self.batches_acm > self.warmup_steps and loss_value > 5. * (self.loss_acm / self.batches_acm)

    def advance(self, loss_value, discard):
        """-> learning rate of this step's update, or None when the batch is discarded."""
        if discard:
            self.discarded += 1
            return None
        self.loss_acm += loss_value
        self.batches_acm += 1
        return inverse_sqrt_lr(self.embed_size, self.batches_acm, self.warmup_steps)


def adam_step(p, g, m, v, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-6):
    """One parameter tensor; no bias correction; decoupled decay added before the lr multiply."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    upd = m / (v.sqrt() + eps) + weight_decay * p
    return p - lr * upd, m, v


def clip_coef(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ factor: min(1, max_norm / (||g||_2 + 1e-6))."""
    tot = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return torch.clamp(max_norm / (tot + 1e-6), max=1.0), tot


def is_no_decay(name):
    return name.endswith('bias') or 'layer_norm' in name
