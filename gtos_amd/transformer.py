"""Transformer / TransformerLayer / MultiheadAttention (decoder self- and cross-attention) on the gfx950 kernels.

Drop-in for /root/reference/generator/transformer.py (same signatures and state_dict keys).  The attention
core is the mode-0 variant of the fused relation-attention kernel; LayerNorm+residual+dropout and the FFN
use the same fused kernels as the graph encoder.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

from . import ops


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., weights_dropout=True):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        ops.check_attention_shape(embed_dim, num_heads)
        self.scaling = self.head_dim ** -0.5
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.weights_dropout = weights_dropout
        self.compute_dtype = torch.float32
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.in_proj_weight, std=0.02)
        nn.init.normal_(self.out_proj.weight, std=0.02)
        nn.init.constant_(self.in_proj_bias, 0.)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, key_padding_mask=None, attn_mask=None, need_weights=False):
        d, H = self.embed_dim, self.num_heads
        qkv_same = query.data_ptr() == key.data_ptr() == value.data_ptr()
        kv_same = key.data_ptr() == value.data_ptr()
        assert key.shape == value.shape
        cd = self.compute_dtype
        query = query.to(cd)
        W, b = self.in_proj_weight, self.in_proj_bias
        if qkv_same:
            qsrc, kvsrc, offs = ops.linear(query, W, b), None, (0, d, 2 * d)
        elif kv_same:
            qsrc = ops.linear(query, W, b, rows=(0, d))
            kvsrc, offs = ops.linear(key.to(cd), W, b, rows=(d, 3 * d)), (0, 0, d)
        else:
            qsrc = ops.linear(query, W, b, rows=(0, d))
            kvsrc = torch.cat([ops.linear(key.to(cd), W, b, rows=(d, 2 * d)),
                               ops.linear(value.to(cd), W, b, rows=(2 * d, 3 * d))], -1)
            offs = (0, 0, d)
        p_w = self.dropout if (self.weights_dropout and self.training) else 0.0
        o, w = ops.attention_core(qsrc, kvsrc, offs, d, H, self.scaling, key_pad=key_padding_mask,
                                  attn_mask=attn_mask, p_drop=p_w, need_weights=need_weights)
        if not self.weights_dropout:
            o = F.dropout(o, p=self.dropout, training=self.training)
        attn = ops.linear(o, self.out_proj.weight, self.out_proj.bias)
        if need_weights:
            w = w.max(dim=3)[0].transpose(1, 2)            # head-max, [tgt_len, bsz, src_len]
        return attn, w


    # ---- incremental decoding (inference only): projected K/V rows are cached instead of re-projecting the prefix
    def project_kv(self, key):
        """[S,B,d] -> packed [S,B,2d] = (W_k key + b_k | W_v key + b_v): the K/V cache rows of `key`."""
        d = self.embed_dim
        return ops.linear(key.to(self.compute_dtype), self.in_proj_weight, self.in_proj_bias, rows=(d, 3 * d))

    def attend_cached(self, query, kv, key_padding_mask=None, need_weights=False):
        """Attention of `query` [T,B,d] over cached projected rows kv [S,B,2d] (eval mode: no dropout)."""
        d, H = self.embed_dim, self.num_heads
        q = ops.linear(query.to(self.compute_dtype), self.in_proj_weight, self.in_proj_bias, rows=(0, d))
        o, w = ops.attention_core(q, kv, (0, 0, d), d, H, self.scaling, key_pad=key_padding_mask, need_weights=need_weights)
        attn = ops.linear(o, self.out_proj.weight, self.out_proj.bias)
        if need_weights:
            w = w.max(dim=3)[0].transpose(1, 2)
        return attn, w


class TransformerLayer(nn.Module):
    def __init__(self, embed_dim, ff_embed_dim, num_heads, dropout, with_external=False, weights_dropout=True):
        super().__init__()
        self.self_attn = MultiheadAttention(embed_dim, num_heads, dropout, weights_dropout)
        self.fc1 = nn.Linear(embed_dim, ff_embed_dim)
        self.fc2 = nn.Linear(ff_embed_dim, embed_dim)
        self.attn_layer_norm = nn.LayerNorm(embed_dim)
        self.ff_layer_norm = nn.LayerNorm(embed_dim)
        self.with_external = with_external
        self.dropout = dropout
        if self.with_external:
            self.external_attn = MultiheadAttention(embed_dim, num_heads, dropout, weights_dropout)
            self.external_layer_norm = nn.LayerNorm(embed_dim)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.fc1.weight, std=0.02)
        nn.init.normal_(self.fc2.weight, std=0.02)
        nn.init.constant_(self.fc1.bias, 0.)
        nn.init.constant_(self.fc2.bias, 0.)

    def forward(self, x, kv=None, self_padding_mask=None, self_attn_mask=None,
                external_memories=None, external_padding_mask=None, need_weights=False):
        p = self.dropout if self.training else 0.0
        cd = self.self_attn.compute_dtype
        # xs: the residual stream (fp32 also in bf16 mode, ops.FP32_STREAM), x: the same values in the compute dtype for the GEMMs
        xs, x = ops.split_stream(x, cd)
        src = x if kv is None else kv
        a, self_attn = self.self_attn(query=x, key=src, value=src, key_padding_mask=self_padding_mask,
                                      attn_mask=self_attn_mask, need_weights=need_weights)
        ln = self.attn_layer_norm
        xs, x = ops.layer_norm_stream(xs, a, ln.weight, ln.bias, p, ln.eps, cd)
        if self.with_external:
            a, external_attn = self.external_attn(query=x, key=external_memories, value=external_memories,
                                                  key_padding_mask=external_padding_mask, need_weights=need_weights)
            ln = self.external_layer_norm
            xs, x = ops.layer_norm_stream(xs, a, ln.weight, ln.bias, p, ln.eps, cd)
        else:
            external_attn = None
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, relu=True, p_drop=p)
        f = ops.linear(h, self.fc2.weight, self.fc2.bias)
        ln = self.ff_layer_norm
        xs, x = ops.layer_norm_stream(xs, f, ln.weight, ln.bias, p, ln.eps, cd)
        return ops.join_stream(xs, x), self_attn, external_attn


    def step(self, x, new_kv_src, self_cache, ext_kv, ext_mask):
        """One decoding step in eval mode.  x [1,N,d]: the query rows; new_kv_src [1,N,d]: the row whose K/V projection
        joins this layer's self-attention cache (x itself in the sentence encoder, the newest token state in the
        inference core); self_cache [t,N,2d] or None; ext_kv [S,N,2d]: cached projection of the graph states.
        Same arithmetic as forward(x, kv=whole prefix, ...) of generator/transformer.py:25-44, without re-projecting
        the prefix.  Returns (x_out, grown cache)."""
        cd = self.self_attn.compute_dtype
        # the same residual stream as forward(): fp32 also in bf16 mode (ops.FP32_STREAM), with its bf16 twin as the GEMM operand, so that
        # teacher-forced forward and incremental decoding of one model round alike (the layer output carries the stream to the next layer)
        xs, x = ops.split_stream(x, cd)
        row = self.self_attn.project_kv(ops.split_stream(new_kv_src, cd)[1])
        cache = row if self_cache is None else torch.cat([self_cache, row], 0)
        a, _ = self.self_attn.attend_cached(x, cache)
        ln = self.attn_layer_norm
        xs, x = ops.layer_norm_stream(xs, a, ln.weight, ln.bias, 0.0, ln.eps, cd)
        if self.with_external:
            a, _ = self.external_attn.attend_cached(x, ext_kv, key_padding_mask=ext_mask)
            ln = self.external_layer_norm
            xs, x = ops.layer_norm_stream(xs, a, ln.weight, ln.bias, 0.0, ln.eps, cd)
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, relu=True)
        f = ops.linear(h, self.fc2.weight, self.fc2.bias)
        ln = self.ff_layer_norm
        xs, x = ops.layer_norm_stream(xs, f, ln.weight, ln.bias, 0.0, ln.eps, cd)
        return ops.join_stream(xs, x), cache


class Transformer(nn.Module):
    def __init__(self, layers, embed_dim, ff_embed_dim, num_heads, dropout, with_external=False, weights_dropout=True):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(layers):
            self.layers.append(TransformerLayer(embed_dim, ff_embed_dim, num_heads, dropout, with_external, weights_dropout))

    def forward(self, x, kv=None, self_padding_mask=None, self_attn_mask=None,
                external_memories=None, external_padding_mask=None):
        for layer in self.layers:
            x, _, _ = layer(x, kv, self_padding_mask, self_attn_mask, external_memories, external_padding_mask)
        return x


def Embedding(num_embeddings, embedding_dim, padding_idx):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, std=0.02)
    nn.init.constant_(m.weight[padding_idx], 0)
    return m


class SelfAttentionMask(nn.Module):
    """Causal mask, True above the diagonal (bool; the reference builds uint8, transformer.py:204-219)."""

    def __init__(self, device, init_size=100):
        super().__init__()
        self.device = device
        self.weights = None
        self._cut = {}

    def forward(self, size):
        if self.weights is None or size > self.weights.size(0):
            self.weights = torch.ones((max(size, 100), max(size, 100)), dtype=torch.bool, device=self.device).triu_(1)
            self._cut = {}
        m = self._cut.get(size)
        if m is None:                  # contiguous [size,size] copy made once per size: the kernels read it as bytes, every layer
            m = self._cut[size] = self.weights[:size, :size].contiguous()
        return m


class SinusoidalPositionalEmbedding(nn.Module):
    """[sin | cos] table, tensor2tensor style (transformer.py:240-281)."""

    def __init__(self, embedding_dim, device, init_size=512):
        super().__init__()
        self.embedding_dim, self.device = embedding_dim, device
        self.weights = self.get_embedding(init_size, embedding_dim).to(device)

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim):
        half = embedding_dim // 2
        f = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
        ang = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
        emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num_embeddings, -1)
        if embedding_dim % 2 == 1:
            emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
        return emb

    def forward(self, input, offset=0):
        seq_len, bsz = input.size()
        mx = seq_len + offset
        if mx > self.weights.size(0):
            self.weights = self.get_embedding(mx, self.embedding_dim).to(self.device)
        return self.weights[offset:offset + seq_len].unsqueeze(1).expand(-1, bsz, -1)
