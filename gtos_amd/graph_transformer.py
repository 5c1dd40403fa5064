"""GraphTransformer / GraphTransformerLayer / RelationMultiheadAttention backed by the gfx950 kernels.

Drop-in for /root/reference/generator/graph_transformer.py: same constructor and forward signatures, same
state_dict keys, same [T,B,C] / relation[n,n,B,d] conventions (relation[j][i] pairs with query i, key j).
Two extras that the reference API does not have, both optional:
  * ``relation`` may be an ``ops.FactoredRelation`` (bank + type ids) instead of the dense tensor;
  * ``compute_dtype`` (fp32 default, or bf16) selects the storage type of activations.
"""
import torch
from torch import nn

from . import ops
from .ops import FactoredRelation


class RelationMultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., weights_dropout=True):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        ops.check_attention_shape(embed_dim, num_heads)
        self.scaling = self.head_dim ** -0.5
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.relation_in_proj = nn.Linear(embed_dim, 2 * embed_dim, bias=False)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.weights_dropout = weights_dropout
        self.compute_dtype = torch.float32
        self.reset_parameters()

    def reset_parameters(self):
        for w in (self.in_proj_weight, self.out_proj.weight, self.relation_in_proj.weight):
            nn.init.normal_(w, std=0.02)
        nn.init.constant_(self.in_proj_bias, 0.)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, relation, key_padding_mask=None, attn_mask=None, need_weights=False):
        d, H = self.embed_dim, self.num_heads
        qkv_same = query.data_ptr() == key.data_ptr() == value.data_ptr()
        kv_same = key.data_ptr() == value.data_ptr()
        tgt_len, bsz, _ = query.shape
        assert key.shape == value.shape
        assert key.shape[0] == tgt_len, "the relation term needs tgt_len == src_len (graph_transformer.py:122-127)"
        cd = self.compute_dtype
        query = query.to(cd)
        if qkv_same:
            qsrc = ops.linear(query, self.in_proj_weight, self.in_proj_bias)
            kvsrc, offs = None, (0, d, 2 * d)
        elif kv_same:
            qsrc = ops.linear(query, self.in_proj_weight, self.in_proj_bias, rows=(0, d))
            kvsrc = ops.linear(key.to(cd), self.in_proj_weight, self.in_proj_bias, rows=(d, 3 * d))
            offs = (0, 0, d)
        else:
            qsrc = ops.linear(query, self.in_proj_weight, self.in_proj_bias, rows=(0, d))
            kvsrc = torch.cat([ops.linear(key.to(cd), self.in_proj_weight, self.in_proj_bias, rows=(d, 2 * d)),
                               ops.linear(value.to(cd), self.in_proj_weight, self.in_proj_bias, rows=(2 * d, 3 * d))], -1)
            offs = (0, 0, d)
        if isinstance(relation, FactoredRelation):
            fact = relation
            bank = relation.bank if relation.bank.dtype == cd else relation.bank.to(cd)
            group = relation.grad_group if bank is relation.bank else None
            rel = relation.take_projection(self)                                           # prefetched on the side stream?
            if rel is None:
                rel = ops.linear(bank, self.relation_in_proj.weight, group=group)           # [R, 2d]
                if ops.PROJ_RECOMPUTE:      # what the attention core needs to make this tensor again in its backward
                    rel._gtos_proj_src = (bank.detach(), ops.compute_weight(self.relation_in_proj.weight, cd))
        else:
            fact = None
            rel = ops.linear(relation.to(cd), self.relation_in_proj.weight)               # [n, n, B, 2d]
        p_w = self.dropout if (self.weights_dropout and self.training) else 0.0
        o, w = ops.attention_core(qsrc, kvsrc, offs, d, H, self.scaling, rel=rel, fact=fact,
                                  key_pad=key_padding_mask, attn_mask=attn_mask, p_drop=p_w, need_weights=need_weights)
        if not self.weights_dropout:
            o = torch.nn.functional.dropout(o, p=self.dropout, training=self.training)
        attn = ops.linear(o, self.out_proj.weight, self.out_proj.bias)
        return attn, w          # w: [tgt_len, src_len, bsz, heads] per-head weights, like the reference


class GraphTransformerLayer(nn.Module):
    def __init__(self, embed_dim, ff_embed_dim, num_heads, dropout, weights_dropout=True):
        super().__init__()
        self.self_attn = RelationMultiheadAttention(embed_dim, num_heads, dropout, weights_dropout)
        self.fc1 = nn.Linear(embed_dim, ff_embed_dim)
        self.fc2 = nn.Linear(ff_embed_dim, embed_dim)
        self.attn_layer_norm = nn.LayerNorm(embed_dim)
        self.ff_layer_norm = nn.LayerNorm(embed_dim)
        self.dropout = dropout
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.fc1.weight, std=0.02)
        nn.init.normal_(self.fc2.weight, std=0.02)
        nn.init.constant_(self.fc1.bias, 0.)
        nn.init.constant_(self.fc2.bias, 0.)

    def forward(self, x, relation, kv=None, self_padding_mask=None, self_attn_mask=None, need_weights=False):
        p = self.dropout if self.training else 0.0
        cd = self.self_attn.compute_dtype
        # xs: the residual stream (fp32 also in bf16 mode, ops.FP32_STREAM), x: the same values in the compute dtype for the GEMMs
        xs, x = ops.split_stream(x, cd)
        src = x if kv is None else kv
        a, self_attn = self.self_attn(query=x, key=src, value=src, relation=relation,
                                      key_padding_mask=self_padding_mask, attn_mask=self_attn_mask,
                                      need_weights=need_weights)
        ln = self.attn_layer_norm
        xs, x = ops.layer_norm_stream(xs, a, ln.weight, ln.bias, p, ln.eps, cd)
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, relu=True, p_drop=p)
        f = ops.linear(h, self.fc2.weight, self.fc2.bias)
        ln = self.ff_layer_norm
        xs, x = ops.layer_norm_stream(xs, f, ln.weight, ln.bias, p, ln.eps, cd)
        return ops.join_stream(xs, x), self_attn


class GraphTransformer(nn.Module):
    def __init__(self, layers, embed_dim, ff_embed_dim, num_heads, dropout, weights_dropout=True):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(layers):
            self.layers.append(GraphTransformerLayer(embed_dim, ff_embed_dim, num_heads, dropout, weights_dropout))

    def forward(self, x, relation, kv=None, self_padding_mask=None, self_attn_mask=None):
        if isinstance(relation, FactoredRelation):
            relation.prefetch_projections([layer.self_attn for layer in self.layers])
        for layer in self.layers:
            x, _ = layer(x, relation, kv, self_padding_mask, self_attn_mask)
        return x

    def get_attn_weights(self, x, relation, kv=None, self_padding_mask=None, self_attn_mask=None):
        attns = []
        for layer in self.layers:
            x, attn = layer(x, relation, kv, self_padding_mask, self_attn_mask, need_weights=True)
            attns.append(attn)
        return torch.stack(attns)


def set_compute_dtype(module, dtype):
    """fp32 (parity mode) or bf16 activations for every gtos_amd sub-module of ``module``."""
    for m in module.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dtype
    return module
