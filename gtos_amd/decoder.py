"""DecodeLayer / TokenGenerator (/root/reference/generator/decoder.py) on the gfx950 kernels.

The decoder's self- and cross-attention, LayerNorms, FFN and vocabulary projections run on the HIP kernels, and so does
the copy/generate mixture: one fused kernel evaluates the NLL of the target without materialising the [T,B,V+copies]
distribution (csrc/copy_nll.hip); inference gets the full log-likelihood row from a second kernel.
"""
import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .transformer import MultiheadAttention, Transformer


def _padded_linear(x, lin):
    """Linear whose input width is not a multiple of 8 (token_size 300 is; small test sizes may not be)."""
    pad = (-x.shape[-1]) % 8
    w = lin.weight
    if pad:
        x, w = F.pad(x, (0, pad)), F.pad(w, (0, pad))
    return ops.linear(x, w, lin.bias)


class TokenGenerator(nn.Module):
    def __init__(self, vocabs, embed_dim, token_size, dropout):
        super().__init__()
        self.alignment_layer = MultiheadAttention(embed_dim, 1, dropout, weights_dropout=False)
        self.alignment_layer_norm = nn.LayerNorm(embed_dim)
        self.transfer = nn.Linear(embed_dim, token_size)
        self.generator = nn.Linear(token_size, vocabs['predictable_token'].size)
        self.diverter = nn.Linear(token_size, 2)
        self.vocabs, self.dropout = vocabs, dropout
        self.reset_parameters()

    def reset_parameters(self):
        for l in (self.transfer, self.diverter, self.generator):
            nn.init.normal_(l.weight, std=0.02)
            nn.init.constant_(l.bias, 0.)

    def forward(self, outs, graph_state, graph_padding_mask, copy_seq, target=None, work=False, align_kv=None,
                tot_ext=None):
        p = self.dropout if self.training else 0.0
        cd = self.alignment_layer.compute_dtype
        outs_s, outs = ops.split_stream(outs, cd)      # (residual stream -- fp32 in bf16 mode --, GEMM operand)
        if align_kv is not None:       # incremental decoding: the graph states' K/V projection is cached
            x, alignment_weight = self.alignment_layer.attend_cached(outs, align_kv, key_padding_mask=graph_padding_mask,
                                                                     need_weights=True)
        else:
            x, alignment_weight = self.alignment_layer(outs, graph_state, graph_state,
                                                       key_padding_mask=graph_padding_mask, need_weights=True)
        ln = self.alignment_layer_norm
        outs = ops.layer_norm_stream(outs_s, x, ln.weight, ln.bias, p, ln.eps, cd)[1]
        hidden = torch.tanh(ops.linear(outs, self.transfer.weight, self.transfer.bias))
        hidden = F.dropout(hidden, p=self.dropout, training=self.training)
        logits = _padded_linear(hidden, self.generator)              # [T,B,V] vocabulary scores
        div = _padded_linear(hidden, self.diverter)                  # [T,B,2] generate-vs-copy scores
        # decoder.py:40-63 (softmaxes, gate * probabilities, scatter_add of the copy mass at the concepts' copy ids,
        # log(p + 1e-12), NLL gather) is ONE kernel: gtos_copy_nll_* for the loss, gtos_copy_ll_fwd for the full row
        if work:
            if tot_ext is None:
                tot_ext = 1 + int(copy_seq.max().item())
            return ops.copy_log_likelihood(logits, div, alignment_weight, copy_seq, tot_ext)
        pad = self.vocabs['predictable_token'].padding_idx
        return ops.copy_nll(logits, div, alignment_weight, copy_seq, target, pad).sum(0)


class DecodeLayer(nn.Module):
    def __init__(self, vocabs, inference_layers, embed_dim, ff_embed_dim, num_heads, token_size, rel_size, dropout):
        super().__init__()
        self.inference_core = Transformer(inference_layers, embed_dim, ff_embed_dim, num_heads, dropout, with_external=True)
        self.token_generator = TokenGenerator(vocabs, embed_dim, token_size, dropout)
        self.dropout, self.vocabs = dropout, vocabs

    def forward(self, probe, graph_state, snt_state, graph_padding_mask, snt_padding_mask, attn_mask,
                copy_seq, target=None, work=False):
        outs = F.dropout(probe, p=self.dropout, training=self.training)
        outs = self.inference_core(outs, kv=snt_state, self_padding_mask=snt_padding_mask, self_attn_mask=attn_mask,
                                   external_memories=graph_state, external_padding_mask=graph_padding_mask)
        if work:
            return self.token_generator(outs, graph_state, graph_padding_mask, copy_seq, work=True)
        token_loss = self.token_generator(outs, graph_state, graph_padding_mask, copy_seq, target=target, work=False)
        token_tot = snt_padding_mask.size(0) - snt_padding_mask.float().sum(0)
        return (token_loss / token_tot).mean()

    def step(self, probe, token_row, caches, mem):
        """Log-likelihoods of the next token, [1,N,V+ext] (decoder.py:85-87 with work=True), for N live hypotheses.
        token_row [1,N,d]: the newest sentence-encoder state; caches: per inference layer [t,N,2d] or None;
        mem: 'inf_ext_kv' (list), 'align_kv', 'graph_padding_mask', 'cp_seq', 'tot_ext' already gathered per hypothesis."""
        outs = probe
        new_caches = []
        for li, layer in enumerate(self.inference_core.layers):
            outs, c = layer.step(outs, token_row, None if caches is None else caches[li], mem['inf_ext_kv'][li],
                                 mem['graph_padding_mask'])
            new_caches.append(c)
        ll = self.token_generator(outs, None, mem['graph_padding_mask'], mem['cp_seq'], work=True,
                                  align_kv=mem['align_kv'], tot_ext=mem['tot_ext'])
        return ll, new_caches
