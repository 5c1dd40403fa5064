"""Generator: the model assembly of /root/reference/generator/generator.py (encode_step / forward) wired to the
HIP-backed modules.  Same constructor arguments and state_dict keys, so reference checkpoints load.

Train-mode ``encode_step`` hands the graph encoder the relation in FACTORED form (bank + type ids): it is the
exact same function as the reference's ``relation.index_select(0, idx).view(n,n,B,d)`` (generator.py:79) but the
[n,n,B,d] tensor is never built.  Eval mode aggregates alternative shortest paths with the gather-mean kernel
(generator.py:83-88).  Inference (work, generator.py:96-110) runs the beam search of gtos_amd.search over projected K/V caches
(decode_step_batched): the reference re-projects the whole prefix in every layer at every step.  ``decode_step`` keeps the
reference's signature (generator.py:119) for callers that bring their own search loop.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .encoder import TokenEncoder, RelationEncoder
from .decoder import DecodeLayer
from .transformer import Transformer, SinusoidalPositionalEmbedding, SelfAttentionMask
from .graph_transformer import GraphTransformer, set_compute_dtype
from .search import Beam, beam_search
from .vocab import lists_to_tensor, strings_to_char_tensor


class Generator(nn.Module):
    def __init__(self, vocabs, word_char_dim, word_dim, concept_char_dim, concept_dim, cnn_filters, char2word_dim,
                 char2concept_dim, rel_dim, rnn_hidden_size, rnn_num_layers, embed_dim, ff_embed_dim, num_heads,
                 dropout, snt_layers, graph_layers, inference_layers, pretrained_file, device, depth_size=32,
                 factored_relation=True):
        super().__init__()
        self.vocabs = vocabs
        self.concept_encoder = TokenEncoder(vocabs['concept'], vocabs['concept_char'], concept_char_dim, concept_dim,
                                            embed_dim, cnn_filters, char2concept_dim, dropout, pretrained_file)
        self.relation_encoder = RelationEncoder(vocabs['relation'], rel_dim, embed_dim, rnn_hidden_size,
                                                rnn_num_layers, dropout)
        self.token_encoder = TokenEncoder(vocabs['token'], vocabs['token_char'], word_char_dim, word_dim, embed_dim,
                                          cnn_filters, char2word_dim, dropout, pretrained_file)
        self.graph_encoder = GraphTransformer(graph_layers, embed_dim, ff_embed_dim, num_heads, dropout)
        self.snt_encoder = Transformer(snt_layers, embed_dim, ff_embed_dim, num_heads, dropout, with_external=True)
        self.embed_dim = embed_dim
        self.embed_scale = math.sqrt(embed_dim)
        self.token_position = SinusoidalPositionalEmbedding(embed_dim, device)
        self.concept_depth = nn.Embedding(depth_size, embed_dim)     # 32 (generator) / 256 (translator/generator.py:39)
        self.token_embed_layer_norm = nn.LayerNorm(embed_dim)
        self.concept_embed_layer_norm = nn.LayerNorm(embed_dim)
        self.self_attn_mask = SelfAttentionMask(device)
        self.decoder = DecodeLayer(vocabs, inference_layers, embed_dim, ff_embed_dim, num_heads, concept_dim, rel_dim, dropout)
        self.dropout = dropout
        self.probe_generator = nn.Linear(embed_dim, embed_dim)
        self.device = device
        self.factored_relation = factored_relation
        self.compute_dtype = torch.float32
        self.grad_sync = None           # train.Trainer (data parallel): places the gradient-segment boundary markers
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.probe_generator.weight, std=0.02)
        nn.init.constant_(self.probe_generator.bias, 0.)
        nn.init.constant_(self.concept_depth.weight, 0.)

    def set_compute_dtype(self, dtype):
        return set_compute_dtype(self, dtype)

    def _concepts(self, inp):
        c = self.embed_scale * self.concept_encoder(inp['concept'], inp['concept_char']) \
            + self.concept_depth(inp['concept_depth']).to(self.compute_dtype)
        ln = self.concept_embed_layer_norm
        c = ops.layer_norm_residual(c, None, ln.weight, ln.bias, 0.0, ln.eps)
        return c, torch.eq(inp['concept'], self.vocabs['concept'].padding_idx)

    def encode_step(self, inp, train=True):
        ops.refresh_side_policy(inp['concept'].device)      # auxiliary stream yes / no for this step (memory: ops.SIDE_STREAMS)
        if 'relation_graphs' in inp:           # a loader batch whose relation section was left to this device (index_prep="device_all")
            from .data import complete_on_device
            complete_on_device(inp, inp['concept'].device)
        concept_repr, concept_mask = self._concepts(inp)
        with ops._Timed("relation_encoder_fwd"):
            bank = self.relation_encoder(inp['relation_bank'], inp['relation_length'], trie=inp.get('relation_trie'))   # [R, d]
        if train and self.grad_sync is not None:
            # everything downstream of these two belongs to gradient segments <= 2 (graph encoder, probe, decoders)
            concept_repr, bank = self.grad_sync.boundary(2, concept_repr, bank)
        if train:
            if self.factored_relation:
                relation = ops.FactoredRelation(bank, inp['relation'], index=inp.get('relation_index'))
            else:
                relation = bank.index_select(0, inp['relation'].reshape(-1)).view(*inp['relation'].size(), -1)
        elif inp['relation'].dim() == 4 and self.factored_relation:
            # generator flavour, eval: [n,n,B,K] alternative paths averaged (generator.py:83-88) -- as a derived bank of the
            # distinct K-tuples + per-pair ids, so the [n,n,B,d] tensor and its projection are never built
            relation = ops.factored_eval_relation(bank.detach(), inp['relation'])
        elif inp['relation'].dim() == 3 and self.factored_relation:
            # translator flavour: one path per pair, the plain lookup (translator/generator.py:73), no autograd graph
            relation = ops.FactoredRelation(bank.detach(), inp['relation'], index=inp.get('relation_index'))
        else:
            relation = ops.relation_gather_mean(bank.detach(), inp['relation'], zero_row0=inp['relation'].dim() == 4)
        with ops._Timed("graph_encoder_fwd"):
            concept_repr = self.graph_encoder(concept_repr, relation, self_padding_mask=concept_mask)
        ops.note_memory(inp['concept'].device)               # a high-water point of the step (ops.refresh_side_policy)
        # (bf16 mode: the encoder returns its fp32 residual stream carrying the bf16 twin; everything downstream is a GEMM operand)
        concept_repr = ops.split_stream(concept_repr, self.compute_dtype)[1]
        probe = torch.tanh(ops.linear(concept_repr[:1], self.probe_generator.weight, self.probe_generator.bias))
        return concept_repr[1:], concept_mask[1:], probe

    def encoder_attn(self, inp):
        with torch.no_grad():
            if 'relation_graphs' in inp:
                from .data import complete_on_device
                complete_on_device(inp, inp['concept'].device)
            concept_repr, concept_mask = self._concepts(inp)
            bank = self.relation_encoder(inp['relation_bank'], inp['relation_length'], trie=inp.get('relation_trie'))
            relation = (ops.factored_eval_relation(bank, inp['relation']) if self.factored_relation
                        else ops.relation_gather_mean(bank, inp['relation'], zero_row0=True))
            return self.graph_encoder.get_attn_weights(concept_repr, relation, self_padding_mask=concept_mask)

    def forward(self, data):
        concept_repr, concept_mask, probe = self.encode_step(data)
        pos = self.token_position(data['token_in']).to(self.compute_dtype)
        token_repr = self.embed_scale * self.token_encoder(data['token_in'], data['token_char_in']) + pos
        ln = self.token_embed_layer_norm
        token_repr = ops.layer_norm_residual(token_repr, None, ln.weight, ln.bias, 0.0, ln.eps)
        token_repr = F.dropout(token_repr, p=self.dropout, training=self.training)
        token_mask = torch.eq(data['token_in'], self.vocabs['token'].padding_idx)
        attn_mask = self.self_attn_mask(data['token_in'].size(0))
        concept_repr = concept_repr.contiguous()
        gs = self.grad_sync
        if gs is not None:          # downstream: sentence encoder (segment 1) and decoder (segment 0)
            token_repr, concept_repr, probe = gs.boundary(1, token_repr, concept_repr, probe)
        token_repr = self.snt_encoder(token_repr, self_padding_mask=token_mask, self_attn_mask=attn_mask,
                                      external_memories=concept_repr, external_padding_mask=concept_mask)
        token_repr = ops.split_stream(token_repr, self.compute_dtype)[1]
        probe = probe.expand_as(token_repr)
        if gs is not None:          # downstream: the decoder only
            probe, concept_repr, token_repr = gs.boundary(0, probe, concept_repr, token_repr)
        out = self.decoder(probe, concept_repr, token_repr, concept_mask, token_mask, attn_mask,
                           data['cp_seq'], target=data['token_out'])
        ops.note_memory(token_repr.device)                   # end of the forward pass: everything saved for backward is alive
        return out

    # ------------------------------------------------------------------------------------------------ inference
    def work(self, data, beam_size, max_time_step, min_time_step=1):
        """Beam search for every graph of the batch (generator.py:96-110).  Returns the finished Beam objects
        (``beam.get_k_best(k, alpha)``)."""
        with torch.no_grad():
            concept_repr, concept_mask, probe = self.encode_step(data, train=False)
            concept_repr = concept_repr.contiguous()
            dec = self.decoder
            memory = {
                'probe': probe,
                'graph_padding_mask': concept_mask,
                'cp_seq': data['cp_seq'],
                'tot_ext': 1 + int(data['cp_seq'].max().item()),
                'local_idx2token': data['local_idx2token'],
                # K/V projections of the graph states for every cross-attention that reads them: computed once
                'snt_ext_kv': [l.external_attn.project_kv(concept_repr) for l in self.snt_encoder.layers],
                'inf_ext_kv': [l.external_attn.project_kv(concept_repr) for l in dec.inference_core.layers],
                'align_kv': dec.token_generator.alignment_layer.project_kv(concept_repr),
            }
            beams = [Beam(beam_size, min_time_step, max_time_step) for _ in range(concept_repr.size(1))]
            beam_search(self, beams, memory)
        return beams

    def prepare_incremental_input(self, step_seq):
        """step_seq: one single-token list per live hypothesis (generator.py:112-117).  Token id and character row of a
        string are cached: beam search feeds the same few thousand strings over and over."""
        if any(len(x) != 1 for x in step_seq):
            token = lists_to_tensor(step_seq, self.vocabs['token'])
            token_char = strings_to_char_tensor(step_seq, self.vocabs['token_char'])
            return token.to(self.device), token_char.to(self.device)
        cache = self.__dict__.setdefault('_step_input_cache', {})
        ids, rows = [], []
        for (w,) in step_seq:
            hit = cache.get(w)
            if hit is None:
                hit = (self.vocabs['token'].token2idx(w), strings_to_char_tensor([[w]], self.vocabs['token_char'])[0, 0].tolist())
                cache[w] = hit
            ids.append(hit[0])
            rows.append(hit[1])
        token = torch.tensor([ids], dtype=torch.int64)
        token_char = torch.tensor([rows], dtype=torch.int64)
        return token.to(self.device), token_char.to(self.device)

    def decode_step_batched(self, tokens, state, memory, beam_of_hyp, offset, topk):
        """One step for N live hypotheses of ALL beams (what gtos_amd.search.beam_search drives).  tokens: their last token
        strings; state: None or {'snt': [cache per sentence-encoder layer], 'inf': [cache per inference layer]}, every cache
        [t,N,2d]; memory: per GRAPH (``work``); beam_of_hyp [N]: graph index of each hypothesis.  Returns (state grown by one
        row, per hypothesis the top-k [(token string, log-likelihood)])."""
        inp = self.prepare_incremental_input([[t] for t in tokens])
        sel = lambda v: v.index_select(1, beam_of_hyp)
        owners = beam_of_hyp.tolist()
        mem = {'graph_padding_mask': sel(memory['graph_padding_mask']), 'cp_seq': sel(memory['cp_seq']), 'probe': sel(memory['probe']),
               'tot_ext': memory['tot_ext'], 'inf_ext_kv': [sel(v) for v in memory['inf_ext_kv']],
               'snt_ext_kv': [sel(v) for v in memory['snt_ext_kv']], 'align_kv': sel(memory['align_kv']),
               'local_idx2token': [memory['local_idx2token'][bi] for bi in owners]}
        snt, inf, results = self._decode_core(inp, None if state is None else state['snt'], None if state is None else state['inf'],
                                              mem, offset, topk)
        return {'snt': snt, 'inf': inf}, results

    def _decode_core(self, inp, snt_state, inf_state, mem, offset, topk):
        """inp = (step_token [1,N], step_token_char [1,N,C]); snt_state / inf_state: per layer [t,N,2d] or None; mem: everything
        already per hypothesis.  -> (new sentence-encoder caches, new inference caches, top-k results)."""
        step_token, step_token_char = inp
        pos = self.token_position(step_token, offset).to(self.compute_dtype)
        x = self.embed_scale * self.token_encoder(step_token, step_token_char) + pos
        ln = self.token_embed_layer_norm
        x = ops.layer_norm_residual(x, None, ln.weight, ln.bias, 0.0, ln.eps)
        snt_caches = []
        for li, layer in enumerate(self.snt_encoder.layers):
            x, c = layer.step(x, x, None if snt_state is None else snt_state[li], mem['snt_ext_kv'][li], mem['graph_padding_mask'])
            snt_caches.append(c)
        ll, inf_caches = self.decoder.step(mem['probe'], x, inf_state, mem)
        topk_scores, topk_token = torch.topk(ll.squeeze(0).float(), topk, 1)
        vocab = self.vocabs['predictable_token']
        results = []
        for s, t, local in zip(topk_scores.tolist(), topk_token.tolist(), mem['local_idx2token']):
            results.append([(local[i] if i in local else vocab.idx2token(i), sc) for sc, i in zip(s, t)])
        return snt_caches, inf_caches, results

    # ---- the reference's decoding interface (generator/generator.py:96-167, generator/search.py:113-166)
    def reference_memory(self, data):
        """``mem_dict`` as the reference's ``work`` builds it (generator.py:100-104: graph_state, graph_padding_mask, probe,
        local_idx2token, cp_seq -- one column / entry per graph), extended by the K/V projections of the graph states for every
        cross-attention that reads them.  The reference's ``search_by_batch`` treats the dictionary generically (tensors are
        ``index_select``-ed along dim 1 per live hypothesis, lists are indexed), so the extra entries travel with the rest and
        ``decode_step`` never re-projects the graph."""
        with torch.no_grad():
            concept_repr, concept_mask, probe = self.encode_step(data, train=False)
            concept_repr = concept_repr.contiguous()
            mem = {'graph_state': concept_repr, 'graph_padding_mask': concept_mask, 'probe': probe,
                   'local_idx2token': data['local_idx2token'], 'cp_seq': data['cp_seq']}
            for li, l in enumerate(self.snt_encoder.layers):
                mem['snt_ext_kv_%d' % li] = l.external_attn.project_kv(concept_repr)
            for li, l in enumerate(self.decoder.inference_core.layers):
                mem['inf_ext_kv_%d' % li] = l.external_attn.project_kv(concept_repr)
            mem['align_kv'] = self.decoder.token_generator.alignment_layer.project_kv(concept_repr)
        return mem

    def decode_step(self, inp, state_dict, mem_dict, offset, topk):
        """The reference's signature and data flow (generator/generator.py:119-167): inp = (step_token [1,N], step_token_char
        [1,N,C]) from ``prepare_incremental_input``; ``mem_dict``: the reference's keys, every entry already gathered per live
        hypothesis by the caller; ``state_dict``: {} at the first step, afterwards what the previous call returned, split / joined
        along dim 1 by the caller.  Returns (new_state_dict, per hypothesis the top-k [(token string, log-likelihood)]).

        The state carried between steps is the PROJECTED K/V rows of every self-attention ('snt_kv_i', 'inf_kv_i', [t,N,2d]) in
        place of the reference's unprojected 'token_repr_i' / 'token_state' histories -- the same information, so the reference
        re-projecting the whole prefix in every layer at every step is not needed; any caller that treats the dictionary
        opaquely (the reference's Beam / search_by_batch do) works unchanged.  ``mem_dict`` entries made by ``reference_memory``
        are used as they are; with only the reference's own five keys the graph projections are computed here."""
        with torch.no_grad():
            dec = self.decoder
            n_snt, n_inf = len(self.snt_encoder.layers), len(dec.inference_core.layers)
            graph = mem_dict.get('graph_state')

            def kv(key, attn):
                v = mem_dict.get(key)
                return v if v is not None else attn.project_kv(graph)
            cp_seq = mem_dict['cp_seq']
            mem = {'graph_padding_mask': mem_dict['graph_padding_mask'], 'cp_seq': cp_seq, 'probe': mem_dict['probe'],
                   'local_idx2token': mem_dict['local_idx2token'],
                   'tot_ext': 1 + int(cp_seq.max().item()),          # like decoder.py:48 on the live hypotheses' copy ids
                   'snt_ext_kv': [kv('snt_ext_kv_%d' % i, l.external_attn) for i, l in enumerate(self.snt_encoder.layers)],
                   'inf_ext_kv': [kv('inf_ext_kv_%d' % i, l.external_attn) for i, l in enumerate(dec.inference_core.layers)],
                   'align_kv': kv('align_kv', dec.token_generator.alignment_layer)}
            first = not state_dict
            snt_state = None if first else [state_dict['snt_kv_%d' % i] for i in range(n_snt)]
            inf_state = None if first else [state_dict['inf_kv_%d' % i] for i in range(n_inf)]
            snt, inf, results = self._decode_core(inp, snt_state, inf_state, mem, offset, topk)
            new_state = {'snt_kv_%d' % i: c for i, c in enumerate(snt)}
            new_state.update({'inf_kv_%d' % i: c for i, c in enumerate(inf)})
        return new_state, results
