"""Batched beam search over the incremental decoder.

Selection rules of /root/reference/generator/search.py (Hypothesis / Beam.update / Beam.completed / get_k_best /
search_by_batch): per sentence, the top-k continuations of every live hypothesis are pooled, <UNK> continuations score
-inf, the pool is sorted by accumulated log-likelihood (stable, descending) and cut to ``beam_size - #finished``;
a continuation ending in <END> finishes its hypothesis (kept only if it has at least ``min_time_step`` tokens); a beam
stops when it holds ``beam_size`` finished hypotheses or after ``max_time_step`` steps; the final ranking divides the
score by ``(1 + len(seq)) ** alpha``.

What differs is the machinery: hypotheses carry no tensors.  The decoder state of ALL live hypotheses of ALL sentences
is one set of K/V-cache tensors ``[t, N, 2d]`` (gtos_amd.generator.Generator.decode_step_batched); a step returns, per beam,
the parent index of every surviving hypothesis, and the caches are re-gathered with ONE index_select per tensor
instead of being split into per-hypothesis slices and concatenated again.
"""
import torch

from .vocab import END, UNK, STR


class Hypothesis(object):
    __slots__ = ("seq", "score")

    def __init__(self, seq, score):
        self.seq = seq          # token strings, starting with <STR>
        self.score = score      # accumulated log-likelihood

    def is_completed(self):
        return self.seq[-1] == END

    def __len__(self):
        return len(self.seq)


class Beam(object):
    """The search frontier of one sentence."""

    def __init__(self, beam_size, min_time_step, max_time_step):
        self.beam_size, self.min_time_step, self.max_time_step = beam_size, min_time_step, max_time_step
        self.hypotheses = [Hypothesis([STR], 0.)]
        self.completed_hypotheses = []
        self.steps = 0

    def advance(self, last_steps):
        """last_steps[h] = [(token, log-likelihood), ...] for live hypothesis h.  Returns the parent index (into the
        old ``hypotheses``) of every hypothesis that stays alive, in their new order."""
        pool = []
        for parent, steps in enumerate(last_steps):
            base = self.hypotheses[parent].score
            for token, ll in steps:
                pool.append((parent, token, float('-inf') if token == UNK else base + ll))
        pool.sort(key=lambda c: c[2], reverse=True)                     # stable: ties keep (parent, rank) order
        pool = pool[:self.beam_size - len(self.completed_hypotheses)]
        alive, parents = [], []
        for parent, token, score in pool:
            hyp = Hypothesis(self.hypotheses[parent].seq + [token], score)
            if hyp.is_completed():
                if len(hyp) - 2 >= self.min_time_step:
                    self.completed_hypotheses.append(hyp)
            else:
                alive.append(hyp)
                parents.append(parent)
        self.hypotheses = alive
        self.steps += 1
        return parents

    def completed(self):
        return len(self.completed_hypotheses) >= self.beam_size or self.steps >= self.max_time_step

    def get_k_best(self, k, alpha):
        if not self.completed_hypotheses:
            self.completed_hypotheses = self.hypotheses
        self.completed_hypotheses.sort(key=lambda h: h.score / ((1 + len(h.seq)) ** alpha), reverse=True)
        return self.completed_hypotheses[:k]


def beam_search(model, beams, memory):
    """Runs all beams to completion.  ``model.decode_step_batched(tokens, state, memory, beam_of_hyp, offset, topk)`` ->
    (state, results); ``state`` is opaque here except that every tensor in it has the hypothesis axis at dim 1."""
    device = memory['probe'].device
    state = None
    while True:
        owners, tokens = [], []
        for bi, beam in enumerate(beams):
            if not beam.completed():
                for hyp in beam.hypotheses:
                    owners.append(bi)
                    tokens.append(hyp.seq[-1])
                    offset = len(hyp.seq) - 1
        if not owners:
            break
        beam_of_hyp = torch.tensor(owners, dtype=torch.int64, device=device)
        state, results = model.decode_step_batched(tokens, state, memory, beam_of_hyp, offset, beams[0].beam_size)
        # hand every beam its slice of the results; collect the flat parent index of each survivor
        keep, pos = [], 0
        for bi, beam in enumerate(beams):
            if beam.completed():
                continue
            n = len(beam.hypotheses)
            parents = beam.advance(results[pos:pos + n])
            if not beam.completed():
                keep.extend(pos + p for p in parents)
            pos += n
        if not keep:
            break
        idx = torch.tensor(keep, dtype=torch.int64, device=device)
        state = {k: [c.index_select(1, idx) for c in v] for k, v in state.items()}
    return beams
