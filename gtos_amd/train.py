"""One training step of the gtos Generator on MI355X, data-parallel over RCCL.

Counterpart of the hot loop of /root/reference/generator/train.py:136-154: forward -> loss -> (abnormal-loss rule)
-> backward -> gradient averaging -> clip -> lr schedule -> Adam -> zero_grad.  Differences by design:
  * gradients live in one flat bucket cut into a few contiguous SEGMENTS in reverse execution order (decoder ->
    sentence encoder -> graph encoder -> input encoders); each segment is all-reduced (RCCL over xGMI, async) from inside
    backward as soon as the autograd graph has passed the segment's boundary, so only the last segment's collective is
    exposed -- instead of 182 blocking per-parameter all-reduces after backward (train.py:74-79);
  * the abnormal-loss skip (train.py:142-145) is decided COLLECTIVELY (max over ranks), because the reference's
    rank-local ``continue`` would desynchronise the collective sequence;
  * dropout streams are decorrelated across ranks by seeding the hash stream with base + rank after the (identical)
    weight initialisation, like train.py:113-116.
"""
import time

import torch
import torch.distributed as dist

from . import ops
from .flat import FlatParams, inverse_sqrt_lr

# segment 0 finishes first in backward.  The LAST segment is reduced after backward() returned (it holds everything whose
# gradient is only known to be complete then: the input encoders, the relation GRU's side-stream weight gradients).
GENERATOR_SEGMENTS = ("decoder.", "snt_encoder.", ("graph_encoder.", "probe_generator."))


def generator_segment_of(name):
    for i, pref in enumerate(GENERATOR_SEGMENTS):
        if name.startswith(pref):
            return i
    return len(GENERATOR_SEGMENTS)


class SegmentBoundaryFn(torch.autograd.Function):
    """Identity on ``tensors``; its backward runs once the gradients of ALL of them are complete, i.e. when every
    autograd node downstream of the boundary has run -- the moment the parameters of segments <= ``seg`` hold their final
    gradients.  ``sync.segment_ready(seg)`` then launches their all-reduce."""

    @staticmethod
    def forward(ctx, sync, seg, *tensors):
        ctx.sync, ctx.seg = sync, seg
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        ctx.sync.segment_ready(ctx.seg)
        return (None, None) + grads


class Trainer:
    def __init__(self, model, embed_dim, warmup_steps=2000, compute_dtype=torch.float32, world_size=1, rank=0,
                 segment_of="auto", base_seed=19940117, overlap=True, force_collectives=False):
        """``force_collectives``: issue the gradient collectives even with world_size == 1 (an initialised process group of one
        rank): the RCCL code path -- async all-reduce of bucket views from inside backward -- on a single GPU."""
        self.model = model
        self.embed_dim, self.warmup_steps = embed_dim, warmup_steps
        self.world_size, self.rank = world_size, rank
        if segment_of == "auto":
            names = [n for n, _ in model.named_parameters()]
            segment_of = generator_segment_of if any(n.startswith("graph_encoder.") for n in names) else None
        self.flat = FlatParams(model, mirror_dtype=compute_dtype, segment_of=segment_of)
        self.batches_acm, self.loss_acm, self.discarded = 0, 0.0, 0
        self.collective = world_size > 1 or force_collectives
        self.overlap = overlap and self.collective and len(self.flat.segments) > 1
        self._launched = 0              # segments [0, _launched) have their all-reduce in flight
        self._works = []
        self.comm_exposed_s = 0.0       # host time spent waiting for collectives after backward (gloo blocks here)
        self._comm_events = []          # (start, end) HIP events around the waits on the compute stream: the GPU-side exposed time
        if self.overlap and hasattr(model, "grad_sync"):
            model.grad_sync = self      # Generator.forward places the SegmentBoundaryFn markers
        if world_size > 1:
            ops.set_seed(base_seed + rank)

    # ---- boundary markers (called from the model's forward)
    def boundary(self, seg, *tensors):
        """Mark the inputs of segment ``seg`` (everything that consumes ``tensors`` belongs to segments <= seg)."""
        if not (self.overlap and torch.is_grad_enabled() and any(t.requires_grad for t in tensors)):
            return tensors
        return SegmentBoundaryFn.apply(self, seg, *tensors)

    def segment_ready(self, seg):
        """Segments 0..seg hold final gradients: put their all-reduce in flight (once, in segment order on every rank)."""
        while self._launched <= seg and self._launched < len(self.flat.segments):
            lo, hi = self.flat.segments[self._launched]
            self._launched += 1
            if hi > lo and self.collective:
                # async: enqueued on the process group's own stream after the work already on the current stream
                self._works.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def all_reduce_grads(self):
        """After backward(): launch what backward did not (at least the last segment) and wait for everything."""
        ops.join_side()                    # deferred side-stream gradient GEMMs (gru.py) land before the bucket is read
        self.segment_ready(len(self.flat.segments) - 1)
        timed = bool(self._works) and self.flat.grad.is_cuda
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        t0 = time.perf_counter()
        for w in self._works:
            w.wait()                       # nccl: the current stream waits for the collective; gloo: blocks the host
        self.comm_exposed_s += time.perf_counter() - t0
        if timed:
            ev[1].record()
            self._comm_events.append(ev)
            if len(self._comm_events) > 512:       # nobody is reading them: keep the list bounded
                del self._comm_events[:256]
        self._works, self._launched = [], 0

    def comm_exposed_ms(self, reset=True):
        """Milliseconds the compute stream spent stalled on gradient collectives since the last call (HIP events around the
        waits; needs the stream to be idle, i.e. call it after a synchronize)."""
        ms = sum(s.elapsed_time(e) for s, e in self._comm_events)
        if reset:
            self._comm_events = []
        return ms

    def step(self, batch):
        """Returns the loss value (float) of this step, or None when the batch was discarded."""
        loss = self.model(batch)
        loss_value = loss.item()
        abnormal = self.batches_acm > self.warmup_steps and loss_value > 5. * (self.loss_acm / self.batches_acm)
        if self.collective and self.batches_acm > self.warmup_steps:
            flag = torch.tensor([1.0 if abnormal else 0.0], device=loss.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            abnormal = bool(flag.item() > 0)
        if abnormal:
            self.discarded += 1
            return None
        self.loss_acm += loss_value
        self.batches_acm += 1
        loss.backward()
        self.all_reduce_grads()
        lr = inverse_sqrt_lr(self.embed_dim, self.batches_acm, self.warmup_steps)
        self.flat.step(lr, gscale=1.0 / self.world_size, max_norm=1.0)
        self.flat.zero_grad()
        return loss_value
