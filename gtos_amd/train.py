"""One training step of the gtos Generator on MI355X, data-parallel over RCCL.

Counterpart of the hot loop of /root/reference/generator/train.py:136-154: forward -> loss -> (abnormal-loss rule)
-> backward -> gradient averaging -> clip -> lr schedule -> Adam -> zero_grad.  Differences by design:
  * gradients live in one flat bucket cut into a few contiguous SEGMENTS in reverse execution order (decoder ->
    sentence encoder -> graph encoder -> input encoders); each segment is all-reduced (RCCL over xGMI, async) from inside
    backward as soon as the autograd graph has passed the segment's boundary, so only the last segment's collective is
    exposed -- instead of 182 blocking per-parameter all-reduces after backward (train.py:74-79);
  * the abnormal-loss skip (train.py:142-145) and the lr schedule are evaluated ON THE DEVICE (gtos_step_control): the host
    never reads the loss between forward and backward, so the launch queue stays full (the reference's ``loss.item()`` in
    front of ``backward`` drains it every step).  A discarded batch still runs its backward; the Adam kernel reads the skip
    flag and leaves parameters and moments untouched.  Data parallel, the decision is COLLECTIVE (MAX all-reduce of the
    flag on the device), because the reference's rank-local ``continue`` would desynchronise the collective sequence;
  * dropout streams are decorrelated across ranks by seeding the hash stream with base + rank after the (identical)
    weight initialisation, like train.py:113-116.
"""
import os
import time

import torch
import torch.distributed as dist

from . import ops
from ._lib import call, ptr, stream
from .flat import FlatParams

# segment 0 finishes first in backward.  The LAST segment is reduced after backward() returned (it holds everything whose
# gradient is only known to be complete then: the input encoders, the relation GRU's side-stream weight gradients).
GENERATOR_SEGMENTS = ("decoder.", "snt_encoder.", ("graph_encoder.", "probe_generator."))
# steps the host may queue ahead of the device (0: unbounded, as up to round 4); see Trainer.step
MAX_STEPS_AHEAD = int(os.environ.get("GTOS_MAX_STEPS_AHEAD", "2"))


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
        dev = self.flat.param.device
        self._state = torch.zeros(3, dtype=torch.float64, device=dev)       # {loss_acm, batches_acm, discarded}, device-resident
        self._flag = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ctl = torch.zeros(2, dtype=torch.float32, device=dev)         # {lr of this step, skip}
        self.steps_issued = 0           # host-side count of step() calls: an upper bound of batches_acm (no device read needed)
        self._counters, self._counters_at = None, -1
        self.collective = world_size > 1 or force_collectives
        self.overlap = overlap and self.collective and len(self.flat.segments) > 1
        self._launched = 0              # segments [0, _launched) have their all-reduce in flight
        self._works = []
        self.comm_exposed_s = 0.0       # host time spent waiting for collectives after backward (gloo blocks here)
        self._comm_events = []          # (start, end) HIP events around the waits on the compute stream: the GPU-side exposed time
        self._done_events = [] if (dev.type == "cuda" and MAX_STEPS_AHEAD > 0) else None    # end-of-step events of the steps in flight
        if self.overlap and hasattr(model, "grad_sync"):
            model.grad_sync = self      # Generator.forward places the SegmentBoundaryFn markers
        if world_size > 1:
            # Replicas must start identical.  The reference relies on every rank seeding torch alike before it builds the model
            # (generator/train.py:98-100); here rank 0's flat parameter buffer is broadcast once (one collective over the whole
            # model, like DistributedDataParallel does at construction), so a rank that was built differently -- another seed, a
            # checkpoint loaded on rank 0 only -- cannot drift silently.
            dist.broadcast(self.flat.param, src=0)
            self.flat.sync_mirror()
            ops.set_seed(base_seed + rank)

    # ---- boundary markers (called from the model's forward)
    def boundary(self, seg, *tensors):
        """Mark the inputs of segment ``seg`` (everything that consumes ``tensors`` belongs to segments <= seg)."""
        if not (self.overlap and torch.is_grad_enabled() and any(t.requires_grad for t in tensors)):
            return tensors
        return SegmentBoundaryFn.apply(self, seg, *tensors)

    def segment_ready(self, seg):
        """Segments 0..seg hold final gradients: put their all-reduce in flight (once, in segment order on every rank)."""
        if self._launched <= seg and self.collective:
            ops.flush_dw(self.flat.grad.device)        # the small layers' batched weight gradients (ops.DW_BATCH) belong to the segment
        while self._launched <= seg and self._launched < len(self.flat.segments):
            lo, hi = self.flat.segments[self._launched]
            self._launched += 1
            if hi > lo and self.collective:
                # async: enqueued on the process group's own stream after the work already on the current stream -- and, when a
                # deferred side-stream product still writes into the bucket (ops.GradAccumGroup.finish_dw), behind that stream too,
                # without stalling this one (ops.behind_side)
                with ops.behind_side(self.flat.grad.device):
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

    # ---- counters of the reference loop, kept on the device.  READING THEM SYNCHRONISES (one device -> host copy): use
    # ``steps_issued`` (host-side, no read) for per-step cadence checks like the reference's ``batches_acm % print_every`` and read
    # the counters only when something is printed / evaluated.  One read serves all three: ``counters()`` copies the state once
    # per issued step and the properties share that copy, so reference-style logging that touches batches_acm, loss_acm and
    # discarded in one iteration costs one synchronisation, not three.  (A discarded batch still ran its backward and its
    # gradient all-reduce -- only the Adam kernel is skipped)
    def counters(self):
        """(loss_acm, batches_acm, discarded) as of the last issued step: ONE host read, cached until the next ``step()``."""
        if self._counters_at != self.steps_issued or self._counters is None:
            la, ba, di = self._state.tolist()
            self._counters, self._counters_at = (float(la), int(ba), int(di)), self.steps_issued
        return self._counters

    @property
    def batches_acm(self):
        return self.counters()[1]

    @property
    def loss_acm(self):
        return self.counters()[0]

    @property
    def discarded(self):
        return self.counters()[2]

    def set_counters(self, batches_acm, loss_acm, discarded=0):
        """Resume from a checkpoint (train.py keeps batches_acm / loss_acm across restarts)."""
        self._state.copy_(torch.tensor([float(loss_acm), float(batches_acm), float(discarded)], dtype=torch.float64))
        self.steps_issued = int(batches_acm) + int(discarded)
        self._counters = None

    def _control(self, phase, loss):
        """gtos_step_control: phase 0 writes this rank's abnormal-loss flag, phase 1 applies the (all-reduced) flag to the
        counters and writes {lr, skip} for the Adam kernel."""
        call("gtos_step_control", phase, ptr(loss), ptr(self._state), ptr(self._flag), int(self.warmup_steps), int(self.embed_dim),
             ptr(self._ctl), stream())

    def step(self, batch, sync=True):
        """One training step.  ``sync=True`` returns the loss value (float), or None when the batch was discarded -- one host
        read AFTER every launch of the step has been queued.  ``sync=False`` returns a PendingLoss (``.value()`` reads it
        later): the host runs ahead into the next step, which is what the launch-bound small configurations need."""
        if self._done_events is not None:
            # Bounded run-ahead.  Nothing in a step reads the device any more, so the host could queue many steps ahead of it -- and the
            # caching allocator can only recycle a block that was freed on one stream while another still used it once the DEVICE has
            # passed that point: an unbounded run-ahead keeps several steps' worth of such blocks in limbo and mallocs new ones (C2:
            # 237 GB reserved for 44 GB allocated).  Waiting here for the end of step k - MAX_STEPS_AHEAD costs nothing while the device
            # is the bottleneck and never triggers when the host is (the event has long completed).
            while len(self._done_events) >= MAX_STEPS_AHEAD:
                self._done_events.pop(0).synchronize()
        loss = self.model(batch)
        loss = loss if loss.dtype == torch.float32 else loss.float()
        self._control(0, loss.detach())
        if self.collective and self.steps_issued > self.warmup_steps:       # batches_acm <= steps_issued: never needed earlier
            dist.all_reduce(self._flag, op=dist.ReduceOp.MAX)
        self._control(1, loss.detach())
        self.steps_issued += 1
        loss.backward()
        self.all_reduce_grads()
        self.flat.step(None, gscale=1.0 / self.world_size, max_norm=1.0, ctl=self._ctl)
        self.flat.zero_grad()
        res = PendingLoss(torch.cat([loss.detach().reshape(1), self._flag]))
        if self._done_events is not None:
            ev = torch.cuda.Event()
            ev.record()
            self._done_events.append(ev)
        return res.value() if sync else res


class GraphedStep:
    """A training step captured once as a hipGraph and replayed: for the configurations whose step is a thousand small launches (C1:
    9.7 ms of Python-paced launches around ~3 ms of kernels), when consecutive batches have the SAME shapes -- the captured launch
    plan, every tensor size and every trie level count belongs to the batch it was captured on.

        gs = GraphedStep(trainer, batch)        # warm-up steps + capture (a few real steps on ``batch``)
        pending = gs()                          # one replay = one training step on ``batch``; PendingLoss like Trainer.step(sync=False)

    New token / concept contents may be copied INTO the captured batch's tensors between replays; a batch with another relation
    structure (other trie level sizes, another number of distinct paths) is another launch plan and needs its own capture.

    Dropout: kernel arguments are frozen by the capture, so the library's kernels fold a device-side epoch into their seeds
    (ops.set_seed_epoch) which the captured step increments first thing; torch's own generator is graph-safe by itself.
    Limits: one process (world_size 1: the gradient collectives' host-side bookkeeping is not captured); a batch that carries its path
    trie (``batch['relation_trie']``: the RelationEncoder then takes the sort order and the step sizes from it -- since round 5 also with
    the reference's dropout semantics, whose packed sequence used to be sized by a host read; a bare bank still costs that read and cannot
    be captured) and its relation tensors already built (device builders run before, not inside, the capture).
    Robustness (ROCm 7.2, torch 2.10): C1 captures and replays correctly (equal to the eager steps at dropout 0).
    At C2, where the auxiliary stream is forked in three places, ``hipStreamEndCapture`` crashed inside the runtime, and with every
    use of that stream switched off a replay hung; with one use it worked and bought nothing (the step is not launch-bound).  Use it
    for launch-bound configurations."""

    def __init__(self, trainer, batch, warmup=3):
        if trainer.collective:
            raise ValueError("GraphedStep: single-process training only")
        dev = trainer.flat.grad.device
        self.trainer, self.batch, self._dev = trainer, batch, dev
        self.epoch = torch.zeros((), dtype=torch.int64, device=dev)
        ops.set_seed_epoch(self.epoch)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):           # torch's capture protocol: a few eager steps on a side stream first (workspaces,
            for _ in range(warmup):             # cuBLAS-style lazy state, the allocator's pools): REAL training steps on ``batch``
                self.epoch.add_(1)
                self._body()
                trainer.steps_issued += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.epoch.add_(1)
            self._out = self._body()
        torch.cuda.synchronize(dev)
        self.replays = 0

    def _body(self):
        t = self.trainer
        # fork the auxiliary stream into the capture before anything else and join it at the end: the step makes the main stream wait
        # for it in places where, on small inputs, it was never given work -- outside a capture a no-op, inside one a dependency on a
        # stream that is not being captured
        main, side = torch.cuda.current_stream(self._dev), ops.side_stream(self._dev)
        side.wait_stream(main)
        loss = t.model(self.batch)
        loss = loss if loss.dtype == torch.float32 else loss.float()
        t._control(0, loss.detach())
        t._control(1, loss.detach())
        loss.backward()
        ops.join_side()
        t.flat.step(None, gscale=1.0, max_norm=1.0, ctl=t._ctl)
        t.flat.zero_grad()
        main.wait_stream(side)
        return torch.cat([loss.detach().reshape(1), t._flag])

    def __call__(self):
        self.graph.replay()
        self.replays += 1
        self.trainer.steps_issued += 1
        return PendingLoss(self._out.clone())

    def close(self):
        ops.set_seed_epoch(None)


class PendingLoss:
    """Loss and skip flag of a step, still on the device."""

    def __init__(self, pair):
        self._pair, self._val = pair, False

    def value(self):
        """float, or None when the batch was discarded by the abnormal-loss rule."""
        if self._val is False:
            v, f = self._pair.tolist()
            self._val = None if f > 0 else v
            self._pair = None
        return self._val
