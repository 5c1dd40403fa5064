"""Data-parallel path on CPU: world_size 2, gloo.  The flat gradient bucket reduced with ONE all-reduce must equal
the reference's per-parameter all_reduce(SUM)/world_size (generator/train.py:74-79), the ranks must draw different
synthetic graphs, and the collective abnormal-loss decision must agree on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gtos_amd.flat import FlatParams
    from gtos_amd import synth
    torch.manual_seed(19940117)                       # identical init on every rank (train.py:98-100)
    model = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.LayerNorm(7), torch.nn.Linear(7, 3))
    ref = [p.detach().clone() for p in model.parameters()]
    flat = FlatParams(model)                          # CPU: views only, no kernels
    assert all(torch.equal(p.detach(), r) for p, r in zip(model.parameters(), ref))
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 13)
    model(x).pow(2).sum().backward()                  # autograd accumulates into the flat bucket views
    per_param = [p.grad.clone() for p in model.parameters()]
    assert float(flat.grad.abs().sum()) > 0
    for g in per_param:                               # the reference way
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g /= world
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)  # ours: one bucket
    ok = all(torch.allclose(p.grad / world, g, rtol=1e-6, atol=1e-7) for p, g in zip(model.parameters(), per_param))
    b, _ = synth.make_batch(5, 2, 6, 5, first_graph=rank * 2)
    sig = torch.tensor([float(b["concept"].sum())])
    sigs = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(sigs, sig)
    flag = torch.tensor([1.0 if rank == 1 else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    q.put((rank, ok, [float(s) for s in sigs], float(flag)))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_per_parameter():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, sigs, flag in res:
        assert ok, "flat all-reduce differs from per-parameter all-reduce on rank %d" % rank
        assert sigs[0] != sigs[1], "ranks drew identical synthetic graphs"
        assert flag == 1.0


# ---------------------------------------------------------------------------------------------------------------------
# Trainer.step itself with world_size 2 (SURVEY section 7 test vi): the HIP optimizer kernels are replaced by the oracle's
# CPU Adam / clip (same arithmetic), everything else -- segment layout, boundary markers, async all-reduce launch order,
# join_side, the 1/W fold, the collective abnormal-loss flag -- is the product code.
def _cpu_control(self, phase, loss):
    """Trainer._control (gtos_step_control on the GPU) through the oracle's restatement of the loop bookkeeping
    (generator/train.py:136-148), on the trainer's own state tensors."""
    from oracle import gtos_oracle as O
    c = O.LoopCounters(self.embed_dim, self.warmup_steps)
    c.loss_acm, c.batches_acm, c.discarded = float(self._state[0]), int(self._state[1]), int(self._state[2])
    if phase == 0:
        self._flag[0] = 1.0 if c.abnormal(float(loss)) else 0.0
        return
    lr = c.advance(float(loss), bool(self._flag[0] > 0))
    self._state.copy_(torch.tensor([c.loss_acm, c.batches_acm, c.discarded], dtype=torch.float64))
    if lr is None:
        self._ctl[1] = 1.0
    else:
        self._ctl[0], self._ctl[1] = lr, 0.0


def _cpu_flat_step(self, lr, gscale=1.0, max_norm=1.0, ctl=None):
    from oracle import gtos_oracle as O
    from gtos_amd import ops
    ops.join_side()
    if ctl is not None:
        if float(ctl[1]) != 0.0:
            self.steps += 1
            return
        lr = float(ctl[0])
    g = self.grad * gscale
    coef, _ = O.clip_coef([g], max_norm)
    for lo, hi, wd in self.adam_ranges:
        p, m, v = O.adam_step(self.param[lo:hi], g[lo:hi] * coef, self.m[lo:hi], self.v[lo:hi], lr, wd)
        self.param[lo:hi].copy_(p)
        self.m[lo:hi].copy_(m)
        self.v[lo:hi].copy_(v)
    self.steps += 1


class _SegModel(torch.nn.Module):
    """Four sub-modules named like the Generator's gradient segments, wired with the same boundary markers."""

    def __init__(self):
        super().__init__()
        lin = lambda: torch.nn.Sequential(torch.nn.Linear(12, 12), torch.nn.Tanh())
        self.concept_encoder, self.graph_encoder, self.snt_encoder, self.decoder = lin(), lin(), lin(), lin()
        self.probe_generator = torch.nn.Linear(12, 12)
        self.grad_sync = None

    def forward(self, x):                     # x [B, 12]; loss = mean over the batch rows
        gs = self.grad_sync
        h = self.concept_encoder(x)
        side = h * 0.5
        if gs is not None:
            h, side = gs.boundary(2, h, side)
        h = self.graph_encoder(h) + side
        probe = self.probe_generator(h)
        if gs is not None:
            h, probe = gs.boundary(1, h, probe)
        t = self.snt_encoder(h)
        if gs is not None:
            t, h, probe = gs.boundary(0, t, h, probe)
        return (self.decoder(t + h) * probe).pow(2).sum(1).mean()


def _trainer_worker(rank, world, port, q, kind, steps):
    import gtos_amd.flat as flat_mod
    import gtos_amd.train as train_mod
    from gtos_amd import ops
    flat_mod.FlatParams.step = _cpu_flat_step
    train_mod.Trainer._control = _cpu_control
    log = []
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        real_ar = dist.all_reduce

        def logged_all_reduce(t, op=dist.ReduceOp.SUM, async_op=False, **kw):
            log.append(("all_reduce", t.numel(), "max" if op == dist.ReduceOp.MAX else "sum", bool(async_op)))
            return real_ar(t, op=op, async_op=async_op, **kw)
        train_mod.dist.all_reduce = logged_all_reduce
    real_join = ops.join_side
    ops.join_side = lambda *a: (log.append(("join_side",)), real_join(*a))[1]
    torch.manual_seed(19940117)
    if kind == "seg":
        model = _SegModel()
        g = torch.Generator().manual_seed(5)
        data = torch.randn(8, 12, generator=g)
        per = 8 // world
        batch = data[rank * per:(rank + 1) * per]
        trainer = train_mod.Trainer(model, 10000, warmup_steps=1, world_size=world, rank=rank,
                                    segment_of=train_mod.generator_segment_of)
        model.grad_sync = trainer if trainer.overlap else None
    else:                                     # the pinned oracle Generator: the real model math, no markers
        from oracle import gtos_oracle as O
        from gtos_amd import synth
        from tests_support import SMALL_VOCAB, SMALL_GEN_ARGS
        vocabs = {k: O.VocabSpec(v, 0) for k, v in SMALL_VOCAB.items()}
        model = O.Generator(vocabs, *SMALL_GEN_ARGS, 32, 64, 4, 0.0, 1, 2, 2)
        model.train()
        per = 4 // world
        batch, _ = synth.make_batch(77, per, 7, 6, vocab=SMALL_VOCAB, first_graph=rank * per)
        trainer = train_mod.Trainer(model, 250000, warmup_steps=1, world_size=world, rank=rank)    # lr = 2e-3 / sqrt(step)
    nseg = len(trainer.flat.segments)
    seg_sizes = [hi - lo for lo, hi in trainer.flat.segments]
    losses, marks = [], []
    real_backward = torch.Tensor.backward

    def marked_backward(self_, *a, **kw):
        r = real_backward(self_, *a, **kw)
        log.append(("backward_end",))         # everything logged so far happened before backward() returned
        return r
    torch.Tensor.backward = marked_backward
    for _ in range(steps):
        log.append(("step_begin",))
        losses.append(trainer.step(batch))
    q.put((rank, losses, trainer.flat.param.detach().numpy().copy(), [n for n, _, _ in trainer.flat.entries], log, marks, nseg, seg_sizes,
           trainer.flat.grad.abs().max().item()))
    if world > 1:
        dist.destroy_process_group()


def _run_trainer(world, kind, steps):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q, kind, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [(r[0], r[1], torch.from_numpy(r[2])) + tuple(r[3:]) for r in res]


def test_trainer_step_two_ranks_segments_overlap_and_equal_one_rank():
    steps = 4
    two = _run_trainer(2, "seg", steps)
    one = _run_trainer(1, "seg", steps)[0]
    for r in two:
        rank, losses, param, names, log, marks, nseg, seg_sizes, gmax = r
        assert nseg == 4 and all(s > 0 for s in seg_sizes)
        assert names == one[3]
        assert gmax == 0.0                                           # zero_grad ran after the optimizer
        # two ranks on half the batch each == one rank on the whole batch (gradient of the global mean)
        torch.testing.assert_close(param, one[2], rtol=1e-4, atol=2e-5)     # Adam's m/sqrt(v) amplifies fp32 summation-order noise
        # event order inside every step
        per_step, cur = [], None
        for e in log:
            if e == ("step_begin",):
                cur = []
                per_step.append(cur)
            else:
                cur.append(e)
        assert len(per_step) == steps
        for s, ev in enumerate(per_step):
            if s >= 2:                                               # batches_acm > warmup_steps: the collective skip flag
                assert ev[0] == ("all_reduce", 1, "max", False)
                ev = ev[1:]
            k = ev.index(("backward_end",))
            # segments 0,1,2 go out from inside backward, async, in order
            assert ev[:k] == [("all_reduce", seg_sizes[i], "sum", True) for i in range(3)], (s, ev)
            after = ev[k + 1:]
            # after backward: join the side stream FIRST, then the last segment; the optimizer joins again (no-op)
            assert after[0] == ("join_side",) and after[1] == ("all_reduce", seg_sizes[3], "sum", True), after
            assert all(e == ("join_side",) for e in after[2:]), after
    mean_two = [(a + b) / 2 for a, b in zip(two[0][1], two[1][1])]
    for a, b in zip(mean_two, one[1]):
        assert abs(a - b) < 1e-5 * max(1.0, abs(b))


def test_trainer_step_real_model_two_ranks_equal_one_rank():
    """The oracle Generator (the real model math on CPU) through Trainer.step: 2 ranks x 2 graphs == 1 rank x 4 graphs
    after 3 optimizer steps -- loss-curve equality at equal global batch."""
    steps = 3
    two = _run_trainer(2, "gen", steps)
    one = _run_trainer(1, "gen", steps)[0]
    assert two[0][3] == one[3]
    for r in two:
        torch.testing.assert_close(r[2], one[2], rtol=2e-4, atol=2e-6)
    assert torch.equal(two[0][2], two[1][2])                          # replicas stay bit-identical
    for a, b, c in zip(two[0][1], two[1][1], one[1]):
        assert abs((a + b) / 2 - c) < 1e-4 * max(1.0, abs(c))


def _bcast_worker(rank, world, port, q):
    import gtos_amd.train as train_mod
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)                      # ranks built DIFFERENTLY on purpose
    model = _SegModel()
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    trainer = train_mod.Trainer(model, 64, warmup_steps=10, world_size=world, rank=rank, segment_of=train_mod.generator_segment_of)
    after = trainer.flat.param.detach().clone()
    views = torch.cat([p.detach().reshape(-1) for _, p, _ in trainer.flat.entries])
    q.put((rank, before.numpy(), after.numpy(), views.numpy()))
    dist.destroy_process_group()


def test_trainer_construction_broadcasts_rank0_parameters():
    """Replicas start identical even when the ranks were built from different seeds: Trainer broadcasts rank 0's flat buffer."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0, v0), (_, b1, a1, v1) = res
    assert not (b0 == b1).all()                          # they really were different
    assert (a0 == a1).all() and (v0 == v1).all()         # ... and are rank 0's afterwards, in the flat buffer and through the views
    import numpy as np
    assert np.array_equal(np.sort(v0), np.sort(b0))      # rank 0 kept its own values (the flat layout only permutes them)


# ---------------------------------------------------------------------------------------------------------------------
# The PRODUCT Generator data-parallel on two gloo ranks, under tests/dryrun.py: every kernel launch is recorded instead of executed
# (numbers are meaningless), but the segment markers of gtos_amd/generator.py, the parameter -> segment map, the async all-reduces
# launched from inside the real model's backward, the collective flag and the optimizer's joins are all the product's own code.
def _product_worker(rank, world, port, q, steps, b_rank=4):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dryrun import DryRun
    import gtos_amd.train as train_mod
    from gtos_amd import synth
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    real_ar = dist.all_reduce

    def logged_all_reduce(t, op=dist.ReduceOp.SUM, async_op=False, **kw):
        t.nan_to_num_(0.0, 0.0, 0.0) if t.is_floating_point() else None       # (uninitialised kernel outputs: keep gloo's arithmetic quiet)
        log.append(("all_reduce", t.numel(), "max" if op == dist.ReduceOp.MAX else "sum", bool(async_op)))
        return real_ar(t, op=op, async_op=async_op, **kw)
    train_mod.dist.all_reduce = logged_all_reduce
    real_backward = torch.Tensor.backward

    def marked_backward(self_, *a, **kw):
        r = real_backward(self_, *a, **kw)
        log.append(("backward_end",))
        return r
    torch.Tensor.backward = marked_backward
    dev = torch.device("cpu")
    with DryRun() as rec:
        torch.manual_seed(19940117)
        model = build_generator(Generator, "C1", dev, factored_relation=True).to(dev)
        model.set_compute_dtype(torch.bfloat16)
        model.train()
        trainer = train_mod.Trainer(model, synth.CONFIGS["C1"]["d"], warmup_steps=1, compute_dtype=torch.bfloat16, world_size=world, rank=rank)
        batch, _ = synth.make_config_batch("C1", rank=rank, B=b_rank)       # rank r holds graphs [b r, b r + b)
        attach_relation_index(attach_path_trie(batch))
        launches = []
        for _ in range(steps):
            log.append(("step_begin",))
            n0 = len(rec.calls)
            trainer.step(batch, sync=False)
            launches.append(len(rec.calls) - n0)
        seg_sizes = [hi - lo for lo, hi in trainer.flat.segments]
        q.put((rank, log, seg_sizes, launches, bool(trainer.overlap), float(batch["concept"].sum())))
    dist.destroy_process_group()


def _check_product_generator(world, steps, b_rank):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_product_worker, args=(r, world, port, q, steps, b_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, log0, seg0, launches0, overlap0, sig0 = res[0]
    assert len(seg0) == 4 and all(s > 0 for s in seg0)
    assert len(set(r[5] for r in res)) == world                           # the ranks hold different graphs
    coll0 = [e for e in log0 if e[0] != "backward_end"]
    for r, log_r, seg_r, launches_r, overlap_r, _ in res:
        assert overlap_r and seg_r == seg0
        assert [e for e in log_r if e[0] != "backward_end"] == coll0, "rank %d issues another collective sequence than rank 0" % r
        # each rank repeats its own launch plan every step (the plans differ between ranks: rank-local graphs, other trie depths)
        assert len(set(launches_r)) == 1 and launches_r[0] > 300
        per_step, cur = [], None
        for e in log_r:
            if e == ("step_begin",):
                cur = []
                per_step.append(cur)
            else:
                cur.append(e)
        assert len(per_step) == steps
        for s, ev in enumerate(per_step):
            flags = [e for e in ev if e[0] == "all_reduce" and e[2] == "max"]
            sums = [e for e in ev if e[0] == "all_reduce" and e[2] == "sum"]
            # the collective abnormal-loss flag: one MAX all-reduce of ONE word per step once steps_issued (= s before the step) > warmup_steps (= 1)
            assert len(flags) == (1 if s >= 2 else 0) and all(e[1] == 1 for e in flags), (r, s, flags)
            assert [e[1] for e in sums] == seg0 and all(e[3] for e in sums)   # the four gradient segments, in order, async
            k = ev.index(("backward_end",))
            inside = [e for e in ev[:k] if e[0] == "all_reduce" and e[2] == "sum"]
            assert [e[1] for e in inside] == seg0[:3], (r, s, ev)            # three of them launched from INSIDE the real model's backward


def test_product_generator_two_ranks_collective_sequence_under_dry_run():
    _check_product_generator(world=2, steps=3, b_rank=4)


def test_product_generator_eight_ranks_collective_sequence_under_dry_run():
    """C4's rank count (8 x MI355X, generator/train.py:167-190) on gloo: eight processes of the PRODUCT Generator under the dry run, one
    graph each -- every rank must issue the same collectives in the same order with the same sizes (four gradient segments, three of them
    from inside backward, the one-word flag all-reduce after the warm-up step): a rank that skipped or reordered one would hang RCCL."""
    _check_product_generator(world=8, steps=3, b_rank=1)
