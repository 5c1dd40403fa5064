"""GPU: a training step captured as a hipGraph (gtos_amd.train.GraphedStep) and the device-side seed epoch behind it
(gtos_set_seed_epoch): kernel arguments are frozen by a capture, the dropout masks must not be."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def test_seed_epoch_changes_the_masks_and_keeps_forward_and_backward_consistent():
    from gtos_amd import ops
    x = torch.randn(512, 256, device=dev())
    r = torch.randn(512, 256, device=dev())
    g, b = torch.ones(256, device=dev()), torch.zeros(256, device=dev())
    up = torch.randn(512, 256, device=dev())
    epoch = torch.zeros((), dtype=torch.int64, device=dev())

    def run():
        ops.set_seed(1234)                                  # the SAME host-side seed every time, like a replayed launch
        xx, rr = x.clone().requires_grad_(), r.clone().requires_grad_()
        y = ops.LayerNormResidualFn.apply(xx, rr, g, b, 0.5, 1e-5, False)
        (y * up).sum().backward()
        return y.detach().clone(), rr.grad.clone()

    y_plain, dr_plain = run()
    try:
        ops.set_seed_epoch(epoch)
        y0, dr0 = run()                                     # epoch 0: seed + 0 -> the plain masks
        assert torch.equal(y0, y_plain) and torch.equal(dr0, dr_plain)
        epoch.add_(1)
        y1, dr1 = run()
        epoch.add_(1)
        y2, dr2 = run()
        assert not torch.equal(y1, y_plain) and not torch.equal(y2, y1)
        # backward drew the mask the forward drew: dr is zero exactly where the dropped residual contributed nothing
        keep1, keep2 = dr1 != 0, dr2 != 0
        assert 0.4 < float(keep1.float().mean()) < 0.6 and not torch.equal(keep1, keep2)
        epoch.fill_(1)
        y1b, dr1b = run()                                   # same epoch again: the same masks (a replay is reproducible from its epoch)
        assert torch.equal(y1b, y1) and torch.equal(dr1b, dr1)
    finally:
        ops.set_seed_epoch(None)
    y_end, _ = run()
    assert torch.equal(y_end, y_plain)
    with pytest.raises(ValueError):
        ops.set_seed_epoch(torch.zeros((), dtype=torch.int32, device=dev()))


# The capture tests run by default since round 5 (C1, the library's default RelationEncoder semantics, ONE stream: the configuration that
# has captured and replayed correctly in every run of rounds 4 and 5), each in a CHILD process with a hard timeout: on this stack (ROCm
# 7.2, torch 2.10) the capture of the multi-stream C2 step segfaulted inside hipStreamEndCapture and one single-stream C2 variant hung at
# replay, and a wedged process must cost its own test, not the rest of the `-m gpu` run.  GTOS_TEST_HIPGRAPH=0 skips
# them; inside the child GTOS_TEST_HIPGRAPH=child runs the body.
_MODE = os.environ.get("GTOS_TEST_HIPGRAPH", "1")
graphs = pytest.mark.skipif(_MODE == "0", reason="hipGraph capture tests switched off: GTOS_TEST_HIPGRAPH=0")


def _in_child(name):
    """Parent: run test ``name`` of this file in a child process (timeout 240 s) and assert it passed; returns True.  Child: returns False."""
    if _MODE == "child":
        return False
    import subprocess
    import sys
    env = dict(os.environ, GTOS_TEST_HIPGRAPH="child")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__ + "::" + name, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    return True


def _trainer(dropout, seed=0, masks="path"):
    from gtos_amd import ops, synth
    from gtos_amd.config import build_generator
    from gtos_amd.encoder import set_relation_mask_sharing
    from gtos_amd.generator import Generator
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    from gtos_amd.train import Trainer
    m = build_generator(Generator, "C1", dev(), dropout=dropout).to(dev())
    m.set_compute_dtype(torch.bfloat16)
    set_relation_mask_sharing(m, masks)          # "path" = the library default (the reference's dropout semantics): since round 5 its
    m.train()                                    # RelationEncoder has no host read (the batch's trie carries the sort order and step sizes)
    batch, _ = synth.make_config_batch("C1", rank=0, B=8)
    attach_relation_index(attach_path_trie(batch))
    batch = {k: (v.to(dev()) if hasattr(v, "to") else v) for k, v in batch.items()}
    ops.set_seed(99)
    torch.manual_seed(seed)
    return Trainer(m, 256, warmup_steps=100, compute_dtype=torch.bfloat16), batch


@graphs
def test_graphed_step_without_dropout_trains_like_the_eager_step():
    """dropout 0: 3 warm-up steps + 5 replays == 8 eager steps, loss by loss (same kernels, same order, same batch)."""
    if _in_child("test_graphed_step_without_dropout_trains_like_the_eager_step"):
        return
    from gtos_amd.train import GraphedStep
    tr_e, batch = _trainer(0.0)
    eager = [tr_e.step(batch) for _ in range(8)]
    tr_g, batch_g = _trainer(0.0)
    gs = GraphedStep(tr_g, batch_g, warmup=3)
    try:
        got = [gs().value() for _ in range(5)]
    finally:
        gs.close()
    assert tr_g.steps_issued == 8 and tr_g.batches_acm == 8
    for a, b in zip(eager[3:], got):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (eager, got)
    assert eager[-1] < eager[0]                              # and it trains


@graphs
def test_graphed_step_draws_new_dropout_masks_in_every_replay():
    if _in_child("test_graphed_step_draws_new_dropout_masks_in_every_replay"):
        return
    from gtos_amd.train import GraphedStep
    tr, batch = _trainer(0.2)
    gs = GraphedStep(tr, batch, warmup=3)
    try:
        w = next(p for p in tr.model.parameters() if p.requires_grad)
        before = w.detach().clone()
        losses = [gs().value() for _ in range(6)]
        assert not torch.equal(before, w.detach())           # the replays update the parameters
    finally:
        gs.close()
    assert all(v is not None and v == v for v in losses)
    assert len(set(round(v, 5) for v in losses)) == len(losses), losses
