"""One training step of the gtos Generator on MI355X, data-parallel over RCCL.

Counterpart of the hot loop of /root/reference/generator/train.py:136-154: forward -> loss -> (abnormal-loss rule)
-> backward -> gradient averaging -> clip -> lr schedule -> Adam -> zero_grad.  Differences by design:
  * gradients live in one flat bucket, so DP is ONE all-reduce (RCCL over xGMI) instead of 182 (train.py:74-79);
  * the abnormal-loss skip (train.py:142-145) is decided COLLECTIVELY (max over ranks), because the reference's
    rank-local ``continue`` would desynchronise the collective sequence.
"""
import torch
import torch.distributed as dist

from . import ops
from .flat import FlatParams, inverse_sqrt_lr


class Trainer:
    def __init__(self, model, embed_dim, warmup_steps=2000, compute_dtype=torch.float32, world_size=1):
        self.model = model
        self.embed_dim, self.warmup_steps = embed_dim, warmup_steps
        self.world_size = world_size
        self.flat = FlatParams(model, mirror_dtype=compute_dtype)
        self.batches_acm, self.loss_acm, self.discarded = 0, 0.0, 0

    def all_reduce_grads(self):
        ops.join_side()                    # deferred side-stream gradient GEMMs (gru.py) land before the bucket is read
        if self.world_size > 1:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM)

    def step(self, batch):
        """Returns the loss value (float) of this step, or None when the batch was discarded."""
        loss = self.model(batch)
        loss_value = loss.item()
        abnormal = self.batches_acm > self.warmup_steps and loss_value > 5. * (self.loss_acm / self.batches_acm)
        if self.world_size > 1 and self.batches_acm > self.warmup_steps:
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
