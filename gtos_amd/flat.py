"""Flat parameter / gradient / optimizer-state buffers and the fused optimizer step.

MI355X-first replacement for the reference's per-parameter machinery (/root/reference/generator/train.py:74-79
``average_gradients`` = 182 blocking all-reduces, :151 ``clip_grad_norm_``, generator/adam.py:28-87 = 5 small kernels
per parameter): every parameter is a view into ONE fp32 buffer, every gradient a view into ONE fp32 bucket, so
data parallelism is a single RCCL all-reduce and the optimizer is three kernel launches (square-norm, Adam on
the weight-decay segment, Adam on the no-decay segment).  Semantics are the reference's: grads averaged over
ranks, global-norm clip to 1.0, Adam(0.9, 0.999, eps 1e-6) without bias correction, decoupled weight decay 1e-4
except for ``bias`` / ``layer_norm`` parameters (train.py:123-132), lr = d^-0.5 min(s^-0.5, s w^-1.5) (train.py:81-83).
"""
import torch

from ._lib import call, ptr, stream
from . import ops as _ops

ALIGN = 8   # elements; keeps every view 32-byte (fp32) / 16-byte (bf16 mirror) aligned for vector loads


def is_no_decay(name):
    return name.endswith('bias') or 'layer_norm' in name


def inverse_sqrt_lr(embed_size, step, warmup_steps):
    return embed_size ** -0.5 * min(step ** -0.5, step * (warmup_steps ** -1.5))


class FlatParams:
    def __init__(self, model, mirror_dtype=None, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        decay = [(n, p) for n, p in named if not is_no_decay(n)]
        nodecay = [(n, p) for n, p in named if is_no_decay(n)]
        self.entries = []
        off = 0
        for n, p in decay + nodecay:
            if len(self.entries) == len(decay):
                self.decay_end = off
            self.entries.append((n, p, off))
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        if len(nodecay) == 0:
            self.decay_end = off
        self.total = off
        dev = named[0][1].device
        self.param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.param)
        self.m = torch.zeros_like(self.param)
        self.v = torch.zeros_like(self.param)
        self.mirror = torch.zeros(off, dtype=mirror_dtype, device=dev) if mirror_dtype not in (None, torch.float32) else None
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        for n, p, o in self.entries:
            k = p.numel()
            self.param[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.param[o:o + k].view(p.shape)
            p.grad = self.grad[o:o + k].view(p.shape)
            if self.mirror is not None:
                p._gtos_mirror = self.mirror[o:o + k].view(p.shape)
        if self.mirror is not None:
            call("gtos_cast_f32_to_bf16", self.total, ptr(self.param), ptr(self.mirror), stream())
        self.weight_decay, self.betas, self.eps = weight_decay, betas, eps
        self.steps = 0

    def zero_grad(self):
        _ops.join_side()
        self.grad.zero_()

    def grad_norm(self, gscale=1.0):
        self.sqnorm.zero_()
        call("gtos_sqnorm", self.total, ptr(self.grad), ptr(self.sqnorm), stream())
        return self.sqnorm.sqrt() * gscale

    def step(self, lr, gscale=1.0, max_norm=1.0):
        _ops.join_side()                   # deferred side-stream gradient work must have landed in self.grad
        """gscale = 1/world_size after a SUM all-reduce.  Clips by global norm then applies Adam."""
        self.sqnorm.zero_()
        call("gtos_sqnorm", self.total, ptr(self.grad), ptr(self.sqnorm), stream())
        b1, b2 = self.betas
        es = 4
        for lo, hi, wd in ((0, self.decay_end, self.weight_decay), (self.decay_end, self.total, 0.0)):
            if hi <= lo:
                continue
            mir = None if self.mirror is None else self.mirror.data_ptr() + lo * self.mirror.element_size()
            call("gtos_adam_step", hi - lo, self.param.data_ptr() + lo * es, self.grad.data_ptr() + lo * es,
                 self.m.data_ptr() + lo * es, self.v.data_ptr() + lo * es, float(lr), b1, b2, self.eps, wd,
                 float(gscale), ptr(self.sqnorm), float(max_norm), mir, stream())
        self.steps += 1
        from . import ops
        ops.PARAM_EPOCH[0] += 1
