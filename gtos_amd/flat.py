"""Flat parameter / gradient / optimizer-state buffers and the fused optimizer step.

MI355X-first replacement for the reference's per-parameter machinery (/root/reference/generator/train.py:74-79
``average_gradients`` = 182 blocking all-reduces, :151 ``clip_grad_norm_``, generator/adam.py:28-87 = 5 small kernels
per parameter): every parameter is a view into ONE fp32 buffer, every gradient a view into ONE fp32 bucket, so
data parallelism is a handful of large RCCL all-reduces over CONTIGUOUS segments of the bucket (train.Trainer launches
each as soon as backward has produced it) and the optimizer is a square-norm launch plus one Adam launch per
(segment, weight-decay class).  Semantics are the reference's: grads averaged over ranks, global-norm clip to 1.0,
Adam(0.9, 0.999, eps 1e-6) without bias correction, decoupled weight decay 1e-4 except for ``bias`` / ``layer_norm``
parameters (train.py:123-132), lr = d^-0.5 min(s^-0.5, s w^-1.5) (train.py:81-83).
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
    """``segment_of(name) -> int`` groups the parameters into contiguous segments of the flat buffers (segment 0 first);
    inside a segment the weight-decay parameters precede the no-decay ones.  Default: one segment."""

    def __init__(self, model, mirror_dtype=None, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6, segment_of=None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        seg_id = (lambda n: 0) if segment_of is None else segment_of
        nseg = 1 + max(seg_id(n) for n, _ in named)
        self.entries = []
        self.segments = []          # (lo, hi) element ranges, one per segment
        self.adam_ranges = []       # (lo, hi, weight decay)
        off = 0
        for s in range(nseg):
            seg_lo = off
            for nodecay in (False, True):
                lo = off
                for n, p in named:
                    if seg_id(n) == s and is_no_decay(n) == nodecay:
                        self.entries.append((n, p, off))
                        off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                if off > lo:
                    self.adam_ranges.append((lo, off, 0.0 if nodecay else weight_decay))
            self.segments.append((seg_lo, off))
        assert len(self.entries) == len(named)
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
        # transposed bf16 mirror of every 2-D weight (the operand of dX = dY W as an NT product), refreshed by ONE launch
        self.mirror_t, self._tr = None, None
        if self.mirror is not None and self.mirror.dtype == torch.bfloat16 and self.param.is_cuda:
            mats = [(o, p.shape[0], p.shape[1]) for n, p, o in self.entries if p.dim() == 2]
            if mats:
                self.mirror_t = torch.zeros_like(self.mirror)
                desc = torch.tensor([[o, r, c] for o, r, c in mats], dtype=torch.int64)
                tiles = [((r + 31) // 32) * ((c + 31) // 32) for _, r, c in mats]
                starts = [0]
                for t_ in tiles[:-1]:
                    starts.append(starts[-1] + t_)
                self._tr = (len(mats), desc.to(dev), torch.tensor(starts, dtype=torch.int32).to(dev), sum(tiles))
                for n, p, o in self.entries:
                    if p.dim() == 2:
                        p._gtos_mirror_t = self.mirror_t[o:o + p.numel()].view(p.shape[1], p.shape[0])
        self.weight_decay, self.betas, self.eps = weight_decay, betas, eps
        self.steps = 0
        self.sync_mirror()
        # a checkpoint loaded after construction writes the fp32 masters in place: refresh everything derived from them
        if hasattr(model, "register_load_state_dict_post_hook"):
            model.register_load_state_dict_post_hook(lambda module, incompatible: self.sync_mirror())

    def sync_mirror(self):
        """Re-derive the bf16 mirror and invalidate cached weight transposes after the fp32 masters were written by
        anything other than step() (load_state_dict, manual edits)."""
        self.check_views()
        if self.mirror is not None:
            call("gtos_cast_f32_to_bf16", self.total, ptr(self.param), ptr(self.mirror), stream())
        self._refresh_transposes()
        _ops.PARAM_EPOCH[0] += 1

    def _refresh_transposes(self):
        if self._tr is not None:
            n_mat, desc, starts, total = self._tr
            call("gtos_transpose_batch_bf16", n_mat, ptr(desc), ptr(starts), total, ptr(self.mirror), ptr(self.mirror_t), stream())

    def check_views(self):
        """model.to()/.float() after construction would silently detach parameters from the flat buffers."""
        base, es = self.param.data_ptr(), 4
        for n, p, o in self.entries:
            if p.data_ptr() != base + o * es:
                raise RuntimeError("parameter %s no longer lives in the flat buffer (was the model moved or cast after "
                                   "FlatParams/Trainer construction?)" % n)

    def zero_grad(self):
        _ops.join_side()
        self.grad.zero_()

    def grad_norm(self, gscale=1.0):
        _ops.join_side()                   # deferred side-stream gradient GEMMs (gru.py) must have landed
        self.sqnorm.zero_()
        call("gtos_sqnorm", self.total, ptr(self.grad), ptr(self.sqnorm), stream())
        return self.sqnorm.sqrt() * gscale

    def step(self, lr, gscale=1.0, max_norm=1.0, ctl=None):
        """gscale = 1/world_size after a SUM all-reduce.  Clips by global norm then applies Adam.  ``ctl``: device fp32[2] =
        {learning rate, skip flag} written by gtos_step_control (train.Trainer); then ``lr`` is ignored and a set skip flag
        leaves parameters and moments untouched -- the whole decision stays on the device."""
        _ops.join_side()                   # deferred side-stream gradient work must have landed in self.grad
        if self.steps == 0:
            self.check_views()
        self.sqnorm.zero_()
        call("gtos_sqnorm", self.total, ptr(self.grad), ptr(self.sqnorm), stream())
        b1, b2 = self.betas
        es = 4
        for lo, hi, wd in self.adam_ranges:
            mir = None if self.mirror is None else self.mirror.data_ptr() + lo * self.mirror.element_size()
            if ctl is not None:
                call("gtos_adam_step_ctl", hi - lo, self.param.data_ptr() + lo * es, self.grad.data_ptr() + lo * es,
                     self.m.data_ptr() + lo * es, self.v.data_ptr() + lo * es, ptr(ctl), b1, b2, self.eps, wd,
                     float(gscale), ptr(self.sqnorm), float(max_norm), mir, stream())
                continue
            call("gtos_adam_step", hi - lo, self.param.data_ptr() + lo * es, self.grad.data_ptr() + lo * es,
                 self.m.data_ptr() + lo * es, self.v.data_ptr() + lo * es, float(lr), b1, b2, self.eps, wd,
                 float(gscale), ptr(self.sqnorm), float(max_norm), mir, stream())
        self._refresh_transposes()
        self.steps += 1
        _ops.PARAM_EPOCH[0] += 1
