"""Autograd glue between PyTorch tensors and the C-ABI HIP kernels (include/gtos_hip.h).

Each ``torch.autograd.Function`` here is the MI355X counterpart of one ATen op sequence of the reference
(cited per function).  Tensors provide device memory; kernels are enqueued on torch's current HIP stream.
"""
import contextlib
import os

import torch

from . import _lib
from ._lib import call, dt, ptr, require_cuda, stream

_seed_state = [0x5DEECE66D]

# Optional kernel timing with HIP events on the launch stream (bench.py's roofline leg): name -> [(start, end)]
PROFILE = None
PROFILE_DETAIL = False     # also time the per-launch kernels (GRU steps, attention backward, segment sums): bench.py's detail pass
GEMM_PROFILE = None        # tools/profile_step.py: (layout, N, K, out dtype) -> [(M, start, end)]


class _Timed:
    """HIP-event span on the current stream, recorded into PROFILE[name] as (start, end, units); ``detail`` spans are only
    taken in bench.py's detail pass (an event pair costs more than a small kernel)."""

    def __init__(self, name, detail=False, units=0):
        self.name, self.units = name, units
        self.on = PROFILE is not None and (PROFILE_DETAIL or not detail)

    def __enter__(self):
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if self.on:
            self.e.record()
            PROFILE.setdefault(self.name, []).append((self.s, self.e, self.units))


def next_seed():
    """Dropout seeds: a host-side counter (replayed identically on every rank with the same start)."""
    _seed_state[0] = (_seed_state[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return _seed_state[0]


def set_seed(s):
    _seed_state[0] = s & 0xFFFFFFFFFFFFFFFF


_SEED_EPOCH = [None]


def set_seed_epoch(epoch):
    """Register (or, with None, clear) the device int64 word every dropout kernel folds into its seed: gtos_set_seed_epoch.  A
    captured training step (train.GraphedStep) freezes the host-side seeds above; the step increments this word once per replay, so
    every replay draws new masks (and the forward and the backward of one replay the same ones).  The tensor is kept alive here."""
    if epoch is not None and not (epoch.is_cuda and epoch.dtype == torch.int64 and epoch.numel() == 1):
        raise ValueError("the seed epoch is one int64 on the GPU")
    torch.cuda.synchronize()
    call("gtos_set_seed_epoch", None if epoch is None else ptr(epoch))
    _SEED_EPOCH[0] = epoch


def _row_major(t):
    """2-D view usable by the GEMM: unit inner stride; returns (tensor, leading dimension)."""
    if t.dim() != 2:
        raise _lib.GtosHipError("gemm operands must be 2-D")
    if (t.stride(1) != 1 and t.shape[1] != 1) or (t.stride(0) == 0 and t.shape[0] > 1):
        t = t.contiguous()           # (also a row-broadcast view -- an expanded probe at batch size 1 -- : the kernels are only ever
                                     #  tested with a positive leading dimension; found by the dry run's operand-extent check)
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t, ld


def _splitk(M, N, K):
    """K splits for weight-gradient shapes (few output tiles, long K).  The kernel runs all tiles of one split on one
    XCD (32 CUs x 4 resident workgroups = 128 slots), so the split count is a multiple of 8 with tiles * splits / 8 <= 128:
    one resident round on every XCD (42 splits of a 24-tile product left two XCDs with 6 x 24 = 144 workgroups, a second
    round, and 565 instead of 800 TF/s).  At least 6 k tiles per split; partial tiles go through the workspace."""
    blocks = ((M + 127) // 128) * ((N + 127) // 128)
    if blocks >= 512 or K < 1024:
        return 1
    sk = min(8 * max(1, 128 // blocks), 256, max(1, WORKSPACE_BYTES // (4 * M * N)))
    kmax = K // 384
    if kmax < sk:
        sk = (kmax // 8) * 8 if kmax >= 8 else kmax
    return max(1, sk)


# A split-K weight gradient launched on the AUXILIARY stream beside the main chain: its workgroups are long-running (one 256x256 tile over
# K / splits rows: ~360 us for the relation projection's dW at C2), take a whole CU each (128 KB of LDS, 2 x 232 registers per SIMD) and
# cannot be preempted, so with 8 tiles x 32 splits = 256 of them the main stream's next launches -- a handful of small kernels between two
# attention backward passes -- wait for the whole product: 0.33 ms per graph-encoder layer at C2 (profiles/r6z_step_phases.txt: the 6,464-row
# in-projection dW on the main stream takes 351 us beside it, 25 us alone).  Capping the product at SIDE_GEMM_MAX_WGS workgroups leaves CUs
# to the main stream; the product gets longer by the same factor and still ends before the next layer's.  (0 = no cap.)
SIDE_GEMM_MAX_WGS = int(os.environ.get("GTOS_SIDE_GEMM_WGS", "192"))


# (WHEN it runs was tried too: deferred to the start of the NEXT layer's attention backward, so that it runs beside those three HBM-bound
# kernels instead of the layer's small ones: 79.1-79.8 ms per step against 77.1-77.5 launched in place, caps of 96-192 -- the long MFMA
# workgroups cost the attention kernels more CUs than the small kernels lose; profiles/r6_ab_switches.txt.)


def _splitk_side(M, N, K):
    """_splitk for a product that runs on the auxiliary stream beside latency-critical main-stream work (see SIDE_GEMM_MAX_WGS)."""
    sk = _splitk(M, N, K)
    if SIDE_GEMM_MAX_WGS > 0 and sk >= 16:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        while sk > 8 and tiles * sk > SIDE_GEMM_MAX_WGS:
            sk -= 8
    return sk


_WS = {}
WORKSPACE_BYTES = 256 << 20


def _workspace(device):
    """Split-K partial-tile workspace: one per (device, stream) -- launches on one stream reuse it in order, launches on
    different streams (the GRU weight gradients run beside the BPTT steps) must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
    return ws


_SIDE = {}
# Prefetch every layer's relation projection on the side stream (they depend on the bank only, not on the layer chain): the
# MFMA-bound GEMMs run beside the HBM-bound attention kernels.  Measured at C2 (same box, round 3): 64.29 -> 63.70 ms per step.
# The attention launches of the timed steps then share the chip with a GEMM, so bench.py takes the in-step duration of the
# roofline kernel from its detail pass, where the overlap is off.  GTOS_PROJ_SIDE=0 switches it off.
PROJ_SIDE = os.environ.get("GTOS_PROJ_SIDE", "1") != "0"
# ... as long as all the layers' projections together stay small beside the rest of the step (7 GB at C2).  At C5 (R = 1.84 M:
# 30 GB of projections alive at once on top of a 150 GB step) the prefetch made the step time scatter between 215 and 330 ms.
PROJ_SIDE_MAX_BYTES = 16 << 30
# GTOS_PROJ_RECOMPUTE=1 (opt-in): the attention core does not keep a layer's projected bank [R, 2d] for its backward but recomputes it
# there from the bank and the layer's weight (the same GEMM, the same bits).  C5: 3.8 GB per layer, 26 GB of the 93.6 GB peak, for one
# more [R,d]x[d,2d] product per layer and step.
PROJ_RECOMPUTE = os.environ.get("GTOS_PROJ_RECOMPUTE", "0") == "1"
# Backward of the relation projections on the side stream (see LinearFn.backward): overlaps only backward kernels.
BWD_SIDE = os.environ.get("GTOS_BWD_SIDE", "1") != "0"
BWD_SIDE_MIN_ROWS = 100000
# Weight (and bias) gradients of the model's SMALL linear layers -- dY^T X over the n*B = 6,464 rows of a graph layer or the T*B = 3,200 rows
# of a decoder layer -- are not launched where autograd reaches them (63 split-K products of ~22 us per C2 step, each with its partial-tile
# reduction, a bias column sum and often a fill: 270 latency-bound launches, 3.6 ms of the main stream) but noted and launched TOGETHER as one
# batched product (gtos_gemm_tn_batch: a workgroup per 256x256 tile of one job over its whole K, added to the flat gradient in place, plus one
# batched column-sum launch) when the gradient bucket is about to be read: flush_dw() from join_side() and, data parallel, in front of every
# segment's all-reduce (train.GradSync.segment_ready).  The operands stay referenced until then.
DW_BATCH = os.environ.get("GTOS_DW_BATCH", "1") != "0"
DW_BATCH_MAX_ROWS = 32768
_DW_PENDING = {}                  # device -> [(dy2, x2, weight-gradient target, bias-gradient target or None)]


def _dw_batchable(dy2, x2, tgt, btgt):
    ok = (DW_BATCH and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and tgt.dtype == torch.float32 and 0 < dy2.shape[0] <= DW_BATCH_MAX_ROWS
          and dy2.shape[1] % 8 == 0 and x2.shape[1] % 8 == 0 and dy2.stride(1) == 1 and x2.stride(1) == 1 and tgt.stride(1) == 1
          and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0 and tgt.stride(0) % 4 == 0 and dy2.stride(0) >= dy2.shape[1]
          and x2.stride(0) >= x2.shape[1] and tgt.stride(0) >= tgt.shape[1]
          and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0 and tgt.data_ptr() % 16 == 0)
    return ok and (btgt is None or (btgt.dtype == torch.float32 and btgt.is_contiguous()))


def flush_dw(device=None):
    """Launch the noted small weight gradients of ``device`` (all devices: None) on the current stream, in the order they were noted."""
    import ctypes
    for dev in list(_DW_PENDING):
        if device is not None and dev != device:
            continue
        jobs = _DW_PENDING.pop(dev)
        n = len(jobs)
        if not n:
            continue
        vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
        A = vp(*[j[0].data_ptr() for j in jobs]); B = vp(*[j[1].data_ptr() for j in jobs]); C = vp(*[j[2].data_ptr() for j in jobs])
        bias = vp(*[(j[3].data_ptr() if j[3] is not None else None) for j in jobs])
        lda = i64(*[j[0].stride(0) for j in jobs]); ldb = i64(*[j[1].stride(0) for j in jobs]); ldc = i64(*[j[2].stride(0) for j in jobs])
        M = i32(*[j[0].shape[1] for j in jobs]); N = i32(*[j[1].shape[1] for j in jobs]); K = i32(*[j[0].shape[0] for j in jobs])
        for j in jobs:
            for t_ in j:
                ptr(t_)                      # (the dry-run recorder learns the storages behind the table's pointers)
        with _Timed("gemm_tn_batch", detail=True, units=sum(2 * j[0].shape[0] * j[0].shape[1] * j[1].shape[1] for j in jobs)):
            with torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext():
                call("gtos_gemm_tn_batch", n, ctypes.addressof(A), ctypes.addressof(lda), ctypes.addressof(M), ctypes.addressof(B), ctypes.addressof(ldb),
                     ctypes.addressof(N), ctypes.addressof(K), ctypes.addressof(C), ctypes.addressof(ldc), ctypes.addressof(bias), stream())
        del jobs


# The auxiliary stream has a price in memory: the caching allocator keeps one pool per stream, blocks freed on one stream never serve the
# other, and the two pools peak at different moments of a step -- measured: C2 31 GB allocated, 94-99 GB reserved with the auxiliary
# stream, 37 GB without (61.6 vs 62.1 ms per step); C5 93 GB allocated, 230-277 GB reserved with it (of 288 GB: one allocator retry from
# a stall), 119 GB without (212 vs 214.6 ms).  GTOS_SIDE_STREAMS: "1" / "0" = always / never use it; "auto" (default) = use it while the
# previous step's sampled peak allocation (note_memory) stays below GTOS_SIDE_MAX_ALLOC_FRACTION (0.25) of the device memory, decided once
# per step (refresh_side_policy, called by Generator.encode_step), with hysteresis on the way back.
SIDE_STREAMS = os.environ.get("GTOS_SIDE_STREAMS", "auto")
SIDE_MAX_ALLOC_FRACTION = float(os.environ.get("GTOS_SIDE_MAX_ALLOC_FRACTION", "0.25"))
_SIDE_POLICY, _DEVICE_BYTES = {}, {}


def side_ok(device):
    """May this step use the auxiliary stream on ``device``?  (The individual switches -- GTOS_PROJ_SIDE, GTOS_BWD_SIDE, GTOS_GRU_SIDE,
    GTOS_GRU_L0_OVERLAP -- apply on top.)"""
    return _SIDE_POLICY.get(device, SIDE_STREAMS != "0")


_STEP_PEAK = {}


def note_memory(device):
    """Sample the allocator's current allocation at one of a step's known high-water points (end of the encoder, end of the forward
    pass, inside the RelationEncoder's backward): a host-side counter read, no synchronisation.  refresh_side_policy() decides from the
    PREVIOUS step's sampled maximum."""
    if device.type == "cuda" and SIDE_STREAMS == "auto":
        cur = torch.cuda.memory_allocated(device)
        if cur > _STEP_PEAK.get(device, 0):
            _STEP_PEAK[device] = cur


def refresh_side_policy(device):
    """Once per step, before its first launch: host-side allocator statistics only (no synchronisation).  "auto": the auxiliary stream
    is dropped when the previous step's sampled peak allocation (note_memory) exceeded SIDE_MAX_ALLOC_FRACTION of the device memory
    and taken back when it fell below 0.8 of that bound -- per step, not per process lifetime (round 4 read torch's process-wide peak
    counter: one large evaluation batch switched the stream off for good, and a user resetting that counter switched it back on).
    Data parallel, every rank decides for itself: the stream changes WHEN gradient products run, never what they compute or the order
    they accumulate in, so ranks that decide differently differ in step time only."""
    if device.type != "cuda":
        return
    if SIDE_STREAMS != "auto":
        _SIDE_POLICY[device] = SIDE_STREAMS != "0"
        return
    total = _DEVICE_BYTES.get(device)
    if total is None:
        total = _DEVICE_BYTES[device] = torch.cuda.get_device_properties(device).total_memory
    peak = _STEP_PEAK.pop(device, None)
    if peak is None:                    # the first step of a process: what is allocated right now (parameters, the batch)
        peak = torch.cuda.memory_allocated(device)
    bound = SIDE_MAX_ALLOC_FRACTION * total
    was = _SIDE_POLICY.get(device, True)
    _SIDE_POLICY[device] = peak <= (bound if was else 0.8 * bound)


def side_stream(device):
    """One auxiliary HIP stream per device for work that is independent of the main chain (see gru.py, FactoredRelation)."""
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)       # (a high-priority auxiliary stream measured no different: profiles/r6_ab_switches.txt)
    return s


_PENDING_SIDE = set()
_HELD = {}


def defer_side_join(device, keep_alive=None):
    """The side stream still holds work whose only consumers are the flat gradient bucket's readers: join_side() must run
    before the bucket is read (Trainer.step / FlatParams.step do it).  ``keep_alive``: tensors that work reads; they are
    released at the join."""
    _PENDING_SIDE.add(device)
    if keep_alive:
        _HELD.setdefault(device, []).extend(keep_alive)


def behind_side(device):
    """Context for launching an asynchronous collective on gradients that deferred side-stream work still writes (the batched
    relation-projection weight gradient): the launch happens ON the side stream -- after making it wait for the main stream -- so the
    process group's stream orders the collective behind both, and the main stream is not stalled.  Without pending work: no-op."""
    import contextlib
    if device not in _PENDING_SIDE or not torch.cuda.is_available():
        return contextlib.nullcontext()
    side = side_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    return torch.cuda.stream(side)


def join_side(device=None):
    """Make the current stream wait for deferred side-stream work (no-op when there is none); the noted small weight gradients go out first."""
    if _DW_PENDING:
        flush_dw(device)
    for dev in list(_PENDING_SIDE):
        if device is None or dev == device:
            torch.cuda.current_stream(dev).wait_stream(side_stream(dev))
            _PENDING_SIDE.discard(dev)
            _HELD.pop(dev, None)


def gemm(a, b, trans_a=False, trans_b=False, out=None, bias=None, relu=False, p_drop=0.0, seed=0,
         accumulate=False, out_dtype=None, splitk=1):
    """out[M,N] (+)= act(op(a) @ op(b) + bias).  a,b share a dtype (fp32 or bf16)."""
    require_cuda(a, b, out, bias)
    a, lda = _row_major(a)
    b, ldb = _row_major(b)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    kb = b.shape[1] if trans_b else b.shape[0]
    if kb != K or a.dtype != b.dtype:
        raise _lib.GtosHipError("gemm shape/dtype mismatch: %s %s ta=%s tb=%s" % (tuple(a.shape), tuple(b.shape), trans_a, trans_b))
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
        accumulate = False
    if splitk > 1 and not accumulate:
        out.zero_()
        accumulate = True
    ldc = out.stride(0) if M > 1 else max(N, out.stride(0))
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    if GEMM_PROFILE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    call("gtos_gemm", dt(a), dt(out), int(trans_a), int(trans_b), M, N, K, ptr(a), lda, ptr(b), ldb, ptr(out), ldc,
         ptr(bias), int(relu), float(p_drop), seed, int(accumulate), splitk,
         ptr(_workspace(a.device)) if splitk > 1 else None, WORKSPACE_BYTES if splitk > 1 else 0, stream())
    if GEMM_PROFILE is not None:
        ev[1].record()
        big = "M" if trans_a else "K"       # weight-gradient GEMMs reduce over the long dimension
        key = ("%s%s" % ("T" if trans_a else "N", "T" if trans_b else "N"), N, M if trans_a else K,
               str(a.dtype)[6:] + ">" + str(out.dtype)[6:], splitk if trans_a else 1, big)
        GEMM_PROFILE.setdefault(key, []).append((K if trans_a else M, ev[0], ev[1]))
    return out


def compute_weight(w, dtype):
    """Parameter in the compute dtype: fp32 master itself, or its bf16 mirror (kept fresh by the fused Adam
    kernel when the parameters live in a FlatParams buffer; otherwise cast on the fly)."""
    if w.dtype == dtype:
        return w
    mirror = getattr(w, "_gtos_mirror", None)
    if mirror is not None and mirror.dtype == dtype:
        return mirror
    return w.detach().to(dtype)


PARAM_EPOCH = [0]          # bumped by FlatParams.step(): invalidates cached weight transposes


def weight_t(param, w, rows=None):
    """Contiguous transpose of the compute-dtype weight ``w`` derived from ``param`` (optionally its row block), cached
    ON the parameter object until the parameters change.  dX = dY W then runs as dY (W^T)^T through the all-DMA
    forward GEMM kernel instead of the slower transposing NN variant."""
    mt = getattr(param, "_gtos_mirror_t", None)
    if mt is not None and w.dtype == mt.dtype and getattr(param, "_gtos_mirror", None) is not None:
        # the flat buffers keep a transposed bf16 mirror of every 2-D weight, refreshed by one launch per optimizer step
        # (flat.FlatParams): views of it replace the per-weight transpose copies
        if rows is None:
            if w.shape == param.shape:
                return mt
        elif rows[0] == "cols":                 # transpose of a column block [.., width] of the weight = a row block of the mirror
            width = w.shape[1]
            return mt[rows[1] * width:(rows[1] + 1) * width]
        elif w.shape[0] == rows[1] - rows[0]:   # transpose of a row block = a column block (row stride = out features)
            return mt[:, rows[0]:rows[1]]
    stamp = (PARAM_EPOCH[0], param._version, param.data_ptr())
    cache = getattr(param, "_gtos_wt", None)
    if cache is None:
        cache = {}
        try:
            param._gtos_wt = cache
        except Exception:
            pass
    ent = cache.get((w.dtype, rows))
    if ent is not None and ent[0] == stamp:
        return ent[1]
    t = w.detach().t().contiguous()
    cache[(w.dtype, rows)] = (stamp, t)
    return t


def _grad_target(p):
    """Pre-allocated .grad (a view of the flat gradient bucket) to accumulate into, or None."""
    if not p.is_leaf:
        return None
    g = p.grad
    if g is not None and g.dtype == torch.float32 and g.is_contiguous():
        return g
    return None


# The relation bank feeds relation_in_proj of every graph-encoder layer.  Instead of L input-gradient products
# [R,2d] x [2d,d] accumulated one after the other (K = 2d = 1024: the 128x128-tile kernel, a read-modify-write of the [R,d]
# accumulator per layer, bf16 rounding L times), the layers' d(rel) are written side by side into ONE [R, L*2d] slab (the
# bank-gradient kernel takes a row stride) and the bank's gradient is ONE product with K = L*2d = 8192 on the 256x256
# deep-K kernel, accumulated in fp32 across all layers.  GTOS_BATCH_DX=0 restores the per-layer products.
BATCH_DX = os.environ.get("GTOS_BATCH_DX", "1") != "0"
# Layers per deep-K product of the slab; 0 (default) = ONE product over all layers after the last one, rounded once.  Chunks of
# 2 layers (K = 2048 each, launched from inside backward on the auxiliary stream) were measured at C2: the products leave the
# critical path but compete with the HBM-bound attention-backward kernels they run beside (4 x 1.45 ms of GEMM instead of 3.1 ms,
# attention backward +1.2 ms per step): 66.23 vs 66.09 ms per step on the same box -- no gain, so the single rounding stays.
DX_CHUNK = 0
# ... as long as the slab stays small beside the step's memory (7 GB at C2); above this the layers fall back to one K = 2d product each
# with a single [R, 2d] output gradient alive at a time (C5: R = 1.88 M, the slab would be 30 GB).
SLAB_MAX_BYTES = int(os.environ.get("GTOS_SLAB_MAX_GB", "16")) << 30
# The same slab also gives the layers' relation_in_proj WEIGHT gradients as one product (GradAccumGroup.finish_dw), launched behind the
# input gradient: round 3's trace shows the eight per-layer products queued in FRONT of it on the side stream (their 128 KB-LDS workgroups
# cannot start beside the attention-backward kernels, so all eight run when those are done) with the main stream waiting 5.9 ms for the
# bank's gradient.  MEASURED (round 4, same box, two alternating pairs, profiles/r4f_*): 61.14 / 61.21 ms per step WITH it against
# 60.67 / 60.55 without -- the main stream does start the RelationEncoder's backward ~3 ms earlier, but the 3.65 TFLOP product then
# competes with the HBM-bound GRU kernels it was meant to hide behind (its 128 KB-LDS workgroups wait for whole CUs) and the step gets
# 0.55 ms LONGER: the step is work-conserving, not latency-bound, at this point.  Off (a module constant since round 6; the parity test of the batched weight gradient sets it).
BATCH_DW = False


class GradAccumGroup:
    """Several LinearFn calls that share ONE input tensor (the relation bank): only the last backward call hands the summed
    input gradient to autograd -- instead of L separate [R,d] tensors that autograd would add pairwise."""

    def __init__(self):
        self.pending = 0          # members whose backward has not run yet
        self.members = 0          # registered in forward
        self.buf = None           # per-member accumulation (fallback path)
        self.slab = None          # [R, members*w]: the members' output gradients side by side
        self.width = 0
        self.handed = 0           # column blocks handed out
        self.flushed = 0          # column blocks already folded into buf
        self.pieces = []          # (column block, W^T [in, w]) of members whose backward has run and that are not folded in yet
        self.dw_jobs = []         # (column block, weight parameter, row block or None): members that left their weight gradient to finish_dw
        self.done_slab = None     # the complete slab, kept for finish_dw after add() has handed the input gradient on

    def register(self):
        self.pending += 1
        self.members += 1

    def grad_slice(self, like):
        """Column block of the slab for the next member's output gradient ([R,w], row stride members*w), or None."""
        R, w = like.shape
        if not BATCH_DX or self.members < 2 or like.dtype != torch.bfloat16:
            return None
        if self.slab is None and R * self.members * w * like.element_size() > SLAB_MAX_BYTES:
            return None            # (C5: 30 GB for 8 layers; the per-member products with one [R, w] gradient alive at a time instead)
        if self.slab is None:
            self.slab = torch.empty((R, self.members * w), dtype=like.dtype, device=like.device)
            self.width, self.handed = w, 0
        if self.handed >= self.members or w != self.width or self.slab.shape[0] != R:
            return None
        j = self.handed
        self.handed += 1
        return self.slab[:, j * w:(j + 1) * w]

    def _block_of(self, dy2):
        if self.slab is None or dy2.stride(0) != self.slab.stride(0) or dy2.shape != (self.slab.shape[0], self.width):
            return None
        off = dy2.data_ptr() - self.slab.data_ptr()
        step = self.width * dy2.element_size()
        return off // step if (0 <= off < self.slab.stride(0) * dy2.element_size() and off % step == 0) else None

    def defers_dw(self, dy2):
        """True when this member's weight gradient can wait for finish_dw(): its output gradient sits in the slab and the slab is
        consumed in one piece at the end (BATCH_DW, no chunked folds)."""
        return BATCH_DW and DX_CHUNK == 0 and self._block_of(dy2) is not None

    def add(self, dy2, wt, shape, dw_job=None):
        j = self._block_of(dy2)
        if j is not None and dw_job is not None:
            self.dw_jobs.append((j,) + tuple(dw_job))
        if j is not None:
            self.pieces.append((j, wt))
        elif self.buf is None:
            self.buf = gemm(dy2, wt, trans_b=True)
        else:
            gemm(dy2, wt, trans_b=True, out=self.buf, accumulate=True)
        self.pending -= 1
        # The slab blocks are handed out in backward order, so blocks [done, done + DX_CHUNK) are complete as soon as DX_CHUNK
        # more members have run: their K = DX_CHUNK * w product goes out NOW (the caller runs this on the auxiliary stream,
        # beside the attention-backward kernels of the layers still to come) instead of ONE K = L * w product after the last
        # layer, which nothing could overlap: 3.1 ms of an otherwise idle main stream at C2.
        self._flush(final=self.pending == 0)
        if self.pending > 0:
            return None
        # (the slab outlives the input gradient only when a deferred weight-gradient product still needs it: 7 GB at C2, up to SLAB_MAX_BYTES)
        out, self.done_slab = self.buf, (self.slab if self.dw_jobs else None)
        self.buf, self.slab, self.pieces, self.flushed = None, None, [], 0
        return out.view(shape)

    def finish_dw(self, x2):
        """The members' weight gradients as ONE product over the slab: dW_all [members*w, in] = slab^T x  (x = the shared input, the
        relation bank) instead of one [w, R] x [R, in] product per member -- the bank is read once instead of once per layer, the
        split-K partial tiles of one well-shaped product replace those of L small ones, and (the point) the product no longer sits in
        front of the input gradient on the side stream: the caller launches it AFTER the main stream has been released behind dX, so it
        runs beside the RelationEncoder's backward instead of in front of it.  Returns the tensors that must outlive the launches."""
        jobs, slab, self.dw_jobs, self.done_slab = self.dw_jobs, self.done_slab, [], None
        if not jobs or slab is None:
            return []
        w = self.width
        lo, hi = min(j for j, _, _ in jobs), max(j for j, _, _ in jobs) + 1
        blk = slab[:, lo * w:hi * w]
        dw_all = gemm(blk, x2, trans_a=True, out_dtype=torch.float32, splitk=_splitk((hi - lo) * w, x2.shape[1], x2.shape[0]))
        for j, weight, rows in jobs:
            tgt = _grad_target(weight)
            if rows is not None:
                tgt = tgt[rows[0]:rows[1]]
            tgt += dw_all[(j - lo) * w:(j - lo + 1) * w]
        return [slab, dw_all, x2]

    def _fold(self, take):
        """buf (+)= slab[:, consecutive blocks of ``take``] @ cat(their W^T)^T -- one deep-K product, fp32 accumulation inside it."""
        j0 = take[0][0]
        blk = self.slab[:, j0 * self.width:(j0 + len(take)) * self.width]
        wcat = take[0][1] if len(take) == 1 else torch.cat([w_ for _, w_ in take], dim=1)          # [in, len(take) * w]
        if self.buf is None:
            self.buf = gemm(blk, wcat, trans_b=True)
        else:
            gemm(blk, wcat, trans_b=True, out=self.buf, accumulate=True)

    def _flush(self, final):
        pieces = sorted(self.pieces, key=lambda t: t[0])
        run = []                                             # the blocks that continue the folded prefix without a gap
        for jj, w_ in pieces:
            if jj != self.flushed + len(run):
                break
            run.append((jj, w_))
        rest = pieces[len(run):]
        step = DX_CHUNK if DX_CHUNK > 0 else max(1, self.members)
        while run and (len(run) >= step or final):
            take, run = run[:step], run[step:]
            self._fold(take)
            self.flushed += len(take)
        if final:
            for piece in rest:                               # blocks that arrived out of order: one by one
                self._fold([piece])
            rest = []
        self.pieces = run + rest


class LinearFn(torch.autograd.Function):
    """y = dropout(relu(x W^T + b)) -- F.linear (+relu +dropout) of the reference's projections and FFN
    (generator/graph_transformer.py:61-63,106-122,166).  Weight/bias gradients are accumulated straight into the
    flat fp32 gradient bucket when one is attached (no per-parameter temporaries)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, p_drop, rows, group=None):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        w = compute_weight(weight, x2.dtype)
        b = bias.detach() if bias is not None else None
        if rows is not None:           # a row block of a packed projection (in_proj_weight[r0:r1])
            w = w[rows[0]:rows[1]]
            b = b[rows[0]:rows[1]] if b is not None else None
        seed = next_seed() if p_drop > 0 else 0
        y = gemm(x2, w, trans_b=True, bias=b, relu=relu, p_drop=p_drop, seed=seed)
        wt = weight_t(weight, w, rows) if ctx.needs_input_grad[0] else None       # [in, out], for dX
        ctx.save_for_backward(x2, wt, y if (relu or p_drop > 0) else None)
        if group is not None and ctx.needs_input_grad[0]:
            group.register()
        ctx.cfg = (relu, p_drop, shp, weight, bias, rows, w.shape[0], group)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wt, y = ctx.saved_tensors
        relu, p_drop, shp, weight, bias, rows, n_out, group = ctx.cfg
        dy2 = dy.reshape(-1, n_out)
        if not (dy2.stride(1) == 1 and dy2.stride(0) >= n_out and y is None) and not dy2.is_contiguous():
            dy2 = dy2.contiguous()           # row-strided views (column blocks of a gradient slab) go to the GEMMs as they are
        if y is not None:
            if relu:
                dy2 = dy2.clone()
                call("gtos_relu_dropout_bwd", dt(dy2), dy2.numel(), ptr(dy2), ptr(y), float(p_drop), stream())
            else:
                raise _lib.GtosHipError("dropout without relu is not fused in LinearFn")
        dx = dw = db = None
        # A grouped projection (relation_in_proj of one layer) whose weight gradient lands in the flat bucket hands
        # nothing to autograd until the LAST member of its group: its two GEMMs can run on the side stream, beside the
        # attention-backward kernels of the main chain.  The last member makes the main stream wait for all of them.
        offload = (BWD_SIDE and side_ok(x2.device) and group is not None and bias is None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                   and _grad_target(weight) is not None and dy2.shape[0] >= BWD_SIDE_MIN_ROWS)   # small problems are launch-bound
        if offload:
            dev = dy2.device
            main, side = torch.cuda.current_stream(dev), side_stream(dev)
            side.wait_stream(main)
            dy2.record_stream(side)
            later = group.defers_dw(dy2)
            with torch.cuda.stream(side):
                dx = group.add(dy2, wt, shp, dw_job=(weight, rows) if later else None)
                if not later:
                    tgt = _grad_target(weight)
                    if rows is not None:
                        tgt = tgt[rows[0]:rows[1]]
                    gemm(dy2, x2, trans_a=True, out=tgt, accumulate=True, splitk=_splitk_side(n_out, x2.shape[1], dy2.shape[0]))
            if dx is not None:
                main.wait_stream(side)           # the main chain continues behind the input gradient ...
                dx.record_stream(main)
                with torch.cuda.stream(side):    # ... and the batched weight gradient runs beside it; the bucket's readers join the side stream
                    held = group.finish_dw(x2)
                if held:
                    defer_side_join(dev, held)
            return dx, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = group.add(dy2, wt, shp) if group is not None else gemm(dy2, wt, trans_b=True).view(shp)
        noted = False
        if ctx.needs_input_grad[1]:
            tgt = _grad_target(weight)
            M, N, K = n_out, x2.shape[1], dy2.shape[0]
            sk = _splitk(M, N, K)
            in_bucket = tgt is not None
            if tgt is None:
                tgt = dw = torch.zeros(weight.shape, dtype=torch.float32, device=dy2.device)
            if rows is not None:
                tgt = tgt[rows[0]:rows[1]]
            btgt = None
            if bias is not None and ctx.needs_input_grad[2] and _grad_target(bias) is not None:
                btgt = _grad_target(bias)
                btgt = btgt[rows[0]:rows[1]] if rows is not None else btgt
            if in_bucket and _dw_batchable(dy2, x2, tgt, btgt):      # a small layer: with the other small layers' gradients at flush_dw()
                _DW_PENDING.setdefault(dy2.device, []).append((dy2, x2, tgt, btgt))
                noted = True
            else:
                gemm(dy2, x2, trans_a=True, out=tgt, accumulate=True, splitk=sk)
        if bias is not None and ctx.needs_input_grad[2] and not (noted and _grad_target(bias) is not None):
            tgt = _grad_target(bias)
            if tgt is None:
                tgt = db = torch.zeros(bias.shape, dtype=torch.float32, device=dy2.device)
            if rows is not None:
                tgt = tgt[rows[0]:rows[1]]
            call("gtos_colsum", dt(dy2), dy2.shape[0], dy2.shape[1], dy2.stride(0), ptr(dy2), ptr(tgt), stream())
        return dx, dw, db, None, None, None, None


def linear(x, weight, bias=None, relu=False, p_drop=0.0, rows=None, group=None):
    return LinearFn.apply(x, weight, bias, relu, float(p_drop), rows, group)


class LayerNormResidualFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(r)) (generator/graph_transformer.py:57-58,64-65).

    ``out32``: bf16 activations on an fp32 residual stream -- x may be fp32 or bf16, r is bf16, and the function returns the pair
    (y fp32 = the next residual / the layer's output, y16 = its bf16 copy, the next projection's MFMA operand); backward adds the
    two incoming gradients inside the kernel.  Without it: one output in x's dtype, as before."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, p_drop, eps, out32):
        d = x.shape[-1]
        x = x.contiguous()
        r = r.contiguous() if r is not None else None
        rows = x.numel() // d
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        seed = next_seed() if (p_drop > 0 and r is not None) else 0
        ctx.set_materialize_grads(False)
        if out32:
            y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            y16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
            rdt = dt(r) if r is not None else _lib.dt(y16)
            call("gtos_ln_residual_fwd2", dt(x), rdt, dt(y), rows, d, ptr(x), ptr(r), float(p_drop), seed, ptr(gamma), ptr(beta),
                 float(eps), ptr(y), ptr(y16), ptr(mean), ptr(rstd), stream())
        else:
            y, y16 = torch.empty_like(x), None
            call("gtos_ln_residual_fwd", dt(x), rows, d, ptr(x), ptr(r), float(p_drop), seed, ptr(gamma), ptr(beta),
                 float(eps), ptr(y), ptr(mean), ptr(rstd), stream())
        ctx.save_for_backward(x, r, mean, rstd)
        ctx.cfg = (p_drop, seed, gamma, beta, out32)
        return (y, y16) if out32 else y

    @staticmethod
    def backward(ctx, dy, dy16=None):
        x, r, mean, rstd = ctx.saved_tensors
        p_drop, seed, gamma, beta, out32 = ctx.cfg
        d = x.shape[-1]
        rows = x.numel() // d
        if dy is None and dy16 is None:
            return None, None, None, None, None, None, None
        need_dr = r is not None and ctx.needs_input_grad[1]
        tg, tb = _grad_target(gamma), _grad_target(beta)
        dg = tg if tg is not None else torch.zeros(d, dtype=torch.float32, device=x.device)
        db = tb if tb is not None else torch.zeros(d, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        if out32:
            dy = dy.contiguous().float() if dy is not None else None
            dy16 = dy16.contiguous() if dy16 is not None else None
            # the gradient of the sub-layer branch equals dx unless dropout masks it; it leaves in r's dtype (bf16), so it is always
            # its own buffer when the stream is fp32
            dr = torch.empty_like(r) if need_dr else None
            rdt = dt(r) if r is not None else 1
            call("gtos_ln_residual_bwd2", dt(x), rdt, 0, rows, d, ptr(dy), ptr(dy16), ptr(x), ptr(r), float(p_drop), seed, ptr(gamma),
                 ptr(mean), ptr(rstd), ptr(dx), ptr(dr), ptr(dg), ptr(db), stream())
        else:
            dy = dy.contiguous()
            dr = torch.empty_like(x) if (need_dr and p_drop > 0) else None
            call("gtos_ln_residual_bwd", dt(x), rows, d, ptr(dy), ptr(x), ptr(r), float(p_drop), seed, ptr(gamma),
                 ptr(mean), ptr(rstd), ptr(dx), ptr(dr), ptr(dg), ptr(db), stream())
            if need_dr and dr is None:
                dr = dx
        return dx, dr, (None if tg is not None else dg), (None if tb is not None else db), None, None, None


def layer_norm_residual(x, r, gamma, beta, p_drop=0.0, eps=1e-5):
    return LayerNormResidualFn.apply(x, r, gamma, beta, float(p_drop), eps, False)


# bf16 activations on an fp32 residual stream (round 4).  The reference's post-LN layers keep x = LayerNorm(x + sublayer(x)) in fp32;
# storing that stream in bf16 costs one rounding of an |x| <= 4 value per layer (1.6e-2 absolute) that the next layer's residual add
# carries on -- the measured bf16 output error of the graph encoder (8e-3 .. 1.3e-2 relative) came from there, not from the bf16
# MFMA operands.  The stream is [rows, d]: 6.6 MB per layer at C2 next to a 0.9 GB relation stream.  (False = the bf16 stream of rounds 1-3: graph-encoder output error 8e-3..1.3e-2 instead of 1.1e-3..1.8e-3 at the same step time.)
FP32_STREAM = True


def split_stream(x, cd):
    """(residual-stream tensor, GEMM operand in the compute dtype ``cd``) of a layer input: in fp32 mode both are x; in bf16 mode the
    operand is the bf16 twin an earlier ``layer_norm_stream`` attached to its fp32 output, or a cast of x."""
    if cd == torch.float32:
        x = x if x.dtype == cd else x.to(cd)
        return x, x
    twin = getattr(x, "_gtos_twin", None)
    if twin is not None and twin.dtype == cd and twin.shape == x.shape:
        return (x if FP32_STREAM else twin), twin
    x16 = x if x.dtype == cd else x.to(cd)
    return (x if (FP32_STREAM and x.dtype == torch.float32) else x16), x16


def join_stream(x32, x16):
    """What a layer hands on: the stream tensor, carrying its compute-dtype twin so that the next consumer needs no cast."""
    if x32 is not x16:
        x32._gtos_twin = x16
    return x32


def layer_norm_stream(xs, r, gamma, beta, p_drop, eps, cd):
    """LayerNorm(xs + dropout(r)) -> (stream tensor, compute-dtype operand): the fp32 pair in bf16 mode with FP32_STREAM, else one
    tensor in ``cd`` returned twice."""
    if cd == torch.bfloat16 and FP32_STREAM:
        return LayerNormResidualFn.apply(xs, r, gamma, beta, float(p_drop), eps, True)
    y = LayerNormResidualFn.apply(xs if xs.dtype == cd else xs.to(cd), r, gamma, beta, float(p_drop), eps, False)
    return y, y


def check_attention_shape(embed_dim, num_heads):
    """The attention kernels' shape boundary (include/gtos_hip.h, gtos_rel_attn_fwd): refuse at construction, loudly, instead
    of failing at the first forward.  The reference asks embed_dim % heads == 0 (graph_transformer.py:75); the kernels read 8
    channels per lane, so the head width must also be a multiple of 8 (and at most 512).  Power-of-two embed_dim <= 512 with a
    power-of-two head width (every shipped configuration) runs on the fast lane map, everything else on the generic one."""
    hd = embed_dim // max(1, num_heads)
    ok = num_heads > 0 and embed_dim % num_heads == 0 and hd % 8 == 0 and hd <= 512
    if not ok:
        raise _lib.GtosHipError("attention shape embed_dim=%d, heads=%d is outside the gfx950 kernels' boundary: embed_dim must "
                                "be a multiple of heads and the head width a multiple of 8, at most 512" % (embed_dim, num_heads))


def _u8(mask):
    """Mask as uint8 bytes for the kernels.  A bool tensor IS one byte per element holding 0/1: reinterpreted, not copied (the
    copy was one ATen launch per attention call)."""
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


class FactoredRelation:
    """The relation operand in factored form: ``bank`` [R,d] (relation-encoder output, one row per distinct
    label path) + ``relation`` [n,n,B] int64 type ids exactly as the reference batches carry them
    (generator/data.py:164-176).  Equivalent to the dense ``bank.index_select(0, relation.view(-1)).view(n,n,B,d)``
    of generator/generator.py:79, which is never materialised.  Index preparation happens once per batch."""

    CHUNK = 32

    def __init__(self, bank, relation, index=None):
        """``index``: the batch's host-built gtos_amd.relindex.RelationIndex (``batch['relation_index']``); without one the
        same arrays are derived here with device sort / search ops."""
        require_cuda(bank, relation)
        self.bank = bank
        self.grad_group = GradAccumGroup()
        n, n2, B = relation.shape
        assert n == n2
        self.n, self.B = n, B
        self._proj = {}
        if not torch.is_grad_enabled():
            # forward only (inference, under no_grad): the kernels read the query-major ids, nothing else
            self.idx_q = (index.idx_q if index is not None and index.matches(bank, relation)
                          else relation.permute(1, 2, 0).contiguous().to(torch.int32))
            self.idx_k = self.pair_sorted = self.chunk_type = self.chunk_start = self.chunk_count = self.chunk_slot = None
            self.xcd_off = self.heavy_types = None
            self.nchunks = 0
            return
        if index is not None and index.matches(bank, relation):
            self.idx_q, self.idx_k, self.pair_sorted = index.idx_q, index.idx_k, index.pair_sorted
            self.chunk_type, self.chunk_start = index.chunk_type, index.chunk_start
            self.chunk_count, self.chunk_slot = index.chunk_count, index.chunk_slot
            self.xcd_off, self.heavy_types, self.nchunks = index.xcd_off, index.heavy_types, index.nchunks
            return
        rel = relation.contiguous()
        self.idx_q = rel.permute(1, 2, 0).contiguous().to(torch.int32)    # [i,b,j] = relation[j,i,b]
        self.idx_k = rel.permute(0, 2, 1).contiguous().to(torch.int32)    # [j,b,i]
        flat = rel.reshape(-1)
        R = bank.shape[0]
        # type ids fit 32 bits: a 32-bit key sort is about twice as fast as the int64 one; the per-type counts come from
        # the sorted keys by binary search instead of an atomics histogram
        skeys, order = torch.sort(flat.to(torch.int32), stable=True)
        bounds = torch.searchsorted(skeys, torch.arange(R + 1, device=flat.device, dtype=torch.int32))
        counts = bounds[1:] - bounds[:-1]
        starts = bounds[:-1]
        nch = torch.clamp((counts + self.CHUNK - 1) // self.CHUNK, min=1)   # empty types get one empty chunk (writes zeros)
        ctype = torch.repeat_interleave(torch.arange(R, device=flat.device), nch)
        first = torch.cumsum(nch, 0) - nch
        local = torch.arange(ctype.numel(), device=flat.device) - first[ctype]
        heavy_types = torch.nonzero(nch > 1).flatten()
        slot_of_type = torch.full((R,), -1, dtype=torch.int64, device=flat.device)
        slot_of_type[heavy_types] = torch.arange(heavy_types.numel(), device=flat.device)
        self.pair_sorted = order.to(torch.int32)
        c_type = ctype.to(torch.int32)
        c_start = (starts[ctype] + local * self.CHUNK).to(torch.int32)
        c_count = torch.clamp(counts[ctype] - local * self.CHUNK, min=0, max=self.CHUNK).to(torch.int32)
        c_slot = slot_of_type[ctype].to(torch.int32)
        # group the chunks by the XCD that owns the graph of their first pair (graph b -> XCD b // (B/8), the attention
        # kernels' map), so the bank-gradient kernel gathers q/k rows from that XCD's private L2
        first = self.pair_sorted[c_start.clamp(max=flat.numel() - 1).long()]
        gb = first % B
        xcd = (gb // (B // 8)) if B % 8 == 0 else (gb % 8)
        xs, perm = torch.sort(xcd.to(torch.int32), stable=True)
        self.xcd_off = torch.searchsorted(xs, torch.arange(9, device=flat.device, dtype=torch.int32)).to(torch.int32)
        self.chunk_type, self.chunk_start = c_type[perm], c_start[perm]
        self.chunk_count, self.chunk_slot = c_count[perm], c_slot[perm]
        self.heavy_types = heavy_types.to(torch.int32)
        self.nchunks = int(ctype.numel())

    def prefetch_projections(self, attns):
        """relation_in_proj(bank) of EVERY layer depends only on the bank, not on the layer chain: compute them on the side
        stream, ahead of the layers, so the MFMA GEMMs run beside the (HBM-bound) attention kernels and the small
        per-layer kernels of the main stream.  autograd runs a node's backward on the stream of its forward, so the
        projection's dX / dW GEMMs overlap the main chain in backward as well."""
        self._proj = {}
        if not (PROJ_SIDE and self.bank.is_cuda and side_ok(self.bank.device) and torch.is_grad_enabled()):
            return
        dev = self.bank.device
        main, side = torch.cuda.current_stream(dev), side_stream(dev)
        if any(m.compute_dtype != self.bank.dtype for m in attns):
            return
        if sum(self.bank.shape[0] * m.relation_in_proj.weight.shape[0] for m in attns) * self.bank.element_size() > PROJ_SIDE_MAX_BYTES:
            return
        side.wait_stream(main)
        self.bank.record_stream(side)
        with torch.cuda.stream(side):
            for m in attns:
                y = linear(self.bank, m.relation_in_proj.weight, group=self.grad_group)
                ev = torch.cuda.Event()
                ev.record(side)
                self._proj[id(m)] = (y, ev)

    def take_projection(self, attn):
        """The prefetched [R,2d] projection for this attention module (the main stream is made to wait for it), or None."""
        hit = self._proj.pop(id(attn), None)
        if hit is None:
            return None
        y, ev = hit
        main = torch.cuda.current_stream(y.device)
        main.wait_event(ev)
        y.record_stream(main)
        return y


class RelAttnFn(torch.autograd.Function):
    """Fused (relation-aware) attention core.  qsrc [T,B,Cq] holds q at channel offset q_off; kvsrc [S,B,Ckv]
    (or qsrc itself when None) holds k, v at k_off, v_off.  rel: None | rarb [S,T,B,2d] | bank-projection [R,2d].
    Returns (o [T,B,d], w [T,S,B,H] or None)."""

    @staticmethod
    def forward(ctx, qsrc, kvsrc, rel, fact, offs, d, H, scale, key_pad, attn_mask, p_drop, need_w):
        require_cuda(qsrc, kvsrc, rel)
        q_off, k_off, v_off = offs
        kv = qsrc if kvsrc is None else kvsrc
        assert qsrc.is_contiguous() and kv.is_contiguous()
        T_, B, Cq = qsrc.shape
        S, _, Ckv = kv.shape
        mode = 0 if rel is None else (2 if fact is not None else 1)
        if mode == 1:
            assert rel.is_contiguous() and tuple(rel.shape) == (S, T_, B, 2 * d)
        o = torch.empty((T_, B, d), dtype=qsrc.dtype, device=qsrc.device)
        lse = torch.empty((T_, B, H), dtype=torch.float32, device=qsrc.device)
        w = torch.empty((T_, S, B, H), dtype=torch.float32, device=qsrc.device) if need_w else None
        seed = next_seed() if p_drop > 0 else 0
        es = qsrc.element_size()
        with _Timed("rel_attn_fwd_mode%d" % mode):
            call("gtos_rel_attn_fwd", dt(qsrc), mode, T_, S, B, H, d,
                 qsrc.data_ptr() + q_off * es, Cq, kv.data_ptr() + k_off * es, Ckv, kv.data_ptr() + v_off * es, Ckv,
                 ptr(rel), ptr(fact.idx_q) if fact is not None else None, ptr(key_pad), ptr(attn_mask),
                 float(scale), float(p_drop), seed, ptr(o), d, ptr(lse), ptr(w), stream())
        # (mode 2, opt-in) the projection is not kept: its source -- the bank and the compute-dtype weight, both alive anyway -- is
        src = getattr(rel, "_gtos_proj_src", None) if (mode == 2 and PROJ_RECOMPUTE) else None
        ctx.save_for_backward(qsrc, kvsrc, rel if src is None else None, o, lse, w, key_pad, attn_mask)
        ctx.cfg = (fact, offs, d, H, scale, p_drop, seed, mode)
        ctx.proj_src = src
        return o, w

    @staticmethod
    def backward(ctx, d_o, d_w):
        qsrc, kvsrc, rel, o, lse, w, key_pad, attn_mask = ctx.saved_tensors
        fact, (q_off, k_off, v_off), d, H, scale, p_drop, seed, mode = ctx.cfg
        if ctx.proj_src is not None:
            bank_, w_ = ctx.proj_src
            rel = gemm(bank_, w_, trans_b=True)              # the forward's product again: same kernel, same operands, same bits
        kv = qsrc if kvsrc is None else kvsrc
        T_, B, Cq = qsrc.shape
        S, _, Ckv = kv.shape
        d_o = d_o.contiguous()
        if d_w is not None:
            d_w = d_w.contiguous().float()
        # every channel of qsrc/kvsrc that is not q/k/v (none in practice) must come back zero
        full_q = (Cq == d) if kvsrc is not None else (Cq == 3 * d)
        dqsrc = (torch.empty_like if full_q else torch.zeros_like)(qsrc)
        dkv = dqsrc if kvsrc is None else (torch.empty_like if Ckv == 2 * d else torch.zeros_like)(kvsrc)
        if mode == 1:
            d_rel = torch.empty_like(rel)
        elif mode == 2:
            # a block of the layers' shared gradient slab; the attention backward writes the rows of the singleton types
            # (flagged ids of the host index), the type-major pass below the rest
            d_rel = fact.grad_group.grad_slice(rel) if ctx.needs_input_grad[2] else None
            if d_rel is None:
                d_rel = torch.empty_like(rel)
        else:
            d_rel = None
        ldr = d_rel.stride(0) if mode == 2 else 0
        pd = torch.empty((T_, S, B, H), dtype=torch.float32, device=qsrc.device)
        gs = torch.empty_like(pd)
        es = qsrc.element_size()
        with _Timed("rel_attn_bwd_q+kv_mode%d" % mode, detail=True):
            call("gtos_rel_attn_bwd", dt(qsrc), mode, T_, S, B, H, d,
                 qsrc.data_ptr() + q_off * es, Cq, kv.data_ptr() + k_off * es, Ckv, kv.data_ptr() + v_off * es, Ckv,
                 ptr(rel), ptr(fact.idx_q) if fact is not None else None, ptr(fact.idx_k) if fact is not None else None,
                 ptr(key_pad), ptr(attn_mask), float(scale), float(p_drop), seed,
                 ptr(o), d, ptr(lse), ptr(w), ptr(d_o), d, ptr(d_w),
                 dqsrc.data_ptr() + q_off * es, Cq, dkv.data_ptr() + k_off * es, Ckv, dkv.data_ptr() + v_off * es, Ckv,
                 ptr(d_rel), ldr, ptr(pd), ptr(gs), stream())
        if mode == 2:
            nh = int(fact.heavy_types.numel())
            heavy = torch.zeros((max(nh, 1), 2 * d), dtype=torch.float32, device=rel.device)
            with _Timed("rel_attn_bwd_bank", detail=True):
                call("gtos_rel_attn_bwd_bank", dt(qsrc), T_, B, H, d,
                     qsrc.data_ptr() + q_off * es, Cq, kv.data_ptr() + k_off * es, Ckv,
                     ptr(rel), ptr(gs), ptr(fact.pair_sorted), ptr(fact.chunk_type), ptr(fact.chunk_start),
                     ptr(fact.chunk_count), ptr(fact.chunk_slot), ptr(fact.xcd_off), fact.nchunks, ptr(d_rel), ldr, ptr(heavy), stream())
            if nh and d_rel.dtype == torch.bfloat16:         # heavy types: fp32 slots rounded straight into their bank rows
                call("gtos_segment_sum_finish", nh, ptr(fact.heavy_types), ptr(heavy), 2 * d, ptr(d_rel), ldr, stream())
            elif nh:
                d_rel[fact.heavy_types.long()] = heavy[:nh].to(d_rel.dtype)
        return dqsrc, (dkv if kvsrc is not None else None), d_rel, None, None, None, None, None, None, None, None, None


def attention_core(qsrc, kvsrc, offs, d, H, scale, rel=None, fact=None, key_pad=None, attn_mask=None,
                   p_drop=0.0, need_weights=False):
    return RelAttnFn.apply(qsrc, kvsrc, rel, fact, offs, d, H, scale, _u8(key_pad), _u8(attn_mask),
                           float(p_drop), need_weights)


def factored_eval_relation(bank, relation):
    """Eval-mode relation operand WITHOUT the dense [n,n,B,d] tensor: relation [n,n,B,K] lists up to K alternative
    shortest paths per pair (0 = <PAD>), and the reference feeds the graph encoder the mean of their bank rows
    (generator/generator.py:83-88).  Many pairs share their K-tuple of types, so the distinct tuples become a derived bank
    (mean of the tuple's rows, same arithmetic as the reference: bank row 0 zeroed, sum, divide by the clamped count) and
    every pair an id into it: the encoder then runs on the factored operand exactly as in training."""
    n, n2, B, K = relation.shape
    tuples, inv = torch.unique(relation.reshape(-1, K), dim=0, return_inverse=True)
    bank_c = relation_gather_mean(bank, tuples, zero_row0=True)                 # [C, d]
    return FactoredRelation(bank_c, inv.view(n, n2, B))


def relation_gather_mean(bank, idx, zero_row0):
    """Dense relation tensor from the bank: train lookup (idx [n,n,B], generator/generator.py:79) or the eval
    mean over alternative shortest paths (idx [n,n,B,K], zero_row0=True, generator/generator.py:83-88).
    Forward only (the train path with gradients uses FactoredRelation)."""
    require_cuda(bank, idx)
    K = idx.shape[-1] if zero_row0 else 1
    lead = idx.shape[:-1] if zero_row0 else idx.shape
    P = idx.numel() // K
    bank = bank.contiguous()
    out = torch.empty((P, bank.shape[1]), dtype=bank.dtype, device=bank.device)
    call("gtos_relation_gather_mean", dt(bank), P, K, bank.shape[1], ptr(bank), ptr(idx.contiguous()), int(zero_row0),
         ptr(out), stream())
    return out.view(*lead, bank.shape[1])


def embed_bwd_workspace(n, V, dim_pad, device):
    """Workspace gtos_embed_rows_bwd sums its per-block tables through (small tables only): one table per 512-row block."""
    if V * dim_pad * 4 > 60 * 1024 or n <= 8 * 512:
        return None
    return torch.empty((min(2048, (n + 511) // 512), V * dim_pad), dtype=torch.float32, device=device)


class EmbedRowsFn(torch.autograd.Function):
    """x[n, 0:dim_pad] = dropout(table[tokens[n]]) zero-padded, in the compute dtype: nn.Embedding + F.dropout of
    RelationEncoder (generator/encoder.py:99-100).  Backward scatters into the (small) table through LDS."""

    @staticmethod
    def forward(ctx, tokens, table, dim_pad, p_drop, dtype, pad_idx=None):
        require_cuda(tokens, table)
        tokens = tokens.contiguous()
        n, (V, dim) = tokens.numel(), table.shape
        out = torch.empty((n, dim_pad), dtype=dtype, device=table.device)
        seed = next_seed() if p_drop > 0 else 0
        call("gtos_embed_rows_fwd", dt(out), n, dim, dim_pad, ptr(tokens), ptr(table), ptr(out), float(p_drop), seed, stream())
        ctx.save_for_backward(tokens)
        ctx.cfg = (table, dim_pad, p_drop, seed, pad_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        tokens, = ctx.saved_tensors
        table, dim_pad, p_drop, seed, pad_idx = ctx.cfg
        dout = dout.contiguous()
        V, dim = table.shape
        tgt = _grad_target(table)
        dtab = None
        if tgt is None:
            tgt = dtab = torch.zeros(table.shape, dtype=torch.float32, device=table.device)
        ws = embed_bwd_workspace(tokens.numel(), V, dim_pad, dout.device)
        call("gtos_embed_rows_bwd", dt(dout), tokens.numel(), V, dim, dim_pad, ptr(tokens), ptr(dout), ptr(tgt),
             float(p_drop), seed, ptr(ws), 0 if ws is None else ws.numel() * 4, stream())
        if pad_idx is not None:
            tgt[pad_idx].zero_()          # nn.Embedding(padding_idx=...): the padding row takes no gradient
        return None, dtab, None, None, None, None


def embed_rows(tokens, table, dim_pad, p_drop, dtype, pad_idx=None):
    return EmbedRowsFn.apply(tokens, table, dim_pad, float(p_drop), dtype, pad_idx)


class HighwayGateFn(torch.autograd.Function):
    """out = sigmoid(gate) * x + (1 - sigmoid(gate)) * relu(new_x) with [new_x | gate] = y = layer(x): the elementwise half of a
    Highway layer (generator/encoder.py:141-149), one kernel per direction."""

    @staticmethod
    def forward(ctx, y, x):
        require_cuda(y, x)
        D = x.shape[-1]
        y2, x2 = y.reshape(-1, 2 * D).contiguous(), x.reshape(-1, D).contiguous()
        out = torch.empty_like(x2)
        call("gtos_highway_fwd", dt(x2), x2.shape[0], D, ptr(y2), ptr(x2), ptr(out), stream())
        ctx.save_for_backward(y2, x2)
        ctx.shapes = (y.shape, x.shape)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        y2, x2 = ctx.saved_tensors
        D = x2.shape[1]
        g = dout.reshape(-1, D).contiguous()
        dy, dx = torch.empty_like(y2), torch.empty_like(x2)
        call("gtos_highway_bwd", dt(x2), x2.shape[0], D, ptr(y2), ptr(x2), ptr(g), ptr(dy), ptr(dx), stream())
        return dy.view(ctx.shapes[0]), dx.view(ctx.shapes[1])


def highway_gate(y, x):
    return HighwayGateFn.apply(y, x)


class MaxReluFn(torch.autograd.Function):
    """relu(y.max(dim=1)[0]) for y [N, L, F]: the char CNN's max over time + ReLU (generator/encoder.py:169-172)."""

    @staticmethod
    def forward(ctx, y):
        require_cuda(y)
        N, L, F_ = y.shape
        y = y.contiguous()
        out = torch.empty((N, F_), dtype=y.dtype, device=y.device)
        arg = torch.empty((N, F_), dtype=torch.uint8, device=y.device)
        call("gtos_max_relu_fwd", dt(y), N, L, F_, ptr(y), ptr(out), ptr(arg), stream())
        ctx.save_for_backward(out, arg)
        ctx.L = L
        return out

    @staticmethod
    def backward(ctx, dout):
        out, arg = ctx.saved_tensors
        N, F_ = out.shape
        g = dout.contiguous()
        dy = torch.empty((N, ctx.L, F_), dtype=out.dtype, device=out.device)
        call("gtos_max_relu_bwd", dt(out), N, ctx.L, F_, ptr(out), ptr(arg), ptr(g), ptr(dy), stream())
        return dy


def max_relu(y):
    return MaxReluFn.apply(y)


class TokenRowFn(torch.autograd.Function):
    """dropout(cat([feat, table[tok]], -1)) zero-padded to a multiple of 8 columns, in feat's dtype: torch.cat + nn.Embedding +
    F.dropout of TokenEncoder.forward (generator/encoder.py:196-199).  Backward: masked gradient back to ``feat`` and scattered
    into the (fp32) embedding table; the padding row of the table takes none."""

    @staticmethod
    def forward(ctx, feat, tok, table, p_drop, pad_idx):
        require_cuda(feat, tok, table)
        lead, Cc = feat.shape[:-1], feat.shape[-1]
        Ct = table.shape[1]
        Cp = (Cc + Ct + 7) // 8 * 8
        f2 = feat.reshape(-1, Cc).contiguous()
        tk = tok.reshape(-1).contiguous()
        out = torch.empty((f2.shape[0], Cp), dtype=feat.dtype, device=feat.device)
        seed = next_seed() if p_drop > 0 else 0
        call("gtos_token_row_fwd", dt(out), f2.shape[0], Cc, Ct, Cp, ptr(f2), ptr(tk), ptr(table), ptr(out), float(p_drop), seed, stream())
        ctx.save_for_backward(tk)
        ctx.cfg = (table, Cc, Ct, Cp, p_drop, seed, pad_idx, feat.shape)
        return out.view(*lead, Cp)

    @staticmethod
    def backward(ctx, dout):
        (tk,) = ctx.saved_tensors
        table, Cc, Ct, Cp, p_drop, seed, pad_idx, fshape = ctx.cfg
        g = dout.reshape(-1, Cp).contiguous()
        dfeat = torch.empty((g.shape[0], Cc), dtype=g.dtype, device=g.device) if ctx.needs_input_grad[0] else None
        tgt, dtab = None, None
        if ctx.needs_input_grad[2]:
            tgt = _grad_target(table)
            if tgt is None:
                tgt = dtab = torch.zeros(table.shape, dtype=torch.float32, device=table.device)
        call("gtos_token_row_bwd", dt(g), g.shape[0], Cc, Ct, Cp, ptr(g), ptr(tk), ptr(dfeat), ptr(tgt), float(p_drop), seed, stream())
        if tgt is not None and pad_idx is not None:
            tgt[pad_idx].zero_()          # nn.Embedding(padding_idx=...): the padding row takes no gradient
        return (dfeat.view(fshape) if dfeat is not None else None), None, dtab, None, None


def token_row(feat, tok, table, p_drop, pad_idx=None):
    return TokenRowFn.apply(feat, tok, table, float(p_drop), pad_idx)


class PermuteRowsFn(torch.autograd.Function):
    """y = x[perm] for a PERMUTATION perm with inverse inv: the backward is the gather g[inv] instead of the
    atomics-based index_add_ that autograd derives for a general index_select."""

    @staticmethod
    def forward(ctx, x, perm, inv):
        ctx.save_for_backward(inv)
        return x.index_select(0, perm)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return g.index_select(0, inv), None, None


def permute_rows(x, perm, inv):
    return PermuteRowsFn.apply(x, perm, inv)


class CopyNllFn(torch.autograd.Function):
    """nll[t,b] = -log(gen_gate * softmax(logits)[target] + copy_gate * sum_{s: cp_seq[s,b] == target} align[t,b,s] + 1e-12),
    0 at padded targets: the generate/copy mixture of TokenGenerator evaluated only where the loss looks
    (generator/decoder.py:40-63), one kernel forward, one backward."""

    @staticmethod
    def forward(ctx, logits, div, align, cp_seq, target, pad_idx):
        require_cuda(logits, div, align, cp_seq, target)
        T_, B, V = logits.shape
        S = cp_seq.shape[0]
        logits, div = logits.contiguous(), div.contiguous()
        align = align.float().contiguous()
        cp_seq, target = cp_seq.contiguous(), target.contiguous()
        nll = torch.empty((T_, B), dtype=torch.float32, device=logits.device)
        lse, p = torch.empty_like(nll), torch.empty_like(nll)
        call("gtos_copy_nll_fwd", dt(logits), T_, B, V, S, ptr(logits), V, ptr(div), ptr(align), ptr(cp_seq), ptr(target),
             int(pad_idx), ptr(nll), ptr(lse), ptr(p), stream())
        ctx.save_for_backward(logits, div, align, cp_seq, target, lse, p)
        ctx.pad_idx = int(pad_idx)
        return nll

    @staticmethod
    def backward(ctx, d_nll):
        logits, div, align, cp_seq, target, lse, p = ctx.saved_tensors
        T_, B, V = logits.shape
        S = cp_seq.shape[0]
        d_logits, d_div = torch.empty_like(logits), torch.empty_like(div)
        d_align = torch.empty_like(align)
        g = d_nll.float().contiguous()          # (an expanded view after .sum(0)): a named local that outlives the launch below
        call("gtos_copy_nll_bwd", dt(logits), T_, B, V, S, ptr(logits), V, ptr(div), ptr(align), ptr(cp_seq), ptr(target),
             ctx.pad_idx, ptr(lse), ptr(p), ptr(g), ptr(d_logits), ptr(d_div), ptr(d_align), stream())
        del g
        return d_logits, d_div, d_align, None, None, None


def copy_nll(logits, div, align, cp_seq, target, pad_idx):
    return CopyNllFn.apply(logits, div, align, cp_seq, target, pad_idx)


def copy_log_likelihood(logits, div, align, cp_seq, tot_ext):
    """Inference form: ll [T,B,tot_ext] fp32 = log(mixture + 1e-12) over the vocabulary and the batch's copy ids."""
    require_cuda(logits, div, align, cp_seq)
    T_, B, V = logits.shape
    logits, div = logits.contiguous(), div.contiguous()
    align = align.float().contiguous()
    cp_seq = cp_seq.contiguous()
    tot = max(int(tot_ext), V)
    ll = torch.empty((T_, B, tot), dtype=torch.float32, device=logits.device)
    call("gtos_copy_ll_fwd", dt(logits), T_, B, V, cp_seq.shape[0], tot, ptr(logits), V, ptr(div), ptr(align), ptr(cp_seq),
         ptr(ll), stream())
    return ll
