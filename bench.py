#!/usr/bin/env python
"""Headline benchmark: graphs/s of one full training step (fwd + bwd + gradient all-reduce + clip + Adam) of the gtos
Generator on synthetic 100-node AMR graphs, batch 64 per GPU, bf16 activations (BASELINE.json configs[1] = "C2").

    python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU; launched by torch.distributed.run, or self-launching)

Prints ONE JSON line (rank 0).  Also reports
  roofline      relation-attention forward kernel: algorithmic bytes per launch (P*2d*s + 4*n*B*d*s + n*B,
                SURVEY.md section 8d) / average launch duration measured with HIP events on the launch stream inside the
                timed region, against 8 TB/s HBM3E;
  cpu_baseline  the CPU oracle (a port of the reference pinned to its golden vectors) timed on the host cores on a
                bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
T_START = time.time()


def log(*a):
    if os.environ.get("GTOS_BENCH_VERBOSE"):
        print("[bench %7.1fs]" % (time.time() - T_START), *a, file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def rank_cores(world):
    """Host threads one of `world` ranks on this box may use: its share of the usable cores, at least one."""
    return max(1, host_cores() // max(1, world))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dense", action="store_true", help="dense relation[n,n,B,d] signature instead of the factored form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-graphs", type=int, default=4,
                    help="micro-batch of the CPU baseline (default 4 keeps the leg near 45 s; SURVEY 8d's protocol is --cpu-graphs 8 "
                         "--cpu-steps 5 --cpu-budget 600, committed once per round under profiles/)")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CPU steps of the full-step leg (after 1 warm-up; time-capped)")
    ap.add_argument("--cpu-budget", type=float, default=45.0, help="wall-clock cap of the CPU baseline in seconds")
    ap.add_argument("--cpu-warmup", type=int, default=1, help="untimed CPU warm-up steps (SURVEY 8d: 2)")
    ap.add_argument("--cpu-protocol", action="store_true",
                    help="SURVEY 8d's CPU protocol live in this run: --cpu-graphs 8 --cpu-warmup 2 --cpu-steps 5 --cpu-budget 600 (about five "
                         "minutes of host time on 16 cores: more than the default run may take, hence a flag)")
    ap.add_argument("--decode", action="store_true",
                    help="secondary benchmark (SURVEY 8f rank 2): beam search over the K/V-cached decoder instead of the train step")
    ap.add_argument("--beam", type=int, default=8)
    ap.add_argument("--max-steps", type=int, default=50)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: B graphs per GPU (the translator's policy, translator/train.py:201); strong: B graphs in TOTAL, "
                         "B/N per GPU (the generator divides its batch budget by the world size, generator/train.py:183)")
    ap.add_argument("--fresh-batches", action="store_true",
                    help="loader in the loop: a NEW batch every step, assembled by gtos_amd.data.AMRLoader (C++ relation batch, path "
                         "tries, relation index) on Prefetcher worker threads and uploaded on a copy stream, like the reference's "
                         "training loop (generator/train.py:136-140, generator/data.py:290-316); default: one pre-built device batch")
    ap.add_argument("--workers", type=int, default=0,
                    help="--fresh-batches: loader workers per rank (default: 1 with the relation section on the device, 4 with --host-relations)")
    ap.add_argument("--loader", default="processes", choices=["processes", "threads"],
                    help="--fresh-batches: worker processes (own interpreter each: no GIL contention with the launch thread) or threads")
    ap.add_argument("--depth", type=int, default=0,
                    help="--fresh-batches: batches in flight (default 2 per worker: a batch spends ~0.2 s between job hand-out and "
                         "upload, three step times at C2, so fewer than that starves the consumer whatever the worker count)")
    ap.add_argument("--relbatch-threads", type=int, default=2, help="--fresh-batches: threads inside one relation-batch build")
    ap.add_argument("--device-tries", nargs="?", const="torch", default="", choices=["torch", "hip"],
                    help="--fresh-batches: the workers build the relation index only; the path tries are built on the GPU on the loader's "
                         "copy stream: 'torch' (default when the flag is given bare) = torch ops (gtos_amd.pathtrie_device), 'hip' = the "
                         "staged HIP builder (gtos_amd.pathtrie_hip: rocPRIM sorts / scans + stage kernels, 2 host reads)")
    ap.add_argument("--host-relations", action="store_true",
                    help="--fresh-batches: the loader workers build relation / bank / index / tries with the C++ host builders (the route of "
                         "rounds 1-3: 0.10 s of host time per C2 batch, 3-4 worker processes per GPU).  Default since round 4: everything on "
                         "the device (= --device-relations --prep-in-worker, ONE worker process), as the loaders themselves default to")
    ap.add_argument("--relation-masks", default="path", choices=["node", "path"],
                    help="training-mode dropout masks of the RelationEncoder: 'path' = drawn per (path, position) like the reference "
                         "(generator/encoder.py:91-92,105; the library default and, since the end of round 4, this script's: one GRU row "
                         "per path and position), 'node' = drawn per node of the prefix / suffix trie and shared by the paths through it "
                         "(opt-in, gtos_amd.encoder.set_relation_mask_sharing: layer 0 once per trie node, layer-1 input gates from "
                         "per-node tables; the headline of rounds 2-3 and of this round's earlier records).  The other mode is measured "
                         "right after the timed region and reported as `node_masks` / `reference_masks`")
    ap.add_argument("--no-masks-leg", action="store_true", help="skip the leg that measures the other --relation-masks mode")
    ap.add_argument("--graph-leg", action="store_true",
                    help="after the timed region (N = 1): the same step replayed from a hipGraph (train.GraphedStep).  Opt-in: on ROCm "
                         "7.2 the capture of the multi-stream C2 step crashes inside hipStreamEndCapture (C1 captures and replays)")
    ap.add_argument("--no-loader-leg", action="store_true",
                    help="skip the loader-in-the-loop leg that follows the timed region of the default (pre-built batch) run")
    ap.add_argument("--device-relations", action="store_true",
                    help="--fresh-batches: the loader ships the flattened graphs only (index_prep='device_all'); relation / bank / length "
                         "(gtos_amd.relbatch_hip), the relation index (gtos_amd.relindex_hip) and the tries (gtos_amd.pathtrie_hip) are built on "
                         "the GPU on the loader's copy stream -- the host keeps the token / character tensors")
    ap.add_argument("--prep-in-worker", action="store_true",
                    help="--fresh-batches with --device-tries / --device-relations: the device-side preparation (and its host reads) on the "
                         "loader's upload thread instead of the training thread")
    ap.add_argument("--pool", type=int, default=0, help="--fresh-batches: graphs in the per-rank item pool (default 4 batches)")
    ap.add_argument("--prewarm-seconds", type=float, default=20.0,
                    help="untimed device pre-warm BEFORE the --warmup steps: windows of 5 training steps until two consecutive windows "
                         "agree within 1.5 %% or this many seconds have passed (0 = off).  A box that has been idle runs its first "
                         "minute 5-20 %% slow (measured: four back-to-back runs of this bench on a fresh box 74.7 / 67.5 / 64.5 / 61.7 ms per "
                         "step); on a warm box this costs two windows")
    ap.add_argument("--prewarm-min-seconds", type=float, default=8.0,
                    help="the pre-warm runs at least this long before the two-windows criterion may end it: two windows of 5 steps agree within "
                         "1.5 %% after a second of work even on a box whose step time is still drifting down slowly (0 = criterion only, as in "
                         "rounds 2-3)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only check the rank launch / rendezvous (gloo, no GPU): every rank prints its rank and exits")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(cfg_name, graphs, steps, budget_s=75.0, warmup=1):
    """Oracle (kind 'port': the CPU restatement pinned to the reference's golden vectors) on the host cores, fp32, on a
    micro-batch of `graphs` graphs of the same config: (i) the full training step, (ii) the graph encoder
    alone (RelationEncoder + GraphTransformer forward + backward).  1 warm-up + up to `steps` timed steps per leg, capped by
    a wall-clock budget so the default run stays bounded."""
    from oracle import gtos_oracle as O
    from gtos_amd import synth
    from gtos_amd.config import generator_args
    from gtos_amd.flat import inverse_sqrt_lr
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu_baseline: cores", cores, "os.cpu_count", os.cpu_count())
    cfg = synth.CONFIGS[cfg_name]
    vocabs = {k: O.VocabSpec(v, 0) for k, v in synth.DEFAULT_VOCAB.items()}
    torch.manual_seed(19940117)
    model = O.Generator(vocabs, depth_size=256 if cfg["kind"] == "dep" else 32, **generator_args(cfg))
    model.train()
    batch, stats = synth.make_config_batch(cfg_name, B=graphs)
    params = [p for p in model.parameters()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    names = [n for n, _ in model.named_parameters()]

    def full_step(i):
        loss = model(batch)
        loss.backward()
        grads = [p.grad for p in params]
        coef, _ = O.clip_coef(grads, 1.0)
        lr = inverse_sqrt_lr(cfg["d"], i, 2000)
        with torch.no_grad():
            for k, p in enumerate(params):
                np_, m[k], v[k] = O.adam_step(p, p.grad * coef, m[k], v[k], lr, 0.0 if O.is_no_decay(names[k]) else 1e-4)
                p.copy_(np_)
                p.grad = None
        return loss.item()

    def encoder_step(i):
        graph, _, probe = model.encode_step(batch)
        (graph.float().sum() + probe.float().sum()).backward()
        for p in params:
            p.grad = None

    def timed(fn, n_steps, budget, warm):
        for _ in range(warmup if warm else 0):
            fn(1)
        t0, k = time.time(), 0
        while k < n_steps and (k == 0 or time.time() - t0 < budget):
            fn(2 + k)
            k += 1
        return (time.time() - t0) / k, k
    t_all = time.time()
    dt_full, k_full = timed(full_step, steps, budget_s * 0.55, True)
    log("cpu_baseline: full step %.1f s x %d" % (dt_full, k_full))
    # the encoder-only leg runs warm (same model, same batch, right after the full steps): no extra warm-up step
    dt_enc, k_enc = timed(encoder_step, max(1, steps // 2), budget_s * 0.2, False)
    return {"value": graphs / dt_full, "unit": "graphs/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "encoder_only": {"value": graphs / dt_enc, "unit": "graphs/s", "timed_steps": k_enc, "s_per_step": round(dt_enc, 2)},
            "sample": ("micro-batch of %d graphs of %s (n=%d, R=%d), fp32, all host cores, full train step (fwd+bwd+clip+Adam), "
                       "%d warm-up + %d timed steps, %.1f s/step; encoder-only leg (RelationEncoder + GraphTransformer fwd+bwd) "
                       "%.1f s/step; %.0f s total") % (
                           graphs, cfg_name, stats["n"], stats["R"], warmup, k_full, dt_full, dt_enc, time.time() - t_all)}


MFMA_PEAK_TFS = 2500.0


def step_ceiling(cfg, stats, s_el=2):
    """SURVEY.md section 8(d), "per-step graphs/s roofline", for THIS batch's n, B, R and mean path length:
        sum over the graph-encoder layers of (Stage R flops + the small GEMMs) / MFMA peak + Stage A fwd+bwd bytes / HBM peak,
        plus Stage G (RelationEncoder, fwd + bwd = 3 x R * len * 3.45 MFLOP) / MFMA peak
    -- the DENSE-signature figures, before decoder and optimizer, as the survey states them (C2: ~20 ms).  Returns (ms, parts)."""
    n, B, d, ff, L = stats["n"], stats["B"], cfg["d"], cfg["ff"], cfg["layers"]
    P = n * n * B
    stage_r = 12.0 * P * d * d                                  # relation projection, fwd + bwd, per layer
    small = 3.0 * 2.0 * n * B * d * (3 * d + d + 2 * ff)        # QKV, output and feed-forward products, fwd + bwd, per layer
    stage_a = 3.0 * P * 2 * d * s_el + 10.0 * n * B * d * s_el  # relation attention, fwd + bwd bytes, per layer
    stage_g = 3.0 * stats["R"] * stats["mean_path_len"] * 3.45e6
    ms_r = L * (stage_r + small) / (MFMA_PEAK_TFS * 1e12) * 1e3
    ms_a = L * stage_a / (HBM_PEAK_GBS * 1e9) * 1e3
    ms_g = stage_g / (MFMA_PEAK_TFS * 1e12) * 1e3
    return ms_r + ms_a + ms_g, {"stage_R_and_small_gemms_ms": round(ms_r, 2), "stage_A_ms": round(ms_a, 2), "stage_G_ms": round(ms_g, 2)}


def build_roofline(a, cfg, stats, model, trainer, batch, prof, dev, cd, detail=True):
    """roofline object of the JSON line.  Headline (SURVEY 8d "Stage A"): the relation-attention forward kernel on the
    reference's DENSE relation signature (rarb[n,n,B,2d] = relation_in_proj(relation) materialised), measured live here with
    HIP events on the real batch -- layer-0 q/k/v of the real concept embeddings, the real projected bank rows gathered by the
    real type ids, the key-padding mask the model passes.  The training step itself runs the FACTORED operand; its launches
    inside the timed region are reported next to it against ITS algorithmic bytes (every bank row once), not Stage A's.
    `kernels`: one row per other dominant kernel from a detail pass (2 extra steps with per-launch events)."""
    from gtos_amd import ops
    n, B, d, H = stats["n"], stats["B"], cfg["d"], cfg["H"]
    s_el = 2 if cd == torch.bfloat16 else 4
    P, R = n * n * B, stats["R"]
    N_rows = int(batch["relation_length"].sum())
    hs = 256

    def span(pr, name):
        ev_ = pr.get(name, [])
        ms = [s_.elapsed_time(e_) for s_, e_, _u in ev_]
        return (sum(ms) / len(ms), len(ms), sum(u for _, _, u in ev_), sum(ms)) if ms else (None, 0, 0, 0.0)

    occ = torch.bincount(batch['relation'].reshape(-1), minlength=R)
    R_single = int((occ == 1).sum())              # types with ONE pair: their bank-gradient row is written by the query-major pass
    R_multi, P_multi = R - R_single, P - R_single
    stage_a = P * 2 * d * s_el + 4 * n * B * d * s_el + n * B
    fact_bytes = R * 2 * d * s_el + P * 4 + 4 * n * B * d * s_el + n * B
    pmc = {}
    import glob
    import re
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rel_attn_pmc.json")),
                       key=lambda f: int(re.match(r"r(\d+)", os.path.basename(f)).group(1)))
    pmc_file = pmc_files[-1] if pmc_files else None          # the newest round's counter profile of the attention kernels
    if a.config == "C2" and a.dtype == "bf16" and pmc_file:
        pmc = json.load(open(pmc_file))
    key = "rel_attn_fwd_mode1" if a.dense else "rel_attn_fwd_mode2"

    def in_step_of(pr, where):
        ms_in, n_in, _, _ = span(pr, key)
        if not ms_in:
            return None
        alg = stage_a if a.dense else fact_bytes
        return {"kernel": "rel_attn_fwd_kernel, %s operand (what the training steps launch)" % ("dense" if a.dense else "factored"),
                "measured_in": where,
                "launches": n_in, "avg_us": round(ms_in * 1e3, 1), "algorithmic_bytes": alg,
                "algorithmic_bytes_formula": "P*2d*s + 4nBd*s + nB" if a.dense else "R*2d*s (each bank row once) + P*4 (type ids) + 4nBd*s + nB",
                "achieved": round(alg / ms_in / 1e6, 1), "frac": round(alg / ms_in / 1e6 / HBM_PEAK_GBS, 4),
                "traffic": pmc.get("dense" if a.dense else "factored", {}).get("rel_attn_fwd_kernel", {}).get("traffic_bytes_per_launch")}
    overlapped = bool(ops.PROJ_SIDE) and not a.dense
    in_step = in_step_of(prof, "the timed region" + (": every launch runs BESIDE the next layers' relation-projection GEMMs on the auxiliary "
                                                     "stream (GTOS_PROJ_SIDE), so its duration is not the kernel's own" if overlapped else ""))

    # ---- dense-signature leg on the real batch (outside the timed region)
    model.eval()
    with torch.no_grad():
        x, mask = model._concepts(batch)
        bank = model.relation_encoder(batch['relation_bank'], batch['relation_length'], trie=batch.get('relation_trie'))
        layer = model.graph_encoder.layers[0].self_attn
        qkv = ops.linear(x, layer.in_proj_weight, layer.in_proj_bias)
        proj = ops.linear(bank, layer.relation_in_proj.weight)                          # [R, 2d]
        rarb = proj.index_select(0, batch['relation'].reshape(-1)).view(n, n, B, 2 * d)  # the dense operand, [j][i] order
        ops.PROFILE = {}
        for _ in range(8):
            ops.attention_core(qkv, None, (0, d, 2 * d), d, H, (d // H) ** -0.5, rel=rarb, key_pad=mask)
        torch.cuda.synchronize()
        ev = ops.PROFILE["rel_attn_fwd_mode1"][2:]
        ops.PROFILE = None
    dms = sum(s_.elapsed_time(e_) for s_, e_, _ in ev) / len(ev)
    del rarb, proj, qkv
    model.train()
    roof = {"bound": "hbm", "achieved": round(stage_a / dms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(stage_a / dms / 1e6 / HBM_PEAK_GBS, 4),
            "traffic": pmc.get("dense", {}).get("rel_attn_fwd_kernel", {}).get("traffic_bytes_per_launch"),
            "traffic_source": ("profiles/" + os.path.basename(pmc_file or "") + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes per operand "
                               "mode (tools/pmc_rel_attn.sh), gfx950 corrections of MI355X_MICROARCH.md") if pmc else None,
            "kernel": "rel_attn_fwd_kernel<bf16,8>, DENSE relation signature rarb[n,n,B,2d] (SURVEY 8d Stage A), real batch: layer-0 "
                      "q/k/v, real projected bank rows, key-padding mask",
            "launches": len(ev), "avg_us": round(dms * 1e3, 1), "algorithmic_bytes": stage_a,
            "measured": "HIP events on the launch stream, live in this run, after the timed region",
            "in_step": in_step}

    # ---- detail pass: per-launch events on the other dominant kernels (they would perturb the timed region)
    rows = []
    if detail and not a.dense and cd == torch.bfloat16:
        ops.PROFILE, ops.PROFILE_DETAIL, ops.GEMM_PROFILE = {}, True, {}
        side_was, ops.BWD_SIDE = ops.BWD_SIDE, False        # one kernel at a time: no auxiliary-stream GEMM beside the timed launches
        proj_was, ops.PROJ_SIDE = ops.PROJ_SIDE, False
        for _ in range(2):
            trainer.step(batch)
        torch.cuda.synchronize()
        dp, gp = ops.PROFILE, ops.GEMM_PROFILE
        ops.PROFILE, ops.PROFILE_DETAIL, ops.GEMM_PROFILE = None, False, None
        ops.BWD_SIDE, ops.PROJ_SIDE = side_was, proj_was
        if overlapped and roof["in_step"] is not None:
            # the figure above is the launch as the step runs it (beside the auxiliary stream's projection GEMMs); the detail pass runs the
            # kernel alone: what the overlap costs it is the difference
            alone = in_step_of(dp, "the detail pass (2 extra training steps, auxiliary-stream overlap off: the kernel alone)")
            if alone:
                roof["in_step"]["alone"] = {k: alone[k] for k in ("measured_in", "launches", "avg_us", "achieved", "frac")}

        def hbm_row(name, label, bytes_per_launch=None, bytes_per_unit=None, note=""):
            ms, cnt, units, tot = span(dp, name)
            if not ms:
                return
            byt = bytes_per_launch if bytes_per_launch is not None else bytes_per_unit * units / cnt
            rows.append({"kernel": label, "bound": "hbm", "launches_per_step": cnt // 2, "avg_us": round(ms * 1e3, 1),
                         "algorithmic_bytes_per_launch": int(byt), "achieved": round(byt / ms / 1e6, 1), "unit": "GB/s",
                         "peak": HBM_PEAK_GBS, "frac": round(byt / ms / 1e6 / HBM_PEAK_GBS, 4), "ms_per_step": round(tot / 2, 2),
                         "bytes": note})
        hbm_row("rel_attn_bwd_q+kv_mode2", "rel_attn_bwd_q_kernel + rel_attn_bwd_kv_kernel (factored, one C call)",
                bytes_per_launch=2 * R * 2 * d * s_el + R_single * 2 * d * s_el + 10 * n * B * d * s_el + 4 * P * H * 4 + 2 * P * 4,
                note="2 passes x R*2d*s bank rows + R_single*2d*s bank-gradient rows of the %d single-pair types written + 10nBd*s (q,k,v,o,do "
                     "read; dq,dk,dv written) + pd,gs [P,H] fp32 written+read + ids" % R_single)
        hbm_row("rel_attn_bwd_bank", "rel_attn_bwd_bank_kernel (types with >= 2 pairs)",
                bytes_per_launch=2 * R_multi * 2 * d * s_el + P_multi * H * 4 + P_multi * 4 + 2 * n * B * d * s_el,
                note="%d multi-pair types: 2d*s bank row read + gradient row written each; their %d pairs: gs [H] fp32 + pair id; q,k rows "
                     "once (they are served by L2)" % (R_multi, P_multi))
        # ---- RelationEncoder, by the evaluation this run used (the spans are named per mode and per layer)
        # the reference's dropout semantics (library default): one GRU row per (path, position) in both layers (gtos_amd.gru.PackedPathGRUFn);
        # these spans carry their launches' algorithmic bytes as units (gtos_amd/gru.py: _step_fwd / _step_bwd_fused)
        dp_ = (-100) % 64 + 100
        hbm_row("gru_step_fwd_packed_l0", "gru_step_fwd_a2w3_kernel (launches under 8192 rows: the single-stage gru_step_fwd_kernel<1>), GRU layer 0 (input product fused: label rows x W_ih inside)", bytes_per_unit=1,
                note="per active row: label-embedding row %d*2 B + entering state h read; gates 4h + new state h + dropped copy h written (h = %d bf16)" % (dp_, hs))
        hbm_row("gru_step_fwd_packed_l1", "gru_step_fwd_a2w3_kernel (launches under 8192 rows: the single-stage gru_step_fwd_kernel<1>), GRU layer 1 (input product fused: layer-0 output rows x W_ih inside)", bytes_per_unit=1,
                note="per active row: layer-0 output row 2h + entering state h read; gates 4h + new state h written")
        hbm_row("gru_step_bwd_packed_l1", "gru_step_bwd_kernel, GRU layer 1: cell tiles + the previous step's input-gradient tiles in one launch",
                bytes_per_unit=1,
                note="per active row: gates 4h + entering state h read, state gradient h read + written, d4 4h written; the previous step's d4 rows "
                     "once (4h where both roles read them); that step's input gradient 2h written (direction 1: read + written)")
        hbm_row("gru_step_bwd_packed_l0", "gru_step_bwd_kernel, GRU layer 0: cell tiles + the previous step's (masked) label-row gradient tiles",
                bytes_per_unit=1,
                note="as layer 1 plus the output gradient row h read per active row; input gradient %d columns" % dp_)
        for l_ in (1, 0):
            ms, cnt, units, tot = span(dp, "gru_dw_grouped_l%d" % l_)
            if ms:
                rows.append({"kernel": "gemm256p_tn_kernel<grouped>, GRU layer %d: dW_ih and dW_hh of a direction as one product d4^T [x | h_prev]" % l_,
                             "bound": "mfma", "launches_per_step": cnt // 2, "avg_us": round(ms * 1e3, 1), "flops_per_launch": int(units / cnt),
                             "achieved": round(units / tot / 1e9, 1), "unit": "TFLOP/s", "peak": MFMA_PEAK_TFS,
                             "frac": round(units / tot / 1e9 / MFMA_PEAK_TFS, 4), "ms_per_step": round(tot / 2, 2),
                             "note": "useful flops 2 N 3h (in + h); runs on the auxiliary stream beside the other direction's HBM-bound steps"})
        # trie-shared masks (--relation-masks node, opt-in): layer 0 per trie node, layer 1 on per-node gate tables
        hbm_row("gru_step_fwd_tables", "gru_l1_fwd_persistent_kernel (trie evaluation: GRU layer 1 step, gate tables gathered, W_hh slice resident in LDS)",
                bytes_per_unit=(4 + 1 + 1 + 6) * hs * 2 + 8,
                note="per active row: gates 4h + new state h written, state h read, two 3h table rows gathered, 2 node ids")
        hbm_row("gru_step_fwd_x", "gru_step_fwd_kernel<1> (trie evaluation: GRU layer 0 on the tries, input product fused)",
                bytes_per_unit=(4 + 1 + 1 + 1) * hs * 2 + 104 * 2 + 4,
                note="per trie node: gates 4h + state h + dropped copy h written, parent state h gathered, embedding row read")
        hbm_row("gru_step_bwd_rows", "gru_step_bwd_kernel (trie evaluation: GRU layer 1)", bytes_per_unit=(4 + 1 + 3 + 2 + 4) * hs * 2,
                note="per active row: gates 4h + h_prev h read, later step's d(hg) 3h read (MFMA operand), dh read+written, d4 4h written")
        hbm_row("segment_sum_rows", "seg_sum_stream_kernel (trie evaluation: gate-table gradients, rows -> trie nodes)",
                bytes_per_unit=3 * hs * 2 + 4, note="per row: d(xg) 3h read + row id; + nodes x 3h written (not counted)")
        grows = []
        for key_, evs in gp.items():
            lay, nn, kk, dts, sk, big = key_
            ms = sum(s_.elapsed_time(e_) for _, s_, e_ in evs)
            long_sum = sum(m_ for m_, _, _ in evs)
            grows.append((ms, lay, nn, kk, dts, sk, len(evs), long_sum))
        for ms, lay, nn, kk, dts, sk, calls, long_sum in sorted(grows, reverse=True)[:4]:
            fl = 2.0 * long_sum * nn * kk
            rows.append({"kernel": "%s %s N=%d %s=%d %s splitk=%d" % (_gemm_kernel_name(lay, nn, kk, sk, long_sum // calls), lay, nn,
                                                                   "M" if lay[0] == "T" else "K", kk, dts, sk),
                         "bound": "mfma", "launches_per_step": calls // 2, "avg_us": round(ms * 1e3 / calls, 1),
                         "flops_per_launch": int(fl / calls), "achieved": round(fl / ms / 1e9, 1), "unit": "TFLOP/s",
                         "peak": MFMA_PEAK_TFS, "frac": round(fl / ms / 1e9 / MFMA_PEAK_TFS, 4), "ms_per_step": round(ms / 2, 2)})
    roof["kernels"] = rows
    roof["kernels_note"] = ("detail pass: 2 extra training steps with a HIP-event pair around every launch of these kernels, the "
                            "auxiliary-stream overlap of the projection-gradient GEMMs switched off so that every launch is timed alone "
                            "(events and serialisation slow the step, so none of this happens inside the timed region); bytes are "
                            "algorithmic, per launch; GEMM rows: the four shapes with the largest time share")
    return roof


def _gemm_kernel_name(lay, nn, kk, sk, long_dim):
    """Which kernel gtos_gemm's dispatcher (csrc/gemm.hip) picks for a bf16 product of this shape; for NT `long_dim` is M
    and kk is K, for TN kk is M and `long_dim` is K."""
    if lay == "NT":
        tiles256 = -(-long_dim // 256) * -(-nn // 256)
        if tiles256 >= 512 and nn >= 256 and kk >= 1024 and kk % 32 == 0:
            return "gemm256q_nt_kernel"
        if tiles256 >= 1024 and nn >= 256 and kk >= 2048:
            return "gemm256_nt_kernel"
    if lay == "TN" and sk > 1 and nn >= 256 and kk >= 256 and long_dim // sk >= 256:
        return "gemm256p_tn_kernel"
    return "gemm_kernel"


def decode_bench(a):
    """Inference benchmark: beam search (generator/work.py flow) on synthetic graphs of the named config, eval-mode batch
    (all shortest paths, K alternatives per pair), random-weight model.  Prints one JSON line: sentences/s, ms per decoder
    step; cpu_baseline = the pinned oracle (full-prefix recompute, no state carried) on --cpu-graphs of the graphs."""
    from gtos_amd import synth
    from gtos_amd.config import generator_args
    from gtos_amd.generator import Generator
    dev = torch.device("cuda:0")
    cfg = synth.CONFIGS[a.config]
    vocabs = synth.synth_vocabs()
    cd = torch.bfloat16 if a.dtype == "bf16" else torch.float32

    def local_vocabs(batch):        # cp_seq ids beyond the predictable vocabulary are per-graph copy ids: give them strings
        pv, cp = vocabs['predictable_token'], batch['cp_seq']
        return [{int(i): "copy%d" % int(i) for i in cp[:, b].tolist() if i >= pv.size} for b in range(cp.shape[1])]

    torch.manual_seed(19940117)
    model = Generator(vocabs, device=dev, depth_size=256 if cfg["kind"] == "dep" else 32, **generator_args(cfg)).to(dev)
    model.set_compute_dtype(cd)
    model.eval()
    batch, stats = synth.make_config_batch(a.config, train=False)
    B = stats["B"]
    from gtos_amd.pathtrie import attach_path_trie
    batch_dev = {k: v.to(dev) for k, v in attach_path_trie(batch).items()}
    batch_dev['local_idx2token'] = local_vocabs(batch)
    model.work(batch_dev, a.beam, 3)                       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        model.encode_step(batch_dev, train=False)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    beams = model.work(batch_dev, a.beam, a.max_steps)
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    steps = max(b.steps for b in beams)
    out = {"metric": "sentences/sec beam search (100-node AMR, batch 64, beam %d, %d steps)" % (a.beam, a.max_steps),
           "value": B / dt_, "unit": "sentences/s", "n_gpus": 1, "higher_is_better": True, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": "%s eval batch: %d graphs, n=%d, beam %d, max %d decoder steps, random weights" % (
               a.config, B, stats["n"], a.beam, a.max_steps)},
           "seconds": dt_, "encode_seconds": t_enc, "decoder_steps": steps,
           "ms_per_decoder_step": 1e3 * (dt_ - t_enc) / max(1, steps)}
    if not a.no_cpu_baseline:
        from oracle import gtos_oracle as O
        cores = host_cores()
        torch.set_num_threads(cores)
        torch.manual_seed(19940117)
        ref = O.Generator(vocabs, depth_size=256 if cfg["kind"] == "dep" else 32, **generator_args(cfg))
        ref.eval()
        small, _ = synth.make_config_batch(a.config, train=False, B=a.cpu_graphs)
        small['local_idx2token'] = local_vocabs(small)
        t0 = time.perf_counter()
        O.generator_work(ref, small, vocabs, a.beam, a.max_steps)
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": a.cpu_graphs / ct, "unit": "sentences/s", "cores": cores, "kind": "port",
                               "sample": "%d graphs of %s, fp32, beam %d, %d steps, %.1f s" % (
                                   a.cpu_graphs, a.config, a.beam, a.max_steps, ct)}
    print(json.dumps(out))


def _gather_ints(value, world, dev):
    """every rank's integer, on every rank (one all-reduce of a rank-indexed vector)"""
    t = torch.zeros(world, device=dev, dtype=torch.float64)
    t[int(os.environ.get("RANK", "0"))] = float(value)
    dist.all_reduce(t)
    return t.tolist()


def make_feed(a, cfg, rank, B_rank, dev, asm_times, free_b=0):
    """Loader in the loop (generator/train.py:136-140 builds every batch inside the step loop; translator/train.py:85-89,144-146 feeds
    them through a producer process and a queue): a pool of synthetic items of the config -> AMRLoader / DependencyLoader (reference
    batching policy, paths re-drawn per batch) -> Prefetcher worker process(es) -> upload on a copy stream.  Route: everything of the
    relation section on the device (default, ONE worker) or on the host (--host-relations).  Returns (feed, loader_info)."""
    import random
    from gtos_amd import data as data_mod
    from gtos_amd import synth
    device_all = a.device_relations or not (a.host_relations or a.device_tries)
    prep_in_worker = a.prep_in_worker or (device_all and not a.device_relations)        # the default route: preparation on the upload thread
    workers = a.workers or (1 if device_all else 4)
    depth = a.depth or 2 * workers
    pool_n = a.pool or 4 * B_rank
    pool_n -= pool_n % B_rank
    prep = "device_all" if device_all else ("device" if a.device_tries else True)
    if cfg["kind"] == "amr":
        vocabs_s = synth.synth_vocabs()
        items, graphs = synth.make_amr_items(a.config, pool_n, first_graph=rank * pool_n, vocabs=vocabs_s)
        unit = data_mod.AMRLoader.size_of(items[0])                       # every item of a config has the same size
        loader = data_mod.AMRLoader(vocabs_s, items, batch_size=B_rank * unit - unit // 2, for_train=True,
                                    rng=random.Random(19940117 + rank), n_threads=a.relbatch_threads, graphs=graphs, index_prep=prep)
    else:
        vocabs_s = synth.dep_vocabs()
        trees = synth.make_dep_trees(a.config, pool_n, first_graph=rank * pool_n, vocabs=vocabs_s)
        unit = data_mod.DependencyLoader.size_of(trees[0])
        loader = data_mod.DependencyLoader(vocabs_s, trees, batch_size=B_rank * unit - unit // 2, for_train=True,
                                           rng=random.Random(19940117 + rank), n_threads=a.relbatch_threads, index_prep=prep)

    def timed_run(job):                                               # runs in the worker: reports its own assembly time
        t_ = time.perf_counter()
        out_ = loader.run_job(job)
        out_["_assembly_s"] = time.perf_counter() - t_
        return out_

    def jobs():
        while True:                                                   # epochs over the pool: reshuffled, paths re-drawn
            yield from loader.jobs()
    tries = a.device_tries or ("hip" if device_all else False)
    if a.loader == "processes":
        feed = data_mod.Prefetcher(jobs(), depth=depth, workers=workers, device=dev, processes=True, runner=timed_run,
                                   device_tries=tries, prep_in_worker=prep_in_worker)
    else:
        feed = data_mod.Prefetcher((lambda j=j: timed_run(j) for j in jobs()), depth=depth, workers=workers, device=dev,
                                   device_tries=tries, prep_in_worker=prep_in_worker)
    info = {"workers": workers, "kind": a.loader, "depth": depth, "relbatch_threads": a.relbatch_threads,
            "loader_class": type(loader).__name__,
            "tries": {"torch": "device (torch ops on the copy stream)", "hip": "device (staged HIP builder on the copy stream)",
                      False: "host (worker)"}[tries],
            "relations": "device (staged HIP builders: relation / bank / index)" if device_all else "host (worker)",
            "device_prep_thread": "upload thread" if prep_in_worker else "training thread",
            "pool_graphs_per_rank": pool_n, "device_memory_free_gb_at_start": round(free_b / 2 ** 30, 1),
            "allocator": "default" if os.environ.get("GTOS_BENCH_NO_ROUNDUP") else "roundup_power2_divisions:16"}
    return feed, info


def main():
    a = parse()
    if a.cpu_protocol:
        a.cpu_graphs, a.cpu_warmup, a.cpu_steps, a.cpu_budget = 8, 2, 5, 600.0
    if a.decode:
        torch.cuda.set_device(0)
        return decode_bench(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher, one rank per GPU (the reference spawns its own ranks too,
        # generator/train.py:173-190)
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (a.gpus, world))
    if a.dry_launch:
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.destroy_process_group()
        print("dry-launch rank %d of %d ok (host threads per rank %d of %d usable cores)" % (rank, world, rank_cores(world), host_cores()), flush=True)
        return
    if os.environ.get("GTOS_ONE_DEVICE"):      # functional check of the N>1 path on a 1-GPU box: every rank on cuda:0, gloo
        local = 0
    # N ranks share the host: each keeps to its share of the usable cores (torch's intra-op pool defaults to ALL of them per process;
    # 8 ranks x (launch thread + loader worker + upload thread) on a 16-core quota would otherwise fight over them)
    cores_rank = rank_cores(world)
    torch.set_num_threads(cores_rank)
    a.relbatch_threads = max(1, min(a.relbatch_threads, cores_rank - 1))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if (a.fresh_batches or not a.no_loader_leg) and not os.environ.get("GTOS_BENCH_NO_ROUNDUP"):
        # every batch has its own bank size (R and the packed row count vary by a percent), so every large buffer of the step
        # changes size from step to step; the caching allocator then keeps growing (a cached 890 MB block cannot serve an 895 MB
        # request) until hipMalloc fails and the cache is flushed -- measured: 63 ms per step for one run, 120-140 ms for the next
        # two on the same box.  Rounding request sizes up to 1/16 of a power of two makes consecutive steps reuse their blocks.
        torch.cuda.memory._set_allocator_settings("roundup_power2_divisions:16")
    # the driver releases a finished process's device memory lazily (a 100 GB process still shows 35-40 % of the VRAM allocated a
    # second after it has gone, none three seconds later): a run started back to back with another one would otherwise begin under
    # memory pressure that is not its own
    for _ in range(24):
        free_b, total_b = torch.cuda.mem_get_info(dev)
        if free_b > 0.9 * total_b:
            break
        time.sleep(0.5)
    log("device memory free %.1f of %.1f GB" % (free_b / 2 ** 30, total_b / 2 ** 30))
    if world > 1:
        dist.init_process_group(os.environ.get("GTOS_DIST_BACKEND", "nccl"), rank=rank, world_size=world)   # nccl = RCCL on ROCm

    from gtos_amd import ops, synth
    from gtos_amd import gru as gru_mod
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.train import Trainer

    cfg = synth.CONFIGS[a.config]
    cd = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    log("building model")
    model = build_generator(Generator, a.config, dev, factored_relation=not a.dense).to(dev)
    model.set_compute_dtype(cd)
    model.train()
    from gtos_amd.encoder import set_relation_mask_sharing
    set_relation_mask_sharing(model, a.relation_masks)
    trainer = Trainer(model, cfg["d"], warmup_steps=2000, compute_dtype=cd, world_size=world, rank=rank)
    B_cfg = cfg["B"]
    if a.scaling == "strong":
        if B_cfg % world:
            raise SystemExit("--scaling strong needs the batch (%d graphs) to divide by --gpus %d" % (B_cfg, world))
        B_rank = B_cfg // world            # the reference generator's policy: batch budget / world_size (generator/train.py:183)
    else:
        B_rank = B_cfg
    log("model on device; generating batch")
    from gtos_amd.pathtrie import attach_path_trie
    from gtos_amd.relindex import attach_relation_index
    loader_info, feed = None, None
    asm_times = []
    if a.fresh_batches:
        feed, loader_info = make_feed(a, cfg, rank, B_rank, dev, asm_times, free_b)
        batch = next(feed)
        stats = {"n": int(batch["concept"].shape[0]), "B": int(batch["concept"].shape[1]), "T": int(batch["token_in"].shape[0]),
                 "R": int(batch["relation_bank"].shape[1]),
                 "mean_path_len": float(batch["relation_length"].float().mean())}
        assert stats["B"] == B_rank, stats
        asm_times.append(batch.pop("_assembly_s"))
    else:
        batch, stats = synth.make_config_batch(a.config, rank=rank, B=B_rank)   # rank r holds graphs [r*B_rank, (r+1)*B_rank)
        attach_relation_index(attach_path_trie(batch))   # host-side index preparation: batch assembly, like the relation bank itself
        batch = {k: v.to(dev) for k, v in batch.items()}
    log("batch", stats)
    ops.set_seed(19940117 + rank)
    wait_s = [0.0]

    def next_batch():
        if feed is None:
            return batch
        t_ = time.perf_counter()
        b_ = next(feed)
        wait_s[0] += time.perf_counter() - t_
        asm_times.append(b_.pop("_assembly_s"))
        return b_

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    prewarm_steps = 0
    if a.prewarm_seconds > 0:
        t_begin, prev = time.perf_counter(), None
        while True:
            t_w = time.perf_counter()
            for _ in range(5):
                trainer.step(next_batch(), sync=False)
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t_w, time.perf_counter() - t_begin], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)       # every rank takes the same decision (the steps hold collectives)
            w_s, el_s = tt.tolist()
            if prewarm_steps == 0 and not os.environ.get("GTOS_BENCH_KEEP_COLD_CACHE"):
                # the first steps of a process allocate in a different order than the steady state (lazy workspaces, first-use caches,
                # autotuned splits) and leave the caching allocator holding blocks the steady state never reuses: hand them back once
                # (C5: 274 GB reserved for 94 GB allocated otherwise)
                torch.cuda.empty_cache()
            prewarm_steps += 5
            log("prewarm window %.1f ms/step" % (200.0 * w_s))
            settled = prev is not None and abs(w_s - prev) <= 0.015 * prev
            if (settled and el_s >= min(a.prewarm_min_seconds, a.prewarm_seconds)) or el_s > a.prewarm_seconds:
                break
            prev = w_s
    for i in range(a.warmup):
        v = trainer.step(next_batch())
        log("warmup step", i, "loss", v)
    sync()
    log("warmup done")
    ops.PROFILE = {}
    trainer.comm_exposed_s = 0.0
    trainer.comm_exposed_ms()
    wait_s[0] = 0.0
    n_asm0 = len(asm_times) if feed is not None else 0
    torch.cuda.reset_peak_memory_stats(dev)
    retries0 = torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)
    t0 = time.perf_counter()
    pend = []
    for _ in range(a.steps):
        # the abnormal-loss rule and the lr schedule run on the device (gtos_step_control): no host read of the loss inside
        # the step; the values are read after the timed region
        pend.append(trainer.step(next_batch(), sync=False))
    sync()
    elapsed = time.perf_counter() - t0
    losses = [p_.value() for p_ in pend]
    ms_ = torch.cuda.memory_stats(dev)
    # allocator view of the timed region: a retry = hipMalloc failed, the cache was flushed (device-wide sync) and the request repeated
    memory_info = {"peak_allocated_gb": round(ms_.get("allocated_bytes.all.peak", 0) / 2 ** 30, 1),
                   "peak_reserved_gb": round(ms_.get("reserved_bytes.all.peak", 0) / 2 ** 30, 1),
                   "alloc_retries_in_timed_region": int(ms_.get("num_alloc_retries", 0) - retries0)}
    my_elapsed = elapsed
    if feed is not None:
        done = asm_times[n_asm0:] or asm_times
        loader_info.update({"host_assembly_s_per_batch": round(sum(done) / max(1, len(done)), 4),
                            "consumer_wait_ms_per_step": round(1e3 * wait_s[0] / a.steps, 3),
                            # of which (whole run, per batch): waiting for a finished batch / host side of the device-side preparation
                            "queue_wait_ms_per_batch": round(1e3 * feed.stats["queue_wait_s"] / max(1, feed.stats["batches"]), 3),
                            "device_prep_ms_per_batch": round(1e3 * feed.stats["device_prep_s"] / max(1, feed.stats["batches"]), 3)})
        feed.close()                                   # the roofline legs below reuse the first batch the feed handed out
        feed = None
    comm_exposed = max(trainer.comm_exposed_s, 1e-3 * trainer.comm_exposed_ms())   # host wait (gloo) / compute-stream stall (RCCL)
    log("timed region done", elapsed)
    prof, ops.PROFILE = ops.PROFILE, None
    # ---- N > 1: the OTHER scaling mode in the same run, so that one command gives both (weak: B graphs per GPU, the translator's
    # policy; strong: B graphs in total, the generator's, generator/train.py:183), and what the collective layer really saw
    other_scaling, rccl_info = None, None
    if world > 1:
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        rccl_info = {"backend": dist.get_backend(), "ranks_seen_by_allreduce": int(seen.item()), "world_size": world,
                     "devices": sorted(set(int(x) for x in _gather_ints(local, world, dev)))}
        try:
            rccl_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                     # noqa: BLE001
            pass
        if not a.fresh_batches and B_cfg % world == 0:
            B_other = B_cfg // world if a.scaling == "weak" else B_cfg
            b2, st2 = synth.make_config_batch(a.config, rank=rank, B=B_other)
            attach_relation_index(attach_path_trie(b2))
            b2 = {k: v.to(dev) for k, v in b2.items()}
            for _ in range(3):
                trainer.step(b2, sync=False)
            sync()
            t_o = time.perf_counter()
            for _ in range(a.steps):
                trainer.step(b2, sync=False)
            sync()
            to = torch.tensor([time.perf_counter() - t_o], device=dev, dtype=torch.float64)
            dist.all_reduce(to, op=dist.ReduceOp.MAX)
            dt_o = float(to.item())
            other_scaling = {"scaling": "strong" if a.scaling == "weak" else "weak", "B_per_gpu": B_other, "global_batch": world * B_other,
                             "ms_per_step": round(1e3 * dt_o / a.steps, 3), "value": round(world * B_other * a.steps / dt_o, 1),
                             "unit": "graphs/s", "steps": a.steps}
            del b2
    # ---- the OTHER dropout-mask mode of the RelationEncoder, same batch, same K steps, right after the timed region (every rank)
    masks_leg = None
    # (N > 1 runs keep to the legs that cannot strand a rank inside a collective: the other scaling mode above; the mask-mode and loader
    #  legs run at N = 1 -- or everywhere with GTOS_BENCH_ALL_LEGS=1)
    all_legs = world == 1 or bool(os.environ.get("GTOS_BENCH_ALL_LEGS"))
    if not a.no_masks_leg and cd == torch.bfloat16 and not a.fresh_batches and all_legs:
        other = "path" if a.relation_masks == "node" else "node"
        set_relation_mask_sharing(model, other)
        # the other mode is another set of buffer sizes: the cache is full of blocks cut for the headline's (with three warm-up steps on
        # top of it the trie mode measured 73.5 ms per step after a per-path headline, 61.7 on its own)
        torch.cuda.empty_cache()
        for _ in range(8):
            trainer.step(batch, sync=False)
        sync()
        t_m = time.perf_counter()
        for _ in range(a.steps):
            trainer.step(batch, sync=False)
        sync()
        tm = torch.tensor([time.perf_counter() - t_m], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt_m = float(tm.item())
        set_relation_mask_sharing(model, a.relation_masks)
        masks_leg = {"relation_masks": other, "ms_per_step": round(1e3 * dt_m / a.steps, 3), "steps": a.steps,
                     "value": round(world * stats["B"] * a.steps / dt_m, 1), "unit": "graphs/s",
                     "vs_headline": round(dt_m / elapsed, 4),
                     "note": ("masks per (path, position): the reference's dropout (generator/encoder.py:91-92,105), the library default; "
                              "one GRU row per path and position in both layers" if other == "path" else
                              "masks per trie node (opt-in): layer 0 once per trie node, layer-1 input gates from per-node tables")}
    # ---- loader in the loop, AFTER the timed region of the default run (every rank: the steps hold collectives): the same K steps,
    # each on a NEW batch from the default loader route (relation section on the device, ONE worker process per GPU).  The headline
    # above is the pre-built batch SURVEY 8d prescribes; this leg says what feeding costs on top of it.
    loader_leg = None
    # (round 5: this leg runs at every N -- one loader worker process per rank, N of them on the host -- the other secondary legs at N = 1)
    if not a.fresh_batches and not a.no_loader_leg and cd == torch.bfloat16 and not a.dense:
        try:
            # the cache is full of blocks cut for ONE batch's sizes; the loader's batches all differ by a percent.  At C2 that costs
            # nothing, at C5 (95 GB allocated, 230+ GB cached) the first varying-size steps overflow the device and every step pays a
            # cache flush (938 ms per step measured, against 225 ms for `--fresh-batches` from a clean start): start the leg clean
            torch.cuda.empty_cache()
            asm2 = []
            feed2, info2 = make_feed(a, cfg, rank, B_rank, dev, asm2, free_b)
            for _ in range(3):
                b_ = next(feed2); b_.pop("_assembly_s", None)
                trainer.step(b_, sync=False)
            sync()
            wait2, t_l = 0.0, time.perf_counter()
            for _ in range(a.steps):
                t_ = time.perf_counter()
                b_ = next(feed2)
                wait2 += time.perf_counter() - t_
                asm2.append(b_.pop("_assembly_s"))
                trainer.step(b_, sync=False)
            sync()
            dt_l = time.perf_counter() - t_l
            tl = torch.tensor([dt_l], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            dt_l = float(tl.item())
            loader_leg = {"ms_per_step": round(1e3 * dt_l / a.steps, 3), "steps": a.steps, "value": round(world * B_rank * a.steps / dt_l, 1),
                          "unit": "graphs/s", "vs_prebuilt_batch": round(dt_l / elapsed, 4),
                          "workers_per_gpu": info2["workers"], "worker_kind": info2["kind"], "loader_class": info2["loader_class"],
                          "relations": info2["relations"], "tries": info2["tries"], "device_prep_thread": info2["device_prep_thread"],
                          "host_assembly_s_per_batch": round(sum(asm2) / max(1, len(asm2)), 4),
                          "consumer_wait_ms_per_step": round(1e3 * wait2 / a.steps, 3),
                          "device_prep_ms_per_batch": round(1e3 * feed2.stats["device_prep_s"] / max(1, feed2.stats["batches"]), 3),
                          "note": "every step a new batch (new graphs order, paths re-drawn, new bank: R varies by a percent), like "
                                  "generator/train.py:136-140; same rank count, same steps, right after the timed region"}
            feed2.close()
            del b_
        except Exception as e:                  # noqa: BLE001  (the leg must never cost the headline line)
            loader_leg = {"error": "%s: %s" % (type(e).__name__, e)}
            if world > 1:
                raise
    # ---- the same step replayed from a hipGraph (train.GraphedStep): what the Python-paced launch sequence costs.  One process, the
    # pre-built batch, the trie evaluation of the RelationEncoder (node masks or no dropout); the capture is not part of the figure.
    graph_leg = None
    if world == 1 and not a.fresh_batches and a.graph_leg and cd == torch.bfloat16 and not a.dense:
        from gtos_amd.train import GraphedStep
        gs = None
        try:
            ops.PROFILE = None
            t_c = time.perf_counter()
            gs = GraphedStep(trainer, batch)
            capture_s = time.perf_counter() - t_c
            for _ in range(3):
                gs()
            torch.cuda.synchronize()
            t_g = time.perf_counter()
            gp = [gs() for _ in range(a.steps)]
            torch.cuda.synchronize()
            dt_g = time.perf_counter() - t_g
            gl = [p_.value() for p_ in gp]
            graph_leg = {"ms_per_step": round(1e3 * dt_g / a.steps, 3), "steps": a.steps, "value": round(stats["B"] * a.steps / dt_g, 1),
                         "unit": "graphs/s", "vs_eager_launches": round(dt_g / elapsed, 4), "capture_s": round(capture_s, 2),
                         "losses_distinct": len(set(round(x, 6) for x in gl if x is not None)),
                         "note": "one torch.cuda.CUDAGraph replay per step on the pre-built batch; dropout masks new in every replay "
                                 "(device-side seed epoch, gtos_set_seed_epoch); parameters keep training across the replays"}
        except Exception as e:                  # noqa: BLE001  (the leg must never cost the headline line)
            graph_leg = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        finally:
            if gs is not None:
                gs.close()
            else:
                try:
                    ops.set_seed_epoch(None)
                except Exception:               # noqa: BLE001
                    pass
    # per-rank view: wall time of the timed region and the compute-stream stall on gradient collectives, gathered on rank 0
    table = torch.zeros((world, 2), device=dev, dtype=torch.float64)
    table[rank, 0], table[rank, 1] = my_elapsed, comm_exposed
    if world > 1:
        dist.all_reduce(table)                         # every rank fills its own row (a gather that gloo does on device tensors too)
    per_rank = table.tolist()
    elapsed = max(r[0] for r in per_rank)              # MAX over ranks: the job is as slow as its slowest rank

    n, B, d, H = stats["n"], stats["B"], cfg["d"], cfg["H"]
    s_el = 2 if cd == torch.bfloat16 else 4
    P, R = n * n * B, stats["R"]

    def span(pr, name):
        ev_ = pr.get(name, [])
        ms = [s_.elapsed_time(e_) for s_, e_, _u in ev_]
        return (sum(ms) / len(ms), len(ms), sum(u for _, _, u in ev_)) if ms else (None, 0, 0)

    components = {"relation_encoder_fwd_ms": span(prof, "relation_encoder_fwd")[0], "relation_gru_bwd_ms": span(prof, "relation_gru_bwd")[0],
                  "graph_encoder_fwd_ms": span(prof, "graph_encoder_fwd")[0],
                  "note": "HIP-event spans on the main stream inside the timed region, per step"}
    roofline = None
    if rank == 0:
        # N > 1: no detail pass (its extra training steps would issue collectives the other ranks do not join)
        roofline = build_roofline(a, cfg, stats, model, trainer, batch, prof, dev, cd,
                                  detail=(world == 1 and not os.environ.get("GTOS_BENCH_NO_DETAIL")))
    if rank == 0 and roofline is not None:
        ceil_ms, parts = step_ceiling(cfg, stats, s_el)
        roofline["step_ceiling_ms"] = round(ceil_ms, 2)
        roofline["step_frac"] = round(ceil_ms / (1e3 * elapsed / a.steps), 4)
        roofline["step_ceiling_parts"] = parts
        roofline["step_ceiling_note"] = ("SURVEY 8d's per-step roofline for this batch (n=%d, B=%d, R=%d, mean path length %.2f): graph-encoder layers at the "
                                         "MFMA peak (Stage R + small GEMMs) and the HBM peak (Stage A fwd+bwd, dense signature) plus Stage G at the MFMA "
                                         "peak; decoder, optimizer and host not included; step_frac = step_ceiling_ms / ms_per_step" % (
                                             n, B, R, stats["mean_path_len"]))
    if rank == 0:
        metric = ("graphs/sec training step (100-node AMR, batch 64)" if a.config in ("C2", "C4") else
                  "graphs/sec training step (%s: %d-node %s graphs, batch %d)" % (a.config, cfg["N"], cfg["kind"], B))
        out = {"metric": metric, "value": world * B * a.steps / elapsed,
               "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": a.scaling,
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "%s: %s %dx%d-node synthetic %s %s, %d-layer d=%d %d-head, "
                                      "full train step (fwd+bwd+allreduce+clip+Adam), dropout 0.2%s" % (
                                          a.config, "generator/" if cfg["kind"] == "amr" else "translator/", B, cfg["N"],
                                          "AMR graphs" if cfg["kind"] == "amr" else "dependency trees", "per GPU" if a.scaling == "weak" else
                                          "per GPU = %d in total over %d GPUs (generator/train.py:183)" % (world * B, world),
                                          cfg["layers"], d, H,
                                          ", a NEW loader-built batch every step" if a.fresh_batches else ", one pre-built device batch"),
                          "n": n, "B_per_gpu": B, "global_batch": world * B, "P": P, "R": R,
                          "mean_path_len": round(stats["mean_path_len"], 2), "T": stats["T"],
                          "relation_operand": "dense" if a.dense else "factored", "parallelism": "dp%d" % world,
                          "relation_gru": ("trie evaluation, dropout masks drawn per trie node and shared by the paths through it (OPT-IN, "
                                           "--relation-masks node; the library default draws them per (path, position) like the reference: "
                                           "see reference_masks)" if (gru_mod.TRIE and a.relation_masks == "node") else
                                           "the reference's dropout semantics (the library default): masks per (path, position), one GRU row "
                                           "each, on the packed-path kernels (gtos_amd.gru.PackedPathGRUFn: no host read, input gradients inside "
                                           "the backward step launches, grouped weight gradients); the trie-shared masks of rounds 2-3 (an opt-in, "
                                           "a different regulariser) are measured in the same run: see node_masks"),
                          "allreduce_exposed_ms_per_step": round(1e3 * max(r[1] for r in per_rank) / a.steps, 3),
                          "per_rank_ms_per_step": [round(1e3 * r[0] / a.steps, 3) for r in per_rank],
                          "per_rank_allreduce_exposed_ms_per_step": [round(1e3 * r[1] / a.steps, 3) for r in per_rank],
                          "rank_spread_ms_per_step": round(1e3 * (max(r[0] for r in per_rank) - min(r[0] for r in per_rank)) / a.steps, 3),
                          "loader": loader_info,
                          "prewarm_steps": prewarm_steps, "device_memory": memory_info,
                          "loss_first": losses[0], "loss_last": losses[-1]},
               "roofline": roofline, "components": components, "loader_in_loop": loader_leg, "hipgraph_replay": graph_leg,
               ("reference_masks" if a.relation_masks == "node" else "node_masks"): masks_leg,
               "other_scaling": other_scaling, "collectives": rccl_info}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config, a.cpu_graphs, a.cpu_steps, a.cpu_budget, a.cpu_warmup)
            out["cpu_baseline"]["leg"] = (
                "SURVEY 8d protocol, live in this run (8 graphs, 2 warm-up + 5 timed steps)" if a.cpu_graphs >= 8 and a.cpu_steps >= 5 else
                "BOUNDED leg (%d graphs, %d warm-up + <= %d timed steps, <= %.0f s): the default run must finish within minutes and SURVEY 8d's "
                "protocol (8 graphs, 2 + 5 steps) takes about five on this host; `python bench.py --cpu-protocol` runs it live, the newest "
                "committed record of that run is quoted as protocol_8d" % (a.cpu_graphs, a.cpu_warmup, a.cpu_steps, a.cpu_budget))
            # SURVEY 8d's protocol (8 graphs, 2 warm-up + >= 5 timed steps: five minutes of host time) is a run of its own
            # (--cpu-graphs 8 --cpu-steps 5 --cpu-warmup 2 --cpu-budget 600); the newest committed record of it is quoted beside the bounded leg
            import glob
            import re
            recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_%s_n1_cpu_protocol.json" % a.config.lower())),
                          key=lambda f: int(re.match(r"r(\d+)", os.path.basename(f)).group(1)))
            if recs and a.cpu_graphs < 8:
                try:
                    pr = json.loads(open(recs[-1]).read().strip().splitlines()[-1])["cpu_baseline"]
                    out["cpu_baseline"]["protocol_8d"] = {"value": pr["value"], "unit": pr["unit"], "cores": pr["cores"], "cpu": pr.get("cpu"),
                                                          "sample": pr["sample"], "source": "profiles/" + os.path.basename(recs[-1]),
                                                          "note": "the 8-graph working set is larger than the bounded leg's: its graphs/s is the lower, "
                                                                  "stricter figure (the bounded leg flatters the CPU)"}
                except Exception as exc:              # a malformed record must not take the bench line down
                    log("cpu protocol record unreadable:", exc)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                     # rank 0 may still be measuring its dense-signature leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
