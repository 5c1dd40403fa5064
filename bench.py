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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--dense", action="store_true", help="dense relation[n,n,B,d] signature instead of the factored form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-graphs", type=int, default=2)
    ap.add_argument("--decode", action="store_true",
                    help="secondary benchmark (SURVEY 8f rank 2): beam search over the K/V-cached decoder instead of the train step")
    ap.add_argument("--beam", type=int, default=8)
    ap.add_argument("--max-steps", type=int, default=50)
    ap.add_argument("--dry-launch", action="store_true",
                    help="only check the rank launch / rendezvous (gloo, no GPU): every rank prints its rank and exits")
    return ap.parse_args()


def cpu_baseline(cfg_name, graphs):
    """Oracle (kind 'port') full training step on the host cores, fp32, on `graphs` graphs of the same config."""
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

    def step(i):
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
    step(1)
    log("cpu_baseline: warm-up step done")
    t0 = time.time()
    n_timed = 0
    while n_timed < 2 and (n_timed == 0 or time.time() - t0 < 15.0):
        step(2 + n_timed)
        n_timed += 1
    dt_ = (time.time() - t0) / n_timed
    return {"value": graphs / dt_, "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": "%d graphs of %s (n=%d, R=%d), fp32, full train step, %d timed steps, %.1f s/step" % (
                graphs, cfg_name, stats["n"], stats["R"], n_timed, dt_)}


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


def main():
    a = parse()
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
        print("dry-launch rank %d of %d ok" % (rank, world), flush=True)
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)   # RCCL on ROCm

    from gtos_amd import ops, synth
    from gtos_amd.config import build_generator
    from gtos_amd.generator import Generator
    from gtos_amd.train import Trainer

    cfg = synth.CONFIGS[a.config]
    cd = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    log("building model")
    model = build_generator(Generator, a.config, dev, factored_relation=not a.dense).to(dev)
    model.set_compute_dtype(cd)
    model.train()
    trainer = Trainer(model, cfg["d"], warmup_steps=2000, compute_dtype=cd, world_size=world, rank=rank)
    log("model on device; generating batch")
    batch, stats = synth.make_config_batch(a.config, rank=rank)        # weak scaling: B graphs per GPU
    log("batch", stats)
    from gtos_amd.pathtrie import attach_path_trie
    attach_path_trie(batch)            # host-side index preparation, part of batch assembly like the relation bank itself
    batch = {k: v.to(dev) for k, v in batch.items()}
    ops.set_seed(19940117 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        v = trainer.step(batch)
        log("warmup step", i, "loss", v)
    sync()
    log("warmup done")
    ops.PROFILE = {}
    t0 = time.perf_counter()
    losses = []
    for _ in range(a.steps):
        losses.append(trainer.step(batch))
    sync()
    elapsed = time.perf_counter() - t0
    log("timed region done", elapsed)
    prof, ops.PROFILE = ops.PROFILE, None
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- roofline of the relation-attention forward kernel (launches inside the timed region)
    n, B, d = stats["n"], stats["B"], cfg["d"]
    s_el = 2 if cd == torch.bfloat16 else 4
    P = n * n * B
    alg_bytes = P * 2 * d * s_el + 4 * n * B * d * s_el + n * B
    key = "rel_attn_fwd_mode1" if a.dense else "rel_attn_fwd_mode2"
    evs = prof.get(key, [])
    avg_ms = sum(s.elapsed_time(e) for s, e in evs) / max(1, len(evs))
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if evs else 0.0
    traffic = None          # HBM bytes per launch from the PMC passes (separate rocprofv3 runs, see the json's "method")
    pmc_file = os.path.join(ROOT, "profiles", "r1_rel_attn_pmc.json")
    if a.config == "C2" and a.dtype == "bf16" and os.path.exists(pmc_file):
        traffic = json.load(open(pmc_file))["traffic_bytes_per_launch"]
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "rel_attn_fwd_kernel (%s relation operand)" % ("dense" if a.dense else "factored"),
                "launches": len(evs), "avg_us": round(avg_ms * 1e3, 1), "algorithmic_bytes": alg_bytes}

    def avg_ms(name):
        ev_ = prof.get(name, [])
        return round(sum(s_.elapsed_time(e_) for s_, e_ in ev_) / max(1, len(ev_)), 3) if ev_ else None
    components = {"relation_encoder_fwd_ms": avg_ms("relation_encoder_fwd"), "relation_gru_bwd_ms": avg_ms("relation_gru_bwd"),
                  "graph_encoder_fwd_ms": avg_ms("graph_encoder_fwd"),
                  "note": "HIP-event spans on the main stream inside the timed region, per step"}

    if rank == 0 and not a.dense and a.config in ("C1", "C2", "C3"):
        # the same kernel on the reference's dense relation signature (rarb[S,T,B,2d] materialised), outside the timed region
        del trainer
        torch.cuda.empty_cache()
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(n, B, 3 * d, generator=g).to(dev, cd)
        rarb = (0.3 * torch.randn(n * n * B, 2 * d, generator=g)).to(dev, cd).view(n, n, B, 2 * d)
        ops.PROFILE = {}
        for _ in range(6):
            ops.attention_core(qkv, None, (0, d, 2 * d), d, cfg["H"], (d // cfg["H"]) ** -0.5, rel=rarb)
        torch.cuda.synchronize()
        ev = ops.PROFILE["rel_attn_fwd_mode1"][1:]
        ops.PROFILE = None
        dms = sum(s_.elapsed_time(e_) for s_, e_ in ev) / len(ev)
        roofline["dense_signature"] = {"avg_us": round(dms * 1e3, 1), "achieved": round(alg_bytes / dms / 1e6, 1),
                                       "frac": round(alg_bytes / dms / 1e6 / HBM_PEAK_GBS, 4)}
    if rank == 0:
        out = {"metric": "graphs/sec training step (100-node AMR, batch 64)", "value": world * B * a.steps / elapsed,
               "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "%s: generator/ %dx%d-node synthetic AMR graphs per GPU, %d-layer d=%d %d-head, "
                                      "full train step (fwd+bwd+allreduce+clip+Adam), dropout 0.2" % (
                                          a.config, B, cfg["N"], cfg["layers"], d, cfg["H"]),
                          "n": n, "B_per_gpu": B, "global_batch": world * B, "P": P, "R": stats["R"],
                          "mean_path_len": round(stats["mean_path_len"], 2), "T": stats["T"],
                          "relation_operand": "dense" if a.dense else "factored", "parallelism": "dp%d" % world,
                          "loss_first": losses[0], "loss_last": losses[-1]},
               "roofline": roofline, "components": components}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config, a.cpu_graphs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
