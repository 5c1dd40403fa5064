"""`python bench.py --gpus N` must launch its own ranks (one per GPU) when it is not already under torch.distributed.run
-- the reference spawns its ranks itself (generator/train.py:173-190).  Checked on CPU with the --dry-launch leg (gloo)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dry-launch rank 0 of 2 ok" in r.stdout and "dry-launch rank 1 of 2 ok" in r.stdout


def test_bench_self_launches_eight_ranks_with_per_rank_thread_caps():
    """`python bench.py --gpus 8 --dry-launch`: the driver's 8-GPU command shape on this box (gloo rendezvous of eight ranks), and every
    rank reports its share of the usable host cores (bench.rank_cores: torch's intra-op pool and the loader threads are capped to it)."""
    import re
    r = _run(["--gpus", "8", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    seen = re.findall(r"dry-launch rank (\d) of 8 ok \(host threads per rank (\d+) of (\d+) usable cores\)", r.stdout)
    assert sorted(int(x[0]) for x in seen) == list(range(8)), r.stdout
    assert all(int(x[1]) == max(1, int(x[2]) // 8) for x in seen)


def test_bench_rejects_mismatched_world_size():
    r = _run(["--gpus", "2", "--dry-launch"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_cpu_baseline_leg_runs_on_a_tiny_config(monkeypatch):
    """bench.py's cpu_baseline leg end to end on the CPU (the oracle on a 2-graph C1 sample): the record carries every field
    the bench line promises -- a formatting slip here would only show up on the GPU box otherwise."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = bench.cpu_baseline("C1", 2, 1, budget_s=5.0, warmup=1)
    assert rec["kind"] == "port" and rec["unit"] == "graphs/s" and rec["value"] > 0 and rec["cores"] >= 1
    assert rec["encoder_only"]["value"] > 0 and "1 warm-up + 1 timed steps" in rec["sample"] and isinstance(rec["cpu"], str)


def test_bench_defaults_are_the_library_defaults(monkeypatch):
    """`python bench.py` with no flags measures what a user of the library gets: the reference's dropout semantics in the RelationEncoder
    (masks per (path, position), `encoder.MASK_SHARING`), one GPU, the BASELINE configuration the metric is quoted on, no opt-in legs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    from gtos_amd import encoder
    assert a.relation_masks == "path" == encoder.MASK_SHARING
    assert a.gpus == 1 and a.config == "C2" and not a.graph_leg and not a.fresh_batches
