import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def host_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota.  The GPU box lists many more
    CPUs than its quota grants, and torch's default of one thread per listed CPU makes every CPU oracle leg crawl (the round-3
    full-size test spent 380 CPU-minutes that way)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


# The loaders' default (index_prep="auto") leaves the relation section of a batch to the GPU when one is visible; the host-side tests of
# this suite compare the HOST builders' arrays with the reference's, whatever box they run on.  GPU tests that want the device route
# pass index_prep="device_all" explicitly (tests/test_zzz_hip_relbatch.py, test_loader_default_*).
os.environ.setdefault("GTOS_INDEX_PREP", "host")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    torch.set_num_threads(host_cores())


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    """{'sd/a': x} -> {'a': tensor(x)} for one prefix."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith(prefix)}


def T(a):
    return torch.from_numpy(np.asarray(a))


def opt_mask(a):
    return None if a.size == 0 else torch.from_numpy(a).bool()


@pytest.fixture
def golden():
    return load_golden
