"""Where the torch-op trie builder (gtos_amd.pathtrie_device) spends its time on the GPU: torch.profiler table of one call at C2 size.
python tools/time_device_tries.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtos_amd import synth                                            # noqa: E402
from gtos_amd.pathtrie_device import build_path_trie_device           # noqa: E402

dev = torch.device("cuda", 0)
batch, _ = synth.make_config_batch("C2", rank=0, B=64)
bank, length = batch["relation_bank"].to(dev), batch["relation_length"].to(dev)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    build_path_trie_device(bank, length)
    torch.cuda.synchronize(); print("build %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
from torch.profiler import profile, ProfilerActivity                  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    build_path_trie_device(bank, length)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=10, max_name_column_width=60))
