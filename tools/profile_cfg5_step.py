#!/usr/bin/env python3
"""Where does the wall time of one cfg5 (object-condensation) step go?  torch.profiler table of one
step (host and device side) of bench.py's TCWorkload.

    python tools/profile_cfg5_step.py [--hits 200000]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hits", type=int, default=200_000)
ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"))
a = ap.parse_args()
args = bench.parse(["--workload", "cfg5", "--events", str(a.hits)])
dev = torch.device("cuda", 0)
wl = bench.TCWorkload(args, 0, 1, dev, dtype=a.dtype)
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
print(f"step {1e3 * (time.perf_counter() - t0) / 3:.2f} ms")
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    wl.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
