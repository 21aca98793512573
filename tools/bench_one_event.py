"""The one-event-per-step extras of bench.py on their own (eager and captured):  python tools/bench_one_event.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for name, a in (("cfg2_hipgraph_bf16", ("bf16", 100)), ("cfg2_hipgraph_f32", ("f32", 100)),
                ("one_event_150k_2M_bf16", ("bf16", 50, 150_000, 2_000_000, 100, "one event of cfg3")),
                ("one_event_150k_2M_f32", ("f32", 30, 150_000, 2_000_000, 100, "one event of cfg3"))):
    r = bench.hipgraph_cfg2(dev, *a)
    print(name, json.dumps({k: r[k] for k in ("eager_ms_per_step", "ms_per_step", "value")}), flush=True)
