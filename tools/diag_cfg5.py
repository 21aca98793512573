"""Per-step wall times and stage times of the cfg5 step with bench.py's timers off / on
(finds one-time host costs that a short timed region would average in): python tools/diag_cfg5.py [f32|bf16]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_tracking_amd import ops
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
args = bench.parse(["--workload", "cfg5"])
wl = bench.TCWorkload(args, 0, 1, torch.device("cuda", 0), dtype=dtype)
def run(n, tag):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); ts.append(1e3*(time.perf_counter()-t0))
    print(tag, [round(t, 1) for t in ts])
run(8, "plain")
wl.stage.on = True; run(4, "stage timers"); 
print({k: round(v["avg_ms"], 2) for k, v in wl.stage.summary().items()})
wl.stage.on = False
t = ops.KernelTimer(); ops.set_kernel_timer(t); run(4, "kernel timer"); ops.set_kernel_timer(None)
wl.stage.on = True; wl.stage.rec.clear(); ops.set_kernel_timer(ops.KernelTimer()); run(4, "both"); ops.set_kernel_timer(None)
print({k: round(v["avg_ms"], 2) for k, v in wl.stage.summary().items()})
