"""cProfile of the first cfg5 step that runs with the kernel timer armed (after reserving the timing
events): python tools/diag_timer.py [f32|bf16]"""
import sys, time, torch, cProfile, pstats, io
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_tracking_amd import ops
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
args = bench.parse(["--workload", "cfg5"])
wl = bench.TCWorkload(args, 0, 1, torch.device("cuda", 0), dtype=dtype)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
ops.reserve_timing_events(2000)
t = ops.KernelTimer(); ops.set_kernel_timer(t)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print("first kernel-timer step", round(dt * 1e3, 1), "ms")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3000])
for _ in range(3):
    t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); print("next", round((time.perf_counter() - t0) * 1e3, 1))
