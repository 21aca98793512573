"""Random wide-hidden shapes (hidden 63 .. 127) through the bf16 backward against the oracle, many rounds,
every round in its own process (a device fault ends a process, not the run).

    python tools/gpu_stress_wide.py [--rounds 20]
"""
import argparse
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=20)
args = ap.parse_args()
bad = 0
for r in range(args.rounds):
    code = ("import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'oracle'); sys.path.insert(0,'.');"
            "import torch, parity_cases as P;"
            f"P.case_mlp_bf16_stress(torch.device('cuda',0), rounds=3, seed={1000 + r}, cases_per_round=8, wide=True);"
            f"P.case_mlp_bf16_stress(torch.device('cuda',0), rounds=2, seed={5000 + r}, cases_per_round=8, wide_io=True);"
            "torch.cuda.synchronize(); print('ok')")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    ok = p.returncode == 0 and p.stdout.strip().endswith("ok")
    bad += not ok
    tail = [l for l in (p.stdout + p.stderr).splitlines() if "fault" in l.lower() or "Error" in l][-2:]
    print(f"round {r}: {'ok' if ok else 'FAILED rc ' + str(p.returncode)} {tail if not ok else ''}", flush=True)
print(f"{args.rounds - bad} / {args.rounds} rounds clean")
sys.exit(1 if bad else 0)
