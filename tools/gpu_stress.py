#!/usr/bin/env python3
"""Repeat the randomized fused-MLP parity stress on the GPU (fault / race hunting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as P
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P.case_mlp_stress("cuda", rounds=int(sys.argv[2]) if len(sys.argv) > 2 else 10, seed=seed)
P.case_mlp("cuda")
print("stress ok seed", seed, flush=True)
