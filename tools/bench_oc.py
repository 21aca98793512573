#!/usr/bin/env python3
"""Config 5 loss leg: CondensationLossRG / Tiger fwd+bwd on a 200k-hit pile-up-like event
(x[N,8], K ~ 2000 particles of interest), GPU vs the CPU oracle on a sub-sample."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gnn_tracking_amd as G
import ref_cpu as O

from gnn_tracking_amd.synthetic import make_pileup_event as event  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    ev = event(500, n)
    d = {k: v.cuda() for k, v in ev.items()}
    for name, cls in (("rg", G.CondensationLossRG), ("tiger", G.CondensationLossTiger)):
        fn = cls(lw_repulsive=1.0, lw_noise=0.1, lw_coward=0.1)
        def step():
            b = d["beta"].clone().requires_grad_(True); x = d["x"].clone().requires_grad_(True)
            r = fn(beta=b, x=x, particle_id=d["particle_id"], reconstructable=d["reconstructable"],
                   pt=d["pt"], eta=d["eta"])
            r.loss.backward()
            return r, b.grad, x.grad
        step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): r, gb, gx = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        mask = O.good_node_mask(ev["pt"], ev["particle_id"], ev["reconstructable"], ev["eta"])
        K = len(torch.unique(ev["particle_id"][mask]))
        print(f"GPU {name}: fwd+bwd {dt*1e3:.1f} ms  N={n} K={K}  "
              + " ".join(f"{k}={float(v):.6g}" for k, v in r.loss_dct.items()))
        if name == "rg":   # the dense Tiger oracle needs N x K temporaries: sub-sample
            ns = 20000
            sub = {k: v[:ns] for k, v in ev.items()}
            m = O.good_node_mask(sub["pt"], sub["particle_id"], sub["reconstructable"], sub["eta"])
            torch.set_num_threads(min(32, os.cpu_count() or 1))
            t0 = time.perf_counter()
            bo = sub["beta"].clone().requires_grad_(True); xo = sub["x"].clone().requires_grad_(True)
            od = O.condensation_loss_rg(beta=bo, x=xo, particle_id=sub["particle_id"], mask=m)
            (od["attractive"] + od["repulsive"] + 0.1 * od["noise"] + 0.1 * od["coward"]).backward()
            dtc = time.perf_counter() - t0
            ds = {k: v[:ns].cuda() for k, v in ev.items()}
            b = ds["beta"].clone().requires_grad_(True); x = ds["x"].clone().requires_grad_(True)
            rs = fn(beta=b, x=x, particle_id=ds["particle_id"], reconstructable=ds["reconstructable"],
                    pt=ds["pt"], eta=ds["eta"])
            rs.loss.backward()
            errs = {k: abs(float(rs.loss_dct[k]) - float(od[k])) / max(1e-12, abs(float(od[k]))) for k in od}
            gerr = float((x.grad.cpu() - xo.grad).abs().max() / xo.grad.abs().max())
            print(f"  CPU oracle rg on first {ns}: {dtc:.2f} s; rel err terms {errs}; grad_x rel err {gerr:.2e}")
