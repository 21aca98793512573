"""The CPU oracle (oracle/ref_cpu.py) against the committed golden vectors, which were
produced by the reference's own modules (oracle/make_golden.py) and include the pinned
known-answer values of the reference's tests/test_losses.py:112-123.  No GPU, no
reference checkout needed."""

import numpy as np
import torch

import parity_cases as P
import ref_cpu as O
from parity_cases import assert_close, load, tt


def _params(z, prefix):
    return {k[len(prefix):]: tt(z[k]) for k in z.files if k.startswith(prefix)}


def test_ec_testgraph_training_step():
    z = load("g1_ec_testgraph.npz")
    p0 = _params(z, "p0/")
    out, loss, grads, p1 = O.ec_training_step(tt(z["x"]), tt(z["edge_index"]), tt(z["edge_attr"]),
                                              tt(z["y"]), p0, model_kwargs=dict(L_ec=1))
    assert_close(out["W"], z["W"], 1e-6, "W")
    assert_close(out["node_embedding"], z["node_embedding"], 1e-6, "node")
    assert_close(out["edge_embedding"], z["edge_embedding"], 1e-6, "edge")
    assert_close(loss, z["loss"], 1e-6, "loss")
    assert abs(float(z["loss"]) - 0.5523450970649719) < 1e-6  # SURVEY.md 8c marker
    for k in p0:
        assert_close(grads[k], z["grad/" + k], 1e-5, "grad " + k)
        assert_close(p1[k], z["p1/" + k], 1e-6, "adam " + k)


def test_ec_variants():
    z = load("g2_ec_variants.npz")
    x, ei, ea, y, pt = (tt(z[k]) for k in ("x", "edge_index", "edge_attr", "y", "pt"))
    for name, kw in P.EC_VARIANTS.items():
        okw = {k: v for k, v in kw.items() if k not in ("hidden_dim", "interaction_node_dim", "interaction_edge_dim")}
        p0 = _params(z, f"{name}/p0/")
        out, loss, grads, _ = O.ec_training_step(x, ei, ea, y, p0, model_kwargs=okw, pt=pt,
                                                 pt_thld=0.9)
        assert_close(out["W"], z[f"{name}/W"], 1e-6, name + " W")
        assert_close(loss, z[f"{name}/loss"], 1e-6, name + " loss")
        for k in p0:
            assert_close(grads[k], z[f"{name}/grad/{k}"], 1e-5, f"{name} grad {k}")


def test_interaction_network_and_resin():
    z = load("g3_in_layer.npz")
    for name in ("std", "odd"):
        p = _params(z, f"{name}/p0/")
        xt, et = O.interaction_network(tt(z[f"{name}/x"]), tt(z[f"{name}/edge_index"]),
                                       tt(z[f"{name}/edge_attr"]), p, "in")
        assert_close(xt, z[f"{name}/x_tilde"], 1e-6, name + " x~")
        assert_close(et, z[f"{name}/e_tilde"], 1e-6, name + " e~")
    z = load("g3b_resin.npz")
    for name, kw in {"skip1": dict(n_layers=3, residual_type="skip1", alpha=0.5),
                     "skip2": dict(n_layers=2, residual_type="skip2", alpha=0.3),
                     "skip_top": dict(n_layers=3, residual_type="skip_top", alpha=0.7),
                     "skip2_bn": dict(n_layers=2, residual_type="skip2", alpha=0.3, add_bn=True)}.items():
        p = _params(z, f"{name}/p0/")
        x, e, es = O.resin(tt(z["x"]), tt(z["edge_index"]), tt(z["edge_attr"]), p, "r",
                           collect_hidden_edge_embeds=True, **kw)
        assert_close(x, z[f"{name}/x_out"], 1e-6, name)
        assert_close(torch.cat(es, 1), z[f"{name}/edge_attrs_cat"], 1e-6, name)


def test_knn_bit_exact():
    z = load("g4_knn.npz")
    for cn in ("tg3", "u2", "u8"):
        x = tt(z[f"{cn}/x"])
        for k in (1, 2, 3, 9):
            for r in (None, 1.0, 0.3):
                ref = tt(z[f"{cn}/k{k}_r{r}"])
                assert torch.equal(O.knn_with_max_radius(x, k, r), ref), (cn, k, r)


def test_condensation_losses_pinned():
    z = load("g5_oc.npz")
    for cn in ("td1", "td2", "td3"):
        t = {k: tt(z[f"{cn}/{k}"]) for k in ("beta", "x", "particle_id", "pt", "eta",
                                              "reconstructable")}
        mask = O.good_node_mask(t["pt"], t["particle_id"], t["reconstructable"], t["eta"])
        for strat, fn in (("tiger", O.condensation_loss_tiger), ("rg", O.condensation_loss_rg)):
            od = fn(beta=t["beta"], x=t["x"], particle_id=t["particle_id"], mask=mask)
            for i, k in enumerate(("attractive", "repulsive", "coward", "noise")):
                assert_close(od[k], z[f"{cn}/f64/{strat}/{k}"], 1e-9, f"{cn} {strat} {k}")
                if cn in ("td1", "td2"):  # the reference's own pinned numbers
                    pinned = float(z[f"{cn}/pinned"][i])
                    assert abs(float(od[k]) - pinned) <= 1e-6 * abs(pinned), (cn, strat, k)


def test_condensation_losses_chunked_oracle():
    """The blocked float64 oracle used at the 200 k-hit size of config 5 reproduces the
    reference's float64 loss terms, total and gradients (G5) whatever the block size."""
    z = load("g5_oc.npz")
    for cn in ("td1", "td2", "td3"):
        t = {k: tt(z[f"{cn}/{k}"]) for k in ("beta", "x", "particle_id", "pt", "eta",
                                              "reconstructable")}
        mask = O.good_node_mask(t["pt"], t["particle_id"], t["reconstructable"], t["eta"])
        for strat in ("tiger", "rg"):
            for chunk in (7, 4096):
                od = O.condensation_loss_chunked(beta=t["beta"], x=t["x"], particle_id=t["particle_id"],
                                                 mask=mask, mode=strat, weights=(1.0, 2.0, 0.25, 0.5),
                                                 chunk=chunk)
                for k in ("attractive", "repulsive", "coward", "noise", "total", "grad_x", "grad_beta"):
                    assert_close(torch.as_tensor(od[k], dtype=torch.float64), z[f"{cn}/f64/{strat}/{k}"], 1e-9,
                                 f"{cn} {strat} chunk={chunk} {k}")


def test_ml_graph_construction_edges():
    z = load("g6_mlgc.npz")
    g1 = load("g1_ec_testgraph.npz")
    x, pid = tt(g1["x"]), tt(g1["particle_id"])
    for k, r in ((4, 1.0), (16, 0.5)):
        ei = O.knn_with_max_radius(x[:, :3], k, r)
        y, f = O.ml_graph_construction_edges(x, pid, ei)
        assert torch.equal(ei, tt(z[f"k{k}_r{r}/edge_index"]))
        assert torch.equal(y, tt(z[f"k{k}_r{r}/y"]))
        assert torch.equal(f, tt(z[f"k{k}_r{r}/edge_attr"]))
    for k, r, rof in ((16, 0.5, 0.5), (4, 1.0, 2.0)):   # training mode's false-edge subsampling (graph_construction.py:373-384)
        ei = O.knn_with_max_radius(x[:, :3], k, r)
        y, f, ei2 = O.ml_graph_construction_edges(x, pid, ei, ratio_of_false=rof)
        assert torch.equal(ei2, tt(z[f"k{k}_r{r}_rof{rof}/edge_index"]))
        assert torch.equal(y, tt(z[f"k{k}_r{r}_rof{rof}/y"]).long())
        assert torch.equal(f, tt(z[f"k{k}_r{r}_rof{rof}/edge_attr"]))


def test_graph_tcn():
    z = load("g7_graph_tcn.npz")
    x, ei, ea, y = (tt(z[k]) for k in ("x", "edge_index", "edge_attr", "y"))
    for name, kw in P.GTCN_VARIANTS.items():
        p0 = {k: v.clone().requires_grad_(True) for k, v in _params(z, f"{name}/p0/").items()}
        okw = P.gtcn_oracle_kwargs(kw, float(z[f"{name}/ec_threshold"]), y=y)
        if kw.get("heterogeneous_node_encoder"):
            okw["layer"] = tt(z["layer"])
        out = O.graph_tcn(x, ei, ea, p0, **okw)
        if out["W"] is not None:
            assert torch.equal(out["ec_edge_mask"], tt(z[f"{name}/ec_edge_mask"]))
            assert torch.equal(out["ec_hit_mask"], tt(z[f"{name}/ec_hit_mask"]))
        for k in ("W", "H", "B"):
            if out[k] is not None:
                assert_close(out[k], z[f"{name}/{k}"], 1e-6 if k != "H" else 1e-5, f"{name} {k}")
        loss = (out["H"] * tt(z[f"{name}/rH"])).sum() + (out["B"] * tt(z[f"{name}/rB"])).sum()
        if "_cls" not in kw:
            loss = loss + O.edge_weight_bce_loss(out["W"], y.float())
        grads = torch.autograd.grad(loss, list(p0.values()), allow_unused=True)
        for (k, v), g in zip(p0.items(), grads):
            g = g if g is not None else torch.zeros_like(v)
            assert_close(g, z[f"{name}/grad/{k}"], 1e-5, f"{name} grad {k}")


def test_hinge_loss_pinned():
    z = load("g8_hinge.npz")
    pinned = {"n_hits_oi": (0.7307405975481213, 11.076146539572338),
              "n_rep_edges": (0.7307405975481213, 0.34612957938781874)}  # reference tests/test_losses.py:194-203
    for cn in ("td1", "td4"):
        t = {k: tt(z[f"{cn}/{k}"]) for k in ("x", "particle_id", "pt", "eta", "reconstructable", "batch",
                                              "true_edge_index")}
        mask = O.good_node_mask(t["pt"], t["particle_id"], t["reconstructable"], t["eta"])
        for norm in pinned:
            od = O.hinge_embedding_loss(x=t["x"], particle_id=t["particle_id"], batch=t["batch"],
                                        true_edge_index=t["true_edge_index"], mask=mask, rep_normalization=norm)
            assert_close(od["attractive"], z[f"{cn}/f64/{norm}/attractive"], 1e-9, f"{cn} att")
            assert_close(od["repulsive"], z[f"{cn}/f64/{norm}/repulsive"], 1e-9, f"{cn} rep")
            if cn == "td1":
                assert abs(float(od["attractive"]) - pinned[norm][0]) < 1e-9
                assert abs(float(od["repulsive"]) - pinned[norm][1]) < 1e-8


def test_gc_fcnn():
    z = load("g9_gc_fcnn.npz")
    for name, depth in (("d1_h40", 1), ("d4_h96", 4)):
        p0 = {"." + k: v for k, v in _params(z, f"{name}/p0/").items()}
        out = O.res_fcnn(tt(z["x"]), p0, "", depth, 0.6) * p0["._latent_normalization"]
        assert_close(out, z[f"{name}/H"], 1e-5, name)


def test_hetero_fcnn():
    z = load("g10_hetero_fcnn.npz")
    x, layer = tt(z["x"]), tt(z["layer"])
    for name, (cls, kw) in P.HETERO_CASES.items():
        p0 = {"." + k: v for k, v in _params(z, f"{name}/p0/").items()}
        if cls == "GraphConstructionHeteroResFCNN":
            out = O.hetero_res_fcnn(x, layer, p0, "", kw["depth"], kw["alpha"])
        else:
            enc = torch.clamp_min(O.hetero_res_fcnn(x, layer, p0, ".encoder", kw["depth_enc"], kw["alpha"]), 0.0)
            out = O.res_fcnn(enc, p0, ".fcnn", kw["depth"], kw["alpha"])
        assert_close(out * p0["._latent_normalization"], z[f"{name}/H"], 1e-5, name)


def test_dbscan_oracle_vs_reference_labels():
    z = load("g11_dbscan.npz")
    for cn in ("d2", "d3", "d8"):
        for eps, mp in P.DBSCAN_TRIALS[:3] + ((1.3, 3),):
            assert np.array_equal(O.dbscan_labels(z[f"{cn}/x"], 1.0, eps, mp), z[f"{cn}/eps{eps}_mp{mp}"])


def test_gc_resin():
    z = load("g12_gc_resin.npz")
    x, ei, ea = tt(z["x"]), tt(z["edge_index"]), tt(z["edge_attr"])
    for name, kw in P.GC_RESIN_CASES.items():
        p0 = _params(z, f"{name}/p0/")
        out = O.graph_construction_resin(x, ei, ea, p0, h_outdim=kw["h_outdim"], n_layers=kw["n_layers"],
                                         alpha=kw["alpha"], alpha_fcnn=kw["alpha_fcnn"])
        assert_close(out, z[f"{name}/H"], 1e-5, name)


def test_focal_losses():
    z = load("g13_focal.npz")
    w, y, ei, pt = tt(z["w"]), tt(z["y"]), tt(z["edge_index"]), tt(z["pt"])
    for name, (cls, kw) in P.FOCAL_CASES.items():
        loss = O.focal_loss(w, y, edge_index=ei, pt=pt, haughty=cls == "HaughtyFocalLoss", **kw)
        assert_close(loss, z[f"{name}/loss"], 1e-6, name)


def test_bf16_contract_vs_reference_autocast():
    """oracle/ref_cpu.py's restatement of the kernels' bf16 rounding contract against the
    reference's OWN modules under ``torch.autocast("cpu", bfloat16)`` (golden G2b): W within
    one bf16 ulp of the reference's bf16-rounded W, embeddings within four ulps of the largest
    entry.  (The contract accumulates in fp32 where autocast accumulates in bf16, so it sits
    closer to the fp32 reference than autocast itself does.)"""
    z, za = load("g2_ec_variants.npz"), load("g2b_ec_bf16_autocast.npz")
    x, ei, ea = tt(z["x"]), tt(z["edge_index"]), tt(z["edge_attr"])
    for name in ("skip1_L3_h40", "skip1_L2_h2", "alpha0"):
        kw = P.EC_VARIANTS[name]
        out = O.ec_for_graph_tcn_bf16(x, ei, ea, _params(z, f"{name}/p0/"), L_ec=kw["L_ec"],
                                      alpha=kw.get("alpha", 0.5))
        assert (out["W"] - tt(za[f"{name}/W"])).abs().max().item() <= P.BF16_PIN_W, name
        for k in ("node_embedding", "edge_embedding"):
            ref = tt(za[f"{name}/{k}"])
            assert (out[k] - ref).abs().max().item() <= P.BF16_PIN_EMB * max(1.0, ref.abs().max().item()), (name, k)
        # and the contract is the better approximation of the fp32 reference
        d_contract = (out["W"] - tt(z[f"{name}/W"])).abs().max().item()
        d_autocast = (tt(za[f"{name}/W"]) - tt(z[f"{name}/W"])).abs().max().item()
        assert d_contract < d_autocast, (name, d_contract, d_autocast)


def test_tc_training_step_oracle():
    """oracle.tc_training_step against the reference's own TCModule step (golden G14)."""
    z = load("g14_tc_step.npz")
    raw = {k: tt(z[k]) for k in ("x", "particle_id", "pt", "eta", "reconstructable")}
    for name, cfg in P.TC_STEP_CASES.items():
        gk = dict(cfg["gtcn"])
        okw = dict(L_ec=gk.pop("L_ec"), L_hc=gk.pop("L_hc"), ec_threshold=float(z[f"{name}/ec_threshold"]))
        for k in ("mask_orphan_nodes", "feed_edge_weights", "use_ec_embeddings_for_hc"):
            if k in gk:
                okw[k] = gk[k]
        graph, out, terms, total, grads, after = O.tc_training_step(
            raw, _params(z, f"{name}/p0/"), mlgc=P.TC_MLGC, gtcn=okw, loss_kind=cfg["loss"], loss_weights=P.TC_LOSS_W)
        assert torch.equal(graph["edge_index"], tt(z[f"{name}/edge_index"]))
        assert_close(total, z[f"{name}/loss"], 1e-6, name + " loss")
        for k, v in grads.items():
            assert_close(v, z[f"{name}/grad/{k}"], 1e-5, f"{name} grad {k}")
            assert_close(after[k], z[f"{name}/p1/{k}"], 1e-6, f"{name} adam {k}")


def test_tc_training_step_oracle_at_event_scale():
    """oracle.tc_training_step against the reference's own TCModule step on a 1500-hit event
    (golden G14b: the cfg5 model of bench.py + a Tiger / orphan-masking variant)."""
    z = load("g14b_tc_step_event.npz")
    raw = {k: tt(z[k]) for k in ("x", "particle_id", "pt", "eta", "reconstructable")}
    for name, cfg in P.TC_STEP_B_CASES.items():
        gk = dict(cfg["gtcn"])
        okw = dict(L_ec=gk.pop("L_ec"), L_hc=gk.pop("L_hc"), ec_threshold=float(z[f"{name}/ec_threshold"]))
        for k in ("mask_orphan_nodes", "feed_edge_weights", "use_ec_embeddings_for_hc", "alpha_latent",
                  "n_embedding_coords"):
            if k in gk:
                okw[k] = gk[k]
        graph, out, terms, total, grads, after = O.tc_training_step(
            raw, _params(z, f"{name}/p0/"), mlgc=P.TC_B_MLGC, gtcn=okw, loss_kind=cfg["loss"],
            loss_weights=cfg["loss_w"])
        assert torch.equal(graph["edge_index"], tt(z[f"{name}/edge_index"]))
        assert torch.equal(out["ec_edge_mask"], tt(z[f"{name}/ec_edge_mask"]))
        assert int(out["ec_edge_mask"].sum()) > 1000
        assert_close(total, z[f"{name}/loss"], 1e-6, name + " loss")
        for k, v in grads.items():
            assert_close(v, z[f"{name}/grad/{k}"], 1e-5, f"{name} grad {k}")
            assert_close(after[k], z[f"{name}/p1/{k}"], 1e-6, f"{name} adam {k}")


def test_ml_training_step_oracle():
    """oracle.ml_training_step against the reference's own MLModule step (golden G15)."""
    z = load("g15_ml_step.npz")
    raw = {k: tt(z[k]) for k in ("x", "particle_id", "pt", "eta", "reconstructable", "batch", "true_edge_index")}
    for name, cfg in P.ML_STEP_CASES.items():
        h, terms, total, grads, after = O.ml_training_step(
            raw, _params(z, f"{name}/p0/"), depth=cfg["model"]["depth"], alpha=cfg["model"].get("alpha", 0.6),
            loss=cfg["loss"], lw_repulsive=cfg["lw_repulsive"])
        assert terms["n_edges_rep"] == int(z[f"{name}/n_edges_rep"])
        assert_close(h, z[f"{name}/H"], 1e-5, name + " H")
        assert_close(total, z[f"{name}/loss"], 1e-6, name + " loss")
        for k, v in grads.items():
            assert_close(v, z[f"{name}/grad/{k}"], 1e-5, f"{name} grad {k}")
            assert_close(after[k], z[f"{name}/p1/{k}"], 1e-6, f"{name} adam {k}")
