"""CPU restatement of the reference hot path (ORACLE - test infrastructure only).

This file restates, in plain functional PyTorch-CPU fp32/fp64 arithmetic, the
algorithms of gnn_tracking's Interaction-Network edge-classification path, its
kNN graph construction and its object-condensation losses.  It is the CHECKER:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package ``gnn_tracking_amd`` never does, and has
no CPU fallback.

Pinning: ``oracle/make_golden.py`` runs the reference's own Python modules
(imported from /root/reference with stand-ins for the third-party packages this
image lacks) on seeded inputs, checks every function below against them, and
commits inputs + reference outputs as ``tests/golden/*.npz``.  The pinned
known-answer values of the reference's ``tests/test_losses.py:112-149`` are part
of those fixtures.  ``tests/test_oracle_golden.py`` re-checks this file against
the fixtures on every run (no reference needed).

Parameters are passed as a flat ``dict[str, Tensor]`` whose keys are exactly the
reference ``state_dict`` keys (e.g.
``ec_resin.network.layers.0.relational_model.layers.0.weight``).

Every function cites the reference file:line it follows (paths relative to
``/root/reference/src/gnn_tracking``).
"""

from __future__ import annotations

import math
from typing import Sequence

import torch
from torch import Tensor

__all__ = [
    "mlp",
    "interaction_network",
    "resin",
    "ec_for_graph_tcn",
    "falsify_low_pt_edges",
    "edge_weight_bce_loss",
    "knn_graph",
    "radius_graph",
    "knn_with_max_radius",
    "ml_graph_construction_edges",
    "good_node_mask",
    "condensation_loss_rg",
    "condensation_loss_tiger",
]


# --------------------------------------------------------------------------- MLP
def mlp(x: Tensor, p: dict, prefix: str, L: int, *, bias: bool = True,
        last_activation: bool = False) -> Tensor:
    """models/mlp.py:18-62.  Linear -> (ReLU -> Linear) x (L-1); weights [out,in].

    ``ModuleList`` index of the l-th Linear is ``2*l`` (ReLUs sit in between).
    """
    for l in range(L):
        w = p[f"{prefix}.layers.{2 * l}.weight"]
        x = x @ w.t()
        if bias:
            x = x + p[f"{prefix}.layers.{2 * l}.bias"]
        if l < L - 1 or last_activation:
            x = torch.clamp_min(x, 0.0)
    return x


# ------------------------------------------------------- Interaction network
def interaction_network(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: dict,
                        prefix: str) -> tuple[Tensor, Tensor]:
    """models/interaction_network.py:54-103 (+ PyG propagate, add / source_to_target).

    j = edge_index[0] (source), i = edge_index[1] (target).
    e~ = relational([x_i, x_j, e]); aggr[n] = sum_{e: i(e)=n} e~[e];
    x~ = object([x, aggr]).
    """
    src, tgt = edge_index[0], edge_index[1]
    m = torch.cat([x[tgt], x[src], edge_attr], dim=1)
    e_tilde = mlp(m, p, f"{prefix}.relational_model", 3)
    aggr = torch.zeros(x.shape[0], e_tilde.shape[1], dtype=x.dtype)
    aggr = aggr.index_add(0, tgt, e_tilde)
    x_tilde = mlp(torch.cat([x, aggr], dim=1), p, f"{prefix}.object_model", 3)
    return x_tilde, e_tilde


def _sqconvex(delta: Tensor, residue: Tensor | None, alpha: float) -> Tensor:
    """models/resin.py:17-42."""
    if residue is None or math.isclose(alpha, 0.0):
        return delta
    return math.sqrt(alpha) * residue + math.sqrt(1 - alpha) * delta


def resin(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: dict, prefix: str, *,
          n_layers: int, alpha: float = 0.5, residual_type: str = "skip1",
          collect_hidden_edge_embeds: bool = False, connect_to: int = 1, add_bn: bool = False):
    """models/resin.py:99-114 (skip1), :153-175 (skip2; ``add_bn``: BatchNorm1d in training
    mode on the inputs of every layer, :143-151), :197-216 (skip_top).

    Returns ``(x, edge_attr, edge_attrs or None)``.
    """
    relu = lambda t: torch.clamp_min(t, 0.0)  # noqa: E731
    ident = lambda t: t  # noqa: E731
    lp = lambda i: f"{prefix}.network.layers.{i}"  # noqa: E731
    edge_attrs = [edge_attr] if collect_hidden_edge_embeds else None
    if residual_type == "skip1":
        for i in range(n_layers):
            act = relu if i > 0 else ident
            dx, edge_attr = interaction_network(act(x), edge_index, act(edge_attr), p, lp(i))
            x = _sqconvex(dx, x, alpha)
            if edge_attrs is not None:
                edge_attrs.append(edge_attr)
    elif residual_type == "skip2":
        if n_layers % 2:
            raise ValueError("Only even number of layers allowed at the moment")
        # NB: the reference iterates ``pairwise(range(n))`` = (0,1),(1,2),...,(n-2,n-1)
        # (resin.py:157), i.e. overlapping pairs: n-1 blocks of two IN applications.
        def bn(kind, i, t):  # batch statistics, biased variance, eps 1e-5 (torch.nn.BatchNorm1d.forward)
            if not add_bn:
                return t
            w, b = p[f"{prefix}.network._{kind}_batch_norms.{i}.weight"], p[f"{prefix}.network._{kind}_batch_norms.{i}.bias"]
            return (t - t.mean(0)) / torch.sqrt(t.var(0, unbiased=False) + 1e-5) * w + b

        for i0 in range(n_layers - 1):
            i1 = i0 + 1
            act0 = relu if i0 > 0 else ident
            hx, he = interaction_network(act0(bn("node", i0, x)), edge_index, act0(bn("edge", i0, edge_attr)), p, lp(i0))
            dx, edge_attr = interaction_network(relu(bn("node", i1, hx)), edge_index, relu(bn("edge", i1, he)), p, lp(i1))
            x = _sqconvex(dx, x, alpha)
            if edge_attrs is not None:
                edge_attrs.append(edge_attr)
    elif residual_type == "skip_top":
        x_res = None
        for i in range(n_layers):
            if i == connect_to:
                x_res = x
            act = relu if i > 0 else ident
            dx, edge_attr = interaction_network(act(x), edge_index, act(edge_attr), p, lp(i))
            x = _sqconvex(dx, x_res, alpha) if x_res is not None else dx
            if edge_attrs is not None:
                edge_attrs.append(edge_attr)
    else:
        raise KeyError(residual_type)
    return x, edge_attr, edge_attrs


# ------------------------------------------------------------ Edge classifier
def ec_for_graph_tcn(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: dict, *,
                     L_ec: int = 3, alpha: float = 0.5, residual_type: str = "skip1",
                     use_intermediate_edge_embeddings: bool = True,
                     use_node_embedding: bool = True, connect_to: int = 1) -> dict:
    """models/edge_classifier.py:89-121."""
    relu = lambda t: torch.clamp_min(t, 0.0)  # noqa: E731
    h = relu(mlp(x, p, "ec_node_encoder", 2, bias=False))
    e = relu(mlp(edge_attr, p, "ec_edge_encoder", 2, bias=False))
    h, e, es = resin(h, edge_index, e, p, "ec_resin", n_layers=L_ec, alpha=alpha,
                     residual_type=residual_type,
                     collect_hidden_edge_embeds=use_intermediate_edge_embeddings,
                     connect_to=connect_to)
    w_in = torch.cat(es, dim=1) if use_intermediate_edge_embeddings else e
    if use_node_embedding:
        w_in = torch.cat([h[edge_index[0]], h[edge_index[1]], w_in], dim=1)
    eps = 0.001
    w = eps + (1 - 2 * eps) * torch.sigmoid(mlp(w_in, p, "W", 3))
    return {"W": w.squeeze(), "node_embedding": h, "edge_embedding": e}


def falsify_low_pt_edges(y: Tensor, edge_index: Tensor | None, pt: Tensor | None,
                         pt_thld: float = 0.0) -> Tensor:
    """metrics/losses/ec.py:71-92."""
    if math.isclose(pt_thld, 0.0):
        return y
    return y.bool() & (pt[edge_index[0]] > pt_thld)


def edge_weight_bce_loss(w: Tensor, y: Tensor, edge_index: Tensor | None = None,
                         pt: Tensor | None = None, pt_thld: float = 0.0) -> Tensor:
    """metrics/losses/ec.py:103-121: mean BCE, log clamped at -100 (torch semantics)."""
    y = falsify_low_pt_edges(y, edge_index, pt, pt_thld).to(w.dtype)
    lw = torch.clamp_min(torch.log(w), -100.0)
    l1w = torch.clamp_min(torch.log(1 - w), -100.0)
    return -(y * lw + (1 - y) * l1w).mean()


# ----------------------------------------------------------- kNN / radius graph
def _sq_dists_seq(xq: Tensor, xc: Tensor) -> Tensor:
    """[Q,C] squared distances, dimension-sequential fused accumulation
    d <- fma(t, t, d), t = xq_d - xc_d  (the arithmetic the HIP kernel uses; the C
    oracle ``oracle/knn_ref.c`` is the bit-exact statement of it - this torch
    version uses float64 FMA emulation: product exact in f64, one f32 rounding)."""
    d = torch.zeros(xq.shape[0], xc.shape[0], dtype=torch.float32)
    for k in range(xq.shape[1]):
        t = (xq[:, k].view(-1, 1) - xc[:, k].view(1, -1)).to(torch.float32)
        # t*t is exact in float64 (24-bit x 24-bit); adding the f32 d and rounding
        # once to f32 reproduces fmaf except for double-rounding corner cases.
        d = (t.double() * t.double() + d.double()).to(torch.float32)
    return d


def knn_graph(x: Tensor, k: int, chunk: int = 1024) -> Tensor:
    """torch_cluster.knn_graph(x, k) semantics as used at
    models/graph_construction.py:233 (no batch, loop=False, source_to_target):
    for every query q the k nearest other points, ascending distance, ties ->
    lower index; row0 = neighbour (source j), row1 = query (target i)."""
    N = x.shape[0]
    kk = min(k, N - 1)
    rows, cols = [], []
    xf = x.to(torch.float32)
    for s in range(0, N, chunk):
        q = xf[s:s + chunk]
        d = _sq_dists_seq(q, xf)
        idx = torch.arange(s, min(s + chunk, N))
        d[torch.arange(q.shape[0]), idx] = float("inf")
        ds, order = torch.sort(d, dim=1, stable=True)
        rows.append(order[:, :kk].reshape(-1))
        cols.append(idx.view(-1, 1).expand(-1, kk).reshape(-1))
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def radius_graph(x: Tensor, r: float, max_num_neighbors: int = 32, chunk: int = 1024,
                 batch: Tensor | None = None) -> Tensor:
    """torch_cluster.radius_graph(x, r, max_num_neighbors, loop=False) as used at
    metrics/losses/oc.py:115-117: neighbours with d < r (we test d^2 < r^2 on the
    sequential-fma squared distance), self excluded, at most ``max_num_neighbors``
    per query (the nearest ones; the reference tests never reach the cap)."""
    N = x.shape[0]
    xf = x.to(torch.float32)
    r2 = torch.tensor(float(r), dtype=torch.float32) ** 2
    rows, cols = [], []
    for s in range(0, N, chunk):
        q = xf[s:s + chunk]
        d = _sq_dists_seq(q, xf)
        idx = torch.arange(s, min(s + chunk, N))
        d[torch.arange(q.shape[0]), idx] = float("inf")
        if batch is not None:
            d = d.masked_fill(batch[idx].view(-1, 1) != batch.view(1, -1), float("inf"))
        ds, order = torch.sort(d, dim=1, stable=True)
        kk = min(max_num_neighbors, N - 1)
        ok = ds[:, :kk] < r2
        rows.append(order[:, :kk][ok])
        cols.append(idx.view(-1, 1).expand(-1, kk)[ok])
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def edge_lengths(x: Tensor, edge_index: Tensor) -> Tensor:
    """||x[e0]-x[e1]||_2 in fp32, dimension-sequential (sqrt of the fma chain)."""
    t = x[edge_index[0]].float() - x[edge_index[1]].float()
    d = torch.zeros(t.shape[0], dtype=torch.float32)
    for k in range(t.shape[1]):
        d = (t[:, k].double() * t[:, k].double() + d.double()).to(torch.float32)
    return torch.sqrt(d)


def knn_with_max_radius(x: Tensor, k: int, max_radius: float | None = None) -> Tensor:
    """models/graph_construction.py:222-237: kNN, then keep ``||x_j - x_i|| < r``
    (strict, L2 norm - not squared - evaluated after the kNN)."""
    ei = knn_graph(x, k)
    if max_radius is not None:
        ei = ei[:, edge_lengths(x, ei) < max_radius]
    return ei


def ml_graph_construction_edges(x: Tensor, particle_id: Tensor, edge_index: Tensor, ratio_of_false=None):
    """models/graph_construction.py:365-367 and :386-393: edge labels (int64
    compare, noise pid<=0 never true) and edge features ``[x_j - x_i, x_j + x_i]``
    with j = edge_index[0], i = edge_index[1].  ``ratio_of_false`` (:373-384, training mode only): the FIRST
    ``int(n_true * ratio)`` false edges are kept, then all true edges - returns the new edge list as a third value."""
    e0, e1 = edge_index[0], edge_index[1]
    y = (particle_id[e0] == particle_id[e1]) & (particle_id[e0] > 0)
    if ratio_of_false:
        n_keep = int(y.sum() * ratio_of_false)
        false_edges = edge_index[:, ~y][:, :n_keep]
        true_edges = edge_index[:, y]
        edge_index = torch.cat((false_edges, true_edges), dim=1)
        y = torch.cat((torch.zeros(false_edges.shape[1]), torch.ones(true_edges.shape[1])))
        e0, e1 = edge_index[0], edge_index[1]
        feat = torch.cat([x[e0] - x[e1], x[e0] + x[e1]], dim=1)
        return y.long(), feat, edge_index
    feat = torch.cat([x[e0] - x[e1], x[e0] + x[e1]], dim=1)
    return y.long(), feat


# ---------------------------------------------------------- condensation losses
def good_node_mask(pt, particle_id, reconstructable, eta, pt_thld=0.9, max_eta=4.0):
    """utils/graph_masks.py:19-28."""
    return (pt > pt_thld) & (particle_id > 0) & (reconstructable > 0) & (eta.abs() < max_eta)


def _condensation_points(beta: Tensor, particle_id: Tensor, mask: Tensor):
    """metrics/losses/oc.py:16-43: per particle of interest (masked hits only) the
    hit with the largest beta.  Returns ``alphas_k`` (global hit indices, ordered by
    ascending particle id) and the boolean per-hit CP flag."""
    idx = torch.nonzero(mask).view(-1)
    pid_m, beta_m = particle_id[idx], beta[idx]
    uniq, inv = torch.unique(pid_m, sorted=True, return_inverse=True)
    K = uniq.shape[0]
    assert K > 0, "No particles found, cannot evaluate loss"
    # arg-max of beta per particle (ties -> lowest hit index), written without the
    # duplicate-index scatter of the reference (oc.py:20-23 relies on last-write-wins,
    # which torch only honours for small single-threaded tensors)
    order = torch.argsort(beta_m, descending=True, stable=True)
    grouped = order[torch.argsort(inv[order], stable=True)]      # by particle, beta descending
    counts = torch.bincount(inv, minlength=K)
    starts = torch.cumsum(counts, 0) - counts
    first = grouped[starts]
    alphas = idx[first]
    is_cp = torch.zeros_like(particle_id, dtype=torch.bool)
    is_cp[alphas] = True
    return alphas, is_cp


def condensation_loss_rg(*, beta, x, particle_id, mask, q_min=0.01,
                         radius_threshold=1.0, max_num_neighbors=256,
                         radius_edges: Tensor | None = None) -> dict:
    """metrics/losses/oc.py:87-161 (the four loss terms, un-weighted)."""
    alphas, is_cp = _condensation_points(beta, particle_id, mask)
    q = torch.arctanh(beta) ** 2 + q_min
    if radius_edges is None:
        radius_edges = radius_graph(x, radius_threshold, max_num_neighbors)
    e0, e1 = radius_edges[0], radius_edges[1]
    # repulsive: edges whose first endpoint is a CP and that join different particles
    keep = is_cp[e0] & (particle_id[e0] != particle_id[e1])
    r0, r1 = e0[keep], e1[keep]
    d2 = ((x[r0] - x[r1]) ** 2).sum(-1)
    vr = ((radius_threshold - torch.sqrt(1e-9 + d2)) * q[r0] * q[r1]).sum()
    # attractive: every masked non-CP hit to the CP of its particle
    non_cp = torch.nonzero(~is_cp & mask).view(-1)
    cp_of = alphas[torch.searchsorted(particle_id[alphas], particle_id[non_cp])]
    va = (((x[non_cp] - x[cp_of]) ** 2).sum(-1) * q[non_cp] * q[cp_of]).sum()
    n_hits = mask.shape[0]
    n_oi = mask.sum()
    K = alphas.shape[0]
    eps = 1e-9
    return {
        "attractive": va / (eps + n_oi - K),
        "repulsive": vr / (eps + (K - 1) * n_hits),
        "coward": (1 - beta[alphas]).mean(),
        "noise": beta[particle_id == 0].mean(),
    }


def condensation_loss_tiger(*, beta, x, particle_id, mask, q_min=0.01) -> dict:
    """metrics/losses/oc.py:251-347 with ``max_n_rep=0`` (no sub-sampling) and
    ``noise_threshold=0``: dense N x K formulation."""
    eps = 1e-9
    uniq = torch.unique(particle_id[mask])
    assert uniq.numel() > 0, "No particles of interest found, cannot evaluate loss"
    att = particle_id.view(-1, 1) == uniq.view(1, -1)          # [N,K]
    q = torch.arctanh(beta) ** 2 + q_min
    alphas = torch.argmax(q.view(-1, 1) * att, dim=0)           # [K]
    qw = q.view(-1, 1) * q[alphas].view(1, -1)
    dist = torch.cdist(x, x[alphas])
    n_hits = mask.shape[0]
    K = alphas.shape[0]
    norm_rep = eps + (K - 1) * n_hits
    norm_att = eps + mask.sum() - K
    v_att = (qw[att] * dist[att] ** 2).sum() / norm_att
    rep = (~att) & (dist < 1)
    v_rep = (qw[rep] * (1 - dist[rep])).sum() / norm_rep
    return {
        "attractive": v_att,
        "repulsive": v_rep,
        "coward": (1 - beta[alphas]).mean(),
        "noise": beta[~(particle_id > 0)].mean(),
        "n_rep": rep.sum(),
    }


def condensation_loss_chunked(*, beta, x, particle_id, mask, mode: str, q_min=0.01, radius=1.0,
                              weights=(1.0, 1.0, 0.0, 0.0), chunk: int = 4096) -> dict:
    """The two functions above in float64 with O(chunk x K) memory: the same sums taken
    over blocks of hits, loss terms and the gradient of ``att + w_rep rep + w_coward coward
    + w_noise noise`` w.r.t. (x, beta) accumulated block by block.  It exists so that the
    200 k-hit event of BASELINE config 5 (K of several thousand: the dense formulation
    needs 7 GB per temporary) has an oracle; tests/test_oracle_golden.py checks it against
    the dense restatements (and through them the reference's pinned values) on the small
    cases.  ``mode``: "rg" (oc.py:87-161; condensation points from the masked hits,
    attraction of masked non-CP hits, repulsion 1 - sqrt(1e-9 + d2) inside the unit ball) or
    "tiger" (oc.py:251-347; condensation points = arg-max charge over ALL hits of a
    particle of interest, attraction of all its hits, repulsion 1 - d)."""
    assert mode in ("rg", "tiger")
    w_att, w_rep, w_cow, w_noise = weights
    N = int(mask.shape[0])
    if mode == "rg":
        alphas, is_cp = _condensation_points(beta, particle_id, mask)
        att_rows = mask & ~is_cp
    else:
        uniq = torch.unique(particle_id[mask])
        assert uniq.numel() > 0, "No particles of interest found, cannot evaluate loss"
        q32 = torch.arctanh(beta) ** 2 + q_min
        idx = torch.nonzero(torch.isin(particle_id, uniq)).view(-1)
        inv = torch.searchsorted(uniq, particle_id[idx])
        order = torch.argsort(q32[idx], descending=True, stable=True)   # ties -> lowest index (argmax)
        grouped = order[torch.argsort(inv[order], stable=True)]
        counts = torch.bincount(inv, minlength=uniq.numel())
        alphas = idx[grouped[torch.cumsum(counts, 0) - counts]]
        att_rows = torch.ones(N, dtype=torch.bool)
    K = int(alphas.shape[0])
    pid_k = particle_id[alphas]
    norm_att = float(1e-9 + mask.sum() - K)   # as the reference: float + int64 tensor promotes to fp32
    norm_rep = 1e-9 + (K - 1) * N
    b = beta.detach().double().requires_grad_(True)
    xx = x.detach().double().requires_grad_(True)
    v_att = v_rep = 0.0
    n_rep = 0
    for s in range(0, N, chunk):
        sl = slice(s, min(s + chunk, N))
        q = torch.arctanh(b) ** 2 + q_min
        xj, xk, qj, qk = xx[sl], xx[alphas], q[sl], q[alphas]
        d2 = ((xj ** 2).sum(1).view(-1, 1) + (xk ** 2).sum(1).view(1, -1) - 2.0 * xj @ xk.T).clamp_min(0.0)
        same = particle_id[sl].view(-1, 1) == pid_k.view(1, -1)
        qq = qj.view(-1, 1) * qk.view(1, -1)
        a_sel = same & att_rows[sl].view(-1, 1)
        att = (qq[a_sel] * d2[a_sel]).sum()
        r_sel = ~same & (d2 < radius * radius)
        dr = torch.sqrt(1e-9 + d2[r_sel]) if mode == "rg" else torch.sqrt(d2[r_sel])
        rep = (qq[r_sel] * (radius - dr)).sum()
        (w_att * att / norm_att + w_rep * rep / norm_rep).backward()
        v_att += float(att.detach())
        v_rep += float(rep.detach())
        n_rep += int(r_sel.sum())
    coward = (1 - b[alphas]).mean()
    noise_sel = (particle_id == 0) if mode == "rg" else ~(particle_id > 0)
    noise = b[noise_sel].mean()
    (w_cow * coward + w_noise * noise).backward()
    out = {"attractive": v_att / norm_att, "repulsive": v_rep / norm_rep, "coward": float(coward.detach()),
           "noise": float(noise.detach()), "n_rep": n_rep, "K": K, "grad_x": xx.grad, "grad_beta": b.grad}
    out["total"] = (w_att * out["attractive"] + w_rep * out["repulsive"] + w_cow * out["coward"]
                    + w_noise * out["noise"])
    return out


# ------------------------------------------------------------- training step (H)
def ec_training_step(x, edge_index, edge_attr, y, params: dict, *, model_kwargs: dict,
                     lr=1e-4, weight_decay=1e-4, pt=None, pt_thld=0.0):
    """training/ec.py:33-53 + training/base.py:106-116: forward, BCE, backward, one
    Adam step.  Returns (out, loss, grads, params_after)."""
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    out = ec_for_graph_tcn(x, edge_index, edge_attr, ps, **model_kwargs)
    loss = edge_weight_bce_loss(out["W"], y.float(), edge_index, pt, pt_thld)
    names = list(ps.keys())
    grads = torch.autograd.grad(loss, [ps[n] for n in names], allow_unused=True)
    for n, g in zip(names, grads):
        ps[n].grad = g if g is not None else torch.zeros_like(ps[n])
    opt = torch.optim.Adam([ps[n] for n in names], lr=lr, weight_decay=weight_decay)
    opt.step()
    return out, loss, {n: ps[n].grad for n in names}, {n: ps[n].detach() for n in names}


def tc_training_step(data: dict, params: dict, *, mlgc: dict, gtcn: dict, loss_kind: str, loss_weights: tuple,
                     lr: float = 1e-3):
    """training/tc.py:50-84 + training/base.py:94-116: ``data_preproc`` = ``MLGraphConstruction(ml=None)``
    (kNN on a slice of the node features, labels, edge features), ``GraphTCN`` forward,
    ``CondensationLossRG`` / ``CondensationLossTiger`` with the post-EC hit mask, backward, one
    Adam step (Lightning's default optimizer call ``torch.optim.Adam(parameters)``).  ``data``:
    x, particle_id, pt, eta, reconstructable.  Returns (built graph, model outputs, loss terms,
    total, grads, params after the step)."""
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    x, pid = data["x"], data["particle_id"]
    s0, s1 = mlgc["embedding_slice"]
    ei = knn_with_max_radius(x[:, s0:s1], mlgc["max_num_neighbors"], mlgc["max_radius"])
    y, edge_attr = ml_graph_construction_edges(x, pid, ei)
    out = graph_tcn(x, ei, edge_attr, ps, **gtcn)
    hm = out["ec_hit_mask"]
    pt, reco, eta = data["pt"][hm], data["reconstructable"][hm], data["eta"]
    if loss_kind == "tiger":  # (oc.py:207-213 vs :394-401: only the Tiger variant slices eta)
        eta = eta[hm]
    mask = good_node_mask(pt, pid[hm], reco, eta)
    fn = condensation_loss_tiger if loss_kind == "tiger" else condensation_loss_rg
    terms = fn(beta=out["B"], x=out["H"], particle_id=pid[hm], mask=mask)
    w_rep, w_cow, w_noise = loss_weights
    total = terms["attractive"] + w_rep * terms["repulsive"] + w_noise * terms["noise"] + w_cow * terms["coward"]
    names = list(ps)
    grads = torch.autograd.grad(total, [ps[n] for n in names], allow_unused=True)
    for n, g in zip(names, grads):
        ps[n].grad = g if g is not None else torch.zeros_like(ps[n])
    # base.py:106-112: the default scheduler ConstantLR(optimizer) scales the learning rate by its
    # default factor 1/3 from construction on (for the first five epochs)
    opt = torch.optim.Adam([ps[n] for n in names], lr=lr)
    torch.optim.lr_scheduler.ConstantLR(opt)
    opt.step()
    graph = {"edge_index": ei, "y": y, "edge_attr": edge_attr}
    return graph, out, terms, total, {n: ps[n].grad for n in names}, {n: ps[n].detach() for n in names}


# ------------------------------------------------------------- compiled kNN oracle
def knn_graph_c(x: Tensor, k: int, max_radius: float | None = None) -> Tensor:
    """Bit-exact kNN(+radius) edge list from the C oracle (oracle/knn_ref.c): the
    arithmetic spec of the HIP kernel (explicit fmaf chain, (d2, index) order)."""
    import ctypes
    import importlib.util
    import pathlib

    here = pathlib.Path(__file__).resolve().parent
    spec = importlib.util.spec_from_file_location("build_oracle", here / "build_oracle.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = ctypes.CDLL(str(mod.build()))
    lib.knn_ref_search.restype = ctypes.c_int64
    lib.knn_ref_search.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p]
    xf = x.detach().to(torch.float32).contiguous().cpu()
    n, dim = xf.shape
    kk = max(1, min(k, n - 1)) if n > 1 else 1
    nbr = torch.zeros(n * kk, dtype=torch.int32)
    cnt = torch.zeros(n, dtype=torch.int32)
    lib.knn_ref_search(xf.data_ptr(), n, dim, kk, float(max_radius or -1.0), nbr.data_ptr(), None,
                       cnt.data_ptr())
    nbr = nbr.view(n, kk)
    keep = torch.arange(kk).view(1, -1) < cnt.view(-1, 1)
    q = torch.arange(n).view(-1, 1).expand(n, kk)
    return torch.stack([nbr[keep].long(), q[keep]])


# ---------------------------------------------------------------------------------------
# bf16-storage mode (BASELINE configs 3/4).  The reference has no bf16 test or golden
# vector: Lightning's precision="bf16-mixed" autocast would run the same modules with
# bf16 matmul inputs/outputs.  PARITY UNPINNED against the reference for this mode; these
# functions restate the rounding contract of the kernels (include/gnntrk.h,
# gnntrk_mlp_forward_bf16) in fp32 torch so the kernels can be checked exactly, and the
# tests additionally bound the distance to the fp32 (golden-pinned) path.
def bf16_round(t: Tensor) -> Tensor:
    """fp32 -> nearest bf16 (ties to even) -> fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


def mlp_bf16_forward(x: Tensor, weights, biases):
    """``x`` [M, in] fp32 holding bf16-representable values.  Returns the fp32
    pre-epilogue output and the list of (bf16-rounded) hidden activations."""
    h = x
    hidden = []
    L = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        z = h.double() @ bf16_round(W).double().T  # products exact, fp32-like accumulate
        if b is not None:
            z = z + bf16_round(b).double()
        z = z.float()
        if i < L - 1:
            h = bf16_round(torch.relu(z))
            hidden.append(h)
        else:
            h = z
    return h, hidden


def mlp_bf16_epilogue(z: Tensor, epilogue: str, ca=0.0, cb=1.0, res=None) -> Tensor:
    if epilogue == "none":
        return bf16_round(z)
    if epilogue == "relu":
        return bf16_round(torch.relu(z))
    if epilogue == "residual":
        return bf16_round(ca * res + cb * z)
    if epilogue == "sigmoid":
        return ca + cb * torch.sigmoid(z)  # fp32 output
    raise ValueError(epilogue)


def mlp_bf16_backward(x: Tensor, weights, biases, g: Tensor, epilogue: str, ca=0.0, cb=1.0):
    """Backward of the bf16 fused MLP with the kernel's rounding points (include/gnntrk.h,
    gnntrk_mlp_backward_bf16).  ``x`` [M, in] = inputs after their ReLU (bf16 values), ``g``
    [M, out] fp32 = sum of the upstream terms.  Returns (gin [M, in] before the input-ReLU
    gate, list of dW, list of db) - dW/db fp32."""
    z, hidden = mlp_bf16_forward(x, weights, biases)
    if epilogue == "relu":
        g = g * (z > 0)
    elif epilogue == "residual":
        g = cb * g
    elif epilogue == "sigmoid":
        s = torch.sigmoid(z)
        g = g * cb * s * (1 - s)
    L = len(weights)
    gk = bf16_round(g.float())
    acts = [x] + hidden  # input of layer i
    dW, db = [None] * L, [None] * L
    for i in range(L - 1, -1, -1):
        dW[i] = (gk.double().T @ acts[i].double()).float()
        db[i] = gk.double().sum(0).float() if biases[i] is not None else None
        gprev = bf16_round((gk.double() @ bf16_round(weights[i]).double()).float())
        if i > 0:
            gprev = gprev * (acts[i] > 0)
        gk = gprev
    return gk, dW, db


def ec_for_graph_tcn_bf16(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: dict, *,
                          L_ec: int = 3, alpha: float = 0.5) -> dict:
    """``ec_for_graph_tcn`` (skip1, all embeddings fed to W) with the bf16-storage rounding
    points of gnn_tracking_amd (edge_classifier.py bf16 branch): inputs, every stored
    node/edge tensor and every hidden activation are bf16; sums are fp32."""
    def run(prefix, L, xin, bias=True):
        W = [p[f"{prefix}.layers.{2 * l}.weight"] for l in range(L)]
        b = [p[f"{prefix}.layers.{2 * l}.bias"] if bias else None for l in range(L)]
        return mlp_bf16_forward(xin, W, b)[0]

    relu = lambda t: torch.clamp_min(t, 0.0)  # noqa: E731
    src, tgt = edge_index[0], edge_index[1]
    h = bf16_round(relu(run("ec_node_encoder", 2, bf16_round(x), bias=False)))
    e = bf16_round(relu(run("ec_edge_encoder", 2, bf16_round(edge_attr), bias=False)))
    es = [e]
    ca, cb = math.sqrt(alpha), math.sqrt(1 - alpha)
    for i in range(L_ec):
        act = relu if i > 0 else (lambda t: t)
        xi, ei = act(h), act(e)
        pre = f"ec_resin.network.layers.{i}"
        et = bf16_round(run(f"{pre}.relational_model", 3, torch.cat([xi[tgt], xi[src], ei], 1)))
        aggr = bf16_round(torch.zeros(h.shape[0], et.shape[1]).index_add(0, tgt, et))
        z = run(f"{pre}.object_model", 3, torch.cat([xi, aggr], 1))
        h = bf16_round(z) if math.isclose(alpha, 0.0) else bf16_round(ca * h + cb * z)
        e = et
        es.append(e)
    w_in = torch.cat([h[src], h[tgt], torch.cat(es, 1)], 1)
    eps = 0.001
    w = eps + (1 - 2 * eps) * torch.sigmoid(run("W", 3, w_in))
    return {"W": w.squeeze(), "node_embedding": h, "edge_embedding": e}


# ---------------------------------------------------------------------------------------
# Graph track-condensation network
def res_fcnn_depth1(x: Tensor, p: dict, prefix: str) -> Tensor:
    """models/mlp.py:65-123 with depth=1, bias=False: normalise -> Linear -> ReLU -> Linear."""
    x = torch.nn.functional.normalize(x, p=2.0, dim=1, eps=1e-12)
    h = torch.clamp_min(x @ p[f"{prefix}._encoder.weight"].t(), 0.0)
    return h @ p[f"{prefix}._decoder.weight"].t()


def graph_tcn(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: dict, *, L_ec: int, L_hc: int,
              alpha_ec: float = 0.5, alpha_hc: float = 0.5, ec_threshold: float = 0.5,
              mask_orphan_nodes: bool = False, feed_edge_weights: bool = False,
              use_ec_embeddings_for_hc: bool = False, alpha_latent: float = 0.0,
              n_embedding_coords: int = 0, prefix: str = "_gtcn", layer: Tensor | None = None,
              heterogeneous_node_encoder: bool = False, ec_kind: str = "model",
              y: Tensor | None = None) -> dict:
    """models/track_condensation_networks.py:236-308 (``ModularGraphTCN.forward`` as built by
    ``GraphTCN``; node encoder homogeneous or, with ``layer``, heterogeneous :209-217)."""
    relu = lambda t: torch.clamp_min(t, 0.0)  # noqa: E731
    if ec_kind == "none":  # GraphTCNForMLGCPipeline (:522-582): no edge classifier, no cut
        h = relu(res_fcnn_depth1(x, p, f"{prefix}.hc_node_encoder"))
        e = relu(mlp(edge_attr, p, f"{prefix}.hc_edge_encoder", 2, bias=False))
        h, _, _ = resin(h, edge_index, e, p, f"{prefix}.hc_in", n_layers=L_hc, alpha=alpha_hc)
        beta = 1e-6 + (1 - 2e-6) * torch.sigmoid(mlp(h, p, f"{prefix}.p_beta", 3))
        hh = mlp(h, p, f"{prefix}.p_cluster", 3) * p[f"{prefix}._latent_normalization"]
        return {"W": None, "H": hh, "B": beta.squeeze(1), "ec_hit_mask": None, "ec_edge_mask": None}
    if ec_kind == "perfect":  # PerfectEdgeClassification with tpr = tnr = 1 (edge_classifier.py:147-163)
        ec = {"W": y.bool().float(), "edge_embedding": None, "node_embedding": None}
        assert not use_ec_embeddings_for_hc
    else:
        pe = {k[len(prefix) + 4:]: v for k, v in p.items() if k.startswith(prefix + ".ec.")}
        ec = ec_for_graph_tcn(x, edge_index, edge_attr, pe, L_ec=L_ec, alpha=alpha_ec)
    w = ec["W"].reshape(-1, 1)
    edge_mask = (w > ec_threshold).squeeze(1)
    ei = edge_index[:, edge_mask]
    ea = edge_attr[edge_mask]
    w_m = w[edge_mask]
    ee = ec["edge_embedding"][edge_mask] if ec["edge_embedding"] is not None else None
    xn, en = x, ec["node_embedding"]
    n = x.shape[0]
    if mask_orphan_nodes:
        connected = ei.flatten().unique()
        hit_mask = torch.zeros(n, dtype=torch.bool)
        hit_mask[connected] = True
        relabel = torch.full((n,), -1, dtype=torch.long)
        relabel[hit_mask] = torch.arange(int(hit_mask.sum()))
        ei = relabel[ei]
        xn, en = x[hit_mask], (en[hit_mask] if en is not None else None)
        layer = layer[hit_mask] if layer is not None else None
    else:
        hit_mask = torch.ones(n, dtype=torch.bool)
    xs, eas = [xn], [ea]
    if use_ec_embeddings_for_hc:
        xs.append(en)
        eas.append(ee)
    if feed_edge_weights:
        eas.append(w_m)
    if heterogeneous_node_encoder:
        h = relu(hetero_res_fcnn(torch.cat(xs, 1), layer, p, f"{prefix}.hc_node_encoder", 2, 0.0))
    else:
        h = relu(res_fcnn_depth1(torch.cat(xs, 1), p, f"{prefix}.hc_node_encoder"))
    e = relu(mlp(torch.cat(eas, 1), p, f"{prefix}.hc_edge_encoder", 2, bias=False))
    h, _, _ = resin(h, ei, e, p, f"{prefix}.hc_in", n_layers=L_hc, alpha=alpha_hc)
    eps = 1e-6
    beta = eps + (1 - 2 * eps) * torch.sigmoid(mlp(h, p, f"{prefix}.p_beta", 3))
    hh = mlp(h, p, f"{prefix}.p_cluster", 3)
    if alpha_latent:
        res = torch.nn.functional.pad(xn[:, :n_embedding_coords], (0, hh.shape[1] - n_embedding_coords))
        hh = math.sqrt(alpha_latent) * res + math.sqrt(1 - alpha_latent) * hh
    hh = hh * p[f"{prefix}._latent_normalization"]
    return {"W": w.squeeze(1), "H": hh, "B": beta.squeeze(1), "ec_hit_mask": hit_mask,
            "ec_edge_mask": edge_mask}


# ---------------------------------------------------------------------------------------
# Metric-learning hinge loss
def hinge_embedding_loss(*, x: Tensor, particle_id: Tensor, batch: Tensor | None, true_edge_index: Tensor,
                         mask: Tensor, r_emb: float = 1.0, max_num_neighbors: int = 256,
                         p_attr: float = 1.0, p_rep: float = 1.0, rep_normalization: str = "n_hits_oi",
                         rep_oi_only: bool = True) -> dict:
    """metrics/losses/metric_learning.py:14-55,94-178: attractive term over the true edges
    that start at a hit of interest, repulsive hinge over the radius-graph edges (same event,
    d < r_emb) that start at a hit of interest and join different particles."""
    eps = 1e-9
    near = radius_graph(x, r_emb, max_num_neighbors=max_num_neighbors, batch=batch)
    rep = near[:, mask[near[0]]] if rep_oi_only else near
    rep = rep[:, particle_id[rep[0]] != particle_id[rep[1]]]
    att = true_edge_index[:, mask[true_edge_index[0]]]
    d_att = torch.linalg.norm(x[att[0]] - x[att[1]], dim=-1)
    v_att = torch.sum(torch.pow(d_att, p_attr)) / (att.shape[1] + eps)
    d_rep = torch.linalg.norm(x[rep[0]] - x[rep[1]], dim=-1)
    if rep_normalization == "n_rep_edges":
        norm_rep = rep.shape[1] + eps
    elif rep_normalization == "n_hits_oi":
        norm_rep = mask.sum() + eps
    elif rep_normalization == "n_att_edges":
        norm_rep = att.shape[1] + eps
    else:
        raise ValueError(f"Normalization {rep_normalization} not recognized.")
    v_rep = torch.sum(torch.relu(r_emb - torch.pow(d_rep, p_rep))) / norm_rep
    return {"attractive": v_att, "repulsive": v_rep, "n_edges_att": att.shape[1],
            "n_edges_rep": rep.shape[1], "n_hits_oi": int(mask.sum())}


def res_fcnn(x: Tensor, p: dict, prefix: str, depth: int, alpha: float) -> Tensor:
    """models/mlp.py:65-123, bias=False, any depth."""
    x = torch.nn.functional.normalize(x, p=2.0, dim=1, eps=1e-12)
    x = x @ p[f"{prefix}._encoder.weight"].t()
    for i in range(depth - 1):
        x = math.sqrt(alpha) * x + math.sqrt(1 - alpha) * (torch.clamp_min(x, 0.0) @ p[f"{prefix}._layers.{i}.weight"].t())
    return torch.clamp_min(x, 0.0) @ p[f"{prefix}._decoder.weight"].t()


def ml_training_step(data: dict, params: dict, *, depth: int, alpha: float, loss: dict, lw_repulsive: float,
                     lr: float = 1e-3):
    """training/ml.py:25-78 + training/base.py:94-116: ``GraphConstructionFCNN`` forward
    (models/graph_construction.py:25-53: ``ResFCNN`` x ``_latent_normalization``),
    ``GraphConstructionHingeEmbeddingLoss``, backward, one Adam step under the default ConstantLR
    (factor 1/3).  ``data``: x, particle_id, pt, eta, reconstructable, batch, true_edge_index.
    Returns (H, loss terms, total, grads, params after the step)."""
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pp = {"." + k: v for k, v in ps.items()}
    h = res_fcnn(data["x"], pp, "", depth, alpha) * ps["_latent_normalization"]
    mask = good_node_mask(data["pt"], data["particle_id"], data["reconstructable"], data["eta"])
    terms = hinge_embedding_loss(x=h, particle_id=data["particle_id"], batch=data["batch"],
                                 true_edge_index=data["true_edge_index"], mask=mask, **loss)
    total = terms["attractive"] + lw_repulsive * terms["repulsive"]
    names = list(ps)
    grads = torch.autograd.grad(total, [ps[n] for n in names], allow_unused=True)
    for n, g in zip(names, grads):
        ps[n].grad = g if g is not None else torch.zeros_like(ps[n])
    opt = torch.optim.Adam([ps[n] for n in names], lr=lr)
    torch.optim.lr_scheduler.ConstantLR(opt)
    opt.step()
    return h, terms, total, {n: ps[n].grad for n in names}, {n: ps[n].detach() for n in names}


def hetero_res_fcnn(x: Tensor, layer: Tensor, p: dict, prefix: str, depth: int, alpha: float) -> Tensor:
    """models/mlp.py:123-178: pixel hits (layer 0..17) and strip hits through separate
    ``ResFCNN`` s, the two embeddings stacked pixel first."""
    pm = (layer >= 0) & (layer < 18)
    return torch.vstack([res_fcnn(x[pm], p, f"{prefix}.pixel_fcnn", depth, alpha),
                         res_fcnn(x[~pm], p, f"{prefix}.strip_fcnn", depth, alpha)])


# ---------------------------------------------------------------------------------------
# ModularGraphTCN glue (models/track_condensation_networks.py:251-262)
def threshold_compact(w: Tensor, threshold: float):
    """``mask = w > threshold`` and the ascending positions boolean indexing keeps."""
    mask = w.reshape(-1) > threshold
    return mask, torch.nonzero(mask).reshape(-1)


def connected_nodes(edge_index: Tensor, num_nodes: int):
    """``connected = edge_index.flatten().unique()`` (ascending), ``index_to_mask`` and the
    relabelling ``Data.subgraph(connected)`` applies to ``edge_index``."""
    connected = edge_index.flatten().unique()
    hit = torch.zeros(num_nodes, dtype=torch.bool)
    hit[connected] = True
    relabel = torch.full((num_nodes,), -1, dtype=torch.long)
    relabel[connected] = torch.arange(connected.numel())
    return hit, connected, relabel[edge_index]


# ---------------------------------------------------------------------------------------
# DBSCAN post-processing
def radius_neighbors(x, radius: float):
    """sklearn ``NearestNeighbors(radius).radius_neighbors(x)`` on its kd-tree path
    (postprocessing/fastrescanner.py:25-39): fp64, squared distance as a sequential sum over
    the features, member iff d2 <= radius**2, self included; neighbours ascending.
    Returns (offsets int64 [n+1], nbr int64 [M], dist float64 [M])."""
    import numpy as np

    x64 = np.asarray(x, dtype=np.float64)
    n, dim = x64.shape
    d2 = np.zeros((n, n))
    for d in range(dim):
        t = x64[:, None, d] - x64[None, :, d]
        d2 = d2 + t * t
    keep = d2 <= radius * radius
    off = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int64)
    src, nbr = np.nonzero(keep)
    return off, nbr.astype(np.int64), np.sqrt(d2[src, nbr])


def dbscan_labels(x, max_eps: float, eps: float, min_pts: int):
    """postprocessing/fastrescanner.py:41-66 with sklearn's ``dbscan_inner`` restated: edges
    with dist <= eps; core = at least ``min_pts`` neighbours (self included); depth-first
    expansion from every unlabelled core point in index order; noise = -1."""
    import numpy as np

    off, nbr, dist = radius_neighbors(x, max(max_eps, eps))
    n = len(off) - 1
    lists = [nbr[off[i]:off[i + 1]][dist[off[i]:off[i + 1]] <= eps] for i in range(n)]
    core = np.array([len(v) >= min_pts for v in lists], dtype=bool)
    labels = np.full(n, -1, dtype=np.int64)
    label_num = 0
    for i0 in range(n):
        if labels[i0] != -1 or not core[i0]:
            continue
        stack, i = [], i0
        while True:
            if labels[i] == -1:
                labels[i] = label_num
                if core[i]:
                    for v in lists[i]:
                        if labels[v] == -1:
                            stack.append(v)
            if not stack:
                break
            i = stack.pop()
        label_num += 1
    return labels


def graph_construction_resin(x: Tensor, edge_index: Tensor, edge_attr: Tensor, p: dict, *, h_outdim: int,
                             n_layers: int, alpha: float = 0.5, alpha_fcnn: float = 0.5) -> Tensor:
    """models/graph_construction.py:136-219 (``GraphConstructionResIN.forward``)."""
    h = mlp(x, p, "_node_encoder", 2, bias=False)
    e = mlp(edge_attr, p, "_edge_encoder", 2, bias=False)
    h, _, _ = resin(h, edge_index, e, p, "_resin", n_layers=n_layers, alpha=alpha)
    delta = mlp(h, p, "_decoder", 2, bias=False)
    return (alpha_fcnn * x[:, :h_outdim] + (1 - alpha_fcnn) * delta) * p["_latent_normalization"]


def focal_loss(w: Tensor, y: Tensor, *, alpha: float = 0.25, gamma: float = 2.0, pos_weight=1.0,
               edge_index: Tensor | None = None, pt: Tensor | None = None, pt_thld: float = 0.0,
               haughty: bool = False) -> Tensor:
    """metrics/losses/ec.py:13-31 with the label handling of ``EdgeWeightFocalLoss`` (:124-150:
    falsified target, scalar pos_weight) or ``HaughtyFocalLoss`` (:153-183: the target as given,
    pos_weight = the falsified label per edge)."""
    yf = y.float()
    if pt_thld > 0.0:
        yf = (y.bool() & (pt[edge_index[0]] > pt_thld)).float()
    t, pw = (y.float(), yf) if haughty else (yf, pos_weight)
    pos = -alpha * pw * (1 - w).pow(gamma) * t * w.log()
    neg = -(1.0 - alpha) * w.pow(gamma) * (1.0 - t) * (1 - w).log()
    return torch.mean(pos + neg)
