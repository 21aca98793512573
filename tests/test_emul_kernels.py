"""The package's Python layer + the REAL kernel sources, executed on the CPU wave64
emulator (tests/emul), against goldens/oracle.  This is what catches layout, indexing
and autograd-glue bugs in the build container, which has no GPU.  Sizes are reduced to
keep the CPU suite at a few minutes; the full cases run under -m gpu."""

import pytest

import parity_cases as P
from emul_util import emulated

pytestmark = pytest.mark.emul


def test_graph_index():
    with emulated():
        P.case_graph_index("cpu")


def test_fused_mlp_forward_backward():
    with emulated():
        P.case_mlp("cpu", shapes=((14, 40, 4, 3), (14, 14, 5, 2), (30, 33, 7, 3)), rows=37)


def test_fused_mlp_stress_small():
    with emulated():
        P.case_mlp_stress("cpu", rounds=1, seed=11, cases_per_round=6, row_choices=(1, 16, 17, 45))


def test_interaction_network_layer():
    with emulated():
        P.case_in_layer("cpu", which=("odd",))


def test_ec_testgraph_training_step():
    with emulated():
        P.case_ec_testgraph("cpu")


def test_ec_variants_subset():
    with emulated():
        P.case_ec_variants("cpu", names=("skip2_L2",))


def test_edge_cases():
    with emulated():
        P.case_edge_cases("cpu")


def test_knn_against_c_oracle_and_goldens():
    with emulated():
        P.case_knn_oracle("cpu", shapes=((130, 8, 3, 0.5), (65, 3, 100, 0.4), (1, 3, 4, None),
                                         (2, 3, 4, None)))
        P.case_knn_goldens("cpu", clouds=("tg3",))
        P.case_ml_graph_construction("cpu")
        P.case_knn_batched("cpu", sizes=(1, 5, 40, 2))


def test_knn_pruned_equals_brute_force():
    with emulated():
        P.case_knn_pruned("cpu", shapes=((150, 8, 16, 1.0), (70, 3, 70, None)), batched_sizes=(1, 5, 70, 2))


def test_condensation_losses_and_mask():
    with emulated():
        P.case_good_node_mask("cpu")
        P.case_condensation_losses("cpu")
        P.case_oc_sampling("cpu")
        P.case_rg_neighbor_cap("cpu", caps=(4,), n_hits=500)


def test_condensation_losses_spatial_passes():
    with emulated():
        P.case_oc_spatial("cpu", cases=("td1",), sampling=False, caps=(), n_cloud=500)


def test_graph_tcn_emulated():
    with emulated():
        P.case_graph_tcn("cpu", names=("all_cut",))  # (test_tc_training_step_emulated runs a full GraphTCN)


def test_hinge_loss_emulated():
    with emulated():
        P.case_hinge_loss("cpu", cases=("td1",))


def test_gc_fcnn_emulated():
    with emulated():
        P.case_gc_fcnn("cpu")
        P.case_pc_transformer("cpu")


def test_hetero_fcnn_emulated():
    with emulated():
        P.case_hetero_fcnn("cpu", names=("hetero_d2",))


def test_graph_cut_emulated():
    with emulated():
        P.case_graph_cut("cpu")


def test_dbscan_emulated():
    with emulated():
        P.case_dbscan("cpu", clouds=("d8",), trials=((0.5, 2), (0.2, 5), (0.45, 6)))
        P.case_dbscan_pruned("cpu", clouds=("d2",), trials=((0.5, 2), (0.45, 6)), extras=False)


def test_full_size_properties_tiny_emulated():
    # the property checks themselves, on a size the emulator finishes in seconds
    with emulated():
        P.case_full_size_properties("cpu", n_events=2, n_nodes=40, n_edges=96, n_hits=130)


def test_gc_resin_emulated():
    with emulated():
        P.case_gc_resin("cpu", names=("h16_l1",))


def test_focal_losses_emulated():
    with emulated():
        P.case_focal_losses("cpu")


def test_edge_ordered_outputs_emulated():
    with emulated():
        P.case_edge_ordered("cpu", name="skip1_L2_h2")


def test_tc_training_step_emulated():
    with emulated():
        P.case_tc_step("cpu", names=("tiger_orphans",))
