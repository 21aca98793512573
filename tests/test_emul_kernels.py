"""The package's Python layer + the REAL kernel sources, executed on the CPU wave64
emulator (tests/emul), against goldens/oracle.  This is what catches layout, indexing
and autograd-glue bugs in the build container, which has no GPU.  Mostly the same cases as
under -m gpu (the emulator runs a lane as a fiber: a case takes seconds); only the full-size
ones are reduced."""

import pytest

import parity_cases as P
from emul_util import emulated

pytestmark = pytest.mark.emul


def test_graph_index():
    with emulated():
        P.case_graph_index("cpu")


def test_node_order():
    with emulated():
        P.case_node_order("cpu", n_hits=1500, n_edges=15000)


def test_graph_index_placed_from_cached_events():
    with emulated():
        P.case_graph_index_place("cpu")
        P.case_graph_index_place("cpu", order=False)


def test_resident_dataset():
    with emulated():
        P.case_resident_dataset("cpu", sizes=((300, 2500), (97, 800), (1, 0), (513, 4000), (50, 333)))


def test_graph_index_carry_and_fused_bce():
    with emulated():
        P.case_graph_index_carry("cpu")
        P.case_ec_carry_equals_gather("cpu")


def test_fused_mlp_forward_backward():
    with emulated():
        P.case_mlp("cpu")


def test_parameter_gradients_added_in_place_equal_autograd_accumulation():
    with emulated():
        P.case_grad_sink("cpu")


def test_fused_mlp_stress_small():
    with emulated():
        P.case_mlp_stress("cpu", rounds=2, seed=11, cases_per_round=8, row_choices=(1, 16, 17, 45, 300))


def test_interaction_network_layer():
    with emulated():
        P.case_in_layer("cpu")


def test_ec_testgraph_training_step():
    with emulated():
        P.case_ec_testgraph("cpu")


def test_ec_variants_subset():
    with emulated():
        P.case_ec_variants("cpu")


def test_edge_cases():
    with emulated():
        P.case_edge_cases("cpu")


def test_knn_against_c_oracle_and_goldens():
    with emulated():
        P.case_knn_oracle("cpu")
        P.case_knn_goldens("cpu", clouds=("tg3",))   # (u2 / u8 are 2 000-point clouds x 12 searches: -m gpu)
        P.case_ml_graph_construction("cpu")
        P.case_knn_batched("cpu")


def test_knn_pruned_equals_brute_force():
    with emulated():
        P.case_knn_pruned("cpu")


def test_condensation_losses_and_mask():
    with emulated():
        P.case_good_node_mask("cpu")
        P.case_condensation_losses("cpu")
        P.case_oc_sampling("cpu")
        P.case_rg_neighbor_cap("cpu", caps=(4, 256), n_hits=1200)


def test_condensation_losses_spatial_passes():
    with emulated():
        P.case_oc_spatial("cpu", cap_hits=1200, n_cloud=2000)


def test_graph_tcn_emulated():
    with emulated():
        P.case_graph_tcn("cpu")


def test_hinge_loss_emulated():
    with emulated():
        P.case_hinge_loss("cpu")


def test_gc_fcnn_emulated():
    with emulated():
        P.case_gc_fcnn("cpu")
        P.case_pc_transformer("cpu")


def test_res_fcnn_and_hinge_kernels_emulated():
    with emulated():
        print("res_fcnn worst weight-gradient error:", P.case_res_fcnn("cpu", rows=(1, 45)))
        P.case_hinge_terms("cpu")


def test_mlp_wide_emulated():
    with emulated():
        P.case_mlp_wide("cpu", rows=(1, 45))


def test_in_edge_wide_emulated():
    with emulated():
        P.case_in_edge_wide("cpu", sizes=((60, 700), (9, 1)))


def test_segment_sum_f32_emulated():
    with emulated():
        P.case_segment_sum_f32("cpu")


def test_hetero_fcnn_emulated():
    with emulated():
        P.case_hetero_fcnn("cpu")


def test_graph_cut_emulated():
    with emulated():
        P.case_graph_cut("cpu")


def test_dbscan_emulated():
    with emulated():
        P.case_dbscan("cpu")
        P.case_dbscan_pruned("cpu", n_big=3000)


def test_full_size_properties_tiny_emulated():
    # the property checks themselves, on a size the emulator finishes in seconds
    with emulated():
        P.case_full_size_properties("cpu", n_events=2, n_nodes=40, n_edges=96, n_hits=130)


def test_gc_resin_emulated():
    with emulated():
        P.case_gc_resin("cpu")


def test_focal_losses_emulated():
    with emulated():
        P.case_focal_losses("cpu")


def test_edge_ordered_outputs_emulated():
    with emulated():
        P.case_edge_ordered("cpu")


def test_tc_training_step_emulated():
    with emulated():
        P.case_tc_step("cpu")


def test_ml_training_step_emulated():
    with emulated():
        P.case_ml_step("cpu", names=("d1_h40_nrep",))


def test_tc_training_step_event_scale_emulated():
    """golden G14b (1500 hits, 21 617 built edges), the Tiger / orphan-masking variant"""
    with emulated():
        P.case_tc_step_event("cpu", names=("tiger_orphans_h24",))


def test_reference_configs_from_class_path():
    with emulated():
        P.case_class_path_configs("cpu")


def test_cfg3_event_case_small():
    with emulated():
        P.case_cfg3_event("cpu", n_hits=1500, n_edges=15000)
