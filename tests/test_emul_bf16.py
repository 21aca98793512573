"""bf16-storage kernels on the wave64 CPU emulator (tests/emul): the same .hip sources
compiled with g++, checked against oracle/ref_cpu.py's restatement of the rounding
contract.  The GPU versions of these cases are in test_gpu_parity.py."""

import pytest

import parity_cases as P
from emul_util import emulated

pytestmark = pytest.mark.emul


def test_mlp_bf16_forward_emulated():
    with emulated():
        P.case_mlp_bf16_forward("cpu")


def test_mlp_bf16_backward_emulated():
    with emulated():
        P.case_mlp_bf16_backward("cpu")


def test_node_tap_emulated():
    with emulated():
        P.case_node_tap("cpu")
        P.case_fold_alias("cpu")


def test_mlp_bf16_backward_many_workgroups_emulated():
    """Enough rows that every grid cap of the backward launchers is reached on the emulated two-CU device
    (three light workgroups per CU for the weight-gradient-only shapes, two otherwise): the partial-sum
    reduction has to read exactly the blocks the launch wrote (the first four default shapes)."""
    with emulated():
        P.case_mlp_bf16_backward("cpu", rows=1500, full=False)


def test_ec_bf16_emulated():
    with emulated():
        P.case_ec_bf16("cpu")


def test_rows_bf16_emulated():
    with emulated():
        P.case_rows_bf16("cpu")


def test_mlp_bf16_stress_emulated():
    with emulated():
        P.case_mlp_bf16_stress("cpu", rounds=2, cases_per_round=6, row_choices=(1, 17, 33, 100))
        P.case_mlp_bf16_stress("cpu", rounds=1, cases_per_round=6, row_choices=(17, 33), seed=29, wide=True)
        P.case_mlp_bf16_stress("cpu", rounds=1, cases_per_round=6, row_choices=(17, 33), seed=31, wide_io=True)


def test_graph_tcn_bf16_emulated():
    with emulated():
        P.case_graph_tcn_bf16("cpu")
        for name, rep in P.case_graph_tcn_bf16_autocast("cpu").items():
            print("GraphTCN bf16 vs reference autocast:", name, {k: float(f"{v:.3g}") for k, v in rep.items()})


def test_bf16_reproducible_emulated():
    with emulated():
        P.case_bf16_reproducible("cpu")


def test_graph_tcn_wide_hidden_bf16_emulated():
    with emulated():
        P.case_graph_tcn_wide_hidden_bf16("cpu", hiddens=(64, 128), n_hits=200, n_edges=1200)


def test_python_shape_rule_and_library_agree_emulated():
    with emulated():
        P.case_bf16_shape_rules_agree("cpu")


def test_batch_gradient_is_sum_of_event_gradients_emulated():
    """The full-size cfg3 backward property test (test_gpu_parity.py) at emulator scale: 4 events x 300
    hits x 2 400 edges, fp32 and bf16 storage."""
    with emulated():
        print(P.case_full_size_backward("cpu", n_events=4, n_nodes=300, n_edges=2400))


def test_mlp_bf16_in_kernel_fold_emulated():
    """gnntrk_gfold on the wave64 emulator (the same kernel sources): carries, carry chains, windows, isolated
    nodes, the tail unit - relational and head shapes."""
    with emulated():
        assert P.case_mlp_bf16_fold("cpu", sizes=(1, 31, 33, 75, 640)) == 3 * 5 * 4
