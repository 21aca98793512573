"""bench.py's multi-rank control flow on CPU (gloo, world size 2) through the SAME code path
a GPU run takes: self-launch of N ranks when torchrun's environment is absent, torchrun's
environment contract, the rank-count check, barrier + max-over-ranks timing, the flat-gradient
all-reduce.  ``--stub`` swaps the HIP workload for a CPU toy model (the line says so)."""

import json
import os
import pathlib
import socket
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}:\n{out}"
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--stub", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and "rccl_ranks" not in d and d["self_launched"] is True
    assert d["steps"] == 3 and d["warmup"] == 1 and d["data"] == "stub"
    assert d["config"]["global_edges_per_step"] == 128          # both ranks' work is counted
    assert abs(d["value"] - 128 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-4 * d["value"]   # (the line carries six significant digits)


def test_self_launch_eight_ranks_dry_run():
    """What the driver's 8-GPU run exercises, without the GPUs: bench.py --gpus 8 starts eight ranks,
    they rendezvous, reduce and print ONE line (world size 8, every rank's work counted)."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--stub", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["backend"] == "gloo" and d["self_launched"] is True
    assert d["config"]["global_edges_per_step"] == 64 * 8
    # the N > 1 line explains itself: every rank's own time (the job's time is their maximum), the
    # collective library's version (RCCL only) and the GPU link matrix where rocm-smi can be asked
    assert len(d["ms_per_step_per_rank"]) == 8 and abs(max(d["ms_per_step_per_rank"]) - d["ms_per_step"]) < 1e-9 * max(1.0, d["ms_per_step"])
    assert set(d["collective"]) == {"rccl_version", "link_type", "hops"} and d["collective"]["rccl_version"] is None


def test_showtopo_parser():
    sys.path.insert(0, str(ROOT))
    import bench

    text = """
============================ ROCm System Management Interface ============================
================================ Weight between two GPUs =================================
       GPU0         GPU1
GPU0   0            15
GPU1   15           0

================================= Hops between two GPUs ==================================
       GPU0         GPU1
GPU0   0            1
GPU1   1            0

=============================== Link Type between two GPUs ===============================
       GPU0         GPU1
GPU0   0            XGMI
GPU1   XGMI         0

======================================= Numa Nodes =======================================
GPU[0]          : (Topology) Numa Node: 0
"""
    t = bench.parse_showtopo(text)
    assert t["hops"] == [[0, 1], [1, 0]] and t["link_type"] == [["0", "XGMI"], ["XGMI", "0"]]
    assert bench.parse_showtopo("no such tool") == {"link_type": None, "hops": None}


def test_a_rank_that_cannot_join_fails_loudly_not_for_ever():
    """World size 2 announced, only rank 0 started: the bounded rendezvous raises (non-zero exit,
    the backend's message on stderr) instead of hanging in the store / barrier."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               GNNTRK_DIST_TIMEOUT_S="8")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--stub", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and "could not be set up" in r.stderr, r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_under_torchrun_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", "2", "--stub", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and "rccl_ranks" not in d and d["self_launched"] is False


def test_rank_count_mismatch_is_refused():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--stub", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no line may be printed"


def test_cfg4_shards_are_balanced_and_rank_count_independent():
    sys.path.insert(0, str(ROOT))
    import bench
    from gnn_tracking_amd import dist as gdist

    sizes = bench.cfg4_event_sizes(256)
    assert sizes == bench.cfg4_event_sizes(256), "the event list must be the same on every rank"
    assert all(100_000 <= h <= 200_000 and e % 2 == 0 and abs(e - h * 40 / 3) <= 2 for h, e in sizes)
    shards = gdist.shard_events([e for _, e in sizes], 8)
    assert sorted(i for s in shards for i in s) == list(range(256))
    loads = [sum(sizes[i][1] for i in s) for s in shards]
    assert max(loads) / (sum(loads) / 8) < 1.01, "greedy assignment leaves > 1 % imbalance"
    # a rank of an N-rank job owns shards r, r + N, ...: every shard is owned exactly once
    for world in (1, 2, 4, 8):
        owned = [j for r in range(world) for j in range(8) if j % world == r]
        assert sorted(owned) == list(range(8))


def test_the_line_stays_small_whatever_rides_along(tmp_path, capsys, monkeypatch):
    """Round 5's default run printed one 22 KB line and the driver recorded ``parsed: null``.  ``bench.emit`` keeps the
    stdout line below ``LINE_LIMIT`` (6 KB) whatever the per-kernel table and the side runs hold: they go to
    ``bench_extra.json`` / stderr in full and into the line as digests."""
    sys.path.insert(0, str(ROOT))
    import bench

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "gpurun_out").mkdir()
    prose = "a long description of a workload, its model, its timed region and what it is not " * 6
    kernels = {f"mlp16_bwd_kernel<1, 3, 2, true, false, 2, IoRelational<{i}> >":
               {"launches": 20, "avg_ms": 2.3711012184619 + i, "tflops": 375.7241542720253, "alg_GBps": 2483.2343529472932,
                "traffic": 5587475190.4, "hbm_GBps": 2356.1, "hbm_frac": 0.2946123456789} for i in range(14)}
    extra = {f"cfg3_variant_number_{i}": {"workload": prose, "steps": 5, "warmup": 2, "ms_per_step": 23.5279123456 + i,
                                          "value": 2720170000.123, "unit": "edges/s", "final_loss": 0.6790112345,
                                          "roofline": {"frac": 0.31040429411841164, "kernel": "k" * 70, "what": prose},
                                          **({"stages": {s: {"calls": 10, "avg_ms": 1.2345678901} for s in
                                                         ("graph_build", "model_forward", "oc_loss_forward", "backward")}}
                                             if i < 2 else {})}
             for i in range(24)}
    line = {"metric": "edges_per_sec_fwd_bwd", "value": 2720170000.123456, "unit": "edges/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 23.52791234567, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "config": {"workload": prose[:400], "global_edges_per_step": 64000000},
            "roofline": {"bound": "hbm", "achieved": 2483.2343529472932, "peak": 8000.0, "unit": "GB/s",
                         "frac": 0.31040429411841164, "traffic": 5587475190.4, "kernel": "k" * 70,
                         "access_floor": {"what": prose, "skeleton_ms": 2.1533010005950928, "frac": 0.34180079784321693},
                         "floors": {f"kernel_{i}": {"skeleton_ms": 1.0, "kernel_ms_isolated": 1.2, "frac": 0.4} for i in range(3)}},
            "cpu_baseline": {"value": 358582.6592397416, "unit": "edges/s", "cores": 16, "kind": "port", "sample": prose[:300],
                             "s_all_iters": [5.6] * 3, "thread_probe_s": {"8": 1.0, "16": 0.7}},
            "parity_check": {"ok": True, "max_abs_W": 1.1387467384338379e-4, "bound": 5e-4, "event": prose},
            "library": {"csrc_sha256": "0" * 16, "lib_sha256": "1" * 16}, "stages": {"allreduce_adam": {"calls": 20, "avg_ms": 0.05}}}
    bench.emit(line, kernels, extra)
    cap = capsys.readouterr()
    out = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(out) == 1, "stdout must hold exactly ONE line"
    assert len(out[0].encode()) <= bench.LINE_LIMIT < 8192
    d = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline",
              "cpu_baseline", "parity_check", "library", "stages", "higher_is_better", "scaling", "vs_baseline", "data"):
        assert k in d, k
    assert d["roofline"]["frac"] == 0.310404 and d["roofline"]["traffic"] > 0 and "access_floor" in d["roofline"]
    assert d["cpu_baseline"]["cores"] == 16 and "thread_probe_s" not in d["cpu_baseline"]
    assert d["extra"]["cfg3_variant_number_3"]["ms_per_step"] == 26.528
    # the full record: side file (both places) and stderr
    for path in (tmp_path / bench.EXTRA_FILE, tmp_path / "gpurun_out" / bench.EXTRA_FILE):
        full = json.loads(path.read_text())
        assert full["extra"]["cfg3_variant_number_3"]["workload"] == prose and len(full["kernels"]) == 14
        assert full["cpu_baseline"]["thread_probe_s"] == {"8": 1.0, "16": 0.7}
    err = [ln for ln in cap.err.splitlines() if ln.startswith("bench_extra ")]
    assert len(err) == 1 and json.loads(err[0][len("bench_extra "):])["extra"].keys() == extra.keys()
    # a record that cannot fit even as digests loses the digests, never the contract keys
    bench.emit(line, {k + "x" * 300: v for k, v in kernels.items()}, {k + "y" * 300: v for k, v in extra.items()})
    out = capsys.readouterr().out.strip()
    assert len(out.encode()) <= bench.LINE_LIMIT and json.loads(out)["roofline"]["frac"] == 0.310404


def test_stub_run_prints_one_small_line():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--stub", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and len(out[0].encode()) < 8192
    d = json.loads(out[0])
    assert d["metric"] == "stub_not_a_measurement" and d["n_gpus"] == 1 and d["steps"] == 2
