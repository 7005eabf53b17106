"""`python3 bench.py --gpus N` without a launcher (bench.py: launch_ranks): the parent starts the ranks as fresh child
processes, relays rank 0's line and supervises them -- a rank that dies, or a run in which nobody makes progress, ends
with ONE line carrying an "error" field and a non-zero exit status instead of ranks stuck in a receive.
(v6_test.c:26-27, 44-45: p, q go into the descriptor; one process per GPU here.)"""
import json
import os
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))


def _bench(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    return p.returncode, json.loads(lines[0]), time.time() - t0, p.stderr


def test_self_launch_without_a_gpu_reports_the_failed_rank():
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the ranks would run (covered by the gpu tests below)")
    rc, line, _, _ = _bench(["--gpus", "2", "--N", "2048", "--tile", "512", "--steps", "1", "--warmup", "0"], {}, 600)
    assert rc != 0 and line["value"] is None and line["n_gpus"] == 2 and line["launcher"] == "self"
    assert "exited with status" in line["error"] and line["failed_rank"] in (0, 1)


@pytest.mark.gpu
def test_self_launched_two_ranks_on_one_gpu_over_gloo():
    """Two ranks sharing cuda:0, tiles moved by gloo: the line of a 2 x 1 run, parsed."""
    rc, line, _, err = _bench(["--gpus", "2", "--N", "8192", "--tile", "512", "--steps", "1", "--warmup", "1"],
                              {"CHOLMI_DIST_BACKEND": "gloo"}, 900)
    assert rc == 0, err[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["launcher"] == "self"
    assert line["residual"] <= 1e-13
    # the grid is measured, not modelled: both candidates timed in the warm-up, the steps on the faster (bench.py: grid_probe)
    gp = line["config"]["grid_probe"]
    assert set(gp) == {"2x1", "1x2", "chosen"} and gp["chosen"] == line["config"]["grid"] and gp[gp["chosen"]] == min(gp["2x1"], gp["1x2"])
    assert sum(line["config"]["schedule_regimes"][k] for k in ("paired", "plain", "halves", "near_column")) == line["config"]["schedule_regimes"]["waves"] - 1
    ex = line["config"]["exchange"]
    assert ex["backend"] == "gloo" and len(ex["rank_update_ms"]) == 2 and all(x > 0 for x in ex["rank_update_ms"])
    assert len(line["config"]["schedule_calibration"]["mfma_probe_tflops"]) == 2


@pytest.mark.gpu
def test_self_launch_ends_a_run_whose_rank_died():
    """Rank 1 dies between warm-up and the timed steps (test switch): rank 0 would wait for its tiles for ever; the
    parent sees the exit status, kills rank 0, prints the error line and fails -- well inside the stall limit."""
    rc, line, dt, _ = _bench(["--gpus", "2", "--N", "4096", "--tile", "512", "--steps", "1", "--warmup", "1",
                              "--stall-timeout", "300"],
                             {"CHOLMI_DIST_BACKEND": "gloo", "CHOLMI_BENCH_KILL_RANK": "1"}, 900)
    assert rc != 0 and line["value"] is None
    assert line["failed_rank"] == 1 and line["exit_status"] == 17 and line["last_seen"].startswith("warmup")
    assert dt < 300


@pytest.mark.gpu
def test_bench_measures_the_update_kernels_hbm_traffic_in_the_run():
    """bench.py: roofline.traffic comes from two child passes under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE) of the same
    workload, not from a committed file.  Small workload here; the figure must be of the order of the kernel-level
    algorithmic bytes (every tile update reads its C tile and two operand tiles and writes C)."""
    import argparse
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = argparse.Namespace(N=8192, tile=512, dtype="f64", seed=42)
    got = bench.live_pmc_traffic(a, budget_s=200.0)
    assert got is not None, "rocprofv3 --pmc child passes failed"
    per_launch, launches, note = got
    assert launches > 0 and "measured in this run" in note
    nt = 8192 // 512
    tile_updates = 2 * sum((nt - 1 - k) * (nt - k) // 2 for k in range(nt))  # two factorisations per child
    algorithmic = 4.0 * 512 * 512 * 8 * tile_updates / launches
    assert 0.3 * algorithmic <= per_launch <= 3.0 * algorithmic, (per_launch, algorithmic)
