"""GPU parity of the worker path: the reference client's DAG (client_distrib.cpp:506-565)
submitted task by task through the in-process ArmoniK-style API, every task executed by
DagCholeskyWorker on the MI355X through the C ABI, compared with the CPU oracle and the
committed golden fixtures.  Also the v6_test-shaped driver."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_worker_path_default_case(cham, orc):
    from dense_linear_app_amd import client
    from dense_linear_app_amd.worker import DagCholeskyWorker, HipTileBackend

    w = DagCholeskyWorker()
    assert isinstance(w.backend, HipTileBackend)
    res = client.run_cholesky_dag(12, 4, worker=w)
    assert res.task_counts == {"POTRF": 3, "TRSM": 3, "SYRK": 3, "GEMM": 1}
    g = np.load(os.path.join(GOLD, "dag_N12_B4.npz"))
    assert np.abs(res.lower_factor() - g["L"]).max() <= 1e-12 * np.abs(g["L"]).max()


def test_worker_path_config1(cham, orc):
    """BASELINE config 1: N=1024, B=256, 20 tasks; identical (to 1e-12) to the oracle DAG."""
    from dense_linear_app_amd import client

    res = client.run_cholesky_dag(1024, 256)
    assert res.task_counts == {"POTRF": 4, "TRSM": 6, "SYRK": 6, "GEMM": 4}
    A = orc.reference_input(1024)
    Lref, info = orc.cholesky_lower(A, 256)
    L = res.lower_factor()
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-12
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13
    g = np.load(os.path.join(GOLD, "dag_N1024_B256.npz"))
    assert np.abs(np.diag(L) - g["diag"]).max() / np.abs(g["diag"]).max() <= 1e-12
    # SYRK semantics survive the round trip: the strict upper triangle of a diagonal
    # tile still holds the client's original entries (W2:416)
    t11 = res.tile(1, 1)
    assert np.array_equal(np.triu(t11, 1), np.triu(A[256:512, 256:512], 1))


def test_worker_path_odd_tile_size(cham, orc):
    """The VM sweep's best NB (448) is not a multiple of the 128 macro tile: staged + padded."""
    from dense_linear_app_amd import client

    N, B = 896, 448
    res = client.run_cholesky_dag(N, B)
    A = orc.reference_input(N)
    Lref, _ = orc.cholesky_lower(A, B)
    assert np.abs(res.lower_factor() - Lref).max() / np.abs(Lref).max() <= 1e-12


def test_worker_path_ragged_fails_like_reference(cham):
    from dense_linear_app_amd import armonik as ak, client

    with pytest.raises(ak.ResultNotAvailable, match="dpotrf info=3"):
        client.run_cholesky_dag(10, 4)


def test_v6_driver(cham):
    from dense_linear_app_amd import driver

    out, err = io.StringIO(), io.StringIO()
    N, NB = 2048, 256
    args = [1, 1, N, NB, NB, NB, NB * NB, N, N, 0, 0, N, N, 1, 1, 42]
    rc = driver.v6_test([str(a) for a in args], out=out, err=err)
    text = out.getvalue()
    assert rc == 0 and "Performance:" in text and "Gflop/s" in text and "PASS" in text
    assert driver.v6_test(["1", "1"], out=out, err=err) == 1 and "Usage:" in err.getvalue()


def test_bench_protocol_csv(cham, tmp_path):
    from dense_linear_app_amd import driver

    p = tmp_path / "bench.csv"
    rows = driver.bench([1024], [256, 512], csv_path=str(p), repeats=3, out=io.StringIO())
    assert len(rows) == 6 and all(r["exit_code"] == 0 for r in rows)
    head = p.read_text().splitlines()[0].split(",")
    assert head[:12] == ["timestamp", "scheduler", "mapping", "ncpu", "ngpu", "N", "NB", "run_idx", "ms",
                         "exit_code", "gflops", "rel_error"]
    assert float(rows[2]["residual_fro"]) < 1e-13 and float(rows[1]["residual_fro"]) == -1.0
    # rel_error keeps the reference's meaning (v6_test.c:72-86 as written): ~0.2 NB/N, every run
    assert all(0.01 < float(r["rel_error"]) < 0.2 for r in rows)
    assert float(rows[5]["rel_error"]) > float(rows[2]["rel_error"])  # grows with NB


def test_worker_path_device_resident_results(cham, orc):
    """ControlPlane(device_results=True): tiles are uploaded once and every version stays in HBM
    (armonik.DeviceBlob).  Same kernels, same inputs; the TRSM tasks reuse the block inverses the POTRF task left
    behind (content tags) where the host-blob path recomputes them from L(k,k): equal to rounding, not to the bit."""
    import time

    from dense_linear_app_amd import armonik as ak, client

    N, B = 2048, 512
    t0 = time.perf_counter()
    host = client.run_cholesky_dag(N, B)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev = client.run_cholesky_dag(N, B, device_results=True)
    t_dev = time.perf_counter() - t0
    assert dev.task_counts == host.task_counts
    blob = dev.plane._results[dev.latest["blk/3/1"]].data
    assert isinstance(blob, ak.DeviceBlob) and blob.tensor.is_cuda
    assert np.abs(dev.lower_factor() - host.lower_factor()).max() <= 1e-13 * np.abs(host.lower_factor()).max()
    A = orc.reference_input(N)
    L = dev.lower_factor()
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13
    print(f"worker path N={N} B={B}: host blobs {t_host * 1e3:.1f} ms, HBM-resident blobs {t_dev * 1e3:.1f} ms")
    # ragged / non-128-multiple tiles go through the padded staging path on the device too
    dev2 = client.run_cholesky_dag(896, 448, device_results=True)
    Lref, _ = orc.cholesky_lower(orc.reference_input(896), 448)
    assert np.abs(dev2.lower_factor() - Lref).max() / np.abs(Lref).max() <= 1e-12


@pytest.mark.parametrize("N,B", [(2048, 512), (1536, 256)])
def test_wave_level_execution_is_bit_identical_to_the_per_task_path(cham, orc, N, B):
    """run_cholesky_dag(device_results=True, batched=True) hands every ready task of a wave to the worker
    at once (ControlPlane.batch_ready -> DagCholeskyWorker.ExecuteBatch -> one chol_tile_batch per op class).  Same
    tasks, same counts, and -- the grouped launch runs the kernels of the one-tile calls -- the same bits."""
    from dense_linear_app_amd import armonik as ak, client
    from dense_linear_app_amd.worker import DagCholeskyWorker

    per_task = client.run_cholesky_dag(N, B, device_results=True)
    w = DagCholeskyWorker()
    wave = client.run_cholesky_dag(N, B, device_results=True, batched=True, worker=w)
    assert wave.task_counts == per_task.task_counts
    total = sum(wave.task_counts.values())
    assert len(wave.plane.executed) == total == len(per_task.plane.executed)
    assert w.batches >= 2 * (N // B - 2) and w.batched_tasks == total  # POTRF tasks included (asynchronous)
    assert isinstance(wave.plane._results[wave.latest["blk/2/1"]].data, ak.DeviceBlob)
    assert np.array_equal(wave.lower_factor(), per_task.lower_factor())
    A = orc.reference_input(N)
    L = wave.lower_factor()
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13


def test_wave_level_client_releases_superseded_tile_versions(cham, orc, monkeypatch):
    """Results are write-once, so every task adds a tile version; the wave-level client releases the superseded ones
    (ResultsClient.delete_results_data behind a wait for the launches that may still read them) once enough have piled
    up.  Forced after every wave here: same bits, and only the final versions still hold data."""
    from dense_linear_app_amd import armonik as ak, client

    N, B = 2048, 256
    plain = client.run_cholesky_dag(N, B, device_results=True, batched=True)
    monkeypatch.setenv("CHOLESKY_RETIRE_BYTES", "1")
    lean = client.run_cholesky_dag(N, B, device_results=True, batched=True)
    assert np.array_equal(lean.lower_factor(), plain.lower_factor())
    res = lean.plane._results
    final = set(lean.latest.values())
    held = [rid for rid, r in res.items() if isinstance(r.data, ak.DeviceBlob)]
    deleted = [rid for rid, r in res.items() if r.status == "deleted"]
    assert set(held) >= final and len(deleted) > 100
    # (the last waves' superseded versions wait for a release that never came: bounded by what two waves produce)
    assert len(held) - len(final) <= 3 * (N // B) ** 2 // 2
    with pytest.raises(ak.ResultNotAvailable):
        ak.ResultsClient(lean.plane).download_result_data(lean.session_id, deleted[0])


def test_wave_level_execution_reports_the_per_task_statuses(cham):
    """A batch with failing members: every task gets the ProcessStatus (text included) that Execute gives it alone,
    and the good ones still run in the grouped launch."""
    import json

    from dense_linear_app_amd import armonik as ak
    from dense_linear_app_amd.worker import DagCholeskyWorker

    B = 128

    def run(batch: bool):
        rng = np.random.default_rng(0)  # the same tiles in both runs
        plane = ak.ControlPlane(device_results=True, batch_ready=batch)
        w = DagCholeskyWorker()
        plane.register_worker("p", w)
        rc, tc, sc = ak.ResultsClient(plane), ak.TasksClient(plane), ak.SessionsClient(plane)
        opts = ak.TaskOptions(partition_id="p", max_retries=0)
        sid = sc.create_session(opts, ["p"])
        ids = rc.create_results_metadata(sid, ["C", "A", "Bm", "short"] + [f"o{q}" for q in range(5)] + [f"payload/{q}" for q in range(5)])
        tiles = {k: rng.standard_normal(B * B) for k in ("C", "A", "Bm")}
        for k, v in tiles.items():
            rc.upload_result_data(sid, ids[k], v.tobytes())
        rc.upload_result_data(sid, ids["short"], np.zeros(7).tobytes())
        payloads = [
            {"op": "GEMM", "B": B, "inC": ids["C"], "inAi": ids["A"], "inAj": ids["Bm"]},
            {"op": "SYRK", "B": B, "inC": ids["C"], "inA": ids["short"]},
            {"op": "NOPE", "B": B},
            {"op": "GEMM", "B": B, "inC": ids["C"], "inAi": ids["A"], "inAj": "missing-id"},
            {"op": "SYRK", "B": B, "inC": ids["C"], "inA": ids["A"]},
        ]
        deps = [[ids["C"], ids["A"], ids["Bm"]], [ids["C"], ids["short"]], [], [ids["C"], ids["A"]], [ids["C"], ids["A"]]]
        tcs = []
        for q, (pl, dp) in enumerate(zip(payloads, deps)):
            rc.upload_result_data(sid, ids[f"payload/{q}"], json.dumps(pl))
            tcs.append(ak.TaskCreation(ids[f"payload/{q}"], [ids[f"o{q}"]], dp))
        tids = tc.submit_tasks(sid, tcs, opts)
        outs = [tc.get_task_output(t) for t in tids]
        good = [rc.download_result_data(sid, ids[f"o{q}"]) for q in (0, 4)]
        return outs, good, w

    (o1, g1, _), (o2, g2, w2) = run(False), run(True)
    assert [o.details() for o in o1] == [o.details() for o in o2]
    assert o2[0].ok() and o2[4].ok() and not o2[1].ok() and not o2[2].ok() and not o2[3].ok()
    assert "Bad block size" in o2[1].details() and o2[2].details() == "Unknown op=NOPE" and "Missing dependency: missing-id" in o2[3].details()
    assert all(np.array_equal(np.frombuffer(a, dtype=np.float64), np.frombuffer(b, dtype=np.float64)) for a, b in zip(g1, g2))
    assert w2.batched_tasks == 2


def test_wave_level_execution_reports_a_failing_potrf_like_the_blocking_client(cham, orc):
    """Wave-level mode enqueues the POTRF tasks too (chol_potrf_batch), so a tile that is not positive definite is
    found out at the end of the run: the POTRF task then carries the reference's message (W2:243-244), its result
    and everything computed from it are aborted, and the client's wait raises as the blocking client's does."""
    from dense_linear_app_amd import armonik as ak, client

    N, B = 1024, 256
    A = orc.reference_input(N)
    A[600, 600] = -5.0
    with pytest.raises(ak.ResultNotAvailable, match="dpotrf info=89"):
        client.run_cholesky_dag(N, B, A=A)
    plane = ak.ControlPlane(device_results=True, batch_ready=True)
    with pytest.raises(ak.ResultNotAvailable, match="dpotrf info=89"):
        client.run_cholesky_dag(N, B, A=A, device_results=True, batched=True, plane=plane)
    outs = [t.output.details() for t in plane._tasks.values() if t.status == "error"]
    assert outs.count("Exception: [Worker][POTF] dpotrf info=89") == 1
    assert all(o == "a data dependency was aborted" or "dpotrf info=89" in o for o in outs) and len(outs) > 1
