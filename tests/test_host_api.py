"""CPU-side tests of the product's host layer: the C-ABI library loads and exports what
include/cholmi.h declares, the client's input construction / parameter parsing / payload
schema match the reference's golden vectors, and the worker + in-process task API follow
worker_distrib.cpp's contract (error strings, ordering, write-once results).  The tile
arithmetic is injected from the oracle here (tests only): no GPU compute is attempted."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from dense_linear_app_amd import _lib, armonik as ak, client
from dense_linear_app_amd.worker import DagCholeskyWorker, handle_json

from oracle_engine import OracleTileBackend

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "golden_inputs.json")))


def sha(a):
    return hashlib.sha256(np.asfortranarray(a).tobytes(order="F")).hexdigest()


# ------------------------------------------------------------------ the boundary
def test_library_loads_and_exports_header_symbols():
    L = _lib.lib()
    syms = _lib.abi_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/cholmi.h but not exported"
    assert b"gfx950" in L.chol_version()


def test_no_cpu_backend():
    """chol_init(ncpu, 0) must refuse: the product never computes on the host."""
    L = _lib.lib()
    assert L.chol_init(4, 0) == -102
    assert b"no CPU backend" in L.chol_last_error()


def test_product_does_not_import_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "dense_linear_app_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libchol_oracle" not in text, f


# ------------------------------------------------------------------ client: inputs
@pytest.mark.parametrize("case", sorted(G["cases"]))
def test_client_input_construction_matches_reference(case):
    c = G["cases"][case]
    N, B = c["N"], c["B"]
    A = client.make_spd_like_chameleon(N, 100.0, "L", 12345)
    assert sha(A) == c["sha_A_before_dominance"]
    client.enforce_strict_diag_dominance(A)
    assert sha(A) == c["sha_A"]
    for name, h in c["tiles_sha"].items():
        _, i, j = name.split("/")
        assert name == client.block_id_from_ij(int(i), int(j))
        assert sha(client.extract_block_from_spd_matrix_colmajor(A, N, B, int(i), int(j))) == h


def test_client_upper_variant_and_ids():
    assert sha(client.make_spd_like_chameleon(12, 100.0, "U", 12345)) == G["upper_N12_sha"]
    for k, v in G["block_ids"].items():
        i, j = map(int, k.split(","))
        assert client.block_id_from_ij(i, j) == v


def test_load_params_matches_reference():
    err = io.StringIO()
    for c in G["load_params"]:
        p = client.load_params(c["argv"], env={}, err=err)
        assert [p.N, p.B] == c["NB"], c
    env = G["load_params_env"]["env"]
    for c in G["load_params_env"]["cases"]:
        p = client.load_params(c["argv"], env=env, err=err)
        assert [p.N, p.B] == c["NB"], c
    for c in G["parse_int_str"]:
        assert client.parse_int_str(c["s"], c["fallback"], "t", err) == c["value"], c
    assert "[CONFIG] Ignoring invalid value for --N='abc', using 12" in err.getvalue()


def test_payload_schema():
    """C1:44-97 field names/order; example of C2:100-103."""
    assert client.make_payload_potrf("id0", 4) == '{"op":"POTRF","B":4,"in":"id0"}'
    assert client.make_payload_trsm("L", "A", 4) == '{"op":"TRSM","B":4,"inL":"L","inA":"A"}'
    assert client.make_payload_syrk("C", "A", 4) == '{"op":"SYRK","B":4,"inC":"C","inA":"A"}'
    assert (client.make_payload_gemm("5c60", "7f3a", "9a7c", 448)
            == '{"op":"GEMM","B":448,"inC":"5c60","inAi":"7f3a","inAj":"9a7c"}')
    p = handle_json(client.make_payload("GEMM", ["c", "ai", "aj"], 8))
    assert (p.op, p.B, p.inC, p.inAi, p.inAj) == ("GEMM", 8, "c", "ai", "aj")
    # explicit routing: a TRSM stays a TRSM whatever the ids look like (C2:175 defect)
    assert handle_json(client.make_payload("TRSM", ["5d99-uuid", "ef13-uuid"], 4)).op == "TRSM"
    with pytest.raises(RuntimeError):
        client.make_payload("TRSM", ["only-one"], 4)


# ------------------------------------------------------------------ worker contract
def _plane_with(worker):
    plane = ak.ControlPlane()
    plane.register_worker(client.PARTITION, worker)
    sid = ak.SessionsClient(plane).create_session(client.default_task_options(), [client.PARTITION])
    return plane, sid


def _submit(plane, sid, payload, deps, blobs):
    rc, tc = ak.ResultsClient(plane), ak.TasksClient(plane)
    ids = rc.create_results_metadata(sid, ["output", "payload"] + list(blobs))
    for name, data in blobs.items():
        rc.upload_result_data(sid, ids[name], data)
    payload = payload.format(**ids)
    rc.upload_result_data(sid, ids["payload"], payload)
    tid = tc.submit_tasks(sid, [ak.TaskCreation(ids["payload"], [ids["output"]], [ids[d] for d in deps])],
                          client.default_task_options())[0]
    return ids, tc.get_task_output(tid)


def test_worker_error_strings():
    w = DagCholeskyWorker(backend=OracleTileBackend())
    plane, sid = _plane_with(w)
    tile = np.eye(4).tobytes()
    # missing dependency: payload names an id that is not among the task's data dependencies
    ids, st = _submit(plane, sid, '{{"op":"POTRF","B":4,"in":"nope"}}', [], {})
    assert not st.ok() and st.details() == "[Worker][POTF] Missing dependency: nope"
    ids, st = _submit(plane, sid, '{{"op":"TRSM","B":4,"inL":"{a}","inA":"zz"}}', ["a"], {"a": tile})
    assert st.details() == "[Worker][TRSM] Missing dependency: zz"
    ids, st = _submit(plane, sid, '{{"op":"SYRK","B":4,"inC":"cc","inA":"{a}"}}', ["a"], {"a": tile})
    assert st.details() == " [Worker][SYRK]Missing dependency: cc"  # sic (W2:370)
    ids, st = _submit(plane, sid, '{{"op":"GEMM","B":4,"inC":"{a}","inAi":"{a}","inAj":"q"}}', ["a"], {"a": tile})
    assert st.details() == "[Worker][GEMM] Missing dependency: q"
    # bad block size (W2:218-220)
    ids, st = _submit(plane, sid, '{{"op":"POTRF","B":4,"in":"{a}"}}', ["a"], {"a": np.ones(15).tobytes()})
    assert st.details() == "[Worker][POTF] Bad block size: expected 16 doubles, got 15"
    # info != 0 -> exception text (W2:243-244, 558-560)
    bad = np.eye(4)
    bad[2, 2] = -1
    ids, st = _submit(plane, sid, '{{"op":"POTRF","B":4,"in":"{a}"}}', ["a"], {"a": bad.tobytes()})
    assert st.details() == "Exception: [Worker][POTF] dpotrf info=3"
    ids, st = _submit(plane, sid, '{{"op":"LU","B":4}}', [], {})
    assert st.details() == "Unknown op=LU"
    ids, st = _submit(plane, sid, "not json", [], {})
    assert st.details().startswith("Exception: ")
    # a failed task aborts its output: waiting on it raises
    with pytest.raises(ak.ResultNotAvailable):
        ak.EventsClient(plane).wait_for_result_availability(sid, [ids["output"]])
    # success path: output blob = raw column-major doubles (W2:251-261)
    ids, st = _submit(plane, sid, '{{"op":"POTRF","B":4,"in":"{a}"}}', ["a"], {"a": (4 * np.eye(4)).tobytes()})
    assert st == ak.ProcessStatus.Ok
    out = np.frombuffer(ak.ResultsClient(plane).download_result_data(sid, ids["output"]))
    assert np.array_equal(out.reshape(4, 4), 2 * np.eye(4))


def test_results_are_write_once():
    plane, sid = _plane_with(DagCholeskyWorker(backend=OracleTileBackend()))
    rc = ak.ResultsClient(plane)
    rid = rc.create_results_metadata(sid, ["x"])["x"]
    rc.upload_result_data(sid, rid, b"abc")
    with pytest.raises(RuntimeError):
        rc.upload_result_data(sid, rid, b"def")


def test_a_crashing_batch_keeps_the_books_of_tasks_that_had_already_sent_their_results():
    """ControlPlane._run_batch's fallback: ExecuteBatch raises after its first task has sent its result.  That task is
    write-once done -- it must still reach `executed`, `completed` and on_task_done; the others run one by one."""
    class Crashy(DagCholeskyWorker):
        def ExecuteBatch(self, handlers):
            self.Execute(handlers[0])
            raise RuntimeError("boom")

    plane = ak.ControlPlane(batch_ready=True)
    plane.register_worker(client.PARTITION, Crashy(backend=OracleTileBackend()))
    sid = ak.SessionsClient(plane).create_session(client.default_task_options(), [client.PARTITION])
    seen = []
    plane.on_task_done = lambda t: seen.append(t.task_id)
    rc, tc = ak.ResultsClient(plane), ak.TasksClient(plane)
    ids = rc.create_results_metadata(sid, ["o1", "o2", "p1", "p2", "a"])
    rc.upload_result_data(sid, ids["a"], (4 * np.eye(4)).tobytes())
    for k in ("p1", "p2"):
        rc.upload_result_data(sid, ids[k], '{"op":"POTRF","B":4,"in":"%s"}' % ids["a"])
    tids = tc.submit_tasks(sid, [ak.TaskCreation(ids["p1"], [ids["o1"]], [ids["a"]]),
                                 ak.TaskCreation(ids["p2"], [ids["o2"]], [ids["a"]])], client.default_task_options())
    ak.EventsClient(plane).wait_for_result_availability(sid, [ids["o1"], ids["o2"]])
    assert sorted(seen) == sorted(tids) and sorted(plane.executed) == sorted(tids)
    assert all(tc.get_task_output(t) == ak.ProcessStatus.Ok for t in tids)


def test_a_task_that_depends_on_deleted_data_fails_instead_of_waiting_for_ever():
    """ResultsClient.delete_results_data leaves the metadata with status "deleted"; a task submitted against it can never
    become ready -- it must end in error (and its outputs aborted), not stay pending (advisor, round 4)."""
    plane, sid = _plane_with(DagCholeskyWorker(backend=OracleTileBackend()))
    rc, tc = ak.ResultsClient(plane), ak.TasksClient(plane)
    ids = rc.create_results_metadata(sid, ["o", "p", "a"])
    rc.upload_result_data(sid, ids["a"], (4 * np.eye(4)).tobytes())
    rc.upload_result_data(sid, ids["p"], '{"op":"POTRF","B":4,"in":"%s"}' % ids["a"])
    rc.delete_results_data(sid, [ids["a"]])
    (tid,) = tc.submit_tasks(sid, [ak.TaskCreation(ids["p"], [ids["o"]], [ids["a"]])], client.default_task_options())
    out = tc.get_task_output(tid)
    assert out is not None and out != ak.ProcessStatus.Ok and "deleted" in str(out.details if hasattr(out, "details") else out)
    with pytest.raises(Exception):
        ak.EventsClient(plane).wait_for_result_availability(sid, [ids["o"]])


def test_client_dag_default_case_matches_golden():
    """C1/C2 default N=12, B=4: 3 waves, 3/3/3/1 tasks; factor equals the committed fixture."""
    res = client.run_cholesky_dag(12, 4, worker=DagCholeskyWorker(backend=OracleTileBackend()))
    assert res.task_counts == {"POTRF": 3, "TRSM": 3, "SYRK": 3, "GEMM": 1}
    g = np.load(os.path.join(GOLD, "dag_N12_B4.npz"))
    assert np.abs(res.lower_factor() - g["L"]).max() <= 1e-13


def test_client_dag_config1_plumbing():
    """BASELINE config 1 (4x4 tiles of 256): 20 tasks 4/6/6/4 in the reference's order."""
    order = []
    w = DagCholeskyWorker(backend=OracleTileBackend())
    plane = ak.ControlPlane()
    orig = w.Execute

    def spy(th):
        order.append(handle_json(th.getPayload()).op)
        return orig(th)

    w.Execute = spy
    res = client.run_cholesky_dag(1024, 256, plane=plane, worker=w)
    assert res.task_counts == {"POTRF": 4, "TRSM": 6, "SYRK": 6, "GEMM": 4}
    assert order[:10] == ["POTRF", "TRSM", "TRSM", "TRSM", "SYRK", "GEMM", "SYRK", "GEMM", "GEMM", "SYRK"]
    g = np.load(os.path.join(GOLD, "dag_N1024_B256.npz"))
    L = res.lower_factor()
    assert np.abs(np.diag(L) - g["diag"]).max() <= 1e-12
    assert np.abs(L[g["probe_i"], g["probe_j"]] - g["probe_v"]).max() <= 1e-12


def test_ragged_matrix_fails_like_the_reference():
    """N not a multiple of B: the zero-padded last diagonal tile is singular (C2:285,299-303):
    POTRF reports info > 0 and the client's wait fails."""
    with pytest.raises(ak.ResultNotAvailable, match="dpotrf info=3"):
        client.run_cholesky_dag(10, 4, worker=DagCholeskyWorker(backend=OracleTileBackend()))


def test_c_driver_builds_as_c99_and_fails_loudly_without_a_gpu():
    """examples/v6_driver.c: the reference driver's call sequence on include/cholmi.h, compiled by
    a C compiler with -std=c99 -pedantic (the ABI is C: no C++, no torch types)."""
    import subprocess

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "all"], stdout=subprocess.DEVNULL)
    exe = os.path.join(root, "examples", "v6_driver")
    r = subprocess.run([exe, "1", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stderr
    import torch

    if not torch.cuda.is_available():
        args = "1 1 256 128 128 128 16384 256 256 0 0 256 256 1 1 42".split()
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


def test_batched_client_is_the_same_dag_with_four_submissions_per_wave():
    """the non-blocking client.  Same tasks, payloads and results; 4 submit_tasks calls per wave (POTRF; the
    TRSMs; the updates of column k+1, i.e. the panel chain's, with TaskOptions.priority 2; the other updates) instead of
    one per task."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from oracle_engine import OracleTileBackend

    from dense_linear_app_amd import armonik as ak, client
    from dense_linear_app_amd.worker import DagCholeskyWorker

    calls = {"n": 0}
    orig = ak.TasksClient.submit_tasks

    def counting(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)

    ak.TasksClient.submit_tasks = counting
    try:
        w = DagCholeskyWorker(backend=OracleTileBackend())
        serial = client.run_cholesky_dag(96, 16, worker=w)
        n_serial = calls["n"]
        calls["n"] = 0
        batched = client.run_cholesky_dag(96, 16, worker=w, batched=True)
        n_batched = calls["n"]
    finally:
        ak.TasksClient.submit_tasks = orig
    assert serial.task_counts == batched.task_counts == {"POTRF": 6, "TRSM": 15, "SYRK": 15, "GEMM": 20}
    assert n_serial == 56 and n_batched == 4 * 6 - 4  # the last wave is a POTRF alone, the one before has no update beyond column k+1
    assert np.array_equal(serial.lower_factor(), batched.lower_factor())


def test_guest_kernels_fit_on_a_cu_beside_a_resident_update_workgroup():
    """The schedule relies on co-residency: two trailing-update workgroups per
    CU, and every kernel of the panel chain able to start on a CU as soon as ONE of them has left
    it.  Read from the kernel descriptors of the built code object: allocated VGPRs per lane
    (512 per SIMD lane, one wave of each workgroup per SIMD) and LDS bytes (160 KiB per CU).
    The diagonal-block kernel once drifted to 406 VGPRs and silently waited for whole CUs to drain."""
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    import codeobj
    from dense_linear_app_amd._lib import LIB_PATH

    res = codeobj.kernel_resources(LIB_PATH)

    def find(*parts):
        hits = [v for k, v in res.items() if all(p in k for p in parts)]
        assert len(hits) == 1, (parts, [k for k in res if parts[0] in k])
        return hits[0]

    for t in ("d", "f"):
        # fp64 ships the eight-wave update (two waves of a workgroup per SIMD), fp32 the four-wave one
        # both ship the eight-wave update (two waves of a workgroup per SIMD)
        upd, per_simd = (find("k_trail_update_w8IdLi3E"), 2) if t == "d" else (find("k_trail_update_w8f"), 2)
        assert 2 * per_simd * upd["vgprs"] <= 512 and 2 * upd["lds"] <= 160 * 1024
        for guest in ("k_potrf_diagI%sE", "k_panel_solveI%sE", "k_panel_updateI%sE", "k_intile_stepI%sE",
                      "k_solve_smallI%sE", "k_small_updateI%sE"):
            g = find(guest % t)
            assert g["vgprs"] + per_simd * upd["vgprs"] <= 512, (guest % t, g, upd)
            assert g["lds"] + upd["lds"] <= 160 * 1024, (guest % t, g, upd)
    # the flow form of the tile POTRF (round 4) is only used for factorisations that are chain-bound from their first
    # wave on, i.e. on a mostly idle chip: its two kernels need not fit beside an update workgroup, but a workgroup of
    # each must fit on one CU together (one wave per SIMD each, the factor's LDS)
    for t in ("d", "f"):
        ff, fr = find("k_flow_factorI%sE" % t), find("k_flow_rowsI%sE" % t)
        assert ff["vgprs"] <= 512 and fr["vgprs"] <= 512 and ff["lds"] + fr["lds"] <= 160 * 1024


def test_v3_long_option_front_end_argument_checks():
    """The reference's v3 driver validates before it touches Chameleon (v3:122-195): the same
    checks run here without a GPU (every path below returns before CHAMELEON_Init)."""
    import io

    from dense_linear_app_amd import driver

    base = {"N": 2048, "NB": 256, "ncpu": 1, "ngpu": 1, "mat": "none", "dtyp": "d", "mb": 256, "nb": 256,
            "bsiz": 65536, "lm": 2048, "ln": 2048, "i": 0, "j": 0, "m": 2048, "n": 2048, "p": 1, "q": 1,
            "bump": 2048, "uplo": "L", "seed": 51}

    def run(**kw):
        d = dict(base, **kw)
        argv = []
        for k, v in d.items():
            if v is not None:
                argv += [f"--{k}", str(v)]
        err = io.StringIO()
        return driver.v3_test(argv, out=io.StringIO(), err=err), err.getvalue()

    rc, err = run(seed=None)
    assert rc == 1 and "all options are required" in err
    assert run(uplo="X") == (1, "Error: invalid --uplo X\n")
    assert run(dtyp="q") == (1, "Error: invalid --dtyp q\n")
    rc, err = run(dtyp="z")
    assert rc == 1 and "complex" in err
    assert run(mb=0)[1] == "Error: dimension arguments must be >0.\n"
    assert run(bsiz=100)[1] == "Error: --bsiz < mb*nb (bsiz=100 mb=256 nb=256).\n"
    assert run(i=4096)[1] == "Error: invalid offsets i=4096 j=0 (lm=2048 ln=2048).\n"
    assert "outside lm=2048" in run(i=512)[1]
    # --mat user: the buffer is whole tiles (v3:205-212); a ragged order is an error line, not a traceback
    rc, err = run(mat="user", lm=2000, ln=2000, m=2000, n=2000, N=2000)
    assert rc == 1 and err.startswith("Error: --mat user needs lm, ln multiples of mb, nb")
    assert driver.v3_test(["--help"], out=io.StringIO(), err=io.StringIO()) == 0
    assert driver.v3_test(["--bogus", "1"], out=io.StringIO(), err=io.StringIO()) == 1
