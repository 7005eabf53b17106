"""The per-task loops of the in-process control plane and of the worker's wave-level execution exist twice: in Python
(armonik.py, worker.py) and in C on the same objects (csrc/fastplane.c -> dense_linear_app_amd/_fastplane.so).  No GPU
needed: a recording backend stands in for the grouped launches (test infrastructure only); what is compared is the
bookkeeping -- which tasks ran in which group with which operand addresses, every task's status and message, every
result's state -- between the two implementations, on whole DAGs and on the error paths the reference defines
(W2:195, 219, 370, 468, 547-549: missing dependencies, short blobs, unknown ops)."""
import numpy as np
import pytest

from dense_linear_app_amd import armonik as ak
from dense_linear_app_amd import client
from dense_linear_app_amd.worker import DagCholeskyWorker, HipTileBackend


class _FakeTensor:
    def __init__(self, nbytes, ptr):
        self.n, self.p = nbytes, ptr

    def numel(self):
        return self.n

    def element_size(self):
        return 1

    def data_ptr(self):
        return self.p


class RecordingBackend(HipTileBackend):
    """chol_tile_batch / chol_potrf_batch replaced by a log; addresses come from a counter, so two runs agree."""

    def __init__(self):
        self.next_ptr = 1 << 30
        self.calls = []

    def sync_inputs(self):
        pass

    def batch_alloc(self, m, B):
        base = self.next_ptr
        self.next_ptr += m * B * B * 8
        return _FakeTensor(m * B * B * 8, base), base

    def tile_batch(self, code, B, m, ptr, urgent=False):
        self.calls.append(("tile", code, B, m, bool(urgent), ptr.copy()))
        return 0

    def potrf_batch(self, B, m, ptr, slots):
        self.calls.append(("potrf", B, m, ptr.copy()))
        for q in range(m):
            slots[q] = len(self.calls) % 4096
        return 0


class _W(DagCholeskyWorker):
    def flush(self):
        self._deferred = []
        return []


@pytest.fixture
def fake_uploads(monkeypatch):
    state = {"ptr": 1 << 20}

    def from_bytes(cls, data):
        p = state["ptr"]
        state["ptr"] += len(data)
        return cls(_FakeTensor(len(data), p))

    monkeypatch.setattr(ak.DeviceBlob, "from_bytes", classmethod(from_bytes))
    return state


def _run(N, B, fast, monkeypatch, fake_uploads):
    if not fast:
        monkeypatch.setattr(ak, "FAST", None)
    fake_uploads["ptr"] = 1 << 20
    plane = ak.ControlPlane(device_results=True, batch_ready=True)
    plane.flush = lambda: None
    be = RecordingBackend()
    w = _W(backend=be)
    A = np.zeros((N, N), order="F")
    r = client.run_cholesky_dag(N, B, plane=plane, worker=w, A=A, device_results=True, batched=True)
    return r, plane, be, w


def _id_numbers(plane):
    """ids are prefix-counter: compare the two runs by the counter relative to the run's first id"""
    first = min(int(k.split("-")[1], 16) for k in plane._results)
    return first


@pytest.mark.skipif(ak.FAST is None, reason="the _fastplane extension is not built")
@pytest.mark.parametrize("N,B", [(1024, 128), (1536, 256), (640, 128)])
def test_c_and_python_loops_run_the_same_dag(N, B, monkeypatch, fake_uploads):
    rf, pf, bf, wf = _run(N, B, True, monkeypatch, fake_uploads)
    rp, pp, bp, wp = _run(N, B, False, monkeypatch, fake_uploads)
    assert rf.task_counts == rp.task_counts and sum(rf.task_counts.values()) == len(pf.executed) == len(pp.executed)
    assert wf.batches == wp.batches and wf.batched_tasks == wp.batched_tasks == len(pf.executed)
    # the same grouped launches, in the same order, with the same operand addresses and content tags
    assert len(bf.calls) == len(bp.calls)
    f0, p0 = _id_numbers(pf), _id_numbers(pp)
    for a, b in zip(bf.calls, bp.calls):
        assert a[:-1] == b[:-1]
        pa, pb = a[-1], b[-1]
        assert np.array_equal(pa[:4], pb[:4])  # c_in, a, b, c_out
        if a[0] == "potrf" or a[1] == 1:
            pass  # (tags hash the result ids, which differ between the runs by the id prefix's counter only: checked below)
        else:
            assert np.array_equal(pa[4], pb[4])
    # every task: same status, Ok output, one attempt; every result: same state, device blobs where expected
    tf = [pf._tasks[t] for t in pf.executed]
    tp = [pp._tasks[t] for t in pp.executed]
    for a, b in zip(tf, tp):
        assert (a.status, a.attempts, a.output) == (b.status, b.attempts, b.output) == ("completed", 1, ak.ProcessStatus.Ok)
        assert int(a.task_id.split("-")[1], 16) - f0 == int(b.task_id.split("-")[1], 16) - p0
        assert [int(x.split("-")[1], 16) - f0 for x in a.data_dependencies] == [int(x.split("-")[1], 16) - p0 for x in b.data_dependencies]
    sf = sorted((int(k.split("-")[1], 16) - f0, r.name, r.status, type(r.data).__name__, getattr(r.data, "ptr", None), getattr(r.data, "nbytes", None))
                for k, r in pf._results.items())
    sp = sorted((int(k.split("-")[1], 16) - p0, r.name, r.status, type(r.data).__name__, getattr(r.data, "ptr", None), getattr(r.data, "nbytes", None))
                for k, r in pp._results.items())
    assert sf == sp
    # priorities reached the launches: per wave the chain's update (column k+1) is its own, urgent, group
    urgent = [c for c in bf.calls if c[0] == "tile" and c[4]]
    assert len([c for c in urgent if c[1] == 4]) == N // B - 1 and all(c[4] for c in bf.calls if c[0] == "tile" and c[1] == 1)


@pytest.mark.skipif(ak.FAST is None, reason="the _fastplane extension is not built")
def test_the_tag_of_a_result_id_is_the_same_in_c_and_python(monkeypatch, fake_uploads):
    from dense_linear_app_amd.worker import _tag_of

    r, plane, be, w = _run(512, 128, True, monkeypatch, fake_uploads)
    trsm = [c for c in be.calls if c[0] == "tile" and c[1] == 1]
    assert trsm
    # the TRSM tasks' tag is the hash of the L(k,k) result id named in their payloads
    by_ptr = {r_.data.ptr: k for k, r_ in plane._results.items() if isinstance(r_.data, ak.DeviceBlob)}
    for c in trsm:
        ptr = c[-1]
        for q in range(c[3]):
            assert int(ptr[4, q]) == _tag_of(by_ptr[int(ptr[1, q])])


def _one_wave_plane(fast, monkeypatch, fake_uploads, mutate):
    """A plane with one TRSM task whose payload / dependencies `mutate` has spoiled; -> (status text, result state)"""
    if not fast:
        monkeypatch.setattr(ak, "FAST", None)
    B = 128
    plane = ak.ControlPlane(device_results=True, batch_ready=True)
    w = _W(backend=RecordingBackend())
    w.async_potrf = True
    plane.register_worker("p", w)
    rc, tcl = ak.ResultsClient(plane), ak.TasksClient(plane)
    sid = ak.SessionsClient(plane).create_session(ak.TaskOptions(partition_id="p"), ["p"])
    ids = rc.create_results_metadata(sid, ["L", "A", "short", "output", "payload"])
    rc.upload_result_data(sid, ids["L"], bytes(8 * B * B))
    rc.upload_result_data(sid, ids["A"], bytes(8 * B * B))
    rc.upload_result_data(sid, ids["short"], bytes(8 * B))
    payload, deps = mutate(ids, B)
    rc.upload_result_data(sid, ids["payload"], payload)
    tids = tcl.submit_tasks(sid, [ak.TaskCreation(ids["payload"], [ids["output"]], deps)], ak.TaskOptions(partition_id="p"))
    t = plane._tasks[tids[0]]
    import re

    text = re.sub(r"[0-9a-f]{12}-[0-9a-f]{8}", "<id>", t.output.details()) if t.output else None  # (the runs' id counters differ)
    return t.status, text, plane._results[ids["output"]].status


CASES = {
    "ok": lambda ids, B: (f'{{"op":"TRSM","B":{B},"inL":"{ids["L"]}","inA":"{ids["A"]}"}}', [ids["L"], ids["A"]]),
    "undeclared dependency": lambda ids, B: (f'{{"op":"TRSM","B":{B},"inL":"{ids["L"]}","inA":"{ids["A"]}"}}', [ids["L"]]),
    "short blob": lambda ids, B: (f'{{"op":"TRSM","B":{B},"inL":"{ids["L"]}","inA":"{ids["short"]}"}}', [ids["L"], ids["short"]]),
    "unknown op": lambda ids, B: (f'{{"op":"LU","B":{B},"in":"{ids["A"]}"}}', [ids["A"]]),
    "odd tile": lambda ids, B: (f'{{"op":"TRSM","B":100,"inL":"{ids["L"]}","inA":"{ids["A"]}"}}', [ids["L"], ids["A"]]),
    "not json": lambda ids, B: ("{op:TRSM}", [ids["L"], ids["A"]]),
    "escaped id": lambda ids, B: (f'{{"op":"TRSM","B":{B},"inL":"\\u0041","inA":"{ids["A"]}"}}', [ids["L"], ids["A"]]),
}


@pytest.mark.skipif(ak.FAST is None, reason="the _fastplane extension is not built")
@pytest.mark.parametrize("case", sorted(CASES))
def test_error_paths_report_the_same_in_c_and_python(case, monkeypatch, fake_uploads):
    f = _one_wave_plane(True, monkeypatch, fake_uploads, CASES[case])
    p = _one_wave_plane(False, monkeypatch, fake_uploads, CASES[case])
    assert f == p, (case, f, p)
    if case == "ok":
        assert f == ("completed", "", "completed")
    else:
        assert f[0] == "error" and f[2] == "aborted", f
    if case == "undeclared dependency":
        assert "Missing dependency" in f[1]
    if case == "unknown op":
        assert f[1] == "Unknown op=LU"


@pytest.mark.skipif(ak.FAST is None, reason="the _fastplane extension is not built")
@pytest.mark.parametrize("fast", [True, False])
def test_plane_rules_hold_in_both(fast, monkeypatch):
    if not fast:
        monkeypatch.setattr(ak, "FAST", None)
    plane = ak.ControlPlane()
    rc, tcl = ak.ResultsClient(plane), ak.TasksClient(plane)
    sid = ak.SessionsClient(plane).create_session(ak.TaskOptions(partition_id="p"), ["p"])
    ids = rc.create_results_metadata(sid, ["a", "b", "payload/0"])
    assert len(set(ids.values())) == 3 and all(plane._results[v].status == "created" and plane._results[v].name == k for k, v in ids.items())
    with pytest.raises(KeyError):
        tcl.submit_tasks(sid, [ak.TaskCreation(ids["payload/0"], [ids["a"]], ["no-such-id"])])
    assert not plane._tasks and not plane._pending  # nothing half-submitted
    rc.upload_results_data(sid, {ids["payload/0"]: "x"})
    assert plane._results[ids["payload/0"]].data == b"x" and plane._results[ids["payload/0"]].status == "completed"
    with pytest.raises(RuntimeError, match="write-once"):
        rc.upload_results_data(sid, {ids["payload/0"]: "y"})
    t = tcl.submit_tasks(sid, [ak.TaskCreation(ids["payload/0"], [ids["a"]], [ids["b"]])])
    assert plane._tasks[t[0]].status == "pending" and plane._pending == t  # b has no data yet


def test_task_creations_in_c_are_the_python_ones():
    """client.submit_batch: TaskCreation{payload, [output], sorted unique dependencies} (C2:480-492) built in C."""
    if ak.FAST is None:
        pytest.skip("_fastplane not built")
    items = [("p0", ["b", "a", "c"]), ("p1", ["x"]), ("p2", ["k", "k", "j"]), ("p3", ["z", "y"]), ("p4", [])]
    pids, outs = [f"pid{i}" for i in range(5)], [f"out{i}" for i in range(5)]
    got = ak.FAST.task_creations(ak.TaskCreation, pids, outs, items)
    want = [ak.TaskCreation(p, [o], sorted(set(it[1]))) for p, o, it in zip(pids, outs, items)]
    assert got == want and all(type(g) is ak.TaskCreation for g in got)
    assert items[0][1] == ["b", "a", "c"]  # (the caller's lists are not touched)
    with pytest.raises(ValueError):
        ak.FAST.task_creations(ak.TaskCreation, pids[:2], outs, items)
    with pytest.raises(TypeError):
        ak.FAST.task_creations(ak.TaskCreation, pids[:1], outs[:1], [("p", "not a list")])
