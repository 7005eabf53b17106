"""In-process stand-in for the ArmoniK client / worker SDK surface the reference uses.

The SDK itself is not in the reference tree (it is cloned at image-build time,
Dockerfile.worker.v4:92, Dockerfile.client.param.v1:28); this module restates only
what the reference's call sites need, with the same names, argument meaning and
error behaviour, so `client.py` and `worker.py` read like the reference files:

  client side (client_distrib.cpp)          here
  ----------------------------------------  -----------------------------------------
  SessionsClient.create_session      :353   SessionsClient.create_session
  ResultsClient.create_results_metadata:373 ResultsClient.create_results_metadata
  ResultsClient.upload_result_data   :413   ResultsClient.upload_result_data
  TasksClient.submit_tasks           :498   TasksClient.submit_tasks
  EventsClient.wait_for_result_availability :499   EventsClient.wait_for_result_availability
  TaskCreation{payload_id, expected_output_keys, data_dependencies} :489-492
  TaskOptions (max_duration, max_retries, priority, partition, app) :331-339

  worker side (worker_distrib.cpp)
  ----------------------------------------
  ArmoniKWorker::Execute(TaskHandler&) -> ProcessStatus   :95-99
  TaskHandler.getPayload / getExpectedResults / getDataDependencies / send_result(...).get()
                                                            :105, 180-186, 261
  ProcessStatus::Ok / ProcessStatus(message)               :195, 267

There is no gRPC, no object store and no network: the "control plane" is a Python
object; results are write-once byte blobs held in host memory, tasks run as soon as
their data dependencies are available, one at a time per worker (as an ArmoniK
polling agent drives one Execute at a time, W2:593).
"""
from __future__ import annotations

import itertools
import uuid
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence


# The per-task loops of this module and of worker.ExecuteBatch exist twice: in Python (below) and in C
# (csrc/fastplane.c, on the same objects).  FAST = None forces the Python ones (tests compare the two; a build
# without the extension works, 3-4 us per task slower).
try:
    from . import _fastplane as FAST
except ImportError:  # pragma: no cover - the extension is built by `make -C dense_linear_app_amd/csrc`
    FAST = None

_ID_PREFIX = uuid.uuid4().hex[:12]
_ID_NEXT = [1]


def _reserve_ids(n: int) -> int:
    """-> the first of n consecutive id numbers"""
    start = _ID_NEXT[0]
    _ID_NEXT[0] = start + n
    return start


def _new_id() -> str:
    """An opaque unique id (the ArmoniK server hands out uuids; here: a per-process random prefix and a counter,
    a tenth of the cost of uuid4 -- it is paid three times per task)."""
    return f"{_ID_PREFIX}-{_reserve_ids(1):08x}"


# ------------------------------------------------------------------------------ value types
@dataclass
class TaskOptions:
    """C2:331-339."""
    max_duration_seconds: int = 3600
    max_retries: int = 3
    priority: int = 1
    partition_id: str = ""
    application_name: str = ""
    application_version: str = ""
    application_namespace: str = ""
    options: Dict[str, str] = field(default_factory=dict)

    def copy(self) -> "TaskOptions":
        return TaskOptions(**{**self.__dict__, "options": dict(self.options)})


@dataclass
class TaskCreation:
    """C2:489-492."""
    payload_id: str = ""
    expected_output_keys: List[str] = field(default_factory=list)
    data_dependencies: List[str] = field(default_factory=list)


class ProcessStatus:
    """armonik::api::worker::ProcessStatus: Ok, or an error carrying a message."""
    Ok: "ProcessStatus"

    def __init__(self, details: str = "", _ok: bool = False):
        self._ok = _ok
        self._details = details

    def ok(self) -> bool:
        return self._ok

    def details(self) -> str:
        return self._details

    def __repr__(self) -> str:
        return "ProcessStatus::Ok" if self._ok else f"ProcessStatus({self._details!r})"

    def __eq__(self, other) -> bool:
        return isinstance(other, ProcessStatus) and (self._ok, self._details) == (other._ok, other._details)

    def __hash__(self):
        return hash((self._ok, self._details))


ProcessStatus.Ok = ProcessStatus("", _ok=True)


class DeviceBlob:
    """A result that lives in HBM: same wire format (raw column-major doubles), different address
    space.  With `ControlPlane(device_results=True)` tiles are uploaded once, every task's inputs
    and output stay on the GPU, and only `download_result_data` copies back -- the MI355X answer
    to the reference's per-task download/upload of every tile through the object store.

    A blob is either a tensor of its own or `count` bytes at `offset` of a parent tensor (the one output
    allocation of a grouped launch, worker.ExecuteBatch).  Blobs written by an ASYNCHRONOUS grouped launch
    carry the epoch of that launch: whoever reads them outside the library's stream calls wait() first."""

    pending_epoch = 0  # grouped launches issued on the library's stream so far
    synced_epoch = 0   # ... and how many of them a chol_sync() has since waited for

    __slots__ = ("_parent", "_offset", "nbytes", "ptr", "epoch", "_view")

    def __init__(self, tensor, offset: int = 0, count: Optional[int] = None, epoch: int = 0, base_ptr: Optional[int] = None):
        self._parent = tensor  # 1-D torch tensor on the GPU (owns the memory)
        self._offset = offset  # in bytes
        self.nbytes = tensor.numel() * tensor.element_size() if count is None else count
        # (base_ptr: the parent's address when the caller already has it -- a grouped launch makes hundreds of blobs)
        self.ptr = (tensor.data_ptr() if base_ptr is None else base_ptr) + offset
        self.epoch = epoch
        self._view = tensor if (offset == 0 and count is None) else None

    @property
    def tensor(self):
        if self._view is None:
            es = self._parent.element_size()
            self._view = self._parent[self._offset // es:(self._offset + self.nbytes) // es]
        return self._view

    def __len__(self) -> int:
        return self.nbytes

    def wait(self) -> None:
        """Make the content visible to other streams / the host."""
        if self.epoch > DeviceBlob.synced_epoch:
            from ._lib import lib

            lib().chol_sync()
            DeviceBlob.synced_epoch = DeviceBlob.pending_epoch

    @classmethod
    def from_bytes(cls, data) -> "DeviceBlob":
        import numpy as np
        import torch

        a = np.frombuffer(bytes(data), dtype=np.uint8)
        return cls(torch.from_numpy(a.copy()).cuda())

    def to_bytes(self) -> bytes:
        self.wait()
        return self.tensor.cpu().numpy().tobytes()


class ResultNotAvailable(RuntimeError):
    """wait_for_result_availability on a result that was aborted or can never be produced."""


class _Done:
    """The future send_result returns (the reference awaits it with .get(), W2:261)."""

    def __init__(self, exc: Optional[BaseException] = None):
        self._exc = exc

    def get(self):
        if self._exc is not None:
            raise self._exc


# ------------------------------------------------------------------------------ worker side
class TaskHandler:
    __slots__ = ("_plane", "_session", "_task", "_deps")

    def __init__(self, plane: "ControlPlane", session_id: str, task: "_Task"):
        self._plane, self._session, self._task = plane, session_id, task
        self._deps = None  # built when first asked for (the wave-level path of the worker looks its few ids up itself)

    def getPayload(self) -> str:
        return self.payload_bytes().decode("utf-8")

    def payload_bytes(self) -> bytes:
        """The payload as the bytes it was uploaded as (getPayload without the copy and the decoding)."""
        data = self._plane._results[self._task.payload_id].data
        if isinstance(data, DeviceBlob):
            data = data.to_bytes()
        return data if isinstance(data, bytes) else bytes(data)

    def getExpectedResults(self) -> List[str]:
        return list(self._task.expected_output_keys)

    def getDataDependencies(self) -> Dict[str, bytes]:
        if self._deps is None:
            res = self._plane._results
            self._deps = {rid: res[rid].data for rid in self._task.data_dependencies}
        return self._deps

    def dependency(self, result_id: str):
        """getDataDependencies().get(result_id) without building the map."""
        if self._deps is not None:
            return self._deps.get(result_id)
        return self._plane._results[result_id].data if result_id in self._task.data_dependencies else None

    def first_expected_result(self) -> str:
        return self._task.expected_output_keys[0]

    def priority(self) -> int:
        """getTaskOptions().priority (C2:335) without the copy."""
        return self._task.options.priority

    def getSessionId(self) -> str:
        return self._session

    def getTaskId(self) -> str:
        return self._task.task_id

    def getTaskOptions(self) -> TaskOptions:
        return self._task.options

    def send_result(self, result_id: str, data) -> _Done:
        try:
            if result_id not in self._task.expected_output_keys:
                raise RuntimeError(f"result {result_id} is not an expected output of task {self._task.task_id}")
            self._plane._complete_result(result_id, data)
            return _Done()
        except Exception as e:  # surfaced by .get(), like a failed gRPC upload
            return _Done(e)


class ArmoniKWorker:
    """Base class: subclasses implement Execute(TaskHandler) -> ProcessStatus (W2:95-99)."""

    def Execute(self, taskHandler: TaskHandler) -> ProcessStatus:  # noqa: N802 (reference name)
        raise NotImplementedError


# ------------------------------------------------------------------------------ control plane
class _Result:
    __slots__ = ("result_id", "name", "session_id", "data", "status")

    def __init__(self, result_id: str, name: str, session_id: str):
        self.result_id, self.name, self.session_id = result_id, name, session_id
        self.data = None
        self.status = "created"  # created | completed | aborted


class _Task:
    __slots__ = ("task_id", "session_id", "payload_id", "expected_output_keys", "data_dependencies", "options", "status",
                 "output", "attempts")

    def __init__(self, task_id, session_id, payload_id, expected_output_keys, data_dependencies, options):
        self.task_id, self.session_id, self.payload_id = task_id, session_id, payload_id
        self.expected_output_keys, self.data_dependencies, self.options = expected_output_keys, data_dependencies, options
        self.status = "pending"  # pending | completed | error
        self.output = None
        self.attempts = 0


class ControlPlane:
    """What stands between the four clients and the worker(s)."""

    def __init__(self, device_results: bool = False, batch_ready: bool = False):
        self.device_results = device_results
        # batch_ready: all tasks that are ready together go to the worker's ExecuteBatch in one call (the agent of
        # the reference drives one Execute at a time, W2:593; this is the wave-level execution of SURVEY 8f.3)
        self.batch_ready = batch_ready
        self._sessions: Dict[str, TaskOptions] = {}
        self._results: Dict[str, _Result] = {}
        self._tasks: Dict[str, _Task] = {}
        self._pending: List[str] = []
        self._workers: Dict[str, ArmoniKWorker] = {}
        self._batch_parts: set = set()  # partitions whose worker takes a whole batch (ExecuteBatch)
        self.executed: List[str] = []  # task ids in execution order
        self.on_task_done: Optional[Callable[[_Task], None]] = None

    # -- deployment
    def register_worker(self, partition_id: str, worker: ArmoniKWorker) -> None:
        self._workers[partition_id] = worker
        if hasattr(worker, "ExecuteBatch"):
            self._batch_parts.add(partition_id)
        else:
            self._batch_parts.discard(partition_id)

    # -- internals
    def _complete_result(self, result_id: str, data) -> None:
        r = self._results.get(result_id)
        if r is None:
            raise KeyError(f"unknown result id {result_id}")
        if r.status == "completed":
            raise RuntimeError(f"result {result_id} is write-once and already has data")
        r.data = data if isinstance(data, DeviceBlob) else bytes(data)
        r.status = "completed"

    def _ready(self, t: _Task) -> bool:
        res = self._results
        if res[t.payload_id].status != "completed":
            return False
        for r in t.data_dependencies:
            if res[r].status != "completed":
                return False
        return True

    def _blocked_forever(self, t: _Task) -> bool:
        # (a dependency whose data was deleted can never complete either: delete_results_data)
        need = [t.payload_id] + t.data_dependencies
        return any(self._results[r].status in ("aborted", "deleted") for r in need)

    def _run(self, t: _Task) -> None:
        worker = self._workers.get(t.options.partition_id)
        if worker is None:
            raise RuntimeError(f"no worker registered for partition {t.options.partition_id!r}")
        while True:
            t.attempts += 1
            try:
                status = worker.Execute(TaskHandler(self, t.session_id, t))
            except Exception as e:  # a crashing worker: ArmoniK retries (max_retries, C2:335)
                if t.attempts <= t.options.max_retries:
                    continue
                status = ProcessStatus(f"worker crashed: {e}")
            break
        t.output = status
        missing = [k for k in t.expected_output_keys if self._results[k].status != "completed"]
        if status.ok() and not missing:
            t.status = "completed"
        else:
            t.status = "error"
            if status.ok():
                t.output = ProcessStatus("task returned Ok without producing " + ",".join(missing))
            for k in missing:
                self._results[k].status = "aborted"
        self.executed.append(t.task_id)
        if self.on_task_done:
            self.on_task_done(t)

    def _run_batch(self, tasks: List[_Task]) -> None:
        """Hand every ready task of one partition to its worker at once (wave-level execution).  The
        worker answers with one ProcessStatus per task; bookkeeping per task is that of _run."""
        worker = self._workers.get(tasks[0].options.partition_id)
        try:
            if FAST is not None and hasattr(worker, "ExecuteTasks"):
                statuses = worker.ExecuteTasks(self, tasks)  # (handlers are made for the tasks that need one only)
            else:
                statuses = worker.ExecuteBatch([TaskHandler(self, t.session_id, t) for t in tasks])
            if len(statuses) != len(tasks):
                raise RuntimeError("ExecuteBatch returned a status list of the wrong length")
        except Exception:  # a crashing batch: fall back to the one-task path with its retry rule
            for t in tasks:
                if t.status != "pending":
                    continue
                if all(self._results[k].status == "completed" for k in t.expected_output_keys):
                    # its results were sent before the batch crashed (write-once: it cannot run again); book it as
                    # the one-task path would have, so that flush() and on_task_done still see it
                    t.attempts += 1
                    t.status, t.output = "completed", ProcessStatus.Ok
                    self.executed.append(t.task_id)
                    if self.on_task_done:
                        self.on_task_done(t)
                elif not any(self._results[k].status == "completed" for k in t.expected_output_keys):
                    self._run(t)
                else:  # some of its results exist, some do not: neither re-runnable nor complete
                    t.attempts += 1
                    t.status, t.output = "error", ProcessStatus("batch crashed after a partial send_result")
                    for k in t.expected_output_keys:
                        if self._results[k].status != "completed":
                            self._results[k].status = "aborted"
                    self.executed.append(t.task_id)
                    if self.on_task_done:
                        self.on_task_done(t)
            return
        res, ok_status, executed, done_cb = self._results, ProcessStatus.Ok, self.executed, self.on_task_done
        if FAST is not None and done_cb is None and isinstance(statuses, list):
            # the common case (Ok, the one result there) booked in C; the rest below
            slow = FAST.book_batch(tasks, statuses, res, executed, ok_status)
            tasks, statuses = [tasks[i] for i in slow], [statuses[i] for i in slow]
        for t, status in zip(tasks, statuses):
            t.attempts += 1
            t.output = status
            keys = t.expected_output_keys
            if status is ok_status and len(keys) == 1 and res[keys[0]].status == "completed":  # (the common case, spelled out)
                t.status = "completed"
                executed.append(t.task_id)
                if done_cb:
                    done_cb(t)
                continue
            missing = [k for k in keys if res[k].status != "completed"]
            if status.ok() and not missing:
                t.status = "completed"
            else:
                t.status = "error"
                if status.ok():
                    t.output = ProcessStatus("task returned Ok without producing " + ",".join(missing))
                for k in missing:
                    self._results[k].status = "aborted"
            self.executed.append(t.task_id)
            if self.on_task_done:
                self.on_task_done(t)

    def flush(self) -> None:
        """Wave-level execution may report a task before its kernels have run (worker.async_potrf).  Wait for the
        workers and apply what they found out since: the task that failed gets its real status, its result and
        everything computed from it is aborted -- what the one-task path would have reported at once."""
        failed = {}
        for w in self._workers.values():
            if hasattr(w, "flush"):
                failed.update(dict(w.flush()))
        if not failed:
            return
        aborted = set()
        for tid in self.executed:
            t = self._tasks[tid]
            mine = [k for k in t.expected_output_keys if k in failed]
            if mine:
                t.status, t.output = "error", ProcessStatus(failed[mine[0]])
            elif any(d in aborted for d in [t.payload_id] + t.data_dependencies):
                t.status, t.output = "error", ProcessStatus("a data dependency was aborted")
            else:
                continue
            for k in t.expected_output_keys:
                self._results[k].status = "aborted"
                self._results[k].data = None
                aborted.add(k)

    def flush_quiet(self) -> bool:
        """Wait for the workers' asynchronous launches WITHOUT consuming what they found out (a later flush() still
        reports it).  -> False if a failure is pending."""
        ok = True
        for w in self._workers.values():
            if hasattr(w, "sync"):
                ok = w.sync() and ok
        return ok

    def _pump(self) -> None:
        progressed = True
        while progressed:
            progressed = False
            if self.batch_ready and self._pending:
                by_part: Dict[str, List[_Task]] = {}
                rest = []
                tasks, ready_fn, workers = self._tasks, self._ready, self._workers
                if FAST is not None:
                    by_part, rest = FAST.split_ready(self._pending, tasks, self._results, self._batch_parts)
                else:
                    for tid in self._pending:  # one pass: the ready tasks of partitions with a batch-capable worker leave the list
                        t = tasks[tid]
                        if ready_fn(t) and hasattr(workers.get(t.options.partition_id), "ExecuteBatch"):
                            by_part.setdefault(t.options.partition_id, []).append(t)
                        else:
                            rest.append(tid)
                if by_part:
                    self._pending = rest
                    for ts in by_part.values():
                        self._run_batch(ts)
                    progressed = True
            for tid in list(self._pending):
                t = self._tasks[tid]
                if self._blocked_forever(t):
                    t.status = "error"
                    gone = any(self._results[r].status == "deleted" for r in [t.payload_id] + t.data_dependencies)
                    t.output = ProcessStatus("a data dependency was deleted" if gone else "a data dependency was aborted")
                    for k in t.expected_output_keys:
                        self._results[k].status = "aborted"
                    self._pending.remove(tid)
                    progressed = True
                elif self._ready(t):
                    self._pending.remove(tid)
                    self._run(t)
                    progressed = True


# ------------------------------------------------------------------------------ client side
class SessionsClient:
    def __init__(self, plane: ControlPlane):
        self._plane = plane

    def create_session(self, default_task_option: TaskOptions, partitions: Sequence[str] = ()) -> str:
        sid = _new_id()
        self._plane._sessions[sid] = default_task_option.copy()
        return sid


class ResultsClient:
    def __init__(self, plane: ControlPlane):
        self._plane = plane

    def create_results_metadata(self, session_id: str, names: Iterable[str]) -> Dict[str, str]:
        """name -> fresh result id (C2:373, 471).  Results are write-once."""
        if session_id not in self._plane._sessions:
            raise KeyError(f"unknown session {session_id}")
        if FAST is not None:
            names = names if isinstance(names, (list, tuple)) else list(names)
            return FAST.create_results(self._plane._results, _Result, _ID_PREFIX, _reserve_ids(len(names)), names, session_id)
        out = {}
        for n in names:
            rid = _new_id()
            self._plane._results[rid] = _Result(rid, n, session_id)
            out[n] = rid
        return out

    def upload_result_data(self, session_id: str, result_id: str, data) -> None:
        if isinstance(data, str):
            data = data.encode("utf-8")
        name = self._plane._results[result_id].name if result_id in self._plane._results else ""
        if self._plane.device_results and not name.startswith("payload") and not isinstance(data, DeviceBlob):
            data = DeviceBlob.from_bytes(data)  # tiles go to HBM once; payloads stay on the host
        self._plane._complete_result(result_id, data)
        self._plane._pump()

    def upload_results_data(self, session_id: str, items: Dict[str, object]) -> None:
        """upload_result_data for several results in one call (one gRPC stream in the SDK's bulk uploads): the tasks
        they complete the inputs of are looked at once, at the end."""
        plane = self._plane
        if FAST is not None:
            ids = list(items)
            datas = [v.encode("utf-8") if isinstance(v, str) else v for v in items.values()]
            # (payloads -- host bytes by rule -- in one go; anything else falls through to the general loop untouched)
            if FAST.complete_many(plane._results, ids, datas, "payload" if plane.device_results else None):
                plane._pump()
                return
        for result_id, data in items.items():
            if isinstance(data, str):
                data = data.encode("utf-8")
            name = plane._results[result_id].name if result_id in plane._results else ""
            if plane.device_results and not name.startswith("payload") and not isinstance(data, DeviceBlob):
                data = DeviceBlob.from_bytes(data)
            plane._complete_result(result_id, data)
        plane._pump()

    def delete_results_data(self, session_id: str, result_ids: Iterable[str]) -> None:
        """ResultsClient.delete_results_data of the SDK: the data behind the ids is released, the metadata stays (status
        "deleted": downloading it, or submitting a task that depends on it, fails).  The caller vouches that nothing that
        reads the data is still running -- for device blobs written or read by asynchronous grouped launches that means
        a worker flush / chol_sync first (client.run_cholesky_dag does exactly that)."""
        results = self._plane._results
        for rid in result_ids:
            r = results[rid]
            r.data = None
            r.status = "deleted"

    def download_result_data(self, session_id: str, result_id: str) -> bytes:
        r = self._plane._results[result_id]
        if r.status != "completed":
            raise ResultNotAvailable(f"result {result_id} is {r.status}")
        return r.data.to_bytes() if isinstance(r.data, DeviceBlob) else r.data


class TasksClient:
    def __init__(self, plane: ControlPlane):
        self._plane = plane

    def submit_tasks(self, session_id: str, task_creations: Sequence[TaskCreation],
                     task_options: Optional[TaskOptions] = None) -> List[str]:
        opts = task_options or self._plane._sessions[session_id]
        shared = opts.copy()  # (one private copy per submission: the tasks of a call share their options)
        ids = []
        results, tasks, pending = self._plane._results, self._plane._tasks, self._plane._pending
        if FAST is not None:
            ids = FAST.submit(tasks, pending, results, task_creations, _Task, _ID_PREFIX, _reserve_ids(len(task_creations)), session_id, shared)
            self._plane._pump()
            return ids
        for tc in task_creations:
            if tc.payload_id not in results:
                raise KeyError(f"submit_tasks: unknown result id {tc.payload_id}")
            for rid in tc.expected_output_keys:
                if rid not in results:
                    raise KeyError(f"submit_tasks: unknown result id {rid}")
            for rid in tc.data_dependencies:
                if rid not in results:
                    raise KeyError(f"submit_tasks: unknown result id {rid}")
            tid = _new_id()
            tasks[tid] = _Task(tid, session_id, tc.payload_id, list(tc.expected_output_keys), list(tc.data_dependencies), shared)
            pending.append(tid)
            ids.append(tid)
        self._plane._pump()
        return ids

    def get_task_output(self, task_id: str) -> Optional[ProcessStatus]:
        return self._plane._tasks[task_id].output


class EventsClient:
    def __init__(self, plane: ControlPlane):
        self._plane = plane

    def wait_for_result_availability(self, session_id: str, result_ids: Sequence[str]) -> None:
        self._plane._pump()
        results = self._plane._results
        for rid in result_ids:
            r = results[rid]
            if r.status == "completed":
                continue
            producer = next((t for t in self._plane._tasks.values() if rid in t.expected_output_keys), None)
            why = producer.output.details() if producer and producer.output else "no task produces it yet"
            raise ResultNotAvailable(f"result {rid} is {r.status}: {why}")
