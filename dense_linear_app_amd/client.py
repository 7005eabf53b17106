"""The Cholesky DAG client: cuts an SPD matrix into tiles and drives the wave DAG.

Mirrors the reference client (cholesky_armonik/w_c_cons_v2/client_construction2/client/
src/client_distrib.cpp, "C2"; payload builders from the v1 client, "C1"):

  load_params / parse_int_str            C2:46-93
  make_payload_{potrf,trsm,syrk,gemm}    C1:44-97   (explicit routing -- see below)
  make_spd_like_chameleon                C2:224-252
  enforce_strict_diag_dominance          C2:255-264
  extract_block_from_spd_matrix_colmajor C2:280-309
  block_id_from_ij                       C2:319-321
  submit_one                             C2:459-503
  the wave loop                          C2:506-565

Payload routing: C2's merged make_payload(ids, B) picks TRSM vs SYRK for two-id tasks by
testing whether ids[0] starts with 'L' (C2:175).  Result ids are UUIDs, so that test
never fires and every TRSM is emitted as a SYRK -- the v2 client cannot factor anything.
This client keeps the JSON schema the worker parses (W2:47-69) and routes ops explicitly,
as the v1 builders do.

The input-construction functions are host code inside libcholmi.so (csrc/host_client.cpp,
std::mt19937_64 like the reference) and work without a GPU.
"""
from __future__ import annotations

import json
import os
import sys
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import armonik as ak
from ._lib import lib


# ------------------------------------------------------------------------------ parameters
@dataclass
class Params:
    """C2:41-44."""
    N: int = 12
    B: int = 4


def parse_int_str(s: str, fallback: int, name: str, err=sys.stderr) -> int:
    """C2:46-57: std::stol semantics (leading whitespace and sign allowed, no trailing
    characters), accepted range 1 .. 2^30, anything else -> fallback with a warning."""
    try:
        t = s.lstrip(" \t\n\v\f\r")
        if not t:
            raise ValueError("empty")
        body = t[1:] if t[0] in "+-" else t
        if not body or not body.isascii() or not body.isdigit():
            raise ValueError("trailing chars")
        v = int(t)
        if v <= 0 or v > (1 << 30):
            raise OverflowError("range")
        return v
    except Exception:
        print(f"[CONFIG] Ignoring invalid value for {name}='{s}', using {fallback}", file=err)
        return fallback


USAGE = ("Usage: app [--N=INT] [--B=INT]\n"
         "   or: app N B\n"
         "Also supported via env: CHOLESKY_N / CHOLESKY_B\n")


def load_params(argv: Sequence[str], env: Optional[Dict[str, str]] = None, err=sys.stderr) -> Params:
    """C2:58-93.  `argv` excludes the program name."""
    env = os.environ if env is None else env
    p = Params()
    if "CHOLESKY_N" in env:
        p.N = parse_int_str(env["CHOLESKY_N"], p.N, "CHOLESKY_N", err)
    if "CHOLESKY_B" in env:
        p.B = parse_int_str(env["CHOLESKY_B"], p.B, "CHOLESKY_B", err)
    positional_seen = 0
    for arg in argv:
        if arg in ("-h", "--help"):
            sys.stdout.write(USAGE)
            raise SystemExit(0)
        if arg.startswith("--N="):
            p.N = parse_int_str(arg[4:], p.N, "--N", err)
            continue
        if arg.startswith("--B="):
            p.B = parse_int_str(arg[4:], p.B, "--B", err)
            continue
        if arg and arg[0] != "-":
            if positional_seen == 0:
                p.N = parse_int_str(arg, p.N, "N", err)
                positional_seen += 1
            elif positional_seen == 1:
                p.B = parse_int_str(arg, p.B, "B", err)
                positional_seen += 1
    if p.N <= 0 or p.B <= 0:
        raise ValueError("N and B must be positive")
    return p


# ------------------------------------------------------------------------------ payloads
def _dump(obj: dict) -> str:
    return json.dumps(obj, separators=(",", ":"), ensure_ascii=False)


def make_payload_potrf(id_in: str, B: int) -> str:
    """C1:44-53."""
    return _dump({"op": "POTRF", "B": int(B), "in": id_in})


def make_payload_trsm(id_Lkk: str, id_Aik: str, B: int) -> str:
    """C1:55-67."""
    return _dump({"op": "TRSM", "B": int(B), "inL": id_Lkk, "inA": id_Aik})


def make_payload_syrk(id_Cii: str, id_Aik: str, B: int) -> str:
    """C1:69-81."""
    return _dump({"op": "SYRK", "B": int(B), "inC": id_Cii, "inA": id_Aik})


def make_payload_gemm(id_Cij: str, id_Aik: str, id_Ajk: str, B: int) -> str:
    """C1:83-97."""
    return _dump({"op": "GEMM", "B": int(B), "inC": id_Cij, "inAi": id_Aik, "inAj": id_Ajk})


def make_payload(op: str, ids: Sequence[str], B: int) -> str:
    """C2:165-194 with the operation named by the caller instead of guessed from ids[0][0]."""
    n = {"POTRF": 1, "TRSM": 2, "SYRK": 2, "GEMM": 3}.get(op)
    if n is None or len(ids) != n:
        raise RuntimeError("Nombre d'arguments non supporté")  # C2:190
    return {"POTRF": make_payload_potrf, "TRSM": make_payload_trsm,
            "SYRK": make_payload_syrk, "GEMM": make_payload_gemm}[op](*ids, B)


# ------------------------------------------------------------------------------ input construction
def make_spd_like_chameleon(N: int, bump: float = 100.0, uplo: str = "L", seed: int = 12345,
                            LDA: Optional[int] = None) -> np.ndarray:
    """C2:224-252 (called at C2:404 with bump=100, 'L', seed=12345)."""
    LDA = N if LDA is None else LDA
    assert N >= 0 and LDA >= max(1, N)
    A = np.zeros((LDA, N), dtype=np.float64, order="F")
    lib().chol_make_spd_like_chameleon(A.ctypes.data, N, LDA, float(bump), uplo.encode()[:1], int(seed))
    return A


def enforce_strict_diag_dominance(A: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """C2:255-264, in place on a column-major array."""
    assert A.flags.f_contiguous and A.dtype == np.float64
    lib().chol_enforce_strict_diag_dominance(A.ctypes.data, A.shape[1], A.shape[0], float(eps))
    return A


def extract_block_from_spd_matrix_colmajor(A: np.ndarray, N: int, B: int, bi: int, bj: int) -> np.ndarray:
    """C2:280-309: zero-padded B x B column-major block (bi, bj)."""
    assert A.flags.f_contiguous and A.dtype == np.float64
    blk = np.empty((B, B), dtype=np.float64, order="F")
    lib().chol_extract_block(A.ctypes.data, N, A.shape[0], B, bi, bj, blk.ctypes.data)
    return blk


def block_id_from_ij(i: int, j: int) -> str:
    """C2:319-321."""
    return f"blk/{i}/{j}"


# ------------------------------------------------------------------------------ the DAG driver
@dataclass
class DagResult:
    N: int
    B: int
    Nb: int
    session_id: str
    latest: Dict[str, str]
    task_counts: Dict[str, int]
    seconds: float
    plane: ak.ControlPlane = field(repr=False, default=None)
    submit_seconds: float = 0.0  # wave-level client: when the last submission returned (the rest of `seconds` is the GPU catching up)

    def tile(self, i: int, j: int) -> np.ndarray:
        data = ak.ResultsClient(self.plane).download_result_data(self.session_id, self.latest[block_id_from_ij(i, j)])
        return np.frombuffer(data, dtype=np.float64).reshape((self.B, self.B), order="F")

    def lower_factor(self) -> np.ndarray:
        """tril(L) assembled from the final tile versions.  (The reference client never
        downloads its result -- C2:457 -- this is for validation.)"""
        n = self.Nb * self.B
        L = np.zeros((n, n), order="F")
        for i in range(self.Nb):
            for j in range(i + 1):
                L[i * self.B:(i + 1) * self.B, j * self.B:(j + 1) * self.B] = self.tile(i, j)
        return np.tril(L)[:self.N, :self.N]


BULK_CHUNK = 192  # tasks per submission of a wave's bulk update (wave-level client only)
PARTITION = "cholesky-cpu-vm"  # C2:330 (the name is the reference's; the work runs on the GPU)


def default_task_options() -> ak.TaskOptions:
    """C2:331-339."""
    return ak.TaskOptions(max_duration_seconds=3600, max_retries=3, priority=1, partition_id=PARTITION,
                          application_name="cholesky-dag", application_version="1.0",
                          application_namespace="benchmarks")


def run_cholesky_dag(N: int, B: int, plane: Optional[ak.ControlPlane] = None, worker=None,
                     A: Optional[np.ndarray] = None, verbose: bool = False, log=sys.stdout,
                     device_results: bool = False, batched: bool = False) -> DagResult:
    """C2:325-568 (main) as a function.  `worker` defaults to DagCholeskyWorker on the GPU.
    device_results=True keeps every tile version in HBM between tasks (armonik.DeviceBlob).
    batched=True is the non-blocking client of SURVEY 8f.3: the same tasks, payloads and
    dependencies, but one create_results_metadata + one submit_tasks + one
    wait_for_result_availability per phase of a wave (POTRF; all TRSM; all SYRK/GEMM) instead of
    four blocking calls per task (C2:471-499)."""
    if plane is None:
        # batched: the tasks that become ready together reach the worker in one ExecuteBatch call (wave-level execution)
        plane = ak.ControlPlane(device_results=device_results, batch_ready=batched)
    if worker is None:
        from .worker import DagCholeskyWorker

        worker = DagCholeskyWorker(verbose=verbose, log=log)
    if batched and device_results and hasattr(worker, "async_potrf"):
        worker.async_potrf = True  # every class of a wave enqueued; failures surface at plane.flush() below
    plane.register_worker(PARTITION, worker)
    taskOptions = default_task_options()
    tasksClient, resultsClient = ak.TasksClient(plane), ak.ResultsClient(plane)
    sessionsClient, eventsClient = ak.SessionsClient(plane), ak.EventsClient(plane)

    Nb = (N + B - 1) // B  # C2:351
    session_id = sessionsClient.create_session(taskOptions, [PARTITION])  # C2:353
    keys = [block_id_from_ij(i, j) for i in range(Nb) for j in range(i + 1)]  # C2:367-371
    id_map = resultsClient.create_results_metadata(session_id, keys)  # C2:373

    if A is None:  # C2:402-405
        A = make_spd_like_chameleon(N, 100.0, "L", 12345)
        enforce_strict_diag_dominance(A)
    for i in range(Nb):  # C2:407-413
        for j in range(i + 1):
            block = extract_block_from_spd_matrix_colmajor(A, N, B, i, j)
            resultsClient.upload_result_data(session_id, id_map[block_id_from_ij(i, j)], block.tobytes(order="F"))

    latest = dict(id_map)  # C2:442
    counts = {"POTRF": 0, "TRSM": 0, "SYRK": 0, "GEMM": 0}

    def submit_one(payload_json: str, data_deps: List[str], partition_id: str) -> str:
        """C2:459-503: fresh output+payload ids, upload payload, submit, block on the output."""
        result_ids = resultsClient.create_results_metadata(session_id, ["output", "payload"])
        output_id, payload_id = result_ids["output"], result_ids["payload"]
        deps = sorted(set(data_deps))  # C2:480-481
        resultsClient.upload_result_data(session_id, payload_id, payload_json)
        tc = ak.TaskCreation(payload_id=payload_id, expected_output_keys=[output_id], data_dependencies=deps)
        per_task_opts = taskOptions.copy()
        per_task_opts.partition_id = partition_id
        if verbose:
            print(f"[CLIENT][SUBMIT_ONE] : payload_json = {payload_json}", file=log)
        tasksClient.submit_tasks(session_id, [tc], per_task_opts)
        eventsClient.wait_for_result_availability(session_id, [output_id])
        return output_id

    names_cache: Dict[int, tuple] = {}
    opts_by_priority: Dict[int, ak.TaskOptions] = {}

    def submit_batch(items: List[tuple], priority: int = 1) -> List[str]:
        """items: (payload_json, data_deps).  One metadata call, one submission, one wait.  priority: TaskOptions.priority of
        the submission (C2:335 sets 1 for everything; the wave-level client raises it for the tasks of the panel chain)."""
        if not items:
            return []
        m = len(items)
        names = names_cache.get(m)
        if names is None:
            names = names_cache[m] = ([f"output/{q}" for q in range(m)], [f"payload/{q}" for q in range(m)])
        onames, pnames = names
        ids = resultsClient.create_results_metadata(session_id, onames + pnames)
        outs, pids = [ids[x] for x in onames], [ids[x] for x in pnames]
        resultsClient.upload_results_data(session_id, {pid: it[0] for pid, it in zip(pids, items)})  # one call, not m
        if ak.FAST is not None and type(items) is list:  # (the same objects, built in C: csrc/fastplane.c task_creations)
            tcs = ak.FAST.task_creations(ak.TaskCreation, pids, outs, items)
        else:
            tcs = [ak.TaskCreation(pid, [oid], sorted(set(it[1]))) for pid, oid, it in zip(pids, outs, items)]
        opts = opts_by_priority.get(priority)
        if opts is None:
            opts = opts_by_priority[priority] = taskOptions.copy()
            opts.partition_id = PARTITION
            opts.priority = priority
        tasksClient.submit_tasks(session_id, tcs, opts)
        eventsClient.wait_for_result_availability(session_id, outs)
        return outs

    t0 = time.perf_counter()
    if batched:
        # The wave loop creates ~10 small objects per task and frees none before the end: the cyclic collector's
        # full passes over them (and over whatever an earlier run left alive) cost more than the loop itself --
        # 52 against 121 ms at N=16384, tile 512, measured round 4.  Nothing here forms a cycle; collection is
        # switched off for the loop and restored after.
        import gc

        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return _run_waves_batched(N, B, Nb, session_id, latest, counts, submit_batch, plane, eventsClient, device_results, t0,
                                      resultsClient, worker)
        finally:
            if gc_was_on:
                gc.enable()
    return _run_waves_serial(N, B, Nb, session_id, latest, counts, submit_one, plane, verbose, log, t0)


def _run_waves_batched(N, B, Nb, session_id, latest, counts, submit_batch, plane, eventsClient, device_results, t0,
                       resultsClient=None, worker=None):
    """C2:506-565, one submission per phase of a wave: POTRF; all TRSM; the updates of column k+1 (what
    POTRF(k+1) and TRSM(k+1) depend on); the other updates.  The first three go out with TaskOptions.priority 2 -- the
    tasks of the panel chain -- so that a worker that executes dependency-driven (worker.ExecuteBatch over the
    library's two-stream executor) runs wave k+1's chain beside wave k's bulk update.  Nothing waits for the GPU here:
    a submission returns when its tasks are enqueued, their outputs are ordered by what reads them."""
    if True:
        # (the same payload texts as make_payload -- json.dumps with compact separators -- written directly: the ids are
        # the control plane's own hex ids, nothing in them needs escaping; asserted once below)
        bid = [[block_id_from_ij(i, j) for j in range(i + 1)] for i in range(Nb)]
        assert make_payload("GEMM", ["a", "b", "c"], B) == f'{{"op":"GEMM","B":{int(B)},"inC":"a","inAi":"b","inAj":"c"}}'
        Bi = int(B)
        # Results are write-once, so every task adds a tile version; with the tiles resident in HBM the superseded
        # versions have to go, or N=65536 / tile 1024 would hold 45 760 x 8 MiB.  The versions wave k supersedes are read
        # by wave k's tasks only, so they may go once everything enqueued up to the end of wave k has run: the worker takes
        # a MARK there (chol_batch_mark), and once RETIRE_BYTES of versions have piled up the oldest waves' versions are
        # released behind a wait for THEIR mark (chol_batch_wait) -- the GPU keeps running the waves submitted since; a
        # full drain here (round 4: chol_sync) left it idle for as long as the host needs to submit the next wave.
        from collections import deque

        retire_limit = int(os.environ.get("CHOLESKY_RETIRE_BYTES", str(24 << 30)))
        retire_q, retired_bytes, tile_bytes = deque(), 0, 8 * Bi * Bi
        can_retire = device_results and resultsClient is not None and hasattr(worker, "mark")
        superseded: Dict[int, list] = {0: [latest[bid[i][0]] for i in range(Nb)]} if can_retire else {}
        # Submission order (every task is submitted after the tasks that produce its inputs, as C2:506-565 requires; the
        # order among independent tasks is the client's to choose): POTRF(k+1) and the TRSMs of panel k+1 need column
        # k+1 only, which the chain's update of wave k has just produced -- they go out BEFORE the bulk of wave k's
        # update, so that the GPU runs the next panel while this thread is still writing the (many) payloads of that bulk.
        def chain_of(k):
            """POTRF(k,k) and TRSM(i,k), i > k: -> the outputs of the TRSMs"""
            kk = bid[k][k]
            latest[kk] = submit_batch([(make_payload("POTRF", [latest[kk]], B), [latest[kk]])], 2)[0]
            counts["POTRF"] += 1
            Lkk = latest[kk]
            rows = range(k + 1, Nb)
            ins = [latest[bid[i][k]] for i in rows]
            outs = submit_batch([(f'{{"op":"TRSM","B":{Bi},"inL":"{Lkk}","inA":"{a}"}}', [Lkk, a]) for a in ins], 2)
            for i, o in zip(rows, outs):
                latest[bid[i][k]] = o
            counts["TRSM"] += len(outs)
            return outs

        outs = chain_of(0) if Nb > 0 else []
        for k in range(Nb):
            while can_retire and retired_bytes > retire_limit and retire_q:
                mark, ids, nbytes = retire_q.popleft()
                worker.wait_mark(mark)
                resultsClient.delete_results_data(session_id, ids)
                retired_bytes -= nbytes
            rows = range(k + 1, Nb)
            # the versions this wave supersedes: column k's inputs (gone once the panel has run) and every tile it updates
            gone_now = []
            if can_retire:
                gone_now = [v for v in superseded.pop(k, [])]
            # column k+1 first and apart (the chain's tasks), then -- behind the next panel -- the rest of the wave's update
            k1 = k + 1
            items1, keys1, items, keys = [], [], [], []
            for i, Aik in zip(rows, outs):
                row = bid[i]
                if i > k1:
                    Cij, Ajk = latest[row[k1]], outs[0]
                    items1.append((f'{{"op":"GEMM","B":{Bi},"inC":"{Cij}","inAi":"{Aik}","inAj":"{Ajk}"}}', [Cij, Aik, Ajk]))
                    keys1.append(row[k1])
                for j in range(k + 2, i):
                    Cij, Ajk = latest[row[j]], latest[bid[j][k]]
                    items.append((f'{{"op":"GEMM","B":{Bi},"inC":"{Cij}","inAi":"{Aik}","inAj":"{Ajk}"}}', [Cij, Aik, Ajk]))
                    keys.append(row[j])
                Cii = latest[row[i]]
                (items1 if i == k1 else items).append((f'{{"op":"SYRK","B":{Bi},"inC":"{Cii}","inA":"{Aik}"}}', [Cii, Aik]))
                (keys1 if i == k1 else keys).append(row[i])
                counts["GEMM"] += i - k - 1
                counts["SYRK"] += 1
            if can_retire:
                gone_now.extend(latest[key] for key in keys1)
                gone_now.extend(latest[key] for key in keys)
            for key, o in zip(keys1, submit_batch(items1, 2)):
                latest[key] = o
            nxt = []
            if k1 < Nb:
                if can_retire:  # (the inputs of the next panel: read by its POTRF / TRSMs only, retired with wave k+1)
                    superseded[k1] = [latest[bid[i][k1]] for i in range(k1, Nb)]
                nxt = chain_of(k1)
            # the bulk of a large wave in submissions of BULK_CHUNK tasks: the GPU starts on the first while this thread
            # is still writing the payloads of the next (N=16384, tile 512: +2.5 %, same-box A/B; a wave's launches
            # stay above one full round of workgroups)
            chunk = BULK_CHUNK
            if len(items) > chunk + chunk // 2:
                for c0 in range(0, len(items), chunk):
                    for key, o in zip(keys[c0:c0 + chunk], submit_batch(items[c0:c0 + chunk])):
                        latest[key] = o
            else:
                for key, o in zip(keys, submit_batch(items)):
                    latest[key] = o
            outs = nxt
            if can_retire:
                nbytes = tile_bytes * len(gone_now)
                retire_q.append((worker.mark(), gone_now, nbytes))
                retired_bytes += nbytes
        t_sub = time.perf_counter() - t0
        if device_results:
            # grouped launches are asynchronous on the library's stream: the run ends when they have; a tile that
            # turned out not to be positive definite fails its POTRF task now, and the wait below raises as the
            # blocking client's wait on that task's output does (C2:499)
            plane.flush()
            eventsClient.wait_for_result_availability(session_id, list(latest.values()))
        return DagResult(N, B, Nb, session_id, latest, counts, time.perf_counter() - t0, plane, t_sub)


def _run_waves_serial(N, B, Nb, session_id, latest, counts, submit_one, plane, verbose, log, t0):
    """C2:506-565 as the reference runs it: four blocking calls per task."""
    for k in range(Nb):  # C2:506
        if verbose:
            print(f"Wave k={k}", file=log)
        Lkk_in = latest[block_id_from_ij(k, k)]  # POTRF(k,k), C2:510-523
        latest[block_id_from_ij(k, k)] = submit_one(make_payload("POTRF", [Lkk_in], B), [Lkk_in], PARTITION)
        counts["POTRF"] += 1
        Lkk = latest[block_id_from_ij(k, k)]
        for i in range(k + 1, Nb):  # TRSM(i,k), C2:526-535
            Aik_in = latest[block_id_from_ij(i, k)]
            latest[block_id_from_ij(i, k)] = submit_one(make_payload("TRSM", [Lkk, Aik_in], B), [Lkk, Aik_in], PARTITION)
            counts["TRSM"] += 1
        for i in range(k + 1, Nb):  # updates, C2:540-560
            Aik_in = latest[block_id_from_ij(i, k)]
            for j in range(k + 1, i + 1):
                if i == j:
                    Cii_in = latest[block_id_from_ij(i, i)]
                    latest[block_id_from_ij(i, i)] = submit_one(
                        make_payload("SYRK", [Cii_in, Aik_in], B), [Cii_in, Aik_in], PARTITION)
                    counts["SYRK"] += 1
                else:
                    Cij_in = latest[block_id_from_ij(i, j)]
                    Ajk_in = latest[block_id_from_ij(j, k)]
                    latest[block_id_from_ij(i, j)] = submit_one(
                        make_payload("GEMM", [Cij_in, Aik_in, Ajk_in], B), [Cij_in, Aik_in, Ajk_in], PARTITION)
                    counts["GEMM"] += 1
        if verbose:
            print(f"Wave k={k} done.", file=log)
    secs = time.perf_counter() - t0
    if verbose:
        print("All waves completed.", file=log)
    return DagResult(N, B, Nb, session_id, latest, counts, secs, plane)


def main(argv: Optional[Sequence[str]] = None) -> int:
    """The reference client's CLI (C2:58-93): app [--N=INT] [--B=INT] | app N B."""
    from . import chameleon as ch
    from .worker import env_int

    P = load_params(sys.argv[1:] if argv is None else argv)
    ch.CHAMELEON_Init(env_int("CHM_NCPU", os.cpu_count() or 1), env_int("CHM_NGPU", 1))
    res = run_cholesky_dag(P.N, P.B)
    L = res.lower_factor()
    A = enforce_strict_diag_dominance(make_spd_like_chameleon(P.N))[:P.N, :P.N]
    r = np.linalg.norm(L @ L.T - A) / np.linalg.norm(A)
    print(f"All waves completed. N={P.N} B={P.B} tasks={res.task_counts} time={res.seconds:.3f}s "
          f"residual={r:.3e}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
