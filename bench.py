#!/usr/bin/env python3
"""Headline benchmark: fp64 TFLOP/s of the N x N SPD tiled Cholesky (N^3/3 flops, the
reference's metric, v6_test.c:60) on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W            (default: N=65536, tile=1024, fp64)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8 ...
  python bench.py --gpus 8 ...      (no launcher: this process starts and supervises the 8 ranks itself and
                                     never touches a GPU; see launch_ranks)

A step is one factorisation of a freshly generated matrix that is already resident in HBM
in tile layout; every step is timed in its own synchronised bracket and the K brackets are
summed, generation happens between them.  The residual of the last step's factor is computed
(untimed) and printed in the line.
The matrix is the same for every GPU count (counter-based generator), so the N-GPU runs
factor the SAME problem: strong scaling.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK = 78.6   # TFLOP/s, MI355X dense fp64 matrix (datasheet)
FP32_PEAK = 157.3  # TFLOP/s, MI355X dense fp32 matrix (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=65536)
    ap.add_argument("--tile", type=int, default=1024)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # the CPU baseline is quoted at N=16384, tile=512 (config 2's shape): ~1.5 s of oracle +
    # ~2 s of vendor LAPACK on 16 cores; --cpu-extra-N adds a second, larger sample to the line
    ap.add_argument("--cpu-N", type=int, default=16384)
    ap.add_argument("--cpu-tile", type=int, default=512)
    ap.add_argument("--cpu-extra-N", type=int, default=20480)
    ap.add_argument("--no-check", action="store_true",
                    help="skip the (untimed) residual of the last step's factor; by default it is in the line")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC pass instead of two child runs under rocprofv3 --pmc")
    ap.add_argument("--no-worker-path", action="store_true",
                    help="skip the task-API leg (the same factorisation submitted task by task through the in-process "
                         "ArmoniK-style client / worker at N=16384, tile 512; ~6 s)")
    ap.add_argument("--worker-path-N", type=int, default=16384)
    ap.add_argument("--worker-path-tile", type=int, default=512)
    ap.add_argument("--stall-timeout", type=float, default=600.0,
                    help="self-launched ranks (--gpus N without a launcher): seconds without a heartbeat from any rank "
                         "after which the run is declared hung, the ranks are killed and an error line is printed")
    return ap.parse_args()


# ---------------------------------------------------------------- heartbeat of a supervised rank
def beat(tag: str) -> None:
    """A rank started by launch_ranks says where it is (one small file per rank; the parent watches the mtimes)."""
    d = os.environ.get("CHOLMI_BENCH_HB")
    if not d:
        return
    try:
        with open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}"), "w") as f:
            f.write(f"{tag} {time.time():.3f}\n")
    except OSError:
        pass


def cpu_baseline(N: int, B: int, seed: int, vendor: bool = True) -> dict:
    """The CPU restatement of the reference path (oracle/, OpenMP over the tiles of a wave)
    timed on this box's host cores on a bounded sample: one factorisation at (N, B)."""
    from oracle import oracle as orc

    # the GPU box's CPU share for one GPU is 16 cores; never more threads than that
    nthreads = min(orc.num_threads(), len(os.sched_getaffinity(0)), int(os.environ.get("CHOLMI_CPU_THREADS", "16")))
    T = orc.plgsy_tiles_lower(N // B, B, float(N), seed)  # the tiles the factorisation reads
    A_lapack = orc.tile_to_lapack(T, N, B)                # (kept for the vendor-LAPACK side line)
    t0 = time.perf_counter()
    info = orc.tiled_potrf(T, N // B, B, nthreads)
    dt = time.perf_counter() - t0
    out = {"value": round(N ** 3 / 3.0 / dt / 1e12, 5), "unit": "TFLOP/s", "cores": nthreads, "kind": "port",
           "sample": f"one full factorisation N={N} tile={B} fp64 plgsy(seed={seed}), {dt:.2f} s, info={info}"}
    if not vendor:
        return out
    # secondary, clearly labelled: the vendor LAPACK that ships with torch, same N, same threads
    try:
        import torch

        torch.set_num_threads(nthreads)
        A = torch.from_numpy(A_lapack)  # lower triangle filled: what torch.linalg.cholesky (upper=False) reads
        t0 = time.perf_counter()
        torch.linalg.cholesky(A)
        dt2 = time.perf_counter() - t0
        out["vendor_lapack"] = {"value": round(N ** 3 / 3.0 / dt2 / 1e12, 5), "unit": "TFLOP/s",
                                "what": f"torch.linalg.cholesky on CPU, N={N}, {nthreads} threads, {dt2:.2f} s"}
    except Exception as e:  # never let the side line break the bench
        out["vendor_lapack"] = {"error": str(e)[:80]}
    return out


def worker_path(N: int, B: int) -> dict:
    """The reference's own submission path (client_distrib.cpp:459-565 -> worker_distrib.cpp:99-564) on the GPU: the
    DAG of POTRF / TRSM / SYRK / GEMM tasks, one payload and one write-once result per task, tiles resident in HBM,
    a wave's ready tasks executed as grouped launches.  Rate = N^3/3 over the time of the wave loop (uploads excluded,
    as the factorisation's timing excludes generation); the factor is checked against the tile matrix it came from."""
    import numpy as np

    from dense_linear_app_amd import client

    rng = np.random.default_rng(7)
    # SPD by dominance, cheap to build: unit-scale noise + N on the diagonal.  Only the tiles on or below the diagonal are
    # ever extracted (C2:407-413) and POTRF / SYRK read the lower triangle only, so the upper tile rows stay zero.
    A = np.zeros((N, N), order="F")
    for j in range(0, N, B):
        A[j:, j:j + B] = rng.uniform(-0.5, 0.5, size=(N - j, min(B, N - j)))
    A[np.diag_indices(N)] += float(N)
    client.run_cholesky_dag(min(N, 4 * B), B, device_results=True, batched=True, A=np.asfortranarray(A[:4 * B, :4 * B]))  # warm
    best = None
    for _ in range(3):
        r = client.run_cholesky_dag(N, B, A=A, device_results=True, batched=True)
        best = r if best is None or r.seconds < best.seconds else best
    n = sum(best.task_counts.values())
    # one tile of the factor against numpy on the host (the full parity of this path is tests/test_gpu_worker.py)
    L00 = np.tril(best.tile(0, 0))
    ref = np.linalg.cholesky(A[:B, :B])
    return {"N": N, "tile": B, "tasks": n, "seconds": round(best.seconds, 4), "tflops": round(N ** 3 / 3.0 / best.seconds / 1e12, 2),
            "us_per_task": round(best.seconds / n * 1e6, 2), "best_of": 3,
            # when the client's last submission returned: the host side of the path (the rest is the GPU catching up)
            "submit_seconds": round(best.submit_seconds, 4), "host_us_per_task": round(best.submit_seconds / n * 1e6, 2),
            "tile00_max_rel_err": float(np.abs(L00 - ref).max() / np.abs(ref).max()),
            "what": "client.run_cholesky_dag(device_results=True, batched=True): wave-level execution behind the task API"}


def live_pmc_traffic(a, budget_s: float = 150.0):
    """HBM bytes per launch of the update kernel, MEASURED for this run's workload: two child runs of this script (one
    factorisation, nothing else) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters
    in KiB, FETCH doubled (MI355X_MICROARCH.md, HBM section).  Children, because the profiler's tool library has to be
    in the process from its start; the program comes straight after `--`.  -> (bytes per launch, launches, note) or
    None when rocprofv3 is not there, a pass fails or the budget runs out (the committed figure is used then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    t_end = time.time() + budget_s
    tot, launches = {}, 0
    tmp = tempfile.mkdtemp(prefix="cholmi_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = t_end - time.time()
            if left < 20:
                return None
            out = os.path.join(tmp, ctr)
            cmd = [rocprof, "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--N", str(a.N), "--tile", str(a.tile), "--dtype", a.dtype, "--seed", str(a.seed), "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline", "--no-worker-path", "--no-check", "--no-live-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left)
            if p.returncode != 0:
                return None
            s, n = 0.0, 0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_trail_update" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        s += float(r["Counter_Value"])
                        n += 1
            if n == 0:
                return None
            tot[ctr], launches = s, n
        # (one un-timed and one bracketed factorisation per child: `launches` counts both)
        return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / launches, launches, \
            "measured in this run: child passes under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB, FETCH doubled per the gfx950 correction)"
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load_pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except Exception:
        return None


def lib_counters() -> int:
    from dense_linear_app_amd._lib import lib

    return lib().chol_debug_device_counters()


def run_single(a) -> dict:
    import torch

    from dense_linear_app_amd import chameleon as ch

    ch.CHAMELEON_Init(1, 1)
    N, B = a.N, a.tile
    dt = ch.ChamRealDouble if a.dtype == "f64" else ch.ChamRealFloat
    d = ch.CHAMELEON_Desc_Create(None, dt, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    probe = ch.mfma_probe(dt, 4)  # the register-only MFMA stream this chip sustains right now (TFLOP/s)
    kernel = ch.update_kernel_name(dt)
    ch.set_profiling(False)
    for w in range(a.warmup):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, a.seed)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        assert info == 0, info
    # timed region: exactly K factorisations, each in its own synchronised bracket; the matrix is
    # regenerated (in HBM, tile layout) BETWEEN the brackets, so nothing but the K factorisations
    # is ever timed.  HIP events around every launch of the dominant kernel (library-side).
    ch.set_profiling(True)
    upd_ms = upd_flops = elapsed = 0.0
    upd_launches = 0
    for s in range(a.steps):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, a.seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)  # synchronous at the ABI (as Chameleon's _Tile calls)
        torch.cuda.synchronize()
        elapsed += time.perf_counter() - t0
        assert info == 0, (s, info)
        st = ch.last_potrf_stats()
        upd_ms += st["update_ms"]
        upd_flops += st["update_flops"]
        upd_launches += st["update_launches"]
    ch.set_profiling(False)
    # one step more, outside the K timed ones, WITHOUT the library's brackets: what the event records and the closing
    # waits of the profiled steps cost (they add no dependency; this puts the number on record)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, a.seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
    torch.cuda.synchronize()
    unprofiled = time.perf_counter() - t0
    assert info == 0, info
    # the factor the last step left in HBM (same matrix, same schedule as the timed ones), checked (untimed)
    res = ch.residual_plgsy(d, float(N), a.seed) if not a.no_check else None
    ch.CHAMELEON_Desc_Destroy(d)
    return {"elapsed": elapsed, "upd_ms": upd_ms, "upd_flops": upd_flops, "upd_launches": upd_launches,
            "regimes": ch.last_potrf_regimes(),  # of the last (un-bracketed) step: the schedule the timed steps ran too
            "residual": res, "probe": probe, "kernel": kernel, "calibration": ch.calibration(),
            "counters": bool(lib_counters()), "unprofiled_ms": unprofiled * 1e3}


def run_multi(a) -> dict:
    """One process per GPU.  The factorisation is ONE call per step -- CHAMELEON_dpotrf_Tile on the
    p x q descriptor: the wave walker of libcholmi with its own RCCL communicators (point-to-point
    groups over xGMI).  torch.distributed (gloo) only bootstraps: it shares the RCCL ids, and carries
    the barriers and the MAX over ranks of the timing.
    CHOLMI_DIST_BACKEND = rccl (default) | gloo (rehearsal: the same walker, tiles moved by gloo, so that
    several ranks can share the GPUs that exist).  If the library's RCCL transport cannot be built the run
    goes on over torch.distributed's own NCCL binding and SAYS SO in the line (exchange.backend, exchange.fallback);
    CHOLMI_DIST_FALLBACK=0 makes that an error instead."""
    import torch
    import torch.distributed as dist

    from dense_linear_app_amd import chameleon as ch
    from dense_linear_app_amd import distributed as dd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    backend = os.environ.get("CHOLMI_DIST_BACKEND", "rccl")
    assert backend in ("rccl", "gloo"), backend
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    beat("init")
    dist.init_process_group("gloo")
    beat("rendezvous")
    grids = dd.candidate_grids(world)
    P, Q = grids[0]
    eng = dd.HipEngine(a.N, a.tile, P, Q, rank, a.dtype, device=local)
    fallback = None
    if backend == "rccl":
        why = ""
        try:
            dd.install_rccl_transport(dist)
            ok = 1
        except Exception as e:
            why = str(e)[:200]
            print(f"[bench] rank {rank}: library RCCL transport unavailable ({why})", file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if os.environ.get("CHOLMI_DIST_FALLBACK", "1") == "0":
                raise SystemExit(f"[bench] rank {rank}: the requested transport (rccl) could not be installed: {why}")
            backend = "torch-nccl"
            fallback = why or "another rank failed to build the RCCL transport"
            tr = dd.TorchTransport(dist, device=local, group=dist.new_group(backend="nccl"))
            tr.install()
    else:
        tr = dd.TorchTransport(dist, device=local)
        tr.install()
    # MEASURE the grid, do not model it: one connecting and one timed factorisation on each candidate (4x2 and 2x4 for
    # eight ranks; new descriptors, the same communicators), the timed steps on the faster.  Both times go into the line.
    grid_probe = None
    if len(grids) > 1:
        grid_probe = {}
        best = None
        for (p2, q2) in grids:
            e2 = eng if (p2, q2) == (P, Q) else dd.HipEngine(a.N, a.tile, p2, q2, rank, a.dtype, device=local)
            tms = 0.0
            for rep_ in range(2):  # the first connects the peers this grid talks to; the second is timed
                e2.generate(float(a.N), a.seed)
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                info = e2.potrf_tile()
                torch.cuda.synchronize()
                dist.barrier()
                tms = (time.perf_counter() - t0) * 1e3
                assert info == 0, info
            tt = torch.tensor([tms], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            grid_probe[f"{p2}x{q2}"] = round(float(tt.item()), 3)
            beat(f"grid{p2}x{q2}")
            if best is None or float(tt.item()) < best[0]:
                if best is not None and best[1] is not eng:
                    best[1].destroy()
                best = (float(tt.item()), e2, p2, q2)
            elif e2 is not eng:
                e2.destroy()
                del e2
        if best[1] is not eng:
            eng.destroy()
        _, eng, P, Q = best
        grid_probe["chosen"] = f"{P}x{Q}"
        torch.cuda.empty_cache()
    factor = eng.potrf_tile
    for w in range(max(1, a.warmup)):  # at least one: RCCL connects its peers on first use
        eng.generate(float(a.N), a.seed)
        info = factor()
        assert info == 0, info
        beat(f"warmup{w}")
    if os.environ.get("CHOLMI_BENCH_KILL_RANK") == str(rank):  # test switch: this rank dies between warm-up and timing
        os._exit(17)
    elapsed = 0.0
    # regenerate between steps, bracket every step by barrier + synchronize and sum the K bracketed times
    for s in range(a.steps):
        eng.generate(float(a.N), a.seed)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        info = factor()
        torch.cuda.synchronize()
        dist.barrier()
        elapsed += time.perf_counter() - t0
        assert info == 0, (s, info)
        beat(f"step{s}")
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    stats = dd.dist_last_stats()
    # one more step, untimed, with the library's HIP-event brackets on: every rank's own update-kernel time and the
    # device time of its whole schedule, so that a scaling curve explains itself (update alone vs what the rank waited)
    ch.set_profiling(True)
    eng.generate(float(a.N), a.seed)
    torch.cuda.synchronize()
    dist.barrier()
    info = factor()
    torch.cuda.synchronize()
    ch.set_profiling(False)
    assert info == 0, info
    st = ch.last_potrf_stats()
    mine = torch.tensor([st["update_ms"], st["total_ms"], float(st["update_launches"]), st["update_flops"]], dtype=torch.float64)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine)
    stats["rank_update_ms"] = [round(float(x[0]), 3) for x in per_rank]
    stats["rank_device_ms"] = [round(float(x[1]), 3) for x in per_rank]
    stats["rank_update_launches"] = [int(x[2]) for x in per_rank]
    stats["rank_update_tflops"] = [round(float(x[3]) / (float(x[0]) * 1e-3) / 1e12, 2) if float(x[0]) > 0 else None for x in per_rank]
    stats["calibration"] = ch.calibration()
    stats["counters"] = bool(lib_counters())
    stats["regimes_rank0"] = ch.last_potrf_regimes()
    stats["grid_probe"] = grid_probe
    beat("profiled")
    # device time of each rank's own schedule in the last step (HIP events on its streams): max and min over ranks --
    # their spread is the block-cyclic imbalance plus what each rank waited for tiles
    dev = torch.tensor([ch.last_potrf_stats()["total_ms"]], dtype=torch.float64)
    dmax, dmin = dev.clone(), dev.clone()
    dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(dmin, op=dist.ReduceOp.MIN)
    stats["device_ms_max"], stats["device_ms_min"] = round(float(dmax.item()), 3), round(float(dmin.item()), 3)
    # the factor of the last timed step, gathered on rank 0 and checked (untimed) against the regenerated matrix
    res = None
    if not a.no_check:
        full = None
        if rank == 0:
            dt = ch.ChamRealDouble if a.dtype == "f64" else ch.ChamRealFloat
            full = ch.CHAMELEON_Desc_Create(None, dt, a.tile, a.tile, a.tile * a.tile, a.N, a.N, 0, 0, a.N, a.N, 1, 1)
        dd.gather_lower(eng.desc, full, 0)
        if rank == 0:
            res = ch.residual_plgsy(full, float(a.N), a.seed)
            ch.CHAMELEON_Desc_Destroy(full)
    out = {"elapsed": float(t.item()), "rank": rank, "grid": f"{P}x{Q}", "residual": res, "dist": stats,
           "backend": backend, "fallback": fallback}
    dist.barrier()
    dist.destroy_process_group()
    return out


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this one (same
    command line; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relay rank 0's JSON line, and
    supervise: a rank that exits non-zero, or --stall-timeout seconds without a heartbeat from any rank, ends the run --
    the remaining ranks are killed (their process groups), ONE line with an "error" field is printed and the exit
    status is non-zero.  This process never imports torch.cuda and never loads libcholmi.so: no GPU call, no exec."""
    n = a.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    hb = tempfile.mkdtemp(prefix="cholmi_bench_hb_")
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                 "CHOLMI_BENCH_HB": hb})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, lines = [], []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True))

    def pump(r, p):  # rank 0's stdout is the bench line; whatever the others print goes to stderr
        for ln in p.stdout:
            if r == 0:
                lines.append(ln)
            else:
                sys.stderr.write(f"[rank {r}] {ln}")
    threads = [threading.Thread(target=pump, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()

    def last_beat() -> float:
        m = 0.0
        for r in range(n):
            try:
                m = max(m, os.path.getmtime(os.path.join(hb, f"rank{r}")))
            except OSError:
                pass
        return m

    def where(r) -> str:
        try:
            return open(os.path.join(hb, f"rank{r}")).read().split()[0]
        except (OSError, IndexError):
            return "not started"

    t_start = time.time()
    error = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, c = bad[0]
            error = {"error": f"rank {r} exited with status {c}", "failed_rank": r, "exit_status": c, "last_seen": where(r)}
            break
        if all(c == 0 for c in codes):
            break
        quiet = time.time() - max(last_beat(), t_start)
        if quiet > a.stall_timeout:
            alive = [r for r, c in enumerate(codes) if c is None]
            error = {"error": f"no rank made progress for {quiet:.0f} s (limit {a.stall_timeout:.0f} s)", "failed_rank": None,
                     "ranks_alive": alive, "last_seen": {str(r): where(r) for r in alive}}
            break
        time.sleep(0.2)
    if error:
        for p in procs:  # the survivors sit in a receive that will never complete: end them, whole process groups
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_kill = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()
    for t in threads:
        t.join(timeout=5)
    if error:  # (the promised single JSON line first: nothing below may replace it with a traceback)
        line = {"metric": "fp64 TFLOP/s for NxN SPD Cholesky (N^3/3 flops / factorisation time)", "value": None,
                "unit": "TFLOP/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup, "launcher": "self"}
        line.update(error)
        print(json.dumps(line), flush=True)
    import shutil

    shutil.rmtree(hb, ignore_errors=True)  # (a rank killed late can still be writing its heartbeat file)
    if error:
        return 3
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            sys.stderr.write(f"[rank 0] {ln}")
    if len(json_lines) != 1:
        print(json.dumps({"error": f"rank 0 printed {len(json_lines)} JSON lines", "n_gpus": n, "value": None}), flush=True)
        return 4
    sys.stdout.write(json_lines[0])
    sys.stdout.flush()
    return 0


def main() -> int:
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 or world > 1:
        r = run_multi(a)
        if r["rank"] != 0:
            return 0
        grid = r["grid"]
    else:
        r = run_single(a)
        grid = "1x1"
    flops = a.N ** 3 / 3.0
    ms_per_step = r["elapsed"] / a.steps * 1e3
    tflops = flops * a.steps / r["elapsed"] / 1e12
    peak = FP64_PEAK if a.dtype == "f64" else FP32_PEAK
    line = {
        "metric": "fp64 TFLOP/s for NxN SPD Cholesky (N^3/3 flops / factorisation time)" if a.dtype == "f64"
        else "fp32 TFLOP/s for NxN SPD Cholesky (N^3/3 flops / factorisation time)",
        "value": round(tflops, 3), "unit": "TFLOP/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"N={a.N} tile={a.tile} {a.dtype} SPD plgsy(bump=N, seed={a.seed}), lower, "
                               f"resident in HBM in Chameleon tile layout", "N": a.N, "tile": a.tile,
                   "grid": grid, "parallelism": f"2D block-cyclic {grid}"},
        "pct_of_mfma_peak": round(100.0 * tflops / (peak * a.gpus), 2),
    }
    if a.gpus == 1:
        if r["upd_ms"] > 0:
            ach = r["upd_flops"] / (r["upd_ms"] * 1e-3) / 1e12
            # no accounting may ever put the kernel above what the matrix cores sustain (the probe, +2 % for clock jitter)
            assert ach <= 1.02 * max(r["probe"], peak), (ach, r["probe"], peak)
            traffic = load_pmc_traffic()
            static = (traffic["hbm_bytes_per_launch"] * traffic["launches"] / (r["upd_launches"] / a.steps)) \
                if traffic and traffic.get("N") == a.N and traffic.get("tile") == a.tile else None
            traffic_source = "profiles/pmc_traffic.json (committed PMC pass, not measured in this run)" if static else None
            live = None if a.no_live_traffic else live_pmc_traffic(a)
            if live:
                static, traffic_source = live[0], live[2]
            line["roofline"] = {
                "bound": "mfma", "kernel": r["kernel"], "achieved": round(ach, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                # the register-only MFMA stream measured in this run: what the chip sustains, beside the datasheet peak
                "peak_probe": round(r["probe"], 2), "frac_of_probe": round(ach / r["probe"], 4) if r["probe"] > 0 else None,
                # HBM bytes per launch of the update kernel: measured by two child passes under rocprofv3 --pmc when they fit
                # the run's budget (live_pmc_traffic), else the committed pass of this workload (profiles/pmc_traffic.json,
                # rescaled to THIS run's launch count) -- traffic_source says which
                "traffic": static, "traffic_source": traffic_source,
                "launches": r["upd_launches"], "avg_launch_ms": round(r["upd_ms"] / max(1, r["upd_launches"]), 4),
                "flops_per_launch": r["upd_flops"] / max(1, r["upd_launches"]),
                # the launches of a wave run side by side: the HIP-event brackets
                # measure the wall time in which they run (the UNION of their intervals), so
                # avg_launch_ms = union / launches; rocprof's per-kernel sum of durations counts the
                # overlap twice (profiles/r02_bench_union_busy.txt: sum / union = 1.24, union = brackets)
                "time_base": "union of concurrent launches",
            }
            c = r["calibration"]
            line["config"]["schedule_calibration"] = {"mfma_probe_tflops": [round(c[0], 2), round(c[2], 2)],
                                                      "diag_step_us": [round(c[1], 1), round(c[3], 1)],
                                                      # False: chol_init's probe left the counter-linked chain off (events only)
                                                      "device_counters": r["counters"],
                                                      "pinned": os.environ.get("CHOLMI_CALIB")}
            # which regime the walker picked for how many of the waves (it picks from the calibration above: two boxes may
            # run two schedules -- this ties the number to the one it ran)
            line["config"]["schedule_regimes"] = r["regimes"]
            # one extra step without the library's HIP-event brackets, beside the mean of the K bracketed ones
            line["unprofiled_ms"] = round(r["unprofiled_ms"], 3)
        if not a.no_worker_path:
            try:
                line["worker_path"] = worker_path(a.worker_path_N, a.worker_path_tile)
            except Exception as e:  # the side leg never breaks the bench line
                line["worker_path"] = {"error": str(e)[:200]}
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.cpu_N, a.cpu_tile, a.seed)
            if a.cpu_extra_N and a.cpu_extra_N != a.cpu_N:
                line["cpu_baseline"]["extra"] = cpu_baseline(a.cpu_extra_N, a.cpu_tile, a.seed, vendor=False)
    if r.get("residual") is not None:
        line["residual"] = r["residual"]
    if r.get("dist"):
        line["config"]["exchange"] = {"backend": r["backend"], "fallback": r.get("fallback"), "host_issue_us_per_wave": round(r["dist"]["issue_us_per_wave"], 1),
                                      "sends_per_rank0": r["dist"]["sends"], "bytes_sent_rank0": r["dist"]["bytes_sent"],
                                      "rank_device_ms_last_step": [r["dist"]["device_ms_min"], r["dist"]["device_ms_max"]],
                                      # from one extra (untimed) step with the library's HIP-event brackets on, rank by rank:
                                      # the update kernel's time (union of its launches), its rate, the rank's whole schedule
                                      "rank_update_ms": r["dist"]["rank_update_ms"], "rank_update_tflops": r["dist"]["rank_update_tflops"],
                                      "rank_update_launches": r["dist"]["rank_update_launches"],
                                      "rank_device_ms_profiled_step": r["dist"]["rank_device_ms"]}
        c = r["dist"]["calibration"]
        line["config"]["schedule_calibration"] = {"mfma_probe_tflops": [round(c[0], 2), round(c[2], 2)],
                                                  "diag_step_us": [round(c[1], 1), round(c[3], 1)],
                                                  "device_counters": r["dist"]["counters"], "pinned": os.environ.get("CHOLMI_CALIB")}
        line["config"]["schedule_regimes"] = r["dist"]["regimes_rank0"]
        if r["dist"].get("grid_probe"):
            # one timed factorisation per candidate grid ahead of the timed steps [ms, max over ranks]; the steps ran on "chosen"
            line["config"]["grid_probe"] = r["dist"]["grid_probe"]
        if os.environ.get("CHOLMI_BENCH_HB"):
            line["launcher"] = "self"
    print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
