#!/usr/bin/env python3
"""Per-task cost of the worker path (ArmoniK-style client -> DagCholeskyWorker -> C ABI) on the GPU."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch, client

ch.CHAMELEON_Init(1, 1)
N, B = int(sys.argv[1]), int(sys.argv[2])
for dev, bat in ((False, False), (True, False), (True, True), (True, True)):
    client.run_cholesky_dag(min(N, 4 * B), B, device_results=dev, batched=bat)  # warm
    t = time.perf_counter()
    r = client.run_cholesky_dag(N, B, device_results=dev, batched=bat)
    dt = time.perf_counter() - t
    n = sum(r.task_counts.values())
    print(f"device_results={dev} batched={bat}: N={N} B={B} {n} tasks, DAG {r.seconds:.3f} s = {r.seconds / n * 1e3:.3f} ms/task "
          f"({N**3 / 3 / r.seconds / 1e12:.3f} TFLOP/s), whole call {dt:.3f} s", flush=True)
A0 = client.enforce_strict_diag_dominance(client.make_spd_like_chameleon(N))
for bat in (False, True):
    pr = cProfile.Profile()
    pr.enable()
    client.run_cholesky_dag(N, B, device_results=True, batched=bat, A=A0)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(26)
    print(f"---- cProfile, device_results=True batched={bat}")
    print(s.getvalue()[:6000])
