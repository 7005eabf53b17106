#!/usr/bin/env python3
"""Host time of the wave-level task path WITHOUT a GPU: the grouped launches and the allocations are stubbed out (a
diagnostic backend, this script only), everything else -- client, control plane, worker, payload reader -- is the
product's.  usage: worker_dryrun.py N B [profile]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
from dense_linear_app_amd import armonik as ak, client
from dense_linear_app_amd.worker import DagCholeskyWorker, HipTileBackend


class _Fake:
    def __init__(self, nbytes, ptr): self.n, self.p = nbytes, ptr
    def numel(self): return self.n
    def element_size(self): return 1
    def data_ptr(self): return self.p


class NullBatchBackend(HipTileBackend):
    next_ptr = 1 << 30
    def sync_inputs(self): pass
    def batch_alloc(self, m, B):
        NullBatchBackend.next_ptr += m * B * B * 8
        return _Fake(m * B * B * 8, NullBatchBackend.next_ptr), NullBatchBackend.next_ptr
    def tile_batch(self, code, B, m, ptr, urgent=False): return 0
    def potrf_batch(self, B, m, ptr, slots): return 0


def fake_from_bytes(cls, data):
    NullBatchBackend.next_ptr += len(data)
    return cls(_Fake(len(data), NullBatchBackend.next_ptr))


ak.DeviceBlob.from_bytes = classmethod(fake_from_bytes)
N, B = int(sys.argv[1]), int(sys.argv[2])
A = np.zeros((N, N), order="F")


class W(DagCholeskyWorker):
    def flush(self): self._deferred = []; return []


def run():
    plane = ak.ControlPlane(device_results=True, batch_ready=True)
    w = W(backend=NullBatchBackend())
    plane.flush = lambda: None
    return client.run_cholesky_dag(N, B, plane=plane, worker=w, A=A, device_results=True, batched=True)


run()
t = time.perf_counter(); r = run(); dt = time.perf_counter() - t
n = sum(r.task_counts.values())
print(f"N={N} B={B}: {n} tasks, DAG loop {r.seconds * 1e3:.1f} ms = {r.seconds / n * 1e6:.2f} us/task (whole call {dt:.3f} s)")
if len(sys.argv) > 3:
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
