#!/usr/bin/env python3
"""The task API's rate (bench.worker_path) at the sizes given: worker_rates.py 16384x512 65536x1024 ..."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import bench
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib

ch.CHAMELEON_Init(1, 1)
for cfg in sys.argv[1:]:
    N, B = (int(x) for x in cfg.split("x"))
    s0 = (C.c_longlong * 4)(); lib().chol_batch_stats(s0)
    r = bench.worker_path(N, B)
    s1 = (C.c_longlong * 4)(); lib().chol_batch_stats(s1)
    r["batches_chain_bulk_waits"] = [s1[i] - s0[i] for i in range(3)]
    print(json.dumps(r), flush=True)
