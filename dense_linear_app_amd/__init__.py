"""MI355X-native tiled Cholesky behind the reference's ArmoniK-style task/worker API.

Layers (reference file each one mirrors, relative to /root/reference):
  chameleon.py   ctypes binding of libcholmi.so, named after the Chameleon calls the
                 reference makes (worker_distrib.cpp:41,78,238,323,416,511; v6_test.c:41-56)
  armonik.py     in-process stand-in for the ArmoniK client/worker SDK surface the
                 reference uses (client_distrib.cpp:341-344,353,373,413,489-499;
                 worker_distrib.cpp:95-99,105,180-186,261)
  worker.py      DagCholeskyWorker.Execute (worker_distrib.cpp:99-564)
  client.py      the DAG driver (client_distrib.cpp:58-93,165-194,224-321,459-565; v1 payload
                 builders client_distrib.cpp(v1):44-97)
  driver.py      whole-matrix resident driver (v6_test.c:7-95) incl. the bench protocol
  distributed.py 2D block-cyclic multi-GPU factorisation over torch.distributed (RCCL)

All arithmetic happens in libcholmi.so (hand-written gfx950 kernels).  There is no CPU
fallback: importing works anywhere, but chameleon.init() raises without a GPU.
"""
__version__ = "0.1.0"
