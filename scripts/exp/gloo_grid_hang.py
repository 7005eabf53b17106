#!/usr/bin/env python3
"""Reproducer harness: chol_potrf_tile on a P x Q descriptor, ranks as processes sharing cuda:0, gloo transport.
   python scripts/exp/gloo_grid_hang.py P Q [N B]     (stack dumps of every rank after 90 s without progress)"""
import faulthandler, os, socket, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def worker(rank, world, port, P, Q, N, B):
    sys.path.insert(0, ROOT)
    faulthandler.dump_traceback_later(90, exit=True)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd
    eng = dd.HipEngine(N, B, P, Q, rank, "f64", device=0)
    eng.generate(float(N), 42)
    tr = dd.TorchTransport(dist, device=0)
    tr.trace = os.environ.get("TRACE") == "1"
    tr.install()
    t0 = time.time()
    info = eng.potrf_tile()
    print(f"rank {rank}: info {info} in {time.time() - t0:.2f} s, stats {dd.dist_last_stats()}", flush=True)
    from dense_linear_app_amd import chameleon as ch
    full = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1) if rank == 0 else None
    dd.gather_lower(eng.desc, full, 0)
    if rank == 0:
        print("residual", ch.residual_plgsy(full, float(N), 42), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    P, Q = int(sys.argv[1]), int(sys.argv[2])
    N, B = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2304, 256)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, P * Q, port, P, Q, N, B)) for r in range(P * Q)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=200)
    for p in procs:
        if p.is_alive():
            p.kill()
    print("exit codes", [p.exitcode for p in procs], flush=True)
