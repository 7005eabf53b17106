#!/usr/bin/env python3
"""Time the distributed driver (HipEngine + BlockCyclicCholesky) with the ranks of a P x Q grid
sharing the one GPU of the test box (gloo for the broadcasts): checks the stream schedule of the
multi-GPU path on real kernels and gives its single-GPU overhead against chol_potrf_tile.
  python scripts/dist_single.py N B [world] [mode]"""
import os, socket, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))


def worker(rank, world, port, N, B, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd
    P, Q = dd.grid_for(world)
    eng = dd.HipEngine(N, B, P, Q, rank, "f64", device=0)
    chol = dd.BlockCyclicCholesky(eng, dist, lookahead=True, panel_mode=mode)
    chol.warm_up()
    for it in range(3):
        eng.generate(float(N), 42)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        info = chol.factorize()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            print(f"world={world} {P}x{Q} mode={mode} N={N} B={B} run{it}: {dt*1e3:.1f} ms  {N**3/3/dt/1e12:.2f} TFLOP/s info={info}", flush=True)
    if world == 1:
        from dense_linear_app_amd import chameleon as ch
        print("residual", ch.residual_plgsy(eng.desc, float(N), 42), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    N, B = int(sys.argv[1]), int(sys.argv[2])
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    mode = sys.argv[4] if len(sys.argv) > 4 else "bcast"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    if world == 1:
        worker(0, 1, port, N, B, mode)
    else:
        import torch.multiprocessing as mp
        mp.spawn(worker, args=(world, port, N, B, mode), nprocs=world, join=True)
