#!/usr/bin/env python3
"""A PREDICTED per-rank timeline of chol_potrf_tile on a p x q grid, without a GPU: the launch graphs the walker issues on
every rank (chol_debug_comm_trace: the recording engine of csrc/sched_check.hip -- the same walker, the same regime
choices) replayed with rough kernel durations and a link model (bandwidth per message stream, latency per group).

Model (deliberately crude; its job is to say in which region of (GB/s, us) the first real 8-GPU run should land, so that
config.exchange of that run is read against something):
  * a launch starts when everything it is ordered behind (stream order, events, counters) has finished;
  * update launches (kind U) of one rank share the chip: they run one at a time; the panel chain's kernels (P, T, Y, L)
    run beside them at their calibrated durations;
  * a transport group starts when its stream reaches it and its partners' groups have started (RCCL's rules, as in
    tests/test_schedule_check.py); it then takes latency + max over its peers of (bytes to/from that peer) / bandwidth --
    xGMI is point to point, one link per peer;
  * durations: t_tile = 2 mb^3 / (eff * probe) per tile update at the full-chip rate, t_panel = the measured panel chain
    (diag step x factor x mb/128), both as chol_init measures them; checked below against the MEASURED one-GPU time and
    the measured rank-alone times with communication free (profiles/r04_rank_alone_grids.txt).
usage: predict_scale.py [N tile]   (default 65536 1024)"""
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd._lib import lib  # noqa: E402


def graphs_of(nt, mb, p, q, t_tile, t_panel):
    L = lib()
    out = []
    for rank in range(p * q):
        buf = C.create_string_buffer(256 << 20)
        rc = L.chol_debug_comm_trace(nt, mb, p, q, rank, t_tile, t_panel, buf, len(buf))
        assert rc == 0, (rank, rc)
        ops = []
        for ln in buf.value.decode().splitlines():
            f = ln.split(" ")
            ops.append({"ch": int(f[1]), "kind": f[2] if f[2] != "-" else f[7], "peer": int(f[3]), "bytes": int(f[4]), "group": int(f[5]),
                        "deps": [int(x) for x in f[6].split(",")] if f[6] != "-" else [], "cost": float(f[8])})
        out.append(ops)
    return out


def replay(graphs, bw, lat):
    """-> finish time [s] of every rank (max over its launches); bw [bytes/s] per peer link (None: communication free)"""
    R = len(graphs)
    fin = [[None] * len(g) for g in graphs]
    groups, sends, recvs = defaultdict(list), defaultdict(list), defaultdict(list)
    for r, g in enumerate(graphs):
        for i, op in enumerate(g):
            if op["ch"] >= 0:
                groups[(r, op["ch"], op["group"])].append(i)
                key = (r, op["peer"], op["ch"]) if op["kind"] == "S" else (op["peer"], r, op["ch"])
                (sends if op["kind"] == "S" else recvs)[key].append((r, i))
    partner = {}
    for key in sends:
        for a, b in zip(sends[key], recvs[key]):
            partner[a], partner[b] = b, a
    group_of = {(k[0], i): k for k, m in groups.items() for i in m}
    gstart = {}
    chip_free = [0.0] * R  # when the rank's update launches may next start
    progress = True
    while progress:
        progress = False
        for r, g in enumerate(graphs):
            for i, op in enumerate(g):
                if fin[r][i] is not None or op["ch"] >= 0:
                    continue
                if any(fin[r][d] is None for d in op["deps"]):
                    continue
                t0 = max([fin[r][d] for d in op["deps"]] + [0.0])
                if op["kind"] == "U":
                    t0 = max(t0, chip_free[r])
                    chip_free[r] = t0 + op["cost"]
                fin[r][i] = t0 + op["cost"]
                progress = True
        for key, members in groups.items():
            r = key[0]
            if key in gstart:
                continue
            deps = [d for i in members for d in graphs[r][i]["deps"] if d not in members]
            if any(fin[r][d] is None for d in deps):
                continue
            gstart[key] = max([fin[r][d] for d in deps] + [0.0])
            progress = True
        for key, members in groups.items():
            r = key[0]
            if key not in gstart or fin[r][members[0]] is not None:
                continue
            pk = [group_of[partner[(r, i)]] for i in members]
            if any(k not in gstart for k in pk):
                continue
            t0 = max([gstart[key]] + [gstart[k] for k in pk])
            per_peer = defaultdict(int)
            for i in members:
                per_peer[graphs[r][i]["peer"]] += graphs[r][i]["bytes"]
            dt = 0.0 if bw is None else lat + max(per_peer.values()) / bw
            for i in members:
                fin[r][i] = t0 + dt
            progress = True
    assert all(f is not None for row in fin for f in row), "the replay did not complete"
    return [max(row) if row else 0.0 for row in fin]


def main():
    N, mb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 1024)
    nt = N // mb
    # the calibration of the driver's round-4 box (BENCH_r04.json): probe 76.4 TFLOP/s, diagonal-block step 51.1 us
    probe, diag_us, eff, step_factor = 76.4e12, 51.1e-6, 0.85, 1.0
    t_tile, t_panel = 2.0 * mb ** 3 / (eff * probe), diag_us * step_factor * (mb // 128)
    print(f"# N={N} tile={mb}: t_tile {t_tile * 1e6:.1f} us, t_panel {t_panel * 1e6:.0f} us (probe {probe / 1e12:.1f} TFLOP/s x {eff}, diagonal-block step {diag_us * 1e6:.1f} us)")
    t1 = max(replay(graphs_of(nt, mb, 1, 1, t_tile, t_panel), None, 0.0))
    print(f"# one GPU: predicted {t1 * 1e3:.0f} ms  (measured: 1357-1370 ms at N=65536 / 1024, 27.0-27.6 ms at N=16384 / 512)")
    grids = [(4, 2), (2, 4), (8, 1)]
    for (p, q) in grids:
        g = graphs_of(nt, mb, p, q, t_tile, t_panel)
        free = replay(g, None, 0.0)
        print(f"# {p}x{q}: communication free: slowest rank {max(free) * 1e3:.1f} ms, fastest {min(free) * 1e3:.1f} ms -> at most {t1 / max(free):.2f}x  "
              f"(measured rank-alone device times, N=65536 / 1024: 4x2 194.2, 2x4 185.3, 8x1 219.6 ms)")
        print(f"  {p}x{q}: predicted speed-up over one GPU;  rows: latency per group [us], columns: bandwidth per peer link [GB/s]")
        bws = [20, 30, 40, 60, 80, 120]
        print("      lat\\bw " + " ".join(f"{b:6d}" for b in bws))
        for lat in (10, 25, 50, 100, 200):
            row = []
            for b in bws:
                t = max(replay(g, b * 1e9, lat * 1e-6))
                row.append(t1 / t)
            print(f"      {lat:6d} " + " ".join(f"{x:6.2f}" for x in row))
    print("# north_star asks for >= 6x on 8 GPUs: the cells >= 6.00 above are where that holds in this model.")


if __name__ == "__main__":
    main()
