"""The wave walker's dependencies, checked without a GPU (csrc/sched_check.hip, chol_debug_schedule_check): the walker
runs over an engine that records every launch with the tiles it reads and writes, every event record / wait and every
counter edge, and any two launches that touch the same tile (or block-inverse workspace), one of them writing, must be
ordered.  The schedule is picked per wave from measured speeds and eight threshold switches:
here every regime is forced at many sizes.  A missing dependency shows on the GPU as a rare wrong digit at best; here
it is a deterministic finding that names the two launches."""
import ctypes as C
import re

import pytest

from dense_linear_app_amd._lib import lib

T512, P512 = 3.8e-6, 700e-6  # seconds per tile update / per panel chain, about what chol_init measures at tile 512


def check(nt, mb, t_tile, t_panel, profiling=0, grid=(1, 1), rank=0):
    buf = C.create_string_buffer(1 << 16)
    n = lib().chol_debug_schedule_check_grid(nt, mb, grid[0], grid[1], rank, t_tile, t_panel, profiling, buf, len(buf))
    rep = buf.value.decode()
    m = re.search(r"(\d+) launches, (\d+) event waits, (\d+) counter edges, (\d+) flow-form waves, (\d+) findings", rep)
    assert m, rep
    return n, rep, [int(x) for x in m.groups()]


SWITCHES = [
    {},
    {"CHOLMI_PAIR_FACTOR": "0"}, {"CHOLMI_PAIR_FACTOR": "1000"},
    {"CHOLMI_PAIR_FACTOR": "0", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PIPE_FACTOR": "0"},
    {"CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000"}, {"CHOLMI_PIPE_FACTOR": "0.02"}, {"CHOLMI_PIPE_FACTOR": "0"},
    {"CHOLMI_PIPE_FACTOR": "0", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PAIR_FACTOR": "1000"},
    {"CHOLMI_HALVES_MAX_ROUNDS": "0", "CHOLMI_PIPE_FACTOR": "0"}, {"CHOLMI_YIELD_FACTOR": "0"}, {"CHOLMI_YIELD_FACTOR": "1000"},
    # the near column, the latency form of column k+1, the flow form late / never / always
    {"CHOLMI_NEAR_FACTOR": "0"}, {"CHOLMI_U1_SMALL": "0"}, {"CHOLMI_NEAR_FACTOR": "0", "CHOLMI_U1_SMALL": "0"},
    {"CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000", "CHOLMI_NEAR_FACTOR": "100", "CHOLMI_U1_SMALL": "64"},
    {"CHOLMI_PIPE_FACTOR": "0.3", "CHOLMI_NEAR_FACTOR": "100", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PAIR_FACTOR": "1000"},
    {"CHOLMI_NEAR_FACTOR": "0.3"},
    {"CHOLMI_FLOW_FACTOR": "100", "CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000"}, {"CHOLMI_FLOW_FACTOR": "0.05"},
    {"CHOLMI_FLOW_RUN_FACTOR": "0"}, {"CHOLMI_FLOW_RUN_FACTOR": "100"},
    # flow-form waves BEHIND waves that are not counter-linked (pairs everywhere they fit, the flow wherever it fits): the
    # row-slab kernel has no counter to poll there and must join the POTRF stream by its event (advisor, round 4)
    {"CHOLMI_PAIR_FACTOR": "0", "CHOLMI_FLOW_FACTOR": "0.1"}, {"CHOLMI_PAIR_FACTOR": "0", "CHOLMI_FLOW_FACTOR": "100", "CHOLMI_PIPE_FACTOR": "100"},
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()) or "default")
def test_every_conflicting_pair_of_launches_is_ordered(env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    launches = 0
    for mb, t_tile, t_panel in ((512, T512, P512), (384, 1.6e-6, 570e-6), (256, 0.5e-6, 350e-6), (1024, 30e-6, 1400e-6), (128, 0.06e-6, 180e-6)):
        for nt in (1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 24, 33, 48):
            for scale in (1.0, 0.05, 20.0):  # the same matrix on a chip whose panel chain is 20x faster / slower against its update
                for prof in (0, 1):
                    n, rep, cnt = check(nt, mb, t_tile, t_panel * scale, prof)
                    assert n == 0, (env, nt, mb, scale, prof, rep)
                    launches += cnt[0]
    assert launches > 10000


def test_the_regimes_are_really_entered():
    """Counts that tell the forced regimes apart (otherwise the sweep above would pass on one schedule)."""
    _, _, d = check(8, 512, T512, P512)           # N = 4096 at tile 512: chain-bound from wave 0 on -> flow form, counters
    assert d[3] == 7 and d[2] >= 7, d
    _, _, d = check(32, 512, T512, P512)          # N = 16384: pairs, halves, then counter-linked waves, no flow form
    assert d[3] == 0 and d[2] >= 5, d
    _, _, e = check(32, 512, T512, P512 * 1e-3)   # a panel chain that costs nothing: nothing but pairs / plain waves, no counters
    assert e[2] == 0 and e[0] < d[0], (d, e)
    _, _, d = check(8, 1024, 30e-6, 1400e-6)      # eight blocks per tile: never the flow form
    assert d[3] == 0, d


@pytest.mark.parametrize("nt,mb,t_tile,t_panel", [(12, 512, T512, P512), (20, 512, T512, P512), (24, 512, T512, P512 * 0.2), (16, 256, 0.5e-6, 350e-6)])
def test_the_checker_sees_a_missing_dependency(nt, mb, t_tile, t_panel, monkeypatch):
    """Self-test by mutation: the checker ignores ONE dependency the walker asked for and must then report a conflict --
    for every counter edge, and for the event waits that are not implied by others (measured: the start / join waits of
    every stream, and per wave the waits whose producer the consumer already follows through a third launch)."""
    n, rep, cnt = check(nt, mb, t_tile, t_panel)
    assert n == 0, rep
    nwaits, ngates = cnt[1], cnt[2]
    for i in range(ngates):
        monkeypatch.setenv("CHOLMI_CHECK_DROP_GATE", str(i))
        n, rep, _ = check(nt, mb, t_tile, t_panel)
        assert n > 0, (i, rep)
    monkeypatch.delenv("CHOLMI_CHECK_DROP_GATE", raising=False)
    seen = 0
    for i in range(nwaits):
        monkeypatch.setenv("CHOLMI_CHECK_DROP_WAIT", str(i))
        n, rep, _ = check(nt, mb, t_tile, t_panel)
        seen += n > 0
    assert seen >= 0.4 * nwaits, (seen, nwaits)


GRID_SWITCHES = [{}, {"CHOLMI_PAIR_FACTOR": "0"}, {"CHOLMI_PAIR_FACTOR": "1000"},
                 {"CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000"}, {"CHOLMI_PIPE_FACTOR": "0"},
                 {"CHOLMI_PAIR_FACTOR": "0", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PIPE_FACTOR": "0"}, {"CHOLMI_YIELD_FACTOR": "1000"}]


@pytest.mark.parametrize("env", GRID_SWITCHES, ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()) or "default")
def test_every_rank_of_a_grid_orders_its_launches_sends_and_receives(env, monkeypatch):
    """Per rank of a p x q grid: a send reads and a receive writes on the stream it is issued on -- tiles, the diagonal /
    head tile buffers (by wave parity) and the panel buffers (by wave mod 4): the rotating buffers' reuse is exactly what a
    timing-dependent test cannot pin down."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    checks = 0
    for grid in ((2, 1), (1, 2), (2, 2), (4, 2), (2, 4), (3, 2), (4, 1), (3, 3)):
        for nt in (1, 2, 3, 5, 8, 13, 24, 40):
            for mb, t_tile, t_panel in ((512, T512, P512), (1024, 30e-6, 1400e-6), (128, 0.06e-6, 180e-6)):
                for scale in (1.0, 0.05, 20.0):
                    for rank in range(grid[0] * grid[1]):
                        n, rep, _ = check(nt, mb, t_tile, t_panel * scale, rank & 1, grid, rank)
                        assert n == 0, (env, grid, rank, nt, mb, scale, rep)
                        checks += 1
    assert checks > 3000


@pytest.mark.parametrize("grid,rank", [((2, 2), 0), ((2, 2), 3), ((4, 2), 3), ((2, 1), 1)])
def test_the_checker_sees_a_missing_dependency_on_a_grid(grid, rank, monkeypatch):
    n, rep, cnt = check(16, 512, T512, P512, 0, grid, rank)
    assert n == 0, rep
    seen = 0
    for i in range(cnt[1]):
        monkeypatch.setenv("CHOLMI_CHECK_DROP_WAIT", str(i))
        n, rep, _ = check(16, 512, T512, P512, 0, grid, rank)
        seen += n > 0
    # (the rest are implied by other paths: the start / join waits, the panel buffers' last readers -- wave k's panel needs every
    # earlier update of its column anyway -- and the waits that keep POTRF(k+2) off block inverses that are still travelling)
    assert seen >= 0.4 * cnt[1], (seen, cnt[1])


def test_a_flow_wave_behind_a_paired_wave_is_ordered_by_its_join_event(monkeypatch):
    """Advisor's finding of round 4 (walker.h: join_flow): PAIR_FACTOR=0 + FLOW_FACTOR=0.1 at nt = 7, tile 512 -- wave 0 is
    counter-linked and pre-joins the flow stream, waves 1-4 go in pairs, wave 5 is a flow wave with no counter to poll.  Its
    row-slab kernel (ST_CX) reads rows of tile (5,5) that the paired updates wrote: the join event is the only thing that
    orders it, and the checker -- which now records that kernel as a launch of its own on ST_CX -- must say so when the
    join is dropped."""
    monkeypatch.setenv("CHOLMI_PAIR_FACTOR", "0")
    monkeypatch.setenv("CHOLMI_FLOW_FACTOR", "0.1")
    n, rep, cnt = check(7, 512, T512, P512)
    assert n == 0 and cnt[3] >= 1, rep
    joins = []
    for i in range(cnt[1]):
        monkeypatch.setenv("CHOLMI_CHECK_DROP_WAIT", str(i))
        n, rep, _ = check(7, 512, T512, P512)
        if "flow stream's join" in rep:
            joins.append((i, n, rep))
    assert joins, "no flow wave joined its stream by an event in this schedule"
    assert all(n > 0 and "row slabs" in rep for _, n, rep in joins), joins


# ------------------------------------------------------------------------------------------------------------------
# The TASK executor (chol_tile_batch / chol_potrf_batch: two streams, per-tile event ordering, the pending POTRF that
# goes out pipelined with its panel's TRSM batch) -- checked through the product's own client, control plane and worker:
# only the tile allocations are stand-ins (addresses that are never dereferenced), the library runs in recording mode.
def _task_path_record(N, B, mutate=-1, retire_bytes=None, monkeypatch=None):
    import ctypes as C

    import numpy as np

    from dense_linear_app_amd import armonik as ak, client
    from dense_linear_app_amd._lib import lib
    from dense_linear_app_amd.worker import DagCholeskyWorker, HipTileBackend

    class _Fake:
        def __init__(self, nbytes, ptr): self.n, self.p = nbytes, ptr
        def numel(self): return self.n
        def element_size(self): return 1
        def data_ptr(self): return self.p

    class Backend(HipTileBackend):  # (tile_batch / potrf_batch are the product's: they call the library)
        nxt = 1 << 30
        def sync_inputs(self): pass
        def batch_alloc(self, m, B):
            base = Backend.nxt  # (fresh addresses, never reused: results are write-once)
            Backend.nxt += m * B * B * 8
            return _Fake(m * B * B * 8, base), base

    def fake_from_bytes(cls, data):
        base = Backend.nxt
        Backend.nxt += len(data)
        return cls(_Fake(len(data), base))

    L = lib()
    assert L.chol_debug_task_record(1, mutate) == 0
    try:
        monkeypatch.setattr(ak.DeviceBlob, "from_bytes", classmethod(fake_from_bytes))
        if retire_bytes is not None:
            monkeypatch.setenv("CHOLESKY_RETIRE_BYTES", str(retire_bytes))
        plane = ak.ControlPlane(device_results=True, batch_ready=True)
        w = DagCholeskyWorker(backend=Backend())
        r = client.run_cholesky_dag(N, B, plane=plane, worker=w, A=np.zeros((N, N), order="F"), device_results=True, batched=True)
        out = (C.c_longlong * 5)()
        assert L.chol_debug_task_check(out) == 0
        return r, list(out), L.chol_last_error().decode()
    finally:
        L.chol_debug_task_record(0, -1)


@pytest.mark.parametrize("N,B", [(1024, 128), (2560, 256), (4096, 512), (6144, 512), (24 * 128, 128)])
def test_the_task_executor_orders_every_conflicting_pair_of_batches(N, B, monkeypatch):
    r, (batches, waits, pairs, unordered, hosted), msg = _task_path_record(N, B, monkeypatch=monkeypatch)
    nb = N // B
    assert sum(r.task_counts.values()) == nb + nb * (nb - 1) // 2 + nb * (nb - 1) // 2 + nb * (nb - 1) * (nb - 2) // 6
    # POTRF + its TRSM batch go out as one pipelined issue that the tracker books as two batches of the chain stream
    assert batches >= 3 * (nb - 1) and waits > 0 and pairs > batches
    assert unordered == 0, msg


def test_the_task_executor_check_holds_with_versions_retired_behind_marks(monkeypatch):
    """Superseded tile versions released behind chol_batch_wait (client: CHOLESKY_RETIRE_BYTES): the host-side waits are
    part of the order (their addresses are not reused by the stand-in allocator, so this exercises marks and waits)."""
    r, (batches, waits, pairs, unordered, hosted), msg = _task_path_record(4096, 256, retire_bytes=8 << 20, monkeypatch=monkeypatch)
    assert unordered == 0, msg


@pytest.mark.parametrize("drop", [0, 1, 2, 5, 9])
def test_the_task_executor_check_sees_a_missing_event_wait(drop, monkeypatch):
    """Self-test by mutation: with ONE cross-stream event wait dropped the checker must report an unordered pair."""
    _, (batches, waits, pairs, unordered, hosted), msg = _task_path_record(4096, 512, mutate=drop, monkeypatch=monkeypatch)
    assert unordered > 0 and "without being ordered" in msg, (drop, waits, msg)


# ------------------------------------------------------------------------------------------------------------------
# The two-communicator transport of a p x q grid, replayed for ALL ranks together under RCCL's rules -- no GPU, no RCCL.
# What the transport needs to be safe (csrc/walker.h): the operations of one communicator run in the
# order they were issued on its stream; a group is one fused operation that starts when its stream reaches it and
# ends when every send and receive in it has met its partner, which must have started too; the k-th send from a to b
# on a channel pairs with the k-th receive b posts from a on that channel.  The host never waits between groups, so
# only these stream-side rules can block.  The replay takes every rank's launch graph from the recording engine
# (chol_debug_comm_trace: every kernel launch and transport call with the launches it is ordered behind) and runs them
# to completion: a cycle -- rank a's group waiting for a receive that rank b would only post behind a group that waits
# for a -- shows as "no progress".
def _launch_graphs(nt, mb, p, q, t_tile, t_panel):
    import ctypes as C

    from dense_linear_app_amd._lib import lib

    L = lib()
    graphs = []
    for rank in range(p * q):
        buf = C.create_string_buffer(64 << 20)
        rc = L.chol_debug_comm_trace(nt, mb, p, q, rank, t_tile, t_panel, buf, len(buf))
        assert rc == 0, (rank, rc)
        ops = []
        for ln in buf.value.decode().splitlines():
            f = ln.split(" ")
            deps = [int(x) for x in f[6].split(",")] if f[6] != "-" else []
            ops.append({"ch": int(f[1]), "kind": f[2], "peer": int(f[3]), "bytes": int(f[4]), "group": int(f[5]), "deps": deps})
        graphs.append(ops)
    return graphs


def _replay(graphs):
    """-> (everything completed?, description of what is stuck)"""
    from collections import defaultdict, deque

    R = len(graphs)
    done = [[False] * len(g) for g in graphs]
    # groups: (rank, channel, group) -> member op ids; FIFO pairing per (src, dst, channel)
    groups = defaultdict(list)
    sends, recvs = defaultdict(list), defaultdict(list)
    for r, g in enumerate(graphs):
        for i, op in enumerate(g):
            if op["ch"] >= 0:
                groups[(r, op["ch"], op["group"])].append(i)
                (sends if op["kind"] == "S" else recvs)[(r, op["peer"], op["ch"]) if op["kind"] == "S" else (op["peer"], r, op["ch"])].append((r, i))
    partner = {}
    for key in set(sends) | set(recvs):
        a, b = sends.get(key, []), recvs.get(key, [])
        assert len(a) == len(b), f"rank {key[0]} sends {len(a)} messages to rank {key[1]} on channel {key[2]}, which posts {len(b)} receives"
        for (rs, i), (rr, j) in zip(a, b):
            assert graphs[rs][i]["bytes"] == graphs[rr][j]["bytes"], (key, graphs[rs][i], graphs[rr][j])
            partner[(rs, i)], partner[(rr, j)] = (rr, j), (rs, i)
    started = set()  # groups whose stream has reached them
    group_of = {(r, i): (r, graphs[r][i]["ch"], graphs[r][i]["group"]) for (r, c, gidx), m in groups.items() for i in m}
    progress = True
    while progress:
        progress = False
        for r, g in enumerate(graphs):
            for i, op in enumerate(g):
                if done[r][i] or op["ch"] >= 0:
                    continue
                if all(done[r][d] for d in op["deps"]):
                    done[r][i] = True
                    progress = True
        for key, members in groups.items():
            r = key[0]
            if key not in started:
                # (a member's dependencies are the group's: the stream's earlier work and what the stream was told to wait
                # for; members of the same group are not each other's prerequisites)
                if all(done[r][d] or d in members for i in members for d in graphs[r][i]["deps"]):
                    started.add(key)
                    progress = True
        for key in list(started):
            r, members = key[0], groups[key]
            if all(done[r][i] for i in members):
                continue
            if all(group_of[partner[(r, i)]] in started for i in members):
                for i in members:
                    done[r][i] = True
                progress = True
    stuck = [(r, i, graphs[r][i]) for r in range(R) for i in range(len(graphs[r])) if not done[r][i]]
    return not stuck, stuck[:6]


@pytest.mark.parametrize("grid", [(1, 2), (2, 1), (2, 2), (4, 2), (2, 4), (3, 3), (8, 1), (1, 8), (3, 2)])
@pytest.mark.parametrize("nt,mb,ratio", [(5, 512, 1.0), (9, 512, 0.05), (16, 1024, 3.0), (13, 256, 0.3)])
def test_the_transport_calls_of_all_ranks_complete_under_rccl_rules(grid, nt, mb, ratio):
    p, q = grid
    graphs = _launch_graphs(nt, mb, p, q, 1e-4 * ratio, 1e-4)
    ok, stuck = _replay(graphs)
    assert ok, stuck
    # channel 1 carries one group per wave on every rank that takes part in it, channel 0 the diagonal and head tiles:
    # the totals over the grid are what SURVEY 8e's messages add up to (every send has its receive: checked in _replay)
    assert sum(1 for g in graphs for op in g if op["ch"] == 1) > 0


def test_the_replay_sees_a_cycle():
    """Self-test: two ranks that each post a receive ahead of the send the other waits for, on ONE channel, cannot finish."""
    mk = lambda kind, peer, group, deps: {"ch": 0, "kind": kind, "peer": peer, "bytes": 8, "group": group, "deps": deps}
    a = [mk("R", 1, 0, []), mk("S", 1, 1, [0])]
    b = [mk("R", 0, 0, []), mk("S", 0, 1, [0])]
    ok, stuck = _replay([a, b])
    assert not ok and len(stuck) == 4
    # ... and the same calls grouped as RCCL wants them (send and receive of a rank in one group) do
    a = [mk("R", 1, 0, []), mk("S", 1, 0, [])]
    b = [mk("R", 0, 0, []), mk("S", 0, 0, [])]
    assert _replay([a, b])[0]


def test_the_predicted_timeline_replays_and_is_monotone_in_the_link():
    """scripts/predict_scale.py (profiles/r05_predicted_scale.txt): every rank's launch graph with rough durations and a
    (bandwidth, latency) link model.  Small case: the replay completes, communication can only add time, a slower link
    can only add more, and four ranks without communication lie between a quarter of one GPU's time and all of it."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("predict_scale", os.path.join(root, "scripts", "predict_scale.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    nt, mb, t_tile, t_panel = 24, 1024, 3.3e-5, 4.0e-4  # (update-bound, as the headline configuration is)
    one = max(ps.replay(ps.graphs_of(nt, mb, 1, 1, t_tile, t_panel), None, 0.0))
    g = ps.graphs_of(nt, mb, 2, 2, t_tile, t_panel)
    free = max(ps.replay(g, None, 0.0))
    fast = max(ps.replay(g, 100e9, 10e-6))
    slow = max(ps.replay(g, 10e9, 100e-6))
    assert 0 < free <= fast <= slow and one / 4 <= free < one
