// The wave walker: the right-looking tiled Cholesky of the reference client (client_distrib.cpp:506-565:
// POTRF(k); TRSM(i,k), i > k; SYRK / GEMM (i,j,k), k < j <= i) as ONE schedule for every descriptor
// chol_potrf_tile accepts -- a whole tiled matrix on one GPU (v6_test.c:44-56 with p = q = 1) and a p x q
// 2D block-cyclic matrix with one process per GPU (v6_test.c:26-27 passes p, q into the descriptor).  The
// single-GPU case is the p = q = 1 instance of the same code: no transport call, no extra stream operation.
//
// Streams per rank (Ops maps the ids to HIP streams; the CPU test engine executes in issue order):
//   ST_PANEL (high)  POTRF(k) as mb/128 diagonal-block steps, on the owner of (k,k)
//   ST_TRSM  (high)  the TRSM steps of this rank's tiles of panel k, one 128-column step behind the POTRF
//                    steps on the owner of (k,k); on the other ranks of its process column the whole local
//                    TRSM once L(k,k) and its block inverses have arrived (head tile first)
//   ST_U1    (mid)   column k+1 by panel k: the SYRK on (k+1,k+1) from the head tile L(k+1,k) first --
//                    POTRF(k+1) waits for nothing else -- then the rest of the column, which TRSM(k+1) needs
//   ST_MAIN  (low)   the columns beyond, beside the column-(k+1) launch
//   ST_CX    (high)  p*q > 1: the small, latency-critical messages -- L(k,k) with its block inverses down its
//                    process column, the head tile to the owner of (k+1,k+1); transport channel 0
//   ST_PX    (high)  p*q > 1: the exchange of panel k (whole parts along process rows, tile by tile to the
//                    process columns that use a tile as column operand); transport channel 1
// Three regimes by the length of a wave's (local) update against its panel chain: panels in pairs (two per
// pass of the far columns) while it is more than two chains long; the columns beyond k+1 as a near and a far
// launch on ST_U1 / ST_MAIN in the mid waves; and, once it is shorter than the chain and p = q = 1, the
// counter-linked form (SyrkPipe: TRSM steps, SYRK slices and the next POTRF launched ahead of time, polling
// device-side counters, no stream event on the chain).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "cholmi_internal.h"

namespace cholmi {

// geometry of one rank's share of an nt x nt tile matrix on a P x Q grid
struct WaveGeo {
  int nt = 0, P = 1, Q = 1, rank = 0, pr = 0, pc = 0, lmt = 0, lnt = 0, mb = 0, nbm = 0;
  size_t esize = 8, tile_bytes = 0, winv_bytes = 0;
  // local tiles of the trailing updates by column: ge[j] strictly-lower tiles, gd[j] diagonal tiles in
  // columns >= j (what chol_desc's work list holds, api.hip: build_worklist)
  std::vector<int> ge, gd;
  void init(int nt_, int mb_, int P_, int Q_, int rank_, size_t esize_) {
    nt = nt_, mb = mb_, P = P_, Q = Q_, rank = rank_, esize = esize_;
    nbm = mb / MACRO;
    pr = rank / Q, pc = rank % Q;
    lmt = nt - pr > 0 ? (nt - pr + P - 1) / P : 0;
    lnt = nt - pc > 0 ? (nt - pc + Q - 1) / Q : 0;
    tile_bytes = (size_t)mb * mb * esize;
    winv_bytes = (size_t)nbm * MACRO * MACRO * esize;
    ge.assign(nt + 2, 0);
    gd.assign(nt + 2, 0);
    int no = 0, nd = 0;
    for (int J = nt - 1; J >= 0; --J) {
      if (J % Q == pc) {
        const int rows = fla(J, pr) < lmt ? lmt - fla(J, pr) : 0;  // local rows with global index > J
        no += rows;
        if (J % P == pr) ++nd;
      }
      ge[J] = no;
      gd[J] = nd;
    }
  }
  int rank_of(int r, int c) const { return r * Q + c; }
  int fla(int k, int p2) const { return (k + P - p2) / P; }  // first local row of process row p2 with global index > k
  void part(int k, int p2, int *il0, int *cnt) const {       // the tiles of panel k owned by process row p2
    *il0 = fla(k, p2);
    const int rows = nt - p2 > 0 ? (nt - p2 + P - 1) / P : 0;
    *cnt = rows - *il0 > 0 ? rows - *il0 : 0;
  }
  int tiles_in(int jlo, int jhi) const {  // local tiles (off-diagonal + diagonal) in columns [jlo, jhi)
    jlo = std::min(jlo, nt), jhi = std::min(jhi, nt);
    return jlo < jhi ? (ge[jlo] - ge[jhi]) + (gd[jlo] - gd[jhi]) : 0;
  }
  int off_in(int jlo, int jhi) const {
    jlo = std::min(jlo, nt), jhi = std::min(jhi, nt);
    return jlo < jhi ? ge[jlo] - ge[jhi] : 0;
  }
  int diag_in(int jlo, int jhi) const {
    jlo = std::min(jlo, nt), jhi = std::min(jhi, nt);
    return jlo < jhi ? gd[jlo] - gd[jhi] : 0;
  }
};

// what a walker needs to know about speeds (seconds): measured once per context and dtype (api.hip: calibrate)
struct WaveCalib {
  double t_tile = 0;   // one tile of the trailing update (2 mb^3 flops) inside the DAG
  double t_panel = 0;  // the panel chain of one wave (mb/128 dependent steps)
};

// switches of the schedule (environment, read once; what each is for: INTEGRATION.md section 5).  Every factor is
// a threshold on (a wave's update time) / (its panel chain's time), both estimated from chol_init's calibration:
//   pair_fac   >= : panels in pairs (0: always, large: never)        yfac     < : the update's waves yield their CU to the chain
//   pipe_fac   <  : the counter-linked chain (0: never)              near_fac < : ... with column k+2 as a launch of its own (0: never)
//   flow_fac   <  : ... with the tile POTRF as a flow, per wave (default -1: by flow_run_fac on the whole factorisation)
struct WaveSwitches {
  double pair_fac = 2.0, yfac = 3.0, pipe_fac = 0.7, flow_fac = -1.0, halves_max_rounds = 24.0, near_fac = 0.7, flow_run_fac = 1.0;
  int u1_small_max = 8;  // counter-linked waves: column k+1 in the latency form while it has at most this many tiles below the diagonal (CHOLMI_U1_SMALL)
  WaveSwitches() {
    if (const char *e = getenv("CHOLMI_PAIR_FACTOR")) pair_fac = atof(e);
    if (const char *e = getenv("CHOLMI_YIELD_FACTOR")) yfac = atof(e);
    if (const char *e = getenv("CHOLMI_PIPE_FACTOR")) pipe_fac = atof(e);
    if (const char *e = getenv("CHOLMI_FLOW_FACTOR")) flow_fac = atof(e);
    if (const char *e = getenv("CHOLMI_FLOW_RUN_FACTOR")) flow_run_fac = atof(e);
    if (const char *e = getenv("CHOLMI_HALVES_MAX_ROUNDS")) halves_max_rounds = atof(e);
    if (const char *e = getenv("CHOLMI_NEAR_FACTOR")) near_fac = atof(e);
    if (const char *e = getenv("CHOLMI_U1_SMALL")) u1_small_max = atoi(e);
  }
};
// panels go in pairs only for tiles up to this edge, and from wave 1 on: wave 0 stays plain, so that its whole update runs
// beside panel 1's chain (with pairs from wave 0 on, the far columns' update by panel 0 is deferred and nothing runs beside it)
constexpr int PAIR_MAX_MB = 1024, PAIR_START = 1;

// two transport channels (include/cholmi.h: chol_transport_t), one per communication stream, so that the
// diagonal tile of wave k+1 never queues behind the exchange of panel k inside one communicator
struct WaveComm {
  chol_transport_t ch[2];
  bool open[2] = {false, false};
  long long nsend = 0, nrecv = 0, bytes_sent = 0;
  int begin(int c) {
    if (open[c]) return 0;
    open[c] = true;
    return ch[c].group_begin(ch[c].ctx);
  }
  int end(int c) {
    if (!open[c]) return 0;
    open[c] = false;
    return ch[c].group_end(ch[c].ctx);
  }
  int send(int c, const void *buf, size_t bytes, int peer, void *stream) {
    if (int rc = begin(c)) return rc;
    ++nsend;
    bytes_sent += (long long)bytes;
    return ch[c].send(ch[c].ctx, buf, bytes, peer, stream);
  }
  int recv(int c, void *buf, size_t bytes, int peer, void *stream) {
    if (int rc = begin(c)) return rc;
    ++nrecv;
    return ch[c].recv(ch[c].ctx, buf, bytes, peer, stream);
  }
};

#define WRC(call)        \
  do {                   \
    int rc_ = (call);    \
    if (rc_) return rc_; \
  } while (0)

// Ops: see HipOps / CbOps in dist.hip.  What the walker asks of it:
//   bool counters() const                  device-side counters usable (SyrkPipe); pipe_ok(): ... and the kernels' switches allow it
//   bool profiling() const, can_split_trsm() const
//   int begin(int nevents, int nt, int sem_per_wave)   events; zeroed info / yield table / counters on ST_MAIN
//   int rec(int ev, int st), wt(int st, int ev)
//   char *tile(int il, int jl)             local storage
//   void *winv(int par), void *alloc(size_t)
//   void *stream(int st)
//   int panel(k, lkk, winv, tiles, ntiles, ev_steps, ev_head, const SyrkPipeReq *sy, const int *wait_sem, int wait_target)
//   int trsm(k, tiles, ntiles, lkk, winv, st)
//   int diag_syrk(k, j, C, A, st)          C(j,j) -= A A^T, A = L(j,k)
//   int update(k, k2, jlo, jhi, what, pan, pan2, yield, st)   what: 1 off-diagonal tiles, 2 diagonal tiles, 3 both
//   int *sem(k, which, per_wave)           counters of wave k
//   int finish(ev_start, ev_stop, brackets, info*)   drain ST_MAIN (every stream has been joined into it), timings, info word
template <class O>
struct Walker {
  O &o;
  const WaveGeo &g;
  WaveComm *cm;  // null when p*q == 1
  WaveCalib cal;
  WaveSwitches sw;
  enum { E_PANEL, E_U1D, E_U1R, E_U2, E_P0, E_P1, E_NEAR, E_PN0, E_PN1, E_TRSM, E_LKK, E_LKKR, E_HEAD, E_CXS, E_SU, E_PER_WAVE };
  enum { F_START, F_STOP, F_JOIN, F_WAVE, F_TRSM, F_U1END, F_COLS, F_CX, F_PX, F_FLOWJ, F_FIXED };
  enum { NBUF = 4 };
  // receive buffers: L(k,k) + block inverses and the head tile by wave parity, panel parts by wave mod NBUF
  char *lkk_buf[2] = {nullptr, nullptr}, *head_buf[2] = {nullptr, nullptr};
  std::vector<char *> pbuf[NBUF];
  PanelRef pan[NBUF];
  int last_lkk[2] = {-1, -1}, last_head[2] = {-1, -1};
  double issue_us = 0, upd_flops = 0;
  int upd_launches = 0, flow_waves = 0;
  // how many waves ran in which regime (chol_last_potrf_regimes): a number can be tied to its schedule afterwards
  enum { R_PAIRED, R_PLAIN, R_HALVES, R_PIPE, R_NEAR1, R_FLOW, R_YIELD, R_U1SMALL, R_COUNT };
  int regimes[R_COUNT] = {};
  std::vector<int> halves_waves;

  Walker(O &ops, const WaveGeo &geo, WaveComm *comm, const WaveCalib &c) : o(ops), g(geo), cm(comm), cal(c) {}
  int ev(int k, int which) const { return E_PER_WAVE * k + which; }
  int fx(int which) const { return E_PER_WAVE * g.nt + which; }
  bool multi() const { return g.P * g.Q > 1; }
  const char *ptile(const PanelRef &p, int i) const {
    return (const char *)p.base[i % g.P] + (size_t)(i / g.P - p.first[i % g.P]) * g.tile_bytes;
  }

  int setup() {
    if (!multi()) return 0;
    const int maxpart = std::max(1, (g.nt + g.P - 1) / g.P);
    for (int par = 0; par < 2; ++par) {
      lkk_buf[par] = (char *)o.alloc(g.tile_bytes + g.winv_bytes);
      head_buf[par] = (char *)o.alloc(g.tile_bytes);
      if (!lkk_buf[par] || !head_buf[par]) return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "distributed potrf: buffer allocation failed");
    }
    for (int s = 0; s < NBUF; ++s) {
      pbuf[s].assign(g.P, nullptr);
      for (int p2 = 0; p2 < g.P; ++p2)
        if (!(pbuf[s][p2] = (char *)o.alloc((size_t)maxpart * g.tile_bytes)))
          return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "distributed potrf: buffer allocation failed");
    }
    return 0;
  }

  // ---- p*q > 1: what moves in wave k, point to point -----------------------------------------
  // WHAT THE TWO-COMMUNICATOR TRANSPORT RELIES ON (RCCL: channel 0 = diagonal and head tiles on ST_CX, channel 1 = panels
  // on ST_PX, one communicator each).  (1) Per channel, every rank issues its groups in the order of the waves, and
  // within a wave diagonal tile before head tile; a send and the receive that meets it are issued for the SAME wave
  // and kind, and the k-th send from a to b on a channel meets the k-th receive b posts from a.  (2) A rank with nothing
  // to do in a group issues nothing (WaveComm opens a group at its first operation).  (3) Nothing on the host waits
  // between two group ends: a group_end enqueues, so every rank gets to post the groups the others are waiting for;
  // what can block is stream-side only -- a group starts when its stream reaches it and ends when all its operations
  // have met their partners.  (4) The two channels are ordered against each other only through events of THIS rank
  // (E_TRSM ahead of the panel's sends, the buffers' last readers), never through the other channel's completion on
  // another rank.  tests/test_schedule_check.py replays the recorded launch graphs of all ranks of nine grids under
  // exactly these rules (test_the_transport_calls_of_all_ranks_complete_under_rccl_rules): no cycle.
  // L(k,k) and the inverses of its 128-blocks: owner -> the ranks of its process column that hold panel tiles
  int diag_send(int k, const char *lkk, const char *winv) {
    const int dr = k % g.P, dc = k % g.Q;
    bool any = false;
    for (int r2 = 0; r2 < g.P; ++r2) {
      int il0, cnt;
      g.part(k, r2, &il0, &cnt);
      if (r2 == dr || cnt <= 0) continue;
      if (!any) WRC(o.wt(ST_CX, ev(k, E_LKK)));
      any = true;
      WRC(cm->send(0, lkk, g.tile_bytes, g.rank_of(r2, dc), o.stream(ST_CX)));
      if (g.winv_bytes) WRC(cm->send(0, winv, g.winv_bytes, g.rank_of(r2, dc), o.stream(ST_CX)));
    }
    WRC(cm->end(0));
    if (any) WRC(o.rec(ev(k, E_CXS), ST_CX));
    return 0;
  }
  int diag_recv(int k) {
    const int dr = k % g.P, dc = k % g.Q, par = k & 1;
    // the buffer's last reader: this rank's TRSM of the last wave of the same parity it received a diagonal tile for
    if (last_lkk[par] >= 0) WRC(o.wt(ST_CX, ev(last_lkk[par], E_TRSM)));
    last_lkk[par] = k;
    WRC(cm->recv(0, lkk_buf[par], g.tile_bytes, g.rank_of(dr, dc), o.stream(ST_CX)));
    if (g.winv_bytes) WRC(cm->recv(0, lkk_buf[par] + g.tile_bytes, g.winv_bytes, g.rank_of(dr, dc), o.stream(ST_CX)));
    WRC(cm->end(0));
    return o.rec(ev(k, E_LKKR), ST_CX);
  }
  // the head tile L(k+1,k), ahead of the rest of the panel, to the owner of (k+1,k+1); -> where it is on this
  // rank (null elsewhere); E_HEAD(k) stands for it on the rank that uses it
  int head_move(int k, const char **head) {
    *head = nullptr;
    if (k + 1 >= g.nt) return 0;
    const int hr = (k + 1) % g.P, h = g.rank_of(hr, k % g.Q), d1 = g.rank_of(hr, (k + 1) % g.Q), par = k & 1;
    if (g.rank == h) {
      const char *t = o.tile((k + 1) / g.P, k / g.Q);
      if (d1 == h) {
        *head = t;  // E_HEAD(k) was recorded by whoever solved it (panel / trsm)
      } else {
        WRC(o.wt(ST_CX, ev(k, E_HEAD)));
        WRC(cm->send(0, t, g.tile_bytes, d1, o.stream(ST_CX)));
        WRC(cm->end(0));
      }
    } else if (g.rank == d1) {
      // the buffer's last reader: the SYRK of the last wave of the same parity whose next diagonal tile was mine
      if (last_head[par] >= 0) WRC(o.wt(ST_CX, ev(last_head[par], E_U1D)));
      last_head[par] = k;
      WRC(cm->recv(0, head_buf[par], g.tile_bytes, h, o.stream(ST_CX)));
      WRC(cm->end(0));
      WRC(o.rec(ev(k, E_HEAD), ST_CX));
      *head = head_buf[par];
    }
    return 0;
  }
  // panel k to everybody who uses it; pan[k % NBUF] says where its tiles are on this rank afterwards
  int exchange(int k) {
    const int dc = k % g.Q, slot = k % NBUF;
    PanelRef &p = pan[slot];
    memset(&p, 0, sizeof p);
    p.P = g.P;
    void *st = o.stream(ST_PX);
    bool waited = false;
    auto before = [&]() -> int {  // once, ahead of the first operation of the group
      if (waited) return 0;
      waited = true;
      if (g.pc == dc) WRC(o.wt(ST_PX, ev(k, E_TRSM)));
      // the buffers' last readers: the updates of waves k - NBUF and (paired) k - NBUF + 1
      if (k >= NBUF - 1) {
        WRC(o.wt(ST_PX, ev(k - NBUF + 1, E_U2)));
        WRC(o.wt(ST_PX, ev(k - NBUF + 1, E_SU)));
      }
      return 0;
    };
    for (int p2 = 0; p2 < g.P; ++p2) {
      int il0, cnt;
      g.part(k, p2, &il0, &cnt);
      p.first[p2] = il0;
      if (g.pr == p2 && g.pc == dc) {
        char *src = cnt > 0 ? o.tile(il0, k / g.Q) : pbuf[slot][p2];
        p.base[p2] = src;
        if (cnt <= 0) continue;
        WRC(before());
        for (int c2 = 0; c2 < g.Q; ++c2)  // along the process row: the whole part
          if (c2 != dc) WRC(cm->send(1, src, (size_t)cnt * g.tile_bytes, g.rank_of(p2, c2), st));
        for (int r2 = 0; r2 < g.P; ++r2) {  // to the other process rows: tile j to process column j mod q ...
          if (r2 == p2) continue;
          for (int t = 0; t < cnt;) {  // ... runs of consecutive tiles for the same rank as one message
            const int c = ((il0 + t) * g.P + p2) % g.Q;
            int t1 = t + 1;
            while (t1 < cnt && ((il0 + t1) * g.P + p2) % g.Q == c) ++t1;
            WRC(cm->send(1, src + (size_t)t * g.tile_bytes, (size_t)(t1 - t) * g.tile_bytes, g.rank_of(r2, c), st));
            t = t1;
          }
        }
      } else {
        char *buf = pbuf[slot][p2];
        p.base[p2] = buf;
        if (cnt <= 0) continue;
        if (g.pr == p2) {
          WRC(before());
          WRC(cm->recv(1, buf, (size_t)cnt * g.tile_bytes, g.rank_of(p2, dc), st));
        } else {
          for (int t = 0; t < cnt;) {  // the sender's runs: consecutive tiles of my process column
            if (((il0 + t) * g.P + p2) % g.Q != g.pc) {
              ++t;
              continue;
            }
            int t1 = t + 1;
            while (t1 < cnt && ((il0 + t1) * g.P + p2) % g.Q == g.pc) ++t1;
            WRC(before());
            WRC(cm->recv(1, buf + (size_t)t * g.tile_bytes, (size_t)(t1 - t) * g.tile_bytes, g.rank_of(p2, dc), st));
            t = t1;
          }
        }
      }
    }
    WRC(cm->end(1));
    if (!waited && g.pc == dc) WRC(o.wt(ST_PX, ev(k, E_TRSM)));
    return o.rec(ev(k, E_PANEL), ST_PX);
  }

  int run(long long *info_out) {
    const auto t_host0 = std::chrono::steady_clock::now();
    const int nt = g.nt, mb = g.mb, nbm = g.nbm, P = g.P, Q = g.Q;
    const bool mr = multi();
    const int sem_per_wave = 3 * nbm + 1 + flow_ctl_lines(nbm);  // SyrkPipe's counters, the flow's control block
    WRC(o.begin(E_PER_WAVE * nt + F_FIXED + nbm + 1, nt, sem_per_wave));
    const int ev_steps = fx(F_FIXED);
    WRC(o.rec(fx(F_START), ST_MAIN));
    for (int st = ST_PANEL; st < (mr ? ST_COUNT : ST_CX + 1); ++st) WRC(o.wt(st, fx(F_START)));  // (their kernels may poll counters zeroed on ST_MAIN; ST_CX: the flow's row-slab kernel)
    bool paired = false, cols_pending = false, had_pairs = false;
    int open_bracket = -1;  // odd wave whose profiling bracket is still open
    int bnd = -1;
    bool prev_halves = false, prev_flow = false, flow_joined = false;
    const bool flags = !mr && o.counters();
    // On a grid only ONE of the chain's edges is local to a rank: POTRF steps -> its own panel tiles' TRSM steps, on the
    // owner of (k,k).  (The other -- last SYRK slice -> next POTRF -- never is: tile (k+1,k+1) belongs to another rank.)
    // That edge runs on counters too (round 4): the TRSM steps are launched ahead and poll D[s] / I[s] instead of waiting
    // for an event per step; everything that crosses a transport call stays an event.
    const bool flags_local = mr && o.counters();
    // (1.0 panel estimates: N <= ~9000-10000 at tile 512 -- with the near column and the latency-form column update of
    // round 4 the form gains 7 % at N = 8192 and nothing at 12288; at 0.7 the rule flipped at 8192 with the box's calibration)
    const bool flow_run = (double)g.tiles_in(1, nt) * cal.t_tile < sw.flow_run_fac * cal.t_panel;
    const int *wait_sem = nullptr;  // what this wave's first diagonal-block step polls, when the last wave raised it
    int wait_target = 0;
    const double b3 = (double)mb * mb * mb;
    const double t_tile = cal.t_tile, t_panel = cal.t_panel;
    const bool prof = o.profiling();
    for (int k = 0; k < nt; ++k) {
      const int dr = k % P, dc = k % Q, par = k & 1;
      const bool in_col = g.pc == dc, own_diag = in_col && g.pr == dr, last = k + 1 >= nt;
      int il0m = 0, cntm = 0;
      g.part(k, g.pr, &il0m, &cntm);
      if (!in_col) cntm = 0;
      // ---- panel k: POTRF on the owner of (k,k), the TRSM steps of its own panel tiles pipelined behind it;
      // the other ranks of the process column solve theirs once L(k,k) has arrived  (C2:510-535)
      // (ST_TRSM needs no event for the start of the wave: its first step waits for the event recorded on
      // ST_PANEL behind the first diagonal-block step, and a record on ST_PANEL costs the chain ~7 us)
      const int local_tiles = g.tiles_in(k + 1, nt);  // this rank's tiles of wave k's update
      const bool pair_first = k >= PAIR_START && ((k - PAIR_START) & 1) == 0;
      if (pair_first)
        paired = mb <= PAIR_MAX_MB && k + 2 < nt && (double)local_tiles * t_tile >= sw.pair_fac * t_panel;
      // Plain (unpaired) wave whose panel chain is (nearly) critical: the SYRK on tile (k+1,k+1) follows the
      // head tile's TRSM steps slice by slice and the chain's cross-stream edges are device-side counters
      // (only while the update is shorter than about a panel chain: the polling workgroups hold CU slots the
      // update would otherwise use -- measured +10 ... +20 % on the waves between pipe_fac and the yield threshold)
      const bool plain_yield = (double)local_tiles * t_tile < sw.yfac * t_panel;
      const bool chain_bound = (double)local_tiles * t_tile < sw.pipe_fac * t_panel;
      const bool pipe = flags && !paired && !last && o.pipe_ok() && plain_yield && chain_bound;
      const bool pipe_local = flags_local && !paired && o.pipe_ok() && plain_yield && chain_bound && own_diag && cntm > 0;
      // ... with its tile POTRF as a flow (kernels.hip: k_flow_factor) when the WHOLE factorisation is chain-bound (wave 0
      // already is): measured round 4, the form gains 14-17 % there (tile 512, N <= 4096) and nothing when only the last
      // waves of a larger matrix use it -- the wave that switches forms pays ~90 us, the rest gains ~40 us each
      // (CHOLMI_FLOW_FACTOR = f > 0: instead, every wave whose update is shorter than f panel estimates)
      const bool flow = pipe && own_diag && cntm > 0 && flow_ctl_lines(nbm) > 0 && o.flow_ok() && flow_applies(nbm) &&
                        (sw.flow_fac > 0 ? (double)local_tiles * t_tile < sw.flow_fac * t_panel : flow_run);
      if (k > 0 && in_col) WRC(o.wt(ST_TRSM, ev(k - 1, E_U1R)));
      // block inverses of L(k,k): two workspaces alternating by wave, so that POTRF(k+1) may overwrite its
      // set while TRSM(k) still reads the other
      const char *head = nullptr;
      if (own_diag) {
        char *lkk = o.tile(k / P, k / Q);
        SyrkPipe sy;
        if (pipe_local) {
          sy.c = nullptr;  // (no SYRK slices: the next diagonal tile is another rank's)
          sy.su = nullptr;
          sy.sem = o.sem(k, 0, sem_per_wave);
        }
        if (pipe) {
          // the tile's earlier writers: U2(k-1), whose range includes column k+1 (or, behind the paired
          // phase, the column launches of the last pair, which precede this on ST_U1)
          // (... or near(k-1): column k+1 alone, on this very stream)
          if (k > 0 && !prev_halves) WRC(o.wt(ST_U1, ev(k - 1, E_U2)));
          sy.c = o.tile((k + 1) / P, (k + 1) / Q);
          sy.su = (hipStream_t)o.stream(ST_U1);
          sy.sem = o.sem(k, 0, sem_per_wave);
          if (flow) {
            sy.fc = o.sem(k, 3 * nbm + 1, sem_per_wave);
            sy.sflow = (hipStream_t)o.stream(ST_CX);
            sy.ev_flow = (hipEvent_t)o.flow_event();
            // The row-slab kernel (ST_CX) reads its rows of tile (k,k).  Behind another flow wave it follows that wave's
            // row-slab kernel in stream order and polls the counter the POTRF's first kernel polls (the last SYRK slice
            // of wave k-1, the tile's last writer).  Otherwise its stream joins the POTRF stream's order by an event --
            // unless that happened a wave ahead (flow_joined, below) AND this wave polls that counter: without a counter
            // to poll (the wave before was not counter-linked) the event is the only thing that orders it.
            sy.join_flow = !prev_flow && (wait_sem == nullptr || !flow_joined);
          }
        }
        // the head tile is this rank's first panel tile only when there is one process row
        const bool head_mine = !last && P == 1;
        WRC(o.panel(k, lkk, o.winv(par), lkk + g.tile_bytes, cntm, ev_steps, head_mine ? ev(k, E_HEAD) : -1,
                    pipe || pipe_local ? &sy : nullptr, wait_sem, wait_target));
        if (mr && P > 1 && !last) {
          WRC(o.rec(ev(k, E_LKK), ST_PANEL));
          WRC(diag_send(k, lkk, (const char *)o.winv(par)));
        }
        // the next wave switches to the flow form: its row-slab kernel's stream joins the POTRF stream's order NOW, a
        // whole wave ahead, so that the event's wake-up (15-30 us) is not on the first flow wave's chain.  (Only behind
        // a counter-linked wave: the row-slab kernel then polls the same counter as the POTRF's first kernel.)
        if (pipe && !flow && !flow_joined && !mr && k + 2 < nt && flow_ctl_lines(nbm) > 0 && o.flow_ok() && flow_applies(nbm) &&
            (sw.flow_fac > 0 ? (double)g.tiles_in(k + 2, nt) * t_tile < sw.flow_fac * t_panel : flow_run)) {
          WRC(o.rec(fx(F_FLOWJ), ST_PANEL));
          WRC(o.wt(ST_CX, fx(F_FLOWJ)));
          flow_joined = true;
        }
      } else if (in_col && cntm > 0) {
        WRC(diag_recv(k));
        WRC(o.wt(ST_TRSM, ev(k, E_LKKR)));
        char *tiles = o.tile(il0m, k / Q);
        const char *lkk = lkk_buf[par], *wv = lkk_buf[par] + g.tile_bytes;
        const bool head_mine = !last && g.pr == (k + 1) % P;  // my first tile is L(k+1,k)
        if (head_mine && cntm > 1 && o.can_split_trsm()) {
          WRC(o.trsm(k, tiles, 1, lkk, wv, ST_TRSM));
          WRC(o.rec(ev(k, E_HEAD), ST_TRSM));
          WRC(o.trsm(k, tiles + g.tile_bytes, cntm - 1, lkk, wv, ST_TRSM));
        } else {
          WRC(o.trsm(k, tiles, cntm, lkk, wv, ST_TRSM));
          if (head_mine) WRC(o.rec(ev(k, E_HEAD), ST_TRSM));
        }
      }
      // with device-side edges ST_PANEL waits for no event between waves: POTRF(k+1)'s first step polls the
      // last slice's counter, which also stands behind TRSM(k) (same stream, earlier), so POTRF(k+2) may
      // reuse TRSM(k)'s workspace
      const bool by_flags = pipe;
      prev_flow = flow;
      if (flow) ++flow_waves, ++regimes[R_FLOW];
      if (pipe || pipe_local) ++regimes[R_PIPE];
      wait_sem = by_flags ? o.sem(k, 3 * nbm, sem_per_wave) : nullptr;
      wait_target = (mb / 64) * (mb / 64 + 1) / 2;
      // TRSM(k) complete on this rank
      if (in_col) WRC(o.rec(ev(k, mr ? E_TRSM : E_PANEL), ST_TRSM));
      if (own_diag && !by_flags) {
        WRC(o.wt(ST_PANEL, ev(k, mr ? E_TRSM : E_PANEL)));  // POTRF(k+2) overwrites this wave's block inverses
        if (mr && P > 1 && !last) WRC(o.wt(ST_PANEL, ev(k, E_CXS)));  // ... which also travel
      }
      if (last) break;
      if (mr) {
        WRC(head_move(k, &head));
        WRC(exchange(k));
      } else {
        PanelRef &p = pan[k % NBUF];
        memset(&p, 0, sizeof p);
        p.P = 1;
        p.base[0] = o.tile(0, k);
        head = o.tile(k + 1, k);
      }
      const PanelRef &pk = pan[k % NBUF];
      const bool next_diag_mine = g.rank_of((k + 1) % P, (k + 1) % Q) == g.rank;
      char *ckk = next_diag_mine ? o.tile((k + 1) / P, (k + 1) / Q) : nullptr;
      // ---- trailing update (C2:540-560)
      if (paired) {
        // Panels in pairs (k-1, k), k odd: the far columns' update by the even panel is deferred and applied
        // together with the odd one in ONE pass of twice the K (k_trail_update, npan = 2).
        //   even k:  U1(k)  = column k+1 by panel k                                       (ST_U1)
        //   odd  k:  U1'(k) = column k+1, Ca = column k+2, Cb = column k+3 by panels k-1, k  (ST_U1, in this order)
        //            big(k) = the columns from k+4 on by panels k-1, k                      (ST_MAIN, beside them)
        // Every column is written by launches of ST_U1 in program order, except by big(); the first launches
        // of ST_U1 on a column big(k) covers are Ca / Cb of wave k+2, which wait for it.  POTRF(k+1) waits
        // for the SYRKs on (k+1,k+1) only, TRSM(k+1) for the rest of column k+1.
        const bool odd = !pair_first;
        // the first pair behind plain waves: their far launch (ST_MAIN) wrote the columns ST_U1 is about to touch
        if (!odd && k > 0 && !had_pairs) WRC(o.wt(ST_U1, ev(k - 1, E_U2)));
        if (prev_halves) {  // ... and their near launch (ST_U1) the columns ST_MAIN is about to touch
          WRC(o.wt(ST_MAIN, ev(k - 1, E_NEAR)));
          prev_halves = false;
        }
        const PanelRef &prev = pan[(odd ? k - 1 : k) % NBUF];
        const PanelRef *p2 = odd ? &pk : nullptr;  // launches: first `prev`, then `pk` when odd
        const PanelRef &p1 = odd ? prev : pk;
        const int k1 = odd ? k - 1 : k, k2 = odd ? k : -1;
        const int n_c1 = g.tiles_in(k + 1, k + 2), n_ca = g.tiles_in(k + 2, k + 3), n_cb = g.tiles_in(k + 3, k + 4),
                  n_big = g.tiles_in(k + 4, nt);
        const int wave_tiles = n_c1 + (odd ? n_ca + n_cb + n_big : 0);
        const bool yield = (double)wave_tiles * t_tile * (odd ? 2 : 1) < sw.yfac * t_panel * (odd ? 2 : 1);
        if (next_diag_mine) {
          if (odd) WRC(o.diag_syrk(k - 1, k + 1, ckk, ptile(prev, k + 1), ST_U1));
          WRC(o.wt(ST_U1, ev(k, E_HEAD)));  // the head tile L(k+1,k) is all the last SYRK needs
          WRC(o.diag_syrk(k, k + 1, ckk, head, ST_U1));
          WRC(o.rec(ev(k, E_U1D), ST_U1));
          WRC(o.wt(ST_PANEL, ev(k, E_U1D)));
        }
        WRC(o.wt(ST_U1, ev(k, E_PANEL)));
        if (!odd && open_bracket < 0 && prof) WRC(o.rec(ev(k, E_P0), ST_U1));
        WRC(o.update(k1, k2, k + 1, k + 2, 1, p1, p2, yield, ST_U1));
        WRC(o.rec(ev(k, E_U1R), ST_U1));
        int timed = 0;
        double fl = 0;
        const int o_c1 = g.off_in(k + 1, k + 2);
        if (odd) {
          if (o_c1 > 0) ++timed, fl += 2.0 * o_c1;
          if (k >= 2) WRC(o.wt(ST_U1, ev(k - 2, E_U2)));  // big(k-2) covered these columns
          WRC(o.update(k1, k2, k + 2, k + 3, 3, p1, p2, yield, ST_U1));
          WRC(o.update(k1, k2, k + 3, k + 4, 3, p1, p2, yield, ST_U1));
          if (n_ca > 0) ++timed, fl += 2.0 * g.off_in(k + 2, k + 3) + g.diag_in(k + 2, k + 3);
          if (n_cb > 0) ++timed, fl += 2.0 * g.off_in(k + 3, k + 4) + g.diag_in(k + 3, k + 4);
          WRC(o.rec(fx(F_COLS), ST_U1));
          cols_pending = true;
          WRC(o.wt(ST_MAIN, ev(k, E_PANEL)));
          if (prof) WRC(o.rec(ev(k, E_P0), ST_MAIN));
          WRC(o.update(k1, k2, k + 4, nt, 3, p1, p2, yield, ST_MAIN));
          if (n_big > 0) ++timed, fl += 2.0 * g.off_in(k + 4, nt) + g.diag_in(k + 4, nt);
          WRC(o.rec(ev(k, E_U2), ST_MAIN));
          // the bracket [P0(k), P1(k)] covers every k_trail_update launch of the pair's update: the
          // two-panel launches of this wave, which start together, and U1(k+1), which runs beside
          // big(k); it is closed at the next wave
          open_bracket = k;
          upd_launches += timed;
          upd_flops += 2.0 * fl * b3;  // two panels per pass
        } else {
          WRC(o.rec(ev(k, E_U2), ST_MAIN));
          if (prof) {
            if (open_bracket >= 0) {
              // waiting for the column launches and for U1(k) on ST_MAIN constrains nothing: big(k+1)
              // needs panel k+1, which comes after all of them
              WRC(o.wt(ST_MAIN, fx(F_COLS)));
              WRC(o.wt(ST_MAIN, ev(k, E_U1R)));
              WRC(o.rec(ev(open_bracket, E_P1), ST_MAIN));
              WRC(o.rec(ev(k, E_P0), ST_MAIN));
              WRC(o.rec(ev(k, E_P1), ST_MAIN));
            } else {
              WRC(o.rec(ev(k, E_P1), ST_U1));  // the very first wave: U1(0) alone, bracketed on its stream
            }
          }
          open_bracket = -1;
          if (o_c1 > 0) ++upd_launches, upd_flops += 2.0 * o_c1 * b3;
        }
        had_pairs = true;
        ++regimes[R_PAIRED];
        if (yield) ++regimes[R_YIELD];
        if (mr) WRC(o.rec(ev(k, E_SU), ST_U1));
        continue;
      }
      if (open_bracket >= 0) {  // the paired phase ended on an odd wave: close its bracket
        if (prof) {
          WRC(o.wt(ST_MAIN, fx(F_COLS)));
          WRC(o.rec(ev(open_bracket, E_P1), ST_MAIN));
        }
        open_bracket = -1;
      }
      const int n_r1o = g.off_in(k + 1, k + 2), n_r1d = g.diag_in(k + 1, k + 2);
      const int n_r2o = g.off_in(k + 2, nt), n_r2d = g.diag_in(k + 2, nt);
      // Give CUs to the next panel's guest workgroups only when that panel is on the critical path, i.e. when
      // this wave's update is not much longer than a panel; otherwise the polling is pure cost.
      const bool yield = (double)(n_r1o + n_r1d + n_r2o + n_r2d) * t_tile < sw.yfac * t_panel;
      const bool split = yield;
      // the SYRK on (k+1,k+1) needs the head tile L(k+1,k) only; everything else the whole panel
      if (!pipe) {
        if (!split) WRC(o.wt(ST_U1, ev(k, E_PANEL)));
        else if (next_diag_mine) WRC(o.wt(ST_U1, ev(k, E_HEAD)));
      }
      if (cols_pending) {  // first plain wave after the paired phase: Cb of the last pair wrote column k+2
        WRC(o.wt(ST_MAIN, fx(F_COLS)));
        cols_pending = false;
      }
      // Plain waves of a few rounds of workgroups: the columns beyond k+1 go out as TWO launches, the near
      // columns [k+2, bnd) on ST_U1 behind the column-(k+1) launch and the far ones [bnd, nt) on ST_MAIN.
      // With a boundary that stays put for several waves each half depends on its own predecessor only
      // (far(k+1) is a subset of far(k), near(k+1) of near(k)), so the last, partly filled round of one launch
      // runs beside full rounds of the other chain's next launch instead of beside nothing; and column k+1 --
      // the next panel -- waits for near(k-1) only.  The boundary moves (then near(k) also waits for far(k-1))
      // when the near part has shrunk under 30 % of the wave.
      const int tiles2 = n_r2o + n_r2d;
      bool halves = sw.halves_max_rounds > 0 && !pipe && split && nt - 1 - k >= 6 &&
                    (double)tiles2 * nbm * nbm / 512.0 < sw.halves_max_rounds;
      // Counter-linked waves (CHOLMI_PIPE_NEAR): the near half is column k+2 alone, every wave.  The SYRK slices of the
      // NEXT wave (tile (k+2,k+2), on ST_U1) then follow near(k) in stream order instead of waiting for the whole far
      // update of this wave, which started only when this panel was complete: the chain looks two columns ahead.
      const bool near1 = pipe && split && k + 3 < nt && (double)local_tiles * t_tile < sw.near_fac * t_panel;
      bool moved = false;
      if (near1) {
        halves = true;
        if (prev_halves && bnd > k + 3) WRC(o.wt(ST_MAIN, ev(k - 1, E_NEAR)));  // (from the wider halves of the waves before)
        bnd = k + 3;
        moved = true;
      } else if (halves) {
        if (!prev_halves || bnd <= k + 2 || g.tiles_in(k + 2, bnd) * 10 < tiles2 * 3) {
          int b = k + 3;
          while (b < nt - 1 && g.tiles_in(k + 2, b) * 2 < tiles2) ++b;
          // (the boundary only ever moves right, far(k) stays a subset of far(k-1); should it not, far(k)
          // waits for near(k-1) as well)
          if (prev_halves && b < bnd) WRC(o.wt(ST_MAIN, ev(k - 1, E_NEAR)));
          bnd = b;
          moved = true;
        }
        if (bnd >= nt) halves = false;
      }
      if (prev_halves && !halves) WRC(o.wt(ST_MAIN, ev(k - 1, E_NEAR)));  // U2(k) covers near(k-1)'s columns
      // column k+1 was in U2(k-1)'s range -- or in near(k-1)'s, which precedes this on ST_U1
      if (k > 0 && !prev_halves) WRC(o.wt(ST_U1, ev(k - 1, E_U2)));
      WRC(o.wt(ST_MAIN, ev(k, E_PANEL)));
      int timed = 0;  // k_trail_update launches inside this wave's profiling bracket
      // (... of tiles up to 512: with 1024 tiles the form lost 2-4 % -- 256 workgroups per tile, K = 1024 each)
      const bool u1s = pipe && !mr && nbm <= 4 && n_r1o > 0 && n_r1o <= sw.u1_small_max;
      if (split) {
        // the panel chain is (nearly) critical: the diagonal tile (k+1,k+1) alone first, POTRF(k+1)
        // needs nothing else; then the rest of column k+1, which TRSM(k+1) needs
        if (!pipe && next_diag_mine) {
          WRC(o.diag_syrk(k, k + 1, ckk, head, ST_U1));
          WRC(o.rec(ev(k, E_U1D), ST_U1));
        }
        WRC(o.wt(ST_U1, ev(k, E_PANEL)));
        if (halves && prof) WRC(o.rec(ev(k, E_PN0), ST_U1));
        if (u1s) {
          WRC(o.update_col_small(k, ST_U1));
        } else {
          WRC(o.update(k, -1, k + 1, k + 2, 1, pk, nullptr, yield, ST_U1));
          if (n_r1o > 0) ++timed;
        }
        WRC(o.rec(ev(k, E_U1R), ST_U1));
        if (halves) {
          if (moved && prev_halves) WRC(o.wt(ST_U1, ev(k - 1, E_U2)));  // columns taken over from far(k-1)
          WRC(o.update(k, -1, k + 2, bnd, 3, pk, nullptr, yield, ST_U1));
          if (g.tiles_in(k + 2, bnd) > 0) ++timed;
          WRC(o.rec(ev(k, E_NEAR), ST_U1));
          if (prof) WRC(o.rec(ev(k, E_PN1), ST_U1));
        }
      } else {
        // the update dwarfs the panel: one launch for the whole column (one tail less per wave)
        WRC(o.update(k, -1, k + 1, k + 2, 3, pk, nullptr, yield, ST_U1));
        if (n_r1o + n_r1d > 0) ++timed;
        WRC(o.rec(ev(k, E_U1D), ST_U1));
        WRC(o.rec(ev(k, E_U1R), ST_U1));
      }
      if (!by_flags && next_diag_mine) WRC(o.wt(ST_PANEL, ev(k, E_U1D)));
      if (prof) WRC(o.rec(ev(k, E_P0), ST_MAIN));
      if (halves) {
        WRC(o.update(k, -1, bnd, nt, 3, pk, nullptr, yield, ST_MAIN));
        if (g.tiles_in(bnd, nt) > 0) ++timed;
        halves_waves.push_back(k);
      } else if (tiles2 > 0) {
        WRC(o.update(k, -1, k + 2, nt, 3, pk, nullptr, yield, ST_MAIN));
        ++timed;
      }
      WRC(o.rec(ev(k, E_U2), ST_MAIN));
      if (prof) {
        // the bracket [P0, P1] on ST_MAIN covers every k_trail_update launch of the wave: U1(k) started
        // with U2(k); waiting for its end here constrains nothing (U2(k+1) needs panel k+1, which needs it).
        // (halves: far(k+1) does NOT need near(k) -- two brackets, [PN0, PN1] on ST_U1 for column k+1 and the
        // near half, and the host takes the union)
        if (!halves) WRC(o.wt(ST_MAIN, ev(k, E_U1R)));
        WRC(o.rec(ev(k, E_P1), ST_MAIN));
      }
      if (mr) WRC(o.rec(ev(k, E_SU), ST_U1));
      prev_halves = halves;
      ++regimes[near1 ? R_NEAR1 : halves ? R_HALVES : R_PLAIN];
      if (yield) ++regimes[R_YIELD];
      if (u1s) ++regimes[R_U1SMALL];
      upd_launches += timed;
      // algorithmic flops of the launches inside the bracket: GEMM 2 B^3 per off-diagonal tile, SYRK B^3
      // per diagonal tile; the diagonal-tile SYRK of the split form is not a k_trail_update
      upd_flops += (2.0 * n_r2o + n_r2d) * b3;
      if (!u1s) upd_flops += (2.0 * n_r1o + (split ? 0 : n_r1d)) * b3;
    }
    if (open_bracket >= 0 && prof) {
      WRC(o.wt(ST_MAIN, fx(F_COLS)));
      WRC(o.rec(ev(open_bracket, E_P1), ST_MAIN));
    }
    (void)had_pairs;
    issue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count();
    // join every stream into ST_MAIN
    const int joins[5][2] = {{F_JOIN, ST_PANEL}, {F_TRSM, ST_TRSM}, {F_U1END, ST_U1}, {F_CX, ST_CX}, {F_PX, ST_PX}};
    for (int i = 0; i < (mr ? 5 : 4); ++i) {  // (one GPU: ST_CX carries the flow's row-slab kernel)
      WRC(o.rec(fx(joins[i][0]), joins[i][1]));
      WRC(o.wt(ST_MAIN, fx(joins[i][0])));
    }
    WRC(o.rec(fx(F_STOP), ST_MAIN));
    std::vector<std::pair<int, int>> brackets;
    if (prof) {
      for (int k = 0; k + 1 < nt; ++k) brackets.emplace_back(ev(k, E_P0), ev(k, E_P1));
      for (int k : halves_waves) brackets.emplace_back(ev(k, E_PN0), ev(k, E_PN1));
    }
    int info = 0;
    WRC(o.finish(fx(F_START), fx(F_STOP), brackets, &info));
    // the smallest positive info wins: MAX-reduce (2^40 - info), 0 = success
    long long v = info > 0 ? (1LL << 40) - info : 0;
    if (mr) WRC(cm->ch[0].allreduce_max(cm->ch[0].ctx, &v));
    *info_out = v == 0 ? 0 : (1LL << 40) - v;
    return 0;
  }
};

}  // namespace cholmi
