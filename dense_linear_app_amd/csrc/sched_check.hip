// Schedule checker (test infrastructure, no GPU needed): the wave walker of walker.h run over an Ops that executes
// nothing and RECORDS -- every launch with the tiles it reads and writes, every event record / wait, every counter a
// stream is gated on -- and then checks that any two launches that touch the same tile (or the same block-inverse
// workspace), one of them writing, are ordered: by their stream, by an event, or by a counter.  The walker's
// dependencies are per regime (pairs, near / far halves, counter-linked waves, the near column, the flow form ...) and a
// missing one shows on the GPU only as a rare wrong digit; here it is a deterministic finding with the two launches'
// names.  On a p x q grid the check is per rank: sends read and receives write (tiles and the rotating receive buffers) on
// the stream they are issued on; what the OTHER ranks do is tests/test_dist_cabi_gloo.py's business.
//
// Granularity: o.panel() is one POTRF launch (ST_PANEL), one TRSM launch over the whole panel (ST_TRSM, behind it), and
// in counter-linked waves one launch for the SYRK slices (ST_U1, behind the TRSM) -- what happens between those three is
// kernels.hip's business (launch_panel_pipelined) and is tested on the GPU.  The flow form of the tile POTRF is TWO
// launches: the factorisation (k_flow_factor, ST_PANEL) and the row slabs (k_flow_rows, ST_CX), each with the
// dependencies its own stream gives it -- the row slabs have ST_PANEL's only through the join event (sy->join_flow) or
// the counter both kernels poll (wait_sem); the two hand panels to each other inside the kernels (not checked here:
// tests/test_gpu_full.py, CHOLMI_FLOW_FENCES A/B) and are therefore exempt from the pairwise check AGAINST EACH OTHER.
#include <bitset>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/cholmi.h"
#include "cholmi_internal.h"
#include "walker.h"

using namespace cholmi;

namespace {

struct TraceOps {
  const WaveGeo &g;
  bool prof;
  struct Op {
    std::string name;
    std::vector<std::pair<long, bool>> acc;  // (slot, written): tile-sized slots of a made-up address space, see slot_of
    std::vector<int> deps;
    int ch = -1, peer = -1, group = -1;  // a transport call: channel, peer, the group (per channel) it belongs to
    bool is_send = false;
    size_t bytes = 0;
    char kind = '-';   // P tile POTRF, T panel TRSM, U update launch, Y one-tile SYRK / slices, L latency-form column
    double cost = 0;   // a rough duration [s] from the calibration the walker was given (scripts/predict_scale.py)
  };
  double t_tile = 0, t_panel = 0;  // the walker's calibration (seconds per tile update at full-chip rate, per panel chain)
  // one round of update workgroups (512 of them, nbm^2 per tile at the full-chip rate of one tile per t_tile)
  double t_round() const { return t_tile * 512.0 / (g.nbm * g.nbm); }
  void cost_of(int id, char kind, double cost) {
    if (id >= 0 && id < (int)ops.size()) ops[id].kind = kind, ops[id].cost = cost;
  }
  std::vector<Op> ops;
  int last_on[ST_COUNT];
  std::set<int> pending[ST_COUNT];       // what the next launch of a stream has to follow beyond the stream's last one
  std::map<int, std::set<int>> ev_deps;  // a recorded event: the launches it stands behind
  std::map<const int *, int> producer;   // a counter: the launch that raises it to its target
  std::vector<int> sems;
  std::vector<std::string> findings;
  std::set<std::pair<int, int>> coop;     // pairs of launches that cooperate through in-kernel counters (flow form)
  int drop_wait = -1, drop_gate = -1, nwaits = 0, ngates = 0;  // self-test: the n-th event wait / gate is ignored
  std::string dropped;
  // the address space: this rank's lmt x lnt tiles, the two block-inverse workspaces, then whatever the walker allocates
  // (receive buffers of a grid), all in units of one tile
  char *const fake_base = reinterpret_cast<char *>(uintptr_t(1) << 40);
  long local_slots = 0, next_slot = 0;

  TraceOps(const WaveGeo &geo, bool profiling) : g(geo), prof(profiling) {
    for (int &l : last_on) l = -1;
    local_slots = (long)g.lmt * std::max(1, g.lnt);
    next_slot = local_slots + 2;
  }
  bool profiling() const { return prof; }
  bool pipe_ok() const { return true; }
  bool counters() const { return !sems.empty(); }
  bool can_split_trsm() const { return true; }
  bool flow_ok() const { return true; }
  void *flow_event() { return nullptr; }
  void *stream(int st) { return reinterpret_cast<void *>(uintptr_t(st + 1)); }
  static int stream_id(void *s) { return (int)reinterpret_cast<uintptr_t>(s) - 1; }
  char *tile(int il, int jl) { return fake_base + ((size_t)il + (size_t)jl * g.lmt) * g.tile_bytes; }
  void *winv(int par) { return fake_base + (size_t)(local_slots + par) * g.tile_bytes; }
  void *alloc(size_t bytes) {
    char *p = fake_base + (size_t)next_slot * g.tile_bytes;
    next_slot += (long)((bytes + g.tile_bytes - 1) / g.tile_bytes);
    return p;
  }
  int *sem(int k, int which, int per_wave) { return sems.data() + ((size_t)per_wave * k + which) * 32; }
  using Acc = std::vector<std::pair<long, bool>>;
  void touch(Acc &a, const void *ptr, size_t bytes, bool write) const {
    if (!ptr || !bytes) return;
    const long s0 = (long)(((const char *)ptr - fake_base) / (long)g.tile_bytes);
    const long s1 = (long)(((const char *)ptr + bytes - 1 - fake_base) / (long)g.tile_bytes);
    for (long t = s0; t <= s1; ++t) a.push_back({t, write});
  }
  std::string slot_name(long t) const {
    char b[96];
    if (t < local_slots) {
      const int il = (int)(t % g.lmt), jl = (int)(t / g.lmt);
      snprintf(b, sizeof b, "tile (%d,%d)", il * g.P + g.pr, jl * g.Q + g.pc);
    } else if (t < local_slots + 2) {
      snprintf(b, sizeof b, "block inverses %ld", t - local_slots);
    } else {
      snprintf(b, sizeof b, "receive buffer slot %ld", t - local_slots - 2);
    }
    return b;
  }

  int begin(int, int nt, int sem_per_wave) {
    sems.assign((size_t)nt * sem_per_wave * 32 + 32, 0);
    return 0;
  }
  // a launch on the streams of `mask`: behind their last launches and everything they have been told to wait for
  int add(unsigned mask, std::string name, Acc acc, std::vector<int> extra = {}) {
    Op op;
    op.name = std::move(name);
    op.acc = std::move(acc);
    std::set<int> d(extra.begin(), extra.end());
    for (int st = 0; st < ST_COUNT; ++st)
      if (mask & (1u << st)) {
        if (last_on[st] >= 0) d.insert(last_on[st]);
        d.insert(pending[st].begin(), pending[st].end());
        pending[st].clear();
      }
    d.erase(-1);
    op.deps.assign(d.begin(), d.end());
    ops.push_back(std::move(op));
    const int id = (int)ops.size() - 1;
    for (int st = 0; st < ST_COUNT; ++st)
      if (mask & (1u << st)) last_on[st] = id;
    return id;
  }
  int rec(int ev, int st) {
    std::set<int> d = pending[st];
    if (last_on[st] >= 0) d.insert(last_on[st]);
    ev_deps[ev] = std::move(d);
    return 0;
  }
  int wt(int st, int ev) {
    if (nwaits++ == drop_wait) {
      char b[120];
      snprintf(b, sizeof b, "(dropped: stream %d's wait for event %d = wave %d kind %d)", st, ev, ev / 15, ev % 15);
      dropped = b;
      return 0;
    }
    auto it = ev_deps.find(ev);
    if (it == ev_deps.end()) {
      char b[160];
      snprintf(b, sizeof b, "stream %d waits for event %d (wave %d, kind %d), which has not been recorded in this factorisation", st, ev,
               ev / 15, ev % 15);
      findings.push_back(b);
      return 0;
    }
    pending[st].insert(it->second.begin(), it->second.end());
    return 0;
  }
  int panel(int k, char *lkk, void *wv, char *tiles, int ntiles, int, int ev_head, const SyrkPipe *sy, const int *wait_sem, int) {
    std::vector<int> extra;
    if (wait_sem && ngates++ != drop_gate) {
      auto it = producer.find(wait_sem);
      if (it == producer.end()) findings.push_back("POTRF(" + std::to_string(k) + ") polls a counter nobody raises");
      else extra.push_back(it->second);
    }
    const bool flow = sy && sy->fc && sy->sflow && flow_applies(g.nbm);
    // the flow's join: an event recorded on ST_PANEL ahead of the factorisation launch, waited for by ST_CX (kernels.hip:
    // launch_panel_pipelined) -- counted and droppable like every other event wait
    std::vector<int> rows_extra = extra;
    if (flow && sy->join_flow) {
      if (nwaits++ == drop_wait) {
        dropped = "(dropped: the flow stream's join of wave " + std::to_string(k) + ")";
      } else {
        if (last_on[ST_PANEL] >= 0) rows_extra.push_back(last_on[ST_PANEL]);
        rows_extra.insert(rows_extra.end(), pending[ST_PANEL].begin(), pending[ST_PANEL].end());
      }
    }
    Acc pa;
    touch(pa, lkk, g.tile_bytes, true);
    touch(pa, wv, std::max<size_t>(1, g.winv_bytes), true);
    const int potrf = add(1u << ST_PANEL, "POTRF(" + std::to_string(k) + ")" + (flow ? " [flow: factor]" : ""), pa, extra);
    cost_of(potrf, 'P', t_panel);
    std::vector<int> after = {potrf};
    if (flow) {
      Acc ra;
      touch(ra, lkk, g.tile_bytes, true);
      touch(ra, wv, std::max<size_t>(1, g.winv_bytes), false);
      const int rows = add(1u << ST_CX, "POTRF(" + std::to_string(k) + ") [flow: row slabs]", ra, rows_extra);
      coop.insert({potrf, rows});
      after.push_back(rows);  // (the TRSM steps poll D[s], raised by the factorisation, and I[s], raised by the row slabs)
    }
    if (ntiles <= 0) return 0;
    Acc acc;
    touch(acc, lkk, g.tile_bytes, false);
    touch(acc, wv, std::max<size_t>(1, g.winv_bytes), false);
    touch(acc, tiles, (size_t)ntiles * g.tile_bytes, true);
    const int trsm = add(1u << ST_TRSM, "TRSM(" + std::to_string(k) + ")", acc, after);
    // (pipelined one step behind the POTRF: what is left behind it is one step and the tiles' products beyond what the
    // steps already did in its shadow)
    cost_of(trsm, 'T', t_panel / g.nbm + 0.5 * ntiles * t_tile / g.nbm);
    if (ev_head >= 0) ev_deps[ev_head] = {trsm};
    if (sy && sy->c && sy->su) {  // (one GPU: the tiles of a column are contiguous)
      Acc sa;
      touch(sa, tiles, g.tile_bytes, false);
      touch(sa, sy->c, g.tile_bytes, true);
      const int sl = add(1u << ST_U1, "SYRK slices(" + std::to_string(k) + ")", sa, {trsm});
      cost_of(sl, 'Y', t_panel / g.nbm);
      producer[sy->sem + 32 * (3 * g.nbm)] = sl;
    }
    return 0;
  }
  int trsm(int k, char *tiles, int ntiles, const char *lkk, const char *wv, int st) {
    Acc acc;
    touch(acc, lkk, g.tile_bytes, false);
    touch(acc, wv, std::max<size_t>(1, g.winv_bytes), false);
    touch(acc, tiles, (size_t)ntiles * g.tile_bytes, true);
    cost_of(add(1u << st, "TRSM(" + std::to_string(k) + ") of " + std::to_string(ntiles) + " tiles [received diagonal tile]", acc), 'T',
            std::max(t_panel / g.nbm * g.nbm * 0.5, 0.5 * ntiles * t_tile));
    return 0;
  }
  int diag_syrk(int k, int j, char *Cjj, const char *A, int st) {
    Acc acc;
    touch(acc, A, g.tile_bytes, false);
    touch(acc, Cjj, g.tile_bytes, true);
    cost_of(add(1u << st, "SYRK(" + std::to_string(j) + "," + std::to_string(j) + ") by panel " + std::to_string(k), acc), 'Y', t_round());
    return 0;
  }
  const char *ptile(const PanelRef &p, int i) const {
    return (const char *)p.base[i % g.P] + (size_t)(i / g.P - p.first[i % g.P]) * g.tile_bytes;
  }
  int update(int k1, int k2, int jlo, int jhi, int what, const PanelRef &p1, const PanelRef *p2, bool, int st) {
    const int nt = g.nt;
    jlo = std::min(jlo, nt), jhi = std::min(jhi, nt);
    Acc acc;
    std::set<int> rows;
    double tiles_eq = 0;  // GEMM tiles + half the SYRK tiles
    for (int j = jlo; j < jhi; ++j) {
      if (j % g.Q != g.pc) continue;
      for (int i = j; i < nt; ++i) {
        if (i % g.P != g.pr || !(i == j ? (what & 2) : (what & 1))) continue;
        touch(acc, tile(i / g.P, j / g.Q), g.tile_bytes, true);
        rows.insert(i), rows.insert(j);
        tiles_eq += i == j ? 0.5 * (1.0 + 1.0 / g.nbm) : 1.0;
      }
    }
    if (acc.empty()) return 0;
    const PanelRef *ps[2] = {&p1, (p2 && k2 >= 0) ? p2 : nullptr};
    for (const PanelRef *p : ps)
      if (p)
        for (int i : rows) touch(acc, ptile(*p, i), g.tile_bytes, false);
    char b[120];
    snprintf(b, sizeof b, "update columns [%d,%d) %sby panel %d%s", jlo, jhi, what == 3 ? "" : what == 1 ? "(off-diagonal) " : "(diagonal) ", k1,
             ps[1] ? (" and " + std::to_string(k2)).c_str() : "");
    const int np = ps[1] ? 2 : 1;
    cost_of(add(1u << st, b, acc), 'U', std::max(t_round() * np, tiles_eq * t_tile * np));
    return 0;
  }
  int update_col_small(int k, int st) {
    const int n = g.nt - k - 2;
    Acc acc;
    touch(acc, tile(k + 1, k), g.tile_bytes, false);
    touch(acc, tile(k + 2, k), (size_t)n * g.tile_bytes, false);
    touch(acc, tile(k + 2, k + 1), (size_t)n * g.tile_bytes, true);
    cost_of(add(1u << st, "column " + std::to_string(k + 1) + " (latency form) by panel " + std::to_string(k), acc), 'L', 0.6 * t_round());
    return 0;
  }
  // the transport of a grid: a send reads, a receive writes, on the stream it is issued on; each call also remembers its
  // channel, peer, size and group (chol_debug_comm_trace hands the whole launch graph out)
  struct Chan {
    TraceOps *o;
    int id;
  } chan[2] = {{this, 0}, {this, 1}};
  int ngroups[2] = {0, 0};
  static int t_begin(void *ctx) {
    Chan *c = (Chan *)ctx;
    ++c->o->ngroups[c->id];
    return 0;
  }
  static int t_end(void *) { return 0; }
  void mark_comm(int id, int ch, bool is_send, int peer, size_t bytes) {
    Op &op = ops[id];
    op.ch = ch, op.is_send = is_send, op.peer = peer, op.bytes = bytes, op.group = ngroups[ch] - 1;
  }
  // every launch of this rank, one per line: "<id> <channel or -1> <S|R|-> <peer> <bytes> <group> <dep,dep,...>"
  std::string op_graph() const {
    std::string out;
    char b[128];
    for (size_t i = 0; i < ops.size(); ++i) {
      const Op &op = ops[i];
      snprintf(b, sizeof b, "%zu %d %s %d %zu %d ", i, op.ch, op.ch < 0 ? "-" : (op.is_send ? "S" : "R"), op.peer, op.bytes, op.group);
      out += b;
      for (size_t d = 0; d < op.deps.size(); ++d) out += (d ? "," : "") + std::to_string(op.deps[d]);
      if (op.deps.empty()) out += "-";
      snprintf(b, sizeof b, " %c %.9f\n", op.kind, op.cost);
      out += b;
    }
    return out;
  }
  static int t_allreduce(void *, long long *) { return 0; }
  static int t_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream) {
    Chan *c = (Chan *)ctx;
    TraceOps *o = c->o;
    Acc acc;
    o->touch(acc, buf, bytes, false);
    o->mark_comm(o->add(1u << stream_id(stream), "send to rank " + std::to_string(peer), acc), c->id, true, peer, bytes);
    return 0;
  }
  static int t_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream) {
    Chan *c = (Chan *)ctx;
    TraceOps *o = c->o;
    Acc acc;
    o->touch(acc, buf, bytes, true);
    o->mark_comm(o->add(1u << stream_id(stream), "receive from rank " + std::to_string(peer), acc), c->id, false, peer, bytes);
    return 0;
  }
  // every pair of launches on one slot, one of them writing: the earlier one must be an ancestor of the later one
  int finish(int, int, const std::vector<std::pair<int, int>> &, int *info) {
    *info = 0;
    const int n = (int)ops.size();
    constexpr int MAXOPS = 32768;
    if (n > MAXOPS) {
      findings.push_back("too many launches for the checker");
      return 0;
    }
    std::vector<std::bitset<MAXOPS>> *reach = new std::vector<std::bitset<MAXOPS>>(n);
    for (int i = 0; i < n; ++i)
      for (int d : ops[i].deps) {
        (*reach)[i] |= (*reach)[d];
        (*reach)[i].set(d);
      }
    std::map<long, std::vector<std::pair<int, bool>>> by_res;
    for (int i = 0; i < n; ++i)
      for (auto &a : ops[i].acc) by_res[a.first].push_back({i, a.second});
    for (auto &kv : by_res) {
      auto &v = kv.second;
      for (size_t b = 1; b < v.size(); ++b)
        for (size_t a = 0; a < b; ++a) {
          if (v[a].first == v[b].first || (!v[a].second && !v[b].second)) continue;
          if (coop.count({v[a].first, v[b].first})) continue;
          if (!(*reach)[v[b].first].test(v[a].first)) {
            char buf[480];
            snprintf(buf, sizeof buf, "%s: '%s' (%s) and '%s' (%s) are not ordered", slot_name(kv.first).c_str(), ops[v[a].first].name.c_str(),
                     v[a].second ? "writes" : "reads", ops[v[b].first].name.c_str(), v[b].second ? "writes" : "reads");
            if (findings.size() < 64) findings.push_back(buf);
          }
        }
    }
    delete reach;
    return 0;
  }
};

}  // namespace

// Test hook (include/cholmi.h).  Returns the number of findings (0: every conflicting pair of launches is ordered),
// < 0 when the walker itself failed; `report` receives the findings, one per line, and a last line with counts.
static int schedule_check_impl(int nt, int mb, int p, int q, int rank, double t_tile, double t_panel, int profiling, char *report, int cap,
                               bool comm_trace) {
  if (nt <= 0 || mb < MACRO || mb % MACRO) return chol_internal_fail(-1, "schedule_check: nt > 0, mb a multiple of 128");
  if (p < 1 || q < 1 || p > MAXP || rank < 0 || rank >= p * q) return chol_internal_fail(-2, "schedule_check: grid");
  WaveGeo g;
  g.init(nt, mb, p, q, rank, 8);
  TraceOps ops(g, profiling != 0);
  ops.t_tile = t_tile, ops.t_panel = t_panel;
  WaveCalib c;
  c.t_tile = t_tile, c.t_panel = t_panel;
  // (self-test of the checker: CHOLMI_CHECK_DROP_WAIT / _DROP_GATE = n makes it ignore the n-th event wait / counter edge the
  // walker asks for -- a schedule with that dependency missing, which it has to report unless the edge was redundant)
  if (const char *e = getenv("CHOLMI_CHECK_DROP_WAIT")) ops.drop_wait = atoi(e);
  if (const char *e = getenv("CHOLMI_CHECK_DROP_GATE")) ops.drop_gate = atoi(e);
  WaveComm cm;
  for (int ch = 0; ch < 2; ++ch) {
    cm.ch[ch].ctx = &ops.chan[ch];
    cm.ch[ch].group_begin = TraceOps::t_begin;
    cm.ch[ch].send = TraceOps::t_send;
    cm.ch[ch].recv = TraceOps::t_recv;
    cm.ch[ch].group_end = TraceOps::t_end;
    cm.ch[ch].allreduce_max = TraceOps::t_allreduce;
  }
  Walker<TraceOps> w(ops, g, p * q > 1 ? &cm : nullptr, c);
  long long info = 0;
  int rc = w.setup();
  if (!rc) rc = w.run(&info);
  if (rc) return rc < 0 ? rc : -rc;
  if (comm_trace) {  // (the transport calls of this rank, in issue order; the tail if the buffer is short is NOT wanted here)
    const std::string gr = ops.op_graph();
    if (!report || (size_t)cap <= gr.size()) return chol_internal_fail(-3, "comm_trace: buffer too small");
    memcpy(report, gr.c_str(), gr.size() + 1);
    return (int)ops.findings.size();
  }
  std::string out;
  for (auto &f : ops.findings) out += f + "\n";
  char tail[200];
  snprintf(tail, sizeof tail, "%zu launches, %d event waits, %d counter edges, %d flow-form waves, %zu findings\n", ops.ops.size(), ops.nwaits,
           ops.ngates, w.flow_waves, ops.findings.size());
  out += tail;
  if (!ops.dropped.empty()) out += ops.dropped + "\n";
  if (report && cap > 0) snprintf(report, (size_t)cap, "%s", out.size() < (size_t)cap ? out.c_str() : out.substr(out.size() - cap + 1).c_str());
  return (int)ops.findings.size();
}
extern "C" int chol_debug_schedule_check_grid(int nt, int mb, int p, int q, int rank, double t_tile, double t_panel, int profiling,
                                              char *report, int cap) {
  return schedule_check_impl(nt, mb, p, q, rank, t_tile, t_panel, profiling, report, cap, false);
}
// The launch graph of rank `rank` of a p x q grid for one factorisation as text, one launch per line in issue order:
// "<id> <channel or -1> <S|R|-> <peer> <bytes> <group> <dep,dep,...|-> <kind> <rough duration [s]>" -- every kernel launch and every transport call with
// what it waits for (stream order, events, counters), the transport calls with channel, peer, size and the group they
// were issued in.  tests/test_schedule_check.py replays the graphs of ALL ranks together under RCCL's rules.
extern "C" int chol_debug_comm_trace(int nt, int mb, int p, int q, int rank, double t_tile, double t_panel, char *out, int cap) {
  return schedule_check_impl(nt, mb, p, q, rank, t_tile, t_panel, 0, out, cap, true);
}
extern "C" int chol_debug_schedule_check(int nt, int mb, double t_tile, double t_panel, int profiling, char *report, int cap) {
  return chol_debug_schedule_check_grid(nt, mb, 1, 1, 0, t_tile, t_panel, profiling, report, cap);
}
