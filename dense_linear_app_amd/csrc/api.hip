// C ABI of libcholmi.so (include/cholmi.h): process-global context, Chameleon-style
// descriptors, the synchronous tile operations the reference worker calls
// (worker_distrib.cpp:238, 323, 416, 511) and the whole-matrix tiled POTRF the
// reference driver calls (v6_test.c:56), run as the reference client's wave DAG
// (client_distrib.cpp:506-565) on two HIP streams with one wave of lookahead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cholmi.h"
#include "cholmi_internal.h"

using namespace cholmi;

extern "C" int chol_internal_dist_potrf(chol_desc *d, int rank);  // dist.hip
extern "C" void chol_internal_dist_finalize(void);

namespace {

constexpr int SEM_SLOTS = 16384;          // device-side counters: 3 mb/128 + 1 per tile column ...
constexpr int TILE_SEM_SETS = 8;          // ... plus rotating sets of 32 for the single-tile POTRF's fused in-tile steps
constexpr int SEM_INTS = (SEM_SLOTS + TILE_SEM_SETS * 32) * 32;  // ... each on a 128-byte line of its own

struct Ctx {
  bool inited = false;
  int device = -1;
  int rank = 0, nranks = 1;
  hipStream_t s_main = nullptr, s_panel = nullptr, s_trsm = nullptr, s_u1 = nullptr;
  void *winv = nullptr;  // inverses of the 128x128 diagonal blocks of L(k,k)
  size_t winv_bytes = 0;
  int *d_info = nullptr;
  int *d_ytab = nullptr;  // per-CU yield requests (kernels.hip: cooperative CU hand-over)
  int *d_sem = nullptr;   // device-side dependency counters of the panel chain (kernels.hip: sem_wait), or null
  unsigned tile_sem_next = 0;
  double *d_acc = nullptr;
  void *stage[3] = {nullptr, nullptr, nullptr};
  size_t stage_bytes[3] = {0, 0, 0};
  std::vector<hipEvent_t> events;
  bool profiling = false;
  // stats of the last whole-matrix potrf
  double total_ms = 0, update_ms = 0, update_flops = 0;
  int update_launches = 0;
  std::string last_error;
};
Ctx g;
std::mutex g_mu;

int fail_hip(hipError_t e, const char *what, int line) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s failed at api.hip:%d: %s", what, line, hipGetErrorString(e));
  g.last_error = buf;
  return CHOL_ERR_HIP;
}
#define HIPCHECK(call)                                        \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return fail_hip(e_, #call, __LINE__); \
  } while (0)

int fail(int code, const char *msg) {
  g.last_error = msg;
  return code;
}

inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

// counters for one single-tile POTRF (launch_potrf_tile): a set of 32, rotating so that consecutive
// factorisations on different streams never share one
int *tile_sems() {
  return g.d_sem ? g.d_sem + (size_t)(SEM_SLOTS + 32 * (g.tile_sem_next++ % TILE_SEM_SETS)) * 32 : nullptr;
}

int ensure_stage(int idx, size_t bytes) {
  if (g.stage_bytes[idx] >= bytes) return 0;
  if (g.stage[idx]) HIPCHECK(hipFree(g.stage[idx]));
  g.stage[idx] = nullptr;
  g.stage_bytes[idx] = 0;
  HIPCHECK(hipMalloc(&g.stage[idx], bytes));
  g.stage_bytes[idx] = bytes;
  return 0;
}

int ensure_events(size_t n) {
  while (g.events.size() < n) {
    hipEvent_t e;
    HIPCHECK(hipEventCreate(&e));
    g.events.push_back(e);
  }
  return 0;
}

bool is_device_ptr(const void *p) {
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: not an error for us
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

LocalMat local_mat(const chol_desc *d, void *base) {
  LocalMat L;
  L.base = base;
  L.lmt = d->lmt;
  L.P = d->p;
  L.Q = d->q;
  L.mb = d->mbi;
  L.bsiz = d->bsizi;
  return L;
}

template <typename T>
T *local_tile(const chol_desc *d, void *base, int I, int J) {
  return reinterpret_cast<T *>(base) + ((long)(I / d->p) + (long)(J / d->q) * d->lmt) * (long)d->bsizi;
}

// the context holds the inverses of at most 32 diagonal 128-blocks (tiles up to 4096): every entry
// that factors, inverts or solves with a tile checks this before any launch writes winv
bool winv_fits(const chol_desc *d) {
  return (size_t)(roundup(d->mbi, MACRO) / MACRO) * MACRO * MACRO * d->esize <= g.winv_bytes;
}
#define CHECK_WINV(d, what) \
  if (!winv_fits(d)) return fail(CHOL_ERR_NOT_SUPPORTED, what ": tile size above 4096")

bool single_tile_square(const chol_desc *d) {
  return d->mt == 1 && d->nt == 1 && d->m == d->n && d->mb == d->nb && d->m == d->mb &&
         d->p == 1 && d->q == 1 && d->i == 0 && d->j == 0;
}

// ---- staging of one B x B tile (any B, host or device) into a padded device tile
struct Staged {
  void *dev = nullptr;  // device tile with ld = ldp
  int ldp = 0;
  bool in_place = false;
};

template <typename T>
int stage_in(const chol_desc *d, int slot, bool identity_pad, Staged *out) {
  const int B = d->mb, Bp = roundup(B, MACRO);
  out->ldp = Bp;
  if (d->on_device && B == Bp) {
    out->dev = d->mat;
    out->in_place = true;
    return 0;
  }
  const size_t bytes = (size_t)Bp * Bp * sizeof(T);
  int rc = ensure_stage(slot, bytes);
  if (rc) return rc;
  T *dst = reinterpret_cast<T *>(g.stage[slot]);
  if (B != Bp) HIPCHECK(hipMemsetAsync(dst, 0, bytes, g.s_main));
  HIPCHECK(hipMemcpy2DAsync(dst, (size_t)Bp * sizeof(T), d->mat, (size_t)B * sizeof(T),
                            (size_t)B * sizeof(T), B, hipMemcpyDefault, g.s_main));
  if (identity_pad) launch_pad_identity<T>(g.s_main, dst, B, Bp);
  out->dev = dst;
  out->in_place = false;
  return 0;
}

template <typename T>
int stage_out(const chol_desc *d, const Staged &st) {
  if (st.in_place) return 0;
  const int B = d->mb;
  HIPCHECK(hipMemcpy2DAsync(d->mat, (size_t)B * sizeof(T), st.dev, (size_t)st.ldp * sizeof(T),
                            (size_t)B * sizeof(T), B, hipMemcpyDefault, g.s_main));
  return 0;
}

int read_info(int *info) {
  HIPCHECK(hipMemcpy(info, g.d_info, sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

// the work-list segments of the tiles in columns [jlo, jhi)
struct ColRange {
  int off, na, offb, nb;
};
static inline ColRange col_range(const chol_desc *d, int jlo, int jhi) {
  ColRange r;
  r.off = d->ge[jhi];
  r.na = d->ge[jlo] - d->ge[jhi];
  r.offb = d->n_off + d->gd[jhi];
  r.nb = d->gd[jlo] - d->gd[jhi];
  return r;
}

// ---- whole-matrix right-looking tiled Cholesky on one GPU --------------------
// Reference order (C2:506-565): POTRF(k); TRSM(i,k), i > k; SYRK / GEMM (i,j,k), k < j <= i.  Here, per
// wave k, four streams:
//   s_panel (high)  POTRF(k) as mb/128 diagonal-block steps
//   s_trsm  (high)  the TRSM steps of panel k, one 128-column step behind the POTRF steps
//   s_u1    (mid)   U1(k): column k+1 by panel k -- the diagonal tile (k+1,k+1) first (POTRF(k+1) waits
//                   for nothing else), then the rest of the column (TRSM(k+1) waits for that)
//   s_main  (low)   U2(k): the columns beyond, beside U1(k): the two touch different tiles, U1's
//                   blocks go first and U2's fill whatever U1 -- at most one round of workgroups
//                   on a mid-size matrix -- leaves idle
// U1(k) waits for U2(k-1), whose range includes column k+1; U2(k) follows U2(k-1) in stream order.
// Three regimes by the length of a wave's update against its panel chain: panels in pairs (two per pass of
// the far columns) while it is more than two chains long; the columns beyond k+1 as a near and a far launch
// on s_u1 / s_main in the mid waves; and, once it is shorter than the chain, the counter-linked form -- TRSM
// steps, SYRK slices and the next POTRF launched ahead of time and polling device-side counters, no stream
// event on the chain (cholmi_internal.h: SyrkPipe; kernels.hip: sem_wait).
template <typename T>
int potrf_full_device(chol_desc *d, void *base) {
  const int nt = d->nt, mb = d->mbi;
  const long bsiz = d->bsizi;
  T *M = reinterpret_cast<T *>(base);
  T *winv = reinterpret_cast<T *>(g.winv);
  const int nbm = mb / MACRO;
  enum { E_PANEL, E_U1D, E_U1R, E_U2, E_P0, E_P1, E_NEAR, E_PN0, E_PN1, E_PER_WAVE };
  enum { F_START, F_STOP, F_JOIN, F_WAVE, F_TRSM, F_U1END, F_HEAD, F_COLS, F_FIXED };
  int rc = ensure_events((size_t)E_PER_WAVE * nt + F_FIXED + nbm + 1);
  if (rc) return rc;
  auto ev = [&](int k, int which) { return g.events[(size_t)E_PER_WAVE * k + which]; };
  hipEvent_t *fixed = &g.events[(size_t)E_PER_WAVE * nt];
  hipEvent_t *ev_steps = fixed + F_FIXED;
  static const bool concurrent = !(getenv("CHOLMI_U1_CONCURRENT") && atoi(getenv("CHOLMI_U1_CONCURRENT")) == 0);
  // tiles up to this edge are updated by two panels per pass (short K-loops pay the per-block
  // prologue / epilogue / C traffic twice as often)
  static const int pair_max_mb = getenv("CHOLMI_PAIR_MAX_MB") ? atoi(getenv("CHOLMI_PAIR_MAX_MB")) : 1024;
  // ... while a wave's update is at least this many panel chains long: deferring half the updates
  // leaves the chip short of work once the panel chain is what a wave waits for
  static const double pair_fac = getenv("CHOLMI_PAIR_FACTOR") ? atof(getenv("CHOLMI_PAIR_FACTOR")) : 2.0;
  bool paired = false, cols_pending = false;  // paired: this wave belongs to a pair (decided at its even wave)
  bool had_pairs = false;
  // experiment, off: keep the chain-critical launches of a chain-bound wave (the head tile's last TRSM step, the
  // SYRK on the next diagonal tile) on s_panel instead of crossing streams.  Measured -1 ... -2 % at N <= 16384
  // (the hops it removes were overlapped work, not latency); CHOLMI_CHAIN_INSTREAM=1 enables it.
  static const bool chain_enabled = getenv("CHOLMI_CHAIN_INSTREAM") && atoi(getenv("CHOLMI_CHAIN_INSTREAM")) != 0;
  static const double yfac = getenv("CHOLMI_YIELD_FACTOR") ? atof(getenv("CHOLMI_YIELD_FACTOR")) : 3.0;
  static const bool syrk_pipe = !(getenv("CHOLMI_SYRK_PIPE") && atoi(getenv("CHOLMI_SYRK_PIPE")) == 0);
  const bool trsm_fused_on = cholmi::g_trsm_fused_min > 0;
  static const double pipe_fac = getenv("CHOLMI_PIPE_FACTOR") ? atof(getenv("CHOLMI_PIPE_FACTOR")) : 0.7;
  int open_bracket = -1;                      // odd wave whose profiling bracket is still open
  // Plain waves of a few rounds of workgroups: the columns beyond k+1 go out as TWO launches, the near
  // columns [k+2, bnd) on s_u1 behind the column-(k+1) launch and the far ones [bnd, nt) on s_main.  With a
  // boundary that stays put for several waves each half depends on its own predecessor only (far(k+1) is a
  // subset of far(k), near(k+1) of near(k)), so the last, partly filled round of one launch runs beside full
  // rounds of the other chain's next launch instead of beside nothing; and column k+1 -- the next panel --
  // waits for near(k-1) only.  The boundary moves (then near(k) also waits for far(k-1)) when the near part
  // has shrunk under 30 % of the wave.
  static const double halves_max_rounds =
      getenv("CHOLMI_HALVES_MAX_ROUNDS") ? atof(getenv("CHOLMI_HALVES_MAX_ROUNDS")) : 24.0;
  int bnd = -1;
  bool prev_halves = false;
  std::vector<int> halves_waves;
  HIPCHECK(hipMemsetAsync(g.d_info, 0, sizeof(int), g.s_main));
  if (g.d_ytab) HIPCHECK(hipMemsetAsync(g.d_ytab, 0, YTAB_ENTRIES * sizeof(int), g.s_main));
  // device-side edges of the panel chain (kernels.hip, sem_wait; cholmi_internal.h, SyrkPipe): 3 nbm + 1
  // counters per wave, the last one = workgroups of the last SYRK slice on tile (k+1,k+1)
  const int sem_per_wave = 3 * nbm + 1;
  const bool flags = g.d_sem && (long)nt * sem_per_wave <= SEM_SLOTS && concurrent;
  if (flags) HIPCHECK(hipMemsetAsync(g.d_sem, 0, (size_t)nt * sem_per_wave * 32 * sizeof(int), g.s_main));
  auto sem = [&](int k, int which) { return g.d_sem + ((size_t)sem_per_wave * k + which) * 32; };
  const int *wait_sem = nullptr;  // what this wave's first diagonal-block step polls, when the last wave raised it
  int wait_target = 0;
  HIPCHECK(hipEventRecord(fixed[F_START], g.s_main));
  HIPCHECK(hipStreamWaitEvent(g.s_panel, fixed[F_START], 0));
  HIPCHECK(hipStreamWaitEvent(g.s_u1, fixed[F_START], 0));
  HIPCHECK(hipStreamWaitEvent(g.s_trsm, fixed[F_START], 0));  // (its kernels may poll counters zeroed above)
  const LocalMat C = local_mat(d, base);
  const double b3 = (double)mb * mb * mb;
  double upd_flops = 0;
  int upd_launches = 0;
  for (int k = 0; k < nt; ++k) {
    // POTRF on the panel stream, the TRSM steps pipelined behind it on their own stream, which
    // also needs the rest of column k updated by panel k-1  (C2:510-535)
    T *lkk = M + ((long)k + (long)k * nt) * bsiz;
    // (s_trsm needs no event for the start of the wave: its first step waits for the event recorded on
    // s_panel behind the first diagonal-block step, and a record on s_panel costs the chain ~7 us)
    if (k > 0) HIPCHECK(hipStreamWaitEvent(g.s_trsm, ev(k - 1, E_U1R), 0));
    // chain: this wave is about as long as its panel chain -- keep the chain's own launches on s_panel
    const double mrem = nt - 1 - k;
    const bool chain = chain_enabled && cholmi::g_intile_small && k + 1 < nt &&
                       mrem * (mrem + 1) / 2 * (2.0 * b3 / 65e12) < yfac * (nbm * 130e-6 * 1.5);
    // block inverses of L(k,k): two workspaces alternating by wave, so that POTRF(k+1) may overwrite
    // its set while TRSM(k) still reads the other
    T *winv_k = winv + (size_t)(k & 1) * (g.winv_bytes / sizeof(T));
    if ((k & 1) == 0) {
      const double m = nt - 1 - k;
      paired = mb <= pair_max_mb && k + 2 < nt &&
               m * (m + 1) / 2 * (2.0 * b3 / 65e12) >= pair_fac * (nbm * 130e-6 * 1.5);
    }
    // Plain (unpaired) wave whose panel chain is (nearly) critical: the SYRK on tile (k+1,k+1) follows
    // the head tile's TRSM steps slice by slice and the chain's cross-stream edges are device-side
    // counters (SyrkPipe).
    const bool plain_yield = mrem * (mrem + 1) / 2 * (2.0 * b3 / 65e12) < yfac * (nbm * 130e-6 * 1.5);
    hipStream_t su_k = concurrent ? g.s_u1 : g.s_main;
    // (only while the update is shorter than about a panel chain: the polling workgroups hold CU slots the
    // update would otherwise use -- measured +10 ... +20 % on the waves between pipe_fac and the yield threshold)
    const bool chain_bound = mrem * (mrem + 1) / 2 * (2.0 * b3 / 65e12) < pipe_fac * (nbm * 130e-6 * 1.5);
    const bool pipe = syrk_pipe && flags && !paired && !chain && k + 1 < nt && cholmi::g_intile_small &&
                      !trsm_fused_on && plain_yield && chain_bound;
    SyrkPipe sy;
    if (pipe) {
      // the tile's earlier writers: U2(k-1), whose range includes column k+1 (or, behind the paired
      // phase, the column launches of the last pair, which precede this on s_u1 / are joined here)
      if (concurrent && k > 0) HIPCHECK(hipStreamWaitEvent(su_k, ev(k - 1, E_U2), 0));
      if (!concurrent && cols_pending) HIPCHECK(hipStreamWaitEvent(su_k, fixed[F_COLS], 0));
      sy.c = M + ((long)(k + 1) + (long)(k + 1) * nt) * bsiz;
      sy.su = su_k;
      sy.sem = sem(k, 0);
    }
    launch_panel_pipelined<T>(g.s_panel, g.s_trsm, ev_steps, lkk, mb, winv_k, g.d_info, k * mb, lkk + bsiz, bsiz,
                              nt - 1 - k, fixed[F_HEAD], chain, k > 0 ? ev(k - 1, E_U1R) : nullptr,
                              pipe ? &sy : nullptr, wait_sem, wait_target);
    // with device-side edges s_panel waits for no event between waves: POTRF(k+1)'s first step polls the
    // last slice's counter, which also stands behind TRSM(k) (same stream, earlier), so POTRF(k+2) may
    // reuse TRSM(k)'s workspace
    const bool by_flags = pipe;
    wait_sem = by_flags ? sem(k, 3 * nbm) : nullptr;
    wait_target = (mb / 64) * (mb / 64 + 1) / 2;
    // TRSM(k) complete = panel k ready (s_trsm has waited for every POTRF step, and for the head tile's
    // in-stream step in chain mode).
    HIPCHECK(hipEventRecord(ev(k, E_PANEL), g.s_trsm));
    // In chain mode s_panel does not wait for it: POTRF(k+1) needs the SYRK only, and TRSM(k) is over
    // before POTRF(k+2) reuses its workspace (the head tile's last step of TRSM(k+1) waits for the
    // earlier ones, which follow TRSM(k) in stream order -- with more than one step per tile).
    if ((!chain || nbm == 1) && !by_flags) HIPCHECK(hipStreamWaitEvent(g.s_panel, ev(k, E_PANEL), 0));
    if (k + 1 >= nt) break;
    // trailing update (C2:540-560)
    auto panel_ref = [&](int kk) {
      PanelRef pr;
      memset(&pr, 0, sizeof pr);
      pr.P = 1;
      pr.base[0] = M + (long)kk * nt * bsiz;
      return pr;
    };
    const PanelRef pan = panel_ref(k);
    if (paired) {
      // Panels in pairs (k-1, k), k odd: the far columns' update by the even panel is deferred and
      // applied together with the odd one in ONE pass of twice the K (k_trail_update, npan = 2).
      //   even k:  U1(k)  = column k+1 by panel k                                       (s_u1)
      //   odd  k:  U1'(k) = column k+1, Ca = column k+2, Cb = column k+3 by panels k-1, k  (s_u1, in this order)
      //            big(k) = the columns from k+4 on by panels k-1, k                      (s_main, beside them)
      // Every column is written by launches of s_u1 in program order, except by big(); the first
      // launches of s_u1 on a column big(k) covers are Ca / Cb of wave k+2, which wait for it.
      // POTRF(k+1) waits for the SYRKs on (k+1,k+1) only, TRSM(k+1) for the rest of column k+1.
      const bool odd = (k & 1) != 0;
      const PanelRef prev = panel_ref(odd ? k - 1 : k);
      const PanelRef *p2 = odd ? &pan : nullptr;       // launches: first `prev`, then `pan` when odd
      const PanelRef &p1 = odd ? prev : pan;
      auto rng = [&](int jlo, int jhi) { return col_range(d, jlo < nt ? jlo : nt, jhi < nt ? jhi : nt); };
      const ColRange c1 = rng(k + 1, k + 2), ca = rng(k + 2, k + 3), cb = rng(k + 3, k + 4), big = rng(k + 4, nt);
      const int wave_tiles = c1.na + c1.nb + (odd ? ca.na + ca.nb + cb.na + cb.nb + big.na + big.nb : 0);
      const double t_tile = 2.0 * b3 / 65e12 * (odd ? 2 : 1), t_panel = nbm * 130e-6 * 1.5;
      const bool yield = (double)wave_tiles * t_tile < yfac * t_panel * (odd ? 2 : 1);
      hipStream_t su = concurrent ? g.s_u1 : g.s_main;
      T *ckk = M + ((long)(k + 1) + (long)(k + 1) * nt) * bsiz;
      if (cholmi::g_intile_small) {
        if (odd) launch_diag_syrk<T>(su, ckk, M + ((long)(k + 1) + (long)(k - 1) * nt) * bsiz, mb);
        HIPCHECK(hipStreamWaitEvent(su, fixed[F_HEAD], 0));  // the head tile L(k+1,k) is all the last SYRK needs
        launch_diag_syrk<T>(su, ckk, M + ((long)(k + 1) + (long)k * nt) * bsiz, mb);
      } else {
        HIPCHECK(hipStreamWaitEvent(su, ev(k, E_PANEL), 0));
        launch_trail_update<T>(su, C, d->d_list, c1.off, 0, c1.offb, c1.nb, p1, yield, p2);
      }
      HIPCHECK(hipEventRecord(ev(k, E_U1D), su));
      HIPCHECK(hipStreamWaitEvent(g.s_panel, ev(k, E_U1D), 0));
      HIPCHECK(hipStreamWaitEvent(su, ev(k, E_PANEL), 0));
      if (!odd && open_bracket < 0 && g.profiling) HIPCHECK(hipEventRecord(ev(k, E_P0), su));
      launch_trail_update<T>(su, C, d->d_list, c1.off, c1.na, c1.offb, 0, p1, yield, p2);
      HIPCHECK(hipEventRecord(ev(k, E_U1R), su));
      int timed = 0;
      double fl = 0;
      if (odd) {
        if (c1.na > 0) ++timed, fl += 2.0 * c1.na;
        if (k >= 2) HIPCHECK(hipStreamWaitEvent(su, ev(k - 2, E_U2), 0));  // big(k-2) covered these columns
        launch_trail_update<T>(su, C, d->d_list, ca.off, ca.na, ca.offb, ca.nb, p1, yield, p2);
        launch_trail_update<T>(su, C, d->d_list, cb.off, cb.na, cb.offb, cb.nb, p1, yield, p2);
        if (ca.na + ca.nb > 0) ++timed, fl += 2.0 * ca.na + ca.nb;
        if (cb.na + cb.nb > 0) ++timed, fl += 2.0 * cb.na + cb.nb;
        HIPCHECK(hipEventRecord(fixed[F_COLS], su));
        cols_pending = true;
        had_pairs = true;
        HIPCHECK(hipStreamWaitEvent(g.s_main, ev(k, E_PANEL), 0));
        if (g.profiling) HIPCHECK(hipEventRecord(ev(k, E_P0), g.s_main));
        launch_trail_update<T>(g.s_main, C, d->d_list, big.off, big.na, big.offb, big.nb, p1, yield, p2);
        if (big.na + big.nb > 0) ++timed, fl += 2.0 * big.na + big.nb;
        HIPCHECK(hipEventRecord(ev(k, E_U2), g.s_main));
        // the bracket [P0(k), P1(k)] covers every k_trail_update launch of the pair's update: the
        // two-panel launches of this wave, which start together, and U1(k+1), which runs beside
        // big(k); it is closed at the next wave
        open_bracket = k;
        upd_launches += timed;
        upd_flops += 2.0 * fl * b3;  // two panels per pass
      } else {
        HIPCHECK(hipEventRecord(ev(k, E_U2), g.s_main));
        if (g.profiling) {
          if (open_bracket >= 0) {
            // waiting for the column launches and for U1(k) on s_main constrains nothing: big(k+1)
            // needs panel k+1, which comes after all of them
            HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_COLS], 0));
            HIPCHECK(hipStreamWaitEvent(g.s_main, ev(k, E_U1R), 0));
            HIPCHECK(hipEventRecord(ev(open_bracket, E_P1), g.s_main));
            HIPCHECK(hipEventRecord(ev(k, E_P0), g.s_main));
            HIPCHECK(hipEventRecord(ev(k, E_P1), g.s_main));
          } else {
            HIPCHECK(hipEventRecord(ev(k, E_P1), su));  // the very first wave: U1(0) alone, bracketed on its stream
          }
        }
        open_bracket = -1;
        if (c1.na > 0) ++upd_launches, upd_flops += 2.0 * c1.na * b3;
      }
      continue;
    }
    if (open_bracket >= 0) {  // the paired phase ended on an odd wave: close its bracket
      if (g.profiling) {
        HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_COLS], 0));
        HIPCHECK(hipEventRecord(ev(open_bracket, E_P1), g.s_main));
      }
      open_bracket = -1;
    }
    const int j2 = k + 2 <= nt ? k + 2 : nt;
    const ColRange r1 = col_range(d, k + 1, j2), r2 = col_range(d, j2, nt);
    // Give CUs to the next panel's guest workgroups only when that panel is on the critical
    // path, i.e. when this wave's update is not much longer than a panel (POTRF ~ (mb/128) x
    // 130 us, one tile update ~ 2 mb^3 / 65 TFLOP/s); otherwise the polling is pure cost.
    const double t_tile = 2.0 * b3 / 65e12;
    const double t_panel = nbm * 130e-6 * 1.5;
    const bool yield = (double)(r1.na + r1.nb + r2.na + r2.nb) * t_tile < yfac * t_panel;
    static const bool split_always = getenv("CHOLMI_SPLIT_U1") != nullptr;  // diagnostic
    const bool split = yield || split_always;
    hipStream_t su = concurrent ? g.s_u1 : g.s_main;
    const bool syrk_instream = chain && split && concurrent && !paired;
    if (syrk_instream) {
      // the SYRK that releases POTRF(k+1) directly behind the head tile's TRSM on s_panel; the earlier
      // writers of tile (k+1,k+1) -- the previous waves' updates -- finished long ago in this regime
      if (k > 0) {
        HIPCHECK(hipStreamWaitEvent(g.s_panel, ev(k - 1, E_U2), 0));
        HIPCHECK(hipStreamWaitEvent(g.s_panel, ev(k - 1, E_U1R), 0));
      }
      if (had_pairs) HIPCHECK(hipStreamWaitEvent(g.s_panel, fixed[F_COLS], 0));
      launch_diag_syrk<T>(g.s_panel, M + ((long)(k + 1) + (long)(k + 1) * nt) * bsiz,
                          M + ((long)(k + 1) + (long)k * nt) * bsiz, mb);
      HIPCHECK(hipEventRecord(ev(k, E_U1D), g.s_panel));
    }
    // the SYRK on (k+1,k+1) needs the head tile L(k+1,k) only; everything else the whole panel
    if (!pipe)
      HIPCHECK(hipStreamWaitEvent(su, (split && cholmi::g_intile_small && !syrk_instream) ? fixed[F_HEAD] : ev(k, E_PANEL), 0));
    if (cols_pending) {  // first plain wave after the paired phase: Cb of the last pair wrote column k+2
      HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_COLS], 0));
      cols_pending = false;
    }
    // halves: see above
    const int tiles2 = r2.na + r2.nb;
    bool halves = halves_max_rounds > 0 && concurrent && !pipe && !syrk_instream && split && nt - 1 - k >= 6 &&
                  (double)tiles2 * nbm * nbm / 512.0 < halves_max_rounds;
    bool moved = false;
    if (halves) {
      auto tiles_in = [&](int jlo, int jhi) {
        const ColRange r = col_range(d, jlo, jhi);
        return r.na + r.nb;
      };
      if (!prev_halves || bnd <= k + 2 || tiles_in(k + 2, bnd) * 10 < tiles2 * 3) {
        int b = k + 3;
        while (b < nt - 1 && tiles_in(k + 2, b) * 2 < tiles2) ++b;
        // (the boundary only ever moves right, far(k) stays a subset of far(k-1); should it not, far(k) waits
        // for near(k-1) as well)
        if (prev_halves && b < bnd) HIPCHECK(hipStreamWaitEvent(g.s_main, ev(k - 1, E_NEAR), 0));
        bnd = b;
        moved = true;
      }
      if (bnd >= nt) halves = false;
    }
    if (prev_halves && !halves) HIPCHECK(hipStreamWaitEvent(g.s_main, ev(k - 1, E_NEAR), 0));  // U2(k) covers near(k-1)'s columns
    if (concurrent) {
      // column k+1 was in U2(k-1)'s range -- or in near(k-1)'s, which precedes this on s_u1
      if (k > 0 && !prev_halves) HIPCHECK(hipStreamWaitEvent(su, ev(k - 1, E_U2), 0));
      HIPCHECK(hipStreamWaitEvent(g.s_main, ev(k, E_PANEL), 0));
    } else if (g.profiling) {
      HIPCHECK(hipEventRecord(ev(k, E_P0), g.s_main));
    }
    int timed = 0;  // k_trail_update launches inside this wave's profiling bracket
    if (split) {
      // the panel chain is (nearly) critical: the diagonal tile (k+1,k+1) alone first, POTRF(k+1)
      // needs nothing else; then the rest of column k+1, which TRSM(k+1) needs
      if (syrk_instream || pipe) {
        // (done above: on s_panel / by launch_panel_pipelined)
      } else if (cholmi::g_intile_small) {
        launch_diag_syrk<T>(su, M + ((long)(k + 1) + (long)(k + 1) * nt) * bsiz,
                            M + ((long)(k + 1) + (long)k * nt) * bsiz, mb);
        HIPCHECK(hipEventRecord(ev(k, E_U1D), su));
      } else {
        launch_trail_update<T>(su, C, d->d_list, r1.off, 0, r1.offb, r1.nb, pan, yield);
        ++timed;
        HIPCHECK(hipEventRecord(ev(k, E_U1D), su));
      }
      HIPCHECK(hipStreamWaitEvent(su, ev(k, E_PANEL), 0));
      if (halves && g.profiling) HIPCHECK(hipEventRecord(ev(k, E_PN0), su));
      launch_trail_update<T>(su, C, d->d_list, r1.off, r1.na, r1.offb, 0, pan, yield);
      if (r1.na > 0) ++timed;
      HIPCHECK(hipEventRecord(ev(k, E_U1R), su));
      if (halves) {
        const ColRange rn = col_range(d, k + 2, bnd);
        if (moved && prev_halves) HIPCHECK(hipStreamWaitEvent(su, ev(k - 1, E_U2), 0));  // columns taken over from far(k-1)
        launch_trail_update<T>(su, C, d->d_list, rn.off, rn.na, rn.offb, rn.nb, pan, yield);
        ++timed;
        HIPCHECK(hipEventRecord(ev(k, E_NEAR), su));
        if (g.profiling) HIPCHECK(hipEventRecord(ev(k, E_PN1), su));
      }
    } else {
      // the update dwarfs the panel: one launch for the whole column (one tail less per wave)
      launch_trail_update<T>(su, C, d->d_list, r1.off, r1.na, r1.offb, r1.nb, pan, yield);
      ++timed;
      HIPCHECK(hipEventRecord(ev(k, E_U1D), su));
      HIPCHECK(hipEventRecord(ev(k, E_U1R), su));
    }
    if (!syrk_instream && !by_flags) HIPCHECK(hipStreamWaitEvent(g.s_panel, ev(k, E_U1D), 0));
    if (concurrent && g.profiling) HIPCHECK(hipEventRecord(ev(k, E_P0), g.s_main));
    if (halves) {
      const ColRange rf = col_range(d, bnd, nt);
      launch_trail_update<T>(g.s_main, C, d->d_list, rf.off, rf.na, rf.offb, rf.nb, pan, yield);
      ++timed;
      halves_waves.push_back(k);
    } else if (r2.na + r2.nb > 0) {
      launch_trail_update<T>(g.s_main, C, d->d_list, r2.off, r2.na, r2.offb, r2.nb, pan, yield);
      ++timed;
    }
    HIPCHECK(hipEventRecord(ev(k, E_U2), g.s_main));
    if (g.profiling) {
      // the bracket [P0, P1] on s_main covers every k_trail_update launch of the wave: U1(k) started
      // with U2(k); waiting for its end here constrains nothing (U2(k+1) needs panel k+1, which needs it).
      // (halves: far(k+1) does NOT need near(k) -- two brackets, [PN0, PN1] on s_u1 for column k+1 and the
      // near half, and the host takes the union)
      if (concurrent && !halves) HIPCHECK(hipStreamWaitEvent(g.s_main, ev(k, E_U1R), 0));
      HIPCHECK(hipEventRecord(ev(k, E_P1), g.s_main));
    }
    prev_halves = halves;
    upd_launches += timed;
    // algorithmic flops of the launches inside the bracket: GEMM 2 B^3 per off-diagonal tile, SYRK B^3
    // per diagonal tile; the diagonal-tile SYRK of the split form is not a k_trail_update
    upd_flops += (2.0 * r2.na + r2.nb) * b3;
    upd_flops += (2.0 * r1.na + ((split && cholmi::g_intile_small) ? 0 : r1.nb)) * b3;
  }
  if (open_bracket >= 0 && g.profiling) {
    HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_COLS], 0));
    HIPCHECK(hipEventRecord(ev(open_bracket, E_P1), g.s_main));
  }
  HIPCHECK(hipEventRecord(fixed[F_JOIN], g.s_panel));
  HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_JOIN], 0));
  HIPCHECK(hipEventRecord(fixed[F_TRSM], g.s_trsm));
  HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_TRSM], 0));
  HIPCHECK(hipEventRecord(fixed[F_U1END], g.s_u1));
  HIPCHECK(hipStreamWaitEvent(g.s_main, fixed[F_U1END], 0));
  HIPCHECK(hipEventRecord(fixed[F_STOP], g.s_main));
  HIPCHECK(hipStreamSynchronize(g.s_main));
  float ms = 0;
  HIPCHECK(hipEventElapsedTime(&ms, fixed[F_START], fixed[F_STOP]));
  g.total_ms = ms;
  g.update_flops = upd_flops;
  g.update_launches = upd_launches;
  g.update_ms = 0;
  if (g.profiling) {
    // union of the brackets (disjoint by construction except around the waves launched as halves)
    std::vector<std::pair<float, float>> iv;
    auto add = [&](hipEvent_t a, hipEvent_t b) -> int {
      float t0 = 0, t1 = 0;
      HIPCHECK(hipEventElapsedTime(&t0, fixed[F_START], a));
      HIPCHECK(hipEventElapsedTime(&t1, fixed[F_START], b));
      if (t1 > t0) iv.emplace_back(t0, t1);
      return 0;
    };
    for (int k = 0; k + 1 < nt; ++k)
      if ((rc = add(ev(k, E_P0), ev(k, E_P1)))) return rc;
    for (int k : halves_waves)
      if ((rc = add(ev(k, E_PN0), ev(k, E_PN1)))) return rc;
    std::sort(iv.begin(), iv.end());
    float hi = -1;
    for (auto &p : iv) {
      if (p.first > hi) g.update_ms += p.second - p.first;
      else if (p.second > hi) g.update_ms += p.second - hi;
      hi = std::max(hi, p.second);
    }
  }
  int info = 0;
  rc = read_info(&info);
  if (rc) return rc;
  if (info > 0 && d->mbi != d->mb)  // stored index -> index in the caller's matrix
    info = ((info - 1) / d->mbi) * d->mb + (info - 1) % d->mbi + 1;
  return info;
}

int build_worklist(chol_desc *d) {
  if (d->mt != d->nt) return 0;  // only square tile grids are factored
  std::vector<int2> off, dg;
  d->ge.assign(d->nt + 2, 0);
  d->gd.assign(d->nt + 2, 0);
  // diagnostic only (single-process descriptors): keep the diagonal tiles where the column order
  // puts them, to measure what moving them to the end of a launch buys
  const char *ord = getenv("CHOLMI_LIST_ORDER");
  const bool interleaved = ord && !strcmp(ord, "interleaved") && d->p * d->q == 1;
  for (int J = d->nt - 1; J >= 0; --J) {
    if (J % d->q == d->pcol)
      for (int I = J; I < d->mt; ++I)
        if (I % d->p == d->prow) ((I == J && !interleaved) ? dg : off).push_back(make_int2(I, J));
    d->ge[J] = (int)off.size();
    d->gd[J] = (int)dg.size();
  }
  d->n_off = (int)off.size();
  off.insert(off.end(), dg.begin(), dg.end());
  if (!off.empty()) {
    HIPCHECK(hipMalloc(&d->d_list, off.size() * sizeof(int2)));
    HIPCHECK(hipMemcpy(d->d_list, off.data(), off.size() * sizeof(int2), hipMemcpyHostToDevice));
  }
  return 0;
}

template <typename T>
static int potrf_impl(chol_desc *A, bool upper_staged = false) {
  CHECK_WINV(A, "potrf_tile");
  if (single_tile_square(A)) {
    Staged st;
    int rc = stage_in<T>(A, 0, /*identity_pad=*/true, &st);
    if (rc) return rc;
    HIPCHECK(hipMemsetAsync(g.d_info, 0, sizeof(int), g.s_main));
    // ChamUpper on a staged tile: A = U^T U with U = L^T -- transpose the staged (identity-padded,
    // multiple-of-128) copy, factor Lower, transpose back: the caller's strict lower triangle comes
    // back exactly as it went in
    if (upper_staged) launch_transpose_inplace<T>(g.s_main, reinterpret_cast<T *>(st.dev), 1, st.ldp);
    launch_potrf_tile<T>(g.s_main, reinterpret_cast<T *>(st.dev), st.ldp, reinterpret_cast<T *>(g.winv),
                         g.d_info, 0, tile_sems());
    if (upper_staged) launch_transpose_inplace<T>(g.s_main, reinterpret_cast<T *>(st.dev), 1, st.ldp);
    rc = stage_out<T>(A, st);
    if (rc) return rc;
    HIPCHECK(hipStreamSynchronize(g.s_main));
    int info = 0;
    rc = read_info(&info);
    return rc ? rc : info;
  }
  if (A->p * A->q != 1) return chol_internal_dist_potrf(A, g.rank);  // dist.hip: the block-cyclic wave loop
  if (A->mt != A->nt || A->lm != A->ln) return fail(-2, "potrf_tile: matrix is not square");
  if (A->on_device) return potrf_full_device<T>(A, A->mat);
  // host-resident tiled matrix: stage the whole matrix through HBM
  if (A->padded) return fail(CHOL_ERR_NOT_SUPPORTED, "potrf_tile: padded image over a host buffer");
  const size_t bytes = (size_t)A->mt * A->nt * A->bsiz * sizeof(T);
  void *dev = nullptr;
  if (hipMalloc(&dev, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(CHOL_ERR_OUT_OF_MEMORY, "potrf_tile: staging allocation failed");
  }
  int rc = 0;
  hipError_t e = hipMemcpy(dev, A->mat, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = potrf_full_device<T>(A, dev);
    if (rc >= 0) e = hipMemcpy(A->mat, dev, bytes, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dev);
  if (e != hipSuccess) return fail_hip(e, "staged potrf copy", __LINE__);
  return rc;
}

template <typename T>
static int trsm_impl(double alpha, chol_desc *L, chol_desc *B) {
  CHECK_WINV(L, "trsm_tile");
  Staged sl, sb;
  int rc = stage_in<T>(L, 0, /*identity_pad=*/true, &sl);
  if (rc) return rc;
  rc = stage_in<T>(B, 1, false, &sb);
  if (rc) return rc;
  T *winv = reinterpret_cast<T *>(g.winv);
  launch_invert_diag<T>(g.s_main, reinterpret_cast<const T *>(sl.dev), sl.ldp, winv);
  launch_trsm_panel<T>(g.s_main, reinterpret_cast<T *>(sb.dev), (long)sb.ldp * sb.ldp, 1,
                       reinterpret_cast<const T *>(sl.dev), winv, sb.ldp, (T)alpha);
  rc = stage_out<T>(B, sb);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(g.s_main));
  return 0;
}

template <typename T>
static int gemm_impl(double alpha, chol_desc *A, chol_desc *B, double beta, chol_desc *C, bool lower) {
  Staged sa, sb, sc;
  int rc = stage_in<T>(A, 0, false, &sa);
  if (rc) return rc;
  if (B == A) {
    sb = sa;
  } else {
    rc = stage_in<T>(B, 1, false, &sb);
    if (rc) return rc;
  }
  rc = stage_in<T>(C, 2, false, &sc);
  if (rc) return rc;
  launch_gemm_nt_tile<T>(g.s_main, reinterpret_cast<const T *>(sa.dev), reinterpret_cast<const T *>(sb.dev),
                         reinterpret_cast<T *>(sc.dev), sc.ldp, (T)alpha, (T)beta, lower);
  rc = stage_out<T>(C, sc);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(g.s_main));
  return 0;
}

// ---------------------------------------------------------------- solve with the factor
// X <- A^{-1} B for A = L L^T already factored (CHAMELEON_dpotrs_Tile(ChamLower, A, B)).
// The kernels of this library are the right-sided NT forms the factorisation needs
// (X = A L^{-T}, C -= A B^T), so the solve runs on Z = B^T:
//   forward   Z(:,k) <- Z(:,k) L(k,k)^{-T};  Z(:,i) -= Z(:,k) L(i,k)^T, i > k      (L Y = B)
//   backward  Z(:,k) <- Z(:,k) L(k,k)^{-1};  Z(:,i) -= Z(:,k) L(k,i),   i < k      (L^T X = Y)
// the backward sweep's operands being transposed tiles (L(k,k)^{-T} from a TRSM of the identity,
// L(k,i)^T from a tile transpose) so that every product is again A B^T.
template <typename T>
int potrs_impl(chol_desc *A, chol_desc *B) {
  const int nt = A->nt, nr = B->nt, mb = A->mbi;
  const long bs = A->bsizi;
  const size_t tb = (size_t)bs * sizeof(T);
  T *La = reinterpret_cast<T *>(A->mat), *Bm = reinterpret_cast<T *>(B->mat);
  T *scr = nullptr;
  if (hipMalloc(&scr, ((size_t)nr * nt + 2 + (size_t)nr + (size_t)nt) * tb) != hipSuccess) {
    (void)hipGetLastError();
    return fail(CHOL_ERR_OUT_OF_MEMORY, "potrs_tile: scratch allocation failed");
  }
  T *Z = scr, *Wt = scr + (size_t)nr * nt * bs, *Tt = Wt + bs, *tmp = Tt + (size_t)nt * bs;  // Tt: nt tiles, tmp: nr tiles
  hipStream_t s = g.s_main;
  T *winv = reinterpret_cast<T *>(g.winv);
  auto Ltile = [&](int i, int j) { return La + ((long)i + (long)j * A->lmt) * bs; };
  auto Ztile = [&](int r, int i) { return Z + ((long)r + (long)i * nr) * bs; };
  // Z(r,i) = B(i,r)^T
  for (int r = 0; r < nr; ++r)
    launch_tiles_transpose<T>(s, Bm + (long)r * B->lmt * bs, bs, Ztile(r, 0), (long)nr * bs, mb, nt);
  for (int k = 0; k < nt; ++k) {  // forward
    launch_invert_diag<T>(s, Ltile(k, k), mb, winv);
    launch_trsm_panel<T>(s, Ztile(0, k), bs, nr, Ltile(k, k), winv, mb, T(1));
    // Z(r,i) -= Z(r,k) L(i,k)^T for every r and i > k: one launch
    launch_gemm_nt_batch<T>(s, Ztile(0, k), bs, nr, Ltile(k + 1, k), bs, nt - 1 - k, Ztile(0, k + 1), bs, (long)nr * bs, mb,
                            T(-1), T(1));
  }
  for (int k = nt - 1; k >= 0; --k) {  // backward
    HIPCHECK(hipMemsetAsync(Wt, 0, tb, s));
    launch_pad_identity<T>(s, Wt, 0, mb);
    launch_invert_diag<T>(s, Ltile(k, k), mb, winv);
    launch_trsm_panel<T>(s, Wt, bs, 1, Ltile(k, k), winv, mb, T(1));  // Wt = L(k,k)^{-T}
    // Z(r,k) <- Z(r,k) L(k,k)^{-1} for every r (out of place, then back: the tiles of a column are contiguous)
    launch_gemm_nt_batch<T>(s, Ztile(0, k), bs, nr, Wt, 0, 1, tmp, bs, 0, mb, T(1), T(0));
    HIPCHECK(hipMemcpyAsync(Ztile(0, k), tmp, (size_t)nr * tb, hipMemcpyDeviceToDevice, s));
    // Z(r,i) -= Z(r,k) L(k,i) for every r and i < k: the k tiles L(k,i)^T in one transpose launch, one product launch
    if (k > 0) {
      launch_tiles_transpose<T>(s, Ltile(k, 0), (long)A->lmt * bs, Tt, bs, mb, k);
      launch_gemm_nt_batch<T>(s, Ztile(0, k), bs, nr, Tt, bs, k, Ztile(0, 0), bs, (long)nr * bs, mb, T(-1), T(1));
    }
  }
  for (int r = 0; r < nr; ++r)  // B(i,r) = Z(r,i)^T
    launch_tiles_transpose<T>(s, Ztile(r, 0), (long)nr * bs, Bm + (long)r * B->lmt * bs, bs, mb, nt);
  hipError_t e = hipStreamSynchronize(s);
  (void)hipFree(scr);
  if (e != hipSuccess) return fail_hip(e, "potrs_tile", __LINE__);
  return 0;
}

}  // namespace

extern "C" {

int chol_internal_fail(int code, const char *msg) { return fail(code, msg); }

const char *chol_version(void) { return "cholmi 0.2 (gfx950)"; }
const char *chol_last_error(void) { return g.last_error.c_str(); }

int chol_set_device(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g.inited && device != g.device) return fail(-1, "chol_set_device after chol_init");
  g.device = device;
  return 0;
}

int chol_set_rank(int rank, int nranks) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(-1, "chol_set_rank: bad rank");
  g.rank = rank;
  g.nranks = nranks;
  return 0;
}

int chol_init(int ncpu, int ngpu) {
  (void)ncpu;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g.inited) return 0;
  if (ngpu < 1)
    return fail(CHOL_ERR_NO_GPU, "chol_init: ngpu must be >= 1 (this library has no CPU backend)");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1) {
    (void)hipGetLastError();
    return fail(CHOL_ERR_NO_GPU, "chol_init: no HIP device visible");
  }
  if (g.device < 0) {
    const char *lr = getenv("LOCAL_RANK");
    g.device = lr ? atoi(lr) % count : 0;
  }
  if (g.device >= count) return fail(CHOL_ERR_NO_GPU, "chol_init: device index out of range");
  HIPCHECK(hipSetDevice(g.device));
  int lo = 0, hi = 0;
  HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIPCHECK(hipStreamCreateWithPriority(&g.s_main, hipStreamNonBlocking, lo));
  HIPCHECK(hipStreamCreateWithPriority(&g.s_panel, hipStreamNonBlocking, hi));
  HIPCHECK(hipStreamCreateWithPriority(&g.s_trsm, hipStreamNonBlocking, hi));
  // the update of column k+1 (what the next panel waits for) runs beside the rest of the wave's
  // update, ahead of it in priority
  HIPCHECK(hipStreamCreateWithPriority(&g.s_u1, hipStreamNonBlocking, lo - 1 > hi ? lo - 1 : hi));
  if (const char *e = getenv("CHOLMI_VARIANT")) cholmi::g_variant = atoi(e);
  if (const char *e = getenv("CHOLMI_INTILE")) cholmi::g_intile_small = strcmp(e, "big") != 0;
  if (const char *e = getenv("CHOLMI_TRSM_SMALL_MAX")) cholmi::g_trsm_small_max = atoi(e);
  {
    // grids that may poll a counter themselves: few enough that, one per CU in the worst case, most CUs
    // stay free of pollers (kernels.hip: k_sem_gate); on a small partition every grid waits behind a gate
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, g.device));
    const int room = (prop.multiProcessorCount - 32) / 4;
    cholmi::g_poll_max_wgs = room < 0 ? 0 : (room < 48 ? room : 48);
  }
  if (const char *e = getenv("CHOLMI_POLL_MAX_WGS")) cholmi::g_poll_max_wgs = atoi(e);
  if (const char *e = getenv("CHOLMI_INTILE_FUSED")) cholmi::g_intile_fused = atoi(e);
  if (const char *e = getenv("CHOLMI_MIN_UNITS")) cholmi::g_min_units = atoi(e);
  if (const char *e = getenv("CHOLMI_TRSM_FUSED_MIN")) cholmi::g_trsm_fused_min = atoi(e);
  if (const char *e = getenv("CHOLMI_LATE_DMA")) cholmi::g_late_dma = atoi(e);
  if (const char *e = getenv("CHOLMI_F32_W8")) cholmi::g_f32_w8 = atoi(e);
  g.winv_bytes = (size_t)32 * MACRO * MACRO * sizeof(double);  // tiles up to 4096
  HIPCHECK(hipMalloc(&g.winv, 2 * g.winv_bytes));  // two sets: the walker alternates them by wave parity
  HIPCHECK(hipMalloc(&g.d_info, sizeof(int)));
  HIPCHECK(hipMemset(g.d_info, 0, sizeof(int)));
  HIPCHECK(hipMalloc(&g.d_acc, 2 * sizeof(double)));
  {
    const char *e = getenv("CHOLMI_YIELD");
    if (!e || atoi(e) != 0) {
      HIPCHECK(hipMalloc(&g.d_ytab, YTAB_ENTRIES * sizeof(int)));
      HIPCHECK(hipMemset(g.d_ytab, 0, YTAB_ENTRIES * sizeof(int)));
      cholmi::g_ytab = g.d_ytab;
    }
  }
  {
    // Device-side edges of the panel chain: usable only if a kernel polling on one of the two panel
    // streams does not keep the other from running (streams that share a hardware queue would
    // deadlock until the poll's bound).  Probed once, both ways, consumer launched first.
    const char *e = getenv("CHOLMI_DEVICE_FLAGS");
    if (!e || atoi(e) != 0) {
      HIPCHECK(hipMalloc(&g.d_sem, SEM_INTS * sizeof(int)));
      HIPCHECK(hipMemset(g.d_sem, 0, SEM_INTS * sizeof(int)));
      bool ok = true;
      const hipStream_t pairs[4][2] = {{g.s_trsm, g.s_panel}, {g.s_u1, g.s_trsm}, {g.s_panel, g.s_u1}, {g.s_panel, g.s_trsm}};
      for (int t = 0; t < 4 && ok; ++t) {  // {consumer, producer}: the three edges of SyrkPipe, and sp / st the other way
        int *sem = g.d_sem + 64 * t, *res = g.d_sem + 64 * t + 32;
        cholmi::launch_sem_probe(pairs[t][0], pairs[t][1], sem, res);
        HIPCHECK(hipStreamSynchronize(pairs[t][0]));
        HIPCHECK(hipStreamSynchronize(pairs[t][1]));
        int r = 0;
        HIPCHECK(hipMemcpy(&r, res, sizeof(int), hipMemcpyDeviceToHost));
        ok = (r == 1);
      }
      if (!ok) {
        (void)hipFree(g.d_sem);
        g.d_sem = nullptr;
      }
    }
  }
  g.inited = true;
  return 0;
}

int chol_finalize(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g.inited) return 0;
  (void)hipDeviceSynchronize();
  chol_internal_dist_finalize();
  for (auto e : g.events) (void)hipEventDestroy(e);
  g.events.clear();
  for (int i = 0; i < 3; ++i) {
    if (g.stage[i]) (void)hipFree(g.stage[i]);
    g.stage[i] = nullptr;
    g.stage_bytes[i] = 0;
  }
  (void)hipFree(g.winv);
  (void)hipFree(g.d_info);
  (void)hipFree(g.d_acc);
  if (g.d_ytab) (void)hipFree(g.d_ytab);
  g.d_ytab = nullptr;
  if (g.d_sem) (void)hipFree(g.d_sem);
  g.d_sem = nullptr;
  cholmi::g_ytab = nullptr;
  (void)hipStreamDestroy(g.s_main);
  (void)hipStreamDestroy(g.s_panel);
  (void)hipStreamDestroy(g.s_trsm);
  (void)hipStreamDestroy(g.s_u1);
  g.winv = nullptr;
  g.d_info = nullptr;
  g.d_acc = nullptr;
  g.s_main = g.s_panel = g.s_trsm = g.s_u1 = nullptr;
  g.inited = false;
  return 0;
}

int chol_desc_create(chol_desc_t **desc, void *mat, int dtype, int mb, int nb, int bsiz, int lm,
                     int ln, int i, int j, int m, int n, int p, int q) {
  if (!desc) return fail(-1, "desc_create: desc is NULL");
  *desc = nullptr;
  if (dtype != CHOL_REAL_DOUBLE && dtype != CHOL_REAL_FLOAT) return fail(-3, "desc_create: dtype");
  if (mb <= 0) return fail(-4, "desc_create: mb");
  if (nb <= 0) return fail(-5, "desc_create: nb");
  if (bsiz != mb * nb) return fail(-6, "desc_create: bsiz != mb*nb");
  if (lm <= 0) return fail(-7, "desc_create: lm");
  if (ln <= 0) return fail(-8, "desc_create: ln");
  if (i < 0 || i >= lm) return fail(-9, "desc_create: i");
  if (j < 0 || j >= ln) return fail(-10, "desc_create: j");
  if (m <= 0 || i + m > lm) return fail(-11, "desc_create: m");
  if (n <= 0 || j + n > ln) return fail(-12, "desc_create: n");
  if (p <= 0 || p > MAXP) return fail(-13, "desc_create: p");
  if (q <= 0) return fail(-14, "desc_create: q");
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "desc_create before chol_init");
  if (i != 0 || j != 0 || m != lm || n != ln) {
    // A sub-matrix view A(i:i+m, j:j+n) of an lm x ln matrix (V6:24-25, 44-45 pass ioff, joff, m, n
    // straight through).  With library-owned storage nothing outside the view can ever be observed
    // through this descriptor -- Chameleon's generator, factorisation and layout conversions all work in
    // view coordinates (dplgsy: entry (r, c) of the view, order m) -- so the view IS an m x n matrix of
    // its own: only the tiles it can address are allocated.  Tile-aligned offsets only (an unaligned
    // view starts with a partial tile, which would change what tile (0,0) means to tile_upload / _download).
    if (mat) return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: sub-matrix views (i,j,m,n) over a user buffer are not supported");
    if (i % mb || j % nb) return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: sub-matrix views need tile-aligned offsets (i % mb == 0, j % nb == 0)");
    if (p * q != 1) return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: sub-matrix views of distributed matrices are not supported");
    lm = m;
    ln = n;
    i = j = 0;
  }
  if (p * q != 1 && p * q != g.nranks)
    return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: p*q must equal the number of ranks (chol_set_rank)");
  chol_desc *d = new chol_desc();
  d->dtype = dtype; d->mb = mb; d->nb = nb; d->bsiz = bsiz; d->lm = lm; d->ln = ln;
  d->i = i; d->j = j; d->m = m; d->n = n; d->p = p; d->q = q;
  d->esize = dtype == CHOL_REAL_DOUBLE ? 8 : 4;
  d->mt = (lm + mb - 1) / mb;
  d->nt = (ln + nb - 1) / nb;
  const int rank = (p * q == 1) ? 0 : g.rank;
  d->prow = rank / q;
  d->pcol = rank % q;
  d->lmt = (d->mt - d->prow + p - 1) / p;
  d->lnt = (d->nt - d->pcol + q - 1) / q;
  if (d->lmt < 0) d->lmt = 0;
  if (d->lnt < 0) d->lnt = 0;
  const bool multi = d->mt > 1 || d->nt > 1;
  d->mbi = mb;
  d->bsizi = bsiz;
  d->padded = false;
  // a matrix smaller than its one tile (lm < mb): same treatment as a ragged edge tile
  const bool partial1 = !multi && (lm != mb || ln != nb);
  if ((multi && (lm % mb || ln % nb || mb % MACRO)) || partial1) {
    // ragged order and/or a tile edge that is not a multiple of 128 (the reference's sweep
    // uses NB = 192 ... 448): the library keeps its own image with tiles rounded up to 128
    // and the identity outside the matrix.  Needs library-owned storage on one process.
    if (mat || p * q != 1 || mb != nb)
      return delete d, fail(CHOL_ERR_NOT_SUPPORTED,
                            "desc_create: ragged / non-128 tiles need mat == NULL, p*q == 1, square tiles");
    d->mbi = roundup(mb, MACRO);
    d->bsizi = d->mbi * d->mbi;
    d->padded = true;
  }
  if (multi && mb != nb)
    return delete d, fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: tiled matrices need mb == nb");
  if (mat) {
    d->mat = mat;
    d->owns = false;
    d->on_device = is_device_ptr(mat);
  } else {
    const size_t bytes = (size_t)std::max(1, d->lmt) * std::max(1, d->lnt) * (size_t)d->bsizi * d->esize;
    hipError_t e = hipMalloc(&d->mat, bytes);
    if (e != hipSuccess) {
      delete d;
      (void)hipGetLastError();
      return fail(CHOL_ERR_OUT_OF_MEMORY, "desc_create: hipMalloc failed");
    }
    d->owns = true;
    d->on_device = true;
    if (d->padded && lm == ln) {  // zero everywhere, identity on the diagonal tiles' diagonals
      const LocalMat Lm = local_mat(d, d->mat);
      if (dtype == CHOL_REAL_DOUBLE)
        launch_plgsy<double>(g.s_main, Lm, d->lnt, 0, 0, 0.0, 0ull, 0, 0, 0);
      else
        launch_plgsy<float>(g.s_main, Lm, d->lnt, 0, 0, 0.0, 0ull, 0, 0, 0);
      (void)hipStreamSynchronize(g.s_main);
    } else {  // library-owned storage starts at zero (tiles a one-sided dplgsy does not touch)
      (void)hipMemsetAsync(d->mat, 0, bytes, g.s_main);
      (void)hipStreamSynchronize(g.s_main);
    }
  }
  if (multi || d->padded) {
    int rc = build_worklist(d);
    if (rc) {
      if (d->owns) (void)hipFree(d->mat);
      delete d;
      return rc;
    }
  }
  *desc = d;
  return 0;
}

int chol_desc_destroy(chol_desc_t **desc) {
  if (!desc || !*desc) return fail(-1, "desc_destroy: NULL");
  chol_desc *d = *desc;
  if (d->d_list) (void)hipFree(d->d_list);
  if (d->owns && d->mat) (void)hipFree(d->mat);
  delete d;
  *desc = nullptr;
  return 0;
}

void *chol_desc_local_ptr(chol_desc_t *d, size_t *bytes) {
  if (!d) return nullptr;
  if (bytes) *bytes = (size_t)d->lmt * d->lnt * (size_t)d->bsizi * d->esize;
  return d->mat;
}

int chol_desc_local_tiles(chol_desc_t *d, int *lmt, int *lnt) {
  if (!d) return fail(-1, "NULL desc");
  if (lmt) *lmt = d->lmt;
  if (lnt) *lnt = d->lnt;
  return 0;
}

// ---------------------------------------------------------------- POTRF
int chol_potrf_tile(int uplo, chol_desc_t *A) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "potrf_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-1, "potrf_tile: uplo");
  if (!A) return fail(-2, "potrf_tile: NULL descriptor");
  std::lock_guard<std::mutex> lk(g_mu);
  if (uplo == CHOL_LOWER)
    return A->dtype == CHOL_REAL_DOUBLE ? potrf_impl<double>(A) : potrf_impl<float>(A);
  // ChamUpper: A = U^T U with U = L^T.  Transpose the stored matrix in place (its upper
  // triangle becomes the lower one), factor Lower, transpose back: the upper triangle now
  // holds U and the caller's strict lower triangle is bit-for-bit what it was.
  const bool one = single_tile_square(A);
  if (!one && (A->p * A->q != 1 || !A->on_device || A->mt != A->nt))
    return fail(CHOL_ERR_NOT_SUPPORTED, "potrf_tile(Upper): single-process device-resident square matrices");
  if (one && !(A->on_device && A->mb % MACRO == 0))
    return A->dtype == CHOL_REAL_DOUBLE ? potrf_impl<double>(A, true) : potrf_impl<float>(A, true);
  auto flip = [&]() {
    if (A->dtype == CHOL_REAL_DOUBLE)
      launch_transpose_inplace<double>(g.s_main, (double *)A->mat, A->nt, A->mbi);
    else
      launch_transpose_inplace<float>(g.s_main, (float *)A->mat, A->nt, A->mbi);
  };
  flip();
  const int rc = A->dtype == CHOL_REAL_DOUBLE ? potrf_impl<double>(A) : potrf_impl<float>(A);
  flip();
  HIPCHECK(hipStreamSynchronize(g.s_main));
  return rc;
}

int chol_trsm_tile(int side, int uplo, int trans, int diag, double alpha, chol_desc_t *A,
                   chol_desc_t *B) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "trsm_tile before chol_init");
  if (side != CHOL_LEFT && side != CHOL_RIGHT) return fail(-1, "trsm_tile: side");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-2, "trsm_tile: uplo");
  if (trans != CHOL_NOTRANS && trans != CHOL_TRANS) return fail(-3, "trsm_tile: trans");
  if (diag != CHOL_NONUNIT && diag != CHOL_UNIT) return fail(-4, "trsm_tile: diag");
  if (!A) return fail(-6, "trsm_tile: A is NULL");
  if (!B) return fail(-7, "trsm_tile: B is NULL");
  if (side != CHOL_RIGHT || uplo != CHOL_LOWER || trans != CHOL_TRANS || diag != CHOL_NONUNIT)
    return fail(CHOL_ERR_NOT_SUPPORTED, "trsm_tile: only (Right, Lower, Trans, NonUnit)");
  if (!single_tile_square(A) || !single_tile_square(B) || A->mb != B->mb || A->dtype != B->dtype)
    return fail(CHOL_ERR_NOT_SUPPORTED, "trsm_tile: needs two 1-tile descriptors of equal size and type");
  std::lock_guard<std::mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? trsm_impl<double>(alpha, A, B) : trsm_impl<float>(alpha, A, B);
}

// ---------------------------------------------------------------- SYRK / GEMM
int chol_syrk_tile(int uplo, int trans, double alpha, chol_desc_t *A, double beta, chol_desc_t *C) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "syrk_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-1, "syrk_tile: uplo");
  if (trans != CHOL_NOTRANS && trans != CHOL_TRANS) return fail(-2, "syrk_tile: trans");
  if (!A) return fail(-4, "syrk_tile: A is NULL");
  if (!C) return fail(-6, "syrk_tile: C is NULL");
  if (uplo != CHOL_LOWER || trans != CHOL_NOTRANS)
    return fail(CHOL_ERR_NOT_SUPPORTED, "syrk_tile: only (Lower, NoTrans)");
  if (!single_tile_square(A) || !single_tile_square(C) || A->mb != C->mb || A->dtype != C->dtype)
    return fail(CHOL_ERR_NOT_SUPPORTED, "syrk_tile: needs two 1-tile descriptors of equal size and type");
  std::lock_guard<std::mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? gemm_impl<double>(alpha, A, A, beta, C, true)
                                      : gemm_impl<float>(alpha, A, A, beta, C, true);
}

int chol_gemm_tile(int transA, int transB, double alpha, chol_desc_t *A, chol_desc_t *B,
                   double beta, chol_desc_t *C) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "gemm_tile before chol_init");
  if (transA != CHOL_NOTRANS && transA != CHOL_TRANS) return fail(-1, "gemm_tile: transA");
  if (transB != CHOL_NOTRANS && transB != CHOL_TRANS) return fail(-2, "gemm_tile: transB");
  if (!A) return fail(-4, "gemm_tile: A is NULL");
  if (!B) return fail(-5, "gemm_tile: B is NULL");
  if (!C) return fail(-7, "gemm_tile: C is NULL");
  if (transA != CHOL_NOTRANS || transB != CHOL_TRANS)
    return fail(CHOL_ERR_NOT_SUPPORTED, "gemm_tile: only (NoTrans, Trans)");
  if (!single_tile_square(A) || !single_tile_square(B) || !single_tile_square(C) || A->mb != C->mb ||
      B->mb != C->mb || A->dtype != C->dtype || B->dtype != C->dtype)
    return fail(CHOL_ERR_NOT_SUPPORTED, "gemm_tile: needs three 1-tile descriptors of equal size and type");
  std::lock_guard<std::mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? gemm_impl<double>(alpha, A, B, beta, C, false)
                                      : gemm_impl<float>(alpha, A, B, beta, C, false);
}

// ---------------------------------------------------------------- generator / layout / residual
int chol_plgsy_tile(double bump, int uplo, chol_desc_t *A, unsigned long long seed) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "plgsy_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER && uplo != CHOL_UPPER_LOWER) return fail(-2, "plgsy_tile: uplo");
  const int side = uplo == CHOL_LOWER ? 1 : uplo == CHOL_UPPER ? 2 : 0;
  if (!A) return fail(-3, "plgsy_tile: NULL descriptor");
  if (!A->on_device) return fail(CHOL_ERR_NOT_SUPPORTED, "plgsy_tile: descriptor must be device-resident");
  if (A->mt == 1 && A->nt == 1 && (A->m != A->mb || A->n != A->nb) && !A->padded)
    return fail(CHOL_ERR_NOT_SUPPORTED, "plgsy_tile: partial single tile");
  std::lock_guard<std::mutex> lk(g_mu);
  const LocalMat L = local_mat(A, A->mat);
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_plgsy<double>(g.s_main, L, A->lnt, A->prow, A->pcol, bump, seed, A->mb, (long)A->lm, side);
  else
    launch_plgsy<float>(g.s_main, L, A->lnt, A->prow, A->pcol, bump, seed, A->mb, (long)A->lm, side);
  HIPCHECK(hipStreamSynchronize(g.s_main));
  return 0;
}

// ---------------------------------------------------------------- V6 validation block
static int resident_whole(const char *what, const chol_desc *d) {
  char buf[160];
  const char *why = nullptr;
  if (!d) why = "NULL descriptor";
  else if (!d->on_device) why = "descriptor must be device-resident";
  else if (d->p * d->q != 1) why = "distributed descriptor";
  else if (d->mb != d->nb) why = "tiles must be square";
  if (!why) return 0;
  snprintf(buf, sizeof buf, "%s: %s", what, why);
  return fail(d ? CHOL_ERR_NOT_SUPPORTED : -2, buf);
}
static bool same_geometry(const chol_desc *a, const chol_desc *b) {
  return a->dtype == b->dtype && a->mb == b->mb && a->nb == b->nb && a->lm == b->lm && a->ln == b->ln &&
         a->mbi == b->mbi && a->lmt == b->lmt && a->lnt == b->lnt;
}
static TileGeo geo_of(const chol_desc *d) {
  TileGeo g;
  g.lmt = d->lmt;
  g.lnt = d->lnt;
  g.mbs = d->mbi;
  g.mbu = d->mb;
  g.m = d->lm;
  g.n = d->ln;
  return g;
}

int chol_lacpy_tile(int uplo, chol_desc_t *A, chol_desc_t *B) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "lacpy_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER && uplo != CHOL_UPPER_LOWER) return fail(-1, "lacpy_tile: uplo");
  int rc = resident_whole("lacpy_tile", A);
  if (rc) return rc;
  rc = resident_whole("lacpy_tile", B);
  if (rc) return rc;
  if (!same_geometry(A, B)) return fail(-3, "lacpy_tile: descriptors differ in shape, tiling or type");
  std::lock_guard<std::mutex> lk(g_mu);
  const int side = uplo == CHOL_LOWER ? 1 : uplo == CHOL_UPPER ? 2 : 0;
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_lacpy<double>(g.s_main, geo_of(A), side, (const double *)A->mat, (double *)B->mat);
  else
    launch_lacpy<float>(g.s_main, geo_of(A), side, (const float *)A->mat, (float *)B->mat);
  HIPCHECK(hipStreamSynchronize(g.s_main));
  return 0;
}

int chol_geadd_tile(int trans, double alpha, chol_desc_t *A, double beta, chol_desc_t *B) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "geadd_tile before chol_init");
  if (trans != CHOL_NOTRANS) return fail(CHOL_ERR_NOT_SUPPORTED, "geadd_tile: only ChamNoTrans (V6:83)");
  int rc = resident_whole("geadd_tile", A);
  if (rc) return rc;
  rc = resident_whole("geadd_tile", B);
  if (rc) return rc;
  if (!same_geometry(A, B)) return fail(-5, "geadd_tile: descriptors differ in shape, tiling or type");
  std::lock_guard<std::mutex> lk(g_mu);
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_geadd<double>(g.s_main, geo_of(A), alpha, (const double *)A->mat, beta, (double *)B->mat);
  else
    launch_geadd<float>(g.s_main, geo_of(A), alpha, (const float *)A->mat, beta, (float *)B->mat);
  HIPCHECK(hipStreamSynchronize(g.s_main));
  return 0;
}

int chol_lange_tile(int norm, chol_desc_t *A, double *value) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "lange_tile before chol_init");
  if (!value) return fail(-3, "lange_tile: NULL value");
  int kind;
  switch (norm) {
    case CHOL_MAX_NORM: kind = 0; break;
    case CHOL_ONE_NORM: kind = 1; break;
    case CHOL_INF_NORM: kind = 2; break;
    case CHOL_FROBENIUS_NORM: kind = 3; break;
    default: return fail(-1, "lange_tile: norm");
  }
  int rc = resident_whole("lange_tile", A);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  const TileGeo ge = geo_of(A);
  double *work = nullptr;
  HIPCHECK(hipMalloc(&work, (size_t)(std::max(ge.m, ge.n) + 2) * sizeof(double)));
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_lange<double>(g.s_main, ge, kind, (const double *)A->mat, work);
  else
    launch_lange<float>(g.s_main, ge, kind, (const float *)A->mat, work);
  double v = 0;
  hipError_t e = hipMemcpyAsync(&v, work, sizeof(double), hipMemcpyDeviceToHost, g.s_main);
  if (e == hipSuccess) e = hipStreamSynchronize(g.s_main);
  (void)hipFree(work);
  if (e != hipSuccess) return fail_hip(e, "lange_tile", __LINE__);
  *value = kind == 3 ? std::sqrt(v) : v;
  return 0;
}

int chol_lauum_tile(int uplo, chol_desc_t *A) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "lauum_tile before chol_init");
  if (uplo != CHOL_LOWER) return fail(CHOL_ERR_NOT_SUPPORTED, "lauum_tile: only ChamLower (V6:80)");
  int rc = resident_whole("lauum_tile", A);
  if (rc) return rc;
  if (A->mt != A->nt || A->lm != A->ln) return fail(-2, "lauum_tile: matrix is not square");
  if (A->mbi % 64) return fail(CHOL_ERR_NOT_SUPPORTED, "lauum_tile: stored tile edge must be a multiple of 64");
  std::lock_guard<std::mutex> lk(g_mu);
  const size_t bytes = (size_t)A->mt * A->nt * A->bsizi * A->esize;
  void *tmp = nullptr;
  if (hipMalloc(&tmp, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(CHOL_ERR_OUT_OF_MEMORY, "lauum_tile: scratch allocation failed");
  }
  // stored-image geometry: copy every stored position of the lower part back (padding included)
  TileGeo ge = geo_of(A);
  ge.mbu = ge.mbs;
  ge.m = (long)ge.lmt * ge.mbs;
  ge.n = (long)ge.lnt * ge.mbs;
  if (A->dtype == CHOL_REAL_DOUBLE) {
    launch_lauum_lower<double>(g.s_main, (const double *)A->mat, (double *)tmp, A->nt, A->mbi);
    launch_lacpy<double>(g.s_main, ge, 1, (const double *)tmp, (double *)A->mat);
  } else {
    launch_lauum_lower<float>(g.s_main, (const float *)A->mat, (float *)tmp, A->nt, A->mbi);
    launch_lacpy<float>(g.s_main, ge, 1, (const float *)tmp, (float *)A->mat);
  }
  hipError_t e = hipStreamSynchronize(g.s_main);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail_hip(e, "lauum_tile", __LINE__);
  return 0;
}

// ---------------------------------------------------------------- solve with the factor
int chol_potrs_tile(int uplo, chol_desc_t *A, chol_desc_t *B) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "potrs_tile before chol_init");
  if (uplo != CHOL_LOWER) return fail(CHOL_ERR_NOT_SUPPORTED, "potrs_tile: only ChamLower");
  int rc = resident_whole("potrs_tile", A);
  if (rc) return rc;
  rc = resident_whole("potrs_tile", B);
  if (rc) return rc;
  if (A->mt != A->nt || A->lm != A->ln) return fail(-2, "potrs_tile: A is not square");
  if (B->lm != A->lm || B->mb != A->mb || B->mbi != A->mbi || B->dtype != A->dtype)
    return fail(-3, "potrs_tile: B must have A's order, tile size and type");
  if (A->mbi % 64) return fail(CHOL_ERR_NOT_SUPPORTED, "potrs_tile: stored tile edge must be a multiple of 64");
  CHECK_WINV(A, "potrs_tile");
  std::lock_guard<std::mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? potrs_impl<double>(A, B) : potrs_impl<float>(A, B);
}

int chol_posv_tile(int uplo, chol_desc_t *A, chol_desc_t *B) {
  if (uplo != CHOL_LOWER) return fail(CHOL_ERR_NOT_SUPPORTED, "posv_tile: only ChamLower");
  const int info = chol_potrf_tile(uplo, A);
  if (info != 0) return info;  // > 0: not positive definite, B untouched (LAPACK dposv)
  return chol_potrs_tile(uplo, A, B);
}

// valid extent of tile (I,J) inside the matrix (edge tiles are smaller)
static inline int tile_rows(const chol_desc *d, int I) { return std::min(d->mb, d->lm - I * d->mb); }
static inline int tile_cols(const chol_desc *d, int J) { return std::min(d->nb, d->ln - J * d->nb); }

int chol_tile_upload(chol_desc_t *d, int I, int J, const void *host_tile) {
  if (!d || !host_tile) return fail(-1, "tile_upload: NULL");
  if (I < 0 || I >= d->mt || J < 0 || J >= d->nt || I % d->p != d->prow || J % d->q != d->pcol)
    return fail(-2, "tile_upload: tile not owned by this process");
  char *dst = reinterpret_cast<char *>(d->mat) +
              ((size_t)(I / d->p) + (size_t)(J / d->q) * d->lmt) * (size_t)d->bsizi * d->esize;
  // host tile: mb x nb, ld = mb (Chameleon tile); only the part inside the matrix is stored
  HIPCHECK(hipMemcpy2D(dst, (size_t)d->mbi * d->esize, host_tile, (size_t)d->mb * d->esize,
                       (size_t)tile_rows(d, I) * d->esize, tile_cols(d, J), hipMemcpyDefault));
  return 0;
}

int chol_tile_download(chol_desc_t *d, int I, int J, void *host_tile) {
  if (!d || !host_tile) return fail(-1, "tile_download: NULL");
  if (I < 0 || I >= d->mt || J < 0 || J >= d->nt || I % d->p != d->prow || J % d->q != d->pcol)
    return fail(-2, "tile_download: tile not owned by this process");
  const char *src = reinterpret_cast<const char *>(d->mat) +
                    ((size_t)(I / d->p) + (size_t)(J / d->q) * d->lmt) * (size_t)d->bsizi * d->esize;
  if (tile_rows(d, I) < d->mb || tile_cols(d, J) < d->nb) memset(host_tile, 0, (size_t)d->bsiz * d->esize);
  HIPCHECK(hipMemcpy2D(host_tile, (size_t)d->mb * d->esize, src, (size_t)d->mbi * d->esize,
                       (size_t)tile_rows(d, I) * d->esize, tile_cols(d, J), hipMemcpyDefault));
  return 0;
}

int chol_lapack_to_tile(const void *A, int lda, chol_desc_t *d) {
  if (!A || !d) return fail(-1, "lapack_to_tile: NULL");
  if (d->p * d->q != 1) return fail(CHOL_ERR_NOT_SUPPORTED, "lapack_to_tile: single-process descriptors only");
  if (lda < d->lm) return fail(-2, "lapack_to_tile: lda");
  for (int J = 0; J < d->nt; ++J)
    for (int I = 0; I < d->mt; ++I) {
      char *dst = reinterpret_cast<char *>(d->mat) + ((size_t)I + (size_t)J * d->lmt) * (size_t)d->bsizi * d->esize;
      const char *src = reinterpret_cast<const char *>(A) + ((size_t)I * d->mb + (size_t)J * d->nb * lda) * d->esize;
      HIPCHECK(hipMemcpy2D(dst, (size_t)d->mbi * d->esize, src, (size_t)lda * d->esize,
                           (size_t)tile_rows(d, I) * d->esize, tile_cols(d, J), hipMemcpyDefault));
    }
  return 0;
}

int chol_tile_to_lapack(chol_desc_t *d, void *A, int lda) {
  if (!A || !d) return fail(-1, "tile_to_lapack: NULL");
  if (d->p * d->q != 1) return fail(CHOL_ERR_NOT_SUPPORTED, "tile_to_lapack: single-process descriptors only");
  if (lda < d->lm) return fail(-3, "tile_to_lapack: lda");
  for (int J = 0; J < d->nt; ++J)
    for (int I = 0; I < d->mt; ++I) {
      const char *src = reinterpret_cast<const char *>(d->mat) + ((size_t)I + (size_t)J * d->lmt) * (size_t)d->bsizi * d->esize;
      char *dst = reinterpret_cast<char *>(A) + ((size_t)I * d->mb + (size_t)J * d->nb * lda) * d->esize;
      HIPCHECK(hipMemcpy2D(dst, (size_t)lda * d->esize, src, (size_t)d->mbi * d->esize,
                           (size_t)tile_rows(d, I) * d->esize, tile_cols(d, J), hipMemcpyDefault));
    }
  return 0;
}

static int residual_common(chol_desc_t *L, double bump, unsigned long long seed, double *rel_fro,
                           double *rel_inf) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "residual before chol_init");
  if (!L || (!rel_fro && !rel_inf)) return fail(-1, "residual: NULL");
  if (L->p * L->q != 1 || !L->on_device || L->mt != L->nt || L->mb != L->nb || L->mbi % MACRO)
    return fail(CHOL_ERR_NOT_SUPPORTED, "residual: single-process device-resident square tiled matrix only");
  std::lock_guard<std::mutex> lk(g_mu);
  const long n = L->lm;
  double *rows = nullptr;
  if (rel_inf) {
    HIPCHECK(hipMalloc(&rows, 2 * (size_t)n * sizeof(double)));
    HIPCHECK(hipMemsetAsync(rows, 0, 2 * (size_t)n * sizeof(double), g.s_main));
  }
  HIPCHECK(hipMemsetAsync(g.d_acc, 0, 2 * sizeof(double), g.s_main));
  if (L->dtype == CHOL_REAL_DOUBLE)
    launch_residual<double>(g.s_main, reinterpret_cast<const double *>(L->mat), L->nt, L->mbi, bump, seed,
                            g.d_acc, L->mb, n, rows);
  else
    launch_residual<float>(g.s_main, reinterpret_cast<const float *>(L->mat), L->nt, L->mbi, bump, seed,
                           g.d_acc, L->mb, n, rows);
  double h[2];
  HIPCHECK(hipMemcpyAsync(h, g.d_acc, sizeof h, hipMemcpyDeviceToHost, g.s_main));
  HIPCHECK(hipStreamSynchronize(g.s_main));
  if (rel_fro) *rel_fro = sqrt(h[0]) / sqrt(h[1]);
  if (rel_inf) {
    std::vector<double> hr(2 * (size_t)n);
    hipError_t e = hipMemcpy(hr.data(), rows, hr.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(rows);
    if (e != hipSuccess) return fail_hip(e, "residual row sums", __LINE__);
    double rmax = 0, amax = 0;
    for (long r = 0; r < n; ++r) {
      rmax = std::max(rmax, hr[r]);
      amax = std::max(amax, hr[n + r]);
    }
    *rel_inf = rmax / (amax > 0 ? amax : 1.0);  // V6:84
  }
  return 0;
}

int chol_residual_plgsy(chol_desc_t *L, double bump, unsigned long long seed, double *rel) {
  return residual_common(L, bump, seed, rel, nullptr);
}

int chol_residual_plgsy_inf(chol_desc_t *L, double bump, unsigned long long seed, double *rel_inf) {
  return residual_common(L, bump, seed, nullptr, rel_inf);
}

// ---------------------------------------------------------------- instrumentation
int chol_last_potrf_stats(double *total_ms, double *update_ms, int *update_launches, double *update_flops) {
  if (total_ms) *total_ms = g.total_ms;
  if (update_ms) *update_ms = g.update_ms;
  if (update_launches) *update_launches = g.update_launches;
  if (update_flops) *update_flops = g.update_flops;
  return 0;
}

int chol_set_profiling(int on) {
  g.profiling = on != 0;
  return 0;
}

int chol_bench_update(chol_desc_t *d, int k, int ablate, int reps, double *ms, double *flops) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "bench_update before chol_init");
  if (!d || !ms || d->p * d->q != 1 || !d->on_device || d->mt != d->nt || k < 0 || k + 1 >= d->nt || reps < 1)
    return fail(-1, "bench_update: arguments");
#ifndef CHOLMI_DIAGNOSTICS
  if (ablate != 0)
    return fail(CHOL_ERR_NOT_SUPPORTED, "bench_update: the ablation twin of the update is a diagnostic build (make DIAG=1)");
#endif
  std::lock_guard<std::mutex> lk(g_mu);
  int rc = ensure_events(2);
  if (rc) return rc;
  PanelRef pan;
  memset(&pan, 0, sizeof pan);
  pan.P = 1;
  pan.base[0] = (char *)d->mat + (size_t)k * d->nt * d->bsizi * d->esize;
  const LocalMat C = local_mat(d, d->mat);
  const ColRange rr = col_range(d, k + 1, d->nt);
  cholmi::g_ablate = ablate;
  float best = 1e30f;
  for (int r = 0; r <= reps; ++r) {
    HIPCHECK(hipEventRecord(g.events[0], g.s_main));
    if (d->dtype == CHOL_REAL_DOUBLE)
      launch_trail_update<double>(g.s_main, C, d->d_list, rr.off, rr.na, rr.offb, rr.nb, pan);
    else
      launch_trail_update<float>(g.s_main, C, d->d_list, rr.off, rr.na, rr.offb, rr.nb, pan);
    HIPCHECK(hipEventRecord(g.events[1], g.s_main));
    HIPCHECK(hipStreamSynchronize(g.s_main));
    float t = 0;
    HIPCHECK(hipEventElapsedTime(&t, g.events[0], g.events[1]));
    if (r > 0 && t < best) best = t;
  }
  cholmi::g_ablate = 0;
  *ms = best;
  const double ntl = (double)(d->nt - 1 - k);
  if (flops) *flops = (ntl * (ntl - 1) + ntl) * (double)d->mbi * d->mbi * d->mbi;
  return 0;
}

int chol_debug_stamps(int enable, unsigned long long *out, int max_pairs) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "debug_stamps before chol_init");
  static unsigned long long *buf = nullptr;
  const size_t bytes = (1 + 8 * 1000) * sizeof(unsigned long long);
  HIPCHECK(hipDeviceSynchronize());
  if (enable) {
    if (!buf) HIPCHECK(hipMalloc(&buf, bytes));
    HIPCHECK(hipMemset(buf, 0, bytes));
    cholmi::g_dbg = buf;
    return 0;
  }
  int n = 0;
  if (buf && out) {
    std::vector<unsigned long long> h(1 + 8 * 1000);
    HIPCHECK(hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost));
    n = (int)std::min<unsigned long long>(h[0], (unsigned long long)std::min(max_pairs, 1000));
    memcpy(out, h.data() + 1, (size_t)n * 8 * sizeof(unsigned long long));
  }
  cholmi::g_dbg = nullptr;
  return n;
}

int chol_mfma_probe(int dtype, int waves_per_simd, double *tflops) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "mfma_probe before chol_init");
  if (!tflops || waves_per_simd < 1 || waves_per_simd > 8) return fail(-2, "mfma_probe: arguments");
  std::lock_guard<std::mutex> lk(g_mu);
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, g.device));
  const int blocks = prop.multiProcessorCount * waves_per_simd, iters = 4000;
  int rc = ensure_stage(0, (size_t)blocks * 256 * sizeof(double));
  if (rc) return rc;
  rc = ensure_events(2);
  if (rc) return rc;
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {
    HIPCHECK(hipEventRecord(g.events[0], g.s_main));
    if (dtype == CHOL_REAL_DOUBLE)
      launch_mfma_probe<double>(g.s_main, (double *)g.stage[0], blocks, iters);
    else
      launch_mfma_probe<float>(g.s_main, (float *)g.stage[0], blocks, iters);
    HIPCHECK(hipEventRecord(g.events[1], g.s_main));
    HIPCHECK(hipStreamSynchronize(g.s_main));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, g.events[0], g.events[1]));
    const double fl = (double)blocks * 4 * iters * 16 * 2048.0;
    if (rep > 0) best = std::max(best, fl / (ms * 1e-3) / 1e12);
  }
  *tflops = best;
  return 0;
}

// ---------------------------------------------------------------- distributed building blocks
int chol_get_info(int *info) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "get_info before chol_init");
  HIPCHECK(hipDeviceSynchronize());
  return read_info(info);
}

int chol_reset_info(void) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "reset_info before chol_init");
  HIPCHECK(hipMemset(g.d_info, 0, sizeof(int)));
  return 0;
}

int chol_wave_potrf(chol_desc_t *d, int k, void *lkk, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_potrf before chol_init");
  if (!d || !lkk) return fail(-1, "wave_potrf: NULL");
  CHECK_WINV(d, "wave_potrf");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CHOL_REAL_DOUBLE)
    launch_potrf_tile<double>(s, (double *)lkk, d->mbi, (double *)g.winv, g.d_info, k * d->mbi, tile_sems());
  else
    launch_potrf_tile<float>(s, (float *)lkk, d->mbi, (float *)g.winv, g.d_info, k * d->mbi, tile_sems());
  HIPCHECK(hipGetLastError());
  return 0;
}

int chol_wave_invert_diag(chol_desc_t *d, void *lkk, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_invert_diag before chol_init");
  if (!d || !lkk) return fail(-1, "wave_invert_diag: NULL");
  CHECK_WINV(d, "wave_invert_diag");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CHOL_REAL_DOUBLE)
    launch_invert_diag<double>(s, (const double *)lkk, d->mbi, (double *)g.winv);
  else
    launch_invert_diag<float>(s, (const float *)lkk, d->mbi, (float *)g.winv);
  HIPCHECK(hipGetLastError());
  return 0;
}

// Move the 128-block inverses of the current L(k,k) (written by chol_wave_potrf) out of /
// into the context workspace, so that they can travel with the broadcast of L(k,k) instead of
// being recomputed on every rank of the process column.
size_t chol_wave_winv_bytes(chol_desc_t *d) {
  return d ? (size_t)(d->mbi / MACRO) * MACRO * MACRO * d->esize : 0;
}

int chol_wave_export_winv(chol_desc_t *d, void *dst, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_export_winv before chol_init");
  if (!d || !dst) return fail(-1, "wave_export_winv: NULL");
  HIPCHECK(hipMemcpyAsync(dst, g.winv, chol_wave_winv_bytes(d), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int chol_wave_import_winv(chol_desc_t *d, const void *src, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_import_winv before chol_init");
  if (!d || !src) return fail(-1, "wave_import_winv: NULL");
  HIPCHECK(hipMemcpyAsync(g.winv, src, chol_wave_winv_bytes(d), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int chol_wave_trsm(chol_desc_t *d, int k, const void *lkk, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_trsm before chol_init");
  if (!d || !lkk) return fail(-1, "wave_trsm: NULL");
  CHECK_WINV(d, "wave_trsm");
  if (k % d->q != d->pcol) return 0;  // this process column holds no tile of panel k
  hipStream_t s = (hipStream_t)stream;
  const int il0 = (k + d->p - d->prow) / d->p;  // first local row with global index > k
  const int cnt = d->lmt - il0;
  if (cnt <= 0) return 0;
  const size_t off = ((size_t)il0 + (size_t)(k / d->q) * d->lmt) * (size_t)d->bsizi;
  if (d->dtype == CHOL_REAL_DOUBLE)
    launch_trsm_panel<double>(s, (double *)d->mat + off, d->bsizi, cnt, (const double *)lkk,
                              (const double *)g.winv, d->mbi, 1.0);
  else
    launch_trsm_panel<float>(s, (float *)d->mat + off, d->bsizi, cnt, (const float *)lkk,
                             (const float *)g.winv, d->mbi, 1.0f);
  HIPCHECK(hipGetLastError());
  return 0;
}

static int wave_update_range(chol_desc_t *d, const void *const *panel_base, const int *panel_first,
                             const ColRange &r, hipStream_t s) {
  if (r.na + r.nb <= 0) return 0;
  PanelRef pan;
  memset(&pan, 0, sizeof pan);
  pan.P = d->p;
  for (int q = 0; q < d->p; ++q) {
    pan.base[q] = panel_base[q];
    pan.first[q] = panel_first[q];
  }
  const LocalMat C = local_mat(d, d->mat);
  if (d->dtype == CHOL_REAL_DOUBLE)
    launch_trail_update<double>(s, C, d->d_list, r.off, r.na, r.offb, r.nb, pan, true);
  else
    launch_trail_update<float>(s, C, d->d_list, r.off, r.na, r.offb, r.nb, pan, true);
  HIPCHECK(hipGetLastError());
  return 0;
}

static inline bool owns_tile(const chol_desc *d, int I, int J) {
  return I % d->p == d->prow && J % d->q == d->pcol;
}

int chol_wave_update(chol_desc_t *d, int k, int jlo, int jhi, const void *const *panel_base,
                     const int *panel_first, int skip_diag, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_update before chol_init");
  if (!d || !panel_base || !panel_first) return fail(-1, "wave_update: NULL");
  if (jlo <= k) jlo = k + 1;
  if (jhi > d->nt) jhi = d->nt;
  if (jlo >= jhi) return 0;
  ColRange r = col_range(d, jlo, jhi);
  // diagonal tiles are listed by column descending: that of column jlo, when this process owns
  // it, is the last one of the segment
  if (skip_diag && owns_tile(d, jlo, jlo)) --r.nb;
  return wave_update_range(d, panel_base, panel_first, r, (hipStream_t)stream);
}

int chol_wave_update_diag(chol_desc_t *d, int k, int j, const void *const *panel_base,
                          const int *panel_first, void *stream) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "wave_update_diag before chol_init");
  if (!d || !panel_base || !panel_first) return fail(-1, "wave_update_diag: NULL");
  if (j <= k || j >= d->nt) return fail(-3, "wave_update_diag: j out of range");
  if (!owns_tile(d, j, j)) return fail(-3, "wave_update_diag: this process does not own tile (j,j)");
  if (cholmi::g_intile_small) {
    const int pr = j % d->p;
    const size_t aoff = (size_t)(j / d->p - panel_first[pr]) * d->bsizi, coff = ((size_t)(j / d->p) + (size_t)(j / d->q) * d->lmt) * d->bsizi;
    if (d->dtype == CHOL_REAL_DOUBLE)
      launch_diag_syrk<double>((hipStream_t)stream, (double *)d->mat + coff, (const double *)panel_base[pr] + aoff, d->mbi);
    else
      launch_diag_syrk<float>((hipStream_t)stream, (float *)d->mat + coff, (const float *)panel_base[pr] + aoff, d->mbi);
    HIPCHECK(hipGetLastError());
    return 0;
  }
  ColRange r;
  r.off = 0;
  r.na = 0;
  r.offb = d->n_off + d->gd[j + 1];
  r.nb = 1;
  return wave_update_range(d, panel_base, panel_first, r, (hipStream_t)stream);
}

}  // extern "C"
