// C ABI of libcholmi.so (include/cholmi.h): process-global context, Chameleon-style
// descriptors, the synchronous tile operations the reference worker calls
// (worker_distrib.cpp:238, 323, 416, 511) and the whole-matrix tiled POTRF the
// reference driver calls (v6_test.c:56), run as the reference client's wave DAG
// (client_distrib.cpp:506-565) on two HIP streams with one wave of lookahead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/cholmi.h"
#include "cholmi_internal.h"

using namespace cholmi;

// dist.hip: the wave walker on a whole tiled matrix -- one GPU (p = q = 1) or this rank's share of a p x q grid
extern "C" int chol_internal_walk(chol_desc *d, void *base, cholmi::RankCtx *r, int rank);
extern "C" void chol_internal_dist_finalize(void);
extern "C" int chol_internal_desc_create(chol_desc_t **desc, void *mat, int dtype, int mb, int nb, int bsiz, int lm, int ln,
                                         int i, int j, int m, int n, int p, int q, int my_rank, int nranks);

namespace {

constexpr int BINFO_SLOTS = 4096;

struct Ctx {
  bool inited = false;
  int rank = 0, nranks = 1;
  RankCtx r;              // streams, workspaces, counters, events of this process's rank
  int *d_ytab = nullptr;  // per-CU yield requests (kernels.hip: cooperative CU hand-over); one per device
  double *d_acc = nullptr;
  void *stage[3] = {nullptr, nullptr, nullptr};
  size_t stage_bytes[3] = {0, 0, 0};
  void *work = nullptr;   // reusable scratch of lange / potrs / tile batches (grown on demand)
  size_t work_bytes = 0;
  // block inverses of the last single tile factored or inverted under a version tag (chol_desc_set_version):
  // the TRSM tasks of a wave all use the L(k,k) the POTRF task before them produced
  const void *wc_ptr = nullptr;
  unsigned long long wc_version = 0;
  int wc_mb = 0, wc_dtype = 0;
  void *wc_winv = nullptr;
  int *d_binfo = nullptr;  // device info words of asynchronously factored tiles (chol_potrf_batch), a ring
  unsigned binfo_next = 0;
  std::string last_error;
};

// ---------------------------------------------------------------------------------------------------------------
// The executor behind the grouped launches of the task API (chol_tile_batch / chol_potrf_batch): dependency-driven on
// TWO streams.  The reference's tasks name their data dependencies (C2:489-499), so a scheduler may start POTRF(k+1)
// as soon as SYRK(k+1,k+1,k) is done; here the batches of the panel chain (POTRF, TRSM, and the updates the caller
// marks CHOL_BATCH_URGENT: column k+1) go to the high-priority chain stream, all other updates to the bulk stream, and
// what orders two batches is what they read and write: every output tile pointer is remembered with the (stream,
// sequence number) of the batch that writes it; a batch that reads a tile written on the OTHER stream waits for that
// batch's event first (one wait per batch at most: the newest such producer).  Results are write-once buffers (a
// batch's outputs are fresh memory), so there is no write-after-read edge to track; chol_sync() drains both streams
// and forgets everything.  Walker-style overlap follows: wave k+1's chain runs beside wave k's bulk update.
// ---------------------------------------------------------------------------------------------------------------
struct TaskExec {
  enum { CHAIN = 0, BULK = 1, EVPOOL = 64 };
  bool ready = false, dirty = false;
  hipStream_t st[2] = {nullptr, nullptr};
  unsigned long long seq[2] = {0, 0}, done[2] = {0, 0};
  unsigned long long waited[2][2] = {{0, 0}, {0, 0}};  // [consumer][producer]: newest batch of `producer` the consumer stream already follows
  hipEvent_t ev[2][EVPOOL] = {};
  unsigned long long ev_seq[2][EVPOOL] = {};
  struct Prod {
    int stream;
    unsigned long long seq;
  };
  std::unordered_map<const void *, Prod> prod;
  // pointer lists of the batches: a pinned host ring mirrored into a device ring, one slice per batch
  char *h_ring = nullptr, *d_ring = nullptr;
  size_t ring_bytes = (size_t)16 << 20, ring_pos = 0;
  struct Slice {
    size_t begin, end;
    int stream;
    unsigned long long seq;
  };
  std::deque<Slice> slices;
  void *winv = nullptr;  // block inverses of the POTRF batches (two sets by parity of the POTRF count), chain stream only
  unsigned long long npotrf = 0;
  long long batches[2] = {0, 0}, cross_waits = 0;  // statistics: batches per stream, event waits between them
  // A POTRF batch of ONE tile is not launched at once: the TRSM batch that follows it in every wave of the DAG (same L,
  // outputs back to back) then goes out WITH it, pipelined over two streams as in the whole-matrix walker -- TRSM step s
  // needs only the diagonal-block step s (launch_panel_pipelined), so a wave's chain is about POTRF + one TRSM step
  // instead of POTRF + TRSM.  Anything else that comes first (another batch, a sync, a query) launches it alone.
  struct Pending {
    bool on = false;
    int dtype = 0, mb = 0, slot = 0;
    const void *a_in = nullptr;
    void *a_out = nullptr;
    unsigned long long version = 0;
  } pend;
  hipStream_t st_trsm = nullptr;       // the TRSM steps of a pipelined panel; joined back into the chain stream after it
  hipEvent_t pev[36] = {};             // its per-step events (tiles up to 4096: 32 steps), then: pre, join
  long long fused_panels = 0;
};
TaskExec tx;
// The executor's ORDERING checked without a GPU (chol_debug_task_record / chol_debug_task_check, tests/
// test_schedule_check.py): in recording mode the same code runs -- dependency tracking, the pending-POTRF / pipelined
// panel logic, ring slices, marks -- but every HIP call is skipped and every batch is logged with the tiles it reads
// and writes, its stream and sequence number, the event waits issued for it and what the host knew to be complete.
// The check: any two batches that touch the same tile, one of them writing it, must be ordered by stream order, by a
// chain of event waits, or by a host-side wait.  `mutate` >= 0 drops that one event wait (the checker's self-test).
struct TxRecord {
  struct Op {
    int stream;
    unsigned long long seq, base[2];  // base: the batches of each stream the host had waited for when this one was issued
    std::vector<const void *> rd, wr;
    std::vector<std::pair<int, unsigned long long>> waits;  // (producer stream, sequence number) of each event wait
  };
  bool on = false;
  int mutate = -1;
  long long nwaits = 0;
  std::vector<Op> ops;
  std::vector<const void *> rd;                                  // reads followed since the last commit
  std::vector<std::pair<int, unsigned long long>> pending[2];   // waits issued on a stream since its last batch
} tx_rec;
#define TXHIP(call)                 \
  do {                              \
    if (!tx_rec.on) HIPCHECK(call); \
  } while (0)
int tx_quiesce();
const double g_task_yield_factor = 6.0;  // (swept 3 ... 9 in round 5: flat above 5)
Ctx g;
std::recursive_mutex g_mu;     // one ABI call at a time on the context
std::mutex g_err_mu; // chol_last_error's buffer

void set_error(const char *msg) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g.last_error = msg;
}
int fail_hip(hipError_t e, const char *what, int line) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s failed at api.hip:%d: %s", what, line, hipGetErrorString(e));
  set_error(buf);
  return CHOL_ERR_HIP;
}
#define HIPCHECK(call)                                        \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return fail_hip(e_, #call, __LINE__); \
  } while (0)

int fail(int code, const char *msg) {
  set_error(msg);
  return code;
}

inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

// counters for one single-tile POTRF (launch_potrf_tile): a set of 32, rotating so that consecutive
// factorisations on different streams never share one
int *tile_sems() {
  return g.r.d_sem ? g.r.d_sem + (size_t)(SEM_SLOTS + 32 * (g.r.tile_sem_next++ % TILE_SEM_SETS)) * 32 : nullptr;
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

int ensure_work(size_t bytes) {
  if (g.work_bytes >= bytes) return 0;
  if (g.work) HIPCHECK(hipFree(g.work));
  g.work = nullptr;
  g.work_bytes = 0;
  bytes = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  HIPCHECK(hipMalloc(&g.work, bytes));
  g.work_bytes = bytes;
  return 0;
}

// the cached block inverses for (tile pointer, version), or null
const void *cached_winv(const void *ptr, unsigned long long version, int mb, int dtype) {
  if (!version || !g.wc_winv || g.wc_ptr != ptr || g.wc_version != version || g.wc_mb != mb || g.wc_dtype != dtype) return nullptr;
  return g.wc_winv;
}
// remember the block inverses in g.r.winv (first set) for (ptr, version); stream-ordered on ST_MAIN
int remember_winv(const void *ptr, unsigned long long version, int mb, int dtype, size_t bytes) {
  if (!version) return 0;
  if (!g.wc_winv) HIPCHECK(hipMalloc(&g.wc_winv, g.r.winv_bytes));
  HIPCHECK(hipMemcpyAsync(g.wc_winv, g.r.winv, bytes, hipMemcpyDeviceToDevice, g.r.st[ST_MAIN]));
  g.wc_ptr = ptr;
  g.wc_version = version;
  g.wc_mb = mb;
  g.wc_dtype = dtype;
  return 0;
}
void forget_winv(const void *ptr) {
  if (g.wc_ptr == ptr) g.wc_ptr = nullptr, g.wc_version = 0;
}

// ---- the task executor (TaskExec above) ------------------------------------------------------------------------
int tx_init() {
  if (tx.ready) return 0;
  if (tx_rec.on) {  // (recording: no streams, no events; the pointer-list ring is plain host memory)
    tx.h_ring = static_cast<char *>(malloc(tx.ring_bytes));
    tx.d_ring = tx.h_ring;
    if (!tx.h_ring) return fail(CHOL_ERR_HIP, "task recorder: out of memory");
    tx.ready = true;
    return 0;
  }
  tx.st[TaskExec::CHAIN] = g.r.st[ST_PANEL];
  tx.st[TaskExec::BULK] = g.r.st[ST_MAIN];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < TaskExec::EVPOOL; ++i) HIPCHECK(hipEventCreateWithFlags(&tx.ev[s][i], hipEventDisableTiming));
  HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&tx.h_ring), tx.ring_bytes, hipHostMallocDefault));
  HIPCHECK(hipMalloc(reinterpret_cast<void **>(&tx.d_ring), tx.ring_bytes));
  HIPCHECK(hipMalloc(&tx.winv, 2 * g.r.winv_bytes));
  tx.st_trsm = g.r.st[ST_TRSM];
  for (int i = 0; i < 36; ++i) HIPCHECK(hipEventCreateWithFlags(&tx.pev[i], hipEventDisableTiming));
  tx.ready = true;
  return 0;
}
void tx_destroy() {
  if (!tx.ready) return;
  if (tx_rec.on) {
    free(tx.h_ring);
    tx = TaskExec();
    return;
  }
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < TaskExec::EVPOOL; ++i)
      if (tx.ev[s][i]) (void)hipEventDestroy(tx.ev[s][i]);
  for (int i = 0; i < 36; ++i)
    if (tx.pev[i]) (void)hipEventDestroy(tx.pev[i]);
  (void)hipHostFree(tx.h_ring);
  (void)hipFree(tx.d_ring);
  (void)hipFree(tx.winv);
  tx = TaskExec();
}
// everything the executor has enqueued is complete afterwards; nothing is remembered
int tx_flush_pending();
int tx_quiesce() {
  if (!tx.ready || !tx.dirty) return 0;
  if (int rc = tx_flush_pending()) return rc;
  TXHIP(hipStreamSynchronize(tx.st[TaskExec::CHAIN]));
  TXHIP(hipStreamSynchronize(tx.st[TaskExec::BULK]));
  for (int s = 0; s < 2; ++s) {
    tx.done[s] = tx.seq[s];
    tx.waited[s][0] = tx.waited[s][1] = 0;
  }
  tx.prod.clear();
  tx.slices.clear();
  tx.dirty = false;
  return 0;
}
// (defined further down; the cached block inverses of the executor die with the context)

// the host waits until batch `seq` of `stream` has run (an event of the pool that was recorded at or after it)
int tx_wait_host(int stream, unsigned long long seq) {
  if (tx.done[stream] >= seq) return 0;
  const int slot = (int)(seq % TaskExec::EVPOOL);
  TXHIP(hipEventSynchronize(tx.ev[stream][slot]));
  tx.done[stream] = std::max(tx.done[stream], tx.ev_seq[stream][slot]);
  return 0;
}
// a slice of the pointer-list rings (same offset on the host and on the device side)
int tx_take(size_t bytes, size_t *off) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes > tx.ring_bytes) return fail(CHOL_ERR_NOT_SUPPORTED, "tile batch: more tasks than one grouped launch takes (split the batch)");
  if (tx.ring_pos + bytes > tx.ring_bytes) tx.ring_pos = 0;
  const size_t b = tx.ring_pos, e = b + bytes;
  for (size_t i = 0; i < tx.slices.size();) {  // slices of earlier batches under the new one: their batches must have run
    const TaskExec::Slice sl = tx.slices[i];
    const bool over = sl.begin < e && b < sl.end;
    if (over) {
      if (int rc = tx_wait_host(sl.stream, sl.seq)) return rc;
    }
    if (over || tx.done[sl.stream] >= sl.seq) tx.slices.erase(tx.slices.begin() + (long)i);
    else ++i;
  }
  tx.ring_pos = e;
  *off = b;
  return 0;
}
// stream X follows the batches (of the other stream) that write the n tiles of each of the `nl` pointer lists
int tx_follow(int X, const void *const *const *lists, int nl, int n) {
  const int Y = 1 - X;
  unsigned long long need = 0;
  if (tx_rec.on)
    for (int l = 0; l < nl; ++l)
      for (int t = 0; lists[l] && t < n; ++t)
        if (lists[l][t]) tx_rec.rd.push_back(lists[l][t]);
  if (!tx.prod.empty())
    for (int l = 0; l < nl; ++l) {
      if (!lists[l]) continue;
      for (int t = 0; t < n; ++t) {
        const void *p = lists[l][t];
        if (!p) continue;
        auto it = tx.prod.find(p);
        if (it != tx.prod.end() && it->second.stream == Y && it->second.seq > need) need = it->second.seq;
      }
    }
  if (need > tx.waited[X][Y]) {
    if (tx.done[Y] < need) {
      // (the pool's event of that slot stands for batch ev_seq >= need of stream Y: a later batch if the slot has been
      // recorded again since -- a longer wait, never a shorter one)
      if (tx_rec.on) {
        if (tx_rec.nwaits++ != tx_rec.mutate) tx_rec.pending[X].push_back({Y, tx.ev_seq[Y][need % TaskExec::EVPOOL]});
      } else {
        HIPCHECK(hipStreamWaitEvent(tx.st[X], tx.ev[Y][need % TaskExec::EVPOOL], 0));
      }
      ++tx.cross_waits;
    }
    tx.waited[X][Y] = need;
  }
  return 0;
}
// the batch just enqueued on X: its event, its ring slice, the tiles it writes
int tx_commit(int X, void *const *outs, int n, size_t slice_b, size_t slice_e) {
  const unsigned long long seq = ++tx.seq[X];
  const int slot = (int)(seq % TaskExec::EVPOOL);
  TXHIP(hipEventRecord(tx.ev[X][slot], tx.st[X]));
  tx.ev_seq[X][slot] = seq;
  if (tx_rec.on) {
    TxRecord::Op op;
    op.stream = X, op.seq = seq, op.base[0] = tx.done[0], op.base[1] = tx.done[1];
    op.rd.swap(tx_rec.rd);
    op.wr.assign(outs, outs + n);
    op.waits.swap(tx_rec.pending[X]);
    tx_rec.ops.push_back(std::move(op));
  }
  for (int t = 0; t < n; ++t) tx.prod[outs[t]] = TaskExec::Prod{X, seq};
  if (slice_e > slice_b) tx.slices.push_back(TaskExec::Slice{slice_b, slice_e, X, seq});
  ++tx.batches[X];
  tx.dirty = true;
  return 0;
}

// block inverses of the executor's POTRF / TRSM batches: two sets, alternating, all on the chain stream; the last one
// written is remembered under (tile pointer, content tag) for the TRSM batch that names the same L
void *tx_winv_set() { return reinterpret_cast<char *>(tx.winv) + (size_t)(tx.npotrf++ & 1) * g.r.winv_bytes; }
struct TxWinv {
  const void *ptr = nullptr, *set = nullptr;
  unsigned long long version = 0;
  int mb = 0, dtype = 0;
} tx_wc;
const void *tx_cached_winv(const void *ptr, unsigned long long version, int mb, int dtype) {
  if (!version || tx_wc.ptr != ptr || tx_wc.version != version || tx_wc.mb != mb || tx_wc.dtype != dtype) return nullptr;
  return tx_wc.set;
}
void tx_remember_winv(const void *ptr, unsigned long long version, int mb, int dtype, const void *set) {
  tx_wc = TxWinv();
  if (!version) return;
  tx_wc.ptr = ptr, tx_wc.version = version, tx_wc.mb = mb, tx_wc.dtype = dtype, tx_wc.set = set;
}

int ensure_events(size_t n) {
  while (g.r.events.size() < n) {
    hipEvent_t e;
    HIPCHECK(hipEventCreate(&e));
    g.r.events.push_back(e);
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
  return (size_t)(roundup(d->mbi, MACRO) / MACRO) * MACRO * MACRO * d->esize <= g.r.winv_bytes;
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
  if (B != Bp) HIPCHECK(hipMemsetAsync(dst, 0, bytes, g.r.st[ST_MAIN]));
  HIPCHECK(hipMemcpy2DAsync(dst, (size_t)Bp * sizeof(T), d->mat, (size_t)B * sizeof(T),
                            (size_t)B * sizeof(T), B, hipMemcpyDefault, g.r.st[ST_MAIN]));
  if (identity_pad) launch_pad_identity<T>(g.r.st[ST_MAIN], dst, B, Bp);
  out->dev = dst;
  out->in_place = false;
  return 0;
}

template <typename T>
int stage_out(const chol_desc *d, const Staged &st) {
  if (st.in_place) return 0;
  const int B = d->mb;
  HIPCHECK(hipMemcpy2DAsync(d->mat, (size_t)B * sizeof(T), st.dev, (size_t)st.ldp * sizeof(T),
                            (size_t)B * sizeof(T), B, hipMemcpyDefault, g.r.st[ST_MAIN]));
  return 0;
}

int read_info(int *info) {
  HIPCHECK(hipMemcpy(info, g.r.d_info, sizeof(int), hipMemcpyDeviceToHost));
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

// ---- sub-matrix views over a user buffer: the library works on a compact image of the view, tiled on its own.
// The user's matrix holds whole mb x nb tiles of bsiz elements (ld = mb, tile (TI, TJ) at (TI + TJ lmt) bsiz); the
// image's tiles are mbi x mbi (mb rounded up to 128, identity outside the matrix).  Entry (r, c) of the view is entry
// (i + r, j + c) of the user's matrix: an image tile overlaps up to 2 x 2 user tiles (the view may start inside one --
// v3_script_cholesky_x_arg_gpt.c:141-142, 186-212 accept any offset), each overlap one strided copy.
int view_sync(chol_desc *d, bool in) {
  const size_t es = d->esize, ldu = (size_t)d->mb * es, ldi = (size_t)d->mbi * es;
  const size_t ti = (size_t)d->bsizi * es, tu = (size_t)d->bsiz * es;
  for (int J = 0; J < d->nt; ++J)
    for (int I = 0; I < d->mt; ++I) {
      char *img = reinterpret_cast<char *>(d->mat) + ((size_t)I + (size_t)J * d->lmt) * ti;
      const int rows = std::min(d->mb, d->lm - I * d->mb), cols = std::min(d->nb, d->ln - J * d->nb);
      for (int c0 = 0; c0 < cols;) {
        const long C = (long)d->user_j + (long)J * d->nb + c0;  // user column
        const int tj = (int)(C / d->nb), cj = (int)(C % d->nb), nc = std::min(cols - c0, d->nb - cj);
        for (int r0 = 0; r0 < rows;) {
          const long R = (long)d->user_i + (long)I * d->mb + r0;
          const int tr = (int)(R / d->mb), ri = (int)(R % d->mb), nr = std::min(rows - r0, d->mb - ri);
          char *u = reinterpret_cast<char *>(d->user_mat) + ((size_t)tr + (size_t)tj * d->user_lmt) * tu + ((size_t)ri + (size_t)cj * d->mb) * es;
          char *m = img + ((size_t)r0 + (size_t)c0 * d->mbi) * es;
          HIPCHECK(hipMemcpy2D(in ? m : u, in ? ldi : ldu, in ? u : m, in ? ldu : ldi, (size_t)nr * es, nc, hipMemcpyDefault));
          r0 += nr;
        }
        c0 += nc;
      }
    }
  return 0;
}
// One entry point's body between the refresh of its views' images and their write-back, all under the context
// lock (recursive: the bodies take it again); a failed refresh returns before the body runs, a failed write-back
// is reported unless the body already failed.  The write-back also follows a body that returned info > 0
// (a partly factored matrix, as LAPACK leaves it).
struct ViewArg {
  chol_desc *d;
  bool write_back;
};
template <typename F>
static int with_views(std::initializer_list<ViewArg> views, F &&body) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (int rc = tx_quiesce()) return rc;  // (the synchronous calls run on ST_MAIN alone: nothing of the task executor may be in flight)
  for (const ViewArg &v : views)
    if (v.d && v.d->user_mat) {
      const int rc = view_sync(v.d, true);
      if (rc) return rc;
    }
  int rc = body();
  for (const ViewArg &v : views)
    if (v.d && v.d->user_mat && v.write_back) {
      const int r2 = view_sync(v.d, false);
      if (r2 && rc >= 0) rc = r2;
    }
  return rc;
}

int build_worklist(chol_desc *d) {
  if (d->mt != d->nt) return 0;  // only square tile grids are factored
  std::vector<int2> off, dg;
  d->ge.assign(d->nt + 2, 0);
  d->gd.assign(d->nt + 2, 0);
  for (int J = d->nt - 1; J >= 0; --J) {
    if (J % d->q == d->pcol)
      for (int I = J; I < d->mt; ++I)
        if (I % d->p == d->prow) (I == J ? dg : off).push_back(make_int2(I, J));
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
    HIPCHECK(hipMemsetAsync(g.r.d_info, 0, sizeof(int), g.r.st[ST_MAIN]));
    // ChamUpper on a staged tile: A = U^T U with U = L^T -- transpose the staged (identity-padded,
    // multiple-of-128) copy, factor Lower, transpose back: the caller's strict lower triangle comes
    // back exactly as it went in
    if (upper_staged) launch_transpose_inplace<T>(g.r.st[ST_MAIN], reinterpret_cast<T *>(st.dev), 1, st.ldp);
    launch_potrf_tile<T>(g.r.st[ST_MAIN], reinterpret_cast<T *>(st.dev), st.ldp, reinterpret_cast<T *>(g.r.winv),
                         g.r.d_info, 0, tile_sems());
    if (upper_staged) launch_transpose_inplace<T>(g.r.st[ST_MAIN], reinterpret_cast<T *>(st.dev), 1, st.ldp);
    // an in-place device tile with a version tag: its block inverses serve the TRSM tasks that name the same content
    if (st.in_place && !upper_staged)
      rc = remember_winv(A->mat, A->version, st.ldp, A->dtype, (size_t)(st.ldp / MACRO) * MACRO * MACRO * sizeof(T));
    else
      forget_winv(A->mat);
    if (rc) return rc;
    rc = stage_out<T>(A, st);
    if (rc) return rc;
    HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
    int info = 0;
    rc = read_info(&info);
    if (!rc && info != 0) forget_winv(A->mat);
    return rc ? rc : info;
  }
  if (A->mt != A->nt || A->lm != A->ln) return fail(-2, "potrf_tile: matrix is not square");
  // one walker for every whole-matrix descriptor: a single GPU is the p = q = 1 case of the block-cyclic schedule
  if (A->p * A->q != 1 || A->on_device) return chol_internal_walk(A, A->mat, &g.r, A->p * A->q != 1 ? g.rank : 0);
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
    rc = chol_internal_walk(A, dev, &g.r, 0);
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
  // the block inverses of L: those the POTRF task of the same content left behind (chol_desc_set_version), else
  // recomputed from the tile (mb/128 diagonal-block launches) and kept under L's tag for the TRSMs that follow
  const T *winv = sl.in_place ? reinterpret_cast<const T *>(cached_winv(L->mat, L->version, sl.ldp, L->dtype)) : nullptr;
  if (!winv) {
    T *w = reinterpret_cast<T *>(g.r.winv);
    launch_invert_diag<T>(g.r.st[ST_MAIN], reinterpret_cast<const T *>(sl.dev), sl.ldp, w);
    if (sl.in_place) {
      rc = remember_winv(L->mat, L->version, sl.ldp, L->dtype, (size_t)(sl.ldp / MACRO) * MACRO * MACRO * sizeof(T));
      if (rc) return rc;
    }
    winv = w;
  }
  forget_winv(B->mat);  // (B is overwritten)
  launch_trsm_panel<T>(g.r.st[ST_MAIN], reinterpret_cast<T *>(sb.dev), (long)sb.ldp * sb.ldp, 1,
                       reinterpret_cast<const T *>(sl.dev), winv, sb.ldp, (T)alpha);
  rc = stage_out<T>(B, sb);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
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
  forget_winv(C->mat);
  launch_gemm_nt_tile<T>(g.r.st[ST_MAIN], reinterpret_cast<const T *>(sa.dev), reinterpret_cast<const T *>(sb.dev),
                         reinterpret_cast<T *>(sc.dev), sc.ldp, (T)alpha, (T)beta, lower);
  rc = stage_out<T>(C, sc);
  if (rc) return rc;
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
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
  if (ensure_work(((size_t)nr * nt + 2 + (size_t)nr + (size_t)nt) * tb)) {
    (void)hipGetLastError();
    return fail(CHOL_ERR_OUT_OF_MEMORY, "potrs_tile: scratch allocation failed");
  }
  T *scr = reinterpret_cast<T *>(g.work);
  T *Z = scr, *Wt = scr + (size_t)nr * nt * bs, *Tt = Wt + bs, *tmp = Tt + (size_t)nt * bs;  // Tt: nt tiles, tmp: nr tiles
  hipStream_t s = g.r.st[ST_MAIN];
  T *winv = reinterpret_cast<T *>(g.r.winv);
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
  HIPCHECK(hipGetLastError());  // (a refused launch configuration must not come back as a wrong solution)
  HIPCHECK(hipStreamSynchronize(s));
  return 0;
}

// ---- calibration: what the walker's regime switches are measured in (walker.h: WaveCalib) ----------------
// One 128 x 128 diagonal-block step alone and the register-only MFMA stream, per dtype, timed once per context.
template <typename T>
static int time_diag_step(RankCtx *r, double *us) {
  T *tile = nullptr, *winv = reinterpret_cast<T *>(r->winv);
  HIPCHECK(hipMalloc(&tile, (size_t)MACRO * MACRO * sizeof(T)));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    // a well-conditioned SPD block: 4 I (the kernel's time does not depend on the values)
    HIPCHECK(hipMemsetAsync(tile, 0, (size_t)MACRO * MACRO * sizeof(T), r->st[ST_PANEL]));
    launch_pad_identity<T>(r->st[ST_PANEL], tile, 0, MACRO);
    HIPCHECK(hipEventRecord(e0, r->st[ST_PANEL]));
    launch_potrf_tile<T>(r->st[ST_PANEL], tile, MACRO, winv, r->d_info, 0, nullptr);
    HIPCHECK(hipEventRecord(e1, r->st[ST_PANEL]));
    HIPCHECK(hipStreamSynchronize(r->st[ST_PANEL]));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(tile);
  HIPCHECK(hipMemset(r->d_info, 0, sizeof(int)));
  *us = best * 1e3;
  return 0;
}

static int mfma_probe_on(RankCtx *r, int dtype, int waves_per_simd, int iters, double *tflops) {
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, r->device));
  const int blocks = prop.multiProcessorCount * waves_per_simd;
  void *buf = nullptr;
  HIPCHECK(hipMalloc(&buf, (size_t)blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {
    HIPCHECK(hipEventRecord(e0, r->st[ST_MAIN]));
    if (dtype == CHOL_REAL_DOUBLE)
      launch_mfma_probe<double>(r->st[ST_MAIN], (double *)buf, blocks, iters);
    else
      launch_mfma_probe<float>(r->st[ST_MAIN], (float *)buf, blocks, iters);
    HIPCHECK(hipEventRecord(e1, r->st[ST_MAIN]));
    HIPCHECK(hipStreamSynchronize(r->st[ST_MAIN]));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)blocks * 4 * iters * 16 * 2048.0;
    if (rep > 0) best = std::max(best, fl / (ms * 1e-3) / 1e12);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  *tflops = best;
  return 0;
}

}  // namespace

namespace cholmi {

void *DevPool::get(size_t bytes) {
  for (auto &b : blks)
    if (!b.used && b.bytes >= bytes) {
      b.used = true;
      return b.p;
    }
  void *p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  blks.push_back({p, bytes, true});
  return p;
}
void DevPool::free_all() {
  for (auto &b : blks) (void)hipFree(b.p);
  blks.clear();
}

RankCtx *main_rank_ctx() { return &g.r; }
int main_rank(int *nranks) {
  if (nranks) *nranks = g.nranks;
  return g.rank;
}

// Streams, workspaces, counters of one rank on `device` (already current).  calib_from: take the measured
// rates from another context of the same device instead of measuring again.
int rank_ctx_create(RankCtx *r, int device, const RankCtx *calib_from) {
  r->device = device;
  int lo = 0, hi = 0;
  HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIPCHECK(hipStreamCreateWithPriority(&r->st[ST_MAIN], hipStreamNonBlocking, lo));
  HIPCHECK(hipStreamCreateWithPriority(&r->st[ST_PANEL], hipStreamNonBlocking, hi));
  HIPCHECK(hipStreamCreateWithPriority(&r->st[ST_TRSM], hipStreamNonBlocking, hi));
  // the update of column k+1 (what the next panel waits for) runs beside the rest of the wave's
  // update, ahead of it in priority
  HIPCHECK(hipStreamCreateWithPriority(&r->st[ST_U1], hipStreamNonBlocking, lo - 1 > hi ? lo - 1 : hi));
  // the two communication streams of a p x q factorisation (walker.h); idle otherwise
  HIPCHECK(hipStreamCreateWithPriority(&r->st[ST_CX], hipStreamNonBlocking, hi));
  HIPCHECK(hipStreamCreateWithPriority(&r->st[ST_PX], hipStreamNonBlocking, hi));
  {
    // The panel chain's streams must not share a dispatch pipe with a stream that carries trailing-update launches
    // (ST_MAIN; ST_U1): such a launch, longer than one round of workgroups, holds back every kernel of a stream behind
    // the same pipe until its last workgroup is placed (kernels.hip: k_pipe_big).  Probe the pairs; replace a victim
    // by a fresh stream (the old one is kept, so that its queue is not handed out again) and probe again.
    {
      hipDeviceProp_t prop;
      HIPCHECK(hipGetDeviceProperties(&prop, device));
      unsigned long long *t = nullptr;
      HIPCHECK(hipMalloc(&t, 64));
      auto blocked = [&](int big, int small, bool *out) -> int {
        HIPCHECK(hipMemset(t, 0, 64));
        cholmi::launch_pipe_probe(r->st[big], r->st[small], t, prop.multiProcessorCount);
        HIPCHECK(hipStreamSynchronize(r->st[big]));
        HIPCHECK(hipStreamSynchronize(r->st[small]));
        unsigned long long h[2];
        HIPCHECK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
        *out = h[1] > h[0] + 3000;  // started more than 30 us behind the big launch's first workgroup: it waited for rounds
        return 0;
      };
      const int victims[4] = {ST_PANEL, ST_TRSM, ST_CX, ST_U1};
      const int mid = lo - 1 > hi ? lo - 1 : hi;
      const int prio_of[ST_COUNT] = {lo, hi, hi, mid, hi, hi};
      for (int vi = 0; vi < 4; ++vi) {
        const int v = victims[vi];
        for (int attempt = 0;; ++attempt) {
          bool b1 = false, b2 = false;
          int rc = blocked(ST_MAIN, v, &b1);
          if (rc) return rc;
          if (v != ST_U1) rc = blocked(ST_U1, v, &b2);
          if (rc) return rc;
          if (!b1 && !b2) break;
          if (attempt == 6) {
            ++r->stream_collisions;
            break;
          }
          r->retired.push_back(r->st[v]);
          HIPCHECK(hipStreamCreateWithPriority(&r->st[v], hipStreamNonBlocking, prio_of[v]));
          ++r->stream_swaps;
        }
      }
      (void)hipFree(t);
      if (getenv("CHOLMI_VERBOSE"))
        fprintf(stderr, "[cholmi] stream pipe probe: %d stream(s) replaced, %d pair(s) still colliding\n", r->stream_swaps, r->stream_collisions);
    }
  }
  r->winv_bytes = (size_t)32 * MACRO * MACRO * sizeof(double);  // tiles up to 4096
  HIPCHECK(hipMalloc(&r->winv, 2 * r->winv_bytes));  // two sets: the walker alternates them by wave parity
  HIPCHECK(hipMalloc(&r->d_info, sizeof(int)));
  HIPCHECK(hipMemset(r->d_info, 0, sizeof(int)));
  {
    // Device-side edges of the panel chain: usable only if a kernel polling on one of the panel streams does
    // not keep the others from running (streams that share a hardware queue would deadlock until the poll's
    // bound).  Probed once, both ways, consumer launched first.
    const char *e = getenv("CHOLMI_DEVICE_FLAGS");
    if (!e || atoi(e) != 0) {
      HIPCHECK(hipMalloc(&r->d_sem, SEM_INTS * sizeof(int)));
      HIPCHECK(hipMemset(r->d_sem, 0, SEM_INTS * sizeof(int)));
      bool ok = true;
      const hipStream_t pairs[4][2] = {{r->st[ST_TRSM], r->st[ST_PANEL]}, {r->st[ST_U1], r->st[ST_TRSM]},
                                       {r->st[ST_PANEL], r->st[ST_U1]}, {r->st[ST_PANEL], r->st[ST_TRSM]}};
      for (int t = 0; t < 4 && ok; ++t) {  // {consumer, producer}: the three edges of SyrkPipe, and sp / st the other way
        int *sem = r->d_sem + 64 * t, *res = r->d_sem + 64 * t + 32;
        cholmi::launch_sem_probe(pairs[t][0], pairs[t][1], sem, res);
        HIPCHECK(hipStreamSynchronize(pairs[t][0]));
        HIPCHECK(hipStreamSynchronize(pairs[t][1]));
        int v = 0;
        HIPCHECK(hipMemcpy(&v, res, sizeof(int), hipMemcpyDeviceToHost));
        ok = (v == 1);
      }
      if (!ok) {
        (void)hipFree(r->d_sem);
        r->d_sem = nullptr;
      } else {
        // the flow form of the tile POTRF runs its row-slab kernel on ST_CX (idle on one GPU; on a grid it carries
        // the small messages, which follow the POTRF anyway): that stream and ST_PANEL must not share a queue either
        r->flow_ok = true;
        const hipStream_t fpairs[2][2] = {{r->st[ST_CX], r->st[ST_PANEL]}, {r->st[ST_PANEL], r->st[ST_CX]}};
        for (int t = 0; t < 2 && r->flow_ok; ++t) {
          int *sem = r->d_sem + 64 * (4 + t), *res = r->d_sem + 64 * (4 + t) + 32;
          cholmi::launch_sem_probe(fpairs[t][0], fpairs[t][1], sem, res);
          HIPCHECK(hipStreamSynchronize(fpairs[t][0]));
          HIPCHECK(hipStreamSynchronize(fpairs[t][1]));
          int v = 0;
          HIPCHECK(hipMemcpy(&v, res, sizeof(int), hipMemcpyDeviceToHost));
          r->flow_ok = (v == 1);
        }
        HIPCHECK(hipMemset(r->d_sem, 0, 1024 * sizeof(int)));
      }
    }
  }
  if (calib_from) {
    for (int i = 0; i < 2; ++i) r->probe_tflops[i] = calib_from->probe_tflops[i], r->diag_us[i] = calib_from->diag_us[i];
    return 0;
  }
  // CHOLMI_CALIB="tf64,us64,tf32,us32": fixed values instead of measured ones (reproducible schedules)
  if (const char *e = getenv("CHOLMI_CALIB")) {
    double v[4];
    if (sscanf(e, "%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3]) == 4) {
      r->probe_tflops[0] = v[0], r->diag_us[0] = v[1], r->probe_tflops[1] = v[2], r->diag_us[1] = v[3];
      return 0;
    }
  }
  int rc = mfma_probe_on(r, CHOL_REAL_DOUBLE, 4, 1000, &r->probe_tflops[0]);  // (also brings the clocks up)
  if (!rc) rc = mfma_probe_on(r, CHOL_REAL_FLOAT, 4, 2000, &r->probe_tflops[1]);
  if (!rc) rc = time_diag_step<double>(r, &r->diag_us[0]);
  if (!rc) rc = time_diag_step<float>(r, &r->diag_us[1]);
  return rc;
}

void rank_ctx_destroy(RankCtx *r) {
  for (auto e : r->events) (void)hipEventDestroy(e);
  r->events.clear();
  if (r->ev_flow) (void)hipEventDestroy(r->ev_flow);
  r->ev_flow = nullptr;
  r->pool.free_all();
  if (r->winv) (void)hipFree(r->winv);
  if (r->d_info) (void)hipFree(r->d_info);
  if (r->d_sem) (void)hipFree(r->d_sem);
  r->winv = nullptr;
  r->d_info = nullptr;
  r->d_sem = nullptr;
  for (int i = 0; i < ST_COUNT; ++i) {
    if (r->st[i]) (void)hipStreamDestroy(r->st[i]);
    r->st[i] = nullptr;
  }
  for (hipStream_t s : r->retired) (void)hipStreamDestroy(s);
  r->retired.clear();
}

}  // namespace cholmi

extern "C" {

int chol_internal_fail(int code, const char *msg) { return fail(code, msg); }

const char *chol_version(void) { return "cholmi 0.3 (gfx950)"; }

// (a copy per thread: the buffer behind the pointer never changes under the caller)
const char *chol_last_error(void) {
  static thread_local std::string copy;
  std::lock_guard<std::mutex> lk(g_err_mu);
  copy = g.last_error;
  return copy.c_str();
}

int chol_set_device(int device) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g.inited && device != g.r.device) return fail(-1, "chol_set_device after chol_init");
  g.r.device = device;
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
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g.inited) return 0;
  if (ngpu < 1)
    return fail(CHOL_ERR_NO_GPU, "chol_init: ngpu must be >= 1 (this library has no CPU backend)");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count < 1) {
    (void)hipGetLastError();
    return fail(CHOL_ERR_NO_GPU, "chol_init: no HIP device visible");
  }
  if (g.r.device < 0) {
    const char *lr = getenv("LOCAL_RANK");
    g.r.device = lr ? atoi(lr) % count : 0;
  }
  if (g.r.device >= count) return fail(CHOL_ERR_NO_GPU, "chol_init: device index out of range");
  HIPCHECK(hipSetDevice(g.r.device));
  if (const char *e = getenv("CHOLMI_TRSM_SMALL_MAX")) cholmi::g_trsm_small_max = atoi(e);
  {
    // grids that may poll a counter themselves: few enough that, one per CU in the worst case, most CUs
    // stay free of pollers (kernels.hip: k_sem_gate); on a small partition every grid waits behind a gate
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, g.r.device));
    const int room = (prop.multiProcessorCount - 32) / 4;
    cholmi::g_poll_max_wgs = room < 0 ? 0 : (room < 48 ? room : 48);
  }
  if (const char *e = getenv("CHOLMI_POLL_MAX_WGS")) cholmi::g_poll_max_wgs = atoi(e);
  if (const char *e = getenv("CHOLMI_FLOW")) cholmi::g_flow = atoi(e);
  if (const char *e = getenv("CHOLMI_FLOW_NBM")) {  // "lo:hi": tiles of lo .. hi 128-blocks may use the flow form (2 <= lo, hi <= 8)
    int lo = 0, hi = 0;
    if (sscanf(e, "%d:%d", &lo, &hi) == 2) cholmi::g_flow_min_nbm = std::max(2, lo), cholmi::g_flow_max_nbm = std::min(8, hi);
  }
  if (const char *e = getenv("CHOLMI_FLOW_FENCES")) cholmi::g_flow_fences = atoi(e);
  if (const char *e = getenv("CHOLMI_INTILE_FUSED")) cholmi::g_intile_fused = atoi(e);
  if (const char *e = getenv("CHOLMI_MIN_UNITS")) cholmi::g_min_units = atoi(e);
  HIPCHECK(hipMalloc(&g.d_acc, 2 * sizeof(double)));
  HIPCHECK(hipMalloc(&g.d_ytab, YTAB_ENTRIES * sizeof(int)));
  HIPCHECK(hipMemset(g.d_ytab, 0, YTAB_ENTRIES * sizeof(int)));
  cholmi::g_ytab = g.d_ytab;
  int rc = rank_ctx_create(&g.r, g.r.device, nullptr);
  if (rc) return rc;
  g.inited = true;
  return 0;
}

int chol_finalize(void) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (!g.inited) return 0;
  (void)hipDeviceSynchronize();
  chol_internal_dist_finalize();
  for (int i = 0; i < 3; ++i) {
    if (g.stage[i]) (void)hipFree(g.stage[i]);
    g.stage[i] = nullptr;
    g.stage_bytes[i] = 0;
  }
  if (g.work) (void)hipFree(g.work);
  g.work = nullptr;
  g.work_bytes = 0;
  if (g.wc_winv) (void)hipFree(g.wc_winv);
  g.wc_winv = nullptr;
  if (g.d_binfo) (void)hipFree(g.d_binfo);
  g.d_binfo = nullptr;
  tx_destroy();
  g.wc_ptr = nullptr;
  g.wc_version = 0;
  (void)hipFree(g.d_acc);
  if (g.d_ytab) (void)hipFree(g.d_ytab);
  g.d_ytab = nullptr;
  cholmi::g_ytab = nullptr;
  rank_ctx_destroy(&g.r);
  g.d_acc = nullptr;
  g.inited = false;
  return 0;
}

int chol_desc_create(chol_desc_t **desc, void *mat, int dtype, int mb, int nb, int bsiz, int lm,
                     int ln, int i, int j, int m, int n, int p, int q) {
  return chol_internal_desc_create(desc, mat, dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q, g.rank, g.nranks);
}

// (rank, nranks: chol_set_rank's values -- or those of one rank of the one-GPU rehearsal, dist.hip)
int chol_internal_desc_create(chol_desc_t **desc, void *mat, int dtype, int mb, int nb, int bsiz, int lm, int ln,
                              int i, int j, int m, int n, int p, int q, int my_rank, int nranks) {
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
  void *view_user = nullptr;
  int view_lmt = 0, view_oi = 0, view_oj = 0;
  if (i != 0 || j != 0 || m != lm || n != ln) {
    // A sub-matrix view A(i:i+m, j:j+n) of an lm x ln matrix (V6:24-25, 44-45 pass ioff, joff, m, n
    // straight through).  With library-owned storage nothing outside the view can ever be observed
    // through this descriptor -- Chameleon's generator, factorisation and layout conversions all work in
    // view coordinates (dplgsy: entry (r, c) of the view, order m) -- so the view IS an m x n matrix of
    // its own: only the tiles it can address are allocated (tile (0,0) of the descriptor = the first mb x nb entries of
    // the VIEW, wherever it starts).
    if (p * q != 1) return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: sub-matrix views of distributed matrices are not supported");
    if (mat) {
      // over a user buffer (v3's --mat user with offsets): the user's matrix is lm x ln in whole mb x nb tiles; the view
      // may start anywhere in it and have any extent.  The library keeps a compact image of the view, tiled on its
      // own, and mirrors it around every operation (with_views / view_sync)
      if (lm % mb || ln % nb || mb != nb)
        return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: a view over a user buffer needs a user matrix of whole square tiles (lm, ln multiples of mb = nb)");
      view_user = mat;
      view_lmt = lm / mb;
      view_oi = i;
      view_oj = j;
      mat = nullptr;
    }
    // (library-owned storage: nothing outside the view can be observed, so an unaligned offset means nothing either)
    lm = m;
    ln = n;
    i = j = 0;
  }
  if (p * q != 1 && p * q != nranks)
    return fail(CHOL_ERR_NOT_SUPPORTED, "desc_create: p*q must equal the number of ranks (chol_set_rank)");
  chol_desc *d = new chol_desc();
  d->dtype = dtype; d->mb = mb; d->nb = nb; d->bsiz = bsiz; d->lm = lm; d->ln = ln;
  d->i = i; d->j = j; d->m = m; d->n = n; d->p = p; d->q = q;
  d->esize = dtype == CHOL_REAL_DOUBLE ? 8 : 4;
  d->mt = (lm + mb - 1) / mb;
  d->nt = (ln + nb - 1) / nb;
  const int rank = (p * q == 1) ? 0 : my_rank;
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
    // and the identity outside the matrix.  Needs library-owned storage (each rank of a p x q grid its own).
    if (mat || mb != nb)
      return delete d, fail(CHOL_ERR_NOT_SUPPORTED,
                            "desc_create: ragged / non-128 tiles need mat == NULL and square tiles");
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
        launch_plgsy<double>(g.r.st[ST_MAIN], Lm, d->lnt, d->prow, d->pcol, 0.0, 0ull, 0, 0, 0);
      else
        launch_plgsy<float>(g.r.st[ST_MAIN], Lm, d->lnt, d->prow, d->pcol, 0.0, 0ull, 0, 0, 0);
      (void)hipStreamSynchronize(g.r.st[ST_MAIN]);
    } else {  // library-owned storage starts at zero (tiles a one-sided dplgsy does not touch)
      (void)hipMemsetAsync(d->mat, 0, bytes, g.r.st[ST_MAIN]);
      (void)hipStreamSynchronize(g.r.st[ST_MAIN]);
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
  d->user_mat = view_user;
  d->user_lmt = view_lmt;
  d->user_i = view_oi;
  d->user_j = view_oj;
  *desc = d;
  return 0;
}

int chol_desc_destroy(chol_desc_t **desc) {
  if (!desc || !*desc) return fail(-1, "desc_destroy: NULL");
  chol_desc *d = *desc;
  if (d->owns) forget_winv(d->mat);
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

int chol_desc_set_version(chol_desc_t *d, unsigned long long version) {
  if (!d) return fail(-1, "desc_set_version: NULL descriptor");
  d->version = version;
  return 0;
}

int chol_sync(void) {
  if (tx_rec.on) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return tx_quiesce();
  }
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "chol_sync before chol_init");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
  return tx_quiesce();
}

}  // extern "C"

// ---------------------------------------------------------------- the tasks of one op class in one grouped launch
// (the executor: TaskExec / tx_* above)

// the pending POTRF (TaskExec::Pending) alone, as chol_potrf_batch would have launched it
template <typename T>
static int flush_pending_t() {
  const TaskExec::Pending p = tx.pend;
  tx.pend.on = false;
  hipStream_t s = tx.st[TaskExec::CHAIN];
  TXHIP(hipMemsetAsync(g.d_binfo + p.slot, 0, sizeof(int), s));
  TXHIP(hipMemcpyAsync(p.a_out, p.a_in, (size_t)p.mb * p.mb * sizeof(T), hipMemcpyDeviceToDevice, s));  // the private copy (W2:212-213)
  T *wset = reinterpret_cast<T *>(tx_winv_set());
  if (!tx_rec.on) launch_potrf_tile<T>(s, reinterpret_cast<T *>(p.a_out), p.mb, wset, g.d_binfo + p.slot, 0, tile_sems());
  forget_winv(p.a_out);
  tx_remember_winv(p.a_out, p.version, p.mb, p.dtype, wset);
  TXHIP(hipGetLastError());
  void *outs[1] = {p.a_out};
  return tx_commit(TaskExec::CHAIN, outs, 1, 0, 0);
}
namespace {
int tx_flush_pending() {
  if (!tx.ready || !tx.pend.on) return 0;
  return tx.pend.dtype == CHOL_REAL_DOUBLE ? flush_pending_t<double>() : flush_pending_t<float>();
}
}  // namespace

// The pending POTRF and the n TRSM tasks of its panel in one pipelined issue (kernels.hip: launch_panel_pipelined): the
// diagonal-block steps on the chain stream, TRSM step s on the TRSM stream behind the event of step s; the TRSM stream
// starts behind an event of the chain stream (it follows everything the chain stream follows) and is joined back into
// it at the end, so that for the dependency tracker the whole panel is ONE batch of the chain stream.
template <typename T>
static int panel_fused_impl(int dtype, int mb, int n, const void *const *c_in, void *const *c_out) {
  const TaskExec::Pending p = tx.pend;
  tx.pend.on = false;
  const int X = TaskExec::CHAIN, nbm = mb / MACRO;
  hipStream_t sp = tx.st[X], st = tx.st_trsm;
  const size_t tb = (size_t)mb * mb * sizeof(T), lb = (size_t)n * sizeof(void *);
  size_t off = 0;
  int rc = tx_take(2 * lb, &off);
  if (rc) return rc;
  char *hl = tx.h_ring + off, *dl = tx.d_ring + off;
  memcpy(hl, c_in, lb);
  memcpy(hl + lb, c_out, lb);
  const void *const *ins[1] = {c_in};
  rc = tx_follow(X, ins, 1, n);  // (the POTRF's own input was followed when it was stashed)
  if (rc) return rc;
  // (the TRSM stream inside this issue is not the recorder's business: it starts behind an event of the chain stream
  // and is joined back into it, so the tracker -- and the recorder -- see ONE batch of the chain stream; the launches
  // inside are launch_panel_pipelined's, the whole-matrix walker's, whose ordering csrc/sched_check.hip covers)
  hipEvent_t ev_pre = tx.pev[34], ev_join = tx.pev[35];
  TXHIP(hipEventRecord(ev_pre, sp));
  TXHIP(hipStreamWaitEvent(st, ev_pre, 0));
  // the private copies: the diagonal tile on the chain stream, the panel tiles on the TRSM stream (beside the first steps)
  TXHIP(hipMemsetAsync(g.d_binfo + p.slot, 0, sizeof(int), sp));
  TXHIP(hipMemcpyAsync(p.a_out, p.a_in, tb, hipMemcpyDeviceToDevice, sp));
  TXHIP(hipMemcpyAsync(dl, hl, 2 * lb, hipMemcpyHostToDevice, st));
  T *wset = reinterpret_cast<T *>(tx_winv_set());
  if (!tx_rec.on) {
    launch_copy_ptrs(st, (const void *const *)dl, (void *const *)(dl + lb), n, (long)tb);
    launch_panel_pipelined<T>(sp, st, tx.pev, reinterpret_cast<T *>(p.a_out), mb, wset, g.d_binfo + p.slot, 0,
                              reinterpret_cast<T *>(c_out[0]), (long)mb * mb, n, nullptr, nullptr, nullptr, 0, tile_sems());
  }
  (void)nbm;
  TXHIP(hipEventRecord(ev_join, st));
  TXHIP(hipStreamWaitEvent(sp, ev_join, 0));
  forget_winv(p.a_out);
  for (int t = 0; t < n; ++t) forget_winv(c_out[t]);
  tx_remember_winv(p.a_out, p.version, mb, dtype, wset);
  TXHIP(hipGetLastError());
  ++tx.fused_panels;
  void *outs1[1] = {p.a_out};
  rc = tx_commit(X, outs1, 1, 0, 0);
  if (rc) return rc;
  if (tx_rec.on) tx_rec.rd.push_back(p.a_out);  // (the solves read the factor just written: same stream)
  return tx_commit(X, c_out, n, off, off + 2 * lb);
}

template <typename T>
static int tile_batch_impl(int op, int dtype, int mb, int n, const void *const *c_in, const void *const *a, const void *const *b,
                           void *const *c_out, const unsigned long long *a_versions, int flags) {
  int rc = tx_init();
  if (rc) return rc;
  const bool update = op != CHOL_BATCH_TRSM;
  if (tx.pend.on) {
    // the TRSM batch of the pending POTRF's own panel: every task solves against that tile, outputs back to back
    bool fuse = !update && tx.pend.dtype == dtype && tx.pend.mb == mb;
    for (int t = 0; fuse && t < n; ++t)
      fuse = a[t] == tx.pend.a_out && (t == 0 || (const char *)c_out[t] == (const char *)c_out[t - 1] + (size_t)mb * mb * sizeof(T));
    if (fuse) return panel_fused_impl<T>(dtype, mb, n, c_in, c_out);
    rc = tx_flush_pending();
    if (rc) return rc;
  }
  const int X = (!update || (flags & CHOL_BATCH_URGENT)) ? TaskExec::CHAIN : TaskExec::BULK;
  hipStream_t s = tx.st[X];
  const size_t tb = (size_t)mb * mb * sizeof(T), lb = (size_t)n * sizeof(void *);
  // the four pointer lists: pinned host slice -> device slice, in stream order ahead of the kernels that read them
  size_t off = 0;
  rc = tx_take(4 * lb, &off);
  if (rc) return rc;
  char *hl = tx.h_ring + off, *dl = tx.d_ring + off;
  memcpy(hl, c_in, lb);
  memcpy(hl + lb, a, lb);
  if (op == CHOL_BATCH_SYRK || !b) memset(hl + 2 * lb, 0, lb);  // (null: a SYRK task)
  else memcpy(hl + 2 * lb, b, lb);
  memcpy(hl + 3 * lb, c_out, lb);
  const void *const *ins[3] = {c_in, a, op == CHOL_BATCH_SYRK ? nullptr : b};
  rc = tx_follow(X, ins, 3, n);
  if (rc) return rc;
  TXHIP(hipMemcpyAsync(dl, hl, 4 * lb, hipMemcpyHostToDevice, s));
  const void *const *d_cin = (const void *const *)dl, *const *d_a = (const void *const *)(dl + lb),
                    *const *d_b = (const void *const *)(dl + 2 * lb);
  void *const *d_out = (void *const *)(dl + 3 * lb);
  for (int t = 0; t < n; ++t) forget_winv(c_out[t]);
  if (update) {
    // ONE out-of-place launch: c_out = c_in - a b^T (the private copy of W2:212-213 is this write).  The update's waves
    // yield their CU to the chain's guest kernels when the batch is short against a panel chain (the walker's rule)
    const int f = dtype == CHOL_REAL_DOUBLE ? 0 : 1;
    const double t_tile = 2.0 * mb * (double)mb * mb / (g.r.probe_tflops[f] * CHOLMI_UPDATE_EFF * 1e12);
    const double t_panel = g.r.diag_us[f] * 1e-6 * CHOLMI_STEP_FACTOR * (mb / MACRO);
    // (6 panel estimates, not the walker's 3: the task path's chain is POTRF, THEN the TRSM, THEN column k+1 -- about twice
    // the walker's pipelined one -- and beside an update that does not yield its diagonal-block steps take 230-370 us
    // instead of 50: profiles/r05_worker_path_trace_N16384_B512.txt)
    const bool yield = X == TaskExec::BULK && g.r.probe_tflops[f] > 0 && n * t_tile < g_task_yield_factor * t_panel;
    if (!tx_rec.on)
      launch_update_ptrs<T>(s, (const T *const *)d_cin, (const T *const *)d_a, (const T *const *)d_b, (T *const *)d_out, n, mb, yield);
  } else {
    // TRSM: private copies, then runs of consecutive tasks with the same L and outputs laid out back to back as panels
    if (!tx_rec.on) launch_copy_ptrs(s, d_cin, d_out, n, (long)tb);
    int t0 = 0;
    while (t0 < n) {
      int t1 = t0 + 1;
      while (t1 < n && a[t1] == a[t0] && (const char *)c_out[t1] == (const char *)c_out[t1 - 1] + tb) ++t1;
      const T *L = reinterpret_cast<const T *>(a[t0]);
      // the block inverses the POTRF task of this L left behind, else recomputed
      const T *w = reinterpret_cast<const T *>(tx_cached_winv(a[t0], a_versions ? a_versions[t0] : 0, mb, dtype));
      // (... or a synchronous chol_potrf_tile on the tagged tile: complete, and not overwritten while this batch runs --
      // every synchronous call waits for the executor first)
      if (!w) w = reinterpret_cast<const T *>(cached_winv(a[t0], a_versions ? a_versions[t0] : 0, mb, dtype));
      if (!w) {
        T *wn = reinterpret_cast<T *>(tx_winv_set());
        if (!tx_rec.on) launch_invert_diag<T>(s, L, mb, wn);
        tx_remember_winv(a[t0], a_versions ? a_versions[t0] : 0, mb, dtype, wn);
        w = wn;
      }
      if (!tx_rec.on) launch_trsm_panel<T>(s, reinterpret_cast<T *>(c_out[t0]), (long)mb * mb, t1 - t0, L, w, mb, T(1));
      t0 = t1;
    }
  }
  TXHIP(hipGetLastError());
  return tx_commit(X, c_out, n, off, off + 4 * lb);
}

// POTRF of n HBM-resident tiles, each on its private copy, info per tile in a device slot (read after chol_sync)
template <typename T>
static int potrf_batch_impl(int dtype, int mb, int n, const void *const *a_in, void *const *a_out,
                            const unsigned long long *versions, int *slots, bool defer) {
  int rc = tx_init();
  if (rc) return rc;
  const int X = TaskExec::CHAIN;
  hipStream_t s = tx.st[X];
  const size_t tb = (size_t)mb * mb * sizeof(T);
  if (!g.d_binfo && !tx_rec.on) {
    HIPCHECK(hipMalloc(&g.d_binfo, BINFO_SLOTS * sizeof(int)));
    HIPCHECK(hipMemset(g.d_binfo, 0, BINFO_SLOTS * sizeof(int)));
  }
  rc = tx_flush_pending();
  if (rc) return rc;
  const void *const *ins[1] = {a_in};
  rc = tx_follow(X, ins, 1, n);
  if (rc) return rc;
  if (n == 1 && defer) {  // (see TaskExec::Pending: launched with the TRSM batch that follows, or alone by whatever comes first)
    const int slot = (int)(g.binfo_next++ % BINFO_SLOTS);
    slots[0] = slot;
    tx.pend.on = true;
    tx.pend.dtype = dtype, tx.pend.mb = mb, tx.pend.slot = slot;
    tx.pend.a_in = a_in[0], tx.pend.a_out = a_out[0];
    tx.pend.version = versions ? versions[0] : 0;
    tx.dirty = true;
    return 0;
  }
  for (int t = 0; t < n; ++t) {
    const int slot = (int)(g.binfo_next++ % BINFO_SLOTS);
    slots[t] = slot;
    TXHIP(hipMemsetAsync(g.d_binfo + slot, 0, sizeof(int), s));
    TXHIP(hipMemcpyAsync(a_out[t], a_in[t], tb, hipMemcpyDeviceToDevice, s));  // the private copy (W2:212-213)
    T *wset = reinterpret_cast<T *>(tx_winv_set());
    if (!tx_rec.on) launch_potrf_tile<T>(s, reinterpret_cast<T *>(a_out[t]), mb, wset, g.d_binfo + slot, 0, tile_sems());
    forget_winv(a_out[t]);
    tx_remember_winv(a_out[t], versions ? versions[t] : 0, mb, dtype, wset);
  }
  TXHIP(hipGetLastError());
  return tx_commit(X, a_out, n, 0, 0);
}

extern "C" {

int chol_potrf_batch(int dtype, int mb, int n, const void *const *a_in, void *const *a_out,
                     const unsigned long long *versions, int *slots, int flags) {
  if (!g.inited && !tx_rec.on) return fail(CHOL_ERR_NOT_INITIALIZED, "potrf_batch before chol_init");
  if (dtype != CHOL_REAL_DOUBLE && dtype != CHOL_REAL_FLOAT) return fail(-1, "potrf_batch: dtype");
  if (mb <= 0 || mb % MACRO || mb > 4096) return fail(CHOL_ERR_NOT_SUPPORTED, "potrf_batch: tile edge must be a multiple of 128, at most 4096");
  if (n < 0 || n > BINFO_SLOTS / 2) return fail(-3, "potrf_batch: n");
  if (n == 0) return 0;
  if (!a_in || !a_out || !slots) return fail(-4, "potrf_batch: NULL pointer list");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const bool defer = (flags & CHOL_BATCH_ASYNC) != 0;
  const int rc = dtype == CHOL_REAL_DOUBLE ? potrf_batch_impl<double>(dtype, mb, n, a_in, a_out, versions, slots, defer)
                                           : potrf_batch_impl<float>(dtype, mb, n, a_in, a_out, versions, slots, defer);
  if (rc) return rc;
  if (!(flags & CHOL_BATCH_ASYNC)) return tx_quiesce();
  return 0;
}

int chol_batch_info(int slot, int *info) {
  if (tx_rec.on) {  // (recording: nothing ran)
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (info) *info = 0;
    return tx_flush_pending();
  }
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "batch_info before chol_init");
  if (!info || slot < 0 || slot >= BINFO_SLOTS || !g.d_binfo) return fail(-1, "batch_info: slot");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (int rc = tx_flush_pending()) return rc;
  if (tx.ready) HIPCHECK(hipStreamSynchronize(tx.st[TaskExec::CHAIN]));  // (every POTRF batch runs on the chain stream)
  HIPCHECK(hipMemcpy(info, g.d_binfo + slot, sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

int chol_tile_batch(int op, int dtype, int mb, int n, const void *const *c_in, const void *const *a,
                    const void *const *b, void *const *c_out, const unsigned long long *a_versions, int flags) {
  if (!g.inited && !tx_rec.on) return fail(CHOL_ERR_NOT_INITIALIZED, "tile_batch before chol_init");
  if (op != CHOL_BATCH_TRSM && op != CHOL_BATCH_SYRK && op != CHOL_BATCH_GEMM && op != CHOL_BATCH_UPDATE) return fail(-1, "tile_batch: op");
  if (dtype != CHOL_REAL_DOUBLE && dtype != CHOL_REAL_FLOAT) return fail(-2, "tile_batch: dtype");
  if (mb <= 0 || mb % MACRO || mb > 4096) return fail(CHOL_ERR_NOT_SUPPORTED, "tile_batch: tile edge must be a multiple of 128, at most 4096");
  if (n < 0) return fail(-4, "tile_batch: n");
  if (n == 0) return 0;
  if (!c_in || !a || !c_out || ((op == CHOL_BATCH_GEMM || op == CHOL_BATCH_UPDATE) && !b)) return fail(-5, "tile_batch: NULL pointer list");
  if (op == CHOL_BATCH_GEMM)
    for (int t = 0; t < n; ++t)
      if (!b[t]) return fail(-7, "tile_batch: NULL b operand of a GEMM task");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const int rc = dtype == CHOL_REAL_DOUBLE ? tile_batch_impl<double>(op, dtype, mb, n, c_in, a, b, c_out, a_versions, flags)
                                           : tile_batch_impl<float>(op, dtype, mb, n, c_in, a, b, c_out, a_versions, flags);
  if (rc) return rc;
  if (!(flags & CHOL_BATCH_ASYNC)) return tx_quiesce();
  return 0;
}

int chol_batch_mark(unsigned long long *mark2) {
  if (!mark2) return fail(-1, "batch_mark: NULL");
  {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int rc = tx_flush_pending()) return rc;
  }
  mark2[0] = tx.seq[TaskExec::CHAIN], mark2[1] = tx.seq[TaskExec::BULK];
  return 0;
}

int chol_batch_wait(const unsigned long long *mark2) {
  if (!mark2) return fail(-1, "batch_wait: NULL");
  if (!tx.ready) return 0;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  for (int s = 0; s < 2; ++s) {
    const unsigned long long want = std::min(mark2[s], tx.seq[s]);
    if (want == 0 || tx.done[s] >= want) continue;
    // (an event of the pool that has been recorded again since stands for a LATER batch of the same stream: waiting
    // for it waits a little longer, never too little)
    if (int rc = tx_wait_host(s, want)) return rc;
  }
  return 0;
}

int chol_batch_stats(long long *out4) {
  if (!out4) return fail(-1, "batch_stats: NULL");
  out4[0] = tx.batches[TaskExec::CHAIN], out4[1] = tx.batches[TaskExec::BULK], out4[2] = tx.cross_waits, out4[3] = (long long)tx.prod.size();
  if (getenv("CHOLMI_VERBOSE")) fprintf(stderr, "[cholmi] task executor: %lld panels issued pipelined (POTRF + its TRSM batch)\n", tx.fused_panels);
  return 0;
}

// ---- the executor's ordering, checked without a GPU (TxRecord above)
int chol_debug_task_record(int on, int mutate) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (on) {
    if (g.inited) return fail(CHOL_ERR_NOT_SUPPORTED, "task recorder: only before chol_init (it replaces the executor's streams)");
    tx_destroy();
    tx_rec = TxRecord();
    tx_rec.on = true;
    tx_rec.mutate = mutate;
    tx_wc = TxWinv();
    return 0;
  }
  if (tx_rec.on) {
    tx_destroy();
    tx_rec.on = false;
  }
  return 0;
}

// out5 = {batches recorded, event waits recorded, conflicting pairs examined, pairs not ordered, host-ordered pairs};
// the first unordered pair is described in chol_last_error()
int chol_debug_task_check(long long *out5) {
  if (!out5) return fail(-1, "task_check: NULL");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const auto &ops = tx_rec.ops;
  const size_t n = ops.size();
  // what of each stream precedes an op: its own stream up to seq - 1, every (stream, seq) it waited for, and what those
  // preceded in turn (clock[i][s] = newest batch of stream s known to be complete before op i starts)
  std::vector<std::array<unsigned long long, 2>> clock(n);
  std::unordered_map<unsigned long long, size_t> index[2];
  long long nwaits = 0, pairs = 0, bad = 0, hosted = 0;
  char first[256] = "";
  std::unordered_map<const void *, size_t> writer;
  std::unordered_map<const void *, std::vector<size_t>> readers;
  for (size_t i = 0; i < n; ++i) {
    const auto &op = ops[i];
    std::array<unsigned long long, 2> c = {op.base[0], op.base[1]};
    auto absorb = [&](int s, unsigned long long q) {
      if (q == 0) return;
      c[s] = std::max(c[s], q);
      auto it = index[s].find(q);
      if (it != index[s].end())
        for (int z = 0; z < 2; ++z) c[z] = std::max(c[z], clock[it->second][z]);
    };
    absorb(op.stream, op.seq - 1);
    for (const auto &w : op.waits) absorb(w.first, w.second), ++nwaits;
    clock[i] = c;
    index[op.stream][op.seq] = i;
    auto ordered = [&](size_t j) {  // op j (issued earlier) completes before op i starts
      const auto &o = ops[j];
      if (o.stream == op.stream) return true;
      if (op.base[o.stream] >= o.seq) {
        ++hosted;
        return true;
      }
      return c[o.stream] >= o.seq;
    };
    auto examine = [&](size_t j, const void *p, const char *kind) {
      if (j == i) return;
      ++pairs;
      if (!ordered(j)) {
        if (!bad)
          snprintf(first, sizeof first, "task executor: batch %llu of stream %d %s tile %p of batch %llu of stream %d without being ordered behind it",
                   op.seq, op.stream, kind, p, ops[j].seq, ops[j].stream);
        ++bad;
      }
    };
    for (const void *p : op.rd) {
      auto it = writer.find(p);
      if (it != writer.end()) examine(it->second, p, "reads");
    }
    for (const void *p : op.wr) {
      auto it = writer.find(p);
      if (it != writer.end()) examine(it->second, p, "overwrites");
      auto rt = readers.find(p);
      if (rt != readers.end())
        for (size_t j : rt->second) examine(j, p, "overwrites what is read by");
    }
    for (const void *p : op.rd) readers[p].push_back(i);
    for (const void *p : op.wr) writer[p] = i, readers.erase(p);
  }
  out5[0] = (long long)n, out5[1] = nwaits, out5[2] = pairs, out5[3] = bad, out5[4] = hosted;
  if (bad) set_error(first);
  return 0;
}

// ---------------------------------------------------------------- POTRF
int chol_potrf_tile(int uplo, chol_desc_t *A) {
  return with_views({{A, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "potrf_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-1, "potrf_tile: uplo");
  if (!A) return fail(-2, "potrf_tile: NULL descriptor");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
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
      launch_transpose_inplace<double>(g.r.st[ST_MAIN], (double *)A->mat, A->nt, A->mbi);
    else
      launch_transpose_inplace<float>(g.r.st[ST_MAIN], (float *)A->mat, A->nt, A->mbi);
  };
  flip();
  const int rc = A->dtype == CHOL_REAL_DOUBLE ? potrf_impl<double>(A) : potrf_impl<float>(A);
  flip();
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
  return rc;
  });
}

int chol_trsm_tile(int side, int uplo, int trans, int diag, double alpha, chol_desc_t *A,
                   chol_desc_t *B) {
  return with_views({{A, false}, {B, true}}, [&]() -> int {
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
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? trsm_impl<double>(alpha, A, B) : trsm_impl<float>(alpha, A, B);
  });
}

// ---------------------------------------------------------------- SYRK / GEMM
int chol_syrk_tile(int uplo, int trans, double alpha, chol_desc_t *A, double beta, chol_desc_t *C) {
  return with_views({{A, false}, {C, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "syrk_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-1, "syrk_tile: uplo");
  if (trans != CHOL_NOTRANS && trans != CHOL_TRANS) return fail(-2, "syrk_tile: trans");
  if (!A) return fail(-4, "syrk_tile: A is NULL");
  if (!C) return fail(-6, "syrk_tile: C is NULL");
  if (uplo != CHOL_LOWER || trans != CHOL_NOTRANS)
    return fail(CHOL_ERR_NOT_SUPPORTED, "syrk_tile: only (Lower, NoTrans)");
  if (!single_tile_square(A) || !single_tile_square(C) || A->mb != C->mb || A->dtype != C->dtype)
    return fail(CHOL_ERR_NOT_SUPPORTED, "syrk_tile: needs two 1-tile descriptors of equal size and type");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? gemm_impl<double>(alpha, A, A, beta, C, true)
                                      : gemm_impl<float>(alpha, A, A, beta, C, true);
  });
}

int chol_gemm_tile(int transA, int transB, double alpha, chol_desc_t *A, chol_desc_t *B,
                   double beta, chol_desc_t *C) {
  return with_views({{A, false}, {B, false}, {C, true}}, [&]() -> int {
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
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  return A->dtype == CHOL_REAL_DOUBLE ? gemm_impl<double>(alpha, A, B, beta, C, false)
                                      : gemm_impl<float>(alpha, A, B, beta, C, false);
  });
}

// ---------------------------------------------------------------- generator / layout / residual
int chol_plgsy_tile(double bump, int uplo, chol_desc_t *A, unsigned long long seed) {
  return with_views({{A, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "plgsy_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER && uplo != CHOL_UPPER_LOWER) return fail(-2, "plgsy_tile: uplo");
  const int side = uplo == CHOL_LOWER ? 1 : uplo == CHOL_UPPER ? 2 : 0;
  if (!A) return fail(-3, "plgsy_tile: NULL descriptor");
  if (!A->on_device) return fail(CHOL_ERR_NOT_SUPPORTED, "plgsy_tile: descriptor must be device-resident");
  if (A->mt == 1 && A->nt == 1 && (A->m != A->mb || A->n != A->nb) && !A->padded)
    return fail(CHOL_ERR_NOT_SUPPORTED, "plgsy_tile: partial single tile");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const LocalMat L = local_mat(A, A->mat);
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_plgsy<double>(g.r.st[ST_MAIN], L, A->lnt, A->prow, A->pcol, bump, seed, A->mb, (long)A->lm, side);
  else
    launch_plgsy<float>(g.r.st[ST_MAIN], L, A->lnt, A->prow, A->pcol, bump, seed, A->mb, (long)A->lm, side);
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
  return 0;
  });
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
  return with_views({{A, false}, {B, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "lacpy_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER && uplo != CHOL_UPPER_LOWER) return fail(-1, "lacpy_tile: uplo");
  int rc = resident_whole("lacpy_tile", A);
  if (rc) return rc;
  rc = resident_whole("lacpy_tile", B);
  if (rc) return rc;
  if (!same_geometry(A, B)) return fail(-3, "lacpy_tile: descriptors differ in shape, tiling or type");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const int side = uplo == CHOL_LOWER ? 1 : uplo == CHOL_UPPER ? 2 : 0;
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_lacpy<double>(g.r.st[ST_MAIN], geo_of(A), side, (const double *)A->mat, (double *)B->mat);
  else
    launch_lacpy<float>(g.r.st[ST_MAIN], geo_of(A), side, (const float *)A->mat, (float *)B->mat);
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
  return 0;
  });
}

int chol_geadd_tile(int trans, double alpha, chol_desc_t *A, double beta, chol_desc_t *B) {
  return with_views({{A, false}, {B, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "geadd_tile before chol_init");
  if (trans != CHOL_NOTRANS) return fail(CHOL_ERR_NOT_SUPPORTED, "geadd_tile: only ChamNoTrans (V6:83)");
  int rc = resident_whole("geadd_tile", A);
  if (rc) return rc;
  rc = resident_whole("geadd_tile", B);
  if (rc) return rc;
  if (!same_geometry(A, B)) return fail(-5, "geadd_tile: descriptors differ in shape, tiling or type");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_geadd<double>(g.r.st[ST_MAIN], geo_of(A), alpha, (const double *)A->mat, beta, (double *)B->mat);
  else
    launch_geadd<float>(g.r.st[ST_MAIN], geo_of(A), alpha, (const float *)A->mat, beta, (float *)B->mat);
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
  return 0;
  });
}

int chol_lange_tile(int norm, chol_desc_t *A, double *value) {
  return with_views({{A, false}}, [&]() -> int {
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
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const TileGeo ge = geo_of(A);
  int rcw = ensure_work((size_t)(std::max(ge.m, ge.n) + 2) * sizeof(double));
  if (rcw) return rcw;
  double *work = reinterpret_cast<double *>(g.work);
  if (A->dtype == CHOL_REAL_DOUBLE)
    launch_lange<double>(g.r.st[ST_MAIN], ge, kind, (const double *)A->mat, work);
  else
    launch_lange<float>(g.r.st[ST_MAIN], ge, kind, (const float *)A->mat, work);
  double v = 0;
  hipError_t e = hipMemcpyAsync(&v, work, sizeof(double), hipMemcpyDeviceToHost, g.r.st[ST_MAIN]);
  if (e == hipSuccess) e = hipStreamSynchronize(g.r.st[ST_MAIN]);
  if (e != hipSuccess) return fail_hip(e, "lange_tile", __LINE__);
  *value = kind == 3 ? std::sqrt(v) : v;
  return 0;
  });
}

int chol_lauum_tile(int uplo, chol_desc_t *A) {
  return with_views({{A, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "lauum_tile before chol_init");
  if (uplo != CHOL_LOWER) return fail(CHOL_ERR_NOT_SUPPORTED, "lauum_tile: only ChamLower (V6:80)");
  int rc = resident_whole("lauum_tile", A);
  if (rc) return rc;
  if (A->mt != A->nt || A->lm != A->ln) return fail(-2, "lauum_tile: matrix is not square");
  if (A->mbi % 64) return fail(CHOL_ERR_NOT_SUPPORTED, "lauum_tile: stored tile edge must be a multiple of 64");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    launch_lauum_lower<double>(g.r.st[ST_MAIN], (const double *)A->mat, (double *)tmp, A->nt, A->mbi);
    launch_lacpy<double>(g.r.st[ST_MAIN], ge, 1, (const double *)tmp, (double *)A->mat);
  } else {
    launch_lauum_lower<float>(g.r.st[ST_MAIN], (const float *)A->mat, (float *)tmp, A->nt, A->mbi);
    launch_lacpy<float>(g.r.st[ST_MAIN], ge, 1, (const float *)tmp, (float *)A->mat);
  }
  hipError_t e = hipStreamSynchronize(g.r.st[ST_MAIN]);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail_hip(e, "lauum_tile", __LINE__);
  return 0;
  });
}

// ---------------------------------------------------------------- solve with the factor
int chol_potrs_tile(int uplo, chol_desc_t *A, chol_desc_t *B) {
  return with_views({{A, false}, {B, true}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "potrs_tile before chol_init");
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-1, "potrs_tile: uplo");
  int rc = resident_whole("potrs_tile", A);
  if (rc) return rc;
  rc = resident_whole("potrs_tile", B);
  if (rc) return rc;
  if (A->mt != A->nt || A->lm != A->ln) return fail(-2, "potrs_tile: A is not square");
  if (B->lm != A->lm || B->mb != A->mb || B->mbi != A->mbi || B->dtype != A->dtype)
    return fail(-3, "potrs_tile: B must have A's order, tile size and type");
  if (A->mbi % 64) return fail(CHOL_ERR_NOT_SUPPORTED, "potrs_tile: stored tile edge must be a multiple of 64");
  CHECK_WINV(A, "potrs_tile");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  // ChamUpper: A = U^T U with U = L^T -- transpose the factor's storage in place around the Lower solve
  // (as chol_potrf_tile does around the Lower factorisation); the strict lower triangle comes back as it was
  auto flip = [&]() {
    if (A->dtype == CHOL_REAL_DOUBLE)
      launch_transpose_inplace<double>(g.r.st[ST_MAIN], (double *)A->mat, A->nt, A->mbi);
    else
      launch_transpose_inplace<float>(g.r.st[ST_MAIN], (float *)A->mat, A->nt, A->mbi);
  };
  if (uplo == CHOL_UPPER) flip();
  rc = A->dtype == CHOL_REAL_DOUBLE ? potrs_impl<double>(A, B) : potrs_impl<float>(A, B);
  if (uplo == CHOL_UPPER) {
    flip();
    HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
  }
  return rc;
  });
}

int chol_posv_tile(int uplo, chol_desc_t *A, chol_desc_t *B) {
  if (uplo != CHOL_LOWER && uplo != CHOL_UPPER) return fail(-1, "posv_tile: uplo");
  const int info = chol_potrf_tile(uplo, A);
  if (info != 0) return info;  // > 0: not positive definite, B untouched (LAPACK dposv)
  return chol_potrs_tile(uplo, A, B);
}

// valid extent of tile (I,J) inside the matrix (edge tiles are smaller)
static inline int tile_rows(const chol_desc *d, int I) { return std::min(d->mb, d->lm - I * d->mb); }
static inline int tile_cols(const chol_desc *d, int J) { return std::min(d->nb, d->ln - J * d->nb); }

int chol_tile_upload(chol_desc_t *d, int I, int J, const void *host_tile) {
  return with_views({{d, true}}, [&]() -> int {
  if (!d || !host_tile) return fail(-1, "tile_upload: NULL");
  if (I < 0 || I >= d->mt || J < 0 || J >= d->nt || I % d->p != d->prow || J % d->q != d->pcol)
    return fail(-2, "tile_upload: tile not owned by this process");
  char *dst = reinterpret_cast<char *>(d->mat) +
              ((size_t)(I / d->p) + (size_t)(J / d->q) * d->lmt) * (size_t)d->bsizi * d->esize;
  // host tile: mb x nb, ld = mb (Chameleon tile); only the part inside the matrix is stored
  HIPCHECK(hipMemcpy2D(dst, (size_t)d->mbi * d->esize, host_tile, (size_t)d->mb * d->esize,
                       (size_t)tile_rows(d, I) * d->esize, tile_cols(d, J), hipMemcpyDefault));
  return 0;
  });
}

int chol_tile_download(chol_desc_t *d, int I, int J, void *host_tile) {
  return with_views({{d, false}}, [&]() -> int {
  if (!d || !host_tile) return fail(-1, "tile_download: NULL");
  if (I < 0 || I >= d->mt || J < 0 || J >= d->nt || I % d->p != d->prow || J % d->q != d->pcol)
    return fail(-2, "tile_download: tile not owned by this process");
  const char *src = reinterpret_cast<const char *>(d->mat) +
                    ((size_t)(I / d->p) + (size_t)(J / d->q) * d->lmt) * (size_t)d->bsizi * d->esize;
  if (tile_rows(d, I) < d->mb || tile_cols(d, J) < d->nb) memset(host_tile, 0, (size_t)d->bsiz * d->esize);
  HIPCHECK(hipMemcpy2D(host_tile, (size_t)d->mb * d->esize, src, (size_t)d->mbi * d->esize,
                       (size_t)tile_rows(d, I) * d->esize, tile_cols(d, J), hipMemcpyDefault));
  return 0;
  });
}

int chol_lapack_to_tile(const void *A, int lda, chol_desc_t *d) {
  return with_views({{d, true}}, [&]() -> int {
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
  });
}

int chol_tile_to_lapack(chol_desc_t *d, void *A, int lda) {
  return with_views({{d, false}}, [&]() -> int {
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
  });
}

static int residual_common(chol_desc_t *L, double bump, unsigned long long seed, double *rel_fro,
                           double *rel_inf) {
  return with_views({{L, false}}, [&]() -> int {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "residual before chol_init");
  if (!L || (!rel_fro && !rel_inf)) return fail(-1, "residual: NULL");
  if (L->p * L->q != 1 || !L->on_device || L->mt != L->nt || L->mb != L->nb || L->mbi % MACRO)
    return fail(CHOL_ERR_NOT_SUPPORTED, "residual: single-process device-resident square tiled matrix only");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  const long n = L->lm;
  double *rows = nullptr;
  if (rel_inf) {
    HIPCHECK(hipMalloc(&rows, 2 * (size_t)n * sizeof(double)));
    HIPCHECK(hipMemsetAsync(rows, 0, 2 * (size_t)n * sizeof(double), g.r.st[ST_MAIN]));
  }
  HIPCHECK(hipMemsetAsync(g.d_acc, 0, 2 * sizeof(double), g.r.st[ST_MAIN]));
  if (L->dtype == CHOL_REAL_DOUBLE)
    launch_residual<double>(g.r.st[ST_MAIN], reinterpret_cast<const double *>(L->mat), L->nt, L->mbi, bump, seed,
                            g.d_acc, L->mb, n, rows);
  else
    launch_residual<float>(g.r.st[ST_MAIN], reinterpret_cast<const float *>(L->mat), L->nt, L->mbi, bump, seed,
                           g.d_acc, L->mb, n, rows);
  double h[2];
  HIPCHECK(hipMemcpyAsync(h, g.d_acc, sizeof h, hipMemcpyDeviceToHost, g.r.st[ST_MAIN]));
  HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
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
  });
}

int chol_residual_plgsy(chol_desc_t *L, double bump, unsigned long long seed, double *rel) {
  return residual_common(L, bump, seed, rel, nullptr);
}

int chol_residual_plgsy_inf(chol_desc_t *L, double bump, unsigned long long seed, double *rel_inf) {
  return residual_common(L, bump, seed, nullptr, rel_inf);
}

// ---------------------------------------------------------------- instrumentation
int chol_last_potrf_stats(double *total_ms, double *update_ms, int *update_launches, double *update_flops) {
  if (total_ms) *total_ms = g.r.total_ms;
  if (update_ms) *update_ms = g.r.update_ms;
  if (update_launches) *update_launches = g.r.update_launches;
  if (update_flops) *update_flops = g.r.update_flops;
  return 0;
}

int chol_set_profiling(int on) {
  g.r.profiling = on != 0;
  return 0;
}

int chol_bench_update(chol_desc_t *d, int k, int ablate, int reps, double *ms, double *flops) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "bench_update before chol_init");
  if (!d || !ms || d->p * d->q != 1 || !d->on_device || d->mt != d->nt || k < 0 || k + 1 >= d->nt || reps < 1)
    return fail(-1, "bench_update: arguments");
  if (ablate != 0)  // (the ablation twin of the update left the library in round 5; git eb8498c has it)
    return fail(CHOL_ERR_NOT_SUPPORTED, "bench_update: ablate must be 0");
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  int rc = ensure_events(2);
  if (rc) return rc;
  PanelRef pan;
  memset(&pan, 0, sizeof pan);
  pan.P = 1;
  pan.base[0] = (char *)d->mat + (size_t)k * d->nt * d->bsizi * d->esize;
  const LocalMat C = local_mat(d, d->mat);
  const ColRange rr = col_range(d, k + 1, d->nt);
  float best = 1e30f;
  for (int r = 0; r <= reps; ++r) {
    HIPCHECK(hipEventRecord(g.r.events[0], g.r.st[ST_MAIN]));
    if (d->dtype == CHOL_REAL_DOUBLE)
      launch_trail_update<double>(g.r.st[ST_MAIN], C, d->d_list, rr.off, rr.na, rr.offb, rr.nb, pan);
    else
      launch_trail_update<float>(g.r.st[ST_MAIN], C, d->d_list, rr.off, rr.na, rr.offb, rr.nb, pan);
    HIPCHECK(hipEventRecord(g.r.events[1], g.r.st[ST_MAIN]));
    HIPCHECK(hipStreamSynchronize(g.r.st[ST_MAIN]));
    float t = 0;
    HIPCHECK(hipEventElapsedTime(&t, g.r.events[0], g.r.events[1]));
    if (r > 0 && t < best) best = t;
  }
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
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  return mfma_probe_on(&g.r, dtype, waves_per_simd, 4000, tflops);
}

// 1: the counter-linked form of chain-bound waves is available (chol_init's probe found the panel streams on
// independent hardware queues and no device-side wait has timed out since); 0: every dependency is a stream
// event -- the case under a profiler that serialises kernels (rocprofv3 --pmc), where the probe fails by design
int chol_debug_device_counters(void) { return g.inited && g.r.d_sem ? 1 : 0; }

int chol_debug_update_kernel(int dtype, char *buf, int buflen) {
  if (!buf || buflen < 8) return fail(-2, "debug_update_kernel: buffer");
  const bool f64 = dtype == CHOL_REAL_DOUBLE;
  if (!f64 && dtype != CHOL_REAL_FLOAT) return fail(-1, "debug_update_kernel: dtype");
  // mirrors the dispatch of launch_trail_update (kernels.hip)
  snprintf(buf, buflen, f64 ? "cholmi::k_trail_update_w8<double, 3>" : "cholmi::k_trail_update_w8f");
  return 0;
}

// What the walker's regime switches are measured in, as taken at chol_init (or from CHOLMI_CALIB):
// out[0..3] = fp64 MFMA probe [TFLOP/s], fp64 diagonal-block step [us], fp32 probe, fp32 step;
// out[4..7] = the derived per-dtype update rate [TFLOP/s] and panel step [us] the walker uses (walker.h: WaveCalib)
int chol_debug_flow_waves(void) { return g.r.flow_waves; }

int chol_last_potrf_regimes(int *out8, int *nt) {
  if (!out8) return fail(-1, "last_potrf_regimes: NULL");
  for (int i = 0; i < 8; ++i) out8[i] = g.r.regimes[i];
  if (nt) *nt = g.r.regimes_nt;
  return 0;
}

int chol_debug_calibration(double *out8) {
  if (!g.inited) return fail(CHOL_ERR_NOT_INITIALIZED, "debug_calibration before chol_init");
  if (!out8) return fail(-1, "debug_calibration: NULL");
  for (int i = 0; i < 2; ++i) {
    out8[2 * i] = g.r.probe_tflops[i];
    out8[2 * i + 1] = g.r.diag_us[i];
    out8[4 + 2 * i] = g.r.probe_tflops[i] * CHOLMI_UPDATE_EFF;
    out8[5 + 2 * i] = g.r.diag_us[i] * CHOLMI_STEP_FACTOR;
  }
  return 0;
}

}  // extern "C"
