// chol_potrf_tile on a whole tiled matrix: the wave walker (walker.h) instantiated for
//   * this process's GPU (HipOps): one GPU (p = q = 1: v6_test.c:44-56) or its share of a p x q 2D block-cyclic
//     matrix, one process per GPU (v6_test.c:26-27 passes p, q into CHAMELEON_Desc_Create; tile (I,J) lives on
//     rank (I mod p) q + (J mod q), owner computes);
//   * caller-supplied tile kernels on the CPU (CbOps, chol_dist_factorize_with): the test hook that lets the
//     CPU suite drive the same schedule -- ownership, addressing, matching of sends and receives, buffer reuse --
//     with the oracle's kernels under gloo.  Never used by chol_potrf_tile.
// and the transports the p x q walker moves tiles with (include/cholmi.h: chol_transport_t, two channels):
//   * RCCL (xGMI inside a node), loaded at run time: two communicators, one per communication stream;
//   * whatever table the application installs (the tests: torch.distributed / gloo);
//   * an in-process asynchronous one (chol_dist_rehearse): p*q ranks as threads of this process on ONE GPU,
//     every send / receive a stream-ordered device copy, no device synchronisation anywhere -- the real
//     kernels, streams, events and buffer reuse of the multi-GPU schedule on the one GPU a test box has.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>  // types and enum values only: the library itself is resolved with dlopen at run time
#define CHOLMI_RCCL_HEADER 1
#else
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef void *ncclComm_t;
typedef int ncclResult_t;
enum { ncclInt8 = 0, ncclInt64 = 4 };
enum { ncclMax = 2 };
#define NCCL_MAJOR 2
#define NCCL_VERSION_CODE 20700
#endif

#include "../../include/cholmi.h"
#include "cholmi_internal.h"
#include "walker.h"

using namespace cholmi;

extern "C" int chol_internal_desc_create(chol_desc_t **desc, void *mat, int dtype, int mb, int nb, int bsiz, int lm, int ln,
                                         int i, int j, int m, int n, int p, int q, int rank, int nranks);  // api.hip

namespace {

int hip_fail(hipError_t e, const char *what) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s failed (walker): %s", what, hipGetErrorString(e));
  return chol_internal_fail(CHOL_ERR_HIP, buf);
}
#define HIPRC(call)                                      \
  do {                                                   \
    hipError_t e_ = (call);                              \
    if (e_ != hipSuccess) return hip_fail(e_, #call);    \
  } while (0)

struct ColRange {
  int off, na, offb, nb;
};
inline ColRange col_range(const chol_desc *d, int jlo, int jhi) {
  jlo = std::min(jlo, d->nt), jhi = std::min(jhi, d->nt);
  ColRange r;
  r.off = d->ge[jhi];
  r.na = d->ge[jlo] - d->ge[jhi];
  r.offb = d->n_off + d->gd[jhi];
  r.nb = d->gd[jlo] - d->gd[jhi];
  return r;
}

// ---------------------------------------------------------------- product ops: HIP kernels and streams
template <typename T>
struct HipOps {
  RankCtx &r;
  chol_desc *d;
  char *base;
  const WaveGeo &g;
  bool reset_ytab;
  bool sem_ok = false;
  LocalMat C;
  HipOps(RankCtx &rc, chol_desc *desc, void *b, const WaveGeo &geo, bool ry) : r(rc), d(desc), base((char *)b), g(geo), reset_ytab(ry) {
    C.base = b;
    C.lmt = desc->lmt;
    C.P = desc->p;
    C.Q = desc->q;
    C.mb = desc->mbi;
    C.bsiz = desc->bsizi;
  }
  bool profiling() const { return r.profiling; }
  bool pipe_ok() const { return true; }
  bool counters() const { return sem_ok; }
  bool can_split_trsm() const { return true; }
  void *stream(int st) { return r.st[st]; }
  char *tile(int il, int jl) { return base + ((size_t)il + (size_t)jl * g.lmt) * g.tile_bytes; }
  void *winv(int par) { return (char *)r.winv + (size_t)par * r.winv_bytes; }
  void *alloc(size_t bytes) { return r.pool.get(bytes); }
  int *sem(int k, int which, int per_wave) { return r.d_sem + ((size_t)per_wave * k + which) * 32; }
  bool flow_ok() const { return r.flow_ok; }
  void *flow_event() {
    if (!r.ev_flow) (void)hipEventCreateWithFlags(&r.ev_flow, hipEventDisableTiming);
    return r.ev_flow;
  }
  int launched() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
  }
  int begin(int nevents, int nt, int sem_per_wave) {
    while ((int)r.events.size() < nevents) {
      hipEvent_t e;
      HIPRC(hipEventCreate(&e));
      r.events.push_back(e);
    }
    HIPRC(hipMemsetAsync(r.d_info, 0, sizeof(int), r.st[ST_MAIN]));
    if (g_ytab && reset_ytab) HIPRC(hipMemsetAsync(g_ytab, 0, YTAB_ENTRIES * sizeof(int), r.st[ST_MAIN]));
    // device-side edges of the panel chain (kernels.hip, sem_wait; cholmi_internal.h, SyrkPipe): 3 nbm + 1
    // counters per wave, the last one = workgroups of the last SYRK slice on tile (k+1,k+1)
    sem_ok = r.d_sem && (long)nt * sem_per_wave <= SEM_SLOTS;
    if (sem_ok) HIPRC(hipMemsetAsync(r.d_sem, 0, (size_t)nt * sem_per_wave * 32 * sizeof(int), r.st[ST_MAIN]));
    return 0;
  }
  int rec(int ev, int st) {
    HIPRC(hipEventRecord(r.events[ev], r.st[st]));
    return 0;
  }
  int wt(int st, int ev) {
    HIPRC(hipStreamWaitEvent(r.st[st], r.events[ev], 0));
    return 0;
  }
  int panel(int k, char *lkk, void *wv, char *tiles, int ntiles, int ev_steps, int ev_head, const SyrkPipe *sy,
            const int *wait_sem, int wait_target) {
    launch_panel_pipelined<T>(r.st[ST_PANEL], r.st[ST_TRSM], &r.events[ev_steps], (T *)lkk, g.mb, (T *)wv, r.d_info,
                              k * g.mb, (T *)tiles, (long)g.mb * g.mb, ntiles, ev_head >= 0 ? r.events[ev_head] : nullptr,
                              sy, wait_sem, wait_target);
    return launched();
  }
  int trsm(int, char *tiles, int ntiles, const char *lkk, const char *wv, int st) {
    launch_trsm_panel<T>(r.st[st], (T *)tiles, (long)g.mb * g.mb, ntiles, (const T *)lkk, (const T *)wv, g.mb, T(1));
    return launched();
  }
  int diag_syrk(int, int, char *Cjj, const char *A, int st) {
    launch_diag_syrk<T>(r.st[st], (T *)Cjj, (const T *)A, g.mb);
    return launched();
  }
  int update(int, int, int jlo, int jhi, int what, const PanelRef &p1, const PanelRef *p2, bool yield, int st) {
    const ColRange c = col_range(d, jlo, jhi);
    launch_trail_update<T>(r.st[st], C, d->d_list, c.off, (what & 1) ? c.na : 0, c.offb, (what & 2) ? c.nb : 0, p1, yield, p2);
    return launched();
  }
  // one GPU: column k+1 below its diagonal tile, by panel k, in the latency form (kernels.hip: launch_col_update_small)
  int update_col_small(int k, int st) {
    launch_col_update_small<T>(r.st[st], (T *)tile(k + 2, k + 1), (const T *)tile(k + 2, k), (const T *)tile(k + 1, k), g.mb,
                               g.nt - k - 2, (long)g.mb * g.mb, nullptr);
    return launched();
  }
  // the streams have been joined into ST_MAIN and `ev_stop` recorded there
  int finish(int ev_start, int ev_stop, const std::vector<std::pair<int, int>> &brackets, int *info) {
    HIPRC(hipStreamSynchronize(r.st[ST_MAIN]));
    float ms = 0;
    HIPRC(hipEventElapsedTime(&ms, r.events[ev_start], r.events[ev_stop]));
    r.total_ms = ms;
    r.update_ms = 0;
    if (r.profiling) {
      // union of the brackets (disjoint by construction except around the waves launched as halves)
      std::vector<std::pair<float, float>> iv;
      for (auto &b : brackets) {
        float t0 = 0, t1 = 0;
        if (hipEventElapsedTime(&t0, r.events[ev_start], r.events[b.first]) != hipSuccess ||
            hipEventElapsedTime(&t1, r.events[ev_start], r.events[b.second]) != hipSuccess) {
          (void)hipGetLastError();  // a bracket of a wave that launched nothing on this rank
          continue;
        }
        if (t1 > t0) iv.emplace_back(t0, t1);
      }
      std::sort(iv.begin(), iv.end());
      float hi = -1;
      for (auto &p : iv) {
        if (p.first > hi) r.update_ms += p.second - p.first;
        else if (p.second > hi) r.update_ms += p.second - hi;
        hi = std::max(hi, p.second);
      }
    }
    HIPRC(hipMemcpy(info, r.d_info, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
  }
};

// ---------------------------------------------------------------- test ops: callbacks, executed in issue order
struct CbOps {
  chol_test_engine_t cb;
  const WaveGeo &g;
  CbOps(const chol_test_engine_t &c, const WaveGeo &geo) : cb(c), g(geo) {}
  bool profiling() const { return false; }
  bool pipe_ok() const { return false; }
  bool counters() const { return false; }
  bool can_split_trsm() const { return false; }  // the engine's trsm callback solves all local tiles of a panel
  void *stream(int) { return nullptr; }
  char *tile(int il, int jl) { return (char *)cb.store + ((size_t)il + (size_t)jl * g.lmt) * g.tile_bytes; }
  void *winv(int) { return nullptr; }
  void *alloc(size_t bytes) { return cb.alloc(cb.ctx, bytes); }
  int *sem(int, int, int) { return nullptr; }
  bool flow_ok() const { return false; }
  void *flow_event() { return nullptr; }
  int update_col_small(int, int) { return 0; }
  int begin(int, int, int) { return 0; }
  int rec(int, int) { return 0; }
  int wt(int, int) { return 0; }
  int panel(int k, char *lkk, void *, char *, int ntiles, int, int, const SyrkPipe *, const int *, int) {
    WRC(cb.potrf(cb.ctx, k, lkk));
    return ntiles > 0 ? cb.trsm(cb.ctx, k, lkk) : 0;
  }
  int trsm(int k, char *, int, const char *lkk, const char *, int) { return cb.trsm(cb.ctx, k, lkk); }
  int diag_syrk(int k, int j, char *, const char *A, int) {
    std::vector<const void *> hb(g.P, A);
    std::vector<int> hf(g.P, 0);
    hf[j % g.P] = j / g.P;
    return cb.update_diag(cb.ctx, k, j, hb.data(), hf.data());
  }
  int update(int k1, int k2, int jlo, int jhi, int what, const PanelRef &p1, const PanelRef *p2, bool, int) {
    const PanelRef *ps[2] = {&p1, p2};
    const int ks[2] = {k1, k2};
    for (int t = 0; t < 2; ++t) {
      if (!ps[t] || ks[t] < 0) continue;
      if (what == 3) {
        WRC(cb.update(cb.ctx, ks[t], jlo, jhi, ps[t]->base, ps[t]->first, 0));
      } else if (what == 1) {
        if (jhi != jlo + 1) return chol_internal_fail(-1, "CbOps: off-diagonal-only update of more than one column");
        WRC(cb.update(cb.ctx, ks[t], jlo, jhi, ps[t]->base, ps[t]->first, 1));
      } else if (what == 2) {
        for (int j = jlo; j < std::min(jhi, g.nt); ++j)
          if (j % g.P == g.pr && j % g.Q == g.pc) WRC(cb.update_diag(cb.ctx, ks[t], j, ps[t]->base, ps[t]->first));
      }
    }
    return 0;
  }
  int finish(int, int, const std::vector<std::pair<int, int>> &, int *info) {
    *info = cb.info(cb.ctx);
    return 0;
  }
};

// ---------------------------------------------------------------- the installed transport (two channels)
chol_transport_t g_tr[2];
bool g_tr_set = false;

// ---------------------------------------------------------------- RCCL transport (loaded on demand)
struct Rccl {
  void *lib = nullptr;
  ncclComm_t comm[2] = {nullptr, nullptr};
  int which[2] = {0, 1};
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, decltype(ncclInt8), int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, decltype(ncclInt8), int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, decltype(ncclInt8), decltype(ncclMax), ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  long long *d_red = nullptr;
  int version = 0;
} g_rccl;

int rccl_fail(const char *what, int rc) {
  char buf[256];
  snprintf(buf, sizeof buf, "RCCL %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString((ncclResult_t)rc) : "?");
  return chol_internal_fail(CHOL_ERR_HIP, buf);
}
int rccl_load() {
  if (g_rccl.lib) return 0;
  // the copy the process already has (torch ships one), else the system's
  const char *names[] = {"librccl.so", "librccl.so.1"};
  for (int pass = 0; pass < 2 && !g_rccl.lib; ++pass)
    for (const char *n : names) {
      g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (g_rccl.lib) break;
    }
  if (!g_rccl.lib) g_rccl.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.lib) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "librccl.so not found");
#define SYM(field, name)                                                  \
  *(void **)(&g_rccl.field) = dlsym(g_rccl.lib, name);                    \
  if (!g_rccl.field) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "librccl.so lacks " name)
  SYM(GetVersion, "ncclGetVersion");
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  // the prototypes and enum values above are those of the header this file was compiled against: the
  // library found at run time must speak the same major version and know point-to-point operations (2.7+)
  int v = 0;
  const int rc = g_rccl.GetVersion(&v);
  if (rc) return rccl_fail("ncclGetVersion", rc);
  g_rccl.version = v;
  const int major = v >= 10000 ? v / 10000 : v / 1000;
  if (major != NCCL_MAJOR || v < 2700) {
    char buf[160];
    snprintf(buf, sizeof buf, "librccl.so reports version code %d; this build needs major version %d with ncclSend/ncclRecv (>= 2.7)", v, NCCL_MAJOR);
    return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, buf);
  }
  return 0;
}
int rccl_group_begin(void *) {
  const int rc = g_rccl.GroupStart();
  return rc ? rccl_fail("ncclGroupStart", rc) : 0;
}
int rccl_group_end(void *) {
  const int rc = g_rccl.GroupEnd();
  return rc ? rccl_fail("ncclGroupEnd", rc) : 0;
}
int rccl_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream) {
  const int rc = g_rccl.Send(buf, bytes, ncclInt8, peer, g_rccl.comm[*(int *)ctx], (hipStream_t)stream);
  return rc ? rccl_fail("ncclSend", rc) : 0;
}
int rccl_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream) {
  const int rc = g_rccl.Recv(buf, bytes, ncclInt8, peer, g_rccl.comm[*(int *)ctx], (hipStream_t)stream);
  return rc ? rccl_fail("ncclRecv", rc) : 0;
}
int rccl_allreduce_max(void *ctx, long long *value) {
  if (!g_rccl.d_red && hipMalloc(&g_rccl.d_red, sizeof(long long)) != hipSuccess)
    return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "RCCL reduction word");
  hipStream_t s = main_rank_ctx()->st[ST_MAIN];
  if (hipMemcpyAsync(g_rccl.d_red, value, sizeof(long long), hipMemcpyHostToDevice, s) != hipSuccess)
    return chol_internal_fail(CHOL_ERR_HIP, "RCCL reduction word upload");
  const int rc = g_rccl.AllReduce(g_rccl.d_red, g_rccl.d_red, 1, ncclInt64, ncclMax, g_rccl.comm[*(int *)ctx], s);
  if (rc) return rccl_fail("ncclAllReduce", rc);
  if (hipMemcpyAsync(value, g_rccl.d_red, sizeof(long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return chol_internal_fail(CHOL_ERR_HIP, "RCCL reduction word download");
  return 0;
}

// ---------------------------------------------------------------- in-process asynchronous transport
// p*q ranks as threads of one process on one GPU (chol_dist_rehearse).  send = a stream-ordered copy into a
// staging block + an event, posted to the (channel, source, destination) mailbox; recv = wait (on the HOST,
// for the post only) for the matching mail, make the stream wait for its event, copy out stream-ordered.
// Exactly the ordering guarantees of ncclSend / ncclRecv on a stream, and nothing more: no device
// synchronisation, so a buffer reused too early or an event edge missing in the walker shows up as wrong data.
struct Hub {
  struct Mail {
    void *staging;
    size_t bytes;
    hipEvent_t ev;
  };
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::tuple<int, int, int>, std::deque<Mail>> box;
  std::vector<void *> blocks;
  std::vector<hipEvent_t> events;
  char *arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  int nranks = 1, arrived = 0, generation = 0;
  long long red = 0;
  bool failed = false;
  void *take(size_t bytes) {  // (mu held)
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (arena && arena_used + need <= arena_bytes) {
      void *p = arena + arena_used;
      arena_used += need;
      return p;
    }
    void *p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) return nullptr;
    blocks.push_back(p);
    return p;
  }
  void release() {
    for (void *p : blocks) (void)hipFree(p);
    blocks.clear();
    for (auto e : events) (void)hipEventDestroy(e);
    events.clear();
    if (arena) (void)hipFree(arena);
    arena = nullptr;
  }
};
struct HubEnd {
  Hub *hub;
  int rank, ch;
};
int hub_group(void *) { return 0; }
int hub_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream) {
  HubEnd *e = (HubEnd *)ctx;
  Hub::Mail m;
  m.bytes = bytes;
  {
    std::lock_guard<std::mutex> lk(e->hub->mu);
    m.staging = e->hub->take(bytes);
  }
  if (!m.staging) return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "rehearsal transport: staging");
  HIPRC(hipEventCreateWithFlags(&m.ev, hipEventDisableTiming));
  HIPRC(hipMemcpyAsync(m.staging, buf, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIPRC(hipEventRecord(m.ev, (hipStream_t)stream));
  {
    std::lock_guard<std::mutex> lk(e->hub->mu);
    e->hub->events.push_back(m.ev);
    e->hub->box[std::make_tuple(e->ch, e->rank, peer)].push_back(m);
  }
  e->hub->cv.notify_all();
  return 0;
}
int hub_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream) {
  HubEnd *e = (HubEnd *)ctx;
  Hub::Mail m;
  {
    std::unique_lock<std::mutex> lk(e->hub->mu);
    auto &q = e->hub->box[std::make_tuple(e->ch, peer, e->rank)];
    if (!e->hub->cv.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty() || e->hub->failed; }) || q.empty()) {
      e->hub->failed = true;
      lk.unlock();
      e->hub->cv.notify_all();
      return chol_internal_fail(CHOL_ERR_HIP, "rehearsal transport: a receive was never matched by a send");
    }
    m = q.front();
    q.pop_front();
  }
  if (m.bytes != bytes) return chol_internal_fail(CHOL_ERR_HIP, "rehearsal transport: send / receive sizes differ");
  HIPRC(hipStreamWaitEvent((hipStream_t)stream, m.ev, 0));
  HIPRC(hipMemcpyAsync(buf, m.staging, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
int hub_allreduce_max(void *ctx, long long *value) {
  HubEnd *e = (HubEnd *)ctx;
  Hub *h = e->hub;
  std::unique_lock<std::mutex> lk(h->mu);
  const int gen = h->generation;
  if (h->arrived == 0) h->red = *value;
  h->red = std::max(h->red, *value);
  if (++h->arrived == h->nranks) {
    h->arrived = 0;
    ++h->generation;
    lk.unlock();
    h->cv.notify_all();
    lk.lock();
  } else if (!h->cv.wait_for(lk, std::chrono::seconds(300), [&] { return h->generation != gen || h->failed; }) || h->failed) {
    h->failed = true;
    return chol_internal_fail(CHOL_ERR_HIP, "rehearsal transport: reduction never completed");
  }
  *value = h->red;
  return 0;
}

// ---------------------------------------------------------------- null transport (timing one rank alone)
int null_group(void *) { return 0; }
int null_send(void *, const void *, size_t, int, void *) { return 0; }
int null_recv(void *, void *buf, size_t bytes, int, void *stream) {
  HIPRC(hipMemsetAsync(buf, 0, bytes, (hipStream_t)stream));  // zeros: every update subtracts nothing, every tile stays finite
  return 0;
}
int null_allreduce(void *, long long *) { return 0; }

// ---------------------------------------------------------------- running the walker
WaveCalib calib_for(const RankCtx *r, int dtype, const WaveGeo &g) {
  const int idx = dtype == CHOL_REAL_DOUBLE ? 0 : 1;
  WaveCalib c;
  const double b3 = (double)g.mb * g.mb * g.mb;
  const double rate = std::max(1.0, r->probe_tflops[idx]) * 1e12 * CHOLMI_UPDATE_EFF;
  c.t_tile = 2.0 * b3 / rate;
  c.t_panel = g.nbm * std::max(1.0, r->diag_us[idx]) * 1e-6 * CHOLMI_STEP_FACTOR * 1.5;
  return c;
}

template <typename T>
int walk_t(chol_desc *d, void *base, RankCtx *r, int rank, WaveComm *cm, bool reset_ytab) {
  WaveGeo g;
  g.init(d->nt, d->mbi, d->p, d->q, rank, d->esize);
  if (g.lmt != d->lmt || g.ge[0] != d->ge[0] || g.gd[0] != d->gd[0])
    return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "potrf_tile: descriptor does not belong to this rank");
  HipOps<T> ops(*r, d, base, g, reset_ytab);
  Walker<HipOps<T>> w(ops, g, cm, calib_for(r, d->dtype, g));
  long long info = 0;
  int rc = w.setup();
  if (!rc) rc = w.run(&info);
  if (rc) {
    // leave nothing half-open behind a failure: close the transport groups, let kernels that poll a counter go
    // (every counter raised past any target), drain the streams
    if (cm) (void)cm->end(0), (void)cm->end(1);
    if (ops.sem_ok) (void)hipMemsetD32Async((hipDeviceptr_t)r->d_sem, 0x3fffffff, (size_t)SEM_SLOTS * 32, r->st[ST_PX]);  // (ST_PX: no polling kernel ever runs on it)
    for (int s = 0; s < ST_COUNT; ++s) (void)hipStreamSynchronize(r->st[s]);
    (void)hipGetLastError();
  }
  r->pool.release_all();
  r->update_flops = w.upd_flops;
  r->update_launches = w.upd_launches;
  r->flow_waves = w.flow_waves;
  for (int i = 0; i < 8; ++i) r->regimes[i] = i < Walker<HipOps<T>>::R_COUNT ? w.regimes[i] : 0;
  r->regimes_nt = g.nt;
  r->issue_us = w.issue_us / (g.nt > 0 ? g.nt : 1);  // per wave
  r->sends = cm ? cm->nsend : 0, r->recvs = cm ? cm->nrecv : 0, r->bytes_sent = cm ? cm->bytes_sent : 0;
  if (rc) return rc;
  if (info >= 0x7ffffffe) {
    // a bounded device-side wait gave up (kernels.hip: sem_wait, k_potrf_diag): the factor is invalid.  Not a
    // LAPACK info: a runtime failure of its own; the counter scheme stays off for the rest of this context, so
    // a caller that regenerates and retries runs on stream events alone
    if (r->d_sem) {
      (void)hipFree(r->d_sem);
      r->d_sem = nullptr;
    }
    return chol_internal_fail(CHOL_ERR_DEVICE_WAIT,
                              "potrf_tile: a device-side wait on the panel chain timed out (GPU shared with a heavy co-tenant?); "
                              "the matrix is left partly factored; device-side counters are now disabled for this context");
  }
  if (info > 0 && d->mbi != d->mb)  // stored index -> index in the caller's matrix
    info = ((info - 1) / d->mbi) * d->mb + (info - 1) % d->mbi + 1;
  return (int)info;
}

int walk_with(chol_desc *d, void *base, RankCtx *r, int rank, WaveComm *cm, bool reset_ytab) {
  if (d->mt != d->nt || d->lm != d->ln) return chol_internal_fail(-2, "potrf_tile: matrix is not square");
  if ((int)d->ge.size() < d->nt + 2)  // (a one-tile matrix on a p x q grid: no work list was built for it)
    return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "potrf_tile: a distributed descriptor needs more than one tile");
  if (d->p * d->q > 1) {
    if (!d->on_device) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "distributed potrf: the local tiles must be device-resident");
    if (d->mbi % MACRO) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "distributed potrf: stored tile edge must be a multiple of 128");
    if (!cm) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "distributed potrf: no transport installed (chol_set_transport / chol_transport_rccl_init)");
  }
  return d->dtype == CHOL_REAL_DOUBLE ? walk_t<double>(d, base, r, rank, cm, reset_ytab)
                                      : walk_t<float>(d, base, r, rank, cm, reset_ytab);
}

}  // namespace

extern "C" {

// called by chol_potrf_tile (api.hip) for every whole tiled matrix
int chol_internal_walk(chol_desc *d, void *base, RankCtx *r, int rank) {
  WaveComm cm;
  WaveComm *pc = nullptr;
  if (d->p * d->q > 1 && g_tr_set) {
    cm.ch[0] = g_tr[0];
    cm.ch[1] = g_tr[1];
    pc = &cm;
  }
  return walk_with(d, base, r, rank, pc, true);
}

int chol_set_transport_channel(int channel, const chol_transport_t *t) {
  if (channel < 0 || channel > 1) return chol_internal_fail(-1, "chol_set_transport_channel: channel must be 0 or 1");
  if (!t || !t->group_begin || !t->send || !t->recv || !t->group_end || !t->allreduce_max)
    return chol_internal_fail(-2, "chol_set_transport_channel: every entry of the table is required");
  g_tr[channel] = *t;
  return 0;
}

int chol_set_transport(const chol_transport_t *t) {
  if (!t) {
    g_tr_set = false;
    return 0;
  }
  if (!t->group_begin || !t->send || !t->recv || !t->group_end || !t->allreduce_max)
    return chol_internal_fail(-1, "chol_set_transport: every entry of the table is required");
  g_tr[0] = g_tr[1] = *t;
  g_tr_set = true;
  return 0;
}

// Test hook: a transport that moves nothing -- sends vanish, receives deliver zeros (stream-ordered).  One process
// can then play ANY single rank of a p x q grid on a one-GPU box and time that rank's real schedule with
// communication taken as free (scripts/dist_issue_time.py; the projection in DESIGN.md section 5).
int chol_set_transport_null(void) {
  chol_transport_t t;
  t.ctx = nullptr;
  t.group_begin = null_group;
  t.send = null_send;
  t.recv = null_recv;
  t.group_end = null_group;
  t.allreduce_max = null_allreduce;
  return chol_set_transport(&t);
}

int chol_transport_rccl_unique_id(void *id256) {
  if (!id256) return chol_internal_fail(-1, "rccl_unique_id: NULL");
  WRC(rccl_load());
  for (int c = 0; c < 2; ++c) {
    ncclUniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy((char *)id256 + c * sizeof id, &id, sizeof id);
  }
  return 0;
}

int chol_transport_rccl_init(const void *id256, int rank, int nranks) {
  if (!id256) return chol_internal_fail(-1, "rccl_init: NULL id");
  if (nranks < 1 || rank < 0 || rank >= nranks) return chol_internal_fail(-2, "rccl_init: rank");
  WRC(rccl_load());
  if (g_rccl.comm[0]) return chol_internal_fail(-1, "rccl_init: communicators already exist");
  for (int c = 0; c < 2; ++c) {
    ncclUniqueId id;
    memcpy(&id, (const char *)id256 + c * sizeof id, sizeof id);
    const int rc = g_rccl.CommInitRank(&g_rccl.comm[c], nranks, id, rank);
    if (rc) {
      if (c == 1) (void)g_rccl.CommDestroy(g_rccl.comm[0]);
      g_rccl.comm[0] = g_rccl.comm[1] = nullptr;
      return rccl_fail("ncclCommInitRank", rc);
    }
  }
  for (int c = 0; c < 2; ++c) {
    chol_transport_t t;
    t.ctx = &g_rccl.which[c];
    t.group_begin = rccl_group_begin;
    t.send = rccl_send;
    t.recv = rccl_recv;
    t.group_end = rccl_group_end;
    t.allreduce_max = rccl_allreduce_max;
    g_tr[c] = t;
  }
  g_tr_set = true;
  return 0;
}

int chol_transport_rccl_version(void) {
  if (rccl_load()) return 0;
  return g_rccl.version;
}

int chol_transport_rccl_finalize(void) {
  if (g_rccl.comm[0]) {
    for (int c = 0; c < 2; ++c) {
      if (g_rccl.comm[c]) (void)g_rccl.CommDestroy(g_rccl.comm[c]);
      g_rccl.comm[c] = nullptr;
    }
    g_tr_set = false;
  }
  if (g_rccl.d_red) (void)hipFree(g_rccl.d_red);
  g_rccl.d_red = nullptr;
  return 0;
}

// Test hook: one message of `bytes` bytes from this rank to itself on EACH channel of the installed transport,
// the two groups issued back to back on the walker's two communication streams (ST_CX, ST_PX), byte-compared.
// On a one-rank RCCL communicator this is ncclSend / ncclRecv to self inside ncclGroupStart / ncclGroupEnd --
// the transport table's entries executing on the hardware a test box has.
int chol_transport_selftest(int self_rank, size_t bytes) {
  if (!g_tr_set) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "transport_selftest: no transport installed");
  if (bytes == 0 || bytes % 8) return chol_internal_fail(-2, "transport_selftest: bytes must be a positive multiple of 8");
  RankCtx *r = main_rank_ctx();
  if (!r->st[ST_CX]) return chol_internal_fail(CHOL_ERR_NOT_INITIALIZED, "transport_selftest before chol_init");
  std::vector<unsigned long long> h(bytes / 8), back(bytes / 8);
  char *dev = nullptr;
  HIPRC(hipMalloc(&dev, 4 * bytes));
  int rc = 0;
  const int sts[2] = {ST_CX, ST_PX};
  for (int c = 0; c < 2 && !rc; ++c) {
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x9e3779b97f4a7c15ull * (i + 1 + 7919 * c);
    if (hipMemcpy(dev + 2 * c * bytes, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(dev + (2 * c + 1) * bytes, 0, bytes) != hipSuccess)
      rc = chol_internal_fail(CHOL_ERR_HIP, "transport_selftest: upload");
  }
  for (int c = 0; c < 2 && !rc; ++c) {  // both groups in flight before either is waited for
    hipStream_t s = r->st[sts[c]];
    rc = g_tr[c].group_begin(g_tr[c].ctx);
    if (!rc) rc = g_tr[c].send(g_tr[c].ctx, dev + 2 * c * bytes, bytes, self_rank, s);
    if (!rc) rc = g_tr[c].recv(g_tr[c].ctx, dev + (2 * c + 1) * bytes, bytes, self_rank, s);
    const int rc2 = g_tr[c].group_end(g_tr[c].ctx);
    if (!rc) rc = rc2;
  }
  for (int c = 0; c < 2; ++c)
    if (hipStreamSynchronize(r->st[sts[c]]) != hipSuccess && !rc) rc = chol_internal_fail(CHOL_ERR_HIP, "transport_selftest: synchronize");
  for (int c = 0; c < 2 && !rc; ++c) {
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x9e3779b97f4a7c15ull * (i + 1 + 7919 * c);
    if (hipMemcpy(back.data(), dev + (2 * c + 1) * bytes, bytes, hipMemcpyDeviceToHost) != hipSuccess)
      rc = chol_internal_fail(CHOL_ERR_HIP, "transport_selftest: download");
    else if (memcmp(back.data(), h.data(), bytes) != 0)
      rc = chol_internal_fail(CHOL_ERR_HIP, c == 0 ? "transport_selftest: channel 0 delivered different bytes" : "transport_selftest: channel 1 delivered different bytes");
  }
  (void)hipFree(dev);
  return rc;
}

int chol_dist_last_stats(double *issue_us_per_wave, long long *sends, long long *recvs, long long *bytes_sent) {
  RankCtx *r = main_rank_ctx();
  if (issue_us_per_wave) *issue_us_per_wave = r->issue_us;
  if (sends) *sends = r->sends;
  if (recvs) *recvs = r->recvs;
  if (bytes_sent) *bytes_sent = r->bytes_sent;
  return 0;
}

// Collect the lower tiles of a p x q descriptor on `root` into a single-process descriptor of the same
// order and tile size (dst is ignored elsewhere): the verification step after a distributed run.
int chol_dist_gather_lower(chol_desc_t *src, chol_desc_t *dst, int root) {
  if (!src) return chol_internal_fail(-1, "dist_gather_lower: NULL source");
  const int P = src->p, Q = src->q, world = P * Q;
  const int rank = world > 1 ? src->prow * Q + src->pcol : 0;
  if (root < 0 || root >= world) return chol_internal_fail(-3, "dist_gather_lower: root");
  if (rank == root) {
    if (!dst) return chol_internal_fail(-2, "dist_gather_lower: NULL destination on the root");
    if (dst->p * dst->q != 1 || dst->lm != src->lm || dst->mbi != src->mbi || dst->dtype != src->dtype || !dst->on_device)
      return chol_internal_fail(-2, "dist_gather_lower: destination must be a device-resident 1 x 1 descriptor of the same order, tile and type");
  }
  if (world > 1 && !g_tr_set) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "dist_gather_lower: no transport installed");
  const size_t tb = (size_t)src->bsizi * src->esize;
  hipStream_t st = main_rank_ctx()->st[ST_MAIN];
  chol_transport_t &tr = g_tr[1];
  for (int J = 0; J < src->nt; ++J) {
    bool grouped = false;
    for (int I = J; I < src->mt; ++I) {
      const int owner = (I % P) * Q + J % Q;
      char *mine = (char *)src->mat + ((size_t)(I / P) + (size_t)(J / Q) * src->lmt) * tb;
      char *there = rank == root ? (char *)dst->mat + ((size_t)I + (size_t)J * dst->lmt) * tb : nullptr;
      if (owner == rank && rank == root) {
        if (hipMemcpyAsync(there, mine, tb, hipMemcpyDeviceToDevice, st) != hipSuccess)
          return chol_internal_fail(CHOL_ERR_HIP, "dist_gather_lower: copy");
      } else if (owner == rank || rank == root) {
        if (!grouped) {
          WRC(tr.group_begin(tr.ctx));
          grouped = true;
        }
        if (owner == rank)
          WRC(tr.send(tr.ctx, mine, tb, root, st));
        else
          WRC(tr.recv(tr.ctx, there, tb, owner, st));
      }
    }
    if (grouped) WRC(tr.group_end(tr.ctx));
  }
  return hipStreamSynchronize(st) == hipSuccess ? 0 : chol_internal_fail(CHOL_ERR_HIP, "dist_gather_lower: synchronize");
}

void chol_internal_dist_finalize(void) { (void)chol_transport_rccl_finalize(); }

// Test hook (include/cholmi.h): the walker over caller-supplied tile kernels.
// mode: 0 every wave plain and split (near / far halves), 1 mixed (pairs while a rank has >= 6 tiles to update),
// 2 every wave in pairs, none split -- the regimes the product picks by measured speed, forced here
int chol_dist_factorize_with(const chol_test_engine_t *engine, const chol_transport_t *transport, int N, int B,
                             int p, int q, int rank, int mode) {
  if (!engine || !engine->store || !engine->alloc || !engine->potrf || !engine->trsm || !engine->update ||
      !engine->update_diag || !engine->info)
    return chol_internal_fail(-1, "dist_factorize_with: engine table incomplete");
  if (p * q > 1 && !transport) return chol_internal_fail(-2, "dist_factorize_with: transport required");
  if (N <= 0 || B <= 0 || N % B || p <= 0 || q <= 0 || p > MAXP || rank < 0 || rank >= p * q)
    return chol_internal_fail(-3, "dist_factorize_with: geometry");
  WaveGeo g;
  g.init(N / B, B, p, q, rank, engine->esize);
  g.winv_bytes = 0;
  CbOps ops(*engine, g);
  WaveComm cm;
  if (transport) cm.ch[0] = cm.ch[1] = *transport;
  WaveCalib c;
  c.t_tile = 1.0;
  c.t_panel = mode == 0 ? 1e30 : mode == 1 ? 3.0 : 1e-30;
  Walker<CbOps> w(ops, g, p * q > 1 ? &cm : nullptr, c);
  long long info = 0;
  int rc = w.setup();
  if (!rc) rc = w.run(&info);
  if (rc && transport) (void)cm.end(0), (void)cm.end(1);
  RankCtx *r = main_rank_ctx();
  r->issue_us = w.issue_us / (g.nt > 0 ? g.nt : 1);  // per wave
  r->sends = cm.nsend, r->recvs = cm.nrecv, r->bytes_sent = cm.bytes_sent;
  if (rc) return rc;
  return (int)info;
}

// Test hook: a p x q factorisation rehearsed on ONE GPU -- p*q ranks as threads of this process, each with its
// own streams, workspaces and local tiles, the real kernels, the in-process asynchronous transport above.  The
// matrix is plgsy(bump, seed) of order N; the factor's lower tiles are gathered into `full` (a device-resident
// 1 x 1 descriptor of the same order, tile size and type).  *ms: wall time of the factorisation (all ranks share
// the GPU: a stress figure, not a scaling figure).  Returns the LAPACK info all ranks agreed on.
int chol_dist_rehearse(int dtype, int N, int mb, int p, int q, double bump, unsigned long long seed,
                       chol_desc_t *full, double *ms) {
  RankCtx *mr = main_rank_ctx();
  if (!mr->st[ST_MAIN]) return chol_internal_fail(CHOL_ERR_NOT_INITIALIZED, "dist_rehearse before chol_init");
  const int R = p * q;
  if (p < 1 || q < 1 || p > MAXP || R > 16) return chol_internal_fail(-4, "dist_rehearse: grid");
  if (N <= 0 || mb <= 0) return chol_internal_fail(-2, "dist_rehearse: N, mb");
  if (!full || full->p * full->q != 1 || full->lm != N || full->mb != mb || full->dtype != dtype || !full->on_device)
    return chol_internal_fail(-8, "dist_rehearse: `full` must be a device-resident 1 x 1 descriptor of order N, tile mb, same type");
  const int mbi = full->mbi;  // stored tile edge (mb rounded up to 128 when N is ragged or mb no multiple of 128)
  std::vector<RankCtx> ctx(R);
  std::vector<chol_desc_t *> desc(R, nullptr);
  std::vector<HubEnd> ends(2 * R);
  Hub hub;
  hub.nranks = R;
  int rc = 0;
  const int nt = (N + mb - 1) / mb;
  const size_t tb = (size_t)mbi * mbi * (dtype == CHOL_REAL_DOUBLE ? 8 : 4);
  // staging for everything that is ever sent: every panel tile to at most p + q - 2 peers, plus diagonal and head tiles
  hub.arena_bytes = (size_t)nt * (nt + 1) / 2 * tb * (size_t)std::max(1, p + q - 2) + (size_t)4 * nt * p * tb + (64u << 20);
  if (hipMalloc(&hub.arena, hub.arena_bytes) != hipSuccess) {
    (void)hipGetLastError();
    hub.arena = nullptr;  // falls back to one allocation per message
  }
  for (int r = 0; r < R && !rc; ++r) {
    rc = rank_ctx_create(&ctx[r], mr->device, mr);
    if (!rc) rc = chol_internal_desc_create(&desc[r], nullptr, dtype, mb, mb, mb * mb, N, N, 0, 0, N, N, p, q, r, R);
    if (!rc) rc = chol_plgsy_tile(bump, CHOL_LOWER, desc[r], seed);
    for (int c = 0; c < 2; ++c) ends[2 * r + c] = HubEnd{&hub, r, c};
  }
  std::vector<int> res(R, 0);
  const auto t0 = std::chrono::steady_clock::now();
  if (!rc) {
    if (g_ytab) (void)hipMemset(g_ytab, 0, YTAB_ENTRIES * sizeof(int));
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r)
      th.emplace_back([&, r] {
        (void)hipSetDevice(mr->device);
        WaveComm cm;
        for (int c = 0; c < 2; ++c) {
          chol_transport_t t;
          t.ctx = &ends[2 * r + c];
          t.group_begin = hub_group;
          t.send = hub_send;
          t.recv = hub_recv;
          t.group_end = hub_group;
          t.allreduce_max = hub_allreduce_max;
          cm.ch[c] = t;
        }
        res[r] = walk_with(desc[r], desc[r]->mat, &ctx[r], r, R > 1 ? &cm : nullptr, false);
        if (res[r] < 0) {
          std::lock_guard<std::mutex> lk(hub.mu);
          hub.failed = true;
          hub.cv.notify_all();
        }
      });
    for (auto &t : th) t.join();
  }
  if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  (void)hipDeviceSynchronize();
  if (!rc) {
    for (int r = 0; r < R; ++r)
      if (res[r] < 0 || (res[r] != res[0] && !rc)) rc = res[r] < 0 ? res[r] : chol_internal_fail(CHOL_ERR_HIP, "dist_rehearse: ranks disagree on info");
    if (!rc) rc = res[0];
  }
  if (rc >= 0) {  // gather the lower tiles
    for (int J = 0; J < nt; ++J)
      for (int I = J; I < nt; ++I) {
        chol_desc_t *s = desc[(I % p) * q + J % q];
        const char *src = (const char *)s->mat + ((size_t)(I / p) + (size_t)(J / q) * s->lmt) * tb;
        char *dst = (char *)full->mat + ((size_t)I + (size_t)J * full->lmt) * tb;
        if (hipMemcpyAsync(dst, src, tb, hipMemcpyDeviceToDevice, mr->st[ST_MAIN]) != hipSuccess)
          rc = chol_internal_fail(CHOL_ERR_HIP, "dist_rehearse: gather");
      }
    (void)hipStreamSynchronize(mr->st[ST_MAIN]);
  }
  for (int r = 0; r < R; ++r) {
    if (desc[r]) (void)chol_desc_destroy(&desc[r]);
    if (ctx[r].st[ST_MAIN]) {
      mr->sends = ctx[0].sends, mr->recvs = ctx[0].recvs, mr->bytes_sent = ctx[0].bytes_sent, mr->issue_us = ctx[0].issue_us;
      rank_ctx_destroy(&ctx[r]);
    }
  }
  hub.release();
  return rc;
}

}  // extern "C"
