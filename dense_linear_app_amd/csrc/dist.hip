// 2D block-cyclic tiled Cholesky over p x q processes (one per GPU) behind chol_potrf_tile:
// the wave loop of the reference client (client_distrib.cpp:506-565) with Chameleon's descriptor
// distribution (p, q: worker_distrib.cpp:77, v6_test.c:26-27, 44-45; always 1 x 1 in the reference).
// Tile (I,J) lives on rank (I mod p) q + (J mod q); owner computes.  What moves per wave k
// (SURVEY 8e), point to point, one transport group each:
//   1. L(k,k) and the inverses of its 128-blocks: owner -> the other ranks of process column k mod q;
//   2. the head tile L(k+1,k) alone, ahead of everything else, to the owner of (k+1,k+1): that rank
//      applies the one SYRK, factors the next diagonal tile and ships it while the rest of panel k
//      is still on the wire (POTRF and its send are off the per-wave critical path);
//   3. panel k: the part of process row r (tiles i = r mod p, contiguous in its owner's storage, no
//      packing) goes whole to the other ranks of that process row -- they need L(i,k) as the row
//      operand of their tiles (i, j) -- and tile by tile to the ranks (r', j mod q), r' != r, which
//      need L(j,k) as the column operand of their tiles (i, j).  Nobody receives a tile it does not
//      use: (1/p + 1/q) of the panel per rank instead of all of it.
// Three streams per rank, receive buffers double-buffered by wave parity, one wave of lookahead:
//   main   U1(k) = columns k+1, k+2 by panel k; U2(k) = the columns beyond
//   side   TRSM(k+1), head tile, exchange of panel k+1 -- while U2(k) runs
//   early  SYRK on (k+1,k+1) from the head tile, POTRF(k+1), its sends
// The loop is written against `Engine` (tile kernels + streams) and chol_transport_t (exchange):
// the product engine launches the HIP kernels through the library's own wave entry points; the test
// engine (chol_dist_factorize_with) lets the CPU suite drive the same loop with the oracle's tile
// kernels under gloo.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cholmi.h"
#include "cholmi_internal.h"

extern "C" int chol_internal_fail(int code, const char *msg);  // api.hip: sets chol_last_error

namespace {

enum { S_MAIN = 0, S_SIDE = 1, S_EARLY = 2 };
enum { EV_HEAD0 = 0, EV_HEAD1 = 1, EV_U1 = 2, EV_DIAG = 3, EV_TMP = 4, EV_COUNT = 5 };

struct Engine {
  int N = 0, B = 0, P = 1, Q = 1, rank = 0, nt = 0, prow = 0, pcol = 0, lmt = 0, lnt = 0;
  size_t esize = 8;
  size_t tile_bytes = 0;
  virtual ~Engine() {}
  virtual char *store() = 0;
  virtual void *alloc(size_t bytes) = 0;
  virtual int potrf(int k, void *lkk, int s) = 0;
  virtual size_t winv_bytes() = 0;
  virtual int export_winv(void *dst, int s) = 0;
  virtual int import_winv(const void *src, int s) = 0;
  virtual int trsm(int k, const void *lkk, int s) = 0;
  virtual int update(int k, int jlo, int jhi, const void *const *bases, const int *firsts, int skip_diag, int s) = 0;
  virtual int update_diag(int k, int j, const void *const *bases, const int *firsts, int s) = 0;
  virtual int copy(void *dst, const void *src, size_t bytes, int s) = 0;
  virtual void *stream(int s) = 0;
  virtual int record(int s, int ev) = 0;
  virtual int wait_event(int s, int ev) = 0;
  virtual int sync() = 0;
  virtual int reset_info() = 0;
  virtual int info(int *out) = 0;
  int wait_stream(int waiter, int waited) {
    int rc = record(waited, EV_TMP);
    return rc ? rc : wait_event(waiter, EV_TMP);
  }
  char *tile_ptr(int il, int jl) { return store() + ((size_t)il + (size_t)jl * lmt) * tile_bytes; }
  void geometry(int N_, int B_, int P_, int Q_, int rank_, size_t esize_) {
    N = N_, B = B_, P = P_, Q = Q_, rank = rank_, esize = esize_;
    nt = (N + B - 1) / B;
    prow = rank / Q, pcol = rank % Q;
    lmt = nt - prow > 0 ? (nt - prow + P - 1) / P : 0;
    lnt = nt - pcol > 0 ? (nt - pcol + Q - 1) / Q : 0;
    tile_bytes = (size_t)B * B * esize;
  }
};

#define RC(call)        \
  do {                  \
    int rc_ = (call);   \
    if (rc_) return rc_; \
  } while (0)

struct Dist {
  Engine &e;
  chol_transport_t tr;
  bool have_tr;
  char *lkk_buf[2] = {nullptr, nullptr}, *head_buf[2] = {nullptr, nullptr};
  std::vector<char *> pbuf[2];
  const void *lkk_ptr[2] = {nullptr, nullptr};
  size_t nw = 0;
  long long nsend = 0, nrecv = 0, bytes_sent = 0;
  bool in_group = false;

  Dist(Engine &eng, const chol_transport_t *t) : e(eng), have_tr(t != nullptr) {
    if (t) tr = *t;
  }
  int rank_of(int pr, int pc) const { return pr * e.Q + pc; }
  int fla(int k, int p2) const { return (k + e.P - p2) / e.P; }  // first local row with global index > k
  void part(int k, int p2, int *il0, int *cnt) const {
    *il0 = fla(k, p2);
    const int rows = e.nt - p2 > 0 ? (e.nt - p2 + e.P - 1) / e.P : 0;
    *cnt = rows - *il0 > 0 ? rows - *il0 : 0;
  }
  int begin() {
    if (!in_group) {
      in_group = true;
      return tr.group_begin(tr.ctx);
    }
    return 0;
  }
  int end() {
    if (in_group) {
      in_group = false;
      return tr.group_end(tr.ctx);
    }
    return 0;
  }
  int send(const void *buf, size_t bytes, int peer, int s) {
    RC(begin());
    ++nsend;
    bytes_sent += (long long)bytes;
    return tr.send(tr.ctx, buf, bytes, peer, e.stream(s));
  }
  int recv(void *buf, size_t bytes, int peer, int s) {
    RC(begin());
    ++nrecv;
    return tr.recv(tr.ctx, buf, bytes, peer, e.stream(s));
  }

  int setup() {
    const int world = e.P * e.Q;
    if (world > 1 && !have_tr) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "distributed potrf: no transport installed (chol_set_transport / chol_transport_rccl_init)");
    nw = e.winv_bytes();
    const int maxpart = (e.nt + e.P - 1) / e.P;
    for (int par = 0; par < 2; ++par) {
      lkk_buf[par] = (char *)e.alloc(e.tile_bytes + nw);
      head_buf[par] = (char *)e.alloc(e.tile_bytes);
      pbuf[par].assign(e.P, nullptr);
      for (int p2 = 0; p2 < e.P; ++p2) pbuf[par][p2] = (char *)e.alloc((size_t)(maxpart > 0 ? maxpart : 1) * e.tile_bytes);
      if (!lkk_buf[par] || !head_buf[par]) return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "distributed potrf: buffer allocation failed");
      for (char *p : pbuf[par])
        if (!p) return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "distributed potrf: buffer allocation failed");
    }
    return 0;
  }

  // L(k,k): POTRF on its owner, then (with its block inverses) to the other ranks of the process column
  int diag(int k, int s) {
    const int pr = k % e.P, pc = k % e.Q, par = k & 1;
    if (e.pcol != pc) return 0;
    const bool last = k + 1 >= e.nt;
    char *buf = lkk_buf[par];
    if (e.prow == pr) {
      char *lkk = e.tile_ptr(k / e.P, k / e.Q);
      RC(e.potrf(k, lkk, s));
      if (e.P > 1 && !last) {
        RC(e.copy(buf, lkk, e.tile_bytes, s));
        if (nw) RC(e.export_winv(buf + e.tile_bytes, s));
        for (int r2 = 0; r2 < e.P; ++r2)
          if (r2 != pr) RC(send(buf, e.tile_bytes + nw, rank_of(r2, pc), s));
        RC(end());
      }
      lkk_ptr[par] = lkk;
    } else {
      if (e.P > 1 && !last) {
        RC(recv(buf, e.tile_bytes + nw, rank_of(pr, pc), s));
        RC(end());
        if (nw) RC(e.import_winv(buf + e.tile_bytes, s));
      }
      lkk_ptr[par] = buf;
    }
    return 0;
  }

  // TRSM of the local tiles of panel k, then its head tile L(k+1,k) to the owner of (k+1,k+1)
  int trsm_and_head(int k, int s, bool send_head, const void **head) {
    const int pc = k % e.Q, par = k & 1;
    *head = nullptr;
    if (e.pcol == pc) RC(e.trsm(k, lkk_ptr[par], s));
    if (!send_head || k + 1 >= e.nt) return 0;
    const int h = rank_of((k + 1) % e.P, pc), d1 = rank_of((k + 1) % e.P, (k + 1) % e.Q);
    if (e.rank == h) {
      char *t = e.tile_ptr((k + 1) / e.P, k / e.Q);
      if (d1 != h) {
        RC(send(t, e.tile_bytes, d1, s));
        RC(end());
      } else {
        *head = t;
      }
    } else if (e.rank == d1) {
      RC(recv(head_buf[par], e.tile_bytes, h, s));
      RC(end());
      *head = head_buf[par];
    }
    return e.record(s, EV_HEAD0 + par);
  }

  // panel k to everybody who uses it; -> where tile i of the panel is on this rank (bases / firsts)
  int exchange(int k, int s, const void **bases, int *firsts) {
    const int pc = k % e.Q, par = k & 1;
    for (int p2 = 0; p2 < e.P; ++p2) {
      int il0, cnt;
      part(k, p2, &il0, &cnt);
      firsts[p2] = il0;
      if (e.prow == p2 && e.pcol == pc) {
        char *src = cnt > 0 ? e.tile_ptr(il0, k / e.Q) : pbuf[par][p2];
        bases[p2] = src;
        if (cnt <= 0) continue;
        for (int c2 = 0; c2 < e.Q; ++c2)  // along the process row: the whole part
          if (c2 != pc) RC(send(src, (size_t)cnt * e.tile_bytes, rank_of(p2, c2), s));
        for (int r2 = 0; r2 < e.P; ++r2) {  // to the other process rows: tile j to process column j mod q
          if (r2 == p2) continue;
          for (int t = 0; t < cnt; ++t) {
            const int j = (il0 + t) * e.P + p2;
            RC(send(src + (size_t)t * e.tile_bytes, e.tile_bytes, rank_of(r2, j % e.Q), s));
          }
        }
      } else {
        char *buf = pbuf[par][p2];
        bases[p2] = buf;
        if (cnt <= 0) continue;
        if (e.prow == p2) {
          RC(recv(buf, (size_t)cnt * e.tile_bytes, rank_of(p2, pc), s));
        } else {
          for (int t = 0; t < cnt; ++t) {
            const int j = (il0 + t) * e.P + p2;
            if (j % e.Q == e.pcol) RC(recv(buf + (size_t)t * e.tile_bytes, e.tile_bytes, rank_of(p2, pc), s));
          }
        }
      }
    }
    return end();
  }

  struct Panel {
    std::vector<const void *> bases;
    std::vector<int> firsts;
    const void *head = nullptr;
  };
  int panel(int k, int s, bool send_head, Panel *p) {
    p->bases.assign(e.P, nullptr);
    p->firsts.assign(e.P, 0);
    RC(trsm_and_head(k, s, send_head, &p->head));
    return exchange(k, s, p->bases.data(), p->firsts.data());
  }

  double issue_us = 0;
  int factorize(bool lookahead, long long *info_out) {
    const auto t0 = std::chrono::steady_clock::now();
    const int nt = e.nt, P = e.P, Q = e.Q;
    RC(e.reset_info());
    if (!lookahead) {
      // the plain wave order on one stream (C2:506-565)
      for (int k = 0; k < nt; ++k) {
        RC(diag(k, S_MAIN));
        if (k + 1 < nt) {
          Panel p;
          RC(panel(k, S_MAIN, false, &p));
          RC(e.update(k, k + 1, nt, p.bases.data(), p.firsts.data(), 0, S_MAIN));
        }
      }
    } else {
      RC(e.wait_stream(S_SIDE, S_MAIN));
      RC(e.wait_stream(S_EARLY, S_MAIN));
      RC(diag(0, S_EARLY));
      RC(e.wait_stream(S_SIDE, S_EARLY));
      Panel cur, nxt;
      if (nt > 1) RC(panel(0, S_SIDE, true, &cur));
      bool have_u1 = false;
      for (int k = 0; k + 1 < nt; ++k) {
        // early: the diagonal tile of the next wave, as soon as the head tile L(k+1,k) is in
        RC(e.wait_event(S_EARLY, EV_HEAD0 + (k & 1)));
        if (have_u1) RC(e.wait_event(S_EARLY, EV_U1));  // (k+1,k+1) carries every update up to wave k-1
        if (rank_of((k + 1) % P, (k + 1) % Q) == e.rank) {
          std::vector<const void *> hb(P, cur.head);
          std::vector<int> hf(P, 0);
          hf[(k + 1) % P] = (k + 1) / P;
          RC(e.update_diag(k, k + 1, hb.data(), hf.data(), S_EARLY));
        }
        RC(diag(k + 1, S_EARLY));
        RC(e.record(S_EARLY, EV_DIAG));
        // main: panel k complete (and received); columns k+1 and k+2 first
        RC(e.wait_stream(S_MAIN, S_SIDE));
        RC(e.update(k, k + 1, k + 3, cur.bases.data(), cur.firsts.data(), 1, S_MAIN));
        RC(e.record(S_MAIN, EV_U1));
        have_u1 = true;
        if (k + 2 < nt) {
          RC(e.wait_event(S_SIDE, EV_U1));
          RC(e.wait_event(S_SIDE, EV_DIAG));
          RC(panel(k + 1, S_SIDE, true, &nxt));
        }
        RC(e.update(k, k + 3, nt, cur.bases.data(), cur.firsts.data(), 0, S_MAIN));
        std::swap(cur, nxt);
      }
      RC(e.wait_stream(S_MAIN, S_SIDE));
      RC(e.wait_stream(S_MAIN, S_EARLY));
    }
    issue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    RC(e.sync());
    int info = 0;
    RC(e.info(&info));
    // the smallest positive info wins: MAX-reduce (2^40 - info), 0 = success
    long long v = info > 0 ? (1LL << 40) - info : 0;
    if (P * Q > 1) RC(tr.allreduce_max(tr.ctx, &v));
    *info_out = v == 0 ? 0 : (1LL << 40) - v;
    return 0;
  }
};

// ---------------------------------------------------------------- product engine: HIP kernels
struct Pool {
  struct Blk {
    void *p;
    size_t bytes;
    bool used;
  };
  std::vector<Blk> blks;
  void *get(size_t bytes) {
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
  void release_all() {
    for (auto &b : blks) b.used = false;
  }
  void free_all() {
    for (auto &b : blks) (void)hipFree(b.p);
    blks.clear();
  }
};
Pool g_pool;
hipStream_t g_streams[3] = {nullptr, nullptr, nullptr};
hipEvent_t g_events[EV_COUNT];
bool g_hip_ready = false;

int hip_ready() {
  if (g_hip_ready) return 0;
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return chol_internal_fail(CHOL_ERR_HIP, "stream priority range");
  // CHOLMI_COMM_CUS = n: the update stream leaves n CUs of every XCD-pair alone, so that the transport's
  // kernels (RCCL) always find a free CU; they cannot raise the yield table themselves.  Off by default:
  // any mask costs the update more than it buys on one GPU.
  const char *cm = getenv("CHOLMI_COMM_CUS");
  const int reserve = cm ? atoi(cm) : 0;
  hipError_t e = hipSuccess;
  if (reserve > 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    e = hipGetDeviceProperties(&prop, dev);
    const int ncu = e == hipSuccess ? prop.multiProcessorCount : 256;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
    for (int c = 0; c < reserve && c < ncu; ++c) {
      const int cu = (int)((long)c * ncu / reserve);  // spread over the chip
      mask[cu / 32] &= ~(1u << (cu % 32));
    }
    e = hipExtStreamCreateWithCUMask(&g_streams[S_MAIN], (uint32_t)mask.size(), mask.data());
  } else {
    e = hipStreamCreateWithPriority(&g_streams[S_MAIN], hipStreamNonBlocking, lo);
  }
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&g_streams[S_SIDE], hipStreamNonBlocking, hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&g_streams[S_EARLY], hipStreamNonBlocking, hi);
  for (int i = 0; i < EV_COUNT && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&g_events[i], hipEventDisableTiming);
  if (e != hipSuccess) return chol_internal_fail(CHOL_ERR_HIP, hipGetErrorString(e));
  g_hip_ready = true;
  return 0;
}

struct HipEngine : Engine {
  chol_desc *d;
  explicit HipEngine(chol_desc *desc, int rank_) : d(desc) {
    geometry(desc->lm, desc->mbi, desc->p, desc->q, rank_, desc->esize);
  }
  char *store() override { return (char *)d->mat; }
  void *alloc(size_t bytes) override { return g_pool.get(bytes); }
  int potrf(int k, void *lkk, int s) override { return chol_wave_potrf(d, k, lkk, g_streams[s]); }
  size_t winv_bytes() override { return chol_wave_winv_bytes(d); }
  int export_winv(void *dst, int s) override { return chol_wave_export_winv(d, dst, g_streams[s]); }
  int import_winv(const void *src, int s) override { return chol_wave_import_winv(d, src, g_streams[s]); }
  int trsm(int k, const void *lkk, int s) override { return chol_wave_trsm(d, k, lkk, g_streams[s]); }
  int update(int k, int jlo, int jhi, const void *const *bases, const int *firsts, int skip_diag, int s) override {
    return chol_wave_update(d, k, jlo, jhi, bases, firsts, skip_diag, g_streams[s]);
  }
  int update_diag(int k, int j, const void *const *bases, const int *firsts, int s) override {
    return chol_wave_update_diag(d, k, j, bases, firsts, g_streams[s]);
  }
  int copy(void *dst, const void *src, size_t bytes, int s) override {
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_streams[s]) == hipSuccess
               ? 0
               : chol_internal_fail(CHOL_ERR_HIP, "hipMemcpyAsync (distributed potrf)");
  }
  void *stream(int s) override { return g_streams[s]; }
  int record(int s, int ev) override {
    return hipEventRecord(g_events[ev], g_streams[s]) == hipSuccess ? 0 : chol_internal_fail(CHOL_ERR_HIP, "hipEventRecord");
  }
  int wait_event(int s, int ev) override {
    return hipStreamWaitEvent(g_streams[s], g_events[ev], 0) == hipSuccess ? 0 : chol_internal_fail(CHOL_ERR_HIP, "hipStreamWaitEvent");
  }
  int sync() override {
    for (int s = 0; s < 3; ++s)
      if (hipStreamSynchronize(g_streams[s]) != hipSuccess) return chol_internal_fail(CHOL_ERR_HIP, "hipStreamSynchronize (distributed potrf)");
    return 0;
  }
  int reset_info() override { return chol_reset_info(); }
  int info(int *out) override { return chol_get_info(out); }
};

// ---------------------------------------------------------------- test engine: callbacks, no streams
struct CbEngine : Engine {
  chol_test_engine_t cb;
  std::vector<void *> owned;
  CbEngine(const chol_test_engine_t &c, int N_, int B_, int P_, int Q_, int rank_) : cb(c) {
    geometry(N_, B_, P_, Q_, rank_, c.esize);
  }
  char *store() override { return (char *)cb.store; }
  void *alloc(size_t bytes) override { return cb.alloc(cb.ctx, bytes); }
  int potrf(int k, void *lkk, int) override { return cb.potrf(cb.ctx, k, lkk); }
  size_t winv_bytes() override { return 0; }
  int export_winv(void *, int) override { return 0; }
  int import_winv(const void *, int) override { return 0; }
  int trsm(int k, const void *lkk, int) override { return cb.trsm(cb.ctx, k, lkk); }
  int update(int k, int jlo, int jhi, const void *const *bases, const int *firsts, int skip_diag, int) override {
    return cb.update(cb.ctx, k, jlo, jhi, bases, firsts, skip_diag);
  }
  int update_diag(int k, int j, const void *const *bases, const int *firsts, int) override {
    return cb.update_diag(cb.ctx, k, j, bases, firsts);
  }
  int copy(void *dst, const void *src, size_t bytes, int) override {
    memcpy(dst, src, bytes);
    return 0;
  }
  void *stream(int) override { return nullptr; }
  int record(int, int) override { return 0; }
  int wait_event(int, int) override { return 0; }
  int sync() override { return 0; }
  int reset_info() override { return 0; }
  int info(int *out) override {
    *out = cb.info(cb.ctx);
    return 0;
  }
};

// ---------------------------------------------------------------- the installed transport
chol_transport_t g_tr;
bool g_tr_set = false;
double g_last_issue_us_per_wave = 0;
long long g_last_sends = 0, g_last_recvs = 0, g_last_bytes = 0;

// ---------------------------------------------------------------- RCCL transport (loaded on demand)
typedef struct {
  char internal[128];
} nccl_uid_t;
struct Rccl {
  void *lib = nullptr;
  void *comm = nullptr;
  int (*GetUniqueId)(nccl_uid_t *) = nullptr;
  int (*CommInitRank)(void **, int, nccl_uid_t, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  long long *d_red = nullptr;
} g_rccl;

int rccl_fail(const char *what, int rc) {
  char buf[256];
  snprintf(buf, sizeof buf, "RCCL %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
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
int rccl_send(void *, const void *buf, size_t bytes, int peer, void *stream) {
  const int rc = g_rccl.Send(buf, bytes, /*ncclInt8*/ 0, peer, g_rccl.comm, (hipStream_t)stream);
  return rc ? rccl_fail("ncclSend", rc) : 0;
}
int rccl_recv(void *, void *buf, size_t bytes, int peer, void *stream) {
  const int rc = g_rccl.Recv(buf, bytes, /*ncclInt8*/ 0, peer, g_rccl.comm, (hipStream_t)stream);
  return rc ? rccl_fail("ncclRecv", rc) : 0;
}
int rccl_allreduce_max(void *, long long *value) {
  if (!g_rccl.d_red && hipMalloc(&g_rccl.d_red, sizeof(long long)) != hipSuccess)
    return chol_internal_fail(CHOL_ERR_OUT_OF_MEMORY, "RCCL reduction word");
  if (hipMemcpy(g_rccl.d_red, value, sizeof(long long), hipMemcpyHostToDevice) != hipSuccess)
    return chol_internal_fail(CHOL_ERR_HIP, "RCCL reduction word upload");
  const int rc = g_rccl.AllReduce(g_rccl.d_red, g_rccl.d_red, 1, /*ncclInt64*/ 4, /*ncclMax*/ 2, g_rccl.comm, nullptr);
  if (rc) return rccl_fail("ncclAllReduce", rc);
  if (hipStreamSynchronize(nullptr) != hipSuccess || hipMemcpy(value, g_rccl.d_red, sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess)
    return chol_internal_fail(CHOL_ERR_HIP, "RCCL reduction word download");
  return 0;
}

}  // namespace

extern "C" {

int chol_set_transport(const chol_transport_t *t) {
  if (!t) {
    g_tr_set = false;
    return 0;
  }
  if (!t->group_begin || !t->send || !t->recv || !t->group_end || !t->allreduce_max)
    return chol_internal_fail(-1, "chol_set_transport: every entry of the table is required");
  g_tr = *t;
  g_tr_set = true;
  return 0;
}

int chol_transport_rccl_unique_id(void *id128) {
  if (!id128) return chol_internal_fail(-1, "rccl_unique_id: NULL");
  RC(rccl_load());
  nccl_uid_t id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof id);
  return 0;
}

int chol_transport_rccl_init(const void *id128, int rank, int nranks) {
  if (!id128) return chol_internal_fail(-1, "rccl_init: NULL id");
  if (nranks < 1 || rank < 0 || rank >= nranks) return chol_internal_fail(-2, "rccl_init: rank");
  RC(rccl_load());
  if (g_rccl.comm) return chol_internal_fail(-1, "rccl_init: communicator already exists");
  nccl_uid_t id;
  memcpy(&id, id128, sizeof id);
  const int rc = g_rccl.CommInitRank(&g_rccl.comm, nranks, id, rank);
  if (rc) return rccl_fail("ncclCommInitRank", rc);
  chol_transport_t t;
  t.ctx = nullptr;
  t.group_begin = rccl_group_begin;
  t.send = rccl_send;
  t.recv = rccl_recv;
  t.group_end = rccl_group_end;
  t.allreduce_max = rccl_allreduce_max;
  return chol_set_transport(&t);
}

int chol_transport_rccl_finalize(void) {
  if (g_rccl.comm) {
    (void)g_rccl.CommDestroy(g_rccl.comm);
    g_rccl.comm = nullptr;
    g_tr_set = false;
  }
  if (g_rccl.d_red) (void)hipFree(g_rccl.d_red);
  g_rccl.d_red = nullptr;
  return 0;
}

int chol_dist_last_stats(double *issue_us_per_wave, long long *sends, long long *recvs, long long *bytes_sent) {
  if (issue_us_per_wave) *issue_us_per_wave = g_last_issue_us_per_wave;
  if (sends) *sends = g_last_sends;
  if (recvs) *recvs = g_last_recvs;
  if (bytes_sent) *bytes_sent = g_last_bytes;
  return 0;
}

// called by chol_potrf_tile (api.hip) for descriptors with p*q > 1
int chol_internal_dist_potrf(chol_desc *d, int rank) {
  if (!d->on_device) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "distributed potrf: the local tiles must be device-resident");
  if (d->mt != d->nt || d->lm != d->ln) return chol_internal_fail(-2, "potrf_tile: matrix is not square");
  if (d->padded || d->mbi != d->mb || d->lm % d->mb) return chol_internal_fail(CHOL_ERR_NOT_SUPPORTED, "distributed potrf: order must be a multiple of the tile, tile a multiple of 128");
  RC(hip_ready());
  HipEngine eng(d, rank);
  Dist dist(eng, g_tr_set ? &g_tr : nullptr);
  int rc = dist.setup();
  long long info = 0;
  static const bool lookahead = !(getenv("CHOLMI_DIST_LOOKAHEAD") && atoi(getenv("CHOLMI_DIST_LOOKAHEAD")) == 0);
  if (!rc) rc = dist.factorize(lookahead, &info);
  if (rc) {  // leave nothing half-open behind a failure: close the transport group, drain the streams
    (void)dist.end();
    (void)eng.sync();
  }
  g_pool.release_all();
  g_last_issue_us_per_wave = dist.issue_us / (eng.nt > 0 ? eng.nt : 1);
  g_last_sends = dist.nsend, g_last_recvs = dist.nrecv, g_last_bytes = dist.bytes_sent;
  if (rc) return rc;
  return (int)info;
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
  RC(hip_ready());
  const size_t tb = (size_t)src->bsizi * src->esize;
  hipStream_t st = g_streams[S_MAIN];
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
          RC(g_tr.group_begin(g_tr.ctx));
          grouped = true;
        }
        if (owner == rank)
          RC(g_tr.send(g_tr.ctx, mine, tb, root, st));
        else
          RC(g_tr.recv(g_tr.ctx, there, tb, owner, st));
      }
    }
    if (grouped) RC(g_tr.group_end(g_tr.ctx));
  }
  return hipStreamSynchronize(st) == hipSuccess ? 0 : chol_internal_fail(CHOL_ERR_HIP, "dist_gather_lower: synchronize");
}

void chol_internal_dist_finalize(void) {
  (void)chol_transport_rccl_finalize();
  g_pool.free_all();
  if (g_hip_ready) {
    for (int s = 0; s < 3; ++s) (void)hipStreamDestroy(g_streams[s]);
    for (int i = 0; i < EV_COUNT; ++i) (void)hipEventDestroy(g_events[i]);
    g_hip_ready = false;
  }
}

int chol_dist_factorize_with(const chol_test_engine_t *engine, const chol_transport_t *transport, int N, int B,
                             int p, int q, int rank, int lookahead) {
  if (!engine || !engine->store || !engine->alloc || !engine->potrf || !engine->trsm || !engine->update ||
      !engine->update_diag || !engine->info)
    return chol_internal_fail(-1, "dist_factorize_with: engine table incomplete");
  if (p * q > 1 && !transport) return chol_internal_fail(-2, "dist_factorize_with: transport required");
  if (N <= 0 || B <= 0 || N % B || p <= 0 || q <= 0 || p > cholmi::MAXP || rank < 0 || rank >= p * q)
    return chol_internal_fail(-3, "dist_factorize_with: geometry");
  CbEngine eng(*engine, N, B, p, q, rank);
  Dist dist(eng, transport);
  long long info = 0;
  int rc = dist.setup();
  if (!rc) rc = dist.factorize(lookahead != 0, &info);
  if (rc) (void)dist.end();
  g_last_issue_us_per_wave = dist.issue_us / (eng.nt > 0 ? eng.nt : 1);
  g_last_sends = dist.nsend, g_last_recvs = dist.nrecv, g_last_bytes = dist.bytes_sent;
  if (rc) return rc;
  return (int)info;
}

}  // extern "C"
