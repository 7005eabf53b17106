// Internal declarations shared by the kernel and API translation units of
// libcholmi.so.  Not part of the ABI (that is include/cholmi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

#include "../../include/cholmi.h"

// The descriptor behind chol_desc_t (include/cholmi.h): Chameleon's CHAM_desc_t arguments plus what
// this library derives from them.
struct chol_desc {
  int dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q;
  int mbi, bsizi;    // stored tile edge / size: mb rounded up to 128 when the library owns a padded image
  bool padded;       // stored tiles are larger than (mb, nb) and/or the last tile row/column is ragged
  int mt, nt;        // global tile grid
  int prow, pcol;    // this process's grid coordinates
  int lmt, lnt;      // local tile grid
  size_t esize;
  void *mat;         // storage as seen by the caller (host or device)
  bool on_device;    // mat is device memory
  bool owns;         // library allocated mat
  // static work list of the trailing updates: the local strictly-lower tiles sorted by column
  // descending (rows ascending inside a column), entries with column >= j in [0, ge[j]); then,
  // from n_off on, the local diagonal tiles sorted by column descending, [n_off, n_off + gd[j])
  int2 *d_list = nullptr;
  std::vector<int> ge, gd;
  int n_off = 0;
  // chol_desc_set_version: names the CONTENT behind `mat` (the worker: a hash of the write-once result id the
  // blob belongs to); lets the library keep the block inverses of a factored tile for the TRSM tasks that follow
  unsigned long long version = 0;
  // A sub-matrix view (i, j, m, n) over a USER buffer (tile-aligned offsets, whole tiles): `mat` is a library-owned
  // compact image of the view's tiles, refreshed from the user's tile matrix before every operation on this
  // descriptor and written back after every operation that modifies it (api.hip: with_views).
  void *user_mat = nullptr;
  int user_lmt = 0;              // tile rows of the user's matrix
  int user_i = 0, user_j = 0;    // the view's first ROW / COLUMN in it (any offset: a view may start inside a tile)
};

extern "C" int chol_internal_fail(int code, const char *msg);  // api.hip: sets chol_last_error, returns code

namespace cholmi {

// streams of one rank's factorisation (walker.h)
enum { ST_MAIN = 0, ST_PANEL, ST_TRSM, ST_U1, ST_CX, ST_PX, ST_COUNT };

constexpr int SEM_SLOTS = 65536;   // device-side counters: 3 mb/128 + 1 per tile column (+ the flow's control block) ...
constexpr int TILE_SEM_SETS = 8;   // ... plus rotating sets of 32 for the single-tile POTRF's fused in-tile steps
constexpr int SEM_INTS = (SEM_SLOTS + TILE_SEM_SETS * 32) * 32;  // ... each on a 128-byte line of its own

// device buffers that live as long as the context (receive buffers of the distributed walker)
struct DevPool {
  struct Blk {
    void *p;
    size_t bytes;
    bool used;
  };
  std::vector<Blk> blks;
  void *get(size_t bytes);
  void release_all() {
    for (auto &b : blks) b.used = false;
  }
  void free_all();
};

// Everything one rank's factorisation runs on: its streams, workspaces, counters, events, timings.  The process
// has one (api.hip: the context behind the C ABI); the one-GPU rehearsal of the distributed walker makes more.
struct RankCtx {
  int device = -1;
  hipStream_t st[ST_COUNT] = {};
  void *winv = nullptr;  // inverses of the 128x128 diagonal blocks of L(k,k), two sets (wave parity)
  size_t winv_bytes = 0;
  int *d_info = nullptr;
  int *d_sem = nullptr;  // device-side dependency counters of the panel chain (kernels.hip: sem_wait), or null
  bool flow_ok = false;  // ... and ST_CX / ST_PANEL have queues of their own: the flow form of the tile POTRF may be used
  unsigned tile_sem_next = 0;
  std::vector<hipEvent_t> events;
  std::vector<hipStream_t> retired;  // streams chol_init replaced because an update stream's launches held theirs back
  int stream_swaps = 0;          // ... how many, and pairs that still collide after the attempts
  int stream_collisions = 0;
  hipEvent_t ev_flow = nullptr;  // joins ST_CX to ST_PANEL once, at the first flow-form wave of a factorisation
  bool profiling = false;
  // stats of the last whole-matrix potrf
  double total_ms = 0, update_ms = 0, update_flops = 0, issue_us = 0;
  int update_launches = 0;
  int flow_waves = 0;  // waves of the last whole-matrix potrf whose tile POTRF ran in flow form (kernels.hip: k_flow_factor)
  int regimes[8] = {};  // ... and its waves per regime (walker.h: Walker::R_*), of regimes_nt waves (chol_last_potrf_regimes)
  int regimes_nt = 0;
  long long sends = 0, recvs = 0, bytes_sent = 0;
  DevPool pool;
  // measured once at creation, by dtype (0 = f64, 1 = f32): the register-only MFMA stream's rate [TFLOP/s]
  // and one 128 x 128 diagonal-block step of the panel chain alone [us]; what the walker's regime switches use
  double probe_tflops[2] = {0, 0}, diag_us[2] = {0, 0};
};
int rank_ctx_create(RankCtx *r, int device, const RankCtx *calib_from);  // api.hip
void rank_ctx_destroy(RankCtx *r);
RankCtx *main_rank_ctx();
int main_rank(int *nranks);

// The trailing update inside the DAG runs at about this fraction of the register-only MFMA stream, and one
// 128-step of the panel chain (diagonal block, in-tile solve and update, launch gaps) takes about this many
// times the diagonal-block kernel alone (rounds 1-2, fp64: 65 of 76.4 TFLOP/s; 130 of 49 us): the units the
// walker's regime thresholds were tuned in, now derived from what chol_init measures per dtype.
constexpr double CHOLMI_UPDATE_EFF = 0.85, CHOLMI_STEP_FACTOR = 2.65;

constexpr int MACRO = 128;  // macro-tile edge: one workgroup's C block, and the
                            // diagonal-block size of the in-tile POTRF/TRSM
constexpr int MAXP = 8;     // max process-grid rows a panel reference can address

// Where the panel tiles L(i,k) of the current wave live.  Tile i is at
// base[i % P] + (i / P - first[i % P]) * bsiz  (elements).  Single GPU: P = 1,
// base[0] = column k of the matrix itself, first[0] = 0 (no copy).
struct PanelRef {
  const void *base[MAXP];
  int first[MAXP];
  int P;
};

// This process's part of a 2D block-cyclic tile matrix.
struct LocalMat {
  void *base;
  int lmt;   // local tile rows (leading dimension of the local tile grid)
  int P, Q;  // process grid
  int mb;    // tile edge (= ld inside a tile)
  long bsiz; // elements per tile
};

extern int g_trsm_small_max;
extern int g_poll_max_wgs;
// the flow form of a counter-linked wave's tile POTRF (kernels.hip: k_flow_factor): does it apply to tiles of nbm 128-blocks
bool flow_applies(int nbm);
extern int g_flow;
extern int g_flow_min_nbm;
extern int g_flow_max_nbm;
extern int g_flow_fences;
extern int g_intile_fused;
extern int g_min_units;
extern unsigned long long *g_dbg;
extern int *g_ytab;
constexpr int YTAB_ENTRIES = 2048;

// ---- launchers (kernels.hip) ---------------------------------------------
// C(i,j) -= L(i,k) L(j,k)^T for the (i,j) pairs in d_list[off .. off+na) followed by
// d_list[offb .. offb+nb) (off-diagonal tiles first, the diagonal tiles -- whose blocks above the
// diagonal exit at once -- last, so that they do not split the 64-block cohorts of an XCD);
// yield: the update's waves give their CU to guest workgroups of the panel chain (kernels.hip);
// pan2 != null: the updates by two panels in one pass (C -= L L^T of `pan`, then of `pan2`)
template <typename T>
void launch_trail_update(hipStream_t s, const LocalMat &C, const int2 *d_list, int off, int na, int offb,
                         int nb, const PanelRef &pan, bool yield = false, const PanelRef *pan2 = nullptr);

// In-tile blocked POTRF of one mb x mb tile (device pointer, ld = mb).  Writes the
// inverses of the MACRO x MACRO diagonal blocks of L to winv (mb/MACRO blocks of
// MACRO*MACRO elements, ld = MACRO).  info: device int, set to info_base + j (1-based)
// at the first non-positive pivot (first writer wins).
template <typename T>
void launch_potrf_tile(hipStream_t s, T *tile, int mb, T *winv, int *d_info, int info_base, int *sem = nullptr);

// C (mb x mb tile, lower part) -= A A^T, A one tile: 64 x 64 blocks, one workgroup each (critical chain)
template <typename T>
void launch_diag_syrk(hipStream_t s, T *C, const T *A, int mb);

// POTRF(tile) on stream sp with the TRSM of `ntiles` contiguous tiles pipelined behind it on
// stream st (ev: mb/MACRO + 1 events).  ev_head (may be null): recorded once the panel is solved.
// chol_init's probe: the consumer kernel goes first, polls *sem (zeroed) for <= ~20 ms and writes 1 (seen) or
// 2 (gave up) to *result; the producer kernel raises *sem
void launch_sem_probe(hipStream_t consumer, hipStream_t producer, int *sem, int *result);
// chol_init's probe of two streams' dispatch paths (kernels.hip: k_pipe_big): t[0] / t[1] = start of the many-round launch on
// `big` / of the one-wave kernel launched right behind it on `small` (100 MHz ticks)
void launch_pipe_probe(hipStream_t big, hipStream_t small, unsigned long long *t, int cus);

// The chain-bound form of a wave (device-side edges, kernels.hip: sem_wait).  The SYRK on the next diagonal
// tile, C(k+1,k+1) -= L(k+1,k) L(k+1,k)^T, is cut into the K = 128 slices of the head tile's block columns
// and issued on `su`; the TRSM steps go to the TRSM stream as before -- but every kernel of the three
// streams is launched ahead of time, without a stream operation, and polls the counter of what it needs:
//   diagonal-block step s  -->  TRSM step s: solve  -->  slice s          in-tile solve s  -->  TRSM step s: update
//   last slice  -->  POTRF(k+1)'s first diagonal-block step (launch_panel_pipelined's wait_sem / wait_target)
struct SyrkPipe {
  void *c;         // tile (k+1,k+1)
  hipStream_t su;  // has already waited for the earlier writers of that tile
  int *sem;        // 3 nbm + 1 zeroed counters of this wave, 32 ints (one 128-byte line) apart; the last one
                   // counts the last slice's workgroups: n (n + 1) / 2, n = mb / 64
  // the tile POTRF as a flow (kernels.hip: k_flow_factor / k_flow_rows), or null: its zeroed control block
  // (flow_lines(nbm, nbm) lines), the stream the row-slab waves run on, an event to join it once
  int *fc = nullptr;
  hipStream_t sflow = nullptr;
  hipEvent_t ev_flow = nullptr;
  bool join_flow = false;  // the flow stream has to join the POTRF stream's order by an event (walker.h)
};
inline int flow_ctl_lines(int nbm) { return nbm >= 2 && nbm <= 8 ? 1 + nbm + 2 * nbm * nbm : 0; }

template <typename T>
void launch_panel_pipelined(hipStream_t sp, hipStream_t st, hipEvent_t *ev, T *lkk, int mb, T *winv,
                            int *d_info, int info_base, T *tiles, long bsiz, int ntiles,
                            hipEvent_t ev_head = nullptr, const SyrkPipe *sy = nullptr, const int *wait_sem = nullptr,
                            int wait_target = 0, int *tile_sem = nullptr);

template <typename T>
void launch_col_update_small(hipStream_t s, T *C, const T *A, const T *B, int mb, int ntiles, long bsiz, int *done);

// winv from an already factored tile
template <typename T>
void launch_invert_diag(hipStream_t s, const T *tile, int mb, T *winv);

// tiles[t] := alpha * tiles[t] * L^{-T}, t < ntiles, tiles contiguous (stride bsiz)
template <typename T>
void launch_trsm_panel(hipStream_t s, T *tiles, long bsiz, int ntiles, const T *lkk, const T *winv,
                       int mb, T alpha);

// generic 1-tile C := alpha A B^T + beta C (lower_only: SYRK semantics)
template <typename T>
void launch_gemm_nt_tile(hipStream_t s, const T *A, const T *B, T *C, int mb, T alpha, T beta,
                         bool lower_only);

// the same product for nz1 x nz2 tiles in one launch: A + z1 sA, B + z2 sB, C + z1 sC1 + z2 sC2
template <typename T>
void launch_gemm_nt_batch(hipStream_t s, const T *A, long sA, int nz1, const T *B, long sB, int nz2, T *C, long sC1,
                          long sC2, int mb, T alpha, T beta);

// cout[z] = cin[z] - a[z] b[z]^T for n tasks given as device arrays of device tile pointers, out of place; b[z] == null:
// a SYRK task (b = a, the lower triangle updated, the strict upper one copied) -- kernels.hip: k_update_ptrs_w8
template <typename T>
void launch_update_ptrs(hipStream_t s, const T *const *cin, const T *const *a, const T *const *b, T *const *cout, int n, int mb,
                        bool yield);
// dst[z] <- src[z], z < n, `bytes` (a multiple of 16) each; src / dst: device arrays of device pointers
void launch_copy_ptrs(hipStream_t s, const void *const *src, void *const *dst, int n, long bytes);

template <typename T>
void launch_plgsy(hipStream_t s, const LocalMat &A, int lnt, int prow, int pcol, double bump,
                  unsigned long long seed, int mbu, long nglob, int side);

// accumulates sum((LL^T - A)^2) and sum(A^2) over the lower triangle (strict part
// counted twice) into acc[0], acc[1] (device doubles).  Single process only.
template <typename T>
void launch_residual(hipStream_t s, const T *Lbase, int Nb, int mb, double bump,
                     unsigned long long seed, double *d_acc, int mbu, long nglob,
                     double *rowsum = nullptr);  // rowsum: 2*nglob doubles (|R| rows, then |A| rows), zeroed

// pad helpers for the staged 1-tile path: dst is ldp x ldp (zeroed), identity on
// the padded part of the diagonal when `unit_pad`.
template <typename T>
void launch_pad_identity(hipStream_t s, T *dst, int n, int ldp);

// in-place transpose of a whole nt x nt tile matrix (mb multiple of 64)
template <typename T>
void launch_transpose_inplace(hipStream_t s, T *M, int nt, int mb);

// ---- launchers (verify_ops.hip): the driver's validation block, v6_test.c:51, 74-85 ----
// Geometry of a single-process stored tile image: lmt x lnt tiles of mbs x mbs elements, of
// which the caller's tile is the leading mbu x mbu part; the matrix is m x n.
struct TileGeo {
  int lmt, lnt, mbs, mbu;
  long m, n;
};
// side: 0 all, 1 on or below the diagonal, 2 on or above
template <typename T>
void launch_lacpy(hipStream_t s, const TileGeo &g, int side, const T *A, T *B);
template <typename T>
void launch_geadd(hipStream_t s, const TileGeo &g, double alpha, const T *A, double beta, T *B);
// work: max(m, n) + 2 device doubles; result in work[0] (kind 0 max, 1 one, 2 inf, 3 sum of squares)
template <typename T>
void launch_lange(hipStream_t s, const TileGeo &g, int kind, const T *A, double *work);
// out(I,J) = sum_{K>=I} L(K,I)^T L(K,J), I >= J (lower part only; out != L)
template <typename T>
void launch_lauum_lower(hipStream_t s, const T *L, T *out, int nt, int mbs);

// out-of-place transposes of `count` mb x mb tiles (mb % 64 == 0)
template <typename T>
void launch_tiles_transpose(hipStream_t s, const T *in, long istride, T *out, long ostride, int mb, int count);

// register-only MFMA stream (blocks x 256 threads, 16 MFMA per wave per iteration)
template <typename T>
void launch_mfma_probe(hipStream_t s, T *out, int blocks, int iters);

}  // namespace cholmi
