/*
 * cholmi.h -- C ABI of libcholmi.so: the MI355X (gfx950) replacement for the
 * Chameleon calls on the reference's tiled-Cholesky path.
 *
 * Every entry point is `extern "C"`, takes plain ints / doubles / pointers, and
 * replaces exactly one call the reference makes into libchameleon.so.  Citations
 * are relative to /root/reference/ :
 *   W2 = cholesky_armonik/w_c_cons_v2/worker_construction2/src/worker_distrib.cpp
 *   C2 = cholesky_armonik/w_c_cons_v2/client_construction2/client/src/client_distrib.cpp
 *   V6 = Cholesky_chameleon_VM/cho/docker_installation_and_bench_files/v6_test.c
 *
 * Conventions
 *   - All tile operations are synchronous at this boundary (they return after
 *     the result is visible in the descriptor's memory), as Chameleon's
 *     non-_Async API is.
 *   - Return value: 0 on success; > 0 a LAPACK-style `info` (POTRF: 1-based
 *     global index of the first non-positive pivot); < 0 the negated 1-based
 *     position of the first invalid argument; <= -100 a runtime failure
 *     (CHOL_ERR_*).  The reference treats any value != 0 as failure
 *     (W2:243, 327, 420).
 *   - There is NO CPU backend.  chol_init(ncpu, 0) fails with CHOL_ERR_NO_GPU:
 *     the product path never falls back to host arithmetic.
 *   - Storage: a descriptor's `mat` may be NULL (the library allocates HBM in
 *     Chameleon tile layout, V6:44), a device pointer (wrapped in place, tiles
 *     stay resident) or a host pointer (W2:78: staged H2D/D2H around every
 *     call, which is what the reference's worker does with its blobs).
 *   - Tile layout (Chameleon descriptor): tile (I,J) of an lmt x lnt tile grid
 *     is `bsiz` contiguous elements at offset (I + J*lmt)*bsiz, column-major
 *     inside with leading dimension mb.  With p*q > 1 each process holds the
 *     tiles with (I mod p, J mod q) == its grid coordinates, packed the same
 *     way over local tile indices (I/p, J/q).
 */
#ifndef CHOLMI_H
#define CHOLMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Chameleon / PLASMA enum values (ChamRealDouble etc. are referenced at W2:78,
 * 238, 323, 416, 511; chameleon.h itself is not in the reference tree). */
enum {
  CHOL_REAL_FLOAT = 2,  /* ChamRealFloat  */
  CHOL_REAL_DOUBLE = 3, /* ChamRealDouble */
  CHOL_NOTRANS = 111,   /* ChamNoTrans */
  CHOL_TRANS = 112,     /* ChamTrans   */
  CHOL_UPPER = 121,     /* ChamUpper   */
  CHOL_LOWER = 122,     /* ChamLower   */
  CHOL_UPPER_LOWER = 123, /* ChamUpperLower (V6:51) */
  CHOL_NONUNIT = 131,   /* ChamNonUnit */
  CHOL_UNIT = 132,      /* ChamUnit    */
  CHOL_LEFT = 141,      /* ChamLeft    */
  CHOL_RIGHT = 142,     /* ChamRight   */
  CHOL_ONE_NORM = 171,       /* ChamOneNorm       */
  CHOL_FROBENIUS_NORM = 174, /* ChamFrobeniusNorm */
  CHOL_INF_NORM = 175,       /* ChamInfNorm (V6:74) */
  CHOL_MAX_NORM = 177        /* ChamMaxNorm       */
};

enum {
  CHOL_SUCCESS = 0,
  CHOL_ERR_NOT_INITIALIZED = -101,
  CHOL_ERR_NO_GPU = -102,        /* no HIP device / ngpu == 0 requested */
  CHOL_ERR_HIP = -103,           /* a HIP runtime call failed (see chol_last_error) */
  CHOL_ERR_NOT_SUPPORTED = -104, /* valid Chameleon usage this build does not cover */
  CHOL_ERR_OUT_OF_MEMORY = -105,
  CHOL_ERR_DEVICE_WAIT = -106    /* a bounded device-side wait inside the library gave up: the result is invalid */
};

typedef struct chol_desc chol_desc_t;

/* CHAMELEON_Init(ncpu, ngpu) W2:589, V6:41 / CHAMELEON_Finalize() V6:93.
 * Process-global and idempotent.  `ncpu` is accepted for signature parity and
 * ignored (host threads do no arithmetic); `ngpu` must be >= 1: this process
 * drives HIP device `chol_set_device`'s value (default: $LOCAL_RANK or 0). */
int chol_init(int ncpu, int ngpu);
int chol_finalize(void);
int chol_set_device(int device); /* call before chol_init */
const char *chol_last_error(void);
const char *chol_version(void);

/* One process per GPU: position of this process in the p x q grid that
 * descriptors with p*q > 1 refer to (Chameleon takes it from MPI).  rank =
 * prow*q + pcol, i.e. tile (I,J) belongs to rank (I mod p)*q + (J mod q). */
int chol_set_rank(int rank, int nranks);

/* CHAMELEON_Desc_Create(&d, mat, dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q)
 * W2:78 (1-tile wrap of a user buffer), V6:44 (mat = NULL: library-owned tile
 * storage) / CHAMELEON_Desc_Destroy W2:256, V6:90-91.
 * Sub-matrix views (i, j, m, n) != (0, 0, lm, ln), any offset: with mat = NULL the view is a matrix of its
 * own; over a user buffer (a matrix of whole mb = nb tiles) the library mirrors the view through a device
 * image around every operation -- entries outside the view are never touched. */
int chol_desc_create(chol_desc_t **desc, void *mat, int dtype, int mb, int nb, int bsiz, int lm,
                     int ln, int i, int j, int m, int n, int p, int q);
int chol_desc_destroy(chol_desc_t **desc);

/* Names the CONTENT behind a 1-tile descriptor's device buffer (the worker: a hash of the write-once result id
 * its blob belongs to; 0 = unnamed).  chol_potrf_tile on a tagged in-place device tile keeps the tile's block
 * inverses; chol_trsm_tile with an L of the same buffer AND tag reuses them instead of re-inverting L(k,k) for
 * every TRSM task of the wave (W2:323 is called once per panel tile with the same L). */
int chol_desc_set_version(chol_desc_t *desc, unsigned long long version);

/* CHAMELEON_dpotrf_Tile(uplo, A) W2:238, V6:56 (spotrf by descriptor dtype).
 * Works on a 1-tile descriptor (worker path), on a whole tiled matrix (driver path: the full wave
 * DAG of C2:506-565 runs on the device) and on a p x q block-cyclic descriptor (one process per GPU,
 * ChamLower; needs a transport, see chol_set_transport below).  ChamUpper is served for device-resident
 * matrices by transposing the storage in place around the Lower factorisation (the strict
 * lower triangle is returned untouched).
 * Returns LAPACK's info (0, or the 1-based index of the first non-positive pivot) or a negative CHOL_ERR_*;
 * CHOL_ERR_DEVICE_WAIT -- never expected -- says that a bounded device-side wait inside the library gave
 * up (the matrix is then left partly factored; the context falls back to stream events for later calls;
 * DESIGN.md, section 4). */
int chol_potrf_tile(int uplo, chol_desc_t *A);

/* CHAMELEON_dtrsm_Tile(side, uplo, trans, diag, alpha, A, B) W2:323.
 * Supported: (Right, Lower, Trans, NonUnit), any alpha; 1-tile descriptors. */
int chol_trsm_tile(int side, int uplo, int trans, int diag, double alpha, chol_desc_t *A,
                   chol_desc_t *B);

/* CHAMELEON_dsyrk_Tile(uplo, trans, alpha, A, beta, C) W2:416.
 * Supported: (Lower, NoTrans), any alpha/beta; the strict upper triangle of C
 * is never read or written. 1-tile descriptors. */
int chol_syrk_tile(int uplo, int trans, double alpha, chol_desc_t *A, double beta, chol_desc_t *C);

/* CHAMELEON_dgemm_Tile(transA, transB, alpha, A, B, beta, C) W2:511.
 * Supported: (NoTrans, Trans), any alpha/beta. 1-tile descriptors. */
int chol_gemm_tile(int transA, int transB, double alpha, chol_desc_t *A, chol_desc_t *B,
                   double beta, chol_desc_t *C);

/* The tasks of ONE op class of a wave in one grouped launch ("submit a whole wave, one wait per
 * wave"; W2:323 / 416 / 511 executed for n tasks at once), on HBM-resident mb x mb tiles (mb a multiple of 128):
 *   CHOL_BATCH_TRSM    c_out[t] = c_in[t] * a[t]^{-T}           (Right, Lower, Trans, NonUnit; a[t] = L(k,k))
 *   CHOL_BATCH_SYRK    c_out[t] = c_in[t] - a[t] a[t]^T, lower triangle (the strict upper triangle is copied)
 *   CHOL_BATCH_GEMM    c_out[t] = c_in[t] - a[t] b[t]^T
 *   CHOL_BATCH_UPDATE  the SYRK and GEMM tasks of a wave together: b[t] == NULL marks a SYRK task
 * c_in / a / b / c_out: HOST arrays of n device pointers (b ignored for TRSM / SYRK).  c_out[t] must not alias an
 * input: every task writes its private copy (W2:212-213) -- SYRK / GEMM / UPDATE in ONE out-of-place launch of the
 * trailing-update kernel (the copy IS the kernel's write: no copy pass), TRSM after a copy launch.  Results agree
 * with n calls of chol_trsm_tile / chol_syrk_tile / chol_gemm_tile to rounding (same products, same K order; the
 * kernels differ, so bit-identity is not promised).
 * a_versions (may be NULL): content tags of the a[t] buffers (chol_desc_set_version's meaning) -- TRSM reuses
 * the block inverses kept for (a[t], tag) by the POTRF batch instead of re-inverting L(k,k).
 * flags: CHOL_BATCH_ASYNC -- return once the work is enqueued (later batches are ordered behind it by what they
 *        read: see below; chol_sync() waits for everything).
 *        CHOL_BATCH_URGENT -- (SYRK / GEMM / UPDATE) the tasks feed the panel chain: column k+1 of wave k.
 * Execution is dependency-driven on two streams: POTRF and TRSM batches and URGENT updates run on a high-priority
 * chain stream, other updates on the bulk stream; a batch that reads a tile written by a batch of the other stream
 * waits for exactly that batch (tile pointers are the keys: results are write-once buffers).  So POTRF / TRSM of
 * wave k+1 run beside the bulk update of wave k, as in the whole-matrix walker, whenever the caller submits column
 * k+1's tasks URGENT and ahead of the rest (the reference's TaskOptions.priority, C2:335, is the natural carrier). */
enum { CHOL_BATCH_TRSM = 1, CHOL_BATCH_SYRK = 2, CHOL_BATCH_GEMM = 3, CHOL_BATCH_UPDATE = 4 };
enum { CHOL_BATCH_ASYNC = 1, CHOL_BATCH_URGENT = 2 };
int chol_tile_batch(int op, int dtype, int mb, int n, const void *const *c_in, const void *const *a,
                    const void *const *b, void *const *c_out, const unsigned long long *a_versions, int flags);
/* Waits for everything enqueued by the batch calls (both streams) and for the library's main stream. */
int chol_sync(void);
/* The POTRF tasks of a wave the same way: a_out[t] <- a_in[t] (private copy), factored Lower in place.  versions
 * (may be NULL) tags a_out[t]'s content, so that its block inverses serve the TRSM batch that follows.  The
 * LAPACK info of task t lands in a device slot: slots[t] (host array of n ints) receives its index, and
 * chol_batch_info(slot, &info) reads it (after waiting for the library's stream).  With CHOL_BATCH_ASYNC the call
 * returns at once: a failing factorisation is discovered when the caller asks. */
int chol_potrf_batch(int dtype, int mb, int n, const void *const *a_in, void *const *a_out,
                     const unsigned long long *versions, int *slots, int flags);
int chol_batch_info(int slot, int *info);
/* A mark = the batches enqueued so far (mark2: two sequence numbers, chain and bulk stream); chol_batch_wait blocks
 * until everything up to a mark has run WITHOUT draining what was enqueued after it -- what a client that releases
 * superseded tile versions needs: the readers of a version were all enqueued before the mark taken after them. */
int chol_batch_mark(unsigned long long *mark2);
int chol_batch_wait(const unsigned long long *mark2);
/* Executor statistics since chol_init: out4 = {batches on the chain stream, batches on the bulk stream, event waits
 * between the two, tile pointers currently remembered} (bench.py / tests: the overlap is really there). */
int chol_batch_stats(long long *out4);

/* CHAMELEON_dplgsy_Tile(bump, uplo, A, seed) V6:46: Chameleon's generator (published
 * core_dplgsy: 64-bit LCG with jump-ahead, entry (i,j), i >= j, = 0.5 - ran_{i + j*m} / 2^64,
 * symmetric, `bump` added to the diagonal), directly in tile layout on the device.  Values
 * depend only on (global row, global col, matrix order m, seed): independent of tile size
 * and process grid.  uplo = ChamLower / ChamUpper: the tiles on that side and the diagonal
 * tiles in full, the other tiles are left as they are (Chameleon's rule; library-allocated
 * storage starts at zero); ChamUpperLower: every tile. */
int chol_plgsy_tile(double bump, int uplo, chol_desc_t *A, unsigned long long seed);

/* The validation block of the reference driver, V6:51 and V6:72-86, on device-resident
 * single-process descriptors of identical geometry:
 *   CHAMELEON_dlacpy_Tile(uplo, A, B)            V6:51, 79   B <- A on the uplo part
 *   CHAMELEON_dlange_Tile(norm, A)               V6:74, 85   *value <- the norm
 *   CHAMELEON_dlauum_Tile(ChamLower, A)          V6:80       tril(A) <- tril(L^T L), L = tril(A)
 *   CHAMELEON_dgeadd_Tile(ChamNoTrans, a, A, b, B) V6:83     B <- a A + b B
 * (s-precision by descriptor dtype).  dlauum takes a scratch copy of the matrix. */
int chol_lacpy_tile(int uplo, chol_desc_t *A, chol_desc_t *B);
int chol_lange_tile(int norm, chol_desc_t *A, double *value);
int chol_lauum_tile(int uplo, chol_desc_t *A);
int chol_geadd_tile(int trans, double alpha, chol_desc_t *A, double beta, chol_desc_t *B);

/* The step after the factor (not called by the reference):
 *   CHAMELEON_dpotrs_Tile(ChamLower, A, B)   B <- A^{-1} B with A = L L^T already factored
 *   CHAMELEON_dposv_Tile(ChamLower, A, B)    factor A, then solve; info > 0: A not SPD, B untouched
 * A: n x n, B: n x nrhs, device-resident single-process descriptors with the same tile size. */
int chol_potrs_tile(int uplo, chol_desc_t *A, chol_desc_t *B);
int chol_posv_tile(int uplo, chol_desc_t *A, chol_desc_t *B);

/* CHAMELEON_Lapack_to_Tile / Tile_to_Lapack equivalents (host LAPACK layout
 * <-> descriptor storage); single-process descriptors only. */
int chol_lapack_to_tile(const void *A, int lda, chol_desc_t *desc);
int chol_tile_to_lapack(chol_desc_t *desc, void *A, int lda);

/* Single-tile transfer for any descriptor this process owns tile (I,J) of. */
int chol_tile_upload(chol_desc_t *desc, int I, int J, const void *host_tile);
int chol_tile_download(chol_desc_t *desc, int I, int J, void *host_tile);

/* ||tril(L) tril(L)^T - A||_F / ||A||_F with A regenerated by the plgsy rule
 * (the check V6:72-87 intended).  L = factored descriptor. */
int chol_residual_plgsy(chol_desc_t *L, double bump, unsigned long long seed, double *rel);
/* The quantity v6_test.c:72-86 prints, computed correctly: ||A - L L^T||_inf / ||A||_inf. */
int chol_residual_plgsy_inf(chol_desc_t *L, double bump, unsigned long long seed, double *rel_inf);

/* ---- client-side host helpers (pure host code, usable without a GPU) ------ */
/* make_spd_like_chameleon C2:224-252 + enforce_strict_diag_dominance C2:255-264 */
void chol_make_spd_like_chameleon(double *A, int N, int LDA, double bump, char uplo,
                                  unsigned long long seed);
void chol_enforce_strict_diag_dominance(double *A, int N, int LDA, double eps);
/* handle_json W2:47-69 for the n payloads of a wave in one call (the worker's wave-level execution):
 * buf = the payloads back to back, payload t = buf[off[t] .. off[t+1]).  op[t]: 1 TRSM, 2 SYRK, 3 GEMM, 4 POTRF, B[t],
 * and id_off / id_len [3 t + r]: where the op's tile ids sit in buf, in the order chol_tile_batch takes them (TRSM
 * {inA, inL}, SYRK {inC, inA}, GEMM {inC, inAi, inAj}, POTRF {in}).  op[t] = 0: not a flat payload of these four ops
 * -- the caller's general JSON parser decides about it, with the reference's messages.  Host only. */
int chol_parse_payloads(const char *buf, const long long *off, int n, int *op, int *B, long long *id_off, int *id_len);
/* extract_block_from_spd_matrix_colmajor C2:280-309 */
void chol_extract_block(const double *A, int N, int LDA, int B, int bi, int bj, double *block);

/* ---- instrumentation ------------------------------------------------------ */
/* Device time (HIP events on the library's own streams) of the last whole-matrix
 * chol_potrf_tile: total ms and the trailing-update kernel's launch count / ms. */
int chol_last_potrf_stats(double *total_ms, double *update_ms, int *update_launches,
                          double *update_flops);
/* The schedule the last whole-matrix chol_potrf_tile actually ran -- the walker picks a regime per wave from speeds
 * measured at chol_init (chol_debug_calibration), so two boxes may run two schedules; with this a number can be tied
 * to its schedule afterwards.  out8: waves that were {0 paired, 1 plain (one far launch), 2 near / far halves,
 * 3 counter-linked chain, 4 ... with the near column, 5 ... with the tile POTRF as a flow, 6 update yields CUs to the
 * chain, 7 column k+1 in the latency form}; 0-2 and 4 partition the waves with an update, 3 / 5-7 are attributes.
 * *nt (may be NULL): the number of waves. */
int chol_last_potrf_regimes(int *out8, int *nt);
/* 1 = bracket the trailing-update launches of every wave (pair of waves) with HIP events
 * on the stream the big launch runs on; the wait that closes a bracket joins launches the
 * next wave's big launch depends on anyway, so no dependency is added (walker.h) -- what
 * remains is the cost of the event records, which bench.py puts on record beside the
 * bracketed steps ("unprofiled_ms").  For bench.py's roofline leg; 0 = off (default). */
int chol_set_profiling(int on);

/* Diagnostic: time the trailing-update launch of wave k alone (best of `reps`), on whatever
 * data the descriptor holds (the matrix is modified).  ablate must be 0 (the ablation twin of rounds 1-4 left the
 * library in round 5: CHOL_ERR_NOT_SUPPORTED). */
int chol_bench_update(chol_desc_t *desc, int k, int ablate, int reps, double *ms, double *flops);

/* Diagnostic: enable = 1 starts recording, per diagonal-block workgroup, 8 words of 100 MHz
 * realtime ticks {start, end, loaded, sum phase A, sum phase B, L stored, block inverses done,
 * factor inverse done}; enable = 0 stops, copies up to max_pairs records (8 words each) to
 * `out` and returns their number. */
int chol_debug_stamps(int enable, unsigned long long *out, int max_pairs);

/* Register-only 16x16x4 MFMA stream on every CU (waves_per_simd = 1..8): the matrix-core
 * rate this chip sustains under load, to quote beside the datasheet peak. */
int chol_mfma_probe(int dtype, int waves_per_simd, double *tflops);

/* Diagnostic: how many waves of the last whole-matrix chol_potrf_tile factored their diagonal tile in the flow form
 * (two persistent launches handing 16-column panels to each other through polled counters). */
int chol_debug_flow_waves(void);

/* Test hook, no GPU needed: the wave walker (one GPU, nt x nt tiles of edge mb) run over an engine that executes nothing and
 * records every launch with the tiles it reads and writes, every event record / wait and every counter a stream is gated
 * on; then every pair of launches that touch the same tile (or block-inverse workspace), one of them writing, must be
 * ordered by a stream, an event or a counter.  t_tile / t_panel [s]: the speeds the walker picks its regimes from (the
 * CHOLMI_* schedule switches apply as in the product); profiling: with the per-wave event brackets.  Returns the number of
 * findings (0 = every conflict ordered; < 0: the walker failed); `report` (cap bytes) receives them, one per line, and a
 * last line "<launches> launches, <w> event waits, <c> counter edges, <n> flow-form waves, <m> findings".
 * CHOLMI_CHECK_DROP_WAIT / CHOLMI_CHECK_DROP_GATE = n: the checker's self-test -- it ignores the n-th event wait / counter
 * edge the walker asks for, i.e. checks a schedule with that dependency missing. */
int chol_debug_schedule_check(int nt, int mb, double t_tile, double t_panel, int profiling, char *report, int cap);
/* ... and rank `rank`'s share of the same factorisation on a p x q grid: its sends read and its receives write (tiles, the
 * diagonal / head tile buffers by wave parity, the panel buffers by wave mod 4) on the streams they are issued on. */
int chol_debug_schedule_check_grid(int nt, int mb, int p, int q, int rank, double t_tile, double t_panel, int profiling,
                                   char *report, int cap);
/* ... and that rank's whole launch graph as text, one launch per line in issue order:
 * "<id> <channel or -1> <S|R|-> <peer> <bytes> <group> <dep,dep,...|-> <kind> <seconds>" (kernel launches: channel -1, kind
 * P / T / U / Y / L and a rough duration from t_tile / t_panel -- scripts/predict_scale.py; transport calls: channel
 * 0 = diagonal and head tiles, 1 = panels, the peer, the size, the group of that channel they were issued in; deps = the
 * launches it is ordered behind by stream order, events and counters).  Returns the checker's findings, < 0 on error. */
int chol_debug_comm_trace(int nt, int mb, int p, int q, int rank, double t_tile, double t_panel, char *out, int cap);

/* The ordering of the task executor (chol_tile_batch / chol_potrf_batch) checked WITHOUT a GPU.
 * chol_debug_task_record(1, mutate) -- only before chol_init -- puts the executor into recording mode: the grouped-launch
 * entry points (and chol_batch_mark / _wait / _info, chol_sync) run their real dependency tracking on whatever pointer
 * values the caller passes (they are never dereferenced), skip every HIP call and log each batch with the tiles it reads
 * and writes, its stream, its event waits and what the host knew to be complete.  mutate >= 0 drops that one event wait
 * (self-test).  chol_debug_task_check(out5) then examines every pair of batches that touch the same tile, one of them
 * writing: out5 = {batches, event waits, pairs examined, pairs NOT ordered, pairs ordered by a host-side wait};
 * chol_last_error() describes the first unordered pair.  chol_debug_task_record(0, -1) leaves the mode. */
int chol_debug_task_record(int on, int mutate);
int chol_debug_task_check(long long *out5);

/* What the walker's regime switches (pairs / halves / counter-linked chain / CU hand-over) are measured in,
 * taken once at chol_init (or from CHOLMI_CALIB="tf64,us64,tf32,us32"): out8[0..3] = fp64 MFMA probe
 * [TFLOP/s], fp64 128 x 128 diagonal-block step alone [us], the same for fp32; out8[4..7] = the derived
 * update rate inside the DAG [TFLOP/s] and panel-chain step [us] per dtype that the thresholds use. */
int chol_debug_calibration(double *out8);
/* Name of the trailing-update kernel launched for `dtype`, as a profiler prints it (bench.py's roofline.kernel). */
int chol_debug_update_kernel(int dtype, char *buf, int buflen);
/* 1 if chain-bound waves can use device-side counters (chol_init's probe found the panel streams on independent
 * hardware queues, and no device-side wait has timed out since), 0 if every dependency is a stream event -- e.g.
 * under a profiler that serialises kernels, where the probe fails by design. */
int chol_debug_device_counters(void);

/* ---- local storage of a descriptor ---------------------------------------- */
void *chol_desc_local_ptr(chol_desc_t *desc, size_t *bytes);
int chol_desc_local_tiles(chol_desc_t *desc, int *lmt, int *lnt);

/* ---- distributed factorisation behind chol_potrf_tile --------------------- */
/* chol_potrf_tile(ChamLower, A) on a descriptor with p*q > 1 (the reference passes p, q from argv
 * straight into CHAMELEON_Desc_Create, V6:26-27, 44-45) runs the 2D block-cyclic wave DAG of all
 * p*q processes inside the library -- the SAME wave walker as on one GPU (csrc/walker.h; p = q = 1 is its
 * instance without transport calls): paired panels, near / far halves, the POTRF steps pipelined with the
 * owner's TRSM steps.  What moves per wave, point to point, nobody receiving a tile it does not use:
 * L(k,k) with its block inverses from its owner to the other ranks of its process column; the head tile
 * L(k+1,k) ahead of everything else to the owner of (k+1,k+1); every panel tile L(i,k) along process row
 * i mod p (whole contiguous parts) and to the ranks of process column i mod q (tile by tile).
 *
 * A transport is a table of stream-ordered point-to-point operations.  `stream` is a hipStream_t;
 * everything issued between group_begin and group_end progresses together (ncclGroupStart /
 * ncclGroupEnd semantics: sends and receives of one group may be matched in any order).
 * allreduce_max is a blocking host-value reduction (the final LAPACK info).
 * The walker drives TWO channels from two streams of its own: channel 0 carries the small latency-critical
 * messages (diagonal tile, head tile), channel 1 the panel exchange -- so that a communicator which executes
 * its operations in issue order (RCCL does) never queues the next diagonal tile behind a panel. */
typedef struct chol_transport {
  void *ctx;
  int (*group_begin)(void *ctx);
  int (*send)(void *ctx, const void *buf, size_t bytes, int peer, void *stream);
  int (*recv)(void *ctx, void *buf, size_t bytes, int peer, void *stream);
  int (*group_end)(void *ctx);
  int (*allreduce_max)(void *ctx, long long *value);
} chol_transport_t;
/* Install (copy) a transport for both channels; NULL removes it.  Ranks are those of chol_set_rank.
 * chol_set_transport_channel replaces the table of one channel (0 or 1) afterwards. */
int chol_set_transport(const chol_transport_t *t);
int chol_set_transport_channel(int channel, const chol_transport_t *t);
/* The RCCL transport (xGMI inside a node): rank 0 creates the id blob (CHOL_RCCL_ID_BYTES: one ncclUniqueId
 * per channel), the application shares it out of band (MPI, torch.distributed store, a file), every rank calls
 * _init -- which builds one communicator per channel on this process's device and installs the transport
 * (ncclSend / ncclRecv in ncclGroupStart / ncclGroupEnd).  librccl is loaded at this call, not at link
 * time; its version (ncclGetVersion) must match the major version of the rccl.h this library was compiled
 * against and be >= 2.7 (point-to-point operations); _version returns the code (0: not loadable). */
#define CHOL_RCCL_ID_BYTES 256
int chol_transport_rccl_unique_id(void *id256);
int chol_transport_rccl_init(const void *id256, int rank, int nranks);
int chol_transport_rccl_finalize(void);
int chol_transport_rccl_version(void);
/* Host time spent issuing the last distributed factorisation (everything but the final
 * synchronisation), in microseconds per wave, and the number of transport operations it posted. */
int chol_dist_last_stats(double *issue_us_per_wave, long long *sends, long long *recvs, long long *bytes_sent);

/* Collect the lower tiles of a p x q descriptor on rank `root` into a device-resident single-process
 * descriptor of the same order, tile size and type (`dst` is ignored on the other ranks): lets the
 * root verify a distributed factorisation with chol_residual_plgsy.  Collective over the p*q ranks. */
int chol_dist_gather_lower(chol_desc_t *src, chol_desc_t *dst, int root);

/* ---- test hooks of the distributed path (never used by chol_potrf_tile) --- */
/* A transport that moves nothing: sends vanish, receives deliver zeros (stream-ordered).  Lets one process play
 * any single rank of a p x q grid on a one-GPU box and time that rank's schedule with communication taken as
 * free; the numerical result is meaningless. */
int chol_set_transport_null(void);

/* One message of `bytes` bytes (a positive multiple of 8) from this rank to itself on EACH channel of the
 * installed transport, both groups in flight together on the walker's two communication streams,
 * byte-compared.  On a one-rank RCCL communicator: ncclSend / ncclRecv to self inside one group. */
int chol_transport_selftest(int self_rank, size_t bytes);

/* The wave walker over caller-supplied tile kernels, so that the distribution logic (ownership,
 * addressing, matching of sends and receives, buffer reuse, the paired / halves regimes) can run without
 * a GPU under any transport.  Tiles are addressed as in a descriptor: local tile (il, jl) of an
 * lmt x lnt local grid at store + (il + jl*lmt) * B*B elements.  All callbacks are synchronous and are
 * called in a valid sequential order.  mode: 0 every wave plain and split, 1 mixed (panels in pairs while a
 * rank has >= 6 tiles to update), 2 every wave in pairs -- the regimes the product picks by measured speed. */
typedef struct chol_test_engine {
  void *ctx;
  void *store;      /* this rank's tiles */
  size_t esize;     /* bytes per element */
  void *(*alloc)(void *ctx, size_t bytes);                       /* receive buffers */
  int (*potrf)(void *ctx, int k, void *lkk);
  int (*trsm)(void *ctx, int k, const void *lkk);
  int (*update)(void *ctx, int k, int jlo, int jhi, const void *const *bases, const int *firsts, int skip_diag);
  int (*update_diag)(void *ctx, int k, int j, const void *const *bases, const int *firsts);
  int (*info)(void *ctx);
} chol_test_engine_t;
int chol_dist_factorize_with(const chol_test_engine_t *engine, const chol_transport_t *transport, int N, int B,
                             int p, int q, int rank, int mode);

/* A p x q factorisation rehearsed on ONE GPU: p*q ranks as threads of this process, each with its own
 * streams, workspaces and local tiles, the real kernels, and an in-process transport whose sends and receives
 * are stream-ordered device copies (no device synchronisation anywhere: the stream / event ordering and the
 * buffer reuse of the multi-GPU schedule are what is exercised).  The matrix is plgsy(bump, seed) of order N
 * (a multiple of mb, mb a multiple of 128); the factor's lower tiles are gathered into `full`, a
 * device-resident 1 x 1 descriptor of the same order, tile size and type.  *ms: wall time (the ranks share the
 * GPU: a stress figure, not a scaling figure).  Returns the LAPACK info the ranks agreed on. */
int chol_dist_rehearse(int dtype, int N, int mb, int p, int q, double bump, unsigned long long seed,
                       chol_desc_t *full, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* CHOLMI_H */
