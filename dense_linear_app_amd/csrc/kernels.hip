// gfx950 (CDNA4) kernels of libcholmi.so: the four tile operations of the
// reference's Cholesky worker (worker_distrib.cpp:238 POTRF, :323 TRSM, :416 SYRK,
// :511 GEMM) plus the on-device generator / residual of the driver path
// (v6_test.c:46, 72-87).  Written for 64-wide wavefronts and the fp64 / fp32
// 16x16x4 MFMA; no other target is supported.
//
// Common structure ("NT core"): one 256-thread workgroup (4 waves, 2 x 2) owns a
// 128 x 128 block of C and computes  acc = A(128 x K) * B(128 x K)^T  with both
// operands column-major (rows contiguous), staged through LDS in K-slices of 16.
// Production form (nt_kloop_paired + nt_epilogue_paired_impl):
//   - global -> LDS by LDS-DMA (global_load_lds, 16 B per lane): one wave instruction moves a
//     whole 128-row column of a slice into the [k][128] LDS image, no staging registers;
//   - LDS -> MFMA fragments by ds_read_b128 with "paired" rows: lane i of a 16-lane group owns
//     16 B of consecutive rows, so one read feeds 2 (fp64) / 4 (fp32) MFMA tiles; the unpadded
//     128-element row stride makes the four 16-lane groups of a read cover the 64 banks once;
//   - the operands are passed swapped (B-fragment as MFMA "A"), so each lane's accumulator holds
//     consecutive ROWS of C in consecutive lanes: the C read-modify-write is 16 B per lane,
//     contiguous per 16 lanes (col-major C);
//   - double-buffered LDS (64 KiB per workgroup), one barrier per K-slice; 2 workgroups per CU
//     (<= 232 VGPR) hide the C epilogue behind the other workgroup's MFMA stream.
// nt_kloop (register-staged, row stride 144) serves the residual kernel only: it carries the operand
// masks that kernel needs (tril of the diagonal tiles).
#include "cholmi_internal.h"

#include <type_traits>

// The device code is written for gfx950 and nothing else: the fp64 / fp32 16x16x4 MFMA layouts, LDS-DMA, and -- what a
// silent retarget would break without a compile error -- hand-offs that rely on this family's in-order VMEM return and
// `sc1` write-through stores instead of release / acquire fences (the flow form of the tile POTRF, the phase-A flags in
// LDS), and inline-asm sequences whose wait states were counted for it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libcholmi's kernels target gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace cholmi {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct Tr;
template <>
struct Tr<double> {
  using acc_t = d4_t;
  using vec_t = d2_t;
  static constexpr int EPV = 2;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // row of accumulator register `reg` held by `lane` (f64 16x16x4 C/D map)
  static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Tr<float> {
  using acc_t = f4_t;
  using vec_t = f4_t;
  static constexpr int EPV = 4;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

constexpr int BK = 16;
constexpr int LROW = MACRO + 16;

// compile-time loop (indices usable as constants inside the body)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}


template <typename T>
struct alignas(16) Smem {
  T a[2][BK][LROW];
  T b[2][BK][LROW];
};

template <typename T>
using Acc = typename Tr<T>::acc_t[4][4];

template <typename T>
__device__ __forceinline__ void acc_zero(Acc<T> &acc) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
}

// acc += A(128 x K) * B(128 x K)^T.  MASKA / MASKB: treat the operand as the lower
// triangle of a square tile whose row `amask + r` only has columns k <= amask + r
// (used by the residual kernel to read tril(L(j,j))).
template <typename T, bool MASKA, bool MASKB>
__device__ __forceinline__ void nt_kloop(const T *__restrict__ A, int lda, const T *__restrict__ B,
                                         int ldb, int K, Acc<T> &acc, Smem<T> &sm, int amask,
                                         int bmask) {
  using vec_t = typename Tr<T>::vec_t;
  constexpr int EPV = Tr<T>::EPV;
  constexpr int TPC = MACRO / EPV;  // threads per k-column
  constexpr int CPP = 256 / TPC;    // k-columns per pass
  constexpr int NP = BK / CPP;      // passes per slice
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1;
  const int lrow = (t % TPC) * EPV, lcol = t / TPC;
  vec_t ra[NP], rb[NP];

  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int kk = k0 + p * CPP + lcol;
      ra[p] = *reinterpret_cast<const vec_t *>(A + lrow + (size_t)kk * lda);
      rb[p] = *reinterpret_cast<const vec_t *>(B + lrow + (size_t)kk * ldb);
      if (MASKA) {
#pragma unroll
        for (int e = 0; e < EPV; ++e)
          if (kk > amask + lrow + e) ra[p][e] = T(0);
      }
      if (MASKB) {
#pragma unroll
        for (int e = 0; e < EPV; ++e)
          if (kk > bmask + lrow + e) rb[p][e] = T(0);
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      *reinterpret_cast<vec_t *>(&sm.a[buf][p * CPP + lcol][lrow]) = ra[p];
      *reinterpret_cast<vec_t *>(&sm.b[buf][p * CPP + lcol][lrow]) = rb[p];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    T af[4], bf[4];
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int kk = ks * 4 + (lane >> 4);
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = sm.a[cur][kk][wr * 64 + a * 16 + (lane & 15)];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = sm.b[cur][kk][wc * 64 + b * 16 + (lane & 15)];
      // stage the next slice into the other LDS buffer under the last MFMA group, so
      // that nothing but the barrier itself is left at the end of the slice
      if (ks == BK / 4 - 1 && kt + 1 < nk) lstore(cur ^ 1);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = Tr<T>::mfma(bf[b], af[a], acc[a][b]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------
// Cooperative CU hand-over.  The fp64 MFMA stream of a trailing-update wave occupies its
// SIMD's issue for the whole 64 cycles of every MFMA, so a latency-bound kernel that lands
// on the same CU (the single-workgroup diagonal-block factorisation, the handful of
// workgroups of the in-tile POTRF steps) runs 4-6x slower than alone, whatever its wave
// priority -- and it is on the critical path of every wave of the DAG.  Such a "guest"
// workgroup therefore raises a counter in a per-CU table (indexed by the hardware XCC / SE /
// SH / CU ids) for its lifetime; the update waves poll their CU's entry once per K-slice
// (one relaxed agent-scope load, issued ahead of the slice's DMA and consumed at the slice's
// barrier) and sleep while it is non-zero.  One CU of 256 pauses for the ~85 us a guest
// needs; nothing else changes.  Bounded spin: a stale entry can only cost time.
// ------------------------------------------------------------------------------
__device__ __forceinline__ int cu_slot() {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // XCC_ID
  return (int)(((((xcc & 7u) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u) + ((hw >> 8) & 15u));
}
struct GuestOnCu {  // RAII-style bracket used by guest kernels (thread 0 only touches memory)
  int *slot;
  __device__ __forceinline__ explicit GuestOnCu(int *tab) : slot(nullptr) {
    if (tab) {
      slot = tab + cu_slot();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __device__ __forceinline__ void leave() {
    if (slot && threadIdx.x == 0) __hip_atomic_fetch_add(slot, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
__device__ __forceinline__ void yield_to_guest(const int *slot) {
  for (int spin = 0; spin < 4000; ++spin) {  // <= ~2 ms
    if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;
    __builtin_amdgcn_s_sleep(20);
  }
}

// ------------------------------------------------------------------------------
// Device-side dependencies for the panel chain.  A dependency that crosses HIP streams costs 10-15 us of
// wake-up on this runtime (scripts/exp/event_cost.hip: 10.5 us alone, 15 us beside other event traffic,
// against 0.9 us for a kernel that is already resident and polls a word), and a chain-bound wave pays it
// twice: POTRF's last diagonal block -> the panel's last TRSM step, and the last SYRK slice -> the next
// POTRF.  On those two edges the consumer kernel is therefore launched WITHOUT a stream wait and polls a
// counter its producer raises: one lane, relaxed agent-scope loads, then an agent-scope acquire, the
// wait for it and the workgroup barrier before any other lane loads (the consumer recipe of
// MI355X_MICROARCH.md); the producer: every wave's stores drained by the workgroup barrier, an
// agent-scope release by one lane, then the counter.  The polls are bounded (~30 s): a consumer that
// gives up reports INT_MAX through `fail` -- a loud error, never a hang -- and chol_init checks on the
// library's own streams that a polling kernel does not block its producer's stream (streams sharing a
// hardware queue would) before the scheme is used at all.
// ------------------------------------------------------------------------------
__device__ __forceinline__ void sem_wait(const int *sem, int target, int *fail) {
  if (sem) {
    if (threadIdx.x == 0) {
      bool ok = false;
      for (int i = 0; i < (1 << 25); ++i) {  // ~30 s: a safety net only (chol_init has excluded the structural deadlock)
        if (__hip_atomic_load(sem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
          ok = true;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!ok && fail) atomicExch(fail, 0x7ffffffe);  // (INT_MAX - 1: told apart from the diagonal-block kernel's INT_MAX)
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void sem_signal(int *sem) {
  if (sem) {
    __syncthreads();  // (drains every wave's stores: s_waitcnt vmcnt(0) ahead of the barrier)
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(sem, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------
// Hand-offs inside one launch (k_flow_factor / k_flow_rows): the payload is stored write-through (sc1) and read
// by sc1 loads (L1 bypassed, served by L2 / the fabric), the storing wave drains its stores (s_waitcnt vmcnt(0))
// before it -- or, behind a workgroup barrier, one lane of its workgroup -- raises the counter, and the consumer
// polls the counter with sc1 loads and loads only after its poll has matched (MI355X_MICROARCH.md, "Valid forms":
// the form with neither a release nor an acquire fence; ~1 us per hop instead of ~3.5).  CHOLMI_FLOW_FENCES=1 adds
// an agent-scope release before every counter add and an acquire after every poll (diagnostic).
// ------------------------------------------------------------------------------
__device__ __forceinline__ double load_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_pair_sc1(double *p, double a, double b) {  // p 16-byte aligned
  d2_t v = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_pair_sc1(float *p, float a, float b) {  // p 8-byte aligned
  typedef float f2_t __attribute__((ext_vector_type(2)));
  f2_t v = {a, b};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// diagnostic (chol_debug_stamps): one 8-word record {start, end, tag, step} of the calling workgroup
__device__ __forceinline__ void dbg_mark(unsigned long long *dbg, int tag, int step, unsigned long long t0) {
  if (dbg && threadIdx.x == 0) {
    const unsigned long long slot = atomicAdd(dbg, 1ull);
    if (slot < 1000) {
      unsigned long long *p = dbg + 1 + 8 * slot;
      p[0] = t0;
      p[1] = __builtin_amdgcn_s_memrealtime();
      p[2] = (unsigned long long)tag;
      p[3] = (unsigned long long)step;
    }
  }
}

// chol_init's check (see above): the consumer is launched first and polls for at most ~20 ms
__global__ void k_sem_probe_wait(const int *sem, int *result) {
  int ok = 2;
  for (int i = 0; i < 20000; ++i) {
    if (__hip_atomic_load(sem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 1) {
      ok = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  *result = ok;
}
__global__ void k_sem_probe_set(int *sem) { sem_signal(sem); }

// A stream's dependency on a counter, as a launch of its own: ONE wave polls, the kernel behind it in the
// stream starts when it ends.  Grids of more than a few dozen workgroups never poll themselves: resident
// pollers hold LDS and registers, and enough of them can keep the very kernel they wait for (an in-tile
// solve, a diagonal block) from finding a CU -- a deadlock until the poll's bound, seen with two processes
// time-sliced on one GPU.  The pollers that remain are single workgroups (the next POTRF's first step) or
// sit behind their producers in the same launch (k_intile_step).
__global__ void k_sem_gate(const int *sem, int target, int *fail) { sem_wait(sem, target, fail); }
extern int g_poll_max_wgs;
// true: the kernel about to be launched (of `wgs` workgroups) may poll `sem` itself; false: a gate was launched
inline bool poll_in_kernel(hipStream_t s, const int *sem, int target, int *fail, long wgs) {
  if (!sem) return false;
  if (wgs <= g_poll_max_wgs) return true;
  k_sem_gate<<<1, 64, 0, s>>>(sem, target, fail);
  return false;
}

// ------------------------------------------------------------------------------
// "Paired" form of the NT core (trailing update): fragment rows are interleaved so that
// lane i of a 16-lane group owns EPL = 16 B / sizeof(T) CONSECUTIVE rows (fp64: rows 2i,
// 2i+1 of a 32-row group; fp32: rows 4i..4i+3 of the 64-row group).  One ds_read_b128
// then delivers the operand element of EPL MFMA tiles at once (half / quarter the LDS
// instructions), the k-row stride is exactly 128 elements (no padding: the four 16-lane
// groups of a b128 read cover the 64 banks once), and in the epilogue every lane reads
// and writes 16 contiguous bytes of C (256 B contiguous per 16 lanes).
// acc[a][b] <-> rows  64*wr + (a/EPL)*16*EPL + EPL*(lane&15) + a%EPL
//               cols  64*wc + (b/EPL)*16*EPL + EPL*drow(lane,reg) + b%EPL
// ------------------------------------------------------------------------------
template <typename T>
struct alignas(16) SmemP {
  T a[2][BK][MACRO];
  T b[2][BK][MACRO];
};

template <typename T>
__device__ __forceinline__ void nt_kloop_paired(const T *__restrict__ A, int lda,
                                                const T *__restrict__ B, int ldb, int K, Acc<T> &acc,
                                                SmemP<T> &sm, const int *yslot = nullptr) {
  using vec_t = typename Tr<T>::vec_t;
  constexpr int EPV = Tr<T>::EPV;  // elements per 16 bytes == EPL
  constexpr int NG = 4 / EPV;      // 16-lane row groups per 64 rows: fp64 2, fp32 1
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1;
  // LDS-DMA staging (global_load_lds_dwordx4): one wave instruction moves one 1 KiB piece
  // (64 lanes x 16 B) straight into the LDS image at a wave-uniform base -- no staging
  // registers, no ds_write.  Piece q of a slice = bytes [q KiB, (q+1) KiB) of the
  // [BK][128] image; wave w moves pieces w, w+4, ...
  constexpr int PIECES = BK * MACRO * (int)sizeof(T) / 1024;  // per operand per slice
  constexpr int EPP = 1024 / (int)sizeof(T);                  // elements per piece
  auto dma = [&](int buf, int k0) {
#pragma unroll
    for (int q = 0; q < PIECES / 4; ++q) {
      const int piece = q * 4 + w;
      const int e = piece * EPP + lane * EPV;  // element index inside the slice image
      const int kk = e / MACRO, r = e % MACRO;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(A + r + (size_t)(k0 + kk) * lda),
          (__attribute__((address_space(3))) void *)(&sm.a[buf][0][0] + piece * EPP), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(B + r + (size_t)(k0 + kk) * ldb),
          (__attribute__((address_space(3))) void *)(&sm.b[buf][0][0] + piece * EPP), 16, 0, 0);
    }
  };
  vec_t fa[2][NG], fb[2][NG];
  const int arow = wr * 64 + EPV * (lane & 15), brow = wc * 64 + EPV * (lane & 15), kq = lane >> 4;
  auto fread = [&](int set, int cur, int ks) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      fa[set][g] = *reinterpret_cast<const vec_t *>(&sm.a[cur][ks * 4 + kq][arow + g * 16 * EPV]);
      fb[set][g] = *reinterpret_cast<const vec_t *>(&sm.b[cur][ks * 4 + kq][brow + g * 16 * EPV]);
    }
  };

  dma(0, 0);
  __syncthreads();
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    int guest = 0;  // requested ahead of the DMA, looked at after the slice's MFMAs
    if (yslot) guest = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fread(0, cur, 0);
    if (kt + 1 < nk) dma(cur ^ 1, (kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      // The group's own fragments were requested a whole group (1024 cycles) ago, so the
      // wait in front of its first MFMA is free; the NEXT group's fragment reads are issued
      // only after that first MFMA -- issued before it, the compiler's lgkmcnt(0) would also
      // wait for them and expose a full LDS round trip in front of every group.
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = Tr<T>::mfma(fb[ks & 1][0][0], fa[ks & 1][0][0], acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < BK / 4) fread((ks + 1) & 1, cur, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (a + b > 0)
            acc[a][b] = Tr<T>::mfma(fb[ks & 1][b / EPV][b % EPV], fa[ks & 1][a / EPV][a % EPV], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (yslot && __builtin_amdgcn_readfirstlane(guest) != 0) yield_to_guest(yslot);
    __syncthreads();
  }
}

template <typename T, bool LOWER>
__device__ __forceinline__ void nt_epilogue_paired_impl(T *__restrict__ C, int ldc, Acc<T> &acc,
                                                        T alpha, T beta) {
  using vec_t = typename Tr<T>::vec_t;
  constexpr int EPV = Tr<T>::EPV, NG = 4 / EPV;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1;
  const int m0 = wr * 64 + EPV * (lane & 15);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    vec_t cv[4][NG];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wc * 64 + (b / EPV) * 16 * EPV + EPV * Tr<T>::drow(lane, r) + b % EPV;
      const T *col = C + (size_t)n * ldc + m0;
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (beta != T(0)) cv[r][g] = *reinterpret_cast<const vec_t *>(col + g * 16 * EPV);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const T v = alpha * acc[g * EPV + e][b][r];
          cv[r][g][e] = (beta != T(0)) ? v + beta * cv[r][g][e] : v;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wc * 64 + (b / EPV) * 16 * EPV + EPV * Tr<T>::drow(lane, r) + b % EPV;
      T *col = C + (size_t)n * ldc + m0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int m = m0 + g * 16 * EPV;
        if (!LOWER || m >= n) {  // all EPV rows at or below the diagonal
          *reinterpret_cast<vec_t *>(col + g * 16 * EPV) = cv[r][g];
        } else {
#pragma unroll
          for (int e = 0; e < EPV; ++e)
            if (m + e >= n) col[g * 16 * EPV + e] = cv[r][g][e];
        }
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void nt_epilogue_paired(T *__restrict__ C, int ldc, Acc<T> &acc, T alpha,
                                                   T beta, bool lower) {
  if (lower)
    nt_epilogue_paired_impl<T, true>(C, ldc, acc, alpha, beta);
  else
    nt_epilogue_paired_impl<T, false>(C, ldc, acc, alpha, beta);
}

template <typename T>
__device__ __forceinline__ const T *panel_tile(const PanelRef &pan, int i, long bsiz) {
  const int p = i % pan.P;
  return reinterpret_cast<const T *>(pan.base[p]) + (long)(i / pan.P - pan.first[p]) * bsiz;
}

// ------------------------------------------------------------------------------
// Trailing update of one wave (GEMM + SYRK tasks of C2:540-560 fused in one launch):
//   C(i,j) -= L(i,k) L(j,k)^T for every (i,j) in list[0..ntiles), diagonal tiles
//   lower-triangle only (dsyrk Lower semantics, W2:416: strict upper untouched).
// blockIdx -> (tile, macro block) is XCD-aware: blocks with equal blockIdx % 8
// share an XCD (and its L2); all macro blocks of one tile, and G consecutive tiles
// of the list, are dealt to the same XCD so the two panel tiles they stream are
// fetched from HBM once per XCD.
// ------------------------------------------------------------------------------
// blockIdx -> (tile, 128x128 block) of a trailing-update launch, XCD-aware (blockIdx & 7 is the XCD
// the workgroup lands on).  The launch has two segments.  A: the na off-diagonal tiles
// list[0 .. na), MT = nbm^2 blocks each; B: the nb diagonal tiles list[offb .. offb+nb), only their
// MTd = nbm(nbm+1)/2 blocks on or below the diagonal, packed -- no workgroup exits early, and the
// diagonal tiles come last: a tile whose blocks leave holes in the middle of a launch splits the
// 64-workgroup cohort of its XCD into phases for the rest of the launch, which costs the L2 its
// operand reuse (measured: HBM-side traffic 1.96x -> 1.23x the algorithmic bytes).
// Per segment the blocks are numbered tile by tile and dealt to the XCDs in UNITS of `unit`
// consecutive blocks, round-robin.  unit = 64 for a big launch: one unit = the 64 workgroup slots
// of an XCD = one whole tile at mb = 1024 (or 64 / MT consecutive tiles of a smaller mb), so that
// the blocks sharing operands run together behind one L2.  A launch of a few tiles (late waves;
// every launch of a rank when the matrix is spread over 8 GPUs) uses smaller units, down to 8
// blocks (until the launch has at least g_min_units units), so that all eight XCDs get the same
// share instead of whole tiles piling up on some.
// blocks_a = blocks of segment A (a multiple of 8 units).
struct BlockMap {
  int2 ij;
  int mi, mj;
};
__device__ __forceinline__ bool map_update_block(const int2 *__restrict__ list, int na, int offb, int nb,
                                                 int nbm, int blocks_a, int unit, BlockMap &out, int b = -1) {
  if (b < 0) b = blockIdx.x;
  if (na == 0 && nb == 1) {
    // one diagonal tile on its own (the SYRK that releases the next POTRF): nothing to share
    // through an L2, so its blocks go round-robin over all XCDs, one workgroup per CU
    int macro = b, mj = 0;
    while (macro >= nbm - mj) {
      macro -= nbm - mj;
      ++mj;
    }
    out.ij = list[offb];
    out.mj = mj;
    out.mi = mj + macro;
    return true;
  }
  if (b < blocks_a) {
    const int MT = nbm * nbm;
    const int x = b & 7, s = b >> 3;
    const int lin = ((s / unit) * 8 + x) * unit + s % unit;
    if (lin >= na * MT) return false;
    const int t = lin / MT, macro = lin - t * MT;
    out.ij = list[t];
    out.mi = macro % nbm;
    out.mj = macro / nbm;
    return true;
  }
  b -= blocks_a;
  const int MTd = nbm * (nbm + 1) / 2;
  // segment B starts on the XCD after the one that took the last unit of segment A
  const int ua = (na * nbm * nbm + unit - 1) / unit;
  const int x = ((b & 7) + 8 - ua % 8) & 7, s = b >> 3;
  const int lin = ((s / unit) * 8 + x) * unit + s % unit;
  if (lin >= nb * MTd) return false;
  const int t = lin / MTd;
  int macro = lin - t * MTd;
  out.ij = list[offb + t];
  int mj = 0;  // lower triangle, column by column: column mj holds nbm - mj blocks
  while (macro >= nbm - mj) {
    macro -= nbm - mj;
    ++mj;
  }
  out.mj = mj;
  out.mi = mj + macro;
  return true;
}

// ------------------------------------------------------------------------------
// The same trailing update with EIGHT waves per workgroup (fp64): 2 x 4 waves, each 64 rows x 32
// columns of the 128 x 128 block, two workgroups per CU = four waves per SIMD instead of two.  Every
// wave still stalls once per K-slice (fragment read after the barrier, the barrier itself, the DMA
// issue); with four independent MFMA streams per SIMD the matrix pipe finds a ready wave more often.
// Costs: 3 fragment reads per 8 MFMAs instead of 4 per 16, and <= 128 VGPRs per wave.
// ------------------------------------------------------------------------------
// MODE bit 0: the next slice's DMA is issued behind the first MFMAs of k-groups 0 and 1 instead of
// in one burst between the barrier and the slice's first MFMA; bit 1: static priority 1 for waves 4-7
template <typename T, int MODE>
__device__ __forceinline__ void nt_kloop_w8(const T *__restrict__ A, int lda, const T *__restrict__ B, int ldb,
                                            int K, typename Tr<T>::acc_t (&acc)[4][2], SmemP<T> &sm,
                                            const int *yslot) {
  static_assert(sizeof(T) == 8, "fp64 only");
  using vec_t = typename Tr<T>::vec_t;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1, i = lane & 15, q = lane >> 4;
  constexpr int PIECES = BK * MACRO * (int)sizeof(T) / 1024, EPP = 1024 / (int)sizeof(T);
  auto dma_piece = [&](int buf, int k0, int p) {
    const int piece = p * 8 + w;
    const int e = piece * EPP + lane * 2;
    const int kk = e / MACRO, r = e % MACRO;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(A + r + (size_t)(k0 + kk) * lda),
        (__attribute__((address_space(3))) void *)(&sm.a[buf][0][0] + piece * EPP), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(B + r + (size_t)(k0 + kk) * ldb),
        (__attribute__((address_space(3))) void *)(&sm.b[buf][0][0] + piece * EPP), 16, 0, 0);
  };
  auto dma = [&](int buf, int k0) {
#pragma unroll
    for (int p = 0; p < PIECES / 8; ++p) dma_piece(buf, k0, p);
  };
  static_assert(PIECES / 8 == 2, "two pieces per operand per wave and slice");
  vec_t fa[2][2], fb[2];
  const int arow = wr * 64 + 2 * i, brow = wc * 32 + 2 * i;
  if (MODE & 2) {
    if (w >= 4) __builtin_amdgcn_s_setprio(1);
  }
  auto fread = [&](int set, int cur, int ks) {
#pragma unroll
    for (int g = 0; g < 2; ++g) fa[set][g] = *reinterpret_cast<const vec_t *>(&sm.a[cur][ks * 4 + q][arow + 32 * g]);
    fb[set] = *reinterpret_cast<const vec_t *>(&sm.b[cur][ks * 4 + q][brow]);
  };
  dma(0, 0);
  __syncthreads();
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    int guest = 0;
    if (yslot) guest = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fread(0, cur, 0);
    if (!(MODE & 1) && kt + 1 < nk) dma(cur ^ 1, (kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = Tr<T>::mfma(fb[ks & 1][0], fa[ks & 1][0][0], acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < BK / 4) fread((ks + 1) & 1, cur, ks + 1);
      if ((MODE & 1) && ks < 2 && kt + 1 < nk) dma_piece(cur ^ 1, (kt + 1) * BK, ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (a + b > 0) acc[a][b] = Tr<T>::mfma(fb[ks & 1][b], fa[ks & 1][a >> 1][a & 1], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (yslot && __builtin_amdgcn_readfirstlane(guest) != 0) yield_to_guest(yslot);
    __syncthreads();
  }
}

// Cout = Cin - acc for one 128 x 128 block (eight waves, the accumulator layout of nt_kloop_w8): lane (i, q) holds rows
// 2i, 2i+1 (+32g) of column 32 wc + 2 (q + 4r) + b.  lower: the block lies on the diagonal of a diagonal tile -- only
// its entries on or below the diagonal are updated (dsyrk Lower, W2:416); OOP (Cout != Cin, the task path's private
// copy W2:212-213 made by writing elsewhere): the entries above it are copied, in place they are left alone.
template <typename T, bool OOP>
__device__ __forceinline__ void w8_epilogue(const T *Cin, T *Cout, int ld,
                                            typename Tr<T>::acc_t (&acc)[4][2], bool lower) {
  using vec_t = typename Tr<T>::vec_t;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1, i = lane & 15, q = lane >> 4;
  const int m0 = wr * 64 + 2 * i;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    vec_t cv[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 2; ++g)
        cv[r][g] = *reinterpret_cast<const vec_t *>(Cin + (long)(wc * 32 + 2 * (q + 4 * r) + b) * ld + m0 + 32 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wc * 32 + 2 * (q + 4 * r) + b;
      T *col = Cout + (long)n * ld;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int m = m0 + 32 * g;
        vec_t v = cv[r][g];
        if (OOP && lower && m < n) {  // (m even, rows m and m + 1: row m is above the diagonal, row m + 1 maybe not)
          if (m + 1 >= n) v[1] -= acc[2 * g + 1][b][r];
          *reinterpret_cast<vec_t *>(col + m) = v;
          continue;
        }
        v[0] -= acc[2 * g][b][r];
        v[1] -= acc[2 * g + 1][b][r];
        if (!lower || m >= n) {
          *reinterpret_cast<vec_t *>(col + m) = v;
        } else {
          if (m + 1 >= n) col[m + 1] = v[1];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, int MODE>
__device__ __forceinline__ void trail_update_w8_block(const LocalMat &C, const int2 *__restrict__ list, int na, int offb,
                                                      int nb, int blocks_a, const PanelRef &pan, int nbm, int unit,
                                                      const int *ytab, const PanelRef &pan2, int npan, SmemP<T> &sm, int b) {
  using vec_t = typename Tr<T>::vec_t;
  BlockMap bm;
  if (!map_update_block(list, na, offb, nb, nbm, blocks_a, unit, bm, b)) return;
  const int2 ij = bm.ij;
  const int mi = bm.mi, mj = bm.mj;
  const bool lower = (ij.x == ij.y) && mi == mj;
  T *Cp = reinterpret_cast<T *>(C.base) + ((long)(ij.x / C.P) + (long)(ij.y / C.Q) * C.lmt) * C.bsiz +
          mi * MACRO + (long)mj * MACRO * C.mb;
  typename Tr<T>::acc_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
  const int *yslot = ytab ? ytab + cu_slot() : nullptr;
  nt_kloop_w8<T, MODE>(panel_tile<T>(pan, ij.x, C.bsiz) + mi * MACRO, C.mb, panel_tile<T>(pan, ij.y, C.bsiz) + mj * MACRO,
                 C.mb, C.mb, acc, sm, yslot);
  if (npan > 1)
    nt_kloop_w8<T, MODE>(panel_tile<T>(pan2, ij.x, C.bsiz) + mi * MACRO, C.mb,
                   panel_tile<T>(pan2, ij.y, C.bsiz) + mj * MACRO, C.mb, C.mb, acc, sm, yslot);
  w8_epilogue<T, false>(Cp, Cp, C.mb, acc, lower);
}

template <typename T, int MODE>
__global__ __launch_bounds__(512, 4) void k_trail_update_w8(LocalMat C, const int2 *__restrict__ list, int na,
                                                             int offb, int nb, int blocks_a, PanelRef pan, int nbm,
                                                             int unit, const int *ytab, PanelRef pan2, int npan) {
  __shared__ SmemP<T> sm;
  trail_update_w8_block<T, MODE>(C, list, na, offb, nb, blocks_a, pan, nbm, unit, ytab, pan2, npan, sm, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------
// fp32 trailing update on eight waves with K-slices of 32.  The fp32 16x16x4 MFMA issues every 32
// cycles (twice the fp64 rate), so with 16-deep slices a wave meets a barrier after 2048 cycles of
// matrix work and the per-slice stalls weigh twice as much as in fp64 (81 % of peak against 86 %).
// Slices of 32 (2 x 32 KiB LDS buffers, two workgroups per CU) restore the fp64 kernel's 4096 matrix
// cycles per wave and barrier at the same four waves per SIMD.  Wave tile 64 rows x 32 columns: the A
// fragment is one 16-byte read (rows 4i..4i+3 of the 64-row group), the B fragment one 8-byte read
// (columns 2i, 2i+1); accumulator register r of tile (a, b) in lane (i, q) is element
// (row 64 wr + 4i + a, column 32 wc + 2 (4q + r) + b).
// ------------------------------------------------------------------------------
constexpr int BKF = 32;
struct alignas(16) SmemF {
  float a[2][BKF][MACRO];
  float b[2][BKF][MACRO];
};
typedef float f2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void nt_kloop_w8f(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                             int ldb, int K, f4_t (&acc)[4][2], SmemF &sm, const int *yslot) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1, i = lane & 15, q = lane >> 4;
  constexpr int PIECES = BKF * MACRO * 4 / 1024, EPP = 256;  // 16 one-KiB pieces per operand and slice
  auto dma = [&](int buf, int k0) {
#pragma unroll
    for (int p = 0; p < PIECES / 8; ++p) {
      const int piece = p * 8 + w;
      const int e = piece * EPP + lane * 4;
      const int kk = e / MACRO, r = e % MACRO;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(A + r + (size_t)(k0 + kk) * lda),
          (__attribute__((address_space(3))) void *)(&sm.a[buf][0][0] + piece * EPP), 16, 0, 0);
      // B image: odd k-rows rotated by 32 columns (on the SOURCE address -- the DMA writes LDS linearly):
      // the 8-byte fragment reads of the two k-rows of a 32-lane group then fall on opposite bank halves
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(B + ((r + 32 * (kk & 1)) & (MACRO - 1)) + (size_t)(k0 + kk) * ldb),
          (__attribute__((address_space(3))) void *)(&sm.b[buf][0][0] + piece * EPP), 16, 0, 0);
    }
  };
  f4_t fa[2];
  f2_t fb[2];
  const int arow = wr * 64 + 4 * i, brow = (wc * 32 + 2 * i - 32 * (q & 1)) & (MACRO - 1);  // (4 ks + q) & 1 == q & 1
  auto fread = [&](int set, int cur, int ks) {
    fa[set] = *reinterpret_cast<const f4_t *>(&sm.a[cur][ks * 4 + q][arow]);
    fb[set] = *reinterpret_cast<const f2_t *>(&sm.b[cur][ks * 4 + q][brow]);
  };
  dma(0, 0);
  __syncthreads();
  const int nk = K / BKF;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    int guest = 0;
    if (yslot) guest = __hip_atomic_load(yslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fread(0, cur, 0);
    if (kt + 1 < nk) dma(cur ^ 1, (kt + 1) * BKF);
#pragma unroll
    for (int ks = 0; ks < BKF / 4; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = Tr<float>::mfma(fb[ks & 1][0], fa[ks & 1][0], acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < BKF / 4) fread((ks + 1) & 1, cur, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (a + b > 0) acc[a][b] = Tr<float>::mfma(fb[ks & 1][b], fa[ks & 1][a], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (yslot && __builtin_amdgcn_readfirstlane(guest) != 0) yield_to_guest(yslot);
    __syncthreads();
  }
}

// Cout = Cin - acc, fp32 (see w8_epilogue): lane (i, q) holds rows 4i .. 4i+3 of column 32 wc + 2 (4q + r) + b
template <bool OOP>
__device__ __forceinline__ void w8f_epilogue(const float *Cin, float *Cout, int ld, f4_t (&acc)[4][2], bool lower) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1, i = lane & 15, q = lane >> 4;
  const int m0 = wr * 64 + 4 * i;
  f4_t cv[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      cv[b][r] = *reinterpret_cast<const f4_t *>(Cin + (long)(wc * 32 + 2 * (4 * q + r) + b) * ld + m0);
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wc * 32 + 2 * (4 * q + r) + b;
      float *col = Cout + (long)n * ld;
      f4_t v = cv[b][r];
      if (OOP && lower && m0 < n) {  // some of the four rows lie above the diagonal: those are copied
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (m0 + a >= n) v[a] -= acc[a][b][r];
        *reinterpret_cast<f4_t *>(col + m0) = v;
        continue;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) v[a] -= acc[a][b][r];
      if (!lower || m0 >= n) {
        *reinterpret_cast<f4_t *>(col + m0) = v;
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (m0 + a >= n) col[m0 + a] = v[a];
      }
    }
}

__global__ __launch_bounds__(512, 4) void k_trail_update_w8f(LocalMat C, const int2 *__restrict__ list, int na,
                                                              int offb, int nb, int blocks_a, PanelRef pan, int nbm,
                                                              int unit, const int *ytab, PanelRef pan2, int npan) {
  __shared__ SmemF sm;
  BlockMap bm;
  if (!map_update_block(list, na, offb, nb, nbm, blocks_a, unit, bm)) return;
  const int2 ij = bm.ij;
  const int mi = bm.mi, mj = bm.mj;
  const bool lower = (ij.x == ij.y) && mi == mj;
  float *Cp = reinterpret_cast<float *>(C.base) + ((long)(ij.x / C.P) + (long)(ij.y / C.Q) * C.lmt) * C.bsiz +
              mi * MACRO + (long)mj * MACRO * C.mb;
  f4_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  const int *yslot = ytab ? ytab + cu_slot() : nullptr;
  nt_kloop_w8f(panel_tile<float>(pan, ij.x, C.bsiz) + mi * MACRO, C.mb,
               panel_tile<float>(pan, ij.y, C.bsiz) + mj * MACRO, C.mb, C.mb, acc, sm, yslot);
  if (npan > 1)
    nt_kloop_w8f(panel_tile<float>(pan2, ij.x, C.bsiz) + mi * MACRO, C.mb,
                 panel_tile<float>(pan2, ij.y, C.bsiz) + mj * MACRO, C.mb, C.mb, acc, sm, yslot);
  w8f_epilogue<false>(Cp, Cp, C.mb, acc, lower);
}

// ------------------------------------------------------------------------------
// The task path's updates (chol_tile_batch: the SYRK / GEMM tasks of a wave, W2:416, 511) on the same eight-wave cores:
// task t reads its operands through device arrays of tile pointers and writes  cout[t] = cin[t] - a[t] b[t]^T  OUT OF
// PLACE -- the private copy every task makes of the tile it updates (W2:212-213) is this write, there is no copy pass.
// b[t] == nullptr marks a SYRK task (b = a, Lower): its blocks above the diagonal copy cin to cout, its diagonal blocks
// update on or below the diagonal and copy above it.  blockIdx -> (task, block) as in map_update_block's segment A: the
// MT = nbm^2 blocks of a task and `unit` consecutive blocks of the list stay on one XCD (the operands of consecutive
// tasks of a wave repeat: same L(i,k) along a row of the trailing matrix).
// ------------------------------------------------------------------------------
__device__ __forceinline__ bool map_ptr_block(int n, int nbm, int unit, int &t, int &mi, int &mj) {
  const int b = blockIdx.x, x = b & 7, sidx = b >> 3, MT = nbm * nbm;
  const long lin = ((long)(sidx / unit) * 8 + x) * unit + sidx % unit;
  if (lin >= (long)n * MT) return false;
  t = (int)(lin / MT);
  const int macro = (int)(lin - (long)t * MT);
  mi = macro % nbm;
  mj = macro / nbm;
  return true;
}
// cout block <- cin block (128 x 128, 512 threads): the blocks of a SYRK task above the diagonal
template <typename T>
__device__ __forceinline__ void copy_block_512(const T *cin, T *cout, int ld) {
  constexpr int EPV = 16 / (int)sizeof(T), VPC = MACRO / EPV;  // 16-byte vectors per column
  for (int e = threadIdx.x; e < MACRO * VPC; e += 512) {
    const int c = e / VPC, r = (e % VPC) * EPV;
    *reinterpret_cast<uint4 *>(cout + r + (long)c * ld) = *reinterpret_cast<const uint4 *>(cin + r + (long)c * ld);
  }
}
template <typename T, int MODE>
__global__ __launch_bounds__(512, 4) void k_update_ptrs_w8(const T *const *__restrict__ cin, const T *const *__restrict__ ap,
                                                            const T *const *__restrict__ bp, T *const *__restrict__ cout,
                                                            int n, int mb, int nbm, int unit, const int *ytab) {
  __shared__ SmemP<T> sm;
  int t, mi, mj;
  if (!map_ptr_block(n, nbm, unit, t, mi, mj)) return;
  const T *A = ap[t], *B = bp[t];
  const bool syrk = B == nullptr;
  const long off = mi * MACRO + (long)mj * MACRO * mb;
  const T *Ci = cin[t] + off;
  T *Co = cout[t] + off;
  if (syrk && mi < mj) {
    copy_block_512<T>(Ci, Co, mb);
    return;
  }
  if (syrk) B = A;
  typename Tr<T>::acc_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
  const int *yslot = ytab ? ytab + cu_slot() : nullptr;
  nt_kloop_w8<T, MODE>(A + mi * MACRO, mb, B + mj * MACRO, mb, mb, acc, sm, yslot);
  w8_epilogue<T, true>(Ci, Co, mb, acc, syrk && mi == mj);
}
__global__ __launch_bounds__(512, 4) void k_update_ptrs_w8f(const float *const *__restrict__ cin, const float *const *__restrict__ ap,
                                                             const float *const *__restrict__ bp, float *const *__restrict__ cout,
                                                             int n, int mb, int nbm, int unit, const int *ytab) {
  __shared__ SmemF sm;
  int t, mi, mj;
  if (!map_ptr_block(n, nbm, unit, t, mi, mj)) return;
  const float *A = ap[t], *B = bp[t];
  const bool syrk = B == nullptr;
  const long off = mi * MACRO + (long)mj * MACRO * mb;
  const float *Ci = cin[t] + off;
  float *Co = cout[t] + off;
  if (syrk && mi < mj) {
    copy_block_512<float>(Ci, Co, mb);
    return;
  }
  if (syrk) B = A;
  f4_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  const int *yslot = ytab ? ytab + cu_slot() : nullptr;
  nt_kloop_w8f(A + mi * MACRO, mb, B + mj * MACRO, mb, mb, acc, sm, yslot);
  w8f_epilogue<true>(Ci, Co, mb, acc, syrk && mi == mj);
}

// X[:, s] := alpha * A[:, s] * Winv_s^T, in place, for row blocks r >= r0 of `ntiles`
// contiguous tiles.  (TRSM by multiplication with the inverted 128x128 diagonal block.)
template <typename T>
__global__ __launch_bounds__(256, 2) void k_panel_solve(T *tiles, long bsiz, int mb, int nbm, int r0,
                                                        int s, const T *__restrict__ winv, T alpha,
                                                        int *ytab, const int *wait_sem = nullptr,
                                                        int wait_target = 0, int *fail = nullptr,
                                                        int *head_sem = nullptr) {
  __shared__ SmemP<T> sm;
  sem_wait(wait_sem, wait_target, fail);
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(2);  // panel chain: ahead of co-resident trailing-update waves
  const int nr = nbm - r0;
  const int tix = blockIdx.x / nr, r = r0 + blockIdx.x % nr;
  T *Ap = tiles + (long)tix * bsiz + r * MACRO + (long)s * MACRO * mb;
  const T *Bp = winv + (long)s * MACRO * MACRO;
  Acc<T> acc;
  acc_zero<T>(acc);
  nt_kloop_paired<T>(Ap, mb, Bp, MACRO, MACRO, acc, sm);
  nt_epilogue_paired<T>(Ap, mb, acc, alpha, T(0), false);
  guest.leave();
  if (tix == 0) sem_signal(head_sem);  // (the first tile's workgroups: nr counts)
}

// A[:, c] := beta*A[:, c] - X[:, s] * L[c, s]^T for c > s (right-looking TRSM step)
template <typename T>
__global__ __launch_bounds__(256, 2) void k_panel_update(T *tiles, long bsiz, int mb, int nbm, int s,
                                                         const T *__restrict__ lkk, T beta, int *ytab,
                                                         const int *wait_sem = nullptr, int wait_target = 0,
                                                         int *fail = nullptr) {
  __shared__ SmemP<T> sm;
  sem_wait(wait_sem, wait_target, fail);
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(2);  // panel chain: ahead of co-resident trailing-update waves
  const int nc = nbm - 1 - s;
  int b = blockIdx.x;
  const int c = s + 1 + b % nc;
  b /= nc;
  const int r = b % nbm, tix = b / nbm;
  T *tile = tiles + (long)tix * bsiz;
  const T *Ap = tile + r * MACRO + (long)s * MACRO * mb;
  const T *Bp = lkk + c * MACRO + (long)s * MACRO * mb;
  T *Cp = tile + r * MACRO + (long)c * MACRO * mb;
  Acc<T> acc;
  acc_zero<T>(acc);
  nt_kloop_paired<T>(Ap, mb, Bp, mb, MACRO, acc, sm);
  nt_epilogue_paired<T>(Cp, mb, acc, T(-1), beta, false);
  guest.leave();
}

// ------------------------------------------------------------------------------
// The in-tile steps of the tile POTRF in small-block form.  They are K = 128 products of a
// handful of blocks on the critical chain (diagonal block -> solve -> update -> next diagonal
// block); in the 128 x 128 NT core one such block is 8 K-slices of a DMA pipeline that never
// fills: 19-22 us, 3x its MFMA time.  Here a workgroup takes a quarter of the work (solve: a
// 32-row slab of the block row, all 128 columns, so that it can run in place; update: a 64 x 64
// block), K in four phases of 32 staged global -> registers -> LDS with the next phase's loads
// in flight under the current phase's MFMAs, one wave = 32 x 32 of the output = 2 x 2 MFMA
// tiles.  LDS image [k][rows + 16]: the row stride puts the four 16-lane k-groups of a fragment
// read on alternating bank halves (conflict-free for 8-byte reads).
// ------------------------------------------------------------------------------
constexpr int SK = 32;
template <typename T, int ROWS>
struct SmallImg {
  T v[SK][ROWS + 16];
};
template <typename T, int ROWS>
__device__ __forceinline__ void small_gload(const T *__restrict__ P, int ld, int k0, T (&r)[ROWS * SK / 256]) {
#pragma unroll
  for (int q = 0; q < ROWS * SK / 256; ++q) {
    const int e = threadIdx.x + 256 * q;
    r[q] = P[(e % ROWS) + (long)(k0 + e / ROWS) * ld];
  }
}
template <typename T, int ROWS>
__device__ __forceinline__ void small_lstore(SmallImg<T, ROWS> &img, const T (&r)[ROWS * SK / 256]) {
#pragma unroll
  for (int q = 0; q < ROWS * SK / 256; ++q) {
    const int e = threadIdx.x + 256 * q;
    img.v[e / ROWS][e % ROWS] = r[q];
  }
}
// acc[a][b] += A(i0 + 16a .. , k) B(j0 + 16b .., k)^T over the SK k's of the images; operands
// swapped so that accumulator register r of lane l is (row i0 + 16a + (l & 15), col j0 + 16b + drow(l, r))
template <typename T, int RA, int RB>
__device__ __forceinline__ void small_mma(const SmallImg<T, RA> &ia, const SmallImg<T, RB> &ib, int i0, int j0,
                                          typename Tr<T>::acc_t (&acc)[2][2]) {
  const int lane = threadIdx.x & 63, lo = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < SK / 4; ++ks) {
    const int k = 4 * ks + kq;
    T af[2], bf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = ia.v[k][i0 + 16 * a + lo];
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[b] = ib.v[k][j0 + 16 * b + lo];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = Tr<T>::mfma(bf[b], af[a], acc[a][b]);
  }
}

// X[slab, 0..127] = alpha * A[slab, 0..127] * Winv_s^T, in place: one workgroup per 32-row slab of
// the block rows r0.. of each of the tiles at tiles + q*bsiz (it reads only its own rows, all of
// them before it writes).  In-tile step: one tile, r0 = s+1; panel TRSM step: all panel tiles, r0 = 0.
template <typename T>
__global__ __launch_bounds__(256, 2) void k_solve_small(T *tiles, long bsiz, int mb, int nbm, int r0, int s,
                                                        const T *__restrict__ winv, T alpha, int *ytab,
                                                        const int *wait_sem = nullptr, int wait_target = 0,
                                                        int *fail = nullptr, int *head_sem = nullptr) {
  __shared__ SmallImg<T, 32> ia;
  __shared__ SmallImg<T, MACRO> ib;
  sem_wait(wait_sem, wait_target, fail);
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(2);
  const int per_tile = 4 * (nbm - r0);
  const int tix = blockIdx.x / per_tile, slab = blockIdx.x % per_tile;
  T *Ap = tiles + (long)tix * bsiz + (long)r0 * MACRO + 32 * slab + (long)s * MACRO * mb;
  const T *Bp = winv + (long)s * MACRO * MACRO;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lo = lane & 15;
  T ra[32 * SK / 256], rb[MACRO * SK / 256];
  typename Tr<T>::acc_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
  small_gload<T, 32>(Ap, mb, 0, ra);
  small_gload<T, MACRO>(Bp, MACRO, 0, rb);
  for (int ph = 0; ph < MACRO / SK; ++ph) {
    small_lstore<T, 32>(ia, ra);
    small_lstore<T, MACRO>(ib, rb);
    __syncthreads();
    if (ph + 1 < MACRO / SK) {
      small_gload<T, 32>(Ap, mb, (ph + 1) * SK, ra);
      small_gload<T, MACRO>(Bp, MACRO, (ph + 1) * SK, rb);
    }
    small_mma<T, 32, MACRO>(ia, ib, 0, 32 * w, acc);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Ap[16 * a + lo + (long)(32 * w + 16 * b + Tr<T>::drow(lane, r)) * mb] = alpha * acc[a][b][r];
  guest.leave();
  if (tix == 0) sem_signal(head_sem);  // (the first tile's workgroups: per_tile counts)
}

// C(r64, c64) -= A(r64, :) B(c64, :)^T over K columns, for the 64 x 64 blocks on or below the
// diagonal of a square C (diagonal blocks write i >= j only), or, full != 0, for every block of a
// rectangular C.  Uses, all on the critical chain: the panel TRSM step A[:, c] -= X[:, s] L(c, s)^T
// (full; blockIdx.z = panel tile),
// the in-tile trailing update (C = the trailing part of the tile, A = B = block column s, K = 128)
// and the single SYRK on the next diagonal tile, C(k+1,k+1) -= L(k+1,k) L(k+1,k)^T (K = mb), which
// releases POTRF(k+1).
template <typename T>
__global__ __launch_bounds__(256, 2) void k_small_update(T *C, long ldc, const T *__restrict__ A,
                                                        const T *__restrict__ B, long ldab, int K, int *ytab,
                                                        int full, long zc, long za, int *signal_sem = nullptr,
                                                        const int *wait_sem = nullptr, int wait_target = 0,
                                                        int *fail = nullptr) {
  __shared__ SmallImg<T, 64> ia, ib;
  const int r64 = blockIdx.x, c64 = blockIdx.y;
  if (!full && c64 > r64) return;
  C += blockIdx.z * zc;  // (panel TRSM step: one z per panel tile, B is the diagonal tile for all)
  A += blockIdx.z * za;
  const T *Ap = A + 64 * r64;
  const T *Bp = B + 64 * c64;
  T *Cp = C + 64 * r64 + 64 * c64 * ldc;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lo = lane & 15;
  const int i0 = 32 * (w & 1), j0 = 32 * (w >> 1);
  T ra[64 * SK / 256], rb[64 * SK / 256];
  typename Tr<T>::acc_t acc[2][2];
  // C up front (nobody else writes this block meanwhile): its load latency -- a cold miss, the block was last
  // written a step or a wave ago -- is then hidden behind the polls and the K loop instead of ending the kernel
  T cv[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[a][b][r] = T(0);
        cv[a][b][r] = Cp[(i0 + 16 * a + lo) + (long)(j0 + 16 * b + Tr<T>::drow(lane, r)) * ldc];
      }
  // (what is polled for is the producer of A / B; the block of C was last written by an earlier launch of
  // this stream, or before the launch that raised an earlier counter of the chain)
  sem_wait(wait_sem, wait_target, fail);
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(2);
  small_gload<T, 64>(Ap, (int)ldab, 0, ra);
  small_gload<T, 64>(Bp, (int)ldab, 0, rb);
  const int nph = K / SK;
  for (int ph = 0; ph < nph; ++ph) {
    small_lstore<T, 64>(ia, ra);
    small_lstore<T, 64>(ib, rb);
    __syncthreads();
    if (ph + 1 < nph) {
      small_gload<T, 64>(Ap, (int)ldab, (ph + 1) * SK, ra);
      small_gload<T, 64>(Bp, (int)ldab, (ph + 1) * SK, rb);
    }
    small_mma<T, 64, 64>(ia, ib, i0, j0, acc);
    __syncthreads();
  }
  const bool dg = !full && (r64 == c64);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * a + lo, j = j0 + 16 * b + Tr<T>::drow(lane, r);
        if (!dg || i >= j) Cp[i + (long)j * ldc] = cv[a][b][r] - acc[a][b][r];
      }
  guest.leave();
  sem_signal(signal_sem);  // (one count per workgroup that has a block: n (n + 1) / 2 of the n x n grid when !full)
}

// One in-tile POTRF step in ONE launch (chain-bound form, section "device-side dependencies"): the first
// 4 nr workgroups are k_solve_small's (X = A Winv_s^T for the 32-row slabs of block rows s+1.. of block column
// s, in place) and count up `cnt`; the others are k_small_update's for the n (n + 1) / 2 lower 64 x 64 blocks
// of the trailing part (n = 2 nr): resident from the start with their block of C loaded, they poll `cnt` for
// the 4 nr solves and apply C -= X X^T.  Workgroups are dispatched in index order, so the solves never wait
// behind the pollers.  Saves the second launch of the step (drain, dispatch, cold loads) on the critical chain.
template <typename T>
struct IntileLds {
  union {
    struct {
      SmallImg<T, 32> a;
      SmallImg<T, MACRO> b;
    } sv;
    struct {
      SmallImg<T, 64> a, b;
    } up;
  };
};
template <typename T>
__global__ __launch_bounds__(256, 2) void k_intile_step(T *tile, int mb, int nbm, int s, const T *__restrict__ winv,
                                                        int *ytab, int *cnt, int *fail) {
  __shared__ IntileLds<T> L;
  const int nr = nbm - 1 - s, nsolve = 4 * nr;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lo = lane & 15;
  typename Tr<T>::acc_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = T(0);
  if ((int)blockIdx.x < nsolve) {
    GuestOnCu guest(ytab);
    __builtin_amdgcn_s_setprio(2);
    T *Ap = tile + (long)(s + 1) * MACRO + 32 * (int)blockIdx.x + (long)s * MACRO * mb;
    const T *Bp = winv + (long)s * MACRO * MACRO;
    T ra[32 * SK / 256], rb[MACRO * SK / 256];
    small_gload<T, 32>(Ap, mb, 0, ra);
    small_gload<T, MACRO>(Bp, MACRO, 0, rb);
    for (int ph = 0; ph < MACRO / SK; ++ph) {
      small_lstore<T, 32>(L.sv.a, ra);
      small_lstore<T, MACRO>(L.sv.b, rb);
      __syncthreads();
      if (ph + 1 < MACRO / SK) {
        small_gload<T, 32>(Ap, mb, (ph + 1) * SK, ra);
        small_gload<T, MACRO>(Bp, MACRO, (ph + 1) * SK, rb);
      }
      small_mma<T, 32, MACRO>(L.sv.a, L.sv.b, 0, 32 * w, acc);
      __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Ap[16 * a + lo + (long)(32 * w + 16 * b + Tr<T>::drow(lane, r)) * mb] = acc[a][b][r];
    guest.leave();
    sem_signal(cnt);
    return;
  }
  // the idx-th lower block, column by column
  int c64 = 0, left = (int)blockIdx.x - nsolve, len = 2 * nr;
  while (left >= len) {
    left -= len;
    --len;
    ++c64;
  }
  const int r64 = c64 + left;
  T *tr = tile + (long)(s + 1) * MACRO * (mb + 1);                        // trailing part of the tile
  const T *xs = tile + (long)(s + 1) * MACRO + (long)s * MACRO * mb;      // block column s below the diagonal
  const T *Ap = xs + 64 * r64, *Bp = xs + 64 * c64;
  T *Cp = tr + 64 * r64 + (long)64 * c64 * mb;
  const int i0 = 32 * (w & 1), j0 = 32 * (w >> 1);
  T cv[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cv[a][b][r] = Cp[(i0 + 16 * a + lo) + (long)(j0 + 16 * b + Tr<T>::drow(lane, r)) * mb];
  sem_wait(cnt, nsolve, fail);
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(2);
  T ra[64 * SK / 256], rb[64 * SK / 256];
  small_gload<T, 64>(Ap, mb, 0, ra);
  small_gload<T, 64>(Bp, mb, 0, rb);
  for (int ph = 0; ph < MACRO / SK; ++ph) {
    small_lstore<T, 64>(L.up.a, ra);
    small_lstore<T, 64>(L.up.b, rb);
    __syncthreads();
    if (ph + 1 < MACRO / SK) {
      small_gload<T, 64>(Ap, mb, (ph + 1) * SK, ra);
      small_gload<T, 64>(Bp, mb, (ph + 1) * SK, rb);
    }
    small_mma<T, 64, 64>(L.up.a, L.up.b, i0, j0, acc);
    __syncthreads();
  }
  const bool dg = r64 == c64;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * a + lo, j = j0 + 16 * b + Tr<T>::drow(lane, r);
        if (!dg || i >= j) Cp[i + (long)j * mb] = cv[a][b][r] - acc[a][b][r];
      }
  guest.leave();
}

// generic one-tile C := alpha*A*B^T + beta*C (GEMM NoTrans/Trans, or SYRK Lower)
// blockIdx.z = z1 + nz1 * z2 selects the tiles of a batch: A + z1 sA, B + z2 sB, C + z1 sC1 + z2 sC2
// (potrs: Z(r,i) -= Z(r,k) L(i,k)^T for all right-hand-side tile rows r and tile columns i in one launch)
template <typename T>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_tile(const T *__restrict__ A,
                                                         const T *__restrict__ B, T *C, int mb,
                                                         int nbm, T alpha, T beta, int lower, int nz1, long sA,
                                                         long sB, long sC1, long sC2) {
  __shared__ SmemP<T> sm;
  const int mi = blockIdx.x, mj = blockIdx.y;
  if (lower && mi < mj) return;
  {
    const int z1 = blockIdx.z % nz1, z2 = blockIdx.z / nz1;
    A += z1 * sA;
    B += z2 * sB;
    C += z1 * sC1 + z2 * sC2;
  }
  Acc<T> acc;
  acc_zero<T>(acc);
  nt_kloop_paired<T>(A + mi * MACRO, mb, B + mj * MACRO, mb, mb, acc, sm);
  nt_epilogue_paired<T>(C + mi * MACRO + (long)mj * MACRO * mb, mb, acc, alpha, beta,
                 lower && mi == mj);
}

// dst[z] <- src[z], `bytes` (a multiple of 16) each: the private copies of the tiles a TRSM batch solves in place
__global__ __launch_bounds__(256) void k_copy_ptrs(const void *const *__restrict__ src, void *const *__restrict__ dst, long bytes) {
  const uint4 *s = reinterpret_cast<const uint4 *>(src[blockIdx.y]);
  uint4 *d = reinterpret_cast<uint4 *>(dst[blockIdx.y]);
  const long n = bytes / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}

// ------------------------------------------------------------------------------
// 128 x 128 diagonal block: lower Cholesky and the inverse of the triangular factor,
// one 256-thread workgroup, the block resident in LDS.
//
// Blocked by 16 columns.  Per 16-column panel:
//   phase A (waves 0,1; registers + v_readlane, no LDS traffic, no barriers): each
//     wave factors the 16x16 diagonal block held one row per lane (lanes 0-15) and, in
//     the same sweep, solves its own 64 rows below it (one row per lane) against the
//     factor -- every L(c,jj) is broadcast once through an SGPR and feeds three FMAs:
//     the diagonal block's own update, the rows below, and the forward substitution
//     that builds the diagonal block's inverse (wave 0, one column per lane);
//   phase B (all waves): trailing update of the 16x16 blocks below/right with the
//     16x16x4 MFMA straight out of LDS.
// Then the inverse of the whole factor, in place over the strictly-lower blocks, by recursive
// halving (invert_level), again on the MFMA.
// factor = 0: only invert an already factored block.  The strict upper triangle of A
// is never read or written.
// ------------------------------------------------------------------------------
__device__ __forceinline__ double rlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// acc(D layout) += First(16x16) * Second(16x16); element (i,k) of First at F[i*fi + k*fk],
// element (k,j) of Second at Sd[k*sk + j*sj].  Lane l ends up with D[drow(l,reg)][l & 15].
// swz = 1 / 2: First / Second is a 16x16 block stored column-major, ld 16, with the rows of column c rotated
// by c (wd_idx): the layout of the diagonal blocks' inverses, which are written and read a column per lane
// (unrotated, all sixteen lanes of such an access fall on one LDS bank; a leading dimension of 17 would
// cost the kilobyte that lets the kernel share a CU with a trailing-update workgroup)
__device__ __forceinline__ int wd_idx(int r, int c) { return ((r + c) & 15) + 16 * c; }
template <typename T>
__device__ __forceinline__ void mm16(typename Tr<T>::acc_t &acc, const T *F, int fi, int fk,
                                     const T *Sd, int sk, int sj, bool negate, int swz = 0) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    T f = (swz == 1) ? F[wd_idx(lo, 4 * q + hi)] : F[lo * fi + (4 * q + hi) * fk];
    const T g = (swz == 2) ? Sd[wd_idx(4 * q + hi, lo)] : Sd[(4 * q + hi) * sk + lo * sj];
    if (negate) f = -f;
    acc = Tr<T>::mfma(f, g, acc);
  }
}

// LDS image of the diagonal block: only the 36 lower 16x16 blocks, each column-major with
// leading dimension 17 (272 elements per block): 77 KiB instead of 129 KiB, so the kernel
// fits on a CU beside one resident trailing-update workgroup (64 KiB) instead of waiting
// for a whole CU to drain.  ld = 17 keeps row reads, column reads and MFMA fragment
// reads of a block free of bank conflicts.
constexpr int DB_LD = 17, DB_SZ = 16 * DB_LD, DB_NP = MACRO / 16;
__device__ __forceinline__ int db_off(int r, int c) {  // block (r >= c)
  return (c * DB_NP - (c * (c - 1)) / 2 + (r - c)) * DB_SZ;
}
__device__ __forceinline__ int db_idx(int i, int j) {  // element (i,j), block row >= block col
  return db_off(i >> 4, j >> 4) + (i & 15) + (j & 15) * DB_LD;
}

// sqrt(d) and 1/sqrt(d) by v_rsq + two coupled Goldschmidt steps (~10 dependent fp64 ops
// instead of the ~25 of IEEE sqrt followed by an IEEE divide).  The fp64 MFMA and the fp64
// VALU share the SIMD's DP units, so beside a trailing update every dependent fp64
// instruction of this kernel waits for a 64-cycle MFMA: the pivot chain is the critical path.
__device__ __forceinline__ void sqrt_rsqrt(double d, double &s, double &rinv) {
  double r = __builtin_amdgcn_rsq(d);
  double g = d * r, h = 0.5 * r;
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  // one correction step on the root itself: g += h * (d - g*g) * ... (h ~ 1/(2 sqrt d))
  const double c = __builtin_fma(-g, g, d);
  s = __builtin_fma(c, h, g);
  rinv = h + h;
}
__device__ __forceinline__ void sqrt_rsqrt(float d, float &s, float &rinv) {
  s = sqrtf(d);
  rinv = 1.0f / s;
}

// One level of the recursive inversion of the lower block-triangular factor (16x16 blocks; the
// diagonal blocks' inverses are in Wd, everything below the diagonal in S).  At half-size H the
// 8 x 8 block matrix splits into 8 / 2H groups [[L11, 0], [L21, L22]] whose W11 = inv(L11) and
// W22 = inv(L22) are already in place; W21 = -W22 L21 W11 in two stages, four tasks each (one per
// wave): T = L21 W11 by rows (a task reads only its own row of L21, holds the H results in
// registers and overwrites that row), then W21 = -W22 T by columns (same argument).  Three
// levels, H = 1, 2, 4: depth 31 block products instead of the 7 dependent block-column sweeps
// of the in-place dtrti2 order.
template <typename T, int H>
__device__ __forceinline__ void invert_level(T *S, T (*Wd)[16 * 16], int w) {
  using acc_t = typename Tr<T>::acc_t;
  const int lane = threadIdx.x & 63, lo = lane & 15;
  const int g0 = (w / H) * 2 * H, o = w % H;
  acc_t acc[H];
  {
    const int a = g0 + H + o;
#pragma unroll
    for (int bb = 0; bb < H; ++bb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[bb][reg] = T(0);
    // term t of every product in turn (each accumulator still sums over c ascending): the H chains are
    // independent and interleave, where one product after the other left the wave waiting for its own MFMAs
#pragma unroll
    for (int t = 0; t < H; ++t)
#pragma unroll
      for (int bb = 0; bb < H - t; ++bb) {
        const int b = g0 + bb, c = b + t;
        const T *Sec = (t == 0) ? Wd[c] : S + db_off(c, b);
        mm16<T>(acc[bb], S + db_off(a, c), 1, DB_LD, Sec, 1, DB_LD, false, (t == 0) ? 2 : 0);
      }
#pragma unroll
    for (int bb = 0; bb < H; ++bb) {
      T *Cb = S + db_off(a, g0 + bb);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Cb[Tr<T>::drow(lane, reg) + lo * DB_LD] = acc[bb][reg];
    }
  }
  __syncthreads();
  {
    const int b = g0 + o;
#pragma unroll
    for (int aa = 0; aa < H; ++aa)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[aa][reg] = T(0);
#pragma unroll
    for (int t = 0; t < H; ++t)
#pragma unroll
      for (int aa = t; aa < H; ++aa) {  // (product aa has terms c = g0 + H .. a, i.e. t = 0 .. aa)
        const int a = g0 + H + aa, c = g0 + H + t;
        const T *Fp = (t == aa) ? Wd[a] : S + db_off(a, c);
        mm16<T>(acc[aa], Fp, 1, DB_LD, S + db_off(c, b), 1, DB_LD, true, (t == aa) ? 1 : 0);
      }
#pragma unroll
    for (int aa = 0; aa < H; ++aa) {
      T *Cb = S + db_off(g0 + H + aa, b);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Cb[Tr<T>::drow(lane, reg) + lo * DB_LD] = acc[aa][reg];
    }
  }
  __syncthreads();
}

// v in the lanes of mask m, zero in the others (m a compile-time constant: two scalar moves, no compare)
__device__ __forceinline__ double keep_lanes(double v, unsigned long long m) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int rl, rh;
  asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4" : "=&v"(rl), "=&v"(rh) : "v"(lo), "v"(hi), "s"(m));
  return __hiloint2double(rh, rl);
}
__device__ __forceinline__ float keep_lanes(float v, unsigned long long m) {
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
  return r;
}
// one LDS store from the lanes of mask m only (the wave runs with all lanes active around it): no compare, no branch
template <int OFF>
__device__ __forceinline__ void lds_store_lanes(unsigned addr, double v, unsigned long long m) {
  asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1 offset:%3\n\ts_mov_b64 exec, -1" : : "v"(addr), "v"(v), "s"(m), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_store_lanes(unsigned addr, float v, unsigned long long m) {
  asm volatile("s_mov_b64 exec, %2\n\tds_write_b32 %0, %1 offset:%3\n\ts_mov_b64 exec, -1" : : "v"(addr), "v"(v), "s"(m), "n"(OFF) : "memory");
}

// Consumer side of phase A: wait until an LDS word is no longer zero (DS operations of the producing wave execute in order and
// the flag is written last, so a raised flag vouches for the data stored before it).  One opaque instruction sequence
// on purpose: written as a C loop, the unrolled copies gave the register allocator a control-flow graph on which the
// kernel needed 400 VGPRs instead of 256.  Bounded (the producer needs ~0.1 us per column; the bound is ~0.5 s of
// polling -- two processes time-sliced on one GPU were seen to exceed a 1 ms bound): on giving up it stores a value
// other than 0 / 1 into DiagLds::failed, which the kernel turns into info = INT_MAX after the panel's barrier -- a
// logic error shows up as a loud failure, never as a hung GPU or a silently wrong factor.  The "memory" clobber keeps
// the loads of the data behind the flag below it.
__device__ __forceinline__ void lds_wait_flag(unsigned flag_addr, unsigned failed_addr) {
  int v, n = 0x3fffff;
  asm volatile(
      "1:\n\t"
      "ds_read_b32 %0, %2\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_eq_u32_e32 vcc, 0, %0\n\t"
      "s_cbranch_vccz 2f\n\t"
      "s_sub_u32 %1, %1, 1\n\t"
      "s_cmp_eq_u32 %1, 0\n\t"
      "s_cbranch_scc0 1b\n\t"
      "ds_write_b32 %3, %3\n\t"
      "2:"
      : "=&v"(v), "+s"(n)
      : "v"(flag_addr), "v"(failed_addr)
      : "vcc", "scc", "memory");
}

template <typename T>
struct DiagLds {
  T S[DB_NP * (DB_NP + 1) / 2 * DB_SZ];
  T Wd[DB_NP][16 * 16];
  T Lrinv[16];      // phase A: MINUS the reciprocals of the pivots of the 16x16 factor being built (its columns go
                    // straight into the block image S).  Cleared to +0 between panels: -1/sqrt(d) is never +0 (it is
                    // negative, -0 for d = +Inf, NaN or -Inf for a bad pivot), so the slot is also the column's flag
  int failed;
};

// The 128 x 128 block at A (leading dimension ld) into the LDS image S (lower 16 x 16 blocks, db_off), zeros above the
// diagonal inside the diagonal blocks.  256 threads; lane rp of wave w4 owns rows 2 rp, 2 rp + 1 (one 16-byte global load
// per column: a wave instruction fetches a whole 128-row column) of the block columns w4 and 7 - w4 -- nine of the 36
// lower blocks per wave; all 32 loads of a lane in flight at once (the kernel starts cold, behind the launch that
// produced the block: latency, not bandwidth).
template <typename T>
__device__ __forceinline__ void load_block_lower(const T *__restrict__ A, int ld, T *S) {
  typedef T pair2_t __attribute__((ext_vector_type(2)));
  constexpr int NB = 16;
  const int t = threadIdx.x, i0 = 2 * (t & 63), br = i0 >> 4, w4 = t >> 6, r = i0 & 15;
  pair2_t v[2][16];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = h ? 7 - w4 : w4;
    const T *src = A + i0 + (size_t)(NB * c) * ld;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      v[h][u] = pair2_t{T(0), T(0)};
      if (br > c || (br == c && r + 1 >= u)) v[h][u] = *reinterpret_cast<const pair2_t *>(src + (size_t)u * ld);
      if (br == c && r < u) v[h][u][0] = T(0);  // (row r is above the diagonal in this column, row r + 1 is on it)
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = h ? 7 - w4 : w4;
    if (br < c) continue;
    T *dst = S + db_off(br, c) + r;
#pragma unroll
    for (int u = 0; u < 16; ++u) dst[u * DB_LD] = v[h][u][0], dst[u * DB_LD + 1] = v[h][u][1];
  }
}

// FLOW (k_flow_factor): the block is already in L.S (loaded, and updated by the earlier steps' products, by the
// caller); every finished 16-column panel of the factor and the inverse of its 16 x 16 diagonal block are
// PUBLISHED for the row-slab waves of k_flow_rows -- write-through (sc1) stores into the tile / winv right after
// phase A, the panel counter raised one phase later, behind the barrier that has drained them -- so the factor
// is not stored again at the end; a failed pivot raises the flow's abort word.
template <typename T>
struct FlowPub {
  int *fpan;   // panels of this diagonal block published so far (0 .. 8)
  int *abort;  // the flow's abort word
};
template <typename T, bool FLOW = false>
__device__ __forceinline__ void potrf_diag_body(T *A, int ld, T *__restrict__ winv, int *info,
                                                int info_base, int factor, DiagLds<T> &L,
                                                unsigned long long *ph = nullptr, const FlowPub<T> *fp = nullptr) {
  constexpr int n = MACRO, NB = 16, NP = DB_NP;
  unsigned long long tA = 0, tB = 0, tl = 0;
#define PH_NOW() (ph ? __builtin_amdgcn_s_memrealtime() : 0ull)
  T *S = L.S;
  T(*Wd)[NB * NB] = L.Wd;
  int &failed = L.failed;
  const int t = threadIdx.x, lane = t & 63, lo = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform by construction: scalar branches
  // Global <-> LDS: thread t owns row (t & 127) and block columns 4 (t >> 7) .. +3; per 16x16
  // block the 16 elements of its row are one base address + constant strides on both sides
  // (no per-element index arithmetic), 16 independent accesses in flight, rows of consecutive
  // threads contiguous in global memory.
  // Global <-> LDS, two ROWS per lane (16 bytes on the global side: one wave instruction moves a whole 128-row column):
  // lane rp of wave w4 owns rows 2 rp, 2 rp + 1 of the block columns w4 and 7 - w4 (nine of the 36 lower blocks per
  // wave); per block the pair's 16 columns are one base address + constant strides on both sides.
  typedef T pair2_t __attribute__((ext_vector_type(2)));
  const int gi0 = 2 * (t & 63), gbr = gi0 >> 4, gw4 = t >> 6;
  if constexpr (!FLOW) load_block_lower<T>(A, ld, S);
  if constexpr (!FLOW) {
    if (t < 16) L.Lrinv[t] = T(0);
    if (t == 0) failed = 0;
    __syncthreads();
  }
  if (ph && t == 0) ph[0] = PH_NOW();  // loaded

  if (factor) {
    for (int p = 0; p < NP; ++p) {
      const int j0 = NB * p;
      tl = PH_NOW();
      // ---- phase A: wave 0 factors the 16x16 diagonal block and publishes each finished
      // column (and 1/pivot) in LDS; waves 1, 2 (rows below, one row per lane) and wave 3 (the
      // block's inverse, one column per lane) follow one column behind, picking the multipliers
      // up as LDS broadcasts.  Only wave 0's pivot chain is on the critical path.
      {
        // LDS pointers with an opaque base: the 272 constant addresses below then encode as one
        // base register + immediate offsets instead of one hoisted VGPR each
        typedef __attribute__((address_space(3))) T lds_t;
        lds_t *Lc = (lds_t *)(S + db_off(p, p)), *Lr = (lds_t *)L.Lrinv;  // column jj of the factor: Lc + jj * DB_LD
        asm volatile("" : "+v"(Lc), "+v"(Lr));
        // lane ids made opaque per panel: otherwise every lane mask, select and address of the
        // unrolled column code below is hoisted out of the panel loop and kept in registers for
        // its whole length (that alone cost > 100 VGPRs and most of the SGPR spills)
        int lane_ = lane, lo_ = lo;
        asm volatile("" : "+v"(lane_), "+v"(lo_));
        if (w == 0) {
          // The 16x16 block lives in ONE MFMA accumulator, spread over all 64 lanes (lane (hi, lo) holds column lo
          // of the rows drow(lane, 0..3)), as E = -A and as the FULL symmetric matrix: row jj of it -- the column
          // being eliminated, indexed by lo -- then sits in one register of the sixteen lanes of group K, which is
          // exactly where the 16x16x4 MFMA wants both operands of a rank-1 update in k-slot K.  A pivot is:
          // read the diagonal entry (v_readlane), v_rsq + Goldschmidt, scale that register (zero in the other
          // lanes), ONE MFMA  E += a a^T , three LDS stores (the column into the block image, -1/pivot, the
          // column's flag) -- about 30 instructions instead of the 58 of the one-row-per-lane form (16 useful
          // lanes, 14 FMAs per pivot for the trailing columns), which is what an in-order wave pays for.  The
          // square roots themselves (the diagonal of the factor) wait for the end of the
          // panel: rows and columns <= jj of E are never touched again (a is zero there), so the diagonal of E still
          // holds every pivot then, and sixteen lanes take the sixteen roots at once.  Same values bit for bit: the
          // scaled column is A * (h + h) as before, the MFMA with three zero k-slots is fma(-a_i, a_c, A(i,c)), an
          // update of E = -A rounds as the update of A does, and the root of a lane is the sequence the wave-uniform
          // code ran.
          using acc_t = typename Tr<T>::acc_t;
          const int hi_ = lane_ >> 4;
          acc_t E;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int i = Tr<T>::drow(lane_, reg);
            E[reg] = -Lc[max(i, lo_) + min(i, lo_) * DB_LD];  // (the block image holds the lower triangle)
          }
          // The next pivot does not wait for the MFMA: d' = A(jj+1,jj+1) - u^2 with u = A(jj+1,jj) / sqrt(d), both
          // entries read (v_readlane) from E as the PREVIOUS update left it -- the value the MFMA puts on the diagonal,
          // bit for bit (same product, same fma).  So the scalar chain rsq -> Goldschmidt -> u -> d' -> rsq ... and the
          // chain MFMA -> row -> scale -> MFMA run side by side instead of in series.  The order of issue is forced
          // (ORDER: an empty asm that ties values together; what produces them is above it, what uses them below) --
          // the wave issues in order, and the compiler's own order put the reads of E behind the whole scalar chain
          // and the next pivot behind the vector work (261 cycles per pivot against 175: scripts/exp/rank1_probe2.hip).
          T dcur = -rlane(E[0], 0);
          static_for<0, NB>([&](auto JJ) {
            constexpr int jj = decltype(JJ)::value;
            constexpr int R = (sizeof(T) == 8) ? jj / 4 : jj % 4, K = (sizeof(T) == 8) ? jj % 4 : jj / 4;
            // the lanes of group K below the diagonal: the multipliers L(lo, jj), lo > jj
            constexpr unsigned long long M = (0xFFFFull & ~((2ull << jj) - 1ull)) << (16 * K);
            const T d = dcur;
            T rinvn;                 // -1 / sqrt(d)
            [[maybe_unused]] T sq32 = T(0);
            T Em, as = T(0), xs = T(0);
            // what the rest of the pivot needs of E (as the previous MFMA left it): row jj beyond the diagonal, zero in
            // every other lane (selected ahead of the product: the MFMA must not read a register an inline-asm
            // instruction has just written -- the compiler inserts the wait states a VALU result needs before an MFMA
            // reads it only behind instructions it knows), -A(jj+1, jj) and -A(jj+1, jj+1)
            auto reads = [&]() {
              Em = keep_lanes(E[R], M);
              if constexpr (jj + 1 < NB) {
                constexpr int R1 = (sizeof(T) == 8) ? (jj + 1) / 4 : (jj + 1) % 4, K1 = (sizeof(T) == 8) ? (jj + 1) % 4 : (jj + 1) / 4;
                as = rlane(E[R], 16 * K + jj + 1);
                xs = rlane(E[R1], 16 * K1 + jj + 1);
              }
            };
            if constexpr (sizeof(T) == 8) {
              // +Inf passes `d > 0` in LAPACK too (dpotf2: sqrt(Inf) = Inf, the column is scaled by 1/Inf = 0,
              // info stays 0).  v_rsq(Inf) = 0; with the product d r taken on min(d, DBL_MAX) the iteration
              // stays at g = h = 0 (instead of Inf * 0 = NaN), so 1/pivot = 0 and the next pivot is left
              // alone, as there; only sqrt(d) itself needs a select (below, at the end of the panel)
              const double r = __builtin_amdgcn_rsq((double)d);
              const double dm = __builtin_fmin((double)d, 1.7976931348623157e308);
              double g = dm * r, h = 0.5 * r;
              double e = __builtin_fma(-h, g, 0.5);
              g = __builtin_fma(g, e, g);
              h = __builtin_fma(h, e, h);
              e = __builtin_fma(-h, g, 0.5);
              asm volatile("" : "+v"(e), "+v"(h));                       // ORDER: the chain up to here ...
              reads();
              asm volatile("" : "+v"(e), "+v"(Em), "+s"(as), "+s"(xs));  // ... the reads of E (the previous MFMA has had ~70 cycles) ...
              h = __builtin_fma(h, e, h);
              rinvn = (T)(-h - h);
            } else {
              T rinv;
              sqrt_rsqrt(d, sq32, rinv);
              asm volatile("" : "+v"(rinv));
              reads();
              asm volatile("" : "+v"(rinv), "+v"(Em), "+s"(as), "+s"(xs));
              rinvn = -rinv;
            }
            if constexpr (jj + 1 < NB) {
              const T u = as * rinvn;
              if constexpr (sizeof(T) == 8) dcur = __builtin_fma(-u, u, -xs);
              else dcur = __builtin_fmaf(-u, u, -xs);
            }
            T a = Em * rinvn;  // (-A)(-1/sqrt d): column jj of the factor below the diagonal, indexed by lo
            asm volatile("" : "+v"(dcur), "+v"(a));                      // ... the next pivot and the column, then MFMA and stores
            if constexpr (jj + 1 < NB) E = Tr<T>::mfma(a, a, E);
            // column jj from the sixteen lanes of group K (zeros on and above the diagonal: the diagonal entry follows
            // at the end of the panel, the strict upper triangle is never read), then -1/pivot, wave-uniform (every
            // lane stores it: same address, same value) -- the column's flag: DS operations of one wave execute in order
            lds_store_lanes<jj * DB_LD * (int)sizeof(T)>((unsigned)(size_t)(Lc + lo_), a, 0xFFFFull << (16 * K));
            // fp32: the diagonal entry at once, from the pivot the chain used (the fp32 MFMA does not round like an fma, so
            // the diagonal of E may differ from it in the last bit -- near a zero pivot that could be the sign)
            if constexpr (sizeof(T) == 4) Lc[jj * (DB_LD + 1)] = sq32;
            Lr[jj] = rinvn;
            asm volatile("" ::: "memory");
          });
          // info (LAPACK: the first pivot that is not > 0, NaN included): exactly the pivots whose reciprocal is not
          // finite (d <= 0 and NaN give NaN or Inf; +Inf gives -0, tiny d a large finite value).  Read back from the
          // LDS (this wave's own stores, in order), sixteen at once -- nothing on the pivots' way.
          int bad = 0;
          {
            const T rv = Lr[lo_];
            const unsigned long long mb_ = __ballot(!(__builtin_fabs((double)rv) < __builtin_huge_val())) & 0xFFFFull;
            if (mb_) bad = __builtin_ctzll(mb_) + 1;
          }
          // the sixteen pivots from the diagonal of E: lane (K(c), c) holds pivot c in register R(c)
          {
            const int rsel = (sizeof(T) == 8) ? (lo_ >> 2) : (lo_ & 3), ksel = (sizeof(T) == 8) ? (lo_ & 3) : (lo_ >> 2);
            T dv = E[0];
            dv = (rsel == 1) ? E[1] : dv;
            dv = (rsel == 2) ? E[2] : dv;
            dv = (rsel == 3) ? E[3] : dv;
            const T d = -dv;
            const bool mine = hi_ == ksel;
            if (bad) {  // the first pivot that was not > 0 (NaN included)
              if (lane_ == 0) {
                atomicCAS(info, 0, info_base + j0 + bad);
                failed = 1;
              }
            } else if (mine && sizeof(T) == 8) {
              T sq;
              if constexpr (sizeof(T) == 8) {
                const double r = __builtin_amdgcn_rsq((double)d);
                const double dm = __builtin_fmin((double)d, 1.7976931348623157e308);
                double g = dm * r, h = 0.5 * r;
                double e = __builtin_fma(-h, g, 0.5);
                g = __builtin_fma(g, e, g);
                h = __builtin_fma(h, e, h);
                e = __builtin_fma(-h, g, 0.5);
                h = __builtin_fma(h, e, h);
                g = __builtin_fma(g, e, g);
                const double cc = __builtin_fma(-g, g, (double)d);
                sq = (T)__builtin_fma(cc, h, g);
                if (__double2hiint((double)d) == 0x7ff00000 && __double2loint((double)d) == 0)
                  sq = (T)__builtin_huge_val();
              } else {
                T rinv;
                sqrt_rsqrt(d, sq, rinv);
              }
              Lc[lo_ * (DB_LD + 1)] = sq;
            }
          }
        } else {
          // Followers: a lane of waves 1, 2 owns one row below the block (a lane beyond the last row repeats that
          // row: same values to the same addresses), a lane of wave 3 one column of the inverse of the 16x16 factor
          // (forward substitution on the identity; lanes 16-63 repeat lanes 0-15).  They take the factor about FOUR
          // columns at a time: one poll of the last column's flag, then the columns (the rows still needed) and their
          // reciprocals in one burst of LDS broadcast reads, then the right-looking updates of the own row.  Column
          // by column (round 4) every column cost a round trip of ~150 cycles through the LDS, more than wave 0 now
          // needs per pivot, and the followers ended a panel several columns behind.
          const bool rows = (w < 3);
          const bool active = rows ? (j0 + NB + 64 * (w - 1) < n) : true;
          if (active) {
            const unsigned flag_lds = (unsigned)(size_t)Lr + (sizeof(T) == 8 ? 4u : 0u);  // (the word with sign and exponent)
            const unsigned failed_lds = (unsigned)(size_t)(__attribute__((address_space(3))) int *)&L.failed;
            const int myrow = j0 + NB + 64 * (w - 1) + lane_;
            const int rr = (rows && myrow < n) ? myrow : n - 1;
            T *Rp = S + db_off(rr >> 4, p) + (rr & 15);
            const unsigned long long okmask = __ballot(rows && myrow < n);
            // (two copies of the unrolled sweep, one per kind of follower: with `rows` tested at run time every store
            // of a column sat behind two scalar branches)
            auto follow = [&](auto ROWS) {
              constexpr bool rows_ = decltype(ROWS)::value;
              T a[NB];
#pragma unroll
              for (int jj = 0; jj < NB; ++jj) a[jj] = rows_ ? Rp[jj * DB_LD] : ((lo_ == jj) ? T(1) : T(0));
              T *Wp = Wd[p] + 16 * lo_;  // wave 3: column lo_ of the inverse, rows rotated (wd_idx)
              // bursts of columns 0-3, 4-7, 8-11, 12-14 and, alone, 15: nothing of column 15 is needed but its
              // reciprocal, so behind the last pivot a follower is ONE LDS round trip, one product and one store away
              // from the barrier
              static_for<0, 5>([&](auto Q) {
                constexpr int q = decltype(Q)::value, c0 = 4 * q - (q == 4), nc = q < 3 ? 4 : (q == 3 ? 3 : 1);
                lds_wait_flag(flag_lds + (c0 + nc - 1) * (unsigned)sizeof(T), failed_lds);
                T col[4][NB], rn[4];
#pragma unroll
                for (int k = 0; k < nc; ++k) {
                  rn[k] = Lr[c0 + k];
#pragma unroll
                  for (int c = c0 + k + 1; c < NB; ++c) col[k][c] = Lc[(c0 + k) * DB_LD + c];
                }
                static_for<0, nc>([&](auto KK) {
                  constexpr int k = decltype(KK)::value, jj = c0 + k;
                  a[jj] *= -rn[k];  // (wave 0 publishes MINUS the reciprocal)
                  // (only the lanes that own a row / a column store)
                  if constexpr (rows_) lds_store_lanes<jj * DB_LD * (int)sizeof(T)>((unsigned)(size_t)(lds_t *)Rp, a[jj], okmask);
                  else lds_store_lanes<0>((unsigned)(size_t)(lds_t *)(Wp + ((jj + lo_) & 15)), a[jj], 0xFFFFull);
                  // (each update pinned where it is written: left alone, the compiler sinks the updates of a column to
                  // just before its scaling -- a left-looking sweep whose whole sum then sits on the way to the barrier,
                  // with every fetched column kept in registers)
                  static_for<jj + 1, NB>([&](auto C) {
                    constexpr int c = decltype(C)::value;
                    a[c] -= a[jj] * col[k][c];
                    asm volatile("" : "+v"(a[c]));
                  });
                });
              });
            };
            if (rows) follow(std::true_type{});
            else follow(std::false_type{});
          }
        }
      }
      __syncthreads();
      if (failed) {
        if (failed != 1 && t == 0) atomicCAS(info, 0, 0x7fffffff);  // a consumer gave up waiting (ColFetch)
        if constexpr (FLOW) {
          if (t == 0) __hip_atomic_store(fp->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
      }
      if (t < 16) L.Lrinv[t] = T(0);  // (the next panel's flags; its pollers start behind the barrier that ends phase B)
      if constexpr (FLOW) {
        // publish panel p: rows 16 p .. 127 of its 16 columns, on or below the diagonal; one wave instruction
        // stores one column (two consecutive rows per lane), so every 128-byte line is written whole by one store
        const int row0 = 2 * (t & 63);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 4 * q + (t >> 6), jg = j0 + col;
          if (row0 + 1 >= jg) {
            const T *src = S + db_off(row0 >> 4, p) + (row0 & 15) + col * DB_LD;
            const T v0 = src[0], v1 = src[1];
            T *dst = A + row0 + (size_t)jg * ld;
            if (row0 >= jg) store_pair_sc1(dst, v0, v1);
            else store_sc1(dst + 1, v1);
          }
        }
        // ... and the inverse of its 16 x 16 diagonal block, zeros above the diagonal included (the diagonal blocks
        // of the block's inverse ARE these: the full inverse stored at the end repeats the same values)
        {
          const int i = t & 15, j = t >> 4;
          const T v = (i >= j) ? Wd[p][wd_idx(i, j)] : T(0);
          store_sc1(winv + (j0 + i) + (size_t)(j0 + j) * n, v);
        }
      }
      tA += PH_NOW() - tl;
      tl = PH_NOW();
      // ---- phase B: S(r,c) -= X(r,p) X(c,p)^T for 16x16 blocks p < c <= r, dealt round-robin to the
      // four waves.  A wave takes its blocks two at a time: one block is a chain of LDS reads, four
      // dependent MFMAs and LDS writes (~500 cycles of latency for ~256 of issue), two independent ones
      // interleave.
      {
        const int nblk = (NP - 1 - p) * (NP - p) / 2;
        auto block_of = [&](int idx, int &r, int &c) {  // idx-th block in column-major order of the trailing part
          c = p + 1;
          int left = idx, len = NP - 1 - p;
          while (left >= len) {
            left -= len;
            --len;
            ++c;
          }
          r = c + left;
        };
        for (int i0 = w; i0 < nblk; i0 += 8) {
          const int i1 = i0 + 4;
          const bool two = i1 < nblk;
          int r0, c0, r1, c1;
          block_of(i0, r0, c0);
          block_of(two ? i1 : i0, r1, c1);
          typename Tr<T>::acc_t acc0, acc1;
          T *Cb0 = S + db_off(r0, c0), *Cb1 = S + db_off(r1, c1);
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            acc0[reg] = Cb0[Tr<T>::drow(lane, reg) + lo * DB_LD];
            acc1[reg] = Cb1[Tr<T>::drow(lane, reg) + lo * DB_LD];
          }
          // D[i][j] = sum_k X(r)[i][k] * X(c)[j][k]
          const T *F0 = S + db_off(r0, p), *G0 = S + db_off(c0, p), *F1 = S + db_off(r1, p), *G1 = S + db_off(c1, p);
          const int hi = lane >> 4;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const T f0 = -F0[lo + (4 * q + hi) * DB_LD], g0 = G0[(4 * q + hi) * DB_LD + lo];
            const T f1 = -F1[lo + (4 * q + hi) * DB_LD], g1 = G1[(4 * q + hi) * DB_LD + lo];
            acc0 = Tr<T>::mfma(f0, g0, acc0);
            acc1 = Tr<T>::mfma(f1, g1, acc1);
          }
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) Cb0[Tr<T>::drow(lane, reg) + lo * DB_LD] = acc0[reg];
          if (two) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Cb1[Tr<T>::drow(lane, reg) + lo * DB_LD] = acc1[reg];
          }
        }
      }
      if constexpr (FLOW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's panel stores have landed
      __syncthreads();
      if constexpr (FLOW) {
        if (t == 0) __hip_atomic_store(fp->fpan, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tB += PH_NOW() - tl;
    }
    if (ph && t == 0) {
      ph[1] = tA;
      ph[2] = tB;
    }
    if constexpr (!FLOW) {  // (FLOW: every panel was stored when it was published)
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        const int c = h ? 7 - gw4 : gw4;
        if (gbr < c) continue;
        pair2_t v[16];
        const T *src = S + db_off(gbr, c) + (gi0 & 15);
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = pair2_t{src[u * DB_LD], src[u * DB_LD + 1]};
        T *dst = A + gi0 + (size_t)(NB * c) * ld;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (gbr > c || (gi0 & 15) >= u) *reinterpret_cast<pair2_t *>(dst + (size_t)u * ld) = v[u];
          else if ((gi0 & 15) + 1 >= u) dst[(size_t)u * ld + 1] = v[u][1];
        }
      }
    }
  }
  if (ph && t == 0) ph[3] = PH_NOW();  // L stored
  // inverses of the eight 16x16 diagonal blocks (one column per lane, forward substitution
  // with L(c,jj) broadcast by v_readlane), two blocks per wave, all waves in parallel
  if (!factor) {  // (when factoring, wave 3 built them during phase A)
    for (int p = w; p < NP; p += 4) {
      T x[NB];
      const T *Db = S + db_off(p, p);  // L(c, jj) at Db[c + jj * DB_LD]: read as LDS broadcasts
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) x[jj] = (lo == jj) ? T(1) : T(0);
#pragma unroll
      for (int jj = 0; jj < NB; ++jj) {
        x[jj] *= T(1) / Db[jj + jj * DB_LD];
#pragma unroll
        for (int c = jj + 1; c < NB; ++c) x[c] -= x[jj] * Db[c + jj * DB_LD];
      }
      if (lane < NB) {
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) Wd[p][wd_idx(jj, lane)] = x[jj];
      }
    }
  }
  // LDS-only barrier: the stores of L to global memory stay in flight across the inversion (a full
  // __syncthreads() waits for them: ~1.5 us of the chain per diagonal block)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (ph && t == 0) ph[4] = PH_NOW();  // Wd done

  // ---- inverse of the whole factor, in place over the strictly-lower blocks
  invert_level<T, 1>(S, Wd, w);
  invert_level<T, 2>(S, Wd, w);
  invert_level<T, 4>(S, Wd, w);
  if (ph && t == 0) ph[5] = PH_NOW();  // phase C done
#undef PH_NOW
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {  // winv: full 128 x 128, zero above the diagonal
    const int c = h ? 7 - gw4 : gw4;
    pair2_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = pair2_t{T(0), T(0)};
    if (gbr == c) {
      const T *src = Wd[c];
      const int r = gi0 & 15;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (r >= u) v[u][0] = src[wd_idx(r, u)];
        if (r + 1 >= u) v[u][1] = src[wd_idx(r + 1, u)];
      }
    } else if (gbr > c) {
      const T *src = S + db_off(gbr, c) + (gi0 & 15);
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = pair2_t{src[u * DB_LD], src[u * DB_LD + 1]};
    }
    T *dst = winv + gi0 + (size_t)(NB * c) * n;
#pragma unroll
    for (int u = 0; u < 16; ++u) *reinterpret_cast<pair2_t *>(dst + (size_t)u * n) = v[u];
  }
}

// dbg (diagnostic, may be null): dbg[0] = running slot counter, then per launch the
// workgroup's own {start, end} in 100 MHz realtime ticks.
// (256, 2): at most 256 VGPRs, so that a wave fits on a SIMD beside one resident trailing-update
// wave (244 VGPRs); otherwise the kernel has to wait for a whole CU to drain.
template <typename T>
__global__ __launch_bounds__(256, 2) void k_potrf_diag(T *A, int ld, T *__restrict__ winv, int *info,
                                                    int info_base, int factor,
                                                    unsigned long long *dbg, int *ytab,
                                                    const int *wait_sem = nullptr, int wait_target = 0,
                                                    int *signal_sem = nullptr) {
  __shared__ DiagLds<T> L;
  __shared__ unsigned long long slot_s;
  sem_wait(wait_sem, wait_target, info);  // (before the CU is asked to yield: the poll may last a while)
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(3);
  unsigned long long t0 = 0;
  unsigned long long *ph = nullptr;
  if (dbg) {
    t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) slot_s = atomicAdd(dbg, 1ull);
    __syncthreads();
    if (slot_s < 1000) ph = dbg + 1 + 8 * slot_s + 2;
  }
  potrf_diag_body<T>(A, ld, winv, info, info_base, factor, L, ph);
  if (ph && threadIdx.x == 0) {
    ph[-2] = t0;
    ph[-1] = __builtin_amdgcn_s_memrealtime();
  }
  guest.leave();
  sem_signal(signal_sem);
}

// ------------------------------------------------------------------------------
// The tile POTRF of a chain-bound wave as a FLOW (round 4): two persistent launches whose workgroups hand their
// results on through polled counters, so that nothing but the factorisation of the 128 x 128 diagonal blocks is
// on the chain -- no kernel boundary, no load, no store of L, no inverse, no in-tile solve / update launch.
//
//   k_flow_factor  one workgroup per diagonal block s.  It loads its block, subtracts the products of the earlier
//                  steps  D_s -= sum_{k<s} X(s,k) X(s,k)^T  itself, 16 columns of X at a time as the row-slab waves
//                  publish them (the last step's panels arrive while block s-1 is still being factored; only its
//                  last panel is waited for), factors (potrf_diag_body<FLOW>), publishing every finished 16-column
//                  panel of L_s and the inverse of its 16 x 16 diagonal block, and ends as k_potrf_diag does
//                  (inverse of the whole block for the panel TRSM on the other stream, counter D[s]).
//   k_flow_rows    one WAVE per 16-row slab of the block rows below the first.  At step s it solves its rows of
//                  block (r, s) against L_s right-looking, one 16-column panel behind the factorisation:
//                  X_p = A_p W_p^T (W_p = inverse of the panel's diagonal block), A_c -= X_p L_s(c,p)^T, c > p --
//                  operands straight from global memory into MFMA registers, the slab in accumulators: the
//                  accumulator of one product IS the B operand of the next (with swapped operands lane (i, q) holds
//                  X[i][drow(q, r)], r = 0..3, exactly the four k's MFMA r wants when the other operand is fetched
//                  in the same order), so there is no LDS and no barrier in this kernel.  Then it applies X(r,s) to
//                  its rows of the blocks (r, c), s < c < r (needs X(c,s) complete), so that block (r, s+1) is
//                  up to date when step s+1 starts.  It raises I[s] (what the panel TRSM's update on the other
//                  stream polls) once per step.
//
// Chain per 128 columns: the diagonal-block factorisation (phases A + B) + one hand-off each way (last panel ->
// rows of block s+1 -> rank-16 update of D_{s+1}): 44-50 us measured (profiles/r04_flow_step_stamps.txt) against 72.
// Why two launches and not one per step: a version with one launch per step and stream (rows, products, the panel
// TRSM and K = 128 slices of column k+1 all following the factorisation step by step) was built and measured this
// round -- kernels of a handful of workgroups that follow a long-lived polling kernel start 25-30 us late and
// their workgroups up to 30 us apart once five queues are active; every such boundary was on the chain.
//
// Flow control block `fc` (ints; every counter on a 128-byte line of its own, zeroed by the host):
//   line 0             abort: a failed pivot or a poll that gave up; every poll watches it, and whoever sees it
//                      raises the counters the OTHER streams poll (D[s], I[s]) and leaves
//   line 1 + s         panels of diagonal block s published (0 .. 8)
//   line 1 + nbm + r nbm + s, ints 0..7   waves of block row r that have published panel p of X(r, s) (0 .. 8)
// Forward progress: both launches must be resident together (nbm + (mb - 128) / 64 workgroups); the walker uses
// the form only for the last, chain-bound waves, where the chip is mostly idle, and every poll is bounded.
// ------------------------------------------------------------------------------
constexpr int FLOW_SPIN = 1 << 25;
#ifndef FLOW_POLL_SLEEP
#define FLOW_POLL_SLEEP 16  // x 64 cycles between two polls of a wave: hundreds of waves poll a handful of lines
#endif
__host__ __device__ inline int flow_lines(int nbm, int block_rows) { return 1 + nbm + block_rows * nbm; }

// one lane polls; false: aborted (or gave up: then it raises the abort word itself and reports INT_MAX - 1)
__device__ __forceinline__ bool flow_wait(const int *ctr, int target, int *fc, int *info, int fences) {
  int ok = 0;
  if ((threadIdx.x & 63) == 0) {
    int i = 0;
    for (; i < FLOW_SPIN; ++i) {
      if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
        ok = 1;
        break;
      }
      if ((i & 7) == 7 && __hip_atomic_load(fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(FLOW_POLL_SLEEP);
    }
    if (i == FLOW_SPIN) {
      atomicExch(info, 0x7ffffffe);
      __hip_atomic_store(fc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  ok = __builtin_amdgcn_readfirstlane(ok);
  if (fences) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("" ::: "memory");
  return ok != 0;
}
// the calling wave's stores have landed; then (fences) an agent-scope release
__device__ __forceinline__ void flow_drain(int fences) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (fences) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// idx-th lower 16 x 16 block of the 8 x 8 block grid, column by column
__host__ __device__ constexpr int flow_blk_c(int idx) {
  int c = 0, len = DB_NP;
  while (idx >= len) {
    idx -= len;
    --len;
    ++c;
  }
  return c;
}
__host__ __device__ constexpr int flow_blk_r(int idx) {
  int c = 0, len = DB_NP;
  while (idx >= len) {
    idx -= len;
    --len;
    ++c;
  }
  return c + idx;
}

template <typename T, int W>
__device__ __forceinline__ void flow_rank16(typename Tr<T>::acc_t (&acc)[9], const T (&xf)[8][4], const T (&nxf)[8][4]) {
  static_for<0, 9>([&](auto Q) {
    constexpr int q = decltype(Q)::value, idx = W + 4 * q, bi = flow_blk_r(idx), bj = flow_blk_c(idx);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q] = Tr<T>::mfma(nxf[bi][r], xf[bj][r], acc[q]);
  });
}
template <typename T, int W>
__device__ __forceinline__ void flow_fold(const typename Tr<T>::acc_t (&acc)[9], T *S) {
  const int lane = threadIdx.x & 63, lo = lane & 15;
  static_for<0, 9>([&](auto Q) {
    constexpr int q = decltype(Q)::value, idx = W + 4 * q, bi = flow_blk_r(idx), bj = flow_blk_c(idx);
    T *Cb = S + db_off(bi, bj);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Cb[Tr<T>::drow(lane, reg) + lo * DB_LD] += acc[q][reg];
  });
}

// D_s (in L.S) -= sum_{k<s} X(s,k) X(s,k)^T, a 16-column panel of X at a time; false: aborted
template <typename T>
__device__ __forceinline__ bool flow_accumulate(const T *tile, int mb, int nbm, int s, int *fc, int *info, int fences,
                                                DiagLds<T> &L) {
  using acc_t = typename Tr<T>::acc_t;
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  acc_t acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = T(0);
  bool ok = true;
  for (int k = 0; k < s && ok; ++k) {
    const int *hp = fc + 32 * (1 + nbm + s * nbm + k);
    if (k < s - 1) ok = flow_wait(hp + 7, 8, fc, info, fences);  // an earlier step: complete long ago
    for (int p = 0; p < DB_NP && ok; ++p) {
      if (k == s - 1) ok = flow_wait(hp + p, 8, fc, info, fences);  // the last step: as its panels are published
      if (!ok) break;
      // the panel as MFMA fragments, the same registers for both operands: X[16 i + lo][16 p + 4 r + hi]
      T xf[8][4], nxf[8][4];
      const T *X = tile + (size_t)(MACRO * s + lo) + (size_t)(MACRO * k + 16 * p + hi) * mb;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) xf[i][r] = load_sc1(X + 16 * i + (size_t)(4 * r) * mb);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) nxf[i][r] = -xf[i][r];
      switch (w) {
        case 0: flow_rank16<T, 0>(acc, xf, nxf); break;
        case 1: flow_rank16<T, 1>(acc, xf, nxf); break;
        case 2: flow_rank16<T, 2>(acc, xf, nxf); break;
        default: flow_rank16<T, 3>(acc, xf, nxf); break;
      }
    }
  }
  if (ok) {
    switch (w) {
      case 0: flow_fold<T, 0>(acc, L.S); break;
      case 1: flow_fold<T, 1>(acc, L.S); break;
      case 2: flow_fold<T, 2>(acc, L.S); break;
      default: flow_fold<T, 3>(acc, L.S); break;
    }
  } else if (lane == 0) {
    L.failed = 3;
  }
  __syncthreads();
  return L.failed == 0;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void k_flow_factor(T *tile, int mb, int nbm, T *winv, int *info, int info_base,
                                                     int *fc, int *ytab, const int *wait_sem, int wait_target,
                                                     int *dsem, int fences, unsigned long long *dbg) {
  __shared__ DiagLds<T> L;
  __shared__ unsigned long long slot_s;
  const int s = blockIdx.x;
  sem_wait(wait_sem, wait_target, info);  // the previous wave's last SYRK slice (or null: the stream's order)
  __builtin_amdgcn_s_setprio(3);
  unsigned long long t0 = 0;
  unsigned long long *ph = nullptr;
  if (dbg) {
    t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) slot_s = atomicAdd(dbg, 1ull);
    __syncthreads();
    if (slot_s < 1000) ph = dbg + 1 + 8 * slot_s + 2;
  }
  T *A = tile + (size_t)s * MACRO * (mb + 1);
  {  // the block as the previous waves' updates left it -> L.S
    const int t = threadIdx.x;
    load_block_lower<T>(A, mb, L.S);
    if (t < 16) L.Lrinv[t] = T(0);
    if (t == 0) L.failed = 0;
    __syncthreads();
  }
  bool live = true;
  if (s > 0) live = flow_accumulate<T>(tile, mb, nbm, s, fc, info, fences, L);
  if (live) {
    // a guest of its CU only while it factors (as a diagonal-block launch is for its lifetime): an update workgroup
    // beside it sleeps for these ~40 us, not for the hundreds this workgroup spends waiting for the earlier steps
    GuestOnCu guest(ytab);
    FlowPub<T> fp;
    fp.fpan = fc + 32 * (1 + s);
    fp.abort = fc;
    potrf_diag_body<T, true>(A, mb, winv + (size_t)s * MACRO * MACRO, info, info_base + s * MACRO, 1, L, ph, &fp);
    guest.leave();
  }
  if (ph && threadIdx.x == 0) {
    ph[-2] = t0;
    ph[-1] = __builtin_amdgcn_s_memrealtime();
  }
  sem_signal(dsem + 32 * s);  // (also when aborted: the TRSM step on the other stream polls it)
}

template <typename T>
__global__ __launch_bounds__(256) void k_flow_rows(T *tile, int mb, int nbm, const T *winv, int *info, int *fc, int *ytab,
                                                 const int *wait_sem, int wait_target, int *isem, int fences,
                                                 unsigned long long *dbg) {
  using acc_t = typename Tr<T>::acc_t;
  const unsigned long long t_in = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
  sem_wait(wait_sem, wait_target, info);
  GuestOnCu guest(ytab);
  __builtin_amdgcn_s_setprio(2);
  const int lane = threadIdx.x & 63, lo = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int R0 = MACRO + 64 * (int)blockIdx.x + 16 * w, rb = R0 >> 7;  // first row of the slab, its block row
  // column offsets of this lane's four accumulator registers inside a 16 x 16 block (x ld)
  size_t dcol[4], dcolw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dcol[r] = (size_t)Tr<T>::drow(lane, r) * mb, dcolw[r] = (size_t)Tr<T>::drow(lane, r) * MACRO;
  acc_t a[8], nxp[8];
  int signalled = 0;  // steps whose I counter this wave has raised
  bool live = true;
  for (int s = 0; s < rb && live; ++s) {
    T *Ab = tile + (size_t)(R0 + lo) + (size_t)(MACRO * s) * mb;  // my rows of block (rb, s)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) a[c][r] = Ab[(size_t)(16 * c) * mb + dcol[r]];
    const T *Ws = winv + (size_t)s * MACRO * MACRO + lo;     // + (16 p) (1 + 128) + dcolw
    const T *Ls = tile + (size_t)s * MACRO * (mb + 1) + lo;  // diagonal block s: + 16 c + (16 p) mb + dcol
    const int *fpan = fc + 32 * (1 + s);
    int *hp = fc + 32 * (1 + nbm + rb * nbm + s);
    const bool critical = rb == s + 1;  // block row s+1: the next diagonal block waits for these panels one by one
    static_for<0, 8>([&](auto P) {
      constexpr int p = decltype(P)::value;
      if (!live) return;
      live = flow_wait(fpan, p + 1, fc, info, fences);
      if (!live) return;
      T wd[4], lf[8][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) wd[r] = load_sc1(Ws + (size_t)(16 * p) * (MACRO + 1) + dcolw[r]);
#pragma unroll
      for (int c = p + 1; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) lf[c][r] = load_sc1(Ls + 16 * c + (size_t)(16 * p) * mb + dcol[r]);
      acc_t x;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = T(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) x = Tr<T>::mfma(wd[r], a[p][r], x);  // X_p^T = W_p A_p^T
#pragma unroll
      for (int r = 0; r < 4; ++r) store_sc1(Ab + (size_t)(16 * p) * mb + dcol[r], x[r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) nxp[p][r] = -x[r];
#pragma unroll
      for (int c = p + 1; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[c] = Tr<T>::mfma(lf[c][r], nxp[p][r], a[c]);  // A_c -= X_p L(c,p)^T
      if (critical) {
        flow_drain(fences);
        if (lane == 0) __hip_atomic_fetch_add(hp + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    });
    if (!live) break;
    if (!critical) {
      flow_drain(fences);
      if (lane < 8) __hip_atomic_fetch_add(hp + lane, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_fetch_add(isem + 32 * s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    signalled = s + 1;
    // X(rb, s) onto my rows of the blocks (rb, c), s < c < rb: A[:, c] -= X(rb,s) X(c,s)^T, X(c,s) complete once the
    // eight waves of block row c have published their last panel.  The fragments of one 16-column output block (its
    // 8 x 4 operand elements) are requested together, those of the next block behind this one's MFMAs.
    for (int c = s + 1; c < rb && live; ++c) {
      live = flow_wait(fc + 32 * (1 + nbm + c * nbm + s) + 7, 8, fc, info, fences);
      if (!live) break;
      T *Cb = tile + (size_t)(R0 + lo) + (size_t)(MACRO * c) * mb;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[j][r] = Cb[(size_t)(16 * j) * mb + dcol[r]];
      const T *Xc = tile + (size_t)(MACRO * c + lo) + (size_t)(MACRO * s) * mb;  // block (c, s)
      T lf[2][8][4];
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) lf[0][p][r] = load_sc1(Xc + (size_t)(16 * p) * mb + dcol[r]);
      static_for<0, 8>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (j + 1 < 8) {
#pragma unroll
          for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) lf[(j + 1) & 1][p][r] = load_sc1(Xc + 16 * (j + 1) + (size_t)(16 * p) * mb + dcol[r]);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[j] = Tr<T>::mfma(lf[j & 1][p][r], nxp[p][r], a[j]);
      });
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cb[(size_t)(16 * j) * mb + dcol[r]] = a[j][r];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave reads the block back at step c)
    }
  }
  // aborted: the counters the panel TRSM's update on the other stream polls still have to come
  if (lane == 0)
    for (int s = signalled; s < rb; ++s) __hip_atomic_fetch_add(isem + 32 * s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  guest.leave();
  if (blockIdx.x == 0) dbg_mark(dbg, 1, rb, t_in);
}

// ------------------------------------------------------------------------------
// plgsy: the matrix CHAMELEON_dplgsy_Tile generates (v6_test.c:46).  Chameleon's published
// generator (coreblas core_dplgsy, from PLASMA) is a 64-bit LCG ran <- a*ran + 1 addressed by
// jump-ahead: entry (i, j), i >= j, of an order-bigM matrix is 0.5 - ran_n * 2^-64 with ran_n
// the state n = i + j*bigM steps after the seed; symmetric; bump added on the diagonal.
// Bit for bit oracle/chol_oracle.c:orc_plgsy_entry, which the reference's recorded rel_error
// values pin (tests/golden/reference_vm_rel_error.json).  The jump composes the affine maps
// x -> a^(2^k) x + c_k of the set bits of n; the 64 maps are compile-time constants.
// ------------------------------------------------------------------------------
struct LcgTab {
  uint64_t a[64], c[64];
};
constexpr LcgTab make_lcg_tab() {
  LcgTab t{};
  uint64_t a = 6364136223846793005ULL, c = 1ULL;
  for (int k = 0; k < 64; ++k) {
    t.a[k] = a;
    t.c[k] = c;
    c *= (a + 1);
    a *= a;
  }
  return t;
}
__constant__ LcgTab c_lcg = make_lcg_tab();

__device__ __forceinline__ uint64_t lcg_jump(uint64_t n, uint64_t seed) {
  uint64_t ran = seed;
  for (int k = 0; n; n >>= 1, ++k)
    if (n & 1) ran = c_lcg.a[k] * ran + c_lcg.c[k];
  return ran;
}
__device__ __forceinline__ double plgsy_entry(double bump, uint64_t seed, long bigM, long i, long j) {
  const uint64_t lo = (uint64_t)(i < j ? i : j), hi = (uint64_t)(i < j ? j : i);
  const uint64_t ran = lcg_jump(hi + lo * (uint64_t)bigM, seed);
  const double v = 0.5 - (double)ran * 5.4210108624275222e-20;
  return (i == j) ? v + bump : v;
}

// mbu = the caller's tile edge, nglob = matrix order.  The stored tile edge A.mb may be
// larger (rounded up to 128) and the last tile row/column may be ragged: positions outside
// the matrix hold the identity (1 on the diagonal of diagonal tiles, 0 elsewhere), which
// leaves the factor of the real part untouched.
// side: 0 = every tile (ChamUpperLower), 1 = tiles on or below the diagonal (ChamLower),
// 2 = on or above (ChamUpper); diagonal tiles are always generated in full, the tiles on
// the other side are left as they are -- Chameleon's rule.
template <typename T>
__global__ __launch_bounds__(256) void k_plgsy(LocalMat A, int lnt, int prow, int pcol, double bump,
                                               unsigned long long seed, int mbu, long nglob, int side) {
  const long per_tile = A.bsiz;
  const long total = (long)A.lmt * lnt * per_tile;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long tl = idx / per_tile, e = idx - tl * per_tile;
    const int il = (int)(tl % A.lmt), jl = (int)(tl / A.lmt);
    const int ii = (int)(e % A.mb), jj = (int)(e / A.mb);
    const long I = (long)il * A.P + prow, J = (long)jl * A.Q + pcol;
    if ((side == 1 && I < J) || (side == 2 && I > J)) continue;
    const long gi = I * mbu + ii, gj = J * mbu + jj;
    T v;
    if (ii < mbu && jj < mbu && gi < nglob && gj < nglob)
      v = (T)plgsy_entry(bump, seed, nglob, gi, gj);
    else
      v = (I == J && ii == jj) ? T(1) : T(0);
    reinterpret_cast<T *>(A.base)[idx] = v;
  }
}

// ------------------------------------------------------------------------------
// Residual: for every 128x128 block on or below the diagonal,
//   R = sum_{kt <= j} tril?(L(i,kt)) tril?(L(j,kt))^T - A(i,j),  A regenerated.
// ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_residual(const T *__restrict__ L, int Nb, int mb, int nbm,
                                                     double bump, unsigned long long seed,
                                                     double *acc_out, int mbu, long nglob,
                                                     double *rowsum) {
  __shared__ Smem<T> sm;
  __shared__ double red[2][4];
  __shared__ double rs[4][MACRO];  // |R| and |A| sums: by block row (0,1) and by block column (2,3)
  const int MT = nbm * nbm;
  const int tix = blockIdx.x / MT, macro = blockIdx.x % MT;
  // tix -> (i >= j), row-major over the lower triangle
  int i = (int)((sqrt(8.0 * (double)tix + 1.0) - 1.0) * 0.5);
  while ((long)i * (i + 1) / 2 > tix) --i;
  while ((long)(i + 1) * (i + 2) / 2 <= tix) ++i;
  const int j = tix - i * (i + 1) / 2;
  const int mi = macro % nbm, mj = macro / nbm;
  if (i == j && mi < mj) return;
  const long bsiz = (long)mb * mb;
  Acc<T> acc;
  acc_zero<T>(acc);
  for (int kt = 0; kt <= j; ++kt) {
    const T *Ap = L + ((long)i + (long)kt * Nb) * bsiz + mi * MACRO;
    const T *Bp = L + ((long)j + (long)kt * Nb) * bsiz + mj * MACRO;
    if (kt < j)
      nt_kloop<T, false, false>(Ap, mb, Bp, mb, mb, acc, sm, 0, 0);
    else if (i != j)
      nt_kloop<T, false, true>(Ap, mb, Bp, mb, mb, acc, sm, 0, mj * MACRO);
    else
      nt_kloop<T, true, true>(Ap, mb, Bp, mb, mb, acc, sm, mi * MACRO, mj * MACRO);
  }
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w & 1, wc = w >> 1;
  double num = 0.0, den = 0.0;
  if (rowsum) {
    for (int q = t; q < 4 * MACRO; q += 256) (&rs[0][0])[q] = 0.0;
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = mi * MACRO + wr * 64 + a * 16 + (lane & 15);
        const int jj = mj * MACRO + wc * 64 + b * 16 + Tr<T>::drow(lane, r);
        const long gi = (long)i * mbu + ii, gj = (long)j * mbu + jj;
        if (ii >= mbu || jj >= mbu || gi >= nglob || gj >= nglob) continue;  // padding
        if (gi < gj) continue;
        const double aij = (double)(T)plgsy_entry(bump, seed, nglob, gi, gj);
        const double d = (double)acc[a][b][r] - aij;
        const double wgt = (gi == gj) ? 1.0 : 2.0;
        num += wgt * d * d;
        den += wgt * aij * aij;
        if (rowsum) {  // infinity norm: row sums of |R| and |A|; (gi,gj), gi > gj, also stands for (gj,gi)
          const int rl = ii - mi * MACRO, cl = jj - mj * MACRO;
          atomicAdd(&rs[0][rl], fabs(d));
          atomicAdd(&rs[1][rl], fabs(aij));
          if (gi != gj) {
            atomicAdd(&rs[2][cl], fabs(d));
            atomicAdd(&rs[3][cl], fabs(aij));
          }
        }
      }
  if (rowsum) {
    __syncthreads();
    for (int q = t; q < MACRO; q += 256) {
      const long gr = (long)i * mbu + mi * MACRO + q, gc = (long)j * mbu + mj * MACRO + q;
      if (mi * MACRO + q < mbu && gr < nglob) {
        atomicAdd(&rowsum[gr], rs[0][q]);
        atomicAdd(&rowsum[nglob + gr], rs[1][q]);
      }
      if (mj * MACRO + q < mbu && gc < nglob) {
        atomicAdd(&rowsum[gc], rs[2][q]);
        atomicAdd(&rowsum[nglob + gc], rs[3][q]);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    num += __shfl_down(num, o, 64);
    den += __shfl_down(den, o, 64);
  }
  if (lane == 0) {
    red[0][w] = num;
    red[1][w] = den;
  }
  __syncthreads();
  if (t == 0) {
    atomicAdd(&acc_out[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&acc_out[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// ------------------------------------------------------------------------------
// In-place transpose of a whole tile-layout matrix (single process): tile (I,J) <-> tile
// (J,I), both transposed, 64x64 sub-blocks through LDS.  Used to run ChamUpper through the
// Lower kernels: A = U^T U with U = L^T, and since the Lower path never touches strictly-upper
// storage, transposing back leaves the caller's strict lower triangle exactly as it was.
// ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_transpose_inplace(T *M, int nt, int mb) {
  __shared__ T sa[64][65], sb[64][65];
  const int nsub = mb / 64;
  // blockIdx.x enumerates tile pairs I >= J (row-major over the lower triangle) x sub-block pairs
  const long per_pair = (long)nsub * nsub;
  const long pair = blockIdx.x / per_pair, sub = blockIdx.x % per_pair;
  int I = (int)((sqrt(8.0 * (double)pair + 1.0) - 1.0) * 0.5);
  while ((long)I * (I + 1) / 2 > pair) --I;
  while ((long)(I + 1) * (I + 2) / 2 <= pair) ++I;
  const int J = (int)(pair - (long)I * (I + 1) / 2);
  const int si = (int)(sub % nsub), sj = (int)(sub / nsub);
  if (I == J && si < sj) return;  // the mirror sub-block of a diagonal tile is handled by (sj, si)
  const long bsiz = (long)mb * mb;
  T *ta = M + ((long)I + (long)J * nt) * bsiz + si * 64 + (long)sj * 64 * mb;  // sub-block (si,sj) of (I,J)
  T *tb = M + ((long)J + (long)I * nt) * bsiz + sj * 64 + (long)si * 64 * mb;  // sub-block (sj,si) of (J,I)
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) {
    sa[c][tx] = ta[tx + (long)c * mb];
    sb[c][tx] = tb[tx + (long)c * mb];
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    ta[tx + (long)c * mb] = sb[tx][c];
    tb[tx + (long)c * mb] = sa[tx][c];
  }
}

template <typename T>
void launch_transpose_inplace(hipStream_t s, T *M, int nt, int mb) {
  const long pairs = (long)nt * (nt + 1) / 2, nsub = mb / 64;
  k_transpose_inplace<T><<<dim3((unsigned)(pairs * nsub * nsub)), 256, 0, s>>>(M, nt, mb);
}
template void launch_transpose_inplace<double>(hipStream_t, double *, int, int);
template void launch_transpose_inplace<float>(hipStream_t, float *, int, int);

template <typename T>
__global__ void k_pad_identity(T *dst, int n, int ldp) {
  const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ldp) dst[i + (size_t)i * ldp] = T(1);
}

// ------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------
// chol_init's second check: a dispatch that cannot place all its workgroups at once (an update launch of more
// than one round) keeps the dispatcher of its hardware queue's pipe busy until the last of them is placed, and the
// kernels of another stream that shares that pipe do not start meanwhile (round 4: workgroups of a four-workgroup
// launch on the flow's stream seen to start 30 us apart; scripts/exp/pipe_probe.hip).  `big`: four rounds of
// workgroups (two per CU by LDS) that spin ~20 us each; `small`: one wave, launched right behind it on the other
// stream, stamps its start.  t[0] = first workgroup of the big launch, t[1] = the small kernel.
__global__ __launch_bounds__(256) void k_pipe_big(unsigned long long *t, int spin_ticks) {
  __shared__ char lds[65536];
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(16);
  if (lds[(threadIdx.x + 1) & 255] == 7) t[2] = 0;
}
__global__ void k_pipe_small(unsigned long long *t) {
  if (threadIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
}
void launch_pipe_probe(hipStream_t big, hipStream_t small, unsigned long long *t, int cus) {
  k_pipe_big<<<8 * cus, 256, 0, big>>>(t, 2000);
  k_pipe_small<<<1, 64, 0, small>>>(t);
}

void launch_sem_probe(hipStream_t consumer, hipStream_t producer, int *sem, int *result) {
  k_sem_probe_wait<<<1, 1, 0, consumer>>>(sem, result);
  k_sem_probe_set<<<1, 64, 0, producer>>>(sem);
}

int *g_ytab = nullptr;                // per-CU yield requests (cooperative CU hand-over), may be null
unsigned long long *g_dbg = nullptr;  // diagnostic stamp buffer (chol_debug_stamps)
int g_min_units = 128;  // a launch is dealt in units small enough to give at least this many (CHOLMI_MIN_UNITS)
int g_intile_fused = 1;  // chain-bound form: an in-tile step's solve and update in one launch (CHOLMI_INTILE_FUSED=0: two)
// ... while the step has at most this many polling update workgroups: nr (2 nr + 1), i.e. tiles up to 1024 + 128 x 3
// (nr <= 10); beyond, resident pollers would queue for CU slots behind each other
constexpr int INTILE_FUSED_MAX = 256;
bool flow_applies(int nbm) { return g_flow && nbm >= g_flow_min_nbm && nbm <= g_flow_max_nbm; }
int g_flow = 1;          // chain-bound waves: the tile POTRF as a flow of polling workgroups (CHOLMI_FLOW=0: diagonal-block + in-tile step launches)
int g_flow_min_nbm = 3, g_flow_max_nbm = 4;  // ... for tiles of this many 128-blocks (CHOLMI_FLOW_NBM=lo:hi): measured round 4, all waves
                                             // in flow form against none -- tile 512: +14 ... +17 % (N = 1536 ... 4096), +5 % (N = 5120, 6144); tile 256: -8 %
                                             // (a two-block tile has one in-tile step to save); tile 1024: -6 ... -12 % (the row waves' products pile up: 21 for the
                                             // last block row)
int g_flow_fences = 0;   // ... 1: agent-scope release / acquire around every hand-off of the flow (CHOLMI_FLOW_FENCES; diagnostic)
int g_poll_max_wgs = 48;  // grids up to this many workgroups poll their counter themselves, larger ones behind a gate (CHOLMI_POLL_MAX_WGS)
int g_trsm_small_max = 64;  // panel TRSM steps in small-block form up to this many 128-row blocks (CHOLMI_TRSM_SMALL_MAX)

// blocks dealt to an XCD at a time (map_update_block) and the padded block counts of the two segments of a launch
struct UpdateGrid {
  int unit;
  long blocks_a, blocks_b;
};
inline UpdateGrid update_grid(long tot_a, long tot_b, bool single_diag) {
  UpdateGrid u;
  u.unit = 64;
  while (u.unit > 8 && (tot_a + tot_b) / u.unit < g_min_units) u.unit >>= 1;
  u.blocks_a = ((tot_a + u.unit - 1) / u.unit + 7) / 8 * 8 * u.unit;
  u.blocks_b = ((tot_b + u.unit - 1) / u.unit + 7) / 8 * 8 * u.unit;
  if (single_diag) u.blocks_a = 0, u.blocks_b = tot_b;  // single diagonal tile: spread, no padding
  return u;
}

template <typename T>
void launch_trail_update(hipStream_t s, const LocalMat &C, const int2 *d_list, int off, int na, int offb,
                         int nb, const PanelRef &pan, bool yield, const PanelRef *pan2) {
  if (na + nb <= 0) return;
  const int npan = pan2 ? 2 : 1;
  const PanelRef &p2 = pan2 ? *pan2 : pan;
  offb -= off;  // the kernels index from d_list + off
  const int nbm = C.mb / MACRO, MT = nbm * nbm, MTd = nbm * (nbm + 1) / 2;
  const UpdateGrid u = update_grid((long)na * MT, (long)nb * MTd, na == 0 && nb == 1);
  const dim3 grid((unsigned)(u.blocks_a + u.blocks_b));
  if constexpr (sizeof(T) == 4)
    k_trail_update_w8f<<<grid, dim3(512), 0, s>>>(C, d_list + off, na, offb, nb, (int)u.blocks_a, pan, nbm, u.unit,
                                                 yield ? g_ytab : nullptr, p2, npan);
  else
    k_trail_update_w8<T, 3><<<grid, dim3(512), 0, s>>>(C, d_list + off, na, offb, nb, (int)u.blocks_a, pan, nbm, u.unit,
                                                       yield ? g_ytab : nullptr, p2, npan);
}

// One 128-column step of the panel TRSM over `ntiles` tiles: X[:, st] = A[:, st] Winv_st^T, then
// A[:, c] -= X[:, st] L(c, st)^T for the block columns c > st.  Few tiles (the late, chain-bound
// waves): small-block kernels, latency; many tiles: the 128 x 128 NT core, throughput.
// Device-side edges of one step (sem_wait), all optional: the solve polls `diag` (the diagonal-block step
// that produced Winv_st; target 1), the update polls `intile` (the workgroups of the in-tile solve that
// produced L(c, st), c > st), and the solve's workgroups of the FIRST tile count up `head`.
struct StepSems {
  const int *diag = nullptr;
  const int *intile = nullptr;
  int intile_target = 0;
  int *head = nullptr;
  int *fail = nullptr;
};
// (returns how many workgroups of the solve belong to the first tile: what `head` counts up to)
template <typename T>
int trsm_step(hipStream_t s, T *tiles, long bsiz, int ntiles, const T *lkk, const T *winv, int mb, int st,
              T alpha, const StepSems &sm = StepSems()) {
  const int nbm = mb / MACRO, nc = nbm - 1 - st;
  if (alpha == T(1) && ntiles * nbm <= g_trsm_small_max) {
    const bool p1 = poll_in_kernel(s, sm.diag, 1, sm.fail, (long)ntiles * nbm * 4);
    k_solve_small<T><<<ntiles * nbm * 4, 256, 0, s>>>(tiles, bsiz, mb, nbm, 0, st, winv, T(1), g_ytab,
                                                      p1 ? sm.diag : nullptr, 1, sm.fail, sm.head);
    if (nc > 0) {
      const bool p2 = poll_in_kernel(s, sm.intile, sm.intile_target, sm.fail, 4L * nbm * nc * ntiles);
      k_small_update<T><<<dim3(2 * nbm, 2 * nc, ntiles), 256, 0, s>>>(
          tiles + (long)(st + 1) * MACRO * mb, mb, tiles + (long)st * MACRO * mb,
          lkk + (long)(st + 1) * MACRO + (long)st * MACRO * mb, mb, MACRO, g_ytab, 1, bsiz, bsiz, nullptr,
          p2 ? sm.intile : nullptr, sm.intile_target, sm.fail);
    }
    return 4 * nbm;
  }
  // alpha is applied once to every column block: in the solve of block 0 and as the beta of the
  // first update of blocks > 0
  const bool p1 = poll_in_kernel(s, sm.diag, 1, sm.fail, (long)ntiles * nbm);
  k_panel_solve<T><<<ntiles * nbm, 256, 0, s>>>(tiles, bsiz, mb, nbm, 0, st, winv, st == 0 ? alpha : T(1), g_ytab,
                                                p1 ? sm.diag : nullptr, 1, sm.fail, sm.head);
  if (nc > 0) {
    const bool p2 = poll_in_kernel(s, sm.intile, sm.intile_target, sm.fail, (long)ntiles * nbm * nc);
    k_panel_update<T><<<ntiles * nbm * nc, 256, 0, s>>>(tiles, bsiz, mb, nbm, st, lkk, st == 0 ? alpha : T(1), g_ytab,
                                                        p2 ? sm.intile : nullptr, sm.intile_target, sm.fail);
  }
  return nbm;
}

// Column k+1 below its diagonal tile in the LATENCY form: C_i -= A_i B^T, i = 0 .. ntiles-1 (tiles bsiz apart, B the head
// tile), in 64 x 64 blocks -- (mb / 64)^2 workgroups per tile instead of the update kernel's (mb / 128)^2 or fewer.  A
// third of that kernel's rate per CU, but a chain-bound wave has the CUs to spare and waits for exactly this launch.
template <typename T>
void launch_col_update_small(hipStream_t s, T *C, const T *A, const T *B, int mb, int ntiles, long bsiz, int *done) {
  // (done, may be null: raised once by each of the (mb / 64)^2 ntiles workgroups)
  if (ntiles > 0) k_small_update<T><<<dim3(mb / 64, mb / 64, ntiles), 256, 0, s>>>(C, mb, A, B, mb, mb, g_ytab, 1, bsiz, bsiz, done);
}
template void launch_col_update_small<double>(hipStream_t, double *, const double *, const double *, int, int, long, int *);
template void launch_col_update_small<float>(hipStream_t, float *, const float *, const float *, int, int, long, int *);

// C(mb x mb, lower) -= A A^T with A one mb x mb tile: the SYRK that releases the next POTRF
template <typename T>
void launch_diag_syrk(hipStream_t s, T *C, const T *A, int mb) {
  k_small_update<T><<<dim3(mb / 64, mb / 64), 256, 0, s>>>(C, mb, A, A, mb, mb, g_ytab, 0, 0, 0);
}

// sem (may be null): mb / 128 counters, 32 ints apart, for the fused in-tile steps (k_intile_step); zeroed here
template <typename T>
void launch_potrf_tile(hipStream_t s, T *tile, int mb, T *winv, int *d_info, int info_base, int *sem) {
  const int nbm = mb / MACRO;
  const bool fused_steps = sem && g_intile_fused && nbm > 1;
  if (fused_steps) (void)hipMemsetAsync(sem, 0, (size_t)nbm * 32 * sizeof(int), s);
  for (int st = 0; st < nbm; ++st) {
    k_potrf_diag<T><<<1, 256, 0, s>>>(tile + (long)st * MACRO * (mb + 1), mb,
                                      winv + (long)st * MACRO * MACRO, d_info, info_base + st * MACRO, 1, g_dbg, g_ytab);
    const int nr = nbm - 1 - st;
    if (nr > 0) {
      // the in-tile POTRF steps are a handful of workgroups on the critical path: guests
      if (fused_steps && nr * (2 * nr + 1) <= INTILE_FUSED_MAX) {
        k_intile_step<T><<<4 * nr + nr * (2 * nr + 1), 256, 0, s>>>(tile, mb, nbm, st, winv, g_ytab, sem + 32 * st, d_info);
      } else {
        k_solve_small<T><<<4 * nr, 256, 0, s>>>(tile, 0, mb, nbm, st + 1, st, winv, T(1), g_ytab);
        T *tr = tile + (long)(st + 1) * MACRO * (mb + 1);  // trailing part of the tile
        const T *xs = tile + (long)(st + 1) * MACRO + (long)st * MACRO * mb;  // block column st below the diagonal
        k_small_update<T><<<dim3(2 * nr, 2 * nr), 256, 0, s>>>(tr, mb, xs, xs, mb, MACRO, g_ytab, 0, 0, 0);
      }
    }
  }
}

// POTRF of the diagonal tile and TRSM of the panel tiles below it, pipelined over two streams:
// TRSM step s needs only Winv_s and the blocks L(c,s), c > s, of the diagonal tile, which exist
// as soon as in-tile step s is done -- so it runs on `st` while the POTRF goes on with step s+1
// on `sp`.  The chain of a wave shrinks from POTRF + TRSM to about POTRF + one TRSM step.
// ev_head (may be null) is recorded on `st` when the panel is solved.
// ev: nbm events.  Both streams must be joined by the caller.
// sy != null: the counter-linked form of a chain-bound wave (SyrkPipe) -- the same launches, plus the SYRK
// on the next diagonal tile in K = 128 slices on sy->su, every dependency between the streams a polled
// counter instead of an event; wait_sem / wait_target: what the first diagonal-block step polls (the
// previous wave's last slice), or null.
template <typename T>
void launch_panel_pipelined(hipStream_t sp, hipStream_t st, hipEvent_t *ev, T *lkk, int mb, T *winv,
                            int *d_info, int info_base, T *tiles, long bsiz, int ntiles, hipEvent_t ev_head,
                            const SyrkPipe *sy, const int *wait_sem, int wait_target, int *tile_sem) {
  const int nbm = mb / MACRO;
  const bool pipe = sy && ntiles > 0;
  // event-linked form with tile_sem (mb / 128 counters, 32 ints apart): the in-tile steps fused as in launch_potrf_tile
  const bool fused_plain = !pipe && tile_sem && g_intile_fused && nbm > 1;
  if (fused_plain) (void)hipMemsetAsync(tile_sem, 0, (size_t)nbm * 32 * sizeof(int), sp);
  // pipe: counters of this wave, one 128-byte slot each -- D[s] the diagonal-block step s, I[s] the in-tile
  // solve of step s (4 nr workgroups), H[s] the head tile's workgroups of TRSM step s, then `done`
  auto slot = [&](int i) { return sy->sem + 32 * i; };
  // the tile POTRF as a flow (k_flow_factor / k_flow_rows): the diagonal-block steps raise D[s] as before, the
  // row-slab waves I[s] (8 (nbm - 1 - s) of them instead of the 4 (nbm - 1 - s) workgroups of the in-tile solve)
  const bool flow = pipe && sy->fc && sy->sflow && flow_applies(nbm);
  if (flow) {
    if (sy->join_flow && sy->ev_flow) {  // the flow stream joins the POTRF stream's order (without it the row-slab kernel
                                         // would sit on the chip, polling, from the moment the host issues it -- and read
                                         // its rows of the tile before the updates of the waves before have written them)
      (void)hipEventRecord(sy->ev_flow, sp);
      (void)hipStreamWaitEvent(sy->sflow, sy->ev_flow, 0);
    }
    k_flow_factor<T><<<nbm, 256, 0, sp>>>(lkk, mb, nbm, winv, d_info, info_base, sy->fc, g_ytab, wait_sem, wait_target,
                                          slot(0), g_flow_fences, g_dbg);
    k_flow_rows<T><<<(mb - MACRO) / 64, 256, 0, sy->sflow>>>(lkk, mb, nbm, winv, d_info, sy->fc, g_ytab, wait_sem, wait_target,
                                                           slot(nbm), g_flow_fences, g_dbg);
  }
  for (int s = 0; s < nbm; ++s) {
    const int nr = nbm - 1 - s;
    if (!flow)
      k_potrf_diag<T><<<1, 256, 0, sp>>>(lkk + (long)s * MACRO * (mb + 1), mb, winv + (long)s * MACRO * MACRO,
                                         d_info, info_base + s * MACRO, 1, g_dbg, g_ytab, s == 0 ? wait_sem : nullptr,
                                         wait_target, pipe ? slot(s) : nullptr);
    if (nr > 0 && !flow) {
      if ((pipe || fused_plain) && g_intile_fused && nr * (2 * nr + 1) <= INTILE_FUSED_MAX) {
        // solve and update of the step in one launch, the update's workgroups polling the solves' counter
        // (which the TRSM step's update on st polls too)
        k_intile_step<T><<<4 * nr + nr * (2 * nr + 1), 256, 0, sp>>>(lkk, mb, nbm, s, winv, g_ytab,
                                                                     pipe ? slot(nbm + s) : tile_sem + 32 * s, d_info);
      } else {
        k_solve_small<T><<<4 * nr, 256, 0, sp>>>(lkk, 0, mb, nbm, s + 1, s, winv, T(1), g_ytab, nullptr, 0, nullptr,
                                                 pipe ? slot(nbm + s) : nullptr);
        T *tr = lkk + (long)(s + 1) * MACRO * (mb + 1);
        const T *xs = lkk + (long)(s + 1) * MACRO + (long)s * MACRO * mb;
        // (launching the update ahead of time on another stream, polling the in-tile solve's counter, the
        // next diagonal-block step polling its own: -1 ... -5 % on st, -10 ... -20 % on su -- it queues
        // behind that stream's own launches)
        k_small_update<T><<<dim3(2 * nr, 2 * nr), 256, 0, sp>>>(tr, mb, xs, xs, mb, MACRO, g_ytab, 0, 0, 0);
      }
    }
    if (ntiles <= 0) continue;
    if (pipe) {
      // No stream operation anywhere on the chain: TRSM step s (on st) and slice s of the SYRK on tile
      // (k+1,k+1) (on su; X = the head tile, C -= X_s X_s^T) are launched ahead of time and poll.
      StepSems ss;
      ss.diag = slot(s);
      ss.intile = slot(nbm + s);
      ss.intile_target = (flow ? 8 : 4) * nr;
      ss.head = sy->c ? slot(2 * nbm + s) : nullptr;
      ss.fail = d_info;
      const int head_wgs = trsm_step<T>(st, tiles, bsiz, ntiles, lkk, winv, mb, s, T(1), ss);
      if (!sy->c) continue;  // (a grid: the POTRF -> TRSM edge alone runs on counters, the next diagonal tile is elsewhere)
      const T *xs = tiles + (long)s * MACRO * mb;
      const int n64 = mb / 64;
      const long nslice = (long)n64 * (n64 + 1) / 2;
      int *done = s == nbm - 1 ? slot(3 * nbm) : nullptr;
      const bool ps = poll_in_kernel(sy->su, slot(2 * nbm + s), head_wgs, d_info, nslice);
      k_small_update<T><<<dim3(n64, n64), 256, 0, sy->su>>>(reinterpret_cast<T *>(sy->c), mb, xs, xs, mb, MACRO, g_ytab, 0, 0,
                                                           0, done, ps ? slot(2 * nbm + s) : nullptr, head_wgs, d_info);
      continue;
    }
    // (recorded behind the in-tile update, not between the solve and the update: an event record
    // between two dependent launches of the chain costs it ~7 us, the TRSM step loses nothing)
    (void)hipEventRecord(ev[s], sp);
    (void)hipStreamWaitEvent(st, ev[s], 0);
    trsm_step<T>(st, tiles, bsiz, ntiles, lkk, winv, mb, s, T(1));
  }
  if (ev_head) (void)hipEventRecord(ev_head, st);
}
template void launch_panel_pipelined<double>(hipStream_t, hipStream_t, hipEvent_t *, double *, int, double *,
                                             int *, int, double *, long, int, hipEvent_t, const SyrkPipe *, const int *, int, int *);
template void launch_panel_pipelined<float>(hipStream_t, hipStream_t, hipEvent_t *, float *, int, float *, int *,
                                            int, float *, long, int, hipEvent_t, const SyrkPipe *, const int *, int, int *);

template <typename T>
void launch_invert_diag(hipStream_t s, const T *tile, int mb, T *winv) {
  const int nbm = mb / MACRO;
  for (int st = 0; st < nbm; ++st)
    k_potrf_diag<T><<<1, 256, 0, s>>>(const_cast<T *>(tile) + (long)st * MACRO * (mb + 1), mb,
                                      winv + (long)st * MACRO * MACRO, nullptr, 0, 0, nullptr, g_ytab);
}

template <typename T>
void launch_trsm_panel(hipStream_t s, T *tiles, long bsiz, int ntiles, const T *lkk, const T *winv,
                       int mb, T alpha) {
  if (ntiles <= 0) return;
  const int nbm = mb / MACRO;
  // the TRSM of the next panel outranks the trailing update whenever the update yields at all (the
  // walker enables that only while the panel chain is the critical path)
  for (int st = 0; st < nbm; ++st) trsm_step<T>(s, tiles, bsiz, ntiles, lkk, winv, mb, st, alpha);
}

template <typename T>
void launch_gemm_nt_tile(hipStream_t s, const T *A, const T *B, T *C, int mb, T alpha, T beta,
                         bool lower_only) {
  const int nbm = mb / MACRO;
  k_gemm_nt_tile<T><<<dim3(nbm, nbm), 256, 0, s>>>(A, B, C, mb, nbm, alpha, beta, lower_only ? 1 : 0, 1, 0, 0, 0, 0);
}

template <typename T>
void launch_gemm_nt_batch(hipStream_t s, const T *A, long sA, int nz1, const T *B, long sB, int nz2, T *C, long sC1,
                          long sC2, int mb, T alpha, T beta) {
  const int nbm = mb / MACRO;
  if (nz1 <= 0 || nz2 <= 0) return;
  // grid.z is a 16-bit quantity: whole z2-slices per launch (z = z1 + nz1 z2), a single slice in pieces of z1
  if (nz1 <= 65535) {
    const int per = std::max(1, 65535 / nz1);
    for (int z2 = 0; z2 < nz2; z2 += per) {
      const int c = std::min(per, nz2 - z2);
      k_gemm_nt_tile<T><<<dim3(nbm, nbm, nz1 * c), 256, 0, s>>>(A, B + (long)z2 * sB, C + (long)z2 * sC2, mb, nbm, alpha, beta, 0,
                                                               nz1, sA, sB, sC1, sC2);
    }
  } else {
    for (int z2 = 0; z2 < nz2; ++z2)
      for (int z1 = 0; z1 < nz1; z1 += 65535) {
        const int c = std::min(65535, nz1 - z1);
        k_gemm_nt_tile<T><<<dim3(nbm, nbm, c), 256, 0, s>>>(A + (long)z1 * sA, B + (long)z2 * sB, C + (long)z1 * sC1 + (long)z2 * sC2,
                                                          mb, nbm, alpha, beta, 0, c, sA, sB, sC1, sC2);
      }
  }
}

template <typename T>
void launch_update_ptrs(hipStream_t s, const T *const *cin, const T *const *a, const T *const *b, T *const *cout, int n, int mb,
                        bool yield) {
  if (n <= 0) return;
  const int nbm = mb / MACRO;
  const UpdateGrid u = update_grid((long)n * nbm * nbm, 0, false);
  const dim3 grid((unsigned)u.blocks_a);
  if constexpr (sizeof(T) == 4)
    k_update_ptrs_w8f<<<grid, dim3(512), 0, s>>>(cin, a, b, cout, n, mb, nbm, u.unit, yield ? g_ytab : nullptr);
  else
    k_update_ptrs_w8<T, 3><<<grid, dim3(512), 0, s>>>(cin, a, b, cout, n, mb, nbm, u.unit, yield ? g_ytab : nullptr);
}
template void launch_update_ptrs<double>(hipStream_t, const double *const *, const double *const *, const double *const *, double *const *, int, int, bool);
template void launch_update_ptrs<float>(hipStream_t, const float *const *, const float *const *, const float *const *, float *const *, int, int, bool);

void launch_copy_ptrs(hipStream_t s, const void *const *src, void *const *dst, int n, long bytes) {
  for (int z0 = 0; z0 < n; z0 += 65535) {
    const int nz = std::min(65535, n - z0);
    k_copy_ptrs<<<dim3((unsigned)std::max(1L, std::min(64L, bytes / (16 * 256 * 4))), nz), 256, 0, s>>>(src + z0, dst + z0, bytes);
  }
}

template <typename T>
void launch_plgsy(hipStream_t s, const LocalMat &A, int lnt, int prow, int pcol, double bump,
                  unsigned long long seed, int mbu, long nglob, int side) {
  k_plgsy<T><<<4096, 256, 0, s>>>(A, lnt, prow, pcol, bump, seed, mbu, nglob, side);
}

template <typename T>
void launch_residual(hipStream_t s, const T *Lbase, int Nb, int mb, double bump,
                     unsigned long long seed, double *d_acc, int mbu, long nglob, double *rowsum) {
  const int nbm = mb / MACRO;
  const long ntile = (long)Nb * (Nb + 1) / 2;
  k_residual<T><<<dim3((unsigned)(ntile * nbm * nbm)), 256, 0, s>>>(Lbase, Nb, mb, nbm, bump, seed, d_acc, mbu, nglob, rowsum);
}

template <typename T>
void launch_pad_identity(hipStream_t s, T *dst, int n, int ldp) {
  if (ldp > n) k_pad_identity<T><<<(ldp - n + 127) / 128, 128, 0, s>>>(dst, n, ldp);
}


// ------------------------------------------------------------------------------
// MFMA issue-rate probe: register-only 16x16x4 MFMA stream, 16 independent
// accumulators per wave, operands random-looking.  Gives the matrix-core ceiling this
// chip actually sustains (clock under load) beside the datasheet peak.
// ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void k_mfma_probe(T *out, int iters) {
  typename Tr<T>::acc_t acc[16];
  const int t = threadIdx.x + blockIdx.x * 256;
  T a = T((t * 2654435761u >> 8) & 0xffff) * T(1.0 / 65536) - T(0.5);
  T b = T((t * 40503u >> 4) & 0xffff) * T(1.0 / 65536) - T(0.5);
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[u][r] = T(u + r);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = Tr<T>::mfma(a, b, acc[u]);
    a = -a;  // keep the sums bounded
  }
  T s = T(0);
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += acc[u][r];
  out[t] = s;
}

template <typename T>
void launch_mfma_probe(hipStream_t s, T *out, int blocks, int iters) {
  k_mfma_probe<T><<<blocks, 256, 0, s>>>(out, iters);
}
template void launch_mfma_probe<double>(hipStream_t, double *, int, int);
template void launch_mfma_probe<float>(hipStream_t, float *, int, int);

#define INSTANTIATE(T)                                                                              \
  template void launch_trail_update<T>(hipStream_t, const LocalMat &, const int2 *, int, int, int,  \
                                       int, const PanelRef &, bool, const PanelRef *);              \
  template void launch_potrf_tile<T>(hipStream_t, T *, int, T *, int *, int, int *);                \
  template void launch_diag_syrk<T>(hipStream_t, T *, const T *, int);                               \
  template void launch_invert_diag<T>(hipStream_t, const T *, int, T *);                            \
  template void launch_trsm_panel<T>(hipStream_t, T *, long, int, const T *, const T *, int, T);    \
  template void launch_gemm_nt_tile<T>(hipStream_t, const T *, const T *, T *, int, T, T, bool);    \
  template void launch_gemm_nt_batch<T>(hipStream_t, const T *, long, int, const T *, long, int, T *, long, long, \
                                        int, T, T);                                                 \
  template void launch_plgsy<T>(hipStream_t, const LocalMat &, int, int, int, double,               \
                                unsigned long long, int, long, int);                                \
  template void launch_residual<T>(hipStream_t, const T *, int, int, double, unsigned long long,    \
                                   double *, int, long, double *);                                  \
  template void launch_pad_identity<T>(hipStream_t, T *, int, int);
INSTANTIATE(double)
INSTANTIATE(float)

}  // namespace cholmi
