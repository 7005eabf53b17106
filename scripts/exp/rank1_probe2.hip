// The decoupled pivot loop of phase A (kernels.hip, wave 0) in isolation: cycles per pivot of variants.
// Build: hipcc --offload-arch=gfx950 -O3 -o rank1_probe2 rank1_probe2.hip
// (with __launch_bounds__(64, 2) on k_piv the compiler picks VGPR-form MFMAs: 244 instead of 261 cycles per pivot, and WRONG
//  results -- the inline-asm v_cndmask then reads the MFMA's destination with no hazard wait: DESIGN.md section 9)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ double rlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double keep_lanes(double v, unsigned long long m) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int rl, rh;
  asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4" : "=&v"(rl), "=&v"(rh) : "v"(lo), "v"(hi), "s"(m));
  return __hiloint2double(rh, rl);
}
template <int OFF>
__device__ __forceinline__ void lds_store_lanes(unsigned addr, double v, unsigned long long m) {
  asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1 offset:%3\n\ts_mov_b64 exec, -1" : : "v"(addr), "v"(v), "s"(m), "n"(OFF) : "memory");
}
// VAR: 0 = as in the kernel; 1 = no LDS stores at all; 2 = column store unmasked (all lanes, own slots); 3 = u, d' pinned AHEAD of the vector work
template <int VAR>
__global__ __launch_bounds__(64) void k_piv(const double *A, double *Lout, unsigned long long *cyc, int reps) {
  const int lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
  __shared__ double Lc[16 * 17 + 64 * 16];
  __shared__ double Lr[16];
  __shared__ int Lf[16];
  typedef __attribute__((address_space(3))) double lds_t;
  unsigned long long c0 = 0, c1 = 0;
  int one = 1;
  asm volatile("" : "+v"(one));
  for (int rep = 0; rep < reps; ++rep) {
    d4_t E;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = hi + 4 * reg;
      E[reg] = -A[max(i, lo) + min(i, lo) * 16];
    }
    if (rep == reps - 1) c0 = __builtin_readcyclecounter();
    double dcur = -rlane(E[0], 0);
    asm volatile("" : "+v"(dcur));
    static_for<0, 16>([&](auto JJ) {
      constexpr int jj = decltype(JJ)::value;
      constexpr int R = jj / 4, K = jj % 4;
      constexpr unsigned long long M = (0xFFFFull & ~((2ull << jj) - 1ull)) << (16 * K);
      if constexpr (VAR == 6) {  // scalar chain + reads, NO MFMA (E constant)
        const double d = dcur;
        const double r = __builtin_amdgcn_rsq(d);
        const double dm = __builtin_fmin(d, 1.7976931348623157e308);
        double g = dm * r, h = 0.5 * r;
        double e = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, e, g);
        h = __builtin_fma(h, e, h);
        e = __builtin_fma(-h, g, 0.5);
        h = __builtin_fma(h, e, h);
        double rinvn = -h - h;
        if constexpr (jj + 1 < 16) {
          constexpr int R1 = (jj + 1) / 4, K1 = (jj + 1) % 4;
          const double as = rlane(E[R], 16 * K + jj + 1);
          const double xs = rlane(E[R1], 16 * K1 + jj + 1);
          const double u = as * rinvn * 1e-3;
          dcur = __builtin_fma(-u, u, -xs);
        }
        asm volatile("" : "+v"(dcur));
        if (jj == 15) Lr[lane & 15] = rinvn;
        return;
      }
      if constexpr (VAR == 7) {  // MFMA chain only: row -> select -> scale by a constant -> MFMA
        const double Em = keep_lanes(E[R], M);
        const double a = Em * 1e-3;
        E = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, E, 0, 0, 0);
        if (jj == 15) Lr[lane & 15] = E[0] + E[1] + E[2] + E[3];
        return;
      }
      if constexpr (VAR == 8) {  // scalar chain alone, no reads of E at all
        const double d = dcur;
        const double r = __builtin_amdgcn_rsq(d);
        const double dm = __builtin_fmin(d, 1.7976931348623157e308);
        double g = dm * r, h = 0.5 * r;
        double e = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, e, g);
        h = __builtin_fma(h, e, h);
        e = __builtin_fma(-h, g, 0.5);
        h = __builtin_fma(h, e, h);
        double rinvn = -h - h;
        const double u = rinvn * 1e-3;
        dcur = __builtin_fma(-u, u, d);
        asm volatile("" : "+v"(dcur));
        if (jj == 15) Lr[lane & 15] = rinvn;
        return;
      }
      if constexpr (VAR >= 4) {
        // forced order: chain up to e2 | the reads of E (behind the previous MFMA) | h2, rinvn, u, a, d' | MFMA | stores
        const double d = dcur;
        const double r = __builtin_amdgcn_rsq(d);
        const double dm = __builtin_fmin(d, 1.7976931348623157e308);
        double g = dm * r, h = 0.5 * r;
        double e = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, e, g);
        h = __builtin_fma(h, e, h);
        e = __builtin_fma(-h, g, 0.5);
        asm volatile("" : "+v"(e), "+v"(h));
        double Em = keep_lanes(E[R], M);
        double as = 0, xs = 0;
        if constexpr (jj + 1 < 16) {
          constexpr int R1 = (jj + 1) / 4, K1 = (jj + 1) % 4;
          as = rlane(E[R], 16 * K + jj + 1);
          xs = rlane(E[R1], 16 * K1 + jj + 1);
        }
        asm volatile("" : "+v"(e), "+v"(Em), "+s"(as), "+s"(xs));
        h = __builtin_fma(h, e, h);
        double rinvn = -h - h;
        if constexpr (jj + 1 < 16) {
          const double u = as * rinvn;
          dcur = __builtin_fma(-u, u, -xs);
        }
        double a = Em * rinvn;
        asm volatile("" : "+v"(dcur), "+v"(a));
        if constexpr (jj + 1 < 16) E = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, E, 0, 0, 0);
        if constexpr (VAR == 4) {
          lds_store_lanes<jj * 17 * 8>((unsigned)(size_t)((lds_t *)Lc + lo), a, 0xFFFFull << (16 * K));
          Lr[jj] = rinvn;
          asm volatile("" ::: "memory");
        } else if (jj == 15) {
          Lr[lane & 15] = a + rinvn;
        }
        return;
      }
      const double d = dcur;
      const double r = __builtin_amdgcn_rsq(d);
      const double dm = __builtin_fmin(d, 1.7976931348623157e308);
      double g = dm * r, h = 0.5 * r;
      double e = __builtin_fma(-h, g, 0.5);
      g = __builtin_fma(g, e, g);
      h = __builtin_fma(h, e, h);
      e = __builtin_fma(-h, g, 0.5);
      h = __builtin_fma(h, e, h);
      double rinvn = -h - h;
      const double Em = keep_lanes(E[R], M);
      if constexpr (jj + 1 < 16) {
        constexpr int R1 = (jj + 1) / 4, K1 = (jj + 1) % 4;
        const double as = rlane(E[R], 16 * K + jj + 1);
        const double xs = rlane(E[R1], 16 * K1 + jj + 1);
        const double u = as * rinvn;
        dcur = __builtin_fma(-u, u, -xs);
        if constexpr (VAR == 3) asm volatile("" : "+v"(dcur), "+v"(rinvn));
      }
      const double a = Em * rinvn;
      if constexpr (jj + 1 < 16) E = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, E, 0, 0, 0);
      if constexpr (VAR != 1) {
        if constexpr (VAR == 2) Lc[16 * 17 + lane * 16 + jj] = a;
        else lds_store_lanes<jj * 17 * 8>((unsigned)(size_t)((lds_t *)Lc + lo), a, 0xFFFFull << (16 * K));
        Lr[jj] = rinvn;
        asm volatile("" ::: "memory");
        __hip_atomic_store(Lf + jj, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
      } else if (jj == 15) {
        Lr[lane & 15] = a + rinvn;
      }
    });
    asm volatile("" : "+v"(dcur), "+a"(E));
    if (rep == reps - 1) c1 = __builtin_readcyclecounter();
  }
  __syncthreads();
  for (int i = lane; i < 256; i += 64) Lout[i] = Lc[(i & 15) + (i >> 4) * 17];
  if (lane == 0) cyc[0] = c1 - c0;
}
template <int VAR>
static void run(const double *dA, const double *Lref, double *dL, unsigned long long *dc, const char *what) {
  k_piv<VAR><<<1, 64>>>(dA, dL, dc, 50);
  (void)hipDeviceSynchronize();
  double L[256];
  unsigned long long c;
  (void)hipMemcpy(L, dL, sizeof L, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int j = 0; j < 16; ++j)
    for (int i = j + 1; i < 16; ++i) {
      const double e = fabs(L[i + 16 * j] - Lref[i + 16 * j]);
      if (e > err || e != e) err = e;
    }
  printf("%-60s max|L - Lref| (below the diagonal) = %.2e   %.1f cycles per pivot\n", what, err, c / 16.0);
}
int main() {
  double A[256], L[256], G[256];
  srand(7);
  for (int i = 0; i < 256; ++i) G[i] = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = 0; k < 16; ++k) s += G[i + 16 * k] * G[j + 16 * k];
      A[i + 16 * j] = s;
    }
  for (int i = 0; i < 256; ++i) L[i] = A[i];
  for (int j = 0; j < 16; ++j) {
    const double s = sqrt(L[j + 16 * j]);
    L[j + 16 * j] = s;
    for (int i = j + 1; i < 16; ++i) L[i + 16 * j] /= s;
    for (int c = j + 1; c < 16; ++c)
      for (int i = c; i < 16; ++i) L[i + 16 * c] -= L[i + 16 * j] * L[c + 16 * j];
  }
  double *dA, *dL;
  unsigned long long *dc;
  (void)hipMalloc(&dA, sizeof A);
  (void)hipMalloc(&dL, sizeof A);
  (void)hipMalloc(&dc, 8);
  (void)hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(dA, L, dL, dc, "as in the kernel");
    run<1>(dA, L, dL, dc, "no LDS stores (results not comparable)");
    run<2>(dA, L, dL, dc, "column store from all lanes, no exec mask (not comparable)");
    run<3>(dA, L, dL, dc, "next pivot pinned ahead of the vector work");
    run<4>(dA, L, dL, dc, "forced order, reads of E inside the chain, 2 stores");
    run<5>(dA, L, dL, dc, "forced order, no stores (not comparable)");
    run<6>(dA, L, dL, dc, "scalar chain + readlanes of a constant E, no MFMA");
    run<7>(dA, L, dL, dc, "MFMA chain only (row, select, scale, MFMA)");
    run<8>(dA, L, dL, dc, "scalar chain alone");
  }
  return 0;
}
