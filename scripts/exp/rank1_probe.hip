// Phase A of the diagonal-block kernel as ONE MFMA per pivot (round 5): the 16x16 block in one fp64 16x16x4 accumulator as
// E = -A (full symmetric), pivot by v_readlane, rank-1 update E += a a^T with a = the scaled row in k-slot jj % 4.
// Question this probe answers: how many wait states does a VALU read of the accumulator need behind the MFMA on gfx950
// (the compiler's hazard table vs the hardware), and how many cycles does a pivot take.
// Build: hipcc --offload-arch=gfx950 -O3 -o rank1_probe rank1_probe.hip
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
template <int NOPS>
__device__ __forceinline__ void wait_mfma(d4_t &E) {
  if constexpr (NOPS == 0) return;
  else if constexpr (NOPS == 1) asm volatile("s_nop 7" : "+v"(E));
  else if constexpr (NOPS == 2) asm volatile("s_nop 15" : "+v"(E));
  else if constexpr (NOPS == 3) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(E));
  else if constexpr (NOPS == 4) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(E));
  else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(E));
}

template <int NOPS>
__global__ __launch_bounds__(64) void k_rank1(const double *A, double *Lout, double *dout, unsigned long long *cyc, int reps) {
  const int lane = threadIdx.x, lo = lane & 15, hi = lane >> 4;
  __shared__ double Lc[256];
  unsigned long long c0 = 0, c1 = 0;
  for (int rep = 0; rep < reps; ++rep) {
    d4_t E;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = hi + 4 * reg;
      E[reg] = -A[(i >= lo) ? i + lo * 16 : lo + i * 16];
    }
    if (rep == reps - 1) c0 = __builtin_readcyclecounter();
    static_for<0, 16>([&](auto JJ) {
      constexpr int jj = decltype(JJ)::value;
      constexpr int R = jj / 4, K = jj % 4;
      constexpr unsigned long long M = (0xFFFFull & ~((2ull << jj) - 1ull)) << (16 * K);
      if constexpr (jj > 0) wait_mfma<NOPS>(E);
      const double d = -rlane(E[R], 16 * K + jj);
      const double Em = keep_lanes(-E[R], M);  // (the select ahead of the chain: an inline-asm result must not feed the MFMA directly)
      const double r = __builtin_amdgcn_rsq(d);
      double g = d * r, h = 0.5 * r;
      double e = __builtin_fma(-h, g, 0.5);
      g = __builtin_fma(g, e, g);
      h = __builtin_fma(h, e, h);
      e = __builtin_fma(-h, g, 0.5);
      h = __builtin_fma(h, e, h);
      const double rinv = h + h;
      const double a = Em * rinv;
      if constexpr (jj + 1 < 16) E = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, E, 0, 0, 0);
      g = __builtin_fma(g, e, g);
      const double cc = __builtin_fma(-g, g, d);
      const double sq = __builtin_fma(cc, h, g);
      if (hi == K) Lc[jj * 16 + lo] = (lo == jj) ? sq : a;
      if (lane == 0 && rep == reps - 1) dout[jj] = d;
    });
    if (rep == reps - 1) c1 = __builtin_readcyclecounter();
  }
  __syncthreads();
  for (int i = lane; i < 256; i += 64) Lout[i] = Lc[i];
  if (lane == 0) cyc[0] = c1 - c0;
}

template <int NOPS>
static void run(const double *dA, const double *Lref, double *dL, double *dd, unsigned long long *dc) {
  k_rank1<NOPS><<<1, 64>>>(dA, dL, dd, dc, 50);
  (void)hipDeviceSynchronize();
  double L[256], d[16];
  unsigned long long c;
  (void)hipMemcpy(L, dL, sizeof L, hipMemcpyDeviceToHost);
  (void)hipMemcpy(d, dd, sizeof d, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double err = 0;
  int firstbad = -1;
  for (int j = 0; j < 16; ++j)
    for (int i = j; i < 16; ++i) {
      const double e = fabs(L[i + 16 * j] - Lref[i + 16 * j]);
      if (!(e <= 1e-12) && firstbad < 0) firstbad = j;
      if (e > err || e != e) err = e;
    }
  if (NOPS == 4) {
    for (int j = 0; j < 3; ++j) {
      printf("col %d: d=%.6f (ref pivot^2 %.6f) diffs:", j, d[j], Lref[j + 16 * j] * Lref[j + 16 * j]);
      for (int i = j; i < 16; ++i) printf(" %.1e", L[i + 16 * j] - Lref[i + 16 * j]);
      printf("\n");
    }
  }
  printf("nops=%d  max|L - Lref| = %.3e  first bad column %d   %llu cycles for 16 pivots = %.1f per pivot\n", NOPS, err, firstbad, c, c / 16.0);
}

int main() {
  double A[256], L[256];
  srand(7);
  double G[256];
  for (int i = 0; i < 256; ++i) G[i] = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = 0; k < 16; ++k) s += G[i + 16 * k] * G[j + 16 * k];
      A[i + 16 * j] = s;
    }
  for (int i = 0; i < 256; ++i) L[i] = A[i];
  for (int j = 0; j < 16; ++j) {  // right-looking, as the kernel
    const double s = sqrt(L[j + 16 * j]);
    L[j + 16 * j] = s;
    for (int i = j + 1; i < 16; ++i) L[i + 16 * j] /= s;
    for (int c = j + 1; c < 16; ++c)
      for (int i = c; i < 16; ++i) L[i + 16 * c] -= L[i + 16 * j] * L[c + 16 * j];
  }
  double *dA, *dL, *dd;
  unsigned long long *dc;
  (void)hipMalloc(&dA, sizeof A);
  (void)hipMalloc(&dL, sizeof A);
  (void)hipMalloc(&dd, 128);
  (void)hipMalloc(&dc, 8);
  (void)hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(dA, L, dL, dd, dc);
    run<1>(dA, L, dL, dd, dc);
    run<2>(dA, L, dL, dd, dc);
    run<3>(dA, L, dL, dd, dc);
    run<4>(dA, L, dL, dd, dc);
    run<5>(dA, L, dL, dd, dc);
  }
  return 0;
}
