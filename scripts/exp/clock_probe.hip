// What clock does a latency-bound workgroup see while the matrix cores of the chip are busy?  One wave runs a dependent chain of fp64
// FMAs (the shape of the diagonal-block factorisation's phase A) for a fixed number of iterations and stamps s_memtime (shader clock)
// and s_memrealtime (100 MHz) around it -- alone, and beside a grid of two workgroups per CU that issue 16x16x4 fp64 MFMAs back to back.
// Build: hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void k_chain(unsigned long long *t, double *sink, int iters, int prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  double x = 1.0 + threadIdx.x * 1e-9, y = 0.999999;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) x = __builtin_fma(x, y, 1e-12);  // one dependent fp64 FMA per iteration
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) t[0] = c1 - c0, t[1] = r1 - r0;
  sink[threadIdx.x] = x;
}
__global__ __launch_bounds__(512) void k_mfma(double *sink, int iters) {
  d4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
  hipStream_t A, B;
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&A, hipStreamNonBlocking, lo);
  hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi);
  unsigned long long *t;
  double *sink;
  hipMalloc(&t, 64);
  hipMalloc(&sink, 8 * 512 * 1024);
  const int iters = 20000;
  hipEvent_t e0;
  for (int load = 0; load < 4; ++load)
    for (int rep = 0; rep < 3; ++rep) {
      const int prio = load >> 1;
      (void)hipDeviceSynchronize();
      if (load & 1) {
        k_mfma<<<512, 512, 0, A>>>(sink, 60000);  // ~10+ ms of MFMA on every CU
        for (volatile int w = 0; w < 30000000; ++w) {}
      }
      (void)hipEventCreate(&e0);
      k_chain<<<1, 64, 0, B>>>(t, sink, iters, prio);
      hipStreamSynchronize(B);
      unsigned long long h[2];
      hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
      printf("%s rep %d: %d dependent fp64 FMAs in %.1f us = %.2f ns each; shader clock %.0f MHz (%llu cycles)\n", (load & 1) ? (prio ? "beside MFMA load, s_setprio 3" : "beside MFMA load              ") : (prio ? "alone, s_setprio 3           " : "alone                         "), rep,
             iters, h[1] / 100.0, h[1] * 10.0 / iters, h[0] / (h[1] / 100.0), h[0]);
      hipDeviceSynchronize();
    }
  return 0;
}
