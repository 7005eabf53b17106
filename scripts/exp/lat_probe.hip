// Dependent-issue latencies on gfx950 that the pivot chain of the diagonal-block kernel is made of: one wave, s_memtime around
// an unrolled chain of N dependent instructions of one kind.  Build: hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ __launch_bounds__(64) void k_lat(unsigned long long *out, double *sink, double seed) {
  double x = seed + threadIdx.x * 1e-9, y = 0.999999, z = 1e-12;
  unsigned long long c0, c1;
  int n = 0;
  // (0) fp64 FMA chain
  c0 = __builtin_readcyclecounter();
  REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (1) fp64 MUL chain
  c0 = __builtin_readcyclecounter();
  REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (2) fp64 FMA, independent (issue rate)
  {
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    c0 = __builtin_readcyclecounter();
    REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z));)
    c1 = __builtin_readcyclecounter();
    x += a0 + a1 + a2 + a3;
  }
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (3) v_rsq_f64 chain (rsq of rsq ...)
  x = 1.5;
  c0 = __builtin_readcyclecounter();
  REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(x));)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (4) readlane -> SGPR -> VALU use -> readlane ... (v_readlane x2 + v_mul with the SGPR pair)
  x = 1.0 + threadIdx.x * 1e-9;
  c0 = __builtin_readcyclecounter();
  REP64(asm volatile("v_readlane_b32 s20, %0, 17\n\tv_readlane_b32 s21, %1, 17\n\tv_mul_f64 %2, s[20:21], %3" : : "v"(__double2loint(x)), "v"(__double2hiint(x)), "v"(x), "v"(y) : "s20", "s21");)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (5) fp64 MFMA 16x16x4 dependent chain (srcC = previous result)
  d4_t E = {x, x, x, x};
  c0 = __builtin_readcyclecounter();
  REP16(E = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, E, 0, 0, 0);)
  asm volatile("s_nop 0" : "+v"(E));
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (6) MFMA -> VALU read of the result -> VALU -> MFMA operand (mul, then MFMA again): the pivot's round trip
  double t = y;
  c0 = __builtin_readcyclecounter();
  REP16(E = __builtin_amdgcn_mfma_f64_16x16x4f64(t, t, E, 0, 0, 0); t = E[0] * y;)
  asm volatile("s_nop 0" : "+v"(E));
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (7) LDS write + read-back round trip (ds_write_b64, ds_read_b64 of the same address, dependent)
  __shared__ double buf[64];
  c0 = __builtin_readcyclecounter();
  REP16(buf[threadIdx.x] = x; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x = buf[threadIdx.x ^ 1] * y; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (8) empty timer
  c0 = __builtin_readcyclecounter();
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (9) fp32 FMA chain
  float f = (float)x, fy = 0.99f, fz = 1e-6f;
  c0 = __builtin_readcyclecounter();
  REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(fy), "v"(fz));)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  // (10) v_accvgpr_read after MFMA (one MFMA, then read): MFMA + read latency
  c0 = __builtin_readcyclecounter();
  REP16(E = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, E, 0, 0, 0); asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(E[0]));)
  c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[n] = c1 - c0; ++n;
  sink[threadIdx.x] = x + E[0] + E[1] + E[2] + E[3] + t + f;
}
int main() {
  unsigned long long *o, h[16];
  double *sink;
  (void)hipMalloc(&o, 128);
  (void)hipMalloc(&sink, 512);
  const char *name[] = {"v_fma_f64 dependent (64)", "v_mul_f64 dependent (64)", "v_fma_f64 independent (64)", "v_rsq_f64 dependent (64)",
                        "readlane x2 + v_mul (64 round trips)", "mfma f64 16x16x4 dependent (16)", "mfma -> v_mul -> mfma (16)",
                        "LDS write + read round trip (16)", "empty", "v_fma_f32 dependent (64)", "mfma + VALU read of result (16)"};
  const int cnt[] = {64, 64, 64, 64, 64, 16, 16, 16, 1, 64, 16};
  for (int rep = 0; rep < 2; ++rep) {
    k_lat<<<1, 64>>>(o, sink, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < 11; ++i) printf("%-42s %6llu cycles = %6.1f each\n", name[i], h[i], (double)(h[i] - h[8]) / cnt[i]);
  return 0;
}
