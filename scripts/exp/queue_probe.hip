// Do five streams of the library's priorities get five hardware queues?  A consumer kernel launched first
// on stream X polls (<= 50 ms) a word that a producer launched afterwards on stream Y sets: if X and Y share a
// hardware queue the consumer gives up.  Build: hipcc --offload-arch=gfx950 -O2 -o queue_probe queue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_wait(const int *f, int *res) {
  int ok = 2;
  for (int i = 0; i < 50000; ++i) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 1) { ok = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  *res = ok;
}
__global__ void k_set(int *f) { __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main() {
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("priority range: least %d greatest %d\n", lo, hi);
  const int prio[6] = {lo, hi, hi, lo - 1 > hi ? lo - 1 : hi, lo - 1 > hi ? lo - 1 : hi, hi};
  const char *name[6] = {"main(low)", "panel(high)", "trsm(high)", "u1(mid)", "fifth(mid)", "sixth(high)"};
  hipStream_t s[6];
  for (int i = 0; i < 6; ++i) hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, prio[i]);
  int *d;
  hipMalloc(&d, 64 * sizeof(int));
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) {
      if (a == b) continue;
      hipMemset(d, 0, 64 * sizeof(int));
      hipDeviceSynchronize();
      k_wait<<<1, 1, 0, s[a]>>>(d, d + 32);
      k_set<<<1, 1, 0, s[b]>>>(d);
      hipDeviceSynchronize();
      int r = 0;
      hipMemcpy(&r, d + 32, sizeof(int), hipMemcpyDeviceToHost);
      if (r != 1) printf("consumer on %-12s blocks producer on %-12s\n", name[a], name[b]);
    }
  printf("done\n");
  return 0;
}
