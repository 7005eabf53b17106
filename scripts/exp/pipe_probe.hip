// Does a dispatch that cannot launch all its workgroups at once (more workgroups than the chip holds) delay the start of
// small kernels on OTHER streams?  Stream A: a grid of 4 rounds of LDS-heavy workgroups that each spin ~30 us.  20 us
// later, streams B0..B7 each get a one-wave kernel that stamps its start.  A stream whose kernel starts only when A's
// launch has drained shares A's dispatch pipe.  Build: hipcc --offload-arch=gfx950 -O2 -o pipe_probe pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <thread>
__global__ void k_big(unsigned long long *t, int spin_us) {
  __shared__ char lds[65536];
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(16);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
  if (lds[(threadIdx.x + 1) & 255] == 7) t[2] = 0;
}
__global__ void k_small(unsigned long long *t) {
  if (threadIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime();
}
int main(int argc, char **argv) {
  const int NS = 10;
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t A, B[NS];
  hipStreamCreateWithPriority(&A, hipStreamNonBlocking, lo);
  for (int i = 0; i < NS; ++i) hipStreamCreateWithPriority(&B[i], hipStreamNonBlocking, i % 2 ? hi : (lo - 1 > hi ? lo - 1 : hi));
  unsigned long long *d;
  hipMalloc(&d, 64 * 8 * (NS + 1));
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d, 0, 64 * 8 * (NS + 1));
    hipDeviceSynchronize();
    k_big<<<2048, 256, 0, A>>>(d, 30);  // 2 per CU by LDS -> 512 resident, 4 rounds of ~30 us
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    for (int i = 0; i < NS; ++i) k_small<<<1, 64, 0, B[i]>>>(d + 8 * (i + 1));
    hipDeviceSynchronize();
    unsigned long long h[8 * (NS + 1)];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("rep %d: big kernel first workgroup at 0, last workgroup ended at %.1f us\n", rep, (h[1] - h[0]) / 100.0);
    for (int i = 0; i < NS; ++i) printf("   stream B%d (%s): small kernel started at %8.1f us\n", i, i % 2 ? "high" : "mid ", ((long long)h[8 * (i + 1)] - (long long)h[0]) / 100.0);
  }
  return 0;
}
