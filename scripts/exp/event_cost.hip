// What does a dependency cost on this runtime?  Tiny kernels stamp s_memrealtime (100 MHz) at their start
// and end; the gap between the end stamp of one and the start stamp of the next is the price of whatever
// sits between them.  Build: hipcc --offload-arch=gfx950 -O2 -o event_cost event_cost.hip
//   a  kernel -> kernel, one stream
//   b  kernel -> hipEventRecord -> kernel, one stream
//   c  kernel -> hipStreamWaitEvent(an event long complete) -> kernel, one stream
//   d  kernel (A) -> record -> wait (B) -> kernel (B): the cross-stream hop
//   e  as d, while two other streams exchange events in a loop of their own
//   f  kernel -> kernel on one stream with a device-side flag instead: the second kernel is launched on
//      ANOTHER stream at once and polls a word the first one sets at its end (release / acquire)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ void k_stamp(unsigned long long *out, int slot, int spin_us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[2 * slot] = t0;
    out[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
// producer sets flag at its end; consumer polls it at its start (bounded)
__global__ void k_stamp_set(unsigned long long *out, int slot, int spin_us, int *flag, int val) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[2 * slot] = t0;
    out[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void k_stamp_pollset(unsigned long long *out, int slot, int spin_us, int *flag) {
  if (threadIdx.x == 0) {
    for (int i = 0; i < 2000000; ++i) {
      if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slot) break;
      __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[2 * slot] = t0;
    out[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flag, slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int R = 200;
  unsigned long long *d;
  int *flag;
  CK(hipMalloc(&d, 4 * R * sizeof(unsigned long long)));
  CK(hipMalloc(&flag, sizeof(int)));
  hipStream_t A, B, X, Y;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&X, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&Y, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(4 * R);
  for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t old;
  CK(hipEventCreateWithFlags(&old, hipEventDisableTiming));
  CK(hipEventRecord(old, A));
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(4 * R);
  auto report = [&](const char *name, int pairs) {
    std::vector<double> g;
    for (int i = 0; i + 1 < pairs; ++i) g.push_back(((double)h[2 * (i + 1)] - (double)h[2 * i + 1]) / 100.0);
    std::sort(g.begin(), g.end());
    printf("%-58s median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, g[g.size() / 2], g[g.size() / 10], g[g.size() * 9 / 10]);
  };
  const int spin = 20;  // each kernel runs ~20 us, like the chain's small kernels
  for (int variant = 0; variant < 8; ++variant) {
    CK(hipMemset(d, 0, 4 * R * sizeof(unsigned long long)));
    CK(hipMemset(flag, 0, sizeof(int)));
    CK(hipDeviceSynchronize());
    const char *name = "";
    switch (variant) {
      case 0:
        name = "a  kernel -> kernel, one stream";
        for (int i = 0; i < R; ++i) k_stamp<<<1, 64, 0, A>>>(d, i, spin);
        break;
      case 1:
        name = "b  kernel -> record -> kernel, one stream";
        for (int i = 0; i < R; ++i) {
          k_stamp<<<1, 64, 0, A>>>(d, i, spin);
          CK(hipEventRecord(ev[i], A));
        }
        break;
      case 2:
        name = "c  kernel -> wait(complete event) -> kernel, one stream";
        for (int i = 0; i < R; ++i) {
          k_stamp<<<1, 64, 0, A>>>(d, i, spin);
          CK(hipStreamWaitEvent(A, old, 0));
        }
        break;
      case 3:
        name = "d  cross-stream hop: kernel(A) -> record -> wait(B) -> kernel(B)";
        for (int i = 0; i < R; ++i) {
          hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
          k_stamp<<<1, 64, 0, s>>>(d, i, spin);
          CK(hipEventRecord(ev[i], s));
          CK(hipStreamWaitEvent(o, ev[i], 0));
        }
        break;
      case 4:
        name = "e  hop d beside two other streams ping-ponging events";
        for (int i = 0; i < R; ++i) {
          hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
          k_stamp<<<1, 64, 0, s>>>(d, i, spin);
          CK(hipEventRecord(ev[i], s));
          CK(hipStreamWaitEvent(o, ev[i], 0));
          for (int r = 0; r < 3; ++r) {
            hipStream_t s2 = ((i + r) & 1) ? X : Y, o2 = ((i + r) & 1) ? Y : X;
            k_stamp<<<1, 64, 0, s2>>>(d, R + (i % R), 3);
            CK(hipEventRecord(ev[R + 3 * (i % R) + r], s2));
            CK(hipStreamWaitEvent(o2, ev[R + 3 * (i % R) + r], 0));
          }
        }
        break;
      case 5:
        name = "f  device flag: consumer launched at once on B, polls word set by A's kernel";
        for (int i = 0; i < R; ++i) k_stamp_pollset<<<1, 64, 0, (i & 1) ? B : A>>>(d, i, spin, flag);
        break;
      case 6:
        name = "g  kernel -> record, record, record -> kernel, one stream";
        for (int i = 0; i < R; ++i) {
          k_stamp<<<1, 64, 0, A>>>(d, i, spin);
          CK(hipEventRecord(ev[i], A));
          CK(hipEventRecord(ev[R + i], A));
          CK(hipEventRecord(ev[2 * R + i], A));
        }
        break;
      case 7:
        name = "h  hop d, the waiting stream also holds a wait for a complete event";
        for (int i = 0; i < R; ++i) {
          hipStream_t s = (i & 1) ? B : A, o = (i & 1) ? A : B;
          k_stamp<<<1, 64, 0, s>>>(d, i, spin);
          CK(hipEventRecord(ev[i], s));
          CK(hipStreamWaitEvent(o, old, 0));
          CK(hipStreamWaitEvent(o, ev[i], 0));
        }
        break;
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, 4 * R * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    report(name, R);
  }
  return 0;
}
