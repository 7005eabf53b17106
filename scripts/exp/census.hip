// Census of hardware placement ids: which (XCC, SE, SH, CU) does each workgroup land on?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k_census(unsigned *out) {
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID bits 0..3
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay resident a little so that all blocks co-reside
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
  const int nb = 512;
  unsigned *d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
  k_census<<<nb, 256>>>(d);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, int> cnt; std::map<unsigned,int> xccs;
  for (int b = 0; b < nb; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cnt[key]++; xccs[xcc]++;
    if (b < 24) printf("block %3d: xcc=%u se=%u sh=%u cu=%u simd=%u wave=%u raw=%08x\n", b, xcc, se, sh, cu, (hw >> 4) & 3, hw & 15, hw);
  }
  printf("distinct (xcc,se,sh,cu) = %zu\n", cnt.size());
  int mx = 0; for (auto &kv : cnt) mx = kv.second > mx ? kv.second : mx;
  printf("max blocks per cu key = %d\n", mx);
  for (auto &kv : xccs) printf("xcc %u: %d blocks\n", kv.first, kv.second);
  // does blockIdx % 8 predict xcc?
  int agree = 0; for (int b = 0; b < nb; ++b) agree += (h[2*b+1] == h[2*(b%8)+1]);
  printf("blocks whose xcc equals that of block (b %% 8): %d / %d\n", agree, nb);
  return 0;
}
