// prints the distinct (XCC_ID, HW_ID se/sh/cu) of the waves of a big grid: the CU census behind the per-CU work-queue heads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void k(unsigned* out) {
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  for (volatile int i = 0; i < 20000; i++) {}
}
int main() {
  const int n = 8192;
  unsigned* d; hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> cus; std::set<unsigned> xs;
  unsigned orall = 0, orx = 0;
  for (int i = 0; i < n; i++) { orall |= h[2 * i]; orx |= h[2 * i + 1]; cus[(h[2 * i + 1] & 0xf) << 16 | ((h[2 * i] >> 8) & 0xffff)]++; xs.insert(h[2 * i + 1] & 0xf); }
  printf("OR of HW_ID %08x, OR of XCC_ID %08x, distinct (xcc, hwid>>8): %zu, xccs %zu\n", orall, orx, cus.size(), xs.size());
  int c = 0; for (auto& kv : cus) { if (c++ < 40) printf("%05x:%d ", kv.first, kv.second); } printf("\n");
  for (int i = 0; i < 24; i++) printf("blk %d: hw %08x xcc %x\n", i, h[2 * i], h[2 * i + 1]);
  return 0;
}
