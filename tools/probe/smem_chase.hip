// smem_chase.hip -- what a dependent chain of node fetches costs on gfx950, by fetch path.
// One wave per block; every wave chases its own random cycle through a table of 16-byte nodes (node.x = next index).
//   path 0: s_load_dwordx4 (scalar cache)           path 1: s_load_dwordx16 of the node's 64-byte line
//   path 2: global_load_dwordx4, all lanes one address, + 4 v_readfirstlane (vector L1)
//   path 3: two independent s_load_dwordx4 chains per wave (memory-level parallelism 2)
//   path 4: two independent global_load chains per wave
// Prints ns per step per wave and steps per microsecond per CU for footprints from scalar-cache-resident to L2-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

template <int PATH>
__global__ void __launch_bounds__(64) chase(const u32x4* tab, uint32_t n, int steps, uint32_t* out) {
  uint32_t i = uni((blockIdx.x * 2654435761u) % n), j = uni((blockIdx.x * 40503u + 12345u) % n);
  const char __attribute__((address_space(4)))* b = (const char __attribute__((address_space(4)))*)(uintptr_t)tab;
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    if (PATH == 0) {
      u32x4 v = *(const u32x4 __attribute__((address_space(4)))*)(b + (i << 4));
      i = uni(v.x); acc += v.y;
    } else if (PATH == 1) {
      u32x16 v = *(const u32x16 __attribute__((address_space(4)))*)(b + ((i & ~3u) << 4));
      const uint32_t k = i & 3u;
      i = uni(k == 0 ? v[0] : (k == 1 ? v[4] : (k == 2 ? v[8] : v[12]))); acc += v[1];
    } else if (PATH == 2) {
      u32x4 v;
      const u32x4* p = tab + i;
      asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
      i = uni(v.x); acc += v.y;
    } else if (PATH == 3) {
      u32x4 v = *(const u32x4 __attribute__((address_space(4)))*)(b + (i << 4));
      u32x4 w = *(const u32x4 __attribute__((address_space(4)))*)(b + (j << 4));
      i = uni(v.x); j = uni(w.x); acc += v.y + w.y;
    } else {
      u32x4 v, w;
      const u32x4 *p = tab + i, *q = tab + j;
      asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %3, off\n s_waitcnt vmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(p), "v"(q) : "memory");
      i = uni(v.x); j = uni(w.x); acc += v.y + w.y;
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = i + j + acc;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const int steps = 4000;
  uint32_t* out; CK(hipMalloc(&out, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# cus %d; ns = per step per wave; rate = steps (node fetches) per microsecond per CU\n", cus);
  printf("%-6s %-10s %-6s %10s %12s\n", "path", "footprint", "w/CU", "ns", "rate/us/CU");
  const size_t foots[] = {2u << 10, 8u << 10, 32u << 10, 256u << 10, 2u << 20, 16u << 20};
  for (size_t fb : foots) {
    const uint32_t n = (uint32_t)(fb / 16);
    std::vector<uint32_t> perm(n), host(4 * (size_t)n);
    for (uint32_t k = 0; k < n; k++) perm[k] = k;
    std::mt19937 g(7); std::shuffle(perm.begin(), perm.end(), g);
    for (uint32_t k = 0; k < n; k++) { host[4 * (size_t)perm[k]] = perm[(k + 1) % n]; host[4 * (size_t)perm[k] + 1] = 1; }  // one cycle through all nodes
    u32x4* tab; CK(hipMalloc(&tab, fb)); CK(hipMemcpy(tab, host.data(), fb, hipMemcpyHostToDevice));
    for (int path = 0; path < 5; path++)
      for (int wpc : {1, 4, 8, 16, 24, 32}) {
        const int grid = cus * wpc;
        for (int rep = 0; rep < 2; rep++) {
          CK(hipEventRecord(e0));
          switch (path) {
            case 0: hipLaunchKernelGGL(chase<0>, dim3(grid), dim3(64), 0, 0, tab, n, steps, out); break;
            case 1: hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(64), 0, 0, tab, n, steps, out); break;
            case 2: hipLaunchKernelGGL(chase<2>, dim3(grid), dim3(64), 0, 0, tab, n, steps, out); break;
            case 3: hipLaunchKernelGGL(chase<3>, dim3(grid), dim3(64), 0, 0, tab, n, steps, out); break;
            default: hipLaunchKernelGGL(chase<4>, dim3(grid), dim3(64), 0, 0, tab, n, steps, out); break;
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep == 1) {
            const double ns = ms * 1e6 / steps;
            const double fetches = (path >= 3 ? 2.0 : 1.0) * wpc;
            printf("%-6d %-10zu %-6d %10.1f %12.1f\n", path, fb, wpc, ns, fetches / (ns * 1e-3));
            fflush(stdout);
          }
        }
      }
    CK(hipFree(tab));
  }
  return 0;
}
