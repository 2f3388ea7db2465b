// bih_build_device.hpp -- `bih` (Bih.hs:211-324) built on the GPU: the same tree as Graph::bih / the reference's
// build_rec, node for node and bit for bit, level by level instead of by recursion.  Included by glome_device.hip.
//
// build_rec looks at a node's objects three times: to sort them into four candidate partitions (bbox centre below the
// node's midpoint on x / y / z, and big-vs-small by surface area), to take each candidate's split planes (max of the
// left boxes' upper bounds, min of the right boxes' lower bounds) and cost, and to hand the chosen halves down.  Here a
// level of the tree is one pass of four kernels over the object array, which stays one contiguous segment per node:
//   k_bb_accumulate  every object adds itself to its node's four candidates (counts, split planes) -- a wave whose lanes
//                    share the node reduces in registers and issues one set of atomics
//   k_bb_decide      every node of the level: leaf (<= 3 objects, or no candidate cheaper than the node itself) or the
//                    cheapest candidate, with build_rec's own comparison chain (Bih.hs:278-285); children are allocated
//   k_bb_scan_*      prefix sums of "goes left" over the array
//   k_bb_scatter     stable partition of every split segment (a leaf's order is the order the reference's list
//                    partition leaves: `nearest` prefers the later item on ties, so the order is part of the result)
// All arithmetic that decides something is fp64 with explicit round-to-nearest multiplies and adds (no contraction), in
// the host builder's operation order.  Max / min of split planes go through atomics on order-preserving integer keys;
// the compare-select folds of the host builder give the same value except for the sign of a zero, which nothing
// downstream can see (the planes are stored with +-delta added, and boxes only feed midpoints and areas).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <vector>

#include "host_graph.hpp"

namespace glome {
namespace bihdev {

struct DevNode {
  int start, count;       // segment of the object array
  double lo[3], hi[3];    // the node's box (the parent's, shrunk along the split axis only)
  double mid[3];
  int leaf, axis, left, right, ksel, nleft;
  double lsplit, rsplit;
};
struct Acc { unsigned int cnt[4]; unsigned long long lmax[4], rmin[4]; };

__device__ __forceinline__ unsigned long long dkey(double x) {  // order-preserving: a < b <=> dkey(a) < dkey(b)
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double d_area(const double* lo, const double* hi) {  // box_area / bbsa, Vec.hs:694-697
  double dx = __dsub_rn(hi[0], lo[0]), dy = __dsub_rn(hi[1], lo[1]), dz = __dsub_rn(hi[2], lo[2]);
  double v = __dmul_rn(2.0, __dadd_rn(__dadd_rn(__dmul_rn(dx, dy), __dmul_rn(dx, dz)), __dmul_rn(dy, dz)));
  return (0 <= v) ? v : 0;
}

struct Objs { const double* lo[3]; const double* hi[3]; const double* mid[3]; const double* area; };

// which side of candidate k object o falls on in a node with midpoint `mid` and area `sa` (true = left)
__device__ __forceinline__ bool side(const Objs& O, int o, int k, const double* mid, double sa) {
  if (k < 3) return O.mid[k][o] < mid[k];
  return O.area[o] > __dmul_rn(sa, 0.4);
}

__global__ void k_bb_prepare(int n, const double* lo0, const double* lo1, const double* lo2, const double* hi0, const double* hi1, const double* hi2,
                             double* m0, double* m1, double* m2, double* area, int* idx, int* nodeof) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  m0[i] = __dmul_rn(__dadd_rn(lo0[i], hi0[i]), 0.5); m1[i] = __dmul_rn(__dadd_rn(lo1[i], hi1[i]), 0.5); m2[i] = __dmul_rn(__dadd_rn(lo2[i], hi2[i]), 0.5);
  double l[3] = {lo0[i], lo1[i], lo2[i]}, h[3] = {hi0[i], hi1[i], hi2[i]};
  area[i] = d_area(l, h);
  idx[i] = i; nodeof[i] = 0;
}

__global__ void k_bb_clear(Acc* acc, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 4; k++) { acc[i].cnt[k] = 0; acc[i].lmax[k] = dkey(-kInfinity); acc[i].rmin[k] = dkey(kInfinity); }
}

__global__ void k_bb_accumulate(int n, Objs O, const int* idx, const int* nodeof, const DevNode* nodes, Acc* acc, int lvl_begin) {  // Bih
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int nd = i < n ? nodeof[i] : -1;
  const bool act = nd >= 0 && nodes[nd].count > 3;  // leaves by size decide nothing
  unsigned int c[4] = {0, 0, 0, 0};
  unsigned long long lm[4], rm[4];
  for (int k = 0; k < 4; k++) { lm[k] = dkey(-kInfinity); rm[k] = dkey(kInfinity); }
  if (act) {
    const DevNode& N = nodes[nd];
    const double sa = d_area(N.lo, N.hi);
    const int o = idx[i];
    for (int k = 0; k < 4; k++) {
      const int ax = k == 3 ? 0 : k;
      if (side(O, o, k, N.mid, sa)) { c[k] = 1; lm[k] = dkey(O.hi[ax][o]); } else rm[k] = dkey(O.lo[ax][o]);
    }
  }
  // wave-uniform node: reduce across the wave, one lane issues the atomics
  const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
  if (am == 0) return;
  const int first = __ffsll((long long)am) - 1;
  const int nd0 = __shfl(nd, first, 64);
  const bool uniform = __builtin_amdgcn_ballot_w64(act && nd != nd0) == 0;
  if (uniform) {
    for (int k = 0; k < 4; k++) {
      for (int d = 32; d >= 1; d >>= 1) {
        c[k] += (unsigned int)__shfl_xor((int)c[k], d, 64);
        unsigned long long a = (unsigned long long)__shfl_xor((long long)lm[k], d, 64), b = (unsigned long long)__shfl_xor((long long)rm[k], d, 64);
        lm[k] = a > lm[k] ? a : lm[k];
        rm[k] = b < rm[k] ? b : rm[k];
      }
    }
    if ((threadIdx.x & 63) == first) {
      Acc& A = acc[nd0 - lvl_begin];
      for (int k = 0; k < 4; k++) { atomicAdd(&A.cnt[k], c[k]); atomicMax(&A.lmax[k], lm[k]); atomicMin(&A.rmin[k], rm[k]); }
    }
  } else if (act) {
    Acc& A = acc[nd - lvl_begin];
    for (int k = 0; k < 4; k++) {
      if (c[k]) { atomicAdd(&A.cnt[k], 1u); atomicMax(&A.lmax[k], lm[k]); } else atomicMin(&A.rmin[k], rm[k]);
    }
  }
}

__global__ void k_bb_decide(DevNode* nodes, const Acc* acc, int lvl_begin, int lvl_end, int* n_nodes, int node_cap, int* error) {
  int nd = lvl_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (nd >= lvl_end) return;
  DevNode& N = nodes[nd];
  N.leaf = 1; N.axis = -1; N.left = N.right = -1; N.lsplit = N.rsplit = 0; N.ksel = -1; N.nleft = 0;
  if (N.count <= 3) return;
  const Acc& A = acc[nd - lvl_begin];
  const double sa = d_area(N.lo, N.hi);
  double lmax[4], rmin[4], cost[4];
  for (int k = 0; k < 4; k++) {
    const int ax = k == 3 ? 0 : k;
    lmax[k] = dunkey(A.lmax[k]); rmin[k] = dunkey(A.rmin[k]);
    double llo[3] = {N.lo[0], N.lo[1], N.lo[2]}, lhi[3] = {N.hi[0], N.hi[1], N.hi[2]}, rlo[3] = {N.lo[0], N.lo[1], N.lo[2]}, rhi[3] = {N.hi[0], N.hi[1], N.hi[2]};
    lhi[ax] = lmax[k]; rlo[ax] = rmin[k];
    const double nl = (double)A.cnt[k], nr = (double)((unsigned int)N.count - A.cnt[k]);
    cost[k] = __dmul_rn(__dadd_rn(__dmul_rn(d_area(llo, lhi), nl), __dmul_rn(d_area(rlo, rhi), nr)), k < 3 ? 1.1 : 1.2);
  }
  const double costorig = __dmul_rn(sa, (double)N.count);
  if (costorig < cost[0] && costorig < cost[1] && costorig < cost[2] && costorig < cost[3]) return;
  int k;
  if (cost[0] < cost[1] && cost[0] < cost[2] && cost[0] < cost[3]) k = 0;
  else if (cost[1] < cost[2] && cost[1] < cost[3]) k = 1;
  else if (cost[1] < cost[3]) k = 2;  // as written in the reference (`costy < costb`, Bih.hs:283)
  else k = 3;
  const int ax = k == 3 ? 0 : k;
  const int l = atomicAdd(n_nodes, 2);
  if (l + 2 > node_cap) { atomicExch(error, 1); return; }
  N.leaf = 0; N.axis = ax; N.left = l; N.right = l + 1; N.ksel = k; N.nleft = (int)A.cnt[k];
  N.lsplit = __dadd_rn(lmax[k], kDelta); N.rsplit = __dsub_rn(rmin[k], kDelta);
  DevNode& L = nodes[l];
  DevNode& R = nodes[l + 1];
  L.start = N.start; L.count = N.nleft; R.start = N.start + N.nleft; R.count = N.count - N.nleft;
  for (int a = 0; a < 3; a++) { L.lo[a] = N.lo[a]; L.hi[a] = N.hi[a]; R.lo[a] = N.lo[a]; R.hi[a] = N.hi[a]; }
  L.hi[ax] = lmax[k]; R.lo[ax] = rmin[k];
  for (int a = 0; a < 3; a++) { L.mid[a] = __dmul_rn(__dadd_rn(L.lo[a], L.hi[a]), 0.5); R.mid[a] = __dmul_rn(__dadd_rn(R.lo[a], R.hi[a]), 0.5); }
}

// ---- Mesh (Mesh.hs:69-113): the same four candidates over the triangles' boxes, but a child's box is the true union of
// its triangles' boxes (all six bounds), every candidate costs x 1.1, a node of fewer than three triangles is a leaf and
// the comparison chain is the straight one.
struct AccM { unsigned int cnt[4]; unsigned long long lo[4][2][3], hi[4][2][3]; };  // [candidate][left, right][axis]
__global__ void k_mb_clear(AccM* acc, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 4; k++) {
    acc[i].cnt[k] = 0;
    for (int sd = 0; sd < 2; sd++) for (int a = 0; a < 3; a++) { acc[i].lo[k][sd][a] = dkey(kInfinity); acc[i].hi[k][sd][a] = dkey(-kInfinity); }  // box_empty
  }
}
__global__ void k_mb_accumulate(int n, Objs O, const int* idx, const int* nodeof, const DevNode* nodes, AccM* acc, int lvl_begin) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int nd = i < n ? nodeof[i] : -1;
  const bool act = nd >= 0 && nodes[nd].count >= 3;
  const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
  if (am == 0) return;
  const int first = __ffsll((long long)am) - 1;
  const int nd0 = __shfl(nd, first, 64);
  const bool uniform = __builtin_amdgcn_ballot_w64(act && nd != nd0) == 0;
  bool sd[4] = {false, false, false, false};
  unsigned long long blo[3], bhi[3];
  for (int a = 0; a < 3; a++) { blo[a] = dkey(kInfinity); bhi[a] = dkey(-kInfinity); }
  if (act) {
    const DevNode& N = nodes[nd];
    const double sa = d_area(N.lo, N.hi);
    const int o = idx[i];
    for (int k = 0; k < 4; k++) sd[k] = side(O, o, k, N.mid, sa);
    for (int a = 0; a < 3; a++) { blo[a] = dkey(O.lo[a][o]); bhi[a] = dkey(O.hi[a][o]); }
  }
  for (int k = 0; k < 4; k++) {
    if (uniform) {
      const unsigned long long lm = __builtin_amdgcn_ballot_w64(act && sd[k]);
      unsigned long long v[2][2][3];  // [left, right][lo, hi][axis]: this lane's box on its side, identities on the other
      for (int a = 0; a < 3; a++) {
        const bool l = act && sd[k], r = act && !sd[k];
        v[0][0][a] = l ? blo[a] : dkey(kInfinity); v[0][1][a] = l ? bhi[a] : dkey(-kInfinity);
        v[1][0][a] = r ? blo[a] : dkey(kInfinity); v[1][1][a] = r ? bhi[a] : dkey(-kInfinity);
      }
      for (int d = 32; d >= 1; d >>= 1)
        for (int s2 = 0; s2 < 2; s2++)
          for (int a = 0; a < 3; a++) {
            unsigned long long x = (unsigned long long)__shfl_xor((long long)v[s2][0][a], d, 64), y = (unsigned long long)__shfl_xor((long long)v[s2][1][a], d, 64);
            v[s2][0][a] = x < v[s2][0][a] ? x : v[s2][0][a];
            v[s2][1][a] = y > v[s2][1][a] ? y : v[s2][1][a];
          }
      if ((threadIdx.x & 63) == first) {
        AccM& A = acc[nd0 - lvl_begin];
        atomicAdd(&A.cnt[k], (unsigned int)__popcll(lm));
        for (int s2 = 0; s2 < 2; s2++) for (int a = 0; a < 3; a++) { atomicMin(&A.lo[k][s2][a], v[s2][0][a]); atomicMax(&A.hi[k][s2][a], v[s2][1][a]); }
      }
    } else if (act) {
      AccM& A = acc[nd - lvl_begin];
      const int s2 = sd[k] ? 0 : 1;
      if (sd[k]) atomicAdd(&A.cnt[k], 1u);
      for (int a = 0; a < 3; a++) { atomicMin(&A.lo[k][s2][a], blo[a]); atomicMax(&A.hi[k][s2][a], bhi[a]); }
    }
  }
}
__global__ void k_mb_decide(DevNode* nodes, const AccM* acc, int lvl_begin, int lvl_end, int* n_nodes, int node_cap, int* error) {
  int nd = lvl_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (nd >= lvl_end) return;
  DevNode& N = nodes[nd];
  N.leaf = 1; N.axis = -1; N.left = N.right = -1; N.lsplit = N.rsplit = 0; N.ksel = -1; N.nleft = 0;
  if (N.count < 3) return;
  const AccM& A = acc[nd - lvl_begin];
  double cost[4], b[4][2][2][3];  // [candidate][left, right][lo, hi][axis]
  for (int k = 0; k < 4; k++) {
    for (int s2 = 0; s2 < 2; s2++) for (int a = 0; a < 3; a++) { b[k][s2][0][a] = dunkey(A.lo[k][s2][a]); b[k][s2][1][a] = dunkey(A.hi[k][s2][a]); }
    const double nl = (double)A.cnt[k], nr = (double)((unsigned int)N.count - A.cnt[k]);
    cost[k] = __dmul_rn(__dadd_rn(__dmul_rn(d_area(b[k][0][0], b[k][0][1]), nl), __dmul_rn(d_area(b[k][1][0], b[k][1][1]), nr)), 1.1);
  }
  const double lcost = __dmul_rn(d_area(N.lo, N.hi), (double)N.count);
  if (lcost < cost[0] && lcost < cost[1] && lcost < cost[2] && lcost < cost[3]) return;
  int k;
  if (cost[0] < cost[1] && cost[0] < cost[2] && cost[0] < cost[3]) k = 0;
  else if (cost[1] < cost[2] && cost[1] < cost[3]) k = 1;
  else if (cost[2] < cost[3]) k = 2;
  else k = 3;
  const int l = atomicAdd(n_nodes, 2);
  if (l + 2 > node_cap) { atomicExch(error, 1); return; }
  N.leaf = 0; N.axis = 0; N.left = l; N.right = l + 1; N.ksel = k; N.nleft = (int)A.cnt[k];
  DevNode& L = nodes[l];
  DevNode& R = nodes[l + 1];
  L.start = N.start; L.count = N.nleft; R.start = N.start + N.nleft; R.count = N.count - N.nleft;
  for (int a = 0; a < 3; a++) {
    L.lo[a] = b[k][0][0][a]; L.hi[a] = b[k][0][1][a]; R.lo[a] = b[k][1][0][a]; R.hi[a] = b[k][1][1][a];
    L.mid[a] = __dmul_rn(__dadd_rn(L.lo[a], L.hi[a]), 0.5); R.mid[a] = __dmul_rn(__dadd_rn(R.lo[a], R.hi[a]), 0.5);
  }
}

// "goes left" per position (0 for objects of leaves and of finished segments) and its prefix sums, 1024 positions per block
constexpr int kScanBlock = 1024;
__device__ __forceinline__ int goes_left(const Objs& O, const int* idx, const int* nodeof, const DevNode* nodes, int i, int n) {
  if (i >= n) return 0;
  int nd = nodeof[i];
  if (nd < 0 || nodes[nd].leaf) return 0;
  const DevNode& N = nodes[nd];
  return side(O, idx[i], N.ksel, N.mid, d_area(N.lo, N.hi)) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_bb_scan_block(int n, Objs O, const int* idx, const int* nodeof, const DevNode* nodes, int* pre, int* blocksum) {
  __shared__ int sh[256];
  const int base = blockIdx.x * kScanBlock + threadIdx.x * 4;
  int f[4], s = 0;
  for (int q = 0; q < 4; q++) { f[q] = goes_left(O, idx, nodeof, nodes, base + q, n); s += f[q]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    int v = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  int run = sh[threadIdx.x] - s;  // exclusive prefix of this thread's four
  for (int q = 0; q < 4; q++) { if (base + q < n) pre[base + q] = run; run += f[q]; }
  if (threadIdx.x == 255) blocksum[blockIdx.x] = sh[255];
}
__global__ void __launch_bounds__(1024) k_bb_scan_sums(int nblocks, int* blocksum) {  // one block: exclusive prefix in place
  __shared__ int sh[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? blocksum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      int t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) blocksum[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
}
__global__ void k_bb_scatter(int n, Objs O, const int* idx, const int* nodeof, const DevNode* nodes, const int* pre, const int* blockoff, int* idx2, int* nodeof2) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int nd = nodeof[i];
  if (nd < 0 || nodes[nd].leaf) { idx2[i] = idx[i]; nodeof2[i] = -1; return; }
  const DevNode& N = nodes[nd];
  auto S = [&](int p) { return pre[p] + blockoff[p / kScanBlock]; };
  const int lbefore = S(i) - S(N.start);  // left-goers of this segment before position i
  const bool left = side(O, idx[i], N.ksel, N.mid, d_area(N.lo, N.hi));
  const int pos = left ? N.start + lbefore : N.start + N.nleft + ((i - N.start) - lbefore);
  idx2[pos] = idx[i];
  nodeof2[pos] = left ? N.left : N.right;
}

#define BB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); goto done; } } while (0)

// The level loop.  boxes: the objects' bounds in input order; bb: the root's box.  Leaves nodes (breadth-first device
// numbering) and the final object order.
template <bool MESH>
inline bool levels(const std::vector<Box3>& boxes, const Box3& bb, hipStream_t st, std::string& err, float* gpu_ms, std::vector<DevNode>& nodes, std::vector<int>& order) {
  using AccT = typename std::conditional<MESH, AccM, Acc>::type;
  const int n = (int)boxes.size();
  const int node_cap = 4 * n + 16, max_levels = 512, acc_cap = n + 2;
  std::vector<double> h((size_t)6 * n);
  for (int i = 0; i < n; i++) {
    const Box3& b = boxes[(size_t)i];
    const double v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
    for (int a = 0; a < 6; a++) h[(size_t)a * n + i] = v[a];
  }
  double* d_box = nullptr; double* d_mid = nullptr; double* d_area = nullptr;
  int* d_idx[2] = {nullptr, nullptr}; int* d_nodeof[2] = {nullptr, nullptr};
  int* d_pre = nullptr; int* d_bsum = nullptr; int* d_cnt = nullptr;
  DevNode* d_nodes = nullptr; AccT* d_acc = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const int nblk = (n + 255) / 256, nscan = (n + kScanBlock - 1) / kScanBlock;
  bool ok = false;
  int lvl_begin = 0, lvl_end = 1, cur = 0, nlev = 0;
  int hcnt[2] = {1, 0};
  DevNode root{};
  Objs O{};
  BB_HIP(hipMalloc((void**)&d_box, sizeof(double) * 6 * n)); BB_HIP(hipMalloc((void**)&d_mid, sizeof(double) * 3 * n)); BB_HIP(hipMalloc((void**)&d_area, sizeof(double) * n));
  for (int q = 0; q < 2; q++) { BB_HIP(hipMalloc((void**)&d_idx[q], sizeof(int) * n)); BB_HIP(hipMalloc((void**)&d_nodeof[q], sizeof(int) * n)); }
  BB_HIP(hipMalloc((void**)&d_pre, sizeof(int) * n)); BB_HIP(hipMalloc((void**)&d_bsum, sizeof(int) * (nscan + 1))); BB_HIP(hipMalloc((void**)&d_cnt, sizeof(int) * 2));
  BB_HIP(hipMalloc((void**)&d_nodes, sizeof(DevNode) * (size_t)node_cap)); BB_HIP(hipMalloc((void**)&d_acc, sizeof(AccT) * (size_t)acc_cap));
  BB_HIP(hipEventCreate(&e0)); BB_HIP(hipEventCreate(&e1));
  BB_HIP(hipMemcpyAsync(d_box, h.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice, st));
  root.start = 0; root.count = n;
  root.lo[0] = bb.lo.x; root.lo[1] = bb.lo.y; root.lo[2] = bb.lo.z; root.hi[0] = bb.hi.x; root.hi[1] = bb.hi.y; root.hi[2] = bb.hi.z;
  { D3 m = box_mid(bb); root.mid[0] = m.x; root.mid[1] = m.y; root.mid[2] = m.z; }
  BB_HIP(hipMemcpyAsync(d_nodes, &root, sizeof(DevNode), hipMemcpyHostToDevice, st));
  BB_HIP(hipMemcpyAsync(d_cnt, hcnt, sizeof(hcnt), hipMemcpyHostToDevice, st));
  for (int a = 0; a < 3; a++) { O.lo[a] = d_box + (size_t)a * n; O.hi[a] = d_box + (size_t)(3 + a) * n; O.mid[a] = d_mid + (size_t)a * n; }
  O.area = d_area;
  BB_HIP(hipEventRecord(e0, st));
  hipLaunchKernelGGL(k_bb_prepare, dim3(nblk), dim3(256), 0, st, n, O.lo[0], O.lo[1], O.lo[2], O.hi[0], O.hi[1], O.hi[2], d_mid, d_mid + n, d_mid + 2 * (size_t)n, d_area, d_idx[0], d_nodeof[0]);
  while (lvl_begin < lvl_end) {
    if (++nlev > max_levels) { err = "device tree build: deeper than 512 levels"; goto done; }
    const int ln = lvl_end - lvl_begin;
    if (ln > acc_cap) { err = "device tree build: level wider than the accumulator pool"; goto done; }
    if constexpr (MESH) {
      hipLaunchKernelGGL(k_mb_clear, dim3((ln + 255) / 256), dim3(256), 0, st, d_acc, ln);
      hipLaunchKernelGGL(k_mb_accumulate, dim3(nblk), dim3(256), 0, st, n, O, d_idx[cur], d_nodeof[cur], d_nodes, d_acc, lvl_begin);
      hipLaunchKernelGGL(k_mb_decide, dim3((ln + 255) / 256), dim3(256), 0, st, d_nodes, d_acc, lvl_begin, lvl_end, d_cnt, node_cap, d_cnt + 1);
    } else {
      hipLaunchKernelGGL(k_bb_clear, dim3((ln + 255) / 256), dim3(256), 0, st, d_acc, ln);
      hipLaunchKernelGGL(k_bb_accumulate, dim3(nblk), dim3(256), 0, st, n, O, d_idx[cur], d_nodeof[cur], d_nodes, d_acc, lvl_begin);
      hipLaunchKernelGGL(k_bb_decide, dim3((ln + 255) / 256), dim3(256), 0, st, d_nodes, d_acc, lvl_begin, lvl_end, d_cnt, node_cap, d_cnt + 1);
    }
    hipLaunchKernelGGL(k_bb_scan_block, dim3(nscan), dim3(256), 0, st, n, O, d_idx[cur], d_nodeof[cur], d_nodes, d_pre, d_bsum);
    hipLaunchKernelGGL(k_bb_scan_sums, dim3(1), dim3(1024), 0, st, nscan, d_bsum);
    hipLaunchKernelGGL(k_bb_scatter, dim3(nblk), dim3(256), 0, st, n, O, d_idx[cur], d_nodeof[cur], d_nodes, d_pre, d_bsum, d_idx[cur ^ 1], d_nodeof[cur ^ 1]);
    BB_HIP(hipMemcpyAsync(hcnt, d_cnt, sizeof(hcnt), hipMemcpyDeviceToHost, st));
    BB_HIP(hipStreamSynchronize(st));
    if (hcnt[1]) { err = "device tree build: node pool exhausted"; goto done; }
    cur ^= 1;
    lvl_begin = lvl_end; lvl_end = hcnt[0];
  }
  BB_HIP(hipEventRecord(e1, st));
  nodes.resize((size_t)lvl_end); order.resize((size_t)n);
  BB_HIP(hipMemcpyAsync(nodes.data(), d_nodes, sizeof(DevNode) * nodes.size(), hipMemcpyDeviceToHost, st));
  BB_HIP(hipMemcpyAsync(order.data(), d_idx[cur], sizeof(int) * n, hipMemcpyDeviceToHost, st));
  BB_HIP(hipStreamSynchronize(st));
  if (gpu_ms) BB_HIP(hipEventElapsedTime(gpu_ms, e0, e1));
  ok = true;
done:
  (void)hipFree(d_box); (void)hipFree(d_mid); (void)hipFree(d_area);
  for (int q = 0; q < 2; q++) { (void)hipFree(d_idx[q]); (void)hipFree(d_nodeof[q]); }
  (void)hipFree(d_pre); (void)hipFree(d_bsum); (void)hipFree(d_cnt); (void)hipFree(d_nodes); (void)hipFree(d_acc);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return ok;
}

// breadth-first device numbering -> the preorder trees of the host builders (explicit stack: deep trees)
template <class TREE, class LEAF, class BRANCH>
inline void to_preorder(const std::vector<DevNode>& nodes, TREE& T, LEAF&& leaf, BRANCH&& branch) {
  T.nodes.clear(); T.depth = 0;
  struct Item { int dev, parent, depth; bool right; };
  std::vector<Item> stack{{0, -1, 0, false}};
  while (!stack.empty()) {
    Item it = stack.back();
    stack.pop_back();
    const DevNode& d = nodes[(size_t)it.dev];
    const int me = (int)T.nodes.size();
    T.nodes.push_back({});
    T.depth = std::max(T.depth, it.depth + 1);
    if (it.parent >= 0) (it.right ? T.nodes[(size_t)it.parent].right : T.nodes[(size_t)it.parent].left) = me;
    if (d.leaf) leaf(T.nodes[(size_t)me], d);
    else {
      branch(T.nodes[(size_t)me], d);
      stack.push_back({d.right, me, it.depth + 1, true});  // popped second: the left subtree is numbered first
      stack.push_back({d.left, me, it.depth + 1, false});
    }
  }
}

// `bih` (Bih.hs:309-324): fills T like Graph::BihBuild, `ids` = the leaf items
inline bool build(const std::vector<Box3>& boxes, const std::vector<int>& ids, const Box3& bb, BihTree& T, hipStream_t st, std::string& err, float* gpu_ms) {
  std::vector<DevNode> nodes;
  std::vector<int> order;
  if (!levels<false>(boxes, bb, st, err, gpu_ms, nodes, order)) return false;
  T.bb = bb;
  to_preorder(nodes, T,
    [&](BihTree::Node& tn, const DevNode& d) {
      tn.leaf = true; tn.lsplit = tn.rsplit = 0; tn.axis = -1; tn.left = tn.right = -1;
      for (int k = 0; k < d.count; k++) tn.items.push_back(ids[(size_t)order[(size_t)(d.start + k)]]);
    },
    [&](BihTree::Node& tn, const DevNode& d) { tn.leaf = false; tn.lsplit = d.lsplit; tn.rsplit = d.rsplit; tn.axis = d.axis; tn.left = tn.right = -1; });
  return true;
}
// the Mesh's BVH (build_tree, Mesh.hs:69-113): fills M.nodes like Graph::MeshBuild; tbb = the triangles' boxes
inline bool build_mesh(const std::vector<Box3>& tbb, MeshData& M, hipStream_t st, std::string& err, float* gpu_ms) {
  std::vector<DevNode> nodes;
  std::vector<int> order;
  if (!levels<true>(tbb, M.bb, st, err, gpu_ms, nodes, order)) return false;
  auto box_of = [](const DevNode& d) { return Box3{{d.lo[0], d.lo[1], d.lo[2]}, {d.hi[0], d.hi[1], d.hi[2]}}; };
  to_preorder(nodes, M,
    [&](MeshData::Node& tn, const DevNode& d) {
      tn.leaf = true; tn.left = tn.right = -1;
      for (int k = 0; k < d.count; k++) tn.tris.push_back(order[(size_t)(d.start + k)]);
    },
    [&](MeshData::Node& tn, const DevNode& d) { tn.leaf = false; tn.lbb = box_of(nodes[(size_t)d.left]); tn.rbb = box_of(nodes[(size_t)d.right]); tn.left = tn.right = -1; });
  return true;
}
#undef BB_HIP

}  // namespace bihdev
}  // namespace glome
