// rccl_stub.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so with the six entry points glome_multi_* resolves
// (glome_device.hip, struct Rccl), so that the RCCL branch of glome_multi_render -- ncclCommInitAll, one group of ncclSend /
// ncclRecv per call on the ranks' own streams, the slab offsets, the stream ordering -- can execute on a box with ONE GPU
// (GLOME_DEBUG_RCCL_LIB=<this library>, GLOME_DEBUG_RCCL_SAME_DEVICE=1).  A send / recv pair of a group becomes, at
// ncclGroupEnd: an event on the sender's stream, a wait for it on the receiver's stream, a hipMemcpyAsync there, and an event
// back so that the sender's stream does not run ahead of the transfer (RCCL's send kernel occupies the sender's stream until
// the data has left).  Nothing in the product links or loads this file unless the debug variable names it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

namespace {
struct Comm { int rank, nranks, device; };
struct Op { bool send; const void* src; void* dst; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
std::vector<Op> g_ops;
int g_depth = 0;
int g_groups = 0, g_pairs = 0;
size_t dtype_bytes(int t) { return (t == 0 || t == 1) ? 1 : ((t == 2 || t == 3 || t == 7) ? 4 : ((t == 4 || t == 5 || t == 8) ? 8 : 2)); }
int flush() {
  std::vector<bool> used(g_ops.size(), false);
  for (size_t i = 0; i < g_ops.size(); i++) {
    if (!g_ops[i].send) continue;
    const Op& s = g_ops[i];
    size_t j = 0;
    for (; j < g_ops.size(); j++)
      if (!used[j] && !g_ops[j].send && g_ops[j].comm->rank == s.peer && g_ops[j].peer == s.comm->rank && g_ops[j].bytes == s.bytes) break;
    if (j == g_ops.size()) { fprintf(stderr, "rccl_stub: send %d -> %d of %zu bytes has no matching recv\n", s.comm->rank, s.peer, s.bytes); g_ops.clear(); return 1; }
    used[i] = used[j] = true;
    const Op& r = g_ops[j];
    hipEvent_t ready, moved;
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&moved, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(ready, s.stream) != hipSuccess || hipStreamWaitEvent(r.stream, ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(r.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return 1;
    if (hipEventRecord(moved, r.stream) != hipSuccess || hipStreamWaitEvent(s.stream, moved, 0) != hipSuccess) return 1;
    (void)hipEventDestroy(ready); (void)hipEventDestroy(moved);  // (destroyed once complete; the work already enqueued keeps them alive)
    g_pairs++;
  }
  for (size_t j = 0; j < g_ops.size(); j++) if (!used[j]) { fprintf(stderr, "rccl_stub: unmatched recv\n"); g_ops.clear(); return 1; }
  g_ops.clear();
  return 0;
}
}  // namespace

extern "C" {
int ncclCommInitAll(void** comms, int n, const int* devs) {
  for (int i = 0; i < n; i++) comms[i] = new Comm{i, n, devs ? devs[i] : i};
  return 0;
}
int ncclCommDestroy(void* c) { delete (Comm*)c; return 0; }
int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() {
  if (--g_depth > 0) return 0;
  g_groups++;
  return flush();
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) {
  g_ops.push_back(Op{true, buf, nullptr, count * dtype_bytes(dtype), peer, (Comm*)comm, st});
  return g_depth > 0 ? 0 : flush();
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t st) {
  g_ops.push_back(Op{false, nullptr, buf, count * dtype_bytes(dtype), peer, (Comm*)comm, st});
  return g_depth > 0 ? 0 : flush();
}
const char* ncclGetErrorString(int) { return "rccl_stub error"; }
// what the test reads back: groups closed and send / recv pairs moved so far
void rccl_stub_counts(int* groups, int* pairs) { *groups = g_groups; *pairs = g_pairs; }
}
