// valu_rate.hip -- wave64 vector-instruction throughput per SIMD on gfx950, by instruction and waves per SIMD: is a plain
// v_fma_f32 a 2-cycle or a 4-cycle slot, and what does the packed form cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define R8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float sv) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f, c = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { R8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (KIND == 1) { R8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if (KIND == 2) { R8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sv), "v"(c));) }
    if (KIND == 3) { R8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (KIND == 4) { R8(asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %4, %4, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %6, %6, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s21, s21, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21", "scc");) }
    if (KIND == 5) { R8(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");) }
    if (KIND == 6) { R8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));) }
    if (KIND == 8) { R8(asm volatile("v_pk_fma_f32 %0, %0, s[20:21], %4\n v_pk_fma_f32 %1, %1, s[20:21], %4\n v_pk_fma_f32 %2, %2, s[20:21], %4\n v_pk_fma_f32 %3, %3, s[20:21], %4\n v_pk_fma_f32 %0, %0, s[20:21], %4\n v_pk_fma_f32 %1, %1, s[20:21], %4\n v_pk_fma_f32 %2, %2, s[20:21], %4\n v_pk_fma_f32 %3, %3, s[20:21], %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc) : "s20", "s21");) }
    if (KIND == 9) { R8(asm volatile("v_pk_fma_f32 %0, %5, s[20:21], %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %5, s[20:21], %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %2, %5, s[20:21], %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %5, s[20:21], %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %5, s[20:21], %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %5, s[20:21], %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %2, %5, s[20:21], %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %5, s[20:21], %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc) : "s20", "s21");) }
    if (KIND == 10) { R8(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %1, %1, %4 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %3, %3, %4 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_mul_f32 %0, %0, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %1, %1, %4 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %3, %3, %4 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));) }
    if (KIND == 11) { R8(asm volatile("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s20, %1\n v_mul_f32 %2, s20, %2\n v_mul_f32 %3, s20, %3\n v_mul_f32 %4, s20, %4\n v_mul_f32 %5, s20, %5\n v_mul_f32 %6, s20, %6\n v_mul_f32 %7, s20, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20");) }
    if (KIND == 12) { R8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 13) { R8(asm volatile("v_mov_b32 %0, s20\n v_mov_b32 %1, s20\n v_mov_b32 %2, s20\n v_mov_b32 %3, s20\n v_mov_b32 %4, s20\n v_mov_b32 %5, s20\n v_mov_b32 %6, s20\n v_mov_b32 %7, s20" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20");) }
    if (KIND == 14) { R8(asm volatile("v_mul_f32 %0, %8, %0\n s_add_u32 s20, s20, 1\n v_mul_f32 %1, %8, %1\n s_add_u32 s21, s21, 1\n v_mul_f32 %2, %8, %2\n s_add_u32 s20, s20, 1\n v_mul_f32 %3, %8, %3\n s_add_u32 s21, s21, 1\n v_mul_f32 %4, %8, %4\n s_add_u32 s20, s20, 1\n v_mul_f32 %5, %8, %5\n s_add_u32 s21, s21, 1\n v_mul_f32 %6, %8, %6\n s_add_u32 s20, s20, 1\n v_mul_f32 %7, %8, %7\n s_add_u32 s21, s21, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "scc");) }
    if (KIND == 15) { R8(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n s_add_u32 s20, s20, 1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n s_add_u32 s21, s21, 1\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n s_add_u32 s20, s20, 1\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n s_add_u32 s21, s21, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "scc");) }
    if (KIND == 16) { R8(asm volatile("v_mul_f32 %0, %8, %0\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_mul_f32 %1, %8, %1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n v_mul_f32 %2, %8, %2\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n v_mul_f32 %3, %8, %3\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22", "s23", "scc");) }
    // ---- round 4: branches (does a branch share the scalar port?)
    if (KIND == 17) { R8(asm volatile("s_branch 1f\n1:\n s_branch 2f\n2:\n s_branch 3f\n3:\n s_branch 4f\n4:\n s_branch 5f\n5:\n s_branch 6f\n6:\n s_branch 7f\n7:\n s_branch 8f\n8:" ::: "scc");) }
    if (KIND == 18) { R8(asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc0 1f\n s_cbranch_scc0 1f\n s_cbranch_scc0 1f\n s_cbranch_scc0 1f\n s_cbranch_scc0 1f\n s_cbranch_scc0 1f\n s_cbranch_scc0 1f\n1:" ::: "scc");) }
    if (KIND == 19) { R8(asm volatile("s_branch 1f\n1:\n s_add_u32 s20, s20, 1\n s_branch 2f\n2:\n s_add_u32 s21, s21, 1\n s_branch 3f\n3:\n s_add_u32 s20, s20, 1\n s_branch 4f\n4:\n s_add_u32 s21, s21, 1\n s_branch 5f\n5:\n s_add_u32 s20, s20, 1\n s_branch 6f\n6:\n s_add_u32 s21, s21, 1\n s_branch 7f\n7:\n s_add_u32 s20, s20, 1\n s_branch 8f\n8:\n s_add_u32 s21, s21, 1" ::: "s20", "s21", "scc");) }
    if (KIND == 20) { R8(asm volatile("s_branch 1f\n1:\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_branch 2f\n2:\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_branch 3f\n3:\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_branch 4f\n4:\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");) }
    if (KIND == 21) { R8(asm volatile("s_branch 1f\n1:\n v_mul_f32 %0, %8, %0\n s_branch 2f\n2:\n v_mul_f32 %1, %8, %1\n s_branch 3f\n3:\n v_mul_f32 %2, %8, %2\n s_branch 4f\n4:\n v_mul_f32 %3, %8, %3\n s_branch 5f\n5:\n v_mul_f32 %4, %8, %4\n s_branch 6f\n6:\n v_mul_f32 %5, %8, %5\n s_branch 7f\n7:\n v_mul_f32 %6, %8, %6\n s_branch 8f\n8:\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 22) { R8(asm volatile("s_branch 1f\n1:\n v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n s_branch 2f\n2:\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n s_branch 3f\n3:\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n s_branch 4f\n4:\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (KIND == 23) { R8(asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc0 1f\n s_add_u32 s20, s20, 1\n s_cmp_eq_u32 0, 0\n s_cbranch_scc0 1f\n s_add_u32 s21, s21, 1\n s_cmp_eq_u32 0, 0\n s_cbranch_scc0 1f\n1:" ::: "s20", "s21", "scc");) }
    if (KIND == 7) { R8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));) }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount, iters = 2000;
  float* out; CK(hipMalloc(&out, (size_t)cus * 32 * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"v_fma_f32 (vgpr)", "v_pk_fma_f32", "v_fma_f32 (sgpr operand)", "v_cmp_lt_f32 -> vcc", "v_fma + s_add alternating (counted: both)", "s_add_u32", "v_pk_mul / v_pk_add", "v_fma_f32 dependent chains of 4", "v_pk_fma_f32 (sgpr pair operand)", "v_pk_fma_f32 (sgpr pair + op_sel broadcast)", "v_pk_mul_f32 (vgpr, op_sel broadcast)", "v_mul_f32 VOP2 (sgpr operand)", "v_mul_f32 VOP2 (vgpr)", "v_mov_b32 v, s", "v_mul VOP2 (vgpr) + s_add alternating (both)", "2 v_mul (vgpr) : 1 s_add (all)", "1 v_mul (vgpr) : 2 s_add (all)", "s_branch (taken)", "s_cbranch_scc0 (not taken; 1 s_cmp : 7)", "s_branch (taken) + s_add alternating (both)", "1 s_branch (taken) : 2 s_add (all)", "s_branch (taken) + v_mul (vgpr) alternating (both)", "1 s_branch (taken) : 2 v_mul (vgpr) (all)", "s_cmp + s_cbranch (not taken) + s_add (all)"};
  printf("# instructions per cycle per SIMD at an assumed 2.4 GHz (64 instr per loop body, %d iterations; kind 1/6: one instruction = 2 results per lane)\n", iters);
  const int first = getenv("KIND_FIRST") ? atoi(getenv("KIND_FIRST")) : 0;
  for (int kind = first; kind < 24; kind++)
    for (int wps : {1, 6, 8}) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        const dim3 g(cus * 4 * wps), b(64);
        switch (kind) {
          case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 8: hipLaunchKernelGGL(k<8>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 9: hipLaunchKernelGGL(k<9>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 10: hipLaunchKernelGGL(k<10>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 11: hipLaunchKernelGGL(k<11>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 12: hipLaunchKernelGGL(k<12>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 13: hipLaunchKernelGGL(k<13>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 14: hipLaunchKernelGGL(k<14>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 15: hipLaunchKernelGGL(k<15>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 16: hipLaunchKernelGGL(k<16>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 17: hipLaunchKernelGGL(k<17>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 18: hipLaunchKernelGGL(k<18>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 19: hipLaunchKernelGGL(k<19>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 20: hipLaunchKernelGGL(k<20>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 21: hipLaunchKernelGGL(k<21>, g, b, 0, 0, out, iters, 1.0001f); break;
          case 22: hipLaunchKernelGGL(k<22>, g, b, 0, 0, out, iters, 1.0001f); break;
          default: hipLaunchKernelGGL(k<23>, g, b, 0, 0, out, iters, 1.0001f); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double n = (kind == 4 || kind == 14 || kind == 19 || kind == 21 ? 128.0 : (kind == 15 || kind == 16 || kind == 20 || kind == 22 ? 96.0 : 64.0)) * iters * wps;  // instructions per SIMD
      printf("%-44s waves/SIMD %d  %8.3f ms  %.3f instr/cycle/SIMD\n", names[kind], wps, ms, n / (ms * 1e-3 * 2.4e9));
      fflush(stdout);
    }
  return 0;
}
