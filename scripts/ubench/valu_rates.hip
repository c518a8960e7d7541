// VALU issue cost of the instructions the merge loop is made of (gfx950): N dependent-free copies per loop iteration,
// 8 waves per SIMD resident, cycles per wave-instruction = elapsed SIMD cycles / instructions per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/ubench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                           \
  __global__ __launch_bounds__(512) void name(uint32_t* out, int iters) {                            \
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 5, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 9, a7 = a0 + 1; \
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;                                           \
    for (int i = 0; i < iters; ++i) { REP8(body) }                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(b0 + b1 + b2 + b3); \
  }

KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_cndmask_init, if (i == 0) asm volatile("v_cmp_gt_u32 vcc, 17, %0" :: "v"(a0) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_cndmask_sgpr, if (i == 0) asm volatile("v_cmp_gt_u32 s[20:21], 17, %0" :: "v"(a0) : "s20", "s21"); asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21");)
KERNEL(k_cndmask_indep, if (i == 0) asm volatile("v_cmp_gt_u32 vcc, 17, %0" :: "v"(a0) : "vcc"); asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %6, %7, %1, vcc\n v_cndmask_b32 %0, %2, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %1, %0, %2\n v_bfi_b32 %2, %3, %2, %4\n v_bfi_b32 %4, %5, %4, %6\n v_bfi_b32 %6, %7, %6, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_cndmask_e64vcc, asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_cmpvcc_cnd, asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_cmpvcc_cnd4, asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_execmov, asm volatile("v_cmp_gt_u32 s[22:23], 40, %0\n s_and_saveexec_b64 s[20:21], s[22:23]\n v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n v_mov_b32 %1, %2\n v_mov_b32 %3, %4\n s_or_b64 exec, exec, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0\n v_mad_u32_u24 %2, %2, %3, %2\n v_mad_u32_u24 %4, %4, %5, %4\n v_mad_u32_u24 %6, %6, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_shl64, asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %5, %1\n v_lshlrev_b64 %2, %6, %2\n v_lshlrev_b64 %3, %7, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
KERNEL(k_pkmov, asm volatile("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]\n v_pk_mov_b32 %1, %2, %1 op_sel:[1,0]\n v_pk_mov_b32 %2, %3, %2 op_sel:[1,0]\n v_pk_mov_b32 %3, %0, %3 op_sel:[1,0]" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));)
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %6, %6, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_cmp, asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %4, %5\n v_cmp_eq_u32 vcc, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
KERNEL(k_cmp_e64, asm volatile("v_cmp_eq_u32 s[20:21], %0, %1\n v_cmp_eq_u32 s[22:23], %2, %3\n v_cmp_eq_u32 s[24:25], %4, %5\n v_cmp_eq_u32 s[26:27], %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
KERNEL(k_cmp_cnd_dep, asm volatile("v_cmp_eq_u32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cmp_eq_u32 s[20:21], %4, %5\n s_nop 1\n v_cndmask_b32 %6, %6, %7, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21");)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, %1, 4\n v_bfe_u32 %2, %2, %3, 4\n v_bfe_u32 %4, %4, %5, 4\n v_bfe_u32 %6, %6, %7, 4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
KERNEL(k_cmp64, asm volatile("v_cmp_ne_u64 vcc, %0, %1\n v_cmp_ne_u64 vcc, %1, %2\n v_cmp_ne_u64 vcc, %2, %3\n v_cmp_ne_u64 vcc, %3, %0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) :: "vcc");)

// branch cost: 32 v_add per iteration plus 8 forward branches (always taken: exec/vcc never zero ... see below)
#define ADD4 "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n"
KERNEL(k_br_none, asm volatile(ADD4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
// skip-branch that is NOT taken (exec != 0): s_cbranch_execz falls through
KERNEL(k_br_nottaken, asm volatile(ADD4 "s_cbranch_execz 1f\n s_nop 0\n 1:\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
// skip-branch that IS taken (exec != 0 -> s_cbranch_execnz jumps over the s_nop)
KERNEL(k_br_taken, asm volatile(ADD4 "s_cbranch_execnz 1f\n s_nop 0\n 1:\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
// the compiler's if-block shape: save exec, mask, skip when empty (taken), restore
KERNEL(k_br_ifskip, asm volatile(ADD4 "s_mov_b64 s[22:23], 0\n s_and_saveexec_b64 s[20:21], s[22:23]\n s_cbranch_execz 1f\n v_mov_b32 %0, %1\n 1:\n s_or_b64 exec, exec, s[20:21]\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_br_ifrun, asm volatile(ADD4 "s_mov_b64 s[22:23], -1\n s_and_saveexec_b64 s[20:21], s[22:23]\n s_cbranch_execz 1f\n v_mov_b32 %0, %1\n 1:\n s_or_b64 exec, exec, s[20:21]\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_salu, asm volatile(ADD4 "s_or_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[20:21], s[24:25]\n s_or_b64 s[24:25], s[22:23], s[20:21]\n s_xor_b64 s[20:21], s[24:25], s[22:23]\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23", "s24", "s25", "scc");)

int main() {
  uint32_t* out; hipMalloc(&out, 256 * 4 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const int iters = 20000, nblk = 256 * 4;      // 4 workgroups of 8 waves per CU = 8 waves per SIMD
  struct { const char* n; void (*k)(uint32_t*, int); int per_iter; } ks[] = {
    {"v_add_u32", k_add, 32}, {"4 v_add (per group)", k_br_none, 8}, {"4 v_add + branch not taken (per group)", k_br_nottaken, 8}, {"4 v_add + branch taken (per group)", k_br_taken, 8},
    {"4 v_add + if-block skipped (per group)", k_br_ifskip, 8}, {"4 v_add + if-block run, 1 v_mov (per group)", k_br_ifrun, 8}, {"4 v_add + 4 dependent s_or/s_and (per group)", k_salu, 8}, {"v_cndmask_b32 (vcc)", k_cndmask, 32}, {"v_cndmask_b32 (vcc set)", k_cndmask_init, 32}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr, 32}, {"v_cndmask_b32 (vcc, 3 distinct regs)", k_cndmask_indep, 32}, {"v_bfi_b32", k_bfi, 32}, {"v_mov_b32", k_mov, 32}, {"v_cndmask_b32_e64 (vcc)", k_cndmask_e64vcc, 32}, {"v_cmp vcc + v_cndmask vcc (per instr)", k_cmpvcc_cnd, 32}, {"v_cmp vcc + 3 v_cndmask vcc (per instr)", k_cmpvcc_cnd4, 32}, {"cmp+saveexec+6 v_mov+restore (per group)", k_execmov, 8}, {"v_mul_lo_u32", k_mul_lo, 32}, {"v_mad_u32_u24", k_mad24, 32},
    {"v_lshlrev_b64", k_shl64, 32}, {"v_pk_mov_b32", k_pkmov, 32}, {"v_max3_f32", k_max3, 32}, {"v_cmp_eq_u32 vcc", k_cmp, 32},
    {"v_cmp_eq_u32 sgpr", k_cmp_e64, 32}, {"v_cmp+s_nop 1+v_cndmask (per pair)", k_cmp_cnd_dep, 16}, {"v_bfe_u32", k_bfe, 32}, {"v_cmp_ne_u64", k_cmp64, 32}};
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.k, dim3(nblk), dim3(512), 0, 0, out, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k.k, dim3(nblk), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    const double inst_per_simd = 8.0 * (double)iters * k.per_iter;        // 8 waves per SIMD
    printf("%-40s %7.3f ms  %.2f cycles per wave-instruction (clock %d MHz)\n", k.n, ms, cycles / inst_per_simd, clk_khz / 1000);
  }
  return 0;
}
