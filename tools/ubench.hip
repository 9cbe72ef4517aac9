// Micro-benchmark: issue cost of the instructions the render kernels are built from (gfx950).
// Each kernel runs ITER iterations of 16 copies of one instruction; 2048 workgroups x 256 threads = 8 waves/SIMD.
// Prints cycles per wave-instruction per SIMD assuming the clock reported by hipDeviceProp (upper bound).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 2048
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, BODY)                                               \
  __global__ __launch_bounds__(256) void NAME(float* out, int lane_sel) { \
    float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;     \
    float e0 = a, e1 = b, e2 = c, e3 = d;                                 \
    int s = lane_sel;                                                     \
    for (int i = 0; i < ITER; i++) { REP16(BODY) }                        \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e0 + e1 + e2 + e3 + (float)s; \
  }
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(k_fma_indep, asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b), "v"(c));)
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&e0) : "v"(*(double*)&e2), "v"(*(double*)&e2));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "v"(b));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0" : "+v"(a));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(a));)
KERNEL(k_readlane, asm volatile("v_readlane_b32 %0, %1, %0\n s_and_b32 %0, %0, 63" : "+s"(s) : "v"(a));)
KERNEL(k_readlane_use, asm volatile("v_readlane_b32 s20, %1, %2\n s_nop 1\n v_add_f32 %0, s20, %0" : "+v"(b) : "v"(a), "s"(s) : "s20");)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");)
KERNEL(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");)
KERNEL(k_dppadd, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));)
KERNEL(k_dppadd4, asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));)
KERNEL(k_bcast, asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a));)
KERNEL(k_cnd_indep, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_cnd_e64, asm volatile("v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %4, %5, s[20:21]\n v_cndmask_b32_e64 %2, %4, %5, s[20:21]\n v_cndmask_b32_e64 %3, %4, %5, s[20:21]" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_pk_indep, asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3" : "+v"(*(double*)&e0), "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e2) : "v"(*(double*)&b));)
KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %4, %4\n v_pk_mul_f32 %1, %4, %4\n v_pk_mul_f32 %2, %4, %4\n v_pk_mul_f32 %3, %4, %4" : "=v"(*(double*)&e0), "=v"(*(double*)&a), "=v"(*(double*)&c), "=v"(*(double*)&e2) : "v"(*(double*)&b));)
KERNEL(k_rl4, asm volatile("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5" : : "v"(e0), "v"(e1), "v"(e2), "v"(e3) : "s20", "s21", "s22", "s23");)
KERNEL(k_rl4s, asm volatile("v_readlane_b32 s20, %0, %4\n v_readlane_b32 s21, %1, %4\n v_readlane_b32 s22, %2, %4\n v_readlane_b32 s23, %3, %4" : : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "s"(s) : "s20", "s21", "s22", "s23");)
KERNEL(k_fma_2sg, asm volatile("v_fma_f32 %0, s20, %4, %0\n v_fma_f32 %1, s20, %4, %1\n v_fma_f32 %2, s20, %4, %2\n v_fma_f32 %3, s20, %4, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b) : "s20");)
KERNEL(k_sub_sg, asm volatile("v_sub_f32 %0, s20, %4\n v_sub_f32 %1, s21, %4\n v_sub_f32 %2, s22, %4\n v_sub_f32 %3, s23, %4" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b) : "s20","s21","s22","s23");)
KERNEL(k_cmp4, asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %0, %1" : : "v"(a), "v"(b) : "s20","s21","s22","s23","s24","s25","s26","s27");)
KERNEL(k_min, asm volatile("v_min_f32 %0, %4, %5\n v_min_f32 %1, %4, %5\n v_min_f32 %2, %4, %5\n v_min_f32 %3, %4, %5" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b), "v"(c));)
KERNEL(k_mul3, asm volatile("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b), "v"(c));)
KERNEL(k_mul_inpl, asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b));)
KERNEL(k_min_inpl, asm volatile("v_min_f32 %0, %4, %0\n v_min_f32 %1, %4, %1\n v_min_f32 %2, %4, %2\n v_min_f32 %3, %4, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b));)
KERNEL(k_fma3, asm volatile("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %5, %6\n v_fma_f32 %2, %4, %5, %6\n v_fma_f32 %3, %4, %5, %6" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b), "v"(c), "v"(d));)
KERNEL(k_sub_inpl, asm volatile("v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %4, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b));)
KERNEL(k_cnd64_inpl, asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b) : "s20", "s21");)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %0, %3\n v_cmp_lt_f32 vcc, %1, %2" : : "v"(e0), "v"(e1), "v"(e2), "v"(e3) : "vcc");)
KERNEL(k_fmac_sg, asm volatile("v_fmac_f32 %0, s20, %4\n v_fmac_f32 %1, s20, %4\n v_fmac_f32 %2, s20, %4\n v_fmac_f32 %3, s20, %4" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b) : "s20");)
KERNEL(k_dsread, asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(float4*)&e0) : "v"(s) : "memory");)
KERNEL(k_dsread_pipe, asm volatile("ds_read_b128 %0, %2\n ds_read_b64 %1, %2 offset:16\n v_fma_f32 %3, %4, %4, %3\n v_fma_f32 %3, %4, %4, %3\n v_fma_f32 %3, %4, %4, %3\n v_fma_f32 %3, %4, %4, %3\n s_waitcnt lgkmcnt(0)" : "=v"(*(float4*)&e0), "=v"(*(double*)&c), "+v"(s), "+v"(a) : "v"(b) : "memory");)
KERNEL(k_salu, asm volatile("s_add_u32 %0, %0, 1\n s_and_b32 %0, %0, 63" : "+s"(s));)
KERNEL(k_sfma_sgpr, asm volatile("v_fma_f32 %0, %1, s20, %0" : "+v"(a) : "v"(b) : "s20");)

template <typename K>
void run(const char* name, K k, float* out, int per_iter_instrs, double clk_ghz) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 2048;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 3);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 3);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double waves = grid * 4.0, instrs = waves * ITER * 16.0 * per_iter_instrs;
  double simd_cycles = ms * 1e-3 * clk_ghz * 1e9 * 1024.0;
  printf("%-16s %8.3f ms  %6.2f cycles/wave-instr/SIMD (8 waves/SIMD)\n", name, ms, simd_cycles / instrs);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double clk = p.clockRate * 1e-6;
  printf("device %s clock %.2f GHz CUs %d\n", p.name, clk, p.multiProcessorCount);
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  run("v_fma dep", k_fma, out, 1, clk);
  run("v_fma x4 indep", k_fma_indep, out, 4, clk);
  run("v_pk_fma dep", k_pkfma, out, 1, clk);
  run("v_mul dep", k_mul, out, 1, clk);
  run("v_exp dep", k_exp, out, 1, clk);
  run("v_rcp dep", k_rcp, out, 1, clk);
  run("readlane+s_and", k_readlane, out, 2, clk);
  run("readlane,nop,add", k_readlane_use, out, 3, clk);
  run("v_cndmask", k_cndmask, out, 1, clk);
  run("v_cmp", k_cmp, out, 1, clk);
  run("nop+dpp add", k_dppadd, out, 2, clk);
  run("dpp add x4", k_dppadd4, out, 4, clk);
  run("nop+bcast15", k_bcast, out, 2, clk);
  run("cndmask x4 indep", k_cnd_indep, out, 4, clk);
  run("cndmask e64 x4", k_cnd_e64, out, 4, clk);
  run("pk_fma x4 indep", k_pk_indep, out, 4, clk);
  run("pk_mul x4 indep", k_pk_mul, out, 4, clk);
  run("readlane imm x4", k_rl4, out, 4, clk);
  run("readlane sgpr x4", k_rl4s, out, 4, clk);
  run("fma sgpr x4", k_fma_2sg, out, 4, clk);
  run("sub sgpr x4", k_sub_sg, out, 4, clk);
  run("v_cmp sgprdst x4", k_cmp4, out, 4, clk);
  run("v_min x4", k_min, out, 4, clk);
  run("mul 3reg x4", k_mul3, out, 4, clk);
  run("mul inplace x4", k_mul_inpl, out, 4, clk);
  run("min inplace x4", k_min_inpl, out, 4, clk);
  run("fma 3src x4", k_fma3, out, 4, clk);
  run("sub inplace x4", k_sub_inpl, out, 4, clk);
  run("cnd64 inplace x4", k_cnd64_inpl, out, 4, clk);
  run("cmp vcc x4", k_cmp_vcc, out, 4, clk);
  run("fmac sgpr x4", k_fmac_sg, out, 4, clk);
  run("ds_read_b128+wait", k_dsread, out, 1, clk);
  run("2ds+4fma+wait /6", k_dsread_pipe, out, 6, clk);
  run("s_add+s_and", k_salu, out, 2, clk);
  run("v_fma sgpr", k_sfma_sgpr, out, 1, clk);
  return 0;
}
