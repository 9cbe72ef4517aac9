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

// ---- round 2: integer / literal / modifier classes met in the render loops, and the loops' trips as instruction sequences ----
// (an asm block that contains s_and_b64 must declare "scc": the loop counter's s_cmp may sit before the block)
#define K4(NAME, INSTR, CLOB) KERNEL(NAME, asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(b), "v"(c), "v"(d) : CLOB);)
#define I_ADDU(D) "v_add_u32 " #D ", %4, %5"
#define I_ADDU_S(D) "v_add_u32 " #D ", s20, %5"
#define I_ADDU_LIT(D) "v_add_u32 " #D ", 0x108, %5"
#define I_ANDLIT(D) "v_and_b32 " #D ", 0xff, %4"
#define I_MAD24(D) "v_mad_u32_u24 " #D ", %4, 48, %5"
#define I_MAD24S(D) "v_mad_u32_u24 " #D ", %4, 48, s20"
#define I_LSHLADD(D) "v_lshl_add_u32 " #D ", %4, 4, %5"
#define I_ADDF_INL(D) "v_add_f32 " #D ", -1.0, %4"
#define I_ADDF_LIT(D) "v_add_f32 " #D ", 0xc0400000, %4"
#define I_MULF_LIT(D) "v_mul_f32 " #D ", 0x3f7d70a4, %4"
#define I_MINF_LIT(D) "v_min_f32 " #D ", 0x3f7d70a4, %4"
#define I_MAXF(D) "v_max_f32 " #D ", %4, %5"
#define I_MED3(D) "v_med3_f32 " #D ", %4, %5, %6"
#define I_MOV(D) "v_mov_b32 " #D ", %4"
#define I_MULNEG(D) "v_mul_f32_e64 " #D ", %4, -%5"
#define I_FMANEG(D) "v_fma_f32 " #D ", %4, %5, -%6"
#define I_SUBINL(D) "v_sub_f32 " #D ", 1.0, %4"
#define I_CMPU(D) "v_cmp_lt_u32 vcc, %4, %5"
#define I_CNDVCC0(D) "v_cndmask_b32 " #D ", 0, %4, vcc"
#define I_MULU24(D) "v_mul_u32_u24 " #D ", %4, %5"
#define I_LSHL(D) "v_lshlrev_b32 " #D ", 4, %4"
#define I_BFE(D) "v_bfe_u32 " #D ", %4, 8, 8"
#define I_CVT(D) "v_cvt_f32_u32 " #D ", %4"
K4(k_addu, I_ADDU, "vcc")
K4(k_addu_s, I_ADDU_S, "s20")
K4(k_addu_lit, I_ADDU_LIT, "vcc")
K4(k_andlit, I_ANDLIT, "vcc")
K4(k_mad24, I_MAD24, "vcc")
K4(k_mad24s, I_MAD24S, "s20")
K4(k_lshladd, I_LSHLADD, "vcc")
K4(k_addf_inl, I_ADDF_INL, "vcc")
K4(k_addf_lit, I_ADDF_LIT, "vcc")
K4(k_mulf_lit, I_MULF_LIT, "vcc")
K4(k_minf_lit, I_MINF_LIT, "vcc")
K4(k_maxf, I_MAXF, "vcc")
K4(k_med3, I_MED3, "vcc")
K4(k_mov, I_MOV, "vcc")
K4(k_mulneg, I_MULNEG, "vcc")
K4(k_fmaneg, I_FMANEG, "vcc")
K4(k_subinl, I_SUBINL, "vcc")
K4(k_cmpu, I_CMPU, "vcc")
K4(k_cndvcc0, I_CNDVCC0, "vcc")
K4(k_mulu24, I_MULU24, "vcc")
K4(k_lshl, I_LSHL, "vcc")
K4(k_bfe, I_BFE, "vcc")
K4(k_cvt, I_CVT, "vcc")
// is the VCC-reading VOP2 v_cndmask as slow in context as back to back? (23 cycles back to back, 4.2 for the VOP3 form with an SGPR pair)
KERNEL(k_cnd_mix, asm volatile("v_cndmask_b32 %0, 0, %4, vcc\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_cmp_cndvcc, asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %0, 0, %4, vcc\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_cmp_cnd64, asm volatile("v_cmp_lt_f32 s[20:21], %4, %5\n v_cndmask_b32_e64 %0, 0, %4, s[20:21]\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_sand_cndvcc, asm volatile("v_cmp_lt_f32 vcc, %4, %5\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32 %0, 0, %4, vcc\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21");)
// the backward's trip as the compiler emits it (32 VALU + 2 SALU; operands from registers)
#define TRIP_HEAD \
  "v_sub_f32 %0, %4, %5\n v_sub_f32 %1, %5, %6\n v_mul_f32_e64 %2, %4, -%0\n v_fmac_f32 %2, %5, %1\n v_mul_f32 %3, %6, %0\n" \
  "v_mul_f32 %2, %1, %2\n v_fmac_f32 %2, %0, %3\n v_exp_f32 %3, %2\n"
#define TRIP_VALU TRIP_HEAD \
  "v_cmp_lt_u32 vcc, %4, %5\n v_cmp_nlt_f32 s[20:21], 0, %2\n" \
  "v_mul_f32 %0, %5, %3\n v_min_f32 %0, 0x3f7d70a4, %0\n v_mul_f32 %1, %4, %5\n s_and_b64 s[20:21], vcc, s[20:21]\n v_cmp_ngt_f32 vcc, s22, %0\n" \
  "v_fmac_f32 %1, %4, %6\n v_fmac_f32 %1, %5, %6\n s_and_b64 vcc, s[20:21], vcc\n v_fmac_f32 %1, %4, %4\n v_cndmask_b32 %0, 0, %0, vcc\n" \
  "v_fmac_f32 %1, %5, %5\n v_sub_f32 %2, 1.0, %0\n v_and_b32 %3, 0xff, %3\n v_cndmask_b32 %3, 0, %3, vcc\n v_rcp_f32 %2, %2\n" \
  "v_mad_u32_u24 %3, %3, 48, s22\n v_mul_f32 %0, %4, %0\n v_fmac_f32 %1, %0, %1\n v_sub_f32 %0, %5, %1\n v_mul_f32 %0, %2, %0\n" \
  "v_fma_f32 %0, %4, %1, -%0\n v_mul_f32 %1, %4, %2\n v_add_u32 %3, s22, %3\n v_mul_f32 %0, %3, %0\n"
#define TRIP_IO : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(b), "v"(c), "v"(d) : "vcc", "scc", "s20", "s21", "s22"
KERNEL(k_trip, asm volatile(TRIP_VALU TRIP_IO);)
// candidate trims: (a) no list-position test; (abc) additionally one v_cndmask (alpha = min(o G_eff, 0.99)) and no address arithmetic
KERNEL(k_trip_a, asm volatile(TRIP_HEAD
  "v_cmp_nlt_f32 s[20:21], 0, %2\n"
  "v_mul_f32 %0, %5, %3\n v_min_f32 %0, 0x3f7d70a4, %0\n v_mul_f32 %1, %4, %5\n v_cmp_ngt_f32 vcc, s22, %0\n"
  "v_fmac_f32 %1, %4, %6\n v_fmac_f32 %1, %5, %6\n s_and_b64 vcc, s[20:21], vcc\n v_fmac_f32 %1, %4, %4\n v_cndmask_b32 %0, 0, %0, vcc\n"
  "v_fmac_f32 %1, %5, %5\n v_sub_f32 %2, 1.0, %0\n v_and_b32 %3, 0xff, %3\n v_cndmask_b32 %3, 0, %3, vcc\n v_rcp_f32 %2, %2\n"
  "v_mad_u32_u24 %3, %3, 48, s22\n v_mul_f32 %0, %4, %0\n v_fmac_f32 %1, %0, %1\n v_sub_f32 %0, %5, %1\n v_mul_f32 %0, %2, %0\n"
  "v_fma_f32 %0, %4, %1, -%0\n v_mul_f32 %1, %4, %2\n v_add_u32 %3, s22, %3\n v_mul_f32 %0, %3, %0" TRIP_IO);)
KERNEL(k_trip_abc, asm volatile(TRIP_HEAD
  "v_cmp_nlt_f32 s[20:21], 0, %2\n"
  "v_mul_f32 %0, %5, %3\n v_mul_f32 %1, %4, %5\n v_cmp_ngt_f32 vcc, s22, %0\n"
  "v_fmac_f32 %1, %4, %6\n v_fmac_f32 %1, %5, %6\n s_and_b64 vcc, s[20:21], vcc\n v_fmac_f32 %1, %4, %4\n v_cndmask_b32 %3, 0, %3, vcc\n"
  "v_fmac_f32 %1, %5, %5\n v_mul_f32 %0, %5, %3\n v_min_f32 %0, 0x3f7d70a4, %0\n v_sub_f32 %2, 1.0, %0\n v_rcp_f32 %2, %2\n"
  "v_mul_f32 %0, %4, %0\n v_fmac_f32 %1, %0, %1\n v_sub_f32 %0, %5, %1\n v_mul_f32 %0, %2, %0\n"
  "v_fma_f32 %0, %4, %1, -%0\n v_mul_f32 %1, %4, %2\n v_mul_f32 %0, %3, %0" TRIP_IO);)
// the forward's trip (25 VALU + 3 SALU)
KERNEL(k_ftrip, asm volatile(TRIP_HEAD
  "v_cmp_nlt_f32 s[20:21], 0, %2\n"
  "v_mul_f32 %0, %5, %3\n v_min_f32 %0, 0x3f7d70a4, %0\n v_cmp_ngt_f32 vcc, s22, %0\n v_sub_f32 %1, 1.0, %0\n v_mul_f32 %1, %1, %6\n"
  "s_and_b64 s[20:21], s[20:21], vcc\n v_cmp_gt_u32 vcc, 0x38d1b717, %1\n s_and_b64 vcc, s[20:21], vcc\n s_andn2_b64 s[20:21], s[20:21], vcc\n"
  "v_cndmask_b32_e64 %0, 0, %0, s[20:21]\n v_mul_f32 %0, %0, %6\n v_cndmask_b32_e64 %2, %2, %1, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
  "v_fmac_f32 %1, %0, %4\n v_fmac_f32 %2, %0, %5\n v_fmac_f32 %3, %0, %6\n v_fmac_f32 %1, %0, %5\n v_fmac_f32 %2, %0, %6\n v_fmac_f32 %3, %0, %4" TRIP_IO);)
// Do the LDS pipe and the VALU overlap? The backward's trip with the LDS traffic of the real loop: three reads of the next entry
// at a per-quad address (b128 + b128 + b96: 44 B per lane), one sub-list byte, one u/v store; LDSKB KB of LDS per wave so that
// 16 (LDSKB = 10) or 32 (5) waves fit a CU like the render kernels. LDSOPS = 0: the same kernel without the LDS instructions.
#define TRIP_LDS \
  "ds_read_b128 v[100:103], %7\n ds_read_b128 v[104:107], %7 offset:16\n ds_read_b96 v[108:110], %7 offset:32\n" \
  "ds_read_u8 v111, %8\n ds_write2st64_b32 %9, %4, %5 offset1:9\n s_waitcnt lgkmcnt(5)\n"
template <int LDSOPS, int LDSKB>
__global__ __launch_bounds__(256) void k_trip_lds(float* out, int lane_sel) {
  __shared__ float lds[4 * LDSKB * 256];
  for (int i = threadIdx.x; i < 4 * LDSKB * 256; i += 256) lds[i] = 0.f;
  __syncthreads();
  float b = 1.0001f, c = 0.5f, d = 0.25f, e0, e1, e2, e3;
  const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned a_ent = w * (LDSKB * 1024) + (lane >> 4) * 48;  // one entry per quad (DPP row)
  const unsigned a_idx = w * (LDSKB * 1024) + 3200 + (lane >> 4) * 80;
  const unsigned a_uv = w * (LDSKB * 1024) + 3600 + lane * 4;
  for (int i = 0; i < ITER; i++) {
    if (LDSOPS) { REP16(asm volatile(TRIP_LDS TRIP_VALU : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(b), "v"(c), "v"(d), "v"(a_ent), "v"(a_idx), "v"(a_uv)
        : "vcc", "scc", "s20", "s21", "s22", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111");) }
    else { REP16(asm volatile(TRIP_VALU : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(b), "v"(c), "v"(d), "v"(a_ent), "v"(a_idx), "v"(a_uv)
        : "vcc", "scc", "s20", "s21", "s22", "memory");) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = e0 + e1 + e2 + e3 + lds[threadIdx.x];
}

static int g_grid = 2048;
template <typename K>
void run(const char* name, K k, float* out, int per_iter_instrs, double clk_ghz) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = g_grid;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 3);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 3);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double waves = grid * 4.0, instrs = waves * ITER * 16.0 * per_iter_instrs;
  double simd_cycles = ms * 1e-3 * clk_ghz * 1e9 * 1024.0;
  printf("%-20s %8.3f ms  %6.2f cycles/wave-instr/SIMD (%d waves/SIMD)\n", name, ms, simd_cycles / instrs, grid / 256);
}
int main() {
  setvbuf(stdout, NULL, _IOLBF, 0);
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
  for (int pass = 0; pass < 2; pass++) {
    g_grid = pass ? 1024 : 2048;  // 8 and 4 waves/SIMD
    run("v_add_u32 x4", k_addu, out, 4, clk);
    run("v_add_u32 sgpr x4", k_addu_s, out, 4, clk);
    run("v_add_u32 lit x4", k_addu_lit, out, 4, clk);
    run("v_and lit x4", k_andlit, out, 4, clk);
    run("mad_u32_u24 x4", k_mad24, out, 4, clk);
    run("mad_u32_u24 sgpr", k_mad24s, out, 4, clk);
    run("lshl_add_u32 x4", k_lshladd, out, 4, clk);
    run("add_f32 inline", k_addf_inl, out, 4, clk);
    run("add_f32 literal", k_addf_lit, out, 4, clk);
    run("mul_f32 literal", k_mulf_lit, out, 4, clk);
    run("min_f32 literal", k_minf_lit, out, 4, clk);
    run("max_f32 x4", k_maxf, out, 4, clk);
    run("med3_f32 x4", k_med3, out, 4, clk);
    run("v_mov x4", k_mov, out, 4, clk);
    run("mul_e64 neg x4", k_mulneg, out, 4, clk);
    run("fma neg x4", k_fmaneg, out, 4, clk);
    run("sub 1.0 x4", k_subinl, out, 4, clk);
    run("cmp_lt_u32 vcc x4", k_cmpu, out, 4, clk);
    run("cndmask 0,v,vcc", k_cndvcc0, out, 4, clk);
    run("mul_u32_u24 x4", k_mulu24, out, 4, clk);
    run("lshlrev x4", k_lshl, out, 4, clk);
    run("bfe_u32 x4", k_bfe, out, 4, clk);
    run("cvt_f32_u32 x4", k_cvt, out, 4, clk);
    run("cnd_vcc+3fma /4", k_cnd_mix, out, 4, clk);
    run("cmp,cnd_vcc,2fma /4", k_cmp_cndvcc, out, 4, clk);
    run("cmp,cnd64 sgpr,2fma", k_cmp_cnd64, out, 4, clk);
    run("cmp,s_and,cnd_vcc /5", k_sand_cndvcc, out, 5, clk);
    run("bwd trip /34", k_trip, out, 34, clk);
    run("bwd trip (a) /32", k_trip_a, out, 32, clk);
    run("bwd trip (abc) /29", k_trip_abc, out, 29, clk);
    run("fwd trip /28", k_ftrip, out, 28, clk);
  }
  g_grid = 1024;
  run("trip no LDS 4w /34", k_trip_lds<0, 10>, out, 34, clk);
  run("trip + LDS 4w /34", k_trip_lds<1, 10>, out, 34, clk);
  g_grid = 2048;
  run("trip no LDS 8w /34", k_trip_lds<0, 5>, out, 34, clk);
  run("trip + LDS 8w /34", k_trip_lds<1, 5>, out, 34, clk);
  return 0;
}
