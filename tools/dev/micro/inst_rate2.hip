// inst_rate2.hip — round 5's extension of inst_rate.hip: the issue cost (clocks per wave64 instruction per SIMD, 8 waves per
// SIMD, every CU) of the instructions a cheaper per-lane node step could be built from — integer VOP2, three-operand integer
// forms, compares, packed 16-bit arithmetic, output / input modifiers (clamp, neg, abs), DPP — and of MIXES of a 2-clock and a
// 4-clock instruction (is the class-weighted VALU-time model additive?).  VERDICT r04 item 1(a).
// build: hipcc --offload-arch=gfx950 -O3 inst_rate2.hip -o inst_rate2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITER = 1024;
#define R8(X) X X X X X X X X
// I4(op-with-%N placeholders): the same instruction on the four accumulators
#define ONE(S, D) S(D)

#define OPS(F)                                                                                                                   \
	F(0, "v_fma_f32", "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3")                  \
	F(1, "v_mul_f32", "v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3")                                  \
	F(2, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_cvt_f32_ubyte1 %3, %4")                 \
	F(3, "v_or_b32", "v_or_b32 %0, %4, %0\n\tv_or_b32 %1, %4, %1\n\tv_or_b32 %2, %4, %2\n\tv_or_b32 %3, %4, %3")                                      \
	F(4, "v_xor_b32", "v_xor_b32 %0, %4, %0\n\tv_xor_b32 %1, %4, %1\n\tv_xor_b32 %2, %4, %2\n\tv_xor_b32 %3, %4, %3")                                  \
	F(5, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0\n\tv_lshlrev_b32 %1, 3, %1\n\tv_lshlrev_b32 %2, 3, %2\n\tv_lshlrev_b32 %3, 3, %3")                   \
	F(6, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0\n\tv_lshrrev_b32 %1, 3, %1\n\tv_lshrrev_b32 %2, 3, %2\n\tv_lshrrev_b32 %3, 3, %3")                   \
	F(7, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0\n\tv_ashrrev_i32 %1, 3, %1\n\tv_ashrrev_i32 %2, 3, %2\n\tv_ashrrev_i32 %3, 3, %3")                   \
	F(8, "v_lshlrev_b32 (vgpr shift)", "v_lshlrev_b32 %0, %5, %0\n\tv_lshlrev_b32 %1, %5, %1\n\tv_lshlrev_b32 %2, %5, %2\n\tv_lshlrev_b32 %3, %5, %3")   \
	F(9, "v_sub_u32", "v_sub_u32 %0, %4, %0\n\tv_sub_u32 %1, %4, %1\n\tv_sub_u32 %2, %4, %2\n\tv_sub_u32 %3, %4, %3")                                  \
	F(10, "v_min_u32", "v_min_u32 %0, %4, %0\n\tv_min_u32 %1, %4, %1\n\tv_min_u32 %2, %4, %2\n\tv_min_u32 %3, %4, %3")                                 \
	F(11, "v_max_u32", "v_max_u32 %0, %4, %0\n\tv_max_u32 %1, %4, %1\n\tv_max_u32 %2, %4, %2\n\tv_max_u32 %3, %4, %3")                                 \
	F(12, "v_min_f32", "v_min_f32 %0, %4, %0\n\tv_min_f32 %1, %4, %1\n\tv_min_f32 %2, %4, %2\n\tv_min_f32 %3, %4, %3")                                 \
	F(13, "v_mul_u32_u24", "v_mul_u32_u24 %0, %4, %0\n\tv_mul_u32_u24 %1, %4, %1\n\tv_mul_u32_u24 %2, %4, %2\n\tv_mul_u32_u24 %3, %4, %3")             \
	F(14, "v_mul_lo_u32", "v_mul_lo_u32 %0, %4, %0\n\tv_mul_lo_u32 %1, %4, %1\n\tv_mul_lo_u32 %2, %4, %2\n\tv_mul_lo_u32 %3, %4, %3")                  \
	F(15, "v_and_or_b32", "v_and_or_b32 %0, %4, %5, %0\n\tv_and_or_b32 %1, %4, %5, %1\n\tv_and_or_b32 %2, %4, %5, %2\n\tv_and_or_b32 %3, %4, %5, %3")   \
	F(16, "v_lshl_or_b32", "v_lshl_or_b32 %0, %4, 2, %0\n\tv_lshl_or_b32 %1, %4, 2, %1\n\tv_lshl_or_b32 %2, %4, 2, %2\n\tv_lshl_or_b32 %3, %4, 2, %3")  \
	F(17, "v_or3_b32", "v_or3_b32 %0, %4, %5, %0\n\tv_or3_b32 %1, %4, %5, %1\n\tv_or3_b32 %2, %4, %5, %2\n\tv_or3_b32 %3, %4, %5, %3")                 \
	F(18, "v_add3_u32", "v_add3_u32 %0, %4, %5, %0\n\tv_add3_u32 %1, %4, %5, %1\n\tv_add3_u32 %2, %4, %5, %2\n\tv_add3_u32 %3, %4, %5, %3")             \
	F(19, "v_bfi_b32", "v_bfi_b32 %0, %4, %5, %0\n\tv_bfi_b32 %1, %4, %5, %1\n\tv_bfi_b32 %2, %4, %5, %2\n\tv_bfi_b32 %3, %4, %5, %3")                 \
	F(20, "v_alignbit_b32", "v_alignbit_b32 %0, %4, %0, 16\n\tv_alignbit_b32 %1, %4, %1, 16\n\tv_alignbit_b32 %2, %4, %2, 16\n\tv_alignbit_b32 %3, %4, %3, 16") \
	F(21, "v_alignbyte_b32", "v_alignbyte_b32 %0, %4, %0, 1\n\tv_alignbyte_b32 %1, %4, %1, 1\n\tv_alignbyte_b32 %2, %4, %2, 1\n\tv_alignbyte_b32 %3, %4, %3, 1") \
	F(22, "v_cmp_lt_u32 (vcc)", "v_cmp_lt_u32 vcc, %4, %5\n\tv_cmp_lt_u32 vcc, %4, %5\n\tv_cmp_lt_u32 vcc, %4, %5\n\tv_cmp_lt_u32 vcc, %4, %5")          \
	F(23, "v_cmp_lt_f32 (sgpr pair, e64)", "v_cmp_lt_f32 s[20:21], %4, %5\n\tv_cmp_lt_f32 s[22:23], %4, %5\n\tv_cmp_lt_f32 s[20:21], %4, %5\n\tv_cmp_lt_f32 s[22:23], %4, %5") \
	F(24, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %4, %0, vcc\n\tv_cndmask_b32 %1, %4, %1, vcc\n\tv_cndmask_b32 %2, %4, %2, vcc\n\tv_cndmask_b32 %3, %4, %3, vcc") \
	F(25, "v_max3_f32 clamp", "v_max3_f32 %0, %4, %5, %0 clamp\n\tv_max3_f32 %1, %4, %5, %1 clamp\n\tv_max3_f32 %2, %4, %5, %2 clamp\n\tv_max3_f32 %3, %4, %5, %3 clamp") \
	F(26, "v_max3_f32", "v_max3_f32 %0, %4, %5, %0\n\tv_max3_f32 %1, %4, %5, %1\n\tv_max3_f32 %2, %4, %5, %2\n\tv_max3_f32 %3, %4, %5, %3")             \
	F(27, "v_add_f32 clamp (e64)", "v_add_f32_e64 %0, %4, %0 clamp\n\tv_add_f32_e64 %1, %4, %1 clamp\n\tv_add_f32_e64 %2, %4, %2 clamp\n\tv_add_f32_e64 %3, %4, %3 clamp") \
	F(28, "v_add_f32 neg (e64)", "v_add_f32_e64 %0, -%4, %0\n\tv_add_f32_e64 %1, -%4, %1\n\tv_add_f32_e64 %2, -%4, %2\n\tv_add_f32_e64 %3, -%4, %3")     \
	F(29, "v_fma_f32 neg", "v_fma_f32 %0, %4, %5, -%0\n\tv_fma_f32 %1, %4, %5, -%1\n\tv_fma_f32 %2, %4, %5, -%2\n\tv_fma_f32 %3, %4, %5, -%3")          \
	F(30, "v_fma_f32 clamp", "v_fma_f32 %0, %4, %5, %0 clamp\n\tv_fma_f32 %1, %4, %5, %1 clamp\n\tv_fma_f32 %2, %4, %5, %2 clamp\n\tv_fma_f32 %3, %4, %5, %3 clamp") \
	F(31, "v_pk_fma_f16", "v_pk_fma_f16 %0, %4, %5, %0\n\tv_pk_fma_f16 %1, %4, %5, %1\n\tv_pk_fma_f16 %2, %4, %5, %2\n\tv_pk_fma_f16 %3, %4, %5, %3")   \
	F(32, "v_pk_max_f16", "v_pk_max_f16 %0, %4, %0\n\tv_pk_max_f16 %1, %4, %1\n\tv_pk_max_f16 %2, %4, %2\n\tv_pk_max_f16 %3, %4, %3")                   \
	F(33, "v_pk_min_f16", "v_pk_min_f16 %0, %4, %0\n\tv_pk_min_f16 %1, %4, %1\n\tv_pk_min_f16 %2, %4, %2\n\tv_pk_min_f16 %3, %4, %3")                   \
	F(34, "v_pk_add_f16", "v_pk_add_f16 %0, %4, %0\n\tv_pk_add_f16 %1, %4, %1\n\tv_pk_add_f16 %2, %4, %2\n\tv_pk_add_f16 %3, %4, %3")                   \
	F(35, "v_pk_mul_f16", "v_pk_mul_f16 %0, %4, %0\n\tv_pk_mul_f16 %1, %4, %1\n\tv_pk_mul_f16 %2, %4, %2\n\tv_pk_mul_f16 %3, %4, %3")                   \
	F(36, "v_pk_add_u16", "v_pk_add_u16 %0, %4, %0\n\tv_pk_add_u16 %1, %4, %1\n\tv_pk_add_u16 %2, %4, %2\n\tv_pk_add_u16 %3, %4, %3")                   \
	F(37, "v_pk_min_u16", "v_pk_min_u16 %0, %4, %0\n\tv_pk_min_u16 %1, %4, %1\n\tv_pk_min_u16 %2, %4, %2\n\tv_pk_min_u16 %3, %4, %3")                   \
	F(38, "v_pk_max_u16", "v_pk_max_u16 %0, %4, %0\n\tv_pk_max_u16 %1, %4, %1\n\tv_pk_max_u16 %2, %4, %2\n\tv_pk_max_u16 %3, %4, %3")                   \
	F(39, "v_pk_mad_u16", "v_pk_mad_u16 %0, %4, %5, %0\n\tv_pk_mad_u16 %1, %4, %5, %1\n\tv_pk_mad_u16 %2, %4, %5, %2\n\tv_pk_mad_u16 %3, %4, %5, %3")   \
	F(40, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %4, %0\n\tv_pk_mul_lo_u16 %1, %4, %1\n\tv_pk_mul_lo_u16 %2, %4, %2\n\tv_pk_mul_lo_u16 %3, %4, %3")    \
	F(41, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 3, %0\n\tv_pk_lshlrev_b16 %1, 3, %1\n\tv_pk_lshlrev_b16 %2, 3, %2\n\tv_pk_lshlrev_b16 %3, 3, %3")    \
	F(42, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %4, %0\n\tv_cvt_pkrtz_f16_f32 %1, %4, %1\n\tv_cvt_pkrtz_f16_f32 %2, %4, %2\n\tv_cvt_pkrtz_f16_f32 %3, %4, %3") \
	F(43, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %4\n\tv_cvt_f32_u32 %1, %4\n\tv_cvt_f32_u32 %2, %4\n\tv_cvt_f32_u32 %3, %4")                             \
	F(44, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %4\n\tv_cvt_u32_f32 %1, %4\n\tv_cvt_u32_f32 %2, %4\n\tv_cvt_u32_f32 %3, %4")                             \
	F(45, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\tv_cvt_pk_u8_f32 %1, %4, 1, %1\n\tv_cvt_pk_u8_f32 %2, %4, 1, %2\n\tv_cvt_pk_u8_f32 %3, %4, 1, %3") \
	F(46, "v_sad_u8", "v_sad_u8 %0, %4, %5, %0\n\tv_sad_u8 %1, %4, %5, %1\n\tv_sad_u8 %2, %4, %5, %2\n\tv_sad_u8 %3, %4, %5, %3")                      \
	F(47, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %4, %5, %0\n\tv_dot4_u32_u8 %1, %4, %5, %1\n\tv_dot4_u32_u8 %2, %4, %5, %2\n\tv_dot4_u32_u8 %3, %4, %5, %3") \
	F(48, "v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
	F(49, "v_add_f32 dpp row_shr", "v_add_f32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %2, %4, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf") \
	F(50, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %4, %1\n\tv_bcnt_u32_b32 %2, %4, %2\n\tv_bcnt_u32_b32 %3, %4, %3")          \
	F(51, "v_med3_u32", "v_med3_u32 %0, %4, %5, %0\n\tv_med3_u32 %1, %4, %5, %1\n\tv_med3_u32 %2, %4, %5, %2\n\tv_med3_u32 %3, %4, %5, %3")            \
	F(52, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %5\n\tv_ldexp_f32 %1, %1, %5\n\tv_ldexp_f32 %2, %2, %5\n\tv_ldexp_f32 %3, %3, %5")                        \
	F(53, "v_addc_co_u32", "v_addc_co_u32 %0, vcc, %4, %0, vcc\n\tv_addc_co_u32 %1, vcc, %4, %1, vcc\n\tv_addc_co_u32 %2, vcc, %4, %2, vcc\n\tv_addc_co_u32 %3, vcc, %4, %3, vcc") \
	F(54, "v_fmac_f32", "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5")                            \
	F(55, "v_fma_f32 (sgpr operand)", "v_fma_f32 %0, %4, s20, %0\n\tv_fma_f32 %1, %4, s20, %1\n\tv_fma_f32 %2, %4, s20, %2\n\tv_fma_f32 %3, %4, s20, %3") \
	F(56, "v_fmac_f32 (sgpr operand)", "v_fmac_f32 %0, s20, %4\n\tv_fmac_f32 %1, s20, %4\n\tv_fmac_f32 %2, s20, %4\n\tv_fmac_f32 %3, s20, %4")          \
	F(57, "v_mul_f32 (sgpr operand)", "v_mul_f32 %0, s20, %0\n\tv_mul_f32 %1, s20, %1\n\tv_mul_f32 %2, s20, %2\n\tv_mul_f32 %3, s20, %3")               \
	F(58, "v_cmp_lt_f32 + v_cndmask (pair)", "v_cmp_lt_f32 vcc, %4, %0\n\tv_cndmask_b32 %0, %4, %0, vcc\n\tv_cmp_lt_f32 vcc, %4, %1\n\tv_cndmask_b32 %1, %4, %1, vcc") \
	F(59, "MIX v_fma_f32 + v_cvt_f32_ubyte (2+2)", "v_fma_f32 %0, %4, %5, %0\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_fma_f32 %2, %4, %5, %2\n\tv_cvt_f32_ubyte1 %3, %4") \
	F(60, "MIX v_fma_f32 + v_max3_f32 (2+2)", "v_fma_f32 %0, %4, %5, %0\n\tv_max3_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_max3_f32 %3, %4, %5, %3") \
	F(61, "MIX v_mul_f32 + v_cndmask (2+2)", "v_mul_f32 %0, %4, %0\n\tv_cndmask_b32 %1, %4, %1, vcc\n\tv_mul_f32 %2, %4, %2\n\tv_cndmask_b32 %3, %4, %3, vcc") \
	F(62, "MIX 3 v_fma_f32 + 1 v_cvt", "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_cvt_f32_ubyte1 %3, %4")  \
	F(63, "v_pk_fma_f32-free: v_fma_f32 3 distinct srcs", "v_fma_f32 %0, %4, %5, %1\n\tv_fma_f32 %1, %4, %5, %2\n\tv_fma_f32 %2, %4, %5, %3\n\tv_fma_f32 %3, %4, %5, %0") \
	F(64, "v_fma_f32 dst!=src (2 srcs same)", "v_fma_f32 %0, %4, %4, %5\n\tv_fma_f32 %1, %4, %4, %5\n\tv_fma_f32 %2, %4, %4, %5\n\tv_fma_f32 %3, %4, %4, %5") \
	F(65, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %4\n\tv_cvt_f32_f16 %1, %4\n\tv_cvt_f32_f16 %2, %4\n\tv_cvt_f32_f16 %3, %4")                             \
	F(66, "v_and_b32", "v_and_b32 %0, %4, %0\n\tv_and_b32 %1, %4, %1\n\tv_and_b32 %2, %4, %2\n\tv_and_b32 %3, %4, %3")                                 \
	F(67, "v_add_u32", "v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_add_u32 %3, %4, %3")                                 \
	F(68, "v_perm_b32", "v_perm_b32 %0, %4, %5, %0\n\tv_perm_b32 %1, %4, %5, %1\n\tv_perm_b32 %2, %4, %5, %2\n\tv_perm_b32 %3, %4, %5, %3")            \
	F(69, "v_mov_b32", "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4")                                                 \
	F(70, "v_max_i32", "v_max_i32 %0, %4, %0\n\tv_max_i32 %1, %4, %1\n\tv_max_i32 %2, %4, %2\n\tv_max_i32 %3, %4, %3")                                 \
	F(71, "v_subrev_f32", "v_subrev_f32 %0, %4, %0\n\tv_subrev_f32 %1, %4, %1\n\tv_subrev_f32 %2, %4, %2\n\tv_subrev_f32 %3, %4, %3")                  \
	F(72, "v_mac-like: v_fmac_f32 chain dep", "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %0, %4, %5")

template <int OP> __global__ __launch_bounds__(256, 8) void k(float *out, uint32_t seed)
{
	float q = __uint_as_float(seed + threadIdx.x * 0x00010001u), b = 1.0009765625f;
	float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	for (int i = 0; i < ITER; i++)
	{
#define F(N, NAME, ASM)                                                                                                          \
	if (OP == N)                                                                                                                 \
	{                                                                                                                            \
		R8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q), "v"(b) : "vcc", "s20", "s21", "s22", "s23");)     \
	}
		OPS(F)
#undef F
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main()
{
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, 0) != hipSuccess)
		return 1;
	const int blocks = p.multiProcessorCount * 8;
	float *out;
	if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess)
		return 1;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
	printf("%d CUs at %d MHz (clocks computed at that rate)\n", p.multiProcessorCount, p.clockRate / 1000);
#define F(N, NAME, ASM)                                                                                                          \
	for (int rep = 0; rep < 2; rep++)                                                                                            \
	{                                                                                                                            \
		(void)hipEventRecord(e0, 0);                                                                                             \
		hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, out, 0x3c003800u);                                               \
		(void)hipEventRecord(e1, 0);                                                                                             \
		(void)hipEventSynchronize(e1);                                                                                           \
		float ms = 0;                                                                                                            \
		(void)hipEventElapsedTime(&ms, e0, e1);                                                                                  \
		const double insts = (double)blocks * 4 * ITER * 32;                                                                     \
		if (rep)                                                                                                                 \
			printf("%-52s %.3f ms  %.2f clocks per instruction per SIMD\n", NAME, ms,                                             \
				   (double)p.multiProcessorCount * 4 * p.clockRate * 1e3 / (insts / (ms * 1e-3)));                               \
	}
	OPS(F)
#undef F
	return 0;
}
