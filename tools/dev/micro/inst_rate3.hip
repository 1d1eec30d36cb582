// inst_rate3.hip — round 5's extension of inst_rate.hip: the issue cost (clocks per wave64 instruction per SIMD, 8 waves per
// SIMD, every CU) of the instructions a cheaper per-lane node step could be built from — integer VOP2, three-operand integer
// forms, compares, packed 16-bit arithmetic, output / input modifiers (clamp, neg, abs), DPP — and of MIXES of a 2-clock and a
// 4-clock instruction (is the class-weighted VALU-time model additive?).  VERDICT r04 item 1(a).
// build: hipcc --offload-arch=gfx950 -O3 inst_rate3.hip -o inst_rate2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITER = 1024;
#define R8(X) X X X X X X X X
// I4(op-with-%N placeholders): the same instruction on the four accumulators
#define ONE(S, D) S(D)

#define OPS(F) \
	F(0, "sel (sgpr pair mask)", "v_cndmask_b32_e64 %0, %4, %0, s[20:21]\n\tv_cndmask_b32_e64 %1, %4, %1, s[20:21]\n\tv_cndmask_b32_e64 %2, %4, %2, s[20:21]\n\tv_cndmask_b32_e64 %3, %4, %3, s[20:21]") \
	F(1, "selvcc", "v_cndmask_b32_e32 %0, %4, %0, vcc\n\tv_cndmask_b32_e32 %1, %4, %1, vcc\n\tv_cndmask_b32_e32 %2, %4, %2, vcc\n\tv_cndmask_b32_e32 %3, %4, %3, vcc") \
	F(2, "cmp (e64 sgpr dst)", "v_cmp_lt_f32_e64 s[22:23], %4, %0\n\tv_cmp_lt_f32_e64 s[22:23], %4, %1\n\tv_cmp_lt_f32_e64 s[22:23], %4, %2\n\tv_cmp_lt_f32_e64 s[22:23], %4, %3") \
	F(3, "cvt+max3", "v_cvt_f32_ubyte1 %0, %4\n\tv_max3_f32 %1, %4, %5, %1\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_max3_f32 %3, %4, %5, %3") \
	F(4, "cvt+sel", "v_cvt_f32_ubyte1 %0, %4\n\tv_cndmask_b32_e64 %1, %4, %1, s[20:21]\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_cndmask_b32_e64 %3, %4, %3, s[20:21]") \
	F(5, "cvt+cmp", "v_cvt_f32_ubyte1 %0, %4\n\tv_cmp_lt_f32_e64 s[22:23], %4, %1\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_cmp_lt_f32_e64 s[22:23], %4, %3") \
	F(6, "max3+sel", "v_max3_f32 %0, %4, %5, %0\n\tv_cndmask_b32_e64 %1, %4, %1, s[20:21]\n\tv_max3_f32 %2, %4, %5, %2\n\tv_cndmask_b32_e64 %3, %4, %3, s[20:21]") \
	F(7, "cvt+lshl", "v_cvt_f32_ubyte1 %0, %4\n\tv_lshlrev_b32 %1, 3, %1\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_lshlrev_b32 %3, 3, %3") \
	F(8, "cvt+min", "v_cvt_f32_ubyte1 %0, %4\n\tv_min_f32 %1, %4, %1\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_min_f32 %3, %4, %3") \
	F(9, "cvt+perm", "v_cvt_f32_ubyte1 %0, %4\n\tv_perm_b32 %1, %4, %5, %1\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_perm_b32 %3, %4, %5, %3") \
	F(10, "cvt,max3,sel,cmp", "v_cvt_f32_ubyte1 %0, %4\n\tv_max3_f32 %1, %4, %5, %1\n\tv_cndmask_b32_e64 %2, %4, %2, s[20:21]\n\tv_cmp_lt_f32_e64 s[22:23], %4, %3") \
	F(11, "2 slow(cvt) + 2 fast(fma) grouped", "v_cvt_f32_ubyte1 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3") \
	F(12, "3 cvt + 1 fma", "v_cvt_f32_ubyte1 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_fma_f32 %3, %4, %5, %3") \
	F(13, "cvt,max3,cmp + 1 fma", "v_cvt_f32_ubyte1 %0, %4\n\tv_max3_f32 %1, %4, %5, %1\n\tv_cmp_lt_f32_e64 s[22:23], %4, %2\n\tv_fma_f32 %3, %4, %5, %3") \
	F(14, "cvt,max3 + fma,and", "v_cvt_f32_ubyte1 %0, %4\n\tv_fma_f32 %1, %4, %5, %1\n\tv_max3_f32 %2, %4, %5, %2\n\tv_and_b32 %3, %4, %3") \
	F(15, "fma+and", "v_fma_f32 %0, %4, %5, %0\n\tv_and_b32 %1, %4, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_and_b32 %3, %4, %3") \
	F(16, "fma+mul", "v_fma_f32 %0, %4, %5, %0\n\tv_mul_f32 %1, %4, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_mul_f32 %3, %4, %3") \
	F(17, "fma(sgpr)+fma", "v_fma_f32 %0, %4, s20, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, s20, %2\n\tv_fma_f32 %3, %4, %5, %3") \
	F(18, "fma(sgpr)+cvt", "v_fma_f32 %0, %4, s20, %0\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_fma_f32 %2, %4, s20, %2\n\tv_cvt_f32_ubyte1 %3, %4") \
	F(19, "and literal", "v_and_b32 %0, 0x3fffffff, %0\n\tv_and_b32 %1, 0x3fffffff, %1\n\tv_and_b32 %2, 0x3fffffff, %2\n\tv_and_b32 %3, 0x3fffffff, %3") \
	F(20, "add literal", "v_add_u32 %0, 0x12345, %0\n\tv_add_u32 %1, 0x12345, %1\n\tv_add_u32 %2, 0x12345, %2\n\tv_add_u32 %3, 0x12345, %3") \
	F(21, "mul literal", "v_mul_f32 %0, 0x3f99999a, %0\n\tv_mul_f32 %1, 0x3f99999a, %1\n\tv_mul_f32 %2, 0x3f99999a, %2\n\tv_mul_f32 %3, 0x3f99999a, %3") \
	F(22, "add inline -1", "v_add_u32 %0, -1, %0\n\tv_add_u32 %1, -1, %1\n\tv_add_u32 %2, -1, %2\n\tv_add_u32 %3, -1, %3") \
	F(23, "rcp+fma", "v_rcp_f32 %0, %4\n\tv_fma_f32 %1, %4, %5, %1\n\tv_rcp_f32 %2, %4\n\tv_fma_f32 %3, %4, %5, %3") \
	F(24, "rcp+cvt", "v_rcp_f32 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_rcp_f32 %2, %4\n\tv_cvt_f32_ubyte1 %3, %4") \
	F(25, "rcp + 3 fma", "v_rcp_f32 %0, %4\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3") \
	F(26, "bfe+cvt", "v_bfe_u32 %0, %4, 8, 8\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_bfe_u32 %2, %4, 8, 8\n\tv_cvt_f32_ubyte1 %3, %4") \
	F(27, "mbcnt", "v_mbcnt_lo_u32_b32 %0, -1, %0\n\tv_mbcnt_lo_u32_b32 %1, -1, %1\n\tv_mbcnt_lo_u32_b32 %2, -1, %2\n\tv_mbcnt_lo_u32_b32 %3, -1, %3") \
	F(28, "readlane+fma", "v_readlane_b32 s20, %4, 3\n\tv_fma_f32 %1, %4, %5, %1\n\tv_readlane_b32 s20, %4, 3\n\tv_fma_f32 %3, %4, %5, %3") \
	F(29, "subu,ashr,xor,and (fast ints)", "v_sub_u32 %0, %4, %0\n\tv_ashrrev_i32 %1, 31, %1\n\tv_xor_b32 %2, %4, %2\n\tv_and_b32 %3, %4, %3") \
	F(30, "pk_fma_f16+cvt", "v_pk_fma_f16 %0, %4, %5, %0\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_pk_fma_f16 %2, %4, %5, %2\n\tv_cvt_f32_ubyte1 %3, %4") \
	F(31, "pk_fma_f16+fma", "v_pk_fma_f16 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f16 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3")

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


// packed fp32 mixes: is v_pk_fma_f32 an instruction of the 4-clock kind (it would then compete with the conversions) or two
// 2-clock ones?
template <int OP> __global__ __launch_bounds__(256, 8) void kpm(double *out, double seed)
{
	double q = seed + threadIdx.x, b = 1.0009765625;
	double a0 = 0, a1 = 0;
	float f0 = 0, f1 = 0, g = (float)threadIdx.x;
	for (int i = 0; i < ITER; i++)
	{
		if (OP == 0)
		{
			R8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_cvt_f32_ubyte1 %2, %6\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_cvt_f32_ubyte1 %3, %6"
							: "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1) : "v"(q), "v"(b), "v"(g));)
		}
		else if (OP == 1)
		{
			R8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %2, %6, %6, %2\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %3, %6, %6, %3"
							: "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1) : "v"(q), "v"(b), "v"(g));)
		}
		else if (OP == 2)
		{
			R8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n\tv_cvt_f32_ubyte1 %2, %6\n\tv_cvt_f32_ubyte2 %3, %6\n\tv_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[1,0,0]"
							: "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1) : "v"(q), "v"(b), "v"(g));)
		}
		else
		{
			R8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_max3_f32 %2, %6, %6, %2\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_cndmask_b32_e64 %3, %6, %3, s[20:21]"
							: "+v"(a0), "+v"(a1), "+v"(f0), "+v"(f1) : "v"(q), "v"(b), "v"(g) : "s20", "s21");)
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + f0 + f1;
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
	{
		double *outd;
		if (hipMalloc(&outd, (size_t)blocks * 256 * 8) != hipSuccess)
			return 1;
		const char *names[4] = {"pk_fma_f32 + cvt (1:1)", "pk_fma_f32 + fma (1:1)", "pk_fma_f32 op_sel + 2 cvt (the slab: 2 planes)", "pk_fma_f32 + max3 / cndmask"};
		for (int op = 0; op < 4; op++)
			for (int rep = 0; rep < 2; rep++)
			{
				(void)hipEventRecord(e0, 0);
				if (op == 0)
					hipLaunchKernelGGL(kpm<0>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				else if (op == 1)
					hipLaunchKernelGGL(kpm<1>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				else if (op == 2)
					hipLaunchKernelGGL(kpm<2>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				else
					hipLaunchKernelGGL(kpm<3>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				(void)hipEventRecord(e1, 0);
				(void)hipEventSynchronize(e1);
				float ms = 0;
				(void)hipEventElapsedTime(&ms, e0, e1);
				const double insts = (double)blocks * 4 * ITER * 32;
				if (rep)
					printf("%-52s %.3f ms  %.2f clocks per instruction per SIMD\n", names[op], ms,
						   (double)p.multiProcessorCount * 4 * p.clockRate * 1e3 / (insts / (ms * 1e-3)));
			}
	}
	return 0;
}
