// inst_rate.hip — issue cost of single vector instructions on gfx950, in clocks per wave64 instruction per SIMD: every wave
// runs ITER x 32 independent instructions of one kind, 8 waves per SIMD, every CU.  What the traversal's node step could be
// rebuilt from is priced here before it is built (v_fma / v_mul / v_add_f32 issue in ~2 clocks, conversions,
// compares and selects in ~4).
// build: hipcc --offload-arch=gfx950 -O3 inst_rate.hip -o inst_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITER = 2048;
#define R8(X) X X X X X X X X

#define OPS(F)                                                                                                                  \
	F(0, "v_fma_f32", "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3")                 \
	F(1, "v_fma_mix_f32 (f16 lo)", "v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[1,0,0]") \
	F(2, "v_fma_mix_f32 (f16 hi)", "v_fma_mix_f32 %0, %4, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %4, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %4, %5, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %4, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]") \
	F(3, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %4\n\tv_cvt_f32_f16 %1, %4\n\tv_cvt_f32_f16 %2, %4\n\tv_cvt_f32_f16 %3, %4")                              \
	F(4, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_cvt_f32_ubyte1 %2, %4\n\tv_cvt_f32_ubyte1 %3, %4")                \
	F(5, "v_max3_f32", "v_max3_f32 %0, %4, %5, %0\n\tv_max3_f32 %1, %4, %5, %1\n\tv_max3_f32 %2, %4, %5, %2\n\tv_max3_f32 %3, %4, %5, %3")             \
	F(6, "v_max_f32", "v_max_f32 %0, %4, %0\n\tv_max_f32 %1, %4, %1\n\tv_max_f32 %2, %4, %2\n\tv_max_f32 %3, %4, %3")                                 \
	F(7, "v_mad_u32_u24", "v_mad_u32_u24 %0, %4, %5, %0\n\tv_mad_u32_u24 %1, %4, %5, %1\n\tv_mad_u32_u24 %2, %4, %5, %2\n\tv_mad_u32_u24 %3, %4, %5, %3") \
	F(8, "v_add_u32", "v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_add_u32 %3, %4, %3")                                  \
	F(9, "v_lshl_add_u32", "v_lshl_add_u32 %0, %4, 3, %0\n\tv_lshl_add_u32 %1, %4, 3, %1\n\tv_lshl_add_u32 %2, %4, 3, %2\n\tv_lshl_add_u32 %3, %4, 3, %3") \
	F(10, "v_and_b32", "v_and_b32 %0, %4, %0\n\tv_and_b32 %1, %4, %1\n\tv_and_b32 %2, %4, %2\n\tv_and_b32 %3, %4, %3")                                 \
	F(11, "v_perm_b32", "v_perm_b32 %0, %4, %5, %0\n\tv_perm_b32 %1, %4, %5, %1\n\tv_perm_b32 %2, %4, %5, %2\n\tv_perm_b32 %3, %4, %5, %3")           \
	F(12, "v_bfe_u32", "v_bfe_u32 %0, %4, 8, 8\n\tv_bfe_u32 %1, %4, 8, 8\n\tv_bfe_u32 %2, %4, 8, 8\n\tv_bfe_u32 %3, %4, 8, 8")                         \
	F(13, "v_cndmask_b32", "v_cndmask_b32 %0, %4, %0, vcc\n\tv_cndmask_b32 %1, %4, %1, vcc\n\tv_cndmask_b32 %2, %4, %2, vcc\n\tv_cndmask_b32 %3, %4, %3, vcc") \
	F(14, "v_cmp_lt_f32", "v_cmp_lt_f32 vcc, %4, %5\n\tv_cmp_lt_f32 vcc, %4, %5\n\tv_cmp_lt_f32 vcc, %4, %5\n\tv_cmp_lt_f32 vcc, %4, %5")               \
	F(15, "v_mov_b32", "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4")                                                \
	F(16, "v_pk_fma_f16", "v_pk_fma_f16 %0, %4, %5, %0\n\tv_pk_fma_f16 %1, %4, %5, %1\n\tv_pk_fma_f16 %2, %4, %5, %2\n\tv_pk_fma_f16 %3, %4, %5, %3")  \
	F(17, "v_pk_max_f16", "v_pk_max_f16 %0, %4, %0\n\tv_pk_max_f16 %1, %4, %1\n\tv_pk_max_f16 %2, %4, %2\n\tv_pk_max_f16 %3, %4, %3")                  \
	F(18, "v_sub_f32", "v_sub_f32 %0, %4, %0\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %4, %2\n\tv_sub_f32 %3, %4, %3")                                 \
	F(19, "v_mul_f32", "v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3")                                 \
	F(20, "v_readlane_b32", "v_readlane_b32 s20, %4, 3\n\tv_readlane_b32 s21, %4, 5\n\tv_readlane_b32 s22, %4, 7\n\tv_readlane_b32 s23, %4, 9")         \
	F(21, "v_min3_u32", "v_min3_u32 %0, %4, %5, %0\n\tv_min3_u32 %1, %4, %5, %1\n\tv_min3_u32 %2, %4, %5, %2\n\tv_min3_u32 %3, %4, %5, %3")            \
	F(22, "v_med3_f32", "v_med3_f32 %0, %4, %5, %0\n\tv_med3_f32 %1, %4, %5, %1\n\tv_med3_f32 %2, %4, %5, %2\n\tv_med3_f32 %3, %4, %5, %3")            \
	F(23, "v_fmac_f32", "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5")                            \
	F(24, "v_fma_f32 (sgpr operand)", "v_fma_f32 %0, %4, s20, %0\n\tv_fma_f32 %1, %4, s20, %1\n\tv_fma_f32 %2, %4, s20, %2\n\tv_fma_f32 %3, %4, s20, %3") \
	F(29, "v_rcp_f32", "v_rcp_f32 %0, %4\n\tv_rcp_f32 %1, %4\n\tv_rcp_f32 %2, %4\n\tv_rcp_f32 %3, %4") \
	F(30, "v_sqrt_f32", "v_sqrt_f32 %0, %4\n\tv_sqrt_f32 %1, %4\n\tv_sqrt_f32 %2, %4\n\tv_sqrt_f32 %3, %4") \
	F(25, "v_mad_mix-free: v_fma_mix_f32 (all f32)", "v_fma_mix_f32 %0, %4, %5, %0\n\tv_fma_mix_f32 %1, %4, %5, %1\n\tv_fma_mix_f32 %2, %4, %5, %2\n\tv_fma_mix_f32 %3, %4, %5, %3")

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

// packed fp32 (two lanes' worth of operands per 64-bit register pair)
template <int OP> __global__ __launch_bounds__(256, 8) void kp(double *out, double seed)
{
	double q = seed + threadIdx.x, b = 1.0009765625;
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	for (int i = 0; i < ITER; i++)
	{
		if (OP == 0)
		{
			R8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_fma_f32 %2, %4, %5, %2\n\tv_pk_fma_f32 %3, %4, %5, %3"
							: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q), "v"(b));)
		}
		else if (OP == 1)
		{
			R8(asm volatile("v_pk_mul_f32 %0, %4, %0\n\tv_pk_mul_f32 %1, %4, %1\n\tv_pk_mul_f32 %2, %4, %2\n\tv_pk_mul_f32 %3, %4, %3"
							: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q), "v"(b));)
		}
		else
		{
			R8(asm volatile("v_pk_add_f32 %0, %4, %0\n\tv_pk_add_f32 %1, %4, %1\n\tv_pk_add_f32 %2, %4, %2\n\tv_pk_add_f32 %3, %4, %3"
							: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q), "v"(b));)
		}
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
	printf("%d CUs at %d MHz\n", p.multiProcessorCount, p.clockRate / 1000);
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
			printf("%-40s %.3f ms  %.2f clocks per instruction per SIMD\n", NAME, ms,                                             \
				   (double)p.multiProcessorCount * 4 * p.clockRate * 1e3 / (insts / (ms * 1e-3)));                               \
	}
	OPS(F)
#undef F
	{
		double *outd;
		if (hipMalloc(&outd, (size_t)blocks * 256 * 8) != hipSuccess)
			return 1;
		const char *names[3] = {"v_pk_fma_f32 (two fmas)", "v_pk_mul_f32 (two muls)", "v_pk_add_f32 (two adds)"};
		for (int op = 0; op < 3; op++)
			for (int rep = 0; rep < 2; rep++)
			{
				(void)hipEventRecord(e0, 0);
				if (op == 0)
					hipLaunchKernelGGL(kp<0>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				else if (op == 1)
					hipLaunchKernelGGL(kp<1>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				else
					hipLaunchKernelGGL(kp<2>, dim3(blocks), dim3(256), 0, 0, outd, 1.5);
				(void)hipEventRecord(e1, 0);
				(void)hipEventSynchronize(e1);
				float ms = 0;
				(void)hipEventElapsedTime(&ms, e0, e1);
				const double insts = (double)blocks * 4 * ITER * 32;
				if (rep)
					printf("%-40s %.3f ms  %.2f clocks per instruction per SIMD\n", names[op], ms,
						   (double)p.multiProcessorCount * 4 * p.clockRate * 1e3 / (insts / (ms * 1e-3)));
			}
	}
	return 0;
}
