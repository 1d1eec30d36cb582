// inst_rate4.hip — round 6: what do gfx950's PACKED 8-bit float conversions cost?  A node step of the per-lane
// traversal spends 24 v_cvt_f32_ubyteN (4.3 clocks each) on its child planes; v_cvt_pk_f32_fp8 / _bf8 and the scalef32 forms produce
// TWO floats per instruction.  If one of them issues like a single 4-clock instruction, a node with outward-rounded 8-bit FLOAT planes
// would halve the conversions (12 instead of 24).  Rows: clocks per wave64 instruction per SIMD (8 waves per SIMD, every CU), alone and
// in the mixes the node step would contain.  Also: a 16-bit plane whose high half IS the float operand (v_and_b32 with a literal
// mask, a 2-clock instruction; v_lshlrev_b32 16 for the low half is of the 4-clock class by round 5's table).
// build: hipcc --offload-arch=gfx950 -O3 inst_rate4.hip -o inst_rate4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITER = 1024;
#define R8(X) X X X X X X X X

// %0..%3: 64-bit accumulators (VGPR pairs), %4: 32-bit source, %5: 32-bit scale / second operand, %6 / %7: 32-bit accumulators
#define OPS(F) \
	F(0, 4, "v_cvt_f32_ubyte1 (baseline)", "v_cvt_f32_ubyte1 %6, %4\n\tv_cvt_f32_ubyte1 %7, %4\n\tv_cvt_f32_ubyte1 %6, %4\n\tv_cvt_f32_ubyte1 %7, %4") \
	F(1, 4, "v_cvt_pk_f32_fp8", "v_cvt_pk_f32_fp8 %0, %4\n\tv_cvt_pk_f32_fp8 %1, %4\n\tv_cvt_pk_f32_fp8 %2, %4\n\tv_cvt_pk_f32_fp8 %3, %4") \
	F(2, 4, "v_cvt_pk_f32_bf8", "v_cvt_pk_f32_bf8 %0, %4\n\tv_cvt_pk_f32_bf8 %1, %4\n\tv_cvt_pk_f32_bf8 %2, %4\n\tv_cvt_pk_f32_bf8 %3, %4") \
	F(3, 4, "v_cvt_pk_f32_fp8 sdwa WORD_1", "v_cvt_pk_f32_fp8_sdwa %0, %4 src0_sel:WORD_1\n\tv_cvt_pk_f32_fp8_sdwa %1, %4 src0_sel:WORD_1\n\tv_cvt_pk_f32_fp8_sdwa %2, %4 src0_sel:WORD_1\n\tv_cvt_pk_f32_fp8_sdwa %3, %4 src0_sel:WORD_1") \
	F(4, 4, "v_cvt_scalef32_pk_f32_fp8", "v_cvt_scalef32_pk_f32_fp8 %0, %4, %5\n\tv_cvt_scalef32_pk_f32_fp8 %1, %4, %5\n\tv_cvt_scalef32_pk_f32_fp8 %2, %4, %5\n\tv_cvt_scalef32_pk_f32_fp8 %3, %4, %5") \
	F(5, 4, "v_cvt_scalef32_pk_f32_fp8 op_sel hi", "v_cvt_scalef32_pk_f32_fp8 %0, %4, %5 op_sel:[1,0,0]\n\tv_cvt_scalef32_pk_f32_fp8 %1, %4, %5 op_sel:[1,0,0]\n\tv_cvt_scalef32_pk_f32_fp8 %2, %4, %5 op_sel:[1,0,0]\n\tv_cvt_scalef32_pk_f32_fp8 %3, %4, %5 op_sel:[1,0,0]") \
	F(6, 4, "v_cvt_scalef32_pk_f32_bf8", "v_cvt_scalef32_pk_f32_bf8 %0, %4, %5\n\tv_cvt_scalef32_pk_f32_bf8 %1, %4, %5\n\tv_cvt_scalef32_pk_f32_bf8 %2, %4, %5\n\tv_cvt_scalef32_pk_f32_bf8 %3, %4, %5") \
	F(7, 4, "v_cvt_f32_fp8", "v_cvt_f32_fp8 %6, %4\n\tv_cvt_f32_fp8 %7, %4\n\tv_cvt_f32_fp8 %6, %4\n\tv_cvt_f32_fp8 %7, %4") \
	F(8, 4, "v_cvt_scalef32_f32_fp8", "v_cvt_scalef32_f32_fp8 %6, %4, %5\n\tv_cvt_scalef32_f32_fp8 %7, %4, %5\n\tv_cvt_scalef32_f32_fp8 %6, %4, %5\n\tv_cvt_scalef32_f32_fp8 %7, %4, %5") \
	F(9, 4, "cvt_pk_fp8 + 2 fma (two planes decoded and used)", "v_cvt_pk_f32_fp8 %0, %4\n\tv_fma_f32 %6, %4, %5, %6\n\tv_fma_f32 %7, %4, %5, %7\n\tv_cvt_pk_f32_fp8 %1, %4") \
	F(10, 4, "2 cvt_ubyte + 2 fma (today's two planes)", "v_cvt_f32_ubyte1 %6, %4\n\tv_fma_f32 %6, %4, %5, %6\n\tv_cvt_f32_ubyte2 %7, %4\n\tv_fma_f32 %7, %4, %5, %7") \
	F(11, 4, "cvt_pk_fp8 + pk_fma_f32", "v_cvt_pk_f32_fp8 %0, %4\n\tv_pk_fma_f32 %1, %2, %3, %1\n\tv_cvt_pk_f32_fp8 %0, %4\n\tv_pk_fma_f32 %1, %2, %3, %1") \
	F(12, 4, "scalef32_pk_fp8 + 2 fma", "v_cvt_scalef32_pk_f32_fp8 %0, %4, %5\n\tv_fma_f32 %6, %4, %5, %6\n\tv_fma_f32 %7, %4, %5, %7\n\tv_cvt_scalef32_pk_f32_fp8 %1, %4, %5") \
	F(13, 4, "v_and_b32 literal mask (high half of a 16-bit plane pair as float)", "v_and_b32 %6, 0xffff0000, %4\n\tv_and_b32 %7, 0xffff0000, %4\n\tv_and_b32 %6, 0xffff0000, %4\n\tv_and_b32 %7, 0xffff0000, %4") \
	F(14, 4, "v_lshlrev_b32 16 (low half as float)", "v_lshlrev_b32 %6, 16, %4\n\tv_lshlrev_b32 %7, 16, %4\n\tv_lshlrev_b32 %6, 16, %4\n\tv_lshlrev_b32 %7, 16, %4") \
	F(15, 4, "and-mask + fma, lshl + fma (a 16-bit pair decoded and used)", "v_and_b32 %6, 0xffff0000, %4\n\tv_fma_f32 %6, %4, %5, %6\n\tv_lshlrev_b32 %7, 16, %4\n\tv_fma_f32 %7, %4, %5, %7") \
	F(16, 4, "v_fma_mix_f32 (f16 plane as fma operand)", "v_fma_mix_f32 %6, %4, %5, %6 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %4, %5, %7 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %6, %4, %5, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %4, %5, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]")

template <int OP> __global__ __launch_bounds__(256, 8) void k(double *out, uint32_t seed)
{
	float q = __uint_as_float(0x38383838u + (seed & 1u) + threadIdx.x * 0x00010001u), b = 1.0009765625f;
	double a0 = 0, a1 = 0, a2 = 1.0, a3 = 1.0;
	float f0 = 0, f1 = 0;
	for (int i = 0; i < ITER; i++)
	{
#define F(N, CNT, NAME, ASM)                                                                                                      \
	if (OP == N)                                                                                                                  \
	{                                                                                                                             \
		R8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1) : "v"(q), "v"(b), "v"(f0), "v"(f1));)      \
	}
		OPS(F)
#undef F
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1;
}

template <int OP> static void run(const char *name, int per_group, double *out, int cus, double ghz)
{
	const int blocks = cus * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < 5; r++)
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, (uint32_t)r);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	// wave-instructions per SIMD: 8 waves x ITER x 8 groups x per_group instructions
	const double insts = 8.0 * ITER * 8.0 * per_group * 5.0;
	const double clocks = ms * 1e-3 * ghz * 1e9;
	printf("%-72s %6.2f clocks per wave64 instruction per SIMD\n", name, clocks / insts);
}

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double ghz = prop.clockRate * 1e-6;
	printf("%s: %d CUs, %.2f GHz nominal\n", prop.name, cus, ghz);
	double *out;
	hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(double));
#define F(N, CNT, NAME, ASM) run<N>(NAME, CNT, out, cus, ghz);
	OPS(F)
#undef F
	return 0;
}
