// inst_rate5.hip — round 6, late: what does a SCALAR instruction cost beside the vector ones?  The packet kernels (k_primary_packet,
// k_shadow_packet) issue 0.7-0.8 SALU instructions per VALU instruction (profiles/r06g_pmc_mix.md: 1.80 G against 2.33 G per primary
// launch): child ordering by the reference lane's entry distances, exec-mask bookkeeping, stack pointer, address arithmetic.
// Rows: clocks per wave-instruction per SIMD at 8 waves per SIMD on every CU — scalar instructions alone, vector ones alone, and
// the two interleaved in one wave's stream (do another wave's scalar instructions hide behind this wave's vector ones?).
// build: hipcc --offload-arch=gfx950 -O3 inst_rate5.hip -o inst_rate5
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITER = 1024;
#define R8(X) X X X X X X X X

// %0..%3: SGPR accumulators, %4..%7: VGPR accumulators, %8: SGPR operand, %9: VGPR operand
#define OPS(F) \
	F(0, 4, 0, "4 s_add_u32", "s_add_u32 %0, %0, %8\n\ts_add_u32 %1, %1, %8\n\ts_add_u32 %2, %2, %8\n\ts_add_u32 %3, %3, %8") \
	F(1, 4, 0, "4 s_min_u32 / s_max_u32", "s_min_u32 %0, %0, %8\n\ts_max_u32 %1, %1, %8\n\ts_min_u32 %2, %2, %8\n\ts_max_u32 %3, %3, %8") \
	F(2, 4, 0, "comparator: s_cmp_lt_u32 + 2 s_cselect_b32 + s_min_u32", "s_cmp_lt_u32 %0, %1\n\ts_cselect_b32 %2, %3, %8\n\ts_cselect_b32 %3, %8, %2\n\ts_min_u32 %0, %0, %1") \
	F(3, 4, 0, "4 s_and_b64-class (s_and_b32 here)", "s_and_b32 %0, %0, %8\n\ts_or_b32 %1, %1, %8\n\ts_and_b32 %2, %2, %8\n\ts_or_b32 %3, %3, %8") \
	F(4, 0, 4, "4 v_fma_f32", "v_fma_f32 %4, %9, %9, %4\n\tv_fma_f32 %5, %9, %9, %5\n\tv_fma_f32 %6, %9, %9, %6\n\tv_fma_f32 %7, %9, %9, %7") \
	F(5, 0, 4, "4 v_max3_f32", "v_max3_f32 %4, %9, %4, %5\n\tv_max3_f32 %5, %9, %5, %6\n\tv_max3_f32 %6, %9, %6, %7\n\tv_max3_f32 %7, %9, %7, %4") \
	F(6, 2, 2, "2 v_fma_f32 + 2 s_add_u32 interleaved", "v_fma_f32 %4, %9, %9, %4\n\ts_add_u32 %0, %0, %8\n\tv_fma_f32 %5, %9, %9, %5\n\ts_add_u32 %1, %1, %8") \
	F(7, 4, 4, "4 v_fma_f32 + 4 s_add_u32 interleaved", "v_fma_f32 %4, %9, %9, %4\n\ts_add_u32 %0, %0, %8\n\tv_fma_f32 %5, %9, %9, %5\n\ts_add_u32 %1, %1, %8\n\tv_fma_f32 %6, %9, %9, %6\n\ts_add_u32 %2, %2, %8\n\tv_fma_f32 %7, %9, %9, %7\n\ts_add_u32 %3, %3, %8") \
	F(8, 4, 4, "4 v_fma_f32 then 4 s_add_u32 (blocks)", "v_fma_f32 %4, %9, %9, %4\n\tv_fma_f32 %5, %9, %9, %5\n\tv_fma_f32 %6, %9, %9, %6\n\tv_fma_f32 %7, %9, %9, %7\n\ts_add_u32 %0, %0, %8\n\ts_add_u32 %1, %1, %8\n\ts_add_u32 %2, %2, %8\n\ts_add_u32 %3, %3, %8") \
	F(9, 4, 4, "4 v_max3_f32 + 4 s_add_u32 interleaved", "v_max3_f32 %4, %9, %4, %5\n\ts_add_u32 %0, %0, %8\n\tv_max3_f32 %5, %9, %5, %6\n\ts_add_u32 %1, %1, %8\n\tv_max3_f32 %6, %9, %6, %7\n\ts_add_u32 %2, %2, %8\n\tv_max3_f32 %7, %9, %7, %4\n\ts_add_u32 %3, %3, %8") \
	F(10, 6, 6, "a node step's shape: 6 x (v_fma, s_cmp / s_cselect ...)", "v_fma_f32 %4, %9, %9, %4\n\ts_cmp_lt_u32 %0, %1\n\tv_fma_f32 %5, %9, %9, %5\n\ts_cselect_b32 %2, %3, %8\n\tv_fma_f32 %6, %9, %9, %6\n\ts_cselect_b32 %3, %8, %2\n\tv_max3_f32 %7, %9, %7, %4\n\ts_min_u32 %0, %0, %1\n\tv_fma_f32 %4, %9, %9, %4\n\ts_max_u32 %1, %0, %1\n\tv_fma_f32 %5, %9, %9, %5\n\ts_add_u32 %2, %2, %8") \
	F(11, 0, 4, "4 v_readlane_b32 (VALU -> SGPR)", "v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 5\n\tv_readlane_b32 %2, %6, 7\n\tv_readlane_b32 %3, %7, 9") \
	F(12, 0, 4, "4 v_writelane_b32", "v_writelane_b32 %4, %8, 3\n\tv_writelane_b32 %5, %8, 5\n\tv_writelane_b32 %6, %8, 7\n\tv_writelane_b32 %7, %8, 9") \
	F(13, 2, 2, "v_readlane + s_add on it + v_fma x2 (dependent hop VALU -> SALU)", "v_readlane_b32 %0, %4, 3\n\ts_add_u32 %1, %0, %8\n\tv_fma_f32 %5, %9, %9, %5\n\ts_add_u32 %2, %1, %8")

template <int OP> __global__ __launch_bounds__(256, 8) void k(float *out, uint32_t seed)
{
	uint32_t s0 = seed, s1 = seed + 1u, s2 = seed + 2u, s3 = seed + 3u;
	const uint32_t sk = seed | 1u;
	float v0 = 0.0f, v1 = 0.0f, v2 = 1.0f, v3 = 1.0f;
	const float q = 1.0009765625f + threadIdx.x * 1e-6f;
	for (int i = 0; i < ITER; i++)
	{
#define F(N, NS, NV, NAME, ASM)                                                                                                   \
	if (OP == N)                                                                                                                  \
	{                                                                                                                             \
		R8(asm volatile(ASM : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(sk), "v"(q) : "scc");) \
	}
		OPS(F)
#undef F
	}
	out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + (float)(s0 + s1 + s2 + s3);
}

template <int OP> static void run(const char *name, int ns, int nv, float *out, int cus, double ghz)
{
	const int blocks = cus * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < 5; r++)
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, (uint32_t)r + 2u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double groups = 8.0 * ITER * 8.0 * 5.0; // per SIMD: 8 waves x ITER x 8 groups x 5 launches
	const double clocks = ms * 1e-3 * ghz * 1e9;
	printf("%-66s %2d scalar + %2d vector: %6.2f clocks per group per SIMD = %5.2f per instruction", name, ns, nv, clocks / groups,
		   clocks / groups / (ns + nv));
	if (ns && nv)
		printf("  (%5.2f per scalar if the vector ones were free, %5.2f per vector if the scalar ones were)", clocks / groups / ns, clocks / groups / nv);
	printf("\n");
}

int main()
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double ghz = prop.clockRate * 1e-6;
	printf("%s: %d CUs, %.2f GHz nominal\n", prop.name, cus, ghz);
	float *out;
	hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
#define F(N, NS, NV, NAME, ASM) run<N>(NAME, NS, NV, out, cus, ghz);
	OPS(F)
#undef F
	hipFree(out);
	return 0;
}
