// valu_calib.hip — validation of the VALU-time model of profiles/summarize.py (valu_busy_frac).  Round 3 had derived "the share
// of SIMD cycles a VALU instruction was executing" as SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x CUs x SIMDs) and read 1.10 for
// a traversal kernel — which a fraction cannot be: SQ_ACTIVE_INST_VALU turned out to equal SQ_INSTS_VALU on this part (it counts
// instructions, the "x 4 cycles" was an assumption), and v_fma / v_mul / v_add_f32 occupy a SIMD for 2 cycles, not 4.  The model
// now weights the instruction classes the counters distinguish: 2 cycles for FMA / MUL / ADD_F32, 16 for transcendentals, 4 for
// the rest.  This kernel IS a saturated VALU by construction — 8 waves per SIMD on every CU, each issuing nothing but independent
// v_fma_f32 — so under the same rocprofv3 --pmc pass the model must read ~1.0 for it (tools/evidence.sh records what it reads in
// stage_counters.json: valu_busy_validation).  It also prints the wave-instruction rate it reached against the nominal clock.
// build: hipcc --offload-arch=gfx950 -O3 valu_calib.hip -o valu_calib ; run: ./valu_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int ITER = 40000;
// (inline asm: written in C the compiler pairs independent fmas into v_pk_fma_f32 and the kernel measures something else)
#define FMA8                                                                                                                                   \
	asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"          \
				 "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"               \
				 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                                            \
				 : "v"(m), "v"(c));

__global__ __launch_bounds__(256, 8) void k_calib_fma(float *out, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const float m = 1.0000001f, c = 1e-9f;
#pragma nounroll
	for (int i = 0; i < ITER; i++)
	{
		FMA8 FMA8 FMA8 FMA8
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, 0) != hipSuccess)
		return 1;
	const int cus = p.multiProcessorCount, blocks = cus * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
	float *out;
	if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess)
		return 1;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	for (int rep = 0; rep < 3; rep++)
	{
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_calib_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0f + rep);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)blocks * 4 * ITER * 32; // wave-instructions
		printf("k_calib_fma: %d CUs, %.3f ms, %.1f G wave-instructions/s = one per %.2f clocks per SIMD at %.2f GHz\n", cus, ms,
			   insts / (ms * 1e-3) / 1e9, (double)cus * 4 * p.clockRate * 1e3 / (insts / (ms * 1e-3)), p.clockRate * 1e-6);
	}
	return 0;
}
