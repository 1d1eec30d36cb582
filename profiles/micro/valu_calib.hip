// valu_calib.hip — calibration of the "share of SIMD cycles a VALU instruction was executing" figure (profiles/summarize.py:
// valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x CUs x SIMDs)).  Derived from two counters of different blocks
// and clocks, the raw ratio came out above 1 for the traversal kernels (1.10 in round 3), which a fraction cannot be.  This
// kernel IS a saturated VALU by construction — 8 waves per SIMD on every CU, each issuing nothing but independent v_fma_f32 —
// so its raw ratio under the same rocprofv3 --pmc pass is what "1.0" reads as on this part; tools/evidence.sh divides every
// kernel's raw ratio by it (stage_counters.json records the factor).  It also prints the wave-instruction rate it reached
// against the clock, i.e. how many cycles a wave64 v_fma_f32 occupies a SIMD (the guide's 2 vs the counters' 4).
// build: hipcc --offload-arch=gfx950 -O3 valu_calib.hip -o valu_calib ; run: ./valu_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int ITER = 40000;
#define FMA8                                                                                                         \
	a0 = __builtin_fmaf(a0, m, c), a1 = __builtin_fmaf(a1, m, c), a2 = __builtin_fmaf(a2, m, c), a3 = __builtin_fmaf(a3, m, c), \
	a4 = __builtin_fmaf(a4, m, c), a5 = __builtin_fmaf(a5, m, c), a6 = __builtin_fmaf(a6, m, c), a7 = __builtin_fmaf(a7, m, c);

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
