// valu_micro.hip — what does a wave64 VALU instruction cost on gfx950 when only some lanes are active?
// The traversal kernels run at 15-30 of 64 lanes active per VALU instruction (profiles/r02*_pmc_sq_lanes.md); whether a
// half-empty wave issues faster decides whether packing the live lanes into one half of the wave is worth anything.
// Every wave runs ITER x 32 independent v_fma_f32 (or another op) under an EXEC mask chosen by `mode`:
//   0: all 64 lanes   1: lanes 0..31   2: lanes 0..15   3: every 4th lane (16 scattered)   4: lanes 32..63   5: lane 0 only
//   6: lanes 0..7   7: lanes 0..3   8: lanes 0..1   9: lanes 0, 16, 32, 48
// Reported: wave-instructions per shader clock per SIMD (s_memtime around the loop, 8 waves per SIMD resident).
// build: hipcc --offload-arch=gfx950 -O3 valu_micro.hip -o valu_micro ; run: ./valu_micro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                                                       \
	do                                                                                                                 \
	{                                                                                                                  \
		hipError_t e = (x);                                                                                            \
		if (e != hipSuccess)                                                                                           \
		{                                                                                                              \
			printf("%s: %s\n", #x, hipGetErrorString(e));                                                              \
			exit(1);                                                                                                   \
		}                                                                                                              \
	} while (0)

constexpr int ITER = 2048;

__device__ __forceinline__ bool lane_on(int mode, uint32_t lane)
{
	switch (mode)
	{
	case 0: return true;
	case 1: return lane < 32;
	case 2: return lane < 16;
	case 3: return (lane & 3) == 0;
	case 4: return lane >= 32;
	case 5: return lane == 0;
	case 6: return lane < 8;
	case 7: return lane < 4;
	case 8: return lane < 2;
	default: return (lane & 15) == 0;
	}
}

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP> __global__ __launch_bounds__(256) void k_valu(float *out, uint64_t *cyc, int mode, float seed)
{
	const uint32_t lane = threadIdx.x & 63;
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const float m = 1.0000001f, c = 1e-9f;
	typedef float v2f __attribute__((ext_vector_type(2)));
	v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
	const v2f pm = {m, m}, pc = {c, c};
	uint64_t t0 = 0, t1 = 0;
	if (lane_on(mode, lane))
	{
		t0 = __builtin_readcyclecounter();
		for (int i = 0; i < ITER; i++)
		{
			if (OP == 0)
			{
				asm volatile(REP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c));
			}
			else if (OP == 1)
			{
				asm volatile(REP8("v_max3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c));
			}
			else if (OP == 2)
			{
				asm volatile(REP8("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %5\n v_cvt_f32_ubyte2 %2, %6\n v_cvt_f32_ubyte3 %3, %7\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
			}
			else if (OP == 3)
			{
				asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c)
							 : "vcc");
			}
			else if (OP == 6)
			{
				// v_cndmask with an explicit SGPR-pair mask (VOP3 form, what the traversal loop's selects compile to)
				asm volatile("s_mov_b32 s10, 0x55555555\n s_mov_b32 s11, 0x55555555\n" REP8("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %9, s[10:11]\n v_cndmask_b32_e64 %3, %3, %9, s[10:11]\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c)
							 : "s10", "s11");
			}
			else if (OP == 7)
			{
				// compares writing VCC / an SGPR pair
				asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %8\n v_cmp_lt_f32_e64 s[10:11], %1, %8\n v_cmp_lt_f32_e32 vcc, %2, %9\n v_cmp_lt_f32_e64 s[10:11], %3, %9\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c)
							 : "vcc", "s10", "s11");
			}
			else if (OP == 8)
			{
				// a compare feeding a select, the pattern of a sorting-network comparator
				asm volatile(REP8("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %0, %1, vcc\n v_cmp_lt_f32_e64 s[10:11], %1, %3\n v_cndmask_b32_e64 %3, %1, %0, s[10:11]\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c)
							 : "vcc", "s10", "s11");
			}
			else if (OP == 9)
			{
				asm volatile(REP8("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %9\n v_max_f32 %3, %3, %9\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c));
			}
			else if (OP == 10)
			{
				// mixed-precision fma: f16 half of src0 (op_sel picks lo / hi) x f32 + f32
				asm volatile(REP8("v_fma_mix_f32 %0, %4, %8, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %5, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %6, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %7, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c));
			}
			else if (OP == 11)
			{
				asm volatile(REP8("v_perm_b32 %0, %4, %5, %8\n v_perm_b32 %1, %5, %6, %8\n v_perm_b32 %2, %6, %7, %9\n v_perm_b32 %3, %7, %4, %9\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
							 : "v"(m), "v"(c));
			}
			else if (OP == 5)
			{
				// 32 packed instructions = 64 fmas per lane
				asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
							 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
							 : "v"(pm), "v"(pc));
			}
			else
			{
				asm volatile(REP8("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n")
							 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
			}
		}
		t1 = __builtin_readcyclecounter();
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
	if (lane == (mode == 4 ? 32 : 0)) cyc[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP> static void run(const char *name, float *out, uint64_t *cyc, uint64_t *h, int blocks)
{
	for (int mode = 0; mode < (OP >= 6 ? 2 : 10); mode++)
	{
		hipEvent_t a, b;
		CHECK(hipEventCreate(&a));
		CHECK(hipEventCreate(&b));
		k_valu<OP><<<blocks, 256>>>(out, cyc, mode, 1.0f);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(a));
		k_valu<OP><<<blocks, 256>>>(out, cyc, mode, 1.0f);
		CHECK(hipEventRecord(b));
		CHECK(hipEventSynchronize(b));
		float ms;
		CHECK(hipEventElapsedTime(&ms, a, b));
		const int waves = blocks * 4;
		CHECK(hipMemcpy(h, cyc, waves * 8, hipMemcpyDeviceToHost));
		double mean = 0;
		for (int i = 0; i < waves; i++) mean += double(h[i]);
		mean /= waves;
		const double insts = double(ITER) * 32;
		// 8 waves share a SIMD: a wave's loop takes `mean` clocks while its SIMD issues 8 x insts wave-instructions
		printf("%-10s mode %d  %8.3f ms  %9.0f clk per wave  %.3f wave-inst/clk/SIMD (in-kernel clock)  %.1f G wave-inst/s chip\n", name, mode, ms, mean,
			   8.0 * insts / mean, double(waves) * insts / ms * 1e-6);
	}
}

int main()
{
	const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD, one round
	float *out;
	uint64_t *cyc;
	CHECK(hipMalloc(&out, size_t(blocks) * 256 * 4));
	CHECK(hipMalloc(&cyc, size_t(blocks) * 4 * 8));
	uint64_t *h = (uint64_t *)malloc(size_t(blocks) * 4 * 8);
	run<0>("v_fma", out, cyc, h, blocks);
	run<1>("v_minmax3", out, cyc, h, blocks);
	run<2>("v_cvt_ub", out, cyc, h, blocks);
	run<3>("v_cndmask", out, cyc, h, blocks);
	run<4>("v_mov", out, cyc, h, blocks);
	run<5>("v_pk_fma", out, cyc, h, blocks);
	run<6>("cndmask_s", out, cyc, h, blocks);
	run<7>("v_cmp", out, cyc, h, blocks);
	run<8>("cmp+cndmsk", out, cyc, h, blocks);
	run<9>("v_min/max", out, cyc, h, blocks);
	run<10>("fma_mix", out, cyc, h, blocks);
	run<11>("v_perm", out, cyc, h, blocks);
	return 0;
}
