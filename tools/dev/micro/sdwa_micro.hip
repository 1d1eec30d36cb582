// sdwa_micro.hip — can the byte -> float decode of the compressed node's planes be cheaper than v_cvt_f32_ubyteN (a 4-cycle
// operation on this part)?  Candidate: v_mul_f32 with an SDWA byte operand — the byte, zero-extended, IS a denormal float
// (q * 2^-149); times 2^127 it is q * 2^-22, a normal float, and the plane fma takes A * 2^22 instead of A.  If that multiply
// issues at v_mul_f32's 2-cycle rate the decode costs half.  Every wave runs ITER x 32 independent operations of one kind.
// build: hipcc --offload-arch=gfx950 -O3 sdwa_micro.hip -o sdwa_micro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int ITER = 4096;
#define R8(X) X X X X X X X X
#define R32(X) R8(X) R8(X) R8(X) R8(X)

template <int OP> __global__ __launch_bounds__(256, 8) void k(float *out, uint32_t seed)
{
	uint32_t q = seed + threadIdx.x * 0x01010101u;
	float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	const float big = 1.7014118e38f; // 2^127
	for (int i = 0; i < ITER; i++)
	{
		if (OP == 0)
		{
			R8(asm volatile("v_cvt_f32_ubyte0 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_cvt_f32_ubyte2 %2, %4\n\tv_cvt_f32_ubyte3 %3, %4"
							: "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)
							: "v"(q));)
		}
		else if (OP == 1)
		{
			R8(asm volatile("v_mul_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
							"v_mul_f32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
							"v_mul_f32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\t"
							"v_mul_f32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
							: "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)
							: "v"(q), "v"(big));)
		}
		else if (OP == 2)
		{
			R8(asm volatile("v_mul_f32 %0, %4, %5\n\tv_mul_f32 %1, %4, %5\n\tv_mul_f32 %2, %4, %5\n\tv_mul_f32 %3, %4, %5"
							: "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)
							: "v"(q), "v"(big));)
		}
		q += (uint32_t)i;
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ void k_check(float *out, uint32_t q)
{
	float a0, a1, a2, a3;
	const float big = 1.7014118e38f;
	asm volatile("v_mul_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
				 "v_mul_f32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
				 "v_mul_f32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\t"
				 "v_mul_f32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
				 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)
				 : "v"(q), "v"(big));
	if (threadIdx.x == 0)
		out[0] = a0 * 4194304.0f, out[1] = a1 * 4194304.0f, out[2] = a2 * 4194304.0f, out[3] = a3 * 4194304.0f;
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
	const char *names[3] = {"v_cvt_f32_ubyteN", "v_mul_f32_sdwa byte", "v_mul_f32"};
	for (int op = 0; op < 3; op++)
		for (int rep = 0; rep < 2; rep++)
		{
			(void)hipEventRecord(e0, 0);
			if (op == 0)
				hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 0x05030201u);
			else if (op == 1)
				hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 0x05030201u);
			else
				hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 0x05030201u);
			(void)hipEventRecord(e1, 0);
			(void)hipEventSynchronize(e1);
			float ms = 0;
			(void)hipEventElapsedTime(&ms, e0, e1);
			const double insts = (double)blocks * 4 * ITER * 32;
			printf("%-22s %.3f ms  %.1f G wave-instructions/s  (%.2f clocks per instruction per SIMD)\n", names[op], ms, insts / (ms * 1e-3) / 1e9,
				   (double)p.multiProcessorCount * 4 * p.clockRate * 1e3 / (insts / (ms * 1e-3)));
		}
	// the value check: the bytes 0x80, 0x01, 0xFF, 0x05 through the SDWA multiply, times 2^22: 128 1 255 5 ?
	hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, out, 0x05FF0180u);
	float h[4];
	(void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
	printf("decode check (expect 128 1 255 5): %g %g %g %g\n", h[0], h[1], h[2], h[3]);
	return 0;
}
