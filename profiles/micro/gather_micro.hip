// gather_micro.hip — what bounds a divergent per-lane node fetch on gfx950?
// Every lane walks a dependent chain of random 128-B records ("nodes") in a table that is L2/Infinity-Cache resident,
// like a BVH.  Variants:
//   direct<N> : the lane itself issues N x global_load_dwordx4 on its record (the traversal kernel's pattern)
//   coop      : 8 lanes fetch one record with one coalesced 128-B access, records pass through LDS to their owner
// build: hipcc --offload-arch=gfx950 -O3 gather_micro.hip -o gather_micro ; run: ./gather_micro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 64;

template <int N> __global__ __launch_bounds__(256) void k_direct(const u4 *tab, uint32_t mask, uint32_t *out)
{
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	uint32_t cur = (tid * 2654435761u) & mask;
	uint32_t acc = 0;
	for (int s = 0; s < STEPS; s++)
	{
		const u4 *rec = tab + size_t(cur) * 8;
		u4 v[N];
#pragma unroll
		for (int j = 0; j < N; j++) v[j] = rec[j];
		uint32_t x = 0;
#pragma unroll
		for (int j = 0; j < N; j++) x += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
		acc += x;
		cur = (x + tid) & mask;
	}
	out[tid] = acc;
}

// 8 lanes per record: lane (g = lane >> 3, p = lane & 7) loads piece p of the record wanted by lane 8k + g, k = 0..7
__global__ __launch_bounds__(256) void k_coop(const u4 *tab, uint32_t mask, uint32_t *out)
{
	__shared__ u4 stage[4][64 * 9]; // 144-B stride per record: conflict-free b128 reads
	__shared__ uint32_t want[4][64];
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t g = lane >> 3, p = lane & 7;
	uint32_t cur = (tid * 2654435761u) & mask;
	uint32_t acc = 0;
	for (int s = 0; s < STEPS; s++)
	{
		want[w][(lane & 7) * 8 + (lane >> 3)] = cur; // record of lane 8k+g sits at g*8+k
		__builtin_amdgcn_wave_barrier();
		uint32_t idx[8];
#pragma unroll
		for (int k = 0; k < 8; k++) idx[k] = want[w][g * 8 + k];
		u4 v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = tab[size_t(idx[k]) * 8 + p];
#pragma unroll
		for (int k = 0; k < 8; k++) stage[w][(8 * k + g) * 9 + p] = v[k];
		__builtin_amdgcn_wave_barrier();
		uint32_t x = 0;
#pragma unroll
		for (int j = 0; j < 8; j++)
		{
			const u4 r = stage[w][lane * 9 + j];
			x += r.x ^ r.y ^ r.z ^ r.w;
		}
		__builtin_amdgcn_wave_barrier();
		acc += x;
		cur = (x + tid) & mask;
	}
	out[tid] = acc;
}

// 8 lanes per record through LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write): instruction k lands the
// record of lane L = 8k + g at stage + L * 128; the 16-byte pieces are XOR-swizzled on the SOURCE side (piece slot j holds
// row j ^ s(L), s(L) = (L >> 1) & 7) so that the owner's ds_read_b128 of row r at slot r ^ s(L) is bank-conflict free.
// ROWS rows are used by the owner (the traversal kernel needs 7 of the 8).
template <int ROWS> __global__ __launch_bounds__(256) void k_coop_dma(const u4 *tab, uint32_t mask, uint32_t *out)
{
	__shared__ u4 stage[4][64 * 8];
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t g = lane >> 3, j = lane & 7;
	const uint32_t sL = (lane >> 1) & 7;
	uint32_t cur = (tid * 2654435761u) & mask;
	uint32_t acc = 0;
	for (int s = 0; s < STEPS; s++)
	{
#pragma unroll
		for (int k = 0; k < 8; k++)
		{
			const uint32_t L = 8 * k + g;
			const uint32_t c = __shfl(cur, (int)L);
			const uint32_t row = j ^ ((L >> 1) & 7);
			if (row < ROWS)
				__builtin_amdgcn_global_load_lds((const void *)(tab + size_t(c) * 8 + row), (__attribute__((address_space(3))) void *)&stage[w][k * 64], 16, 0, 0);
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		uint32_t x = 0;
#pragma unroll
		for (int r = 0; r < ROWS; r++)
		{
			const u4 v = stage[w][lane * 8 + (r ^ sL)];
			x += v.x ^ v.y ^ v.z ^ v.w;
		}
		acc += x;
		cur = (x + tid) & mask;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	out[tid] = acc;
}

// 4 lanes per record: lane (q = lane >> 2, c = lane & 3) loads 32 B at record + 32 c of the record its quad walks
__global__ __launch_bounds__(256) void k_quad(const u4 *tab, uint32_t mask, uint32_t *out)
{
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	const uint32_t c = threadIdx.x & 3u, ray = tid >> 2;
	uint32_t cur = (ray * 2654435761u) & mask;
	uint32_t acc = 0;
	for (int s = 0; s < STEPS; s++)
	{
		const u4 *rec = tab + size_t(cur) * 8 + c * 2;
		const u4 a = rec[0], b = rec[1];
		uint32_t x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
		// quad reduction (what a closest-child selection would do)
		x += __shfl_xor(x, 1);
		x += __shfl_xor(x, 2);
		acc += x;
		cur = (x + ray) & mask;
	}
	out[tid] = acc;
}

template <typename F> static float time_ms(F &&launch, int reps)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	launch();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; i++) launch();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main()
{
	const uint32_t threads = 256 * 1024 * 8; // 2M lanes = 8192 blocks
	uint32_t *out;
	CHECK(hipMalloc(&out, threads * 4));
	for (uint32_t log2n : {14u, 19u, 21u}) // 2 MiB (L2), 64 MiB (Infinity Cache), 256 MiB
	{
		const uint32_t n = 1u << log2n;
		std::vector<uint32_t> h(size_t(n) * 32);
		uint32_t s = 12345;
		for (auto &x : h) s = s * 1664525u + 1013904223u, x = s >> 3;
		u4 *tab;
		CHECK(hipMalloc(&tab, size_t(n) * 128));
		CHECK(hipMemcpy(tab, h.data(), size_t(n) * 128, hipMemcpyHostToDevice));
		const double visits = double(threads) * STEPS;
		auto report = [&](const char *name, float ms, int bytes) {
			printf("table %4u MiB  %-9s %7.3f ms  %6.1f Gvisit/s  %6.2f TB/s\n", n >> 13, name, ms, visits / ms * 1e-6,
				   visits * bytes / ms * 1e-9);
		};
		report("direct<2>", time_ms([&] { k_direct<2><<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 32);
		report("direct<4>", time_ms([&] { k_direct<4><<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 64);
		report("direct<8>", time_ms([&] { k_direct<8><<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 128);
		report("direct<7>", time_ms([&] { k_direct<7><<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 112);
		report("dma<8>", time_ms([&] { k_coop_dma<8><<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 128);
		report("dma<7>", time_ms([&] { k_coop_dma<7><<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 112);
		report("coop", time_ms([&] { k_coop<<<threads / 256, 256>>>(tab, n - 1, out); }, 5), 128);
		// same number of record visits: 4 lanes per ray
		{
			uint32_t *out4;
			CHECK(hipMalloc(&out4, size_t(threads) * 16));
			report("quad", time_ms([&] { k_quad<<<threads * 4 / 256, 256>>>(tab, n - 1, out4); }, 5), 128);
			CHECK(hipFree(out4));
		}
		CHECK(hipFree(tab));
	}
	return 0;
}
