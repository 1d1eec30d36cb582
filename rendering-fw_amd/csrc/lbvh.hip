// lbvh.hip — BVH construction on the device: Morton order + Karras' parallel hierarchy over chunks of
// four Morton-consecutive triangles, emitted in the reference's BVH2 node layout (bvh_node.h:23-28: children of an
// inner node adjacent, node 1 unused) and fitted bottom-up by the refit kernels of kernels.hip.
//
//   k_lbvh_bounds    per triangle: union of the triangle boxes (ordered-int atomics)                 -> bounds[6]
//   k_lbvh_morton    per triangle: 63-bit Morton code of the centroid in those bounds                -> keys, vals
//   radix sort       rocPRIM radix_sort_pairs on (code, triangle)   [a library primitive, like a GEMM would be]
//   k_lbvh_hierarchy per inner node of the chunk tree: range + split (Karras 2012), children's slots
//   k_lbvh_emit      per inner node / per chunk: rt::Node in device form + parent links
//   k_lbvh_leaf_ids  per leaf slot: the triangle id the refit kernel resolves vertices from
//
// The host side (rfwhip_api.cpp, builder=device) then runs launch_refit on the mesh-local arrays, downloads nodes,
// parents and leaf-ordered vertices, collapses to 4-wide nodes and places the mesh like a host-built one.  The SAH
// builder of bvh_build.cpp stays the default: its trees are what the traversal numbers in DESIGN.md rest on.
#include "kernels.h"
#include "rt_core.h"

#include <stdint.h>
#include <string.h>

#if !defined(RFWHIP_HOST_EMULATION)
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#else
#include <algorithm>
#include <vector>
#endif

namespace rtk
{
using namespace rt;

constexpr uint32_t CHUNK = LBVH_CHUNK; // Morton-consecutive triangles per leaf
constexpr uint32_t LEAF_REF = 0x80000000u; // child reference: chunk (leaf) index instead of inner-node index

// floats as unsigned keys that order like the floats (for atomicMin / atomicMax)
RT_FN uint32_t float_key(float f)
{
	const uint32_t b = fbits(f);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
RT_FN float key_float(uint32_t k)
{
	const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
	return ubits(b);
}
RT_FN void tri_corners(const f4 *verts, const uint32_t *indices, uint32_t t, f3 &a, f3 &b, f3 &c)
{
	uint32_t i0 = 3u * t, i1 = i0 + 1u, i2 = i0 + 2u;
	if (indices)
		i0 = indices[3ull * t], i1 = indices[3ull * t + 1], i2 = indices[3ull * t + 2];
	a = xyz(verts[i0]), b = xyz(verts[i1]), c = xyz(verts[i2]);
}
// 21 bits per axis -> 63-bit code (a 30-bit code leaves many of a million triangles with equal keys, and equal keys
// split by index, not by space)
RT_FN uint64_t expand21(uint32_t v)
{
	uint64_t x = v & 0x1FFFFFu;
	x = (x | (x << 32)) & 0x1F00000000FFFFull;
	x = (x | (x << 16)) & 0x1F0000FF0000FFull;
	x = (x | (x << 8)) & 0x100F00F00F00F00Full;
	x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
	x = (x | (x << 2)) & 0x1249249249249249ull;
	return x;
}
RT_FN uint64_t morton_item(const f4 *verts, const uint32_t *indices, const uint32_t *bounds, uint32_t t)
{
	f3 a, b, c;
	tri_corners(verts, indices, t, a, b, c);
	const f3 ctr = ((a + b) + c) * (1.0f / 3.0f);
	const float lo[3] = {key_float(bounds[0]), key_float(bounds[1]), key_float(bounds[2])};
	const float hi[3] = {key_float(bounds[3]), key_float(bounds[4]), key_float(bounds[5])};
	const float p[3] = {ctr.x, ctr.y, ctr.z};
	uint32_t q[3];
	for (int k = 0; k < 3; k++)
	{
		const float e = hi[k] - lo[k];
		float u = e > 0.0f ? (p[k] - lo[k]) / e : 0.0f;
		u = fminf(fmaxf(u, 0.0f), 1.0f);
		q[k] = (uint32_t)(u * 2097151.0f);
	}
	return (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
}

// Karras 2012: common-prefix length of the keys of chunks i and j; equal keys fall back to the indices.
RT_FN int lbvh_delta(const uint64_t *keys, int m, int i, int j)
{
	if (j < 0 || j >= m)
		return -1;
	const uint64_t x = keys[CHUNK * (uint32_t)i] ^ keys[CHUNK * (uint32_t)j];
#if defined(__HIP_DEVICE_COMPILE__)
	return x ? __clzll((long long)x) : 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
#else
	return x ? __builtin_clzll(x) : 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
#endif
}
// inner node i of the tree over m chunks: children + where they will live (the pair of slots behind inner node i)
RT_FN void hierarchy_item(const uint64_t *keys, int m, uint32_t *child, uint32_t *slot_inner, uint32_t *slot_leaf, int i)
{
	const int d = lbvh_delta(keys, m, i, i + 1) - lbvh_delta(keys, m, i, i - 1) >= 0 ? 1 : -1;
	const int dmin = lbvh_delta(keys, m, i, i - d);
	int lmax = 2;
	while (lbvh_delta(keys, m, i, i + lmax * d) > dmin)
		lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2)
		if (lbvh_delta(keys, m, i, i + (l + t) * d) > dmin)
			l += t;
	const int j = i + l * d;
	const int dnode = lbvh_delta(keys, m, i, j);
	int s = 0;
	for (int t = (l + 1) / 2;; t = (t + 1) / 2)
	{
		if (lbvh_delta(keys, m, i, i + (s + t) * d) > dnode)
			s += t;
		if (t <= 1)
			break;
	}
	const int gamma = i + s * d + (d < 0 ? d : 0);
	const int lo = i < j ? i : j, hi = i < j ? j : i;
	const uint32_t left = (lo == gamma) ? (LEAF_REF | (uint32_t)gamma) : (uint32_t)gamma;
	const uint32_t right = (hi == gamma + 1) ? (LEAF_REF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
	child[2 * i] = left, child[2 * i + 1] = right;
	const uint32_t pair = 2u * (uint32_t)i + 2u;
	if (left & LEAF_REF)
		slot_leaf[left & ~LEAF_REF] = pair;
	else
		slot_inner[left] = pair;
	if (right & LEAF_REF)
		slot_leaf[right & ~LEAF_REF] = pair + 1u;
	else
		slot_inner[right] = pair + 1u;
}
RT_FN void emit_inner_item(Node *nodes, int *parents, const uint32_t *slot_inner, uint32_t i)
{
	const uint32_t s = i == 0u ? 0u : slot_inner[i];
	Node n;
	memset(&n, 0, sizeof(n));
	n.left_first = (int)make_entry((int)(2u * i + 2u), -1, false);
	n.count = -1;
	nodes[s] = n;
	parents[2u * i + 2u] = (int)s, parents[2u * i + 3u] = (int)s;
	if (i == 0u)
	{
		parents[0] = -1, parents[1] = -1;
		memset(&n, 0, sizeof(n)); // the unused slot next to the root
		nodes[1] = n;
	}
}
RT_FN void emit_leaf_item(Node *nodes, const uint32_t *slot_leaf, uint32_t tri_count, uint32_t k)
{
	const uint32_t first = CHUNK * k;
	const uint32_t cnt = tri_count - first < CHUNK ? tri_count - first : CHUNK;
	Node n;
	memset(&n, 0, sizeof(n));
	n.left_first = (int)make_entry((int)first, (int)cnt, false);
	n.count = (int)cnt;
	nodes[slot_leaf[k]] = n;
}

#if !defined(RFWHIP_HOST_EMULATION)

__global__ void __launch_bounds__(256) k_lbvh_bounds(const f4 *verts, const uint32_t *indices, uint32_t n, uint32_t *bounds)
{
	float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
	for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u)
	{
		f3 a, b, c;
		tri_corners(verts, indices, t, a, b, c);
		lo[0] = fminf(lo[0], fminf(a.x, fminf(b.x, c.x))), hi[0] = fmaxf(hi[0], fmaxf(a.x, fmaxf(b.x, c.x)));
		lo[1] = fminf(lo[1], fminf(a.y, fminf(b.y, c.y))), hi[1] = fmaxf(hi[1], fmaxf(a.y, fmaxf(b.y, c.y)));
		lo[2] = fminf(lo[2], fminf(a.z, fminf(b.z, c.z))), hi[2] = fmaxf(hi[2], fmaxf(a.z, fmaxf(b.z, c.z)));
	}
	// wave reduction, then one atomic per wave and component
	for (int k = 0; k < 3; k++)
	{
		for (int o = 32; o > 0; o >>= 1)
			lo[k] = fminf(lo[k], __shfl_down(lo[k], o)), hi[k] = fmaxf(hi[k], __shfl_down(hi[k], o));
		if ((threadIdx.x & 63u) == 0u)
		{
			atomicMin(&bounds[k], float_key(lo[k]));
			atomicMax(&bounds[3 + k], float_key(hi[k]));
		}
	}
}
__global__ void __launch_bounds__(256) k_lbvh_morton(const f4 *verts, const uint32_t *indices, const uint32_t *bounds, uint32_t n,
												  uint64_t *keys, uint32_t *vals)
{
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t < n)
		keys[t] = morton_item(verts, indices, bounds, t), vals[t] = t;
}
__global__ void __launch_bounds__(256) k_lbvh_hierarchy(const uint64_t *keys, int m, uint32_t *child, uint32_t *slot_inner,
													 uint32_t *slot_leaf)
{
	const int i = (int)(blockIdx.x * 256u + threadIdx.x);
	if (i < m - 1)
		hierarchy_item(keys, m, child, slot_inner, slot_leaf, i);
}
__global__ void __launch_bounds__(256) k_lbvh_emit(Node *nodes, int *parents, const uint32_t *slot_inner, const uint32_t *slot_leaf,
												uint32_t m, uint32_t tri_count)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i + 1u < m)
		emit_inner_item(nodes, parents, slot_inner, i);
	if (i < m)
		emit_leaf_item(nodes, slot_leaf, tri_count, i);
}
__global__ void __launch_bounds__(256) k_lbvh_leaf_ids(f4 *tri_verts, const uint32_t *sorted_tris, uint32_t n)
{
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < n)
		tri_verts[3ull * s] = mk4(0.0f, 0.0f, 0.0f, ubits(sorted_tris[s]));
}

size_t lbvh_scratch_bytes(uint32_t tri_count)
{
	size_t sort_tmp = 0;
	uint64_t *k = nullptr;
	uint32_t *v = nullptr;
	(void)rocprim::radix_sort_pairs(nullptr, sort_tmp, k, k, v, v, (size_t)tri_count, 0u, 63u, (hipStream_t)0);
	const size_t n = tri_count, m = (n + CHUNK - 1) / CHUNK;
	// bounds[8] | keys (8 B) | vals | keys_sorted (8 B) | vals_sorted | child[2m] | slot_inner[m] | slot_leaf[m] | sort temp
	return 256 + 24 * n + 16 * m + 64 * 8 + sort_tmp + 256;
}

int launch_lbvh_build(const f4 *verts, const uint32_t *indices, uint32_t tri_count, void *scratch, size_t scratch_bytes,
					  Node *nodes, int *parents, f4 *tri_verts, uint32_t *flags, float bounds_out_device[6], stream_t s)
{
	(void)bounds_out_device;
	hipStream_t st = (hipStream_t)s;
	const uint32_t n = tri_count, m = (n + CHUNK - 1u) / CHUNK;
	if (m < 2u)
		return 1; // a single leaf: the caller builds those on the host
	uint8_t *p = (uint8_t *)scratch;
	auto take = [&](size_t bytes) {
		uint8_t *r = p;
		p += (bytes + 63) & ~size_t(63);
		return r;
	};
	uint32_t *bounds = (uint32_t *)take(32);
	uint64_t *keys = (uint64_t *)take(8ull * n), *keys2 = (uint64_t *)take(8ull * n);
	uint32_t *vals = (uint32_t *)take(4ull * n), *vals2 = (uint32_t *)take(4ull * n);
	uint32_t *child = (uint32_t *)take(8ull * m), *slot_inner = (uint32_t *)take(4ull * m), *slot_leaf = (uint32_t *)take(4ull * m);
	size_t sort_tmp = 0;
	(void)rocprim::radix_sort_pairs(nullptr, sort_tmp, keys, keys2, vals, vals2, (size_t)n, 0u, 63u, st);
	void *tmp = take(sort_tmp);
	if ((size_t)(p - (uint8_t *)scratch) > scratch_bytes)
		return 2;
	const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
	(void)hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, st);
	const uint32_t blocks = (n + 255u) / 256u;
	hipLaunchKernelGGL(k_lbvh_bounds, dim3(blocks < 2048u ? blocks : 2048u), dim3(256), 0, st, verts, indices, n, bounds);
	hipLaunchKernelGGL(k_lbvh_morton, dim3(blocks), dim3(256), 0, st, verts, indices, bounds, n, keys, vals);
	if (rocprim::radix_sort_pairs(tmp, sort_tmp, keys, keys2, vals, vals2, (size_t)n, 0u, 63u, st) != hipSuccess)
		return 3;
	hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((m + 255u) / 256u), dim3(256), 0, st, keys2, (int)m, child, slot_inner, slot_leaf);
	hipLaunchKernelGGL(k_lbvh_emit, dim3((m + 255u) / 256u), dim3(256), 0, st, nodes, parents, slot_inner, slot_leaf, m, n);
	hipLaunchKernelGGL(k_lbvh_leaf_ids, dim3(blocks), dim3(256), 0, st, tri_verts, vals2, n);
	// boxes: the refit kernels (leaf-ordered vertices from the ids, leaf boxes, bottom-up merge)
	launch_refit(nodes, 0u, parents, 2u * m, tri_verts, 0u, verts, indices, n, flags, s);
	return hipGetLastError() == hipSuccess ? 0 : 4;
}

#else // host emulation: the same items, plain loops; std::stable_sort stands in for the radix sort

size_t lbvh_scratch_bytes(uint32_t tri_count) { return 64 + 48ull * tri_count; }

int launch_lbvh_build(const f4 *verts, const uint32_t *indices, uint32_t tri_count, void *, size_t, Node *nodes, int *parents,
					  f4 *tri_verts, uint32_t *flags, float *, stream_t s)
{
	const uint32_t n = tri_count, m = (n + CHUNK - 1u) / CHUNK;
	if (m < 2u)
		return 1;
	uint32_t bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
	for (uint32_t t = 0; t < n; t++)
	{
		f3 a, b, c;
		tri_corners(verts, indices, t, a, b, c);
		const float lo[3] = {fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z))};
		const float hi[3] = {fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z))};
		for (int k = 0; k < 3; k++)
		{
			bounds[k] = std::min(bounds[k], float_key(lo[k]));
			bounds[3 + k] = std::max(bounds[3 + k], float_key(hi[k]));
		}
	}
	std::vector<uint64_t> keys(n);
	std::vector<uint32_t> vals(n), order(n);
	for (uint32_t t = 0; t < n; t++)
		keys[t] = morton_item(verts, indices, bounds, t), order[t] = t;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
	std::vector<uint64_t> keys2(n);
	for (uint32_t t = 0; t < n; t++)
		keys2[t] = keys[order[t]], vals[t] = order[t];
	std::vector<uint32_t> child(2ull * m), slot_inner(m), slot_leaf(m);
	for (int i = 0; i + 1 < (int)m; i++)
		hierarchy_item(keys2.data(), (int)m, child.data(), slot_inner.data(), slot_leaf.data(), i);
	for (uint32_t i = 0; i + 1u < m; i++)
		emit_inner_item(nodes, parents, slot_inner.data(), i);
	for (uint32_t k = 0; k < m; k++)
		emit_leaf_item(nodes, slot_leaf.data(), n, k);
	for (uint32_t sidx = 0; sidx < n; sidx++)
		tri_verts[3ull * sidx] = mk4(0.0f, 0.0f, 0.0f, ubits(vals[sidx]));
	launch_refit(nodes, 0u, parents, 2u * m, tri_verts, 0u, verts, indices, n, flags, s);
	return 0;
}

#endif

} // namespace rtk
