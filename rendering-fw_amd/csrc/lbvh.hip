// lbvh.hip — BVH construction on the device, end to end: nothing but four counters and the root box comes
// back to the host.
//
//   k_lbvh_bounds / k_lbvh_morton / rocPRIM radix sort     triangles in Morton order (63-bit codes of the centroids)
//   k_ploc_init                                            one cluster per triangle, in that order
//   k_ploc_nearest / k_ploc_merge / rocPRIM select         parallel locally-ordered clustering (Meister & Bittner 2018):
//                                                          every cluster looks RT_PLOC_RADIUS neighbours to either side for
//                                                          the partner with the smallest union box; mutual choices merge.
//                                                          Bottom-up agglomeration by surface area — the cost the
//                                                          reference's SAH (bvh_node.h:136-233) minimises top-down.  Nodes
//                                                          are emitted in the reference's BVH2 layout (children adjacent,
//                                                          node 1 unused, bvh_node.h:23-28), one triangle per BVH2 leaf.
//   k_subtree_sizes                                        triangles below every node (bottom-up, like the refit)
//   k_collapse_level (one launch per tree level)           the 4-wide collapse of bvh_build.cpp on the device: open the
//                                                          largest inner child until four; a subtree of <= 4 triangles
//                                                          becomes ONE leaf entry and its triangles are written in
//                                                          depth-first order (so every leaf's triangles are contiguous);
//                                                          the node is quantised (rt::pack_boxes4c) where it is made;
//                                                          worst-case traversal-stack need tracked like bvh::stack_need4
//
// The host builder of bvh_build.cpp stays the default; this one is for meshes whose topology changes.
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

// ---- clusters ------------------------------------------------------------------------------------------------------------
#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 64 // swept on the 1 M-triangle terrain: 8 / 16 / 32 / 64 -> 4.35 / 4.29 / 4.21 / 4.03 ms per frame (SAH tree: 3.4), build time flat
#endif
constexpr int PLOC_RADIUS = RT_PLOC_RADIUS;
constexpr uint32_t CL_LEAF = 0x80000000u; // Cluster::ref: sorted triangle position instead of a child-pair index
struct alignas(16) Cluster
{
	float lo[3];
	uint32_t ref;
	float hi[3];
	uint32_t pad;
};
struct BuildCounters
{
	uint32_t nodes;		 // next free BVH2 node (pairs: starts at 2)
	uint32_t merged;	 // merges of the current clustering pass
	uint32_t nodes4;	 // next free 4-wide node (starts at 1: the root)
	uint32_t stack_need; // worst-case traversal-stack entries (bvh::stack_need4)
	uint32_t queue[2];	 // collapse tasks of the current / next level
	uint32_t pad[2];
};
struct CollapseTask
{
	uint32_t n2, idx4, start, above;
};
RT_FN float box_area(const float lo[3], const float hi[3])
{
	const float e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
	return e0 * e1 + e0 * e2 + e1 * e2;
}
RT_FN void ploc_init_item(Cluster *cl, const f4 *verts, const uint32_t *indices, const uint32_t *sorted_tris, uint32_t s)
{
	f3 a, b, c;
	tri_corners(verts, indices, sorted_tris[s], a, b, c);
	Cluster k;
	// per-triangle boxes are grown by 1e-5 (bvh_tree.cpp:412), node boxes once more (bvh_node.h:218-219): as leaf_bounds()
	k.lo[0] = fminf(a.x, fminf(b.x, c.x)) - 2e-5f, k.hi[0] = fmaxf(a.x, fmaxf(b.x, c.x)) + 2e-5f;
	k.lo[1] = fminf(a.y, fminf(b.y, c.y)) - 2e-5f, k.hi[1] = fmaxf(a.y, fmaxf(b.y, c.y)) + 2e-5f;
	k.lo[2] = fminf(a.z, fminf(b.z, c.z)) - 2e-5f, k.hi[2] = fmaxf(a.z, fmaxf(b.z, c.z)) + 2e-5f;
	k.ref = CL_LEAF | s, k.pad = 0u;
	cl[s] = k;
}
// the neighbour within PLOC_RADIUS whose union with cluster i has the smallest area; ties go to the lower index
RT_FN void ploc_nearest_item(const Cluster *cl, uint32_t n, uint32_t *nearest, uint32_t i)
{
	const Cluster me = cl[i];
	const uint32_t j0 = i > (uint32_t)PLOC_RADIUS ? i - (uint32_t)PLOC_RADIUS : 0u;
	const uint32_t j1 = i + (uint32_t)PLOC_RADIUS < n - 1u ? i + (uint32_t)PLOC_RADIUS : n - 1u;
	float best = 3.0e38f;
	uint32_t bj = i;
	for (uint32_t j = j0; j <= j1; j++)
	{
		if (j == i)
			continue;
		const Cluster o = cl[j];
		float lo[3], hi[3];
		for (int a = 0; a < 3; a++)
			lo[a] = fminf(me.lo[a], o.lo[a]), hi[a] = fmaxf(me.hi[a], o.hi[a]);
		const float ar = box_area(lo, hi);
		if (ar < best)
			best = ar, bj = j;
	}
	nearest[i] = bj;
}
RT_FN Node cluster_node(const Cluster &k)
{
	Node n;
	for (int a = 0; a < 3; a++)
		n.bmin[a] = k.lo[a], n.bmax[a] = k.hi[a];
	if (k.ref & CL_LEAF) // the sorted position stays here until the collapse assigns the depth-first slot
		n.left_first = (int)k.ref, n.count = 1;
	else
		n.left_first = (int)make_entry((int)k.ref, -1, false), n.count = -1;
	return n;
}
// writes cluster k's node record into slot `at`; the children of an inner cluster learn where their parent lives
RT_FN void place_cluster(Node *nodes, int *parents, const Cluster &k, uint32_t at)
{
	nodes[at] = cluster_node(k);
	if (!(k.ref & CL_LEAF))
		parents[k.ref] = (int)at, parents[k.ref + 1u] = (int)at;
}
// mutual nearest neighbours merge (the lower index keeps the merged cluster); `force` pairs up neighbours 2k, 2k+1 instead
// (used when a pass found no mutual pair — possible only with exact area ties — so that every pass makes progress)
RT_FN void ploc_merge_item(const Cluster *cl, uint32_t n, const uint32_t *nearest, Cluster *out, uint32_t *keep, Node *nodes,
						   int *parents, BuildCounters *bc, bool force, uint32_t i)
{
	uint32_t j = nearest[i];
	bool mutual = j != i && nearest[j] == i;
	if (force)
	{
		j = i ^ 1u;
		mutual = j < n;
	}
	if (!mutual)
	{
		out[i] = cl[i], keep[i] = 1u;
		return;
	}
	if (i > j)
	{
		keep[i] = 0u;
		return;
	}
	const Cluster a = cl[i], b = cl[j];
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t pair = atomicAdd(&bc->nodes, 2u);
	atomicAdd(&bc->merged, 1u);
#else
	const uint32_t pair = bc->nodes;
	bc->nodes += 2u, bc->merged += 1u;
#endif
	place_cluster(nodes, parents, a, pair);
	place_cluster(nodes, parents, b, pair + 1u);
	Cluster m;
	for (int k = 0; k < 3; k++)
		m.lo[k] = fminf(a.lo[k], b.lo[k]), m.hi[k] = fmaxf(a.hi[k], b.hi[k]);
	m.ref = pair, m.pad = 0u;
	out[i] = m, keep[i] = 1u;
}

// ---- subtree sizes (triangles below every node): leaves start, the second child to arrive at a parent adds up ----------------
RT_FN uint32_t node_left(const Node &n) { return (uint32_t)n.left_first & ENTRY_INDEX_MASK; }

// ---- 4-wide collapse + quantisation + depth-first triangle order --------------------------------------------------------------
// the <= MAX triangles of a small subtree in depth-first order: their vertices go to tri_verts[start ..], their BVH2 leaves get
// their final device entries
RT_FN uint32_t emit_small_subtree(Node *nodes, uint32_t root, uint32_t start, f4 *tri_verts, const f4 *verts, const uint32_t *indices,
								  const uint32_t *sorted_tris)
{
	uint32_t stack[2 * MAX_LEAF_PRIMS], sp = 0, pos = start;
	stack[sp++] = root;
	while (sp)
	{
		const uint32_t i = stack[--sp];
		Node &n = nodes[i];
		if (n.count < 0)
		{
			const uint32_t l = node_left(n);
			stack[sp++] = l + 1u, stack[sp++] = l; // left first
			continue;
		}
		const uint32_t tri = sorted_tris[(uint32_t)n.left_first & ~CL_LEAF];
		f3 a, b, c;
		tri_corners(verts, indices, tri, a, b, c);
		tri_verts[3ull * pos] = mk4(a.x, a.y, a.z, ubits(tri));
		tri_verts[3ull * pos + 1] = mk4(b.x, b.y, b.z, 1.0f);
		tri_verts[3ull * pos + 2] = mk4(c.x, c.y, c.z, TRI_EPS); // (w: the triangle's determinant threshold, rt::tri_test)
		n.left_first = (int)make_entry((int)pos, 1, false);
		pos++;
	}
	return pos - start;
}
#ifndef RT_DEVICE_MAX_LEAF
#define RT_DEVICE_MAX_LEAF 4
#endif
RT_FN void collapse_item(const CollapseTask &t, Node *nodes, const uint32_t *sizes, Node4c *nodes4, uint32_t *src4, f4 *tri_verts,
						 const f4 *verts, const uint32_t *indices, const uint32_t *sorted_tris, BuildCounters *bc,
						 CollapseTask *next, uint32_t next_cap)
{
	const uint32_t MAXL = RT_DEVICE_MAX_LEAF;
	uint32_t kids[4], kstart[4];
	const uint32_t l = node_left(nodes[t.n2]);
	kids[0] = l, kids[1] = l + 1u, kstart[0] = t.start, kstart[1] = t.start + sizes[l];
	int nk = 2;
	while (nk < 4)
	{
		int best = -1;
		float best_area = -1.0f;
		for (int k = 0; k < nk; k++)
			if (sizes[kids[k]] > MAXL)
			{
				const float ar = box_area(nodes[kids[k]].bmin, nodes[kids[k]].bmax);
				if (ar > best_area)
					best = k, best_area = ar;
			}
		if (best < 0)
			break;
		const uint32_t cl = node_left(nodes[kids[best]]);
		const uint32_t st = kstart[best];
		kids[best] = cl, kstart[best] = st;
		kids[nk] = cl + 1u, kstart[nk] = st + sizes[cl];
		nk++;
	}
	const uint32_t here = t.above + (uint32_t)(nk - 1);
#if defined(__HIP_DEVICE_COMPILE__)
	atomicMax(&bc->stack_need, here);
#else
	bc->stack_need = here > bc->stack_need ? here : bc->stack_need;
#endif
	Node4c out;
	float lo[3][4], hi[3][4];
	bool valid[4];
	for (int k = 0; k < 4; k++)
	{
		valid[k] = k < nk;
		out.entry[k] = ENTRY_EMPTY;
		src4[4u * t.idx4 + (uint32_t)k] = valid[k] ? kids[k] : 0xFFFFFFFFu;
		for (int a = 0; a < 3; a++)
			lo[a][k] = valid[k] ? nodes[kids[k]].bmin[a] : 0.0f, hi[a][k] = valid[k] ? nodes[kids[k]].bmax[a] : 0.0f;
		if (!valid[k])
			continue;
		const uint32_t sz = sizes[kids[k]];
		if (sz <= MAXL)
		{
			emit_small_subtree(nodes, kids[k], kstart[k], tri_verts, verts, indices, sorted_tris);
			out.entry[k] = make_entry((int)kstart[k], (int)sz, false);
		}
		else
		{
#if defined(__HIP_DEVICE_COMPILE__)
			const uint32_t c4 = atomicAdd(&bc->nodes4, 1u);
			const uint32_t q = atomicAdd(&bc->queue[1], 1u);
#else
			const uint32_t c4 = bc->nodes4++;
			const uint32_t q = bc->queue[1]++;
#endif
			out.entry[k] = make_entry((int)c4, -1, false);
			if (q < next_cap)
			{
				CollapseTask nt;
				nt.n2 = kids[k], nt.idx4 = c4, nt.start = kstart[k], nt.above = here;
				next[q] = nt;
			}
		}
	}
	pack_boxes4c(out, lo, hi, valid);
	nodes4[t.idx4] = out;
}
// entries of a mesh-local tree -> scene-wide indices (rfwhip_update places the mesh behind others)
RT_FN void rebase_node_item(Node *nodes, uint32_t node_base, uint32_t tri_base, uint32_t i)
{
	Node &n = nodes[i];
	if (n.count > 0)
		n.left_first = (int)(((uint32_t)n.left_first & ~ENTRY_FIRST_MASK) | ((((uint32_t)n.left_first & ENTRY_FIRST_MASK) + tri_base) & ENTRY_FIRST_MASK));
	else if (n.count < 0)
		n.left_first = (int)((uint32_t)n.left_first + node_base);
}
RT_FN void rebase_node4_item(Node4c *nodes4, uint32_t n4_base, uint32_t tri_base, uint32_t i)
{
	Node4c &n = nodes4[i];
	for (int k = 0; k < 4; k++)
	{
		const uint32_t e = n.entry[k];
		if (e == ENTRY_EMPTY)
			continue;
		if (e & ENTRY_LEAF)
			n.entry[k] = (e & ~ENTRY_FIRST_MASK) | (((e & ENTRY_FIRST_MASK) + tri_base) & ENTRY_FIRST_MASK);
		else
			n.entry[k] = e + n4_base;
	}
}

// subtree sizes: leaves count 1 and walk up; the second child to arrive at a parent (agent-scope counter) adds the two
RT_FN bool subtree_leaf(const Node &n) { return n.count > 0; }

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
__global__ void __launch_bounds__(256) k_ploc_init(Cluster *cl, const f4 *verts, const uint32_t *indices, const uint32_t *sorted_tris, uint32_t n)
{
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < n)
		ploc_init_item(cl, verts, indices, sorted_tris, s);
}
__global__ void __launch_bounds__(256) k_ploc_nearest(const Cluster *cl, uint32_t n, uint32_t *nearest)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n)
		ploc_nearest_item(cl, n, nearest, i);
}
__global__ void __launch_bounds__(256) k_ploc_merge(const Cluster *cl, uint32_t n, const uint32_t *nearest, Cluster *out, uint32_t *keep,
												 Node *nodes, int *parents, BuildCounters *bc, int force)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n)
		ploc_merge_item(cl, n, nearest, out, keep, nodes, parents, bc, force != 0, i);
}
__global__ void k_ploc_root(const Cluster *cl, Node *nodes, int *parents)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
	{
		place_cluster(nodes, parents, cl[0], 0u);
		parents[0] = -1, parents[1] = -1;
		Node z;
		memset(&z, 0, sizeof(z));
		nodes[1] = z;
	}
}
__global__ void __launch_bounds__(256) k_subtree_sizes(const Node *nodes, const int *parents, uint32_t node_count, uint32_t *sizes, uint32_t *flags)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= node_count || i == 1u || !subtree_leaf(nodes[i]))
		return;
	sizes[i] = 1u;
	int cur = (int)i;
	for (;;)
	{
		const int parent = parents[cur];
		if (parent < 0)
			break;
		__threadfence();
		if (atomicAdd(&flags[parent], 1u) == 0u)
			break;
		__threadfence();
		const uint32_t l = node_left(nodes[parent]);
		const uint32_t a = __hip_atomic_load(&sizes[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t b = __hip_atomic_load(&sizes[l + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		sizes[parent] = a + b;
		cur = parent;
	}
}
__global__ void __launch_bounds__(256) k_collapse_level(const CollapseTask *tasks, uint32_t count, Node *nodes, const uint32_t *sizes,
													 Node4c *nodes4, uint32_t *src4, f4 *tri_verts, const f4 *verts, const uint32_t *indices,
													 const uint32_t *sorted_tris, BuildCounters *bc, CollapseTask *next, uint32_t next_cap)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < count)
		collapse_item(tasks[i], nodes, sizes, nodes4, src4, tri_verts, verts, indices, sorted_tris, bc, next, next_cap);
}
__global__ void k_next_level(BuildCounters *bc)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
		bc->queue[0] = bc->queue[1], bc->queue[1] = 0u;
}
__global__ void __launch_bounds__(256) k_rebase_nodes(Node *nodes, uint32_t count, uint32_t node_base, uint32_t tri_base)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < count)
		rebase_node_item(nodes, node_base, tri_base, i);
}
__global__ void __launch_bounds__(256) k_rebase_nodes4(Node4c *nodes4, uint32_t count, uint32_t n4_base, uint32_t tri_base)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < count)
		rebase_node4_item(nodes4, n4_base, tri_base, i);
}

struct BuildScratch
{
	uint32_t *bounds;
	uint64_t *keys, *keys2;
	uint32_t *vals, *vals2;
	Cluster *cl[2];
	uint32_t *nearest, *keep, *sizes;
	CollapseTask *tasks[2];
	BuildCounters *bc;
	uint32_t *selected;
	void *tmp;
	size_t tmp_bytes, total;
};
static BuildScratch carve(void *scratch, uint32_t n)
{
	BuildScratch b;
	uint8_t *p = (uint8_t *)scratch;
	auto take = [&](size_t bytes) {
		uint8_t *r = p;
		p += (bytes + 255) & ~size_t(255);
		return r;
	};
	b.bounds = (uint32_t *)take(32);
	b.bc = (BuildCounters *)take(sizeof(BuildCounters));
	b.selected = (uint32_t *)take(16);
	b.keys = (uint64_t *)take(8ull * n), b.keys2 = (uint64_t *)take(8ull * n);
	b.vals = (uint32_t *)take(4ull * n), b.vals2 = (uint32_t *)take(4ull * n);
	b.cl[0] = (Cluster *)take(sizeof(Cluster) * (size_t)n), b.cl[1] = (Cluster *)take(sizeof(Cluster) * (size_t)n);
	b.nearest = (uint32_t *)take(4ull * n), b.keep = (uint32_t *)take(4ull * n);
	b.sizes = (uint32_t *)take(8ull * n);
	b.tasks[0] = (CollapseTask *)take(sizeof(CollapseTask) * (size_t)n), b.tasks[1] = (CollapseTask *)take(sizeof(CollapseTask) * (size_t)n);
	size_t sort_tmp = 0, sel_tmp = 0;
	(void)rocprim::radix_sort_pairs(nullptr, sort_tmp, b.keys, b.keys2, b.vals, b.vals2, (size_t)n, 0u, 63u, (hipStream_t)0);
	(void)rocprim::select(nullptr, sel_tmp, b.cl[0], b.keep, b.cl[1], b.selected, (size_t)n, (hipStream_t)0);
	b.tmp_bytes = sort_tmp > sel_tmp ? sort_tmp : sel_tmp;
	b.tmp = take(b.tmp_bytes);
	b.total = (size_t)(p - (uint8_t *)scratch);
	return b;
}

size_t device_build_scratch_bytes(uint32_t tri_count) { return carve(nullptr, tri_count).total + 256; }

int launch_device_build(const f4 *verts, const uint32_t *indices, uint32_t tri_count, void *scratch, size_t scratch_bytes, Node *nodes,
						int *parents, uint32_t *flags, Node4c *nodes4, uint32_t *src4, f4 *tri_verts, DeviceBuildResult *out, stream_t s)
{
	hipStream_t st = (hipStream_t)s;
	const uint32_t n = tri_count;
	if (n < 2u)
		return 1; // a single leaf: the caller builds those on the host
	const BuildScratch b = carve(scratch, n);
	if (b.total > scratch_bytes)
		return 2;
	const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
	(void)hipMemcpyAsync(b.bounds, init, sizeof(init), hipMemcpyHostToDevice, st);
	BuildCounters h;
	memset(&h, 0, sizeof(h));
	h.nodes = 2u, h.nodes4 = 1u, h.queue[0] = 1u;
	(void)hipMemcpyAsync(b.bc, &h, sizeof(h), hipMemcpyHostToDevice, st);
	const uint32_t blocks = (n + 255u) / 256u;
	hipLaunchKernelGGL(k_lbvh_bounds, dim3(blocks < 2048u ? blocks : 2048u), dim3(256), 0, st, verts, indices, n, b.bounds);
	hipLaunchKernelGGL(k_lbvh_morton, dim3(blocks), dim3(256), 0, st, verts, indices, b.bounds, n, b.keys, b.vals);
	size_t tmp_bytes = b.tmp_bytes;
	if (rocprim::radix_sort_pairs(b.tmp, tmp_bytes, b.keys, b.keys2, b.vals, b.vals2, (size_t)n, 0u, 63u, st) != hipSuccess)
		return 3;
	// ---- clustering ----
	hipLaunchKernelGGL(k_ploc_init, dim3(blocks), dim3(256), 0, st, b.cl[0], verts, indices, b.vals2, n);
	uint32_t cur = n;
	int in = 0;
	const uint32_t zero = 0u;
	// Every pass merges the mutual nearest-neighbour pairs (about a third of the clusters: ~25 passes for 1 M triangles); a pass
	// that finds no mutual pair pairs neighbours in Morton order instead.  The pass count is bounded: input that makes the
	// clustering crawl (5 = "did not converge") is handed to the host builder by the caller instead of looping on host round trips.
	uint32_t passes = 0, max_passes = 64u;
	for (uint32_t m = n; m > 1u; m >>= 1)
		max_passes += 8u;
	while (cur > 1u)
	{
		if (++passes > max_passes)
			return 5;
		const dim3 g((cur + 255u) / 256u);
		hipLaunchKernelGGL(k_ploc_nearest, g, dim3(256), 0, st, b.cl[in], cur, b.nearest);
		uint32_t merged = 0;
		for (int force = 0; force < 2 && merged == 0u; force++)
		{
			(void)hipMemcpyAsync(&b.bc->merged, &zero, 4, hipMemcpyHostToDevice, st);
			hipLaunchKernelGGL(k_ploc_merge, g, dim3(256), 0, st, b.cl[in], cur, b.nearest, b.cl[in ^ 1], b.keep, nodes, parents, b.bc, force);
			if (hipMemcpyAsync(&merged, &b.bc->merged, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
				return 4;
		}
		if (merged == 0u)
			return 5;
		// the surviving clusters, order kept, back into the other array
		tmp_bytes = b.tmp_bytes;
		if (rocprim::select(b.tmp, tmp_bytes, b.cl[in ^ 1], b.keep, b.cl[in], b.selected, (size_t)cur, st) != hipSuccess)
			return 6;
		cur -= merged;
	}
	hipLaunchKernelGGL(k_ploc_root, dim3(1), dim3(64), 0, st, b.cl[in], nodes, parents);
	// ---- sizes, collapse ----
	const uint32_t node_count = 2u * n;
	(void)hipMemsetAsync(flags, 0, sizeof(uint32_t) * node_count, st);
	(void)hipMemsetAsync(b.sizes, 0, sizeof(uint32_t) * node_count, st);
	hipLaunchKernelGGL(k_subtree_sizes, dim3((node_count + 255u) / 256u), dim3(256), 0, st, nodes, parents, node_count, b.sizes, flags);
	const CollapseTask root = {0u, 0u, 0u, 0u};
	(void)hipMemcpyAsync(b.tasks[0], &root, sizeof(root), hipMemcpyHostToDevice, st);
	int q = 0;
	for (uint32_t count = 1u; count > 0u;)
	{
		hipLaunchKernelGGL(k_collapse_level, dim3((count + 255u) / 256u), dim3(256), 0, st, b.tasks[q], count, nodes, b.sizes, nodes4, src4,
						   tri_verts, verts, indices, b.vals2, b.bc, b.tasks[q ^ 1], n);
		hipLaunchKernelGGL(k_next_level, dim3(1), dim3(64), 0, st, b.bc);
		if (hipMemcpyAsync(&count, &b.bc->queue[0], 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
			return 7;
		q ^= 1;
	}
	Node rootn;
	if (hipMemcpyAsync(&h, b.bc, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
		hipMemcpyAsync(&rootn, nodes, sizeof(Node), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
		return 8;
	out->node_count = node_count, out->node4_count = h.nodes4, out->stack_need = h.stack_need;
	for (int a = 0; a < 3; a++)
		out->bmin[a] = rootn.bmin[a], out->bmax[a] = rootn.bmax[a];
	return hipGetLastError() == hipSuccess ? 0 : 9;
}

void launch_rebase(Node *nodes, uint32_t node_count, uint32_t node_base, Node4c *nodes4, uint32_t n4_count, uint32_t n4_base,
				   uint32_t tri_base, stream_t s)
{
	if (node_count)
		hipLaunchKernelGGL(k_rebase_nodes, dim3((node_count + 255u) / 256u), dim3(256), 0, (hipStream_t)s, nodes, node_count, node_base, tri_base);
	if (n4_count)
		hipLaunchKernelGGL(k_rebase_nodes4, dim3((n4_count + 255u) / 256u), dim3(256), 0, (hipStream_t)s, nodes4, n4_count, n4_base, tri_base);
}

#else // host emulation: the same items, plain loops; std::stable_sort stands in for the radix sort

size_t device_build_scratch_bytes(uint32_t) { return 64; }

int launch_device_build(const f4 *verts, const uint32_t *indices, uint32_t tri_count, void *, size_t, Node *nodes, int *parents,
						uint32_t *flags, Node4c *nodes4, uint32_t *src4, f4 *tri_verts, DeviceBuildResult *out, stream_t)
{
	const uint32_t n = tri_count;
	if (n < 2u)
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
	std::vector<uint32_t> sorted(n);
	for (uint32_t t = 0; t < n; t++)
		keys[t] = morton_item(verts, indices, bounds, t), sorted[t] = t;
	std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
	std::vector<Cluster> cl(n), nxt(n);
	std::vector<uint32_t> nearest(n), keep(n);
	for (uint32_t s = 0; s < n; s++)
		ploc_init_item(cl.data(), verts, indices, sorted.data(), s);
	BuildCounters bc;
	memset(&bc, 0, sizeof(bc));
	bc.nodes = 2u, bc.nodes4 = 1u;
	uint32_t cur = n;
	while (cur > 1u)
	{
		for (uint32_t i = 0; i < cur; i++)
			ploc_nearest_item(cl.data(), cur, nearest.data(), i);
		for (int force = 0; force < 2; force++)
		{
			bc.merged = 0u;
			for (uint32_t i = 0; i < cur; i++)
				ploc_merge_item(cl.data(), cur, nearest.data(), nxt.data(), keep.data(), nodes, parents, &bc, force != 0, i);
			if (bc.merged)
				break;
		}
		if (!bc.merged)
			return 5;
		uint32_t w = 0;
		for (uint32_t i = 0; i < cur; i++)
			if (keep[i])
				cl[w++] = nxt[i];
		cur = w;
	}
	place_cluster(nodes, parents, cl[0], 0u);
	parents[0] = -1, parents[1] = -1;
	memset(&nodes[1], 0, sizeof(Node));
	const uint32_t node_count = 2u * n;
	std::vector<uint32_t> sizes(node_count, 0u);
	memset(flags, 0, sizeof(uint32_t) * node_count);
	for (uint32_t i = 0; i < node_count; i++)
	{
		if (i == 1u || !subtree_leaf(nodes[i]))
			continue;
		sizes[i] = 1u;
		int c = (int)i;
		for (;;)
		{
			const int parent = parents[c];
			if (parent < 0 || flags[parent]++ == 0u)
				break;
			const uint32_t l = node_left(nodes[parent]);
			sizes[parent] = sizes[l] + sizes[l + 1u];
			c = parent;
		}
	}
	std::vector<CollapseTask> qa(1), qb(n);
	qa[0] = CollapseTask{0u, 0u, 0u, 0u};
	while (!qa.empty())
	{
		bc.queue[1] = 0u;
		for (const CollapseTask &t : qa)
			collapse_item(t, nodes, sizes.data(), nodes4, src4, tri_verts, verts, indices, sorted.data(), &bc, qb.data(), n);
		qa.assign(qb.begin(), qb.begin() + bc.queue[1]);
	}
	out->node_count = node_count, out->node4_count = bc.nodes4, out->stack_need = bc.stack_need;
	for (int a = 0; a < 3; a++)
		out->bmin[a] = nodes[0].bmin[a], out->bmax[a] = nodes[0].bmax[a];
	return 0;
}

void launch_rebase(Node *nodes, uint32_t node_count, uint32_t node_base, Node4c *nodes4, uint32_t n4_count, uint32_t n4_base,
				   uint32_t tri_base, stream_t)
{
	for (uint32_t i = 0; i < node_count; i++)
		rebase_node_item(nodes, node_base, tri_base, i);
	for (uint32_t i = 0; i < n4_count; i++)
		rebase_node4_item(nodes4, n4_base, tri_base, i);
}

#endif

} // namespace rtk
