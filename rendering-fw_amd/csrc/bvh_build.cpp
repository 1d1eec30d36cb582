// bvh_build.cpp — binned-SAH BVH2 builder (host, C++17).  See bvh_build.h.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <numeric>
#include <thread>

namespace bvh
{
namespace
{

struct Box
{
	float mn[3], mx[3];
	void reset()
	{
		for (int a = 0; a < 3; a++)
			mn[a] = 1e34f, mx[a] = -1e34f;
	}
	void grow(const float *lo, const float *hi)
	{
		for (int a = 0; a < 3; a++)
			mn[a] = std::min(mn[a], lo[a]), mx[a] = std::max(mx[a], hi[a]);
	}
	void grow(const Box &b) { grow(b.mn, b.mx); }
	float half_area() const
	{
		const float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
		return std::max(0.0f, e0 * e1 + e0 * e2 + e1 * e2);
	}
};

#ifndef RT_SAH_BINS
#define RT_SAH_BINS 16
#endif
constexpr int BINS = RT_SAH_BINS;
#ifndef RT_SAH_TRAVERSAL_COST
#define RT_SAH_TRAVERSAL_COST 1.0f
#endif
constexpr float TRAVERSAL_COST = RT_SAH_TRAVERSAL_COST; // relative to one triangle test
constexpr size_t PARALLEL_MIN = 1u << 15;

struct Builder
{
	const float *bmin, *bmax;
	std::vector<float> cen; // n x 3
	int max_leaf, depth_limit;
	rt::Node *nodes;
	int *parents;
	uint32_t *order;
	std::atomic<int> max_depth{0};
	std::atomic<int> tasks{0};
	int max_tasks = 1;

	void set_node(int idx, const Box &b, float pad, int left_first, int count)
	{
		for (int a = 0; a < 3; a++)
			nodes[idx].bmin[a] = b.mn[a] - pad, nodes[idx].bmax[a] = b.mx[a] + pad;
		nodes[idx].left_first = left_first;
		nodes[idx].count = count;
	}

	// Builds the subtree of node `idx` over order[first, first+count) using node slots [pool, pool + 2*count - 2).
	// Every subtree owns a private slot range, so parallel subtrees never contend and the layout is deterministic.
	void subdivide(int idx, uint32_t first, uint32_t count, int depth, int pool)
	{
		int d = max_depth.load();
		while (depth > d && !max_depth.compare_exchange_weak(d, depth))
		{
		}
		if ((int)count <= 1)
			return;
		// bounds of centroids
		float cmn[3] = {1e34f, 1e34f, 1e34f}, cmx[3] = {-1e34f, -1e34f, -1e34f};
		for (uint32_t i = 0; i < count; i++)
		{
			const float *c = &cen[3ull * order[first + i]];
			for (int a = 0; a < 3; a++)
				cmn[a] = std::min(cmn[a], c[a]), cmx[a] = std::max(cmx[a], c[a]);
		}
		Box nb;
		for (int a = 0; a < 3; a++)
			nb.mn[a] = nodes[idx].bmin[a], nb.mx[a] = nodes[idx].bmax[a];
		const float parent_area = std::max(nb.half_area(), 1e-30f);

		int levels_needed = 0;
		for (uint32_t c = count; c > (uint32_t)max_leaf; c = (c + 1) / 2)
			levels_needed++;
		const bool budget_short = depth + levels_needed + 1 >= depth_limit;

		int best_axis = -1, best_bin = -1;
		float best_cost = 1e34f;
		if (!budget_short)
		{
			for (int axis = 0; axis < 3; axis++)
			{
				const float ext = cmx[axis] - cmn[axis];
				if (!(ext > 0.0f))
					continue;
				Box bb[BINS];
				uint32_t bc[BINS];
				for (int b = 0; b < BINS; b++)
					bb[b].reset(), bc[b] = 0;
				const float scale = (float)BINS / ext;
				for (uint32_t i = 0; i < count; i++)
				{
					const uint32_t p = order[first + i];
					int b = (int)((cen[3ull * p + axis] - cmn[axis]) * scale);
					b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
					bb[b].grow(&bmin[3ull * p], &bmax[3ull * p]);
					bc[b]++;
				}
				float right_area[BINS];
				uint32_t right_cnt[BINS];
				Box acc;
				acc.reset();
				uint32_t n = 0;
				for (int b = BINS - 1; b > 0; b--)
				{
					acc.grow(bb[b]);
					n += bc[b];
					right_area[b] = acc.half_area();
					right_cnt[b] = n;
				}
				acc.reset();
				n = 0;
				for (int b = 0; b < BINS - 1; b++)
				{
					acc.grow(bb[b]);
					n += bc[b];
					if (n == 0 || right_cnt[b + 1] == 0)
						continue;
					const float cost = acc.half_area() * (float)n + right_area[b + 1] * (float)right_cnt[b + 1];
					if (cost < best_cost)
						best_cost = cost, best_axis = axis, best_bin = b;
				}
			}
		}
		const float split_cost = TRAVERSAL_COST + best_cost / parent_area;
		if ((int)count <= max_leaf && (best_axis < 0 || (float)count <= split_cost))
			return; // leaf

		uint32_t mid;
		if (best_axis >= 0)
		{
			const float ext = cmx[best_axis] - cmn[best_axis];
			const float scale = (float)BINS / ext;
			const float lo = cmn[best_axis];
			const int bbin = best_bin, ax = best_axis;
			uint32_t *it = std::partition(order + first, order + first + count, [&](uint32_t p) {
				int b = (int)((cen[3ull * p + ax] - lo) * scale);
				b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
				return b <= bbin;
			});
			mid = (uint32_t)(it - (order + first));
		}
		else
		{
			// no usable SAH plane (identical centroids) or depth budget short: median split on the widest axis
			int ax = 0;
			for (int a = 1; a < 3; a++)
				if (cmx[a] - cmn[a] > cmx[ax] - cmn[ax])
					ax = a;
			mid = count / 2;
			std::nth_element(order + first, order + first + mid, order + first + count, [&](uint32_t a, uint32_t b) {
				const float ca = cen[3ull * a + ax], cb = cen[3ull * b + ax];
				return ca < cb || (ca == cb && a < b);
			});
		}
		if (mid == 0 || mid == count)
			mid = count / 2;

		Box lb, rb;
		lb.reset(), rb.reset();
		for (uint32_t i = 0; i < mid; i++)
			lb.grow(&bmin[3ull * order[first + i]], &bmax[3ull * order[first + i]]);
		for (uint32_t i = mid; i < count; i++)
			rb.grow(&bmin[3ull * order[first + i]], &bmax[3ull * order[first + i]]);
		const int left = pool;
		set_node(left, lb, 1e-5f, (int)first, (int)mid);
		set_node(left + 1, rb, 1e-5f, (int)(first + mid), (int)(count - mid));
		parents[left] = idx, parents[left + 1] = idx;
		nodes[idx].left_first = left;
		nodes[idx].count = -1;
		// a subtree over k primitives has at most 2k - 2 descendant nodes: [pool, pool+2) is the child pair, the left
		// subtree's descendants take [pool+2, pool+2*mid), the right one's start at pool + 2*mid
		const int lpool = pool + 2, rpool = pool + 2 * (int)mid;
		if (count >= PARALLEL_MIN && tasks.load() < max_tasks)
		{
			tasks.fetch_add(1);
			auto fut = std::async(std::launch::async, [&]() { subdivide(left, first, mid, depth + 1, lpool); });
			subdivide(left + 1, first + mid, count - mid, depth + 1, rpool);
			fut.get();
			tasks.fetch_sub(1);
		}
		else
		{
			subdivide(left, first, mid, depth + 1, lpool);
			subdivide(left + 1, first + mid, count - mid, depth + 1, rpool);
		}
	}
};

} // namespace

void build(const float *bmin, const float *bmax, size_t n, int max_leaf, int depth_limit, Result &out)
{
	out.nodes.clear(), out.order.clear(), out.parents.clear();
	out.max_depth = 0;
	if (max_leaf < 1)
		max_leaf = 1;
	if (max_leaf > rt::MAX_LEAF_PRIMS)
		max_leaf = rt::MAX_LEAF_PRIMS;
	if (n == 0)
		return;
	// sparse build: subtree over k primitives owns 2k node slots; compacted afterwards
	std::vector<rt::Node> sparse(2 * n + 2);
	std::vector<int> sparents(2 * n + 2, -1);
	std::memset(sparse.data(), 0, sparse.size() * sizeof(rt::Node));
	out.order.resize(n);
	std::iota(out.order.begin(), out.order.end(), 0u);
	Builder b;
	b.bmin = bmin, b.bmax = bmax;
	b.cen.resize(3 * n);
	Box root;
	root.reset();
	for (size_t i = 0; i < n; i++)
	{
		for (int a = 0; a < 3; a++)
			b.cen[3 * i + a] = 0.5f * (bmin[3 * i + a] + bmax[3 * i + a]);
		root.grow(&bmin[3 * i], &bmax[3 * i]);
	}
	b.max_leaf = max_leaf, b.depth_limit = depth_limit;
	b.nodes = sparse.data(), b.parents = sparents.data(), b.order = out.order.data();
	b.max_tasks = (int)std::max(1u, std::thread::hardware_concurrency());
	b.set_node(0, root, 0.0f, 0, (int)n);
	b.subdivide(0, 0, (uint32_t)n, 0, 2);
	out.max_depth = b.max_depth.load();

	// compaction in depth-first preorder (children pairs stay adjacent and 64-byte aligned: pairs start at even
	// indices because the root pair (0,1) does)
	out.nodes.reserve(2 * n + 4);
	out.parents.reserve(2 * n + 4);
	out.nodes.push_back(sparse[0]);
	out.parents.push_back(-1);
	rt::Node unused;
	std::memset(&unused, 0, sizeof(unused));
	unused.count = 0;
	out.nodes.push_back(unused);
	out.parents.push_back(-1);
	std::vector<int> stack;
	stack.push_back(0);
	while (!stack.empty())
	{
		const int dst = stack.back();
		stack.pop_back();
		if (out.nodes[dst].count >= 0)
			continue;
		const int src_left = out.nodes[dst].left_first;
		const int new_left = (int)out.nodes.size();
		out.nodes.push_back(sparse[src_left]);
		out.nodes.push_back(sparse[src_left + 1]);
		out.parents.push_back(dst);
		out.parents.push_back(dst);
		out.nodes[dst].left_first = new_left;
		stack.push_back(new_left + 1);
		stack.push_back(new_left);
	}
}

namespace
{
constexpr size_t BFS_TOP_NODES = 1024;
float node_area(const rt::Node &n)
{
	const float e0 = n.bmax[0] - n.bmin[0], e1 = n.bmax[1] - n.bmin[1], e2 = n.bmax[2] - n.bmin[2];
	return e0 * e1 + e0 * e2 + e1 * e2;
}
// children of the 4-wide node made from BVH2 node n2: open the inner child with the largest area until four
int pick_children(const Result &b, int n2, int kids[4])
{
	kids[0] = b.nodes[n2].left_first, kids[1] = b.nodes[n2].left_first + 1, kids[2] = kids[3] = -1;
	int nk = 2;
	while (nk < 4)
	{
		int best = -1;
		float best_area = -1.0f;
		for (int k = 0; k < nk; k++)
			if (b.nodes[kids[k]].count < 0 && node_area(b.nodes[kids[k]]) > best_area)
				best = k, best_area = node_area(b.nodes[kids[k]]);
		if (best < 0)
			break;
		const int l = b.nodes[kids[best]].left_first;
		kids[best] = l;
		kids[nk++] = l + 1;
	}
	return nk;
}
void fill_boxes(const Result &b, const int kids[4], int nk, rt::Node4 &nd)
{
	for (int k = 0; k < 4; k++)
	{
		if (k < nk)
		{
			const rt::Node &c = b.nodes[kids[k]];
			for (int a = 0; a < 3; a++)
				nd.lo[a][k] = c.bmin[a], nd.hi[a][k] = c.bmax[a];
			nd.src[k] = (uint32_t)kids[k];
			nd.entry[k] = 0;
		}
		else
		{
			for (int a = 0; a < 3; a++)
				nd.lo[a][k] = 1e34f, nd.hi[a][k] = 1e34f; // a far-away point: tmax > tmin never holds for it
			nd.src[k] = 0xFFFFFFFFu;
			nd.entry[k] = rt::ENTRY_EMPTY;
		}
	}
}
// depth-first: a subtree's nodes are contiguous (what the caches like below the top of the tree)
uint32_t collapse_node(const Result &b, int n2, bool tlas, std::vector<rt::Node4> &out)
{
	const uint32_t idx = (uint32_t)out.size();
	out.emplace_back();
	int kids[4];
	const int nk = pick_children(b, n2, kids);
	rt::Node4 nd;
	fill_boxes(b, kids, nk, nd);
	out[idx] = nd;
	for (int k = 0; k < nk; k++)
	{
		const rt::Node &c = b.nodes[kids[k]];
		uint32_t e;
		if (c.count >= 0)
			e = rt::make_entry(c.left_first, c.count, tlas);
		else
			e = rt::make_entry((int)collapse_node(b, kids[k], tlas, out), -1, tlas);
		out[idx].entry[k] = e;
	}
	return idx;
}
} // namespace

bool collapse4(const Result &bvh2, bool tlas, std::vector<rt::Node4> &out)
{
	out.clear();
	if (bvh2.nodes.empty() || bvh2.nodes[0].count >= 0)
		return false;
	out.reserve(bvh2.nodes.size() / 3 + 2);
	// The top of the tree is laid out breadth-first, so that a prefix of the array is "the top levels" (the traversal
	// kernels keep such a prefix in LDS); below it every subtree is depth-first.
	struct Item
	{
		int n2;
		uint32_t idx;
	};
	std::vector<Item> queue;
	out.emplace_back();
	queue.push_back({0, 0u});
	for (size_t head = 0; head < queue.size(); head++)
	{
		const Item it = queue[head];
		int kids[4];
		const int nk = pick_children(bvh2, it.n2, kids);
		rt::Node4 nd;
		fill_boxes(bvh2, kids, nk, nd);
		out[it.idx] = nd;
		for (int k = 0; k < nk; k++)
		{
			const rt::Node &c = bvh2.nodes[kids[k]];
			uint32_t e;
			if (c.count >= 0)
				e = rt::make_entry(c.left_first, c.count, tlas);
			else if (out.size() < BFS_TOP_NODES)
			{
				const uint32_t cidx = (uint32_t)out.size();
				out.emplace_back();
				queue.push_back({kids[k], cidx});
				e = rt::make_entry((int)cidx, -1, tlas);
			}
			else
				e = rt::make_entry((int)collapse_node(bvh2, kids[k], tlas, out), -1, tlas);
			out[it.idx].entry[k] = e;
		}
	}
	return true;
}

int stack_need4(const std::vector<rt::Node4> &nodes4)
{
	if (nodes4.empty())
		return 0;
	// iterative walk carrying the entries pending above each node (relative entries: inner = index into nodes4)
	std::vector<std::pair<uint32_t, int>> todo;
	todo.push_back({0u, 0});
	int worst = 0;
	while (!todo.empty())
	{
		const auto [idx, above] = todo.back();
		todo.pop_back();
		const rt::Node4 &n = nodes4[idx];
		int kids = 0;
		for (int k = 0; k < 4; k++)
			kids += n.entry[k] != rt::ENTRY_EMPTY;
		const int here = above + (kids > 0 ? kids - 1 : 0);
		worst = here > worst ? here : worst;
		for (int k = 0; k < 4; k++)
			if (n.entry[k] != rt::ENTRY_EMPTY && !(n.entry[k] & rt::ENTRY_LEAF))
				todo.push_back({n.entry[k] & rt::ENTRY_INDEX_MASK, here});
	}
	return worst;
}

} // namespace bvh
