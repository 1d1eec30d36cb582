// bvh_build.h — host-side BVH2 construction for the HIP rendercore.
#pragma once
#include "rt_types.h"
#include <stddef.h>
#include <vector>

namespace bvh
{

struct Result
{
	std::vector<rt::Node> nodes;  // node 0 = root, node 1 unused, children always at (left_first, left_first + 1)
	std::vector<uint32_t> order;  // leaf slot -> original primitive
	std::vector<int> parents;	  // parent node of every node (-1 for the root / unused slot)
	int max_depth = 0;
};

// Binned-SAH top-down build over primitive boxes (bmin/bmax: n x 3 floats).  Leaves hold 1..max_leaf primitives;
// the tree never gets deeper than depth_limit (falls back to median splits when the budget runs short).
// The node layout and the "children boxes grown by 1e-5" rule follow RFW/system/bvh/include/bvh/bvh_node.h:23-28,
// 215-226; the split search itself is our own (16 centroid bins per axis, full SAH with traversal cost), because
// the reference delegates construction to an external crate (RFW/system/bvh/src/bvh_tree.cpp:74-95).
void build(const float *bmin, const float *bmax, size_t n, int max_leaf, int depth_limit, Result &out);

// Collapse the binary tree into 4-wide nodes (greedy: open the inner child with the largest surface area until four
// children).  Entries are relative: inner = index into `out`, leaf = make_entry(first, count) of the BVH2 leaf; `tlas`
// sets ENTRY_TLAS on every entry.  Returns false when the root itself is a leaf (no 4-wide node needed).
bool collapse4(const Result &bvh2, bool tlas, std::vector<rt::Node4> &out);

// Worst-case number of traversal-stack entries a ray can hold while walking this 4-wide tree: the maximum over root-to-leaf
// paths of the sum of (children - 1) over the nodes on the path (every visit pushes at most the children it does not
// descend into; a leaf entry in hand is not on the stack).  0 for an empty tree.
int stack_need4(const std::vector<rt::Node4> &nodes4);

} // namespace bvh
