// kernels.h — host-callable launchers of the wavefront stages (implemented in kernels.hip).
#pragma once
#include "rt_types.h"
#include <stddef.h>

namespace rtk
{

struct Params
{
	rt::SceneView sc;
	rt::WaveView wv;
	rt::CamView cam;
	rt::FrameView fr;
	uint32_t depth;		// pathLength of this wave
	uint32_t max_depth; // MAX_PATH_LENGTH
	uint32_t parity_no_jitter;
	uint32_t lds_first; // Node4 range every traversal workgroup keeps in LDS: the top of the largest BLAS
	uint32_t lds_count; // (0 = off, <= max_lds_nodes())
	uint32_t queue;		// which WaveCounters::work[] row this launch pulls its chunks from
	uint32_t group;		// chunks per XCD group (one row of tiles for the primary wave)
	uint32_t refill;	// incoherent waves: lanes that finish a ray pull the next one (persistent lanes)
	uint32_t textured;	// some material carries a texture / normal map: the shade kernel variant with the texture layers
};

enum GenMode
{
	GEN_BUFFER = 0, // rays come from the wave buffers (depth >= 1)
	GEN_PT = 1,		// generate pt primary rays   (CUDART generatePrimaryRay)
	GEN_PARITY = 2, // generate parity primary rays (EmbreeRT GenerateRay8 draw order)
	GEN_RANGED = 3	// rays from the wave buffers with per-ray (t_min, t_max) in the w components (rfwhip_trace_rays)
};

typedef void *stream_t; // hipStream_t

// capacity of the LDS top-of-tree cache the kernels were built with
uint32_t max_lds_nodes();

// device properties used for grid sizing
void set_device_cus(int cus);

// zero the per-render wave counters (ext/shadow/probe) and set ext[0] = primary_count
void launch_init_counters(rt::WaveCounters *c, uint32_t primary_count, stream_t s);
void launch_set_ext_count(rt::WaveCounters *c, uint32_t depth, uint32_t count, stream_t s);
void launch_rng_states(uint32_t *states, const uint32_t base_state[4], const uint32_t *jump_table,
					   uint32_t packets_per_sample, uint32_t spp, stream_t s);
void launch_extend(const Params &p, int gen, bool count, uint32_t max_items, stream_t s);
bool primary_packet_form(const Params &p, uint32_t max_items); // the pt primary wave of such a launch fills WaveView::hit0_done
void launch_shade_parity(const Params &p, bool count, uint32_t max_items, stream_t s);
uint32_t queue_pad(uint32_t max_items); // extra slots per queue and launch for the void entries of unfinished blocks
void launch_shade_pt(const Params &p, uint32_t max_items, stream_t s);
void launch_connect(const Params &p, bool count, uint32_t max_items, stream_t s);
// the connection wave of depth 0 in packet form: runs of the shadow queue sorted by the chosen light's bin (FrameView::shadow_bins),
// one wave-uniform occlusion traversal per 64 rays of the sorted order
void launch_shadow_packets(const Params &p, bool count, uint32_t max_items, stream_t s);
// the extension rays of pe.depth and the shadow rays of pa.depth (= pe.depth - 1) in one launch (both with persistent lanes)
void launch_trace_fused(const Params &pe, const Params &pa, bool count, uint32_t max_items, stream_t s);
void launch_resolve(const Params &p, stream_t s);
// rfwhip_kat: `function` (RFWHIP_KAT_*) on n records of 24 floats -> n records of 8 floats (device pointers)
void launch_kat(const Params &p, int function, const float *in, float *out, uint32_t n, stream_t s);
// out: local layout (local_rows x W) when full == 0, else full image (H x W; world must be 1)
void launch_present(const Params &p, rt::f4 *out, float scale, int full, stream_t s);
void launch_deinterleave(const rt::f4 *gathered, rt::f4 *out, uint32_t W, uint32_t H, uint32_t local_rows,
						 uint32_t world, stream_t s);
// bottom-up refit of one BLAS after its vertices changed: leaf_order[i] = original primitive of leaf slot i
// nodes / tri_verts are the scene-wide arrays (device entries carry absolute indices); node_base / tri_base locate the
// BLAS in them; parents are BLAS-relative
void launch_refit(rt::Node *nodes, uint32_t node_base, const int *parents, uint32_t node_count, rt::f4 *tri_verts,
				  uint32_t tri_base, const rt::f4 *verts, const uint32_t *indices, uint32_t tri_count, uint32_t *flags,
				  stream_t s);

// Flat instances (rfwhip_update): the triangles of a mesh whose one instance has the identity transform are reached without
// entering an instance, so every one of them carries that instance's index (w of its second leaf-ordered vertex).
void launch_stamp_instance(rt::f4 *tri_verts, uint32_t tri_count, uint32_t instance, stream_t s);

// after a refit of the BVH2 boxes: re-quantise the child boxes of the compressed 4-wide nodes of the same BLAS; src4 = four
// BLAS-relative BVH2 node indices per 4-wide node (Node4::src of the host's collapse)
void launch_refresh4(rt::Node4c *nodes4, const uint32_t *src4, uint32_t count4, const rt::Node *blas_nodes2, stream_t s);
// traversal-stack entries the packet form of the primary wave holds: the 64 lanes of one VGPR minus the sentinel at the
// bottom and two slots of slack above the top (its three-entry push writes unconditionally); the host uses the per-lane
// kernels for a scene whose trees could need more
constexpr uint32_t PACKET_STACK = 61;
// every compressed node once more with float planes (rt::Node4f), for the packet traversal's scalar fetches
void launch_expand4(const rt::Node4c *nodes4, rt::Node4f *out, uint32_t count4, stream_t s);
// BVH construction on the device (lbvh.hip), end to end in mesh-local arrays (node_base = tri_base = n4_base = 0):
//   nodes / parents / flags   2 n entries   BVH2 in the reference's layout, device entries, one triangle per leaf
//   nodes4 / src4             <= n / 4 n    the compressed 4-wide nodes the rays fetch, breadth-first, + their BVH2 sources
//   tri_verts                 3 n           triangles in leaf (depth-first) order, w of vertex 0 = primitive id
// Only the result record comes back (the call synchronises the stream).  Returns 0, or 1 when the mesh has fewer than two
// triangles (build it on the host), > 1 on an error.
struct DeviceBuildResult
{
	uint32_t node_count, node4_count, stack_need;
	float bmin[3], bmax[3];
};
size_t device_build_scratch_bytes(uint32_t tri_count);
int launch_device_build(const rt::f4 *verts, const uint32_t *indices, uint32_t tri_count, void *scratch, size_t scratch_bytes,
						rt::Node *nodes, int *parents, uint32_t *flags, rt::Node4c *nodes4, uint32_t *src4, rt::f4 *tri_verts,
						DeviceBuildResult *out, stream_t s);
// mesh-local entries -> scene-wide indices, in place (rfwhip_update places a device-built mesh with two copies and this)
void launch_rebase(rt::Node *nodes, uint32_t node_count, uint32_t node_base, rt::Node4c *nodes4, uint32_t n4_count, uint32_t n4_base,
				   uint32_t tri_base, stream_t s);
// linear-blend skinning on the device (geometry/gltf/mesh.cpp:31-45): verts/vnormals <- base * sum(w_k * M[j_k]);
// mats: joint_count column-major 4x4
void launch_skin_vertices(rt::f4 *verts, rt::f4 *vnormals, const rt::f4 *base_verts, const rt::f4 *base_normals,
						  const uint32_t *joints4, const rt::f4 *weights4, const float *mats, uint32_t joint_count,
						  uint32_t vertex_count, stream_t s);
// morph targets on the device (geometry/gltf/mesh.cpp:127-147): verts / vnormals <- base + sum_j weights[j] * target_j;
// tgt_pos / tgt_nrm: [target][vertex] float4, weights: device array of target_count floats
void launch_morph_vertices(rt::f4 *verts, rt::f4 *vnormals, const rt::f4 *base_verts, const rt::f4 *base_normals,
						   const rt::f4 *tgt_pos, const rt::f4 *tgt_nrm, const float *weights, uint32_t target_count,
						   uint32_t vertex_count, stream_t s);
// update_triangles() of the same file for the shading records: vertex normals of the three corners + face normal
void launch_skin_shade(rt::TriShade *shade, const rt::f4 *verts, const rt::f4 *vnormals, const uint32_t *indices,
					   uint32_t tri_count, stream_t s);

} // namespace rtk
