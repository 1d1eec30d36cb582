// rt_types.h — data layout of the HIP rendercore in HBM (shared by host code and kernels).
//
// Everything the kernels touch lives in a handful of flat device arrays, described by SceneView / WaveView.
// Layout decisions:
//   * BVH2 nodes keep the reference's 32-byte layout (RFW/system/bvh/include/bvh/bvh_node.h:23-28) so that the two
//     children of an inner node (always adjacent, pair-aligned to 64 B) arrive as four 16-byte loads.
//   * Triangles are stored TWICE: `tri_verts` in BVH-leaf order for intersection (3 x float4, w of vertex 0 carries
//     the original primitive id, so no index indirection at test time), and `tri_shade` (+ `tri_uv`) in mesh order for shading
//     (4 x float4 = 64 B = one memory sector, + 32 B of texture coordinates only textured hits read, instead of the
//     reference's 160-byte AoS Triangle, structs.h:24-60).
//   * Ray / hit / throughput records are arrays of float4 indexed by the (compacted) path index: one 16-byte
//     access per lane per attribute, 1 KiB per wave instruction.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) && !defined(RFWHIP_HOST_EMULATION)
#include <hip/hip_runtime.h>
#define RT_FN __host__ __device__ __forceinline__
#define RT_DEVICE_BUILD 1
#else
#define RT_FN inline
#endif

namespace rt
{

struct alignas(16) f4
{
	float x, y, z, w;
};
struct f3
{
	float x, y, z;
};

#ifndef RT_STRIP_ROWS
#define RT_STRIP_ROWS 8
#endif
constexpr uint32_t STRIP_ROWS = RT_STRIP_ROWS; // rows per ownership strip, a multiple of the 8-row tiles
// Strip ownership: strips are dealt to the ranks in periods of `world`, forwards in even periods and backwards in odd
// ones (0 1 .. w-1 | w-1 .. 1 0 | 0 1 ..): with the plain round-robin every strip of rank r lies 8 r rows below rank 0's,
// and where the cost of a row grows down the image (terrain under sky) the ranks' times were in rank order, 6 % apart
// at 8 ranks; the serpentine cancels that gradient.
RT_FN uint32_t strip_owner(uint32_t strip, uint32_t world)
{
	const uint32_t k = strip / world, pos = strip % world;
	return (k & 1u) ? world - 1u - pos : pos;
}
RT_FN uint32_t strip_of_local(uint32_t local_strip, uint32_t rank, uint32_t world)
{
	return local_strip * world + ((local_strip & 1u) ? world - 1u - rank : rank);
}
constexpr uint32_t TILE = 8;		  // 8x8 pixel tile = one wave64
// Determinant threshold of the triangle test (bvh_tree.cpp:174) in the triangle's own space; every leaf-ordered triangle carries
// its threshold in w of its third vertex: TRI_EPS, or TRI_EPS |det M| for a triangle the world tree holds in world space.
constexpr float TRI_EPS = 1e-6f;
constexpr uint32_t ENTRY_LEAF = 0x80000000u;
constexpr uint32_t ENTRY_TLAS = 0x40000000u;
constexpr uint32_t ENTRY_SENTINEL = 0xFFFFFFFFu;
constexpr uint32_t ENTRY_FIRST_MASK = 0x07FFFFFFu; // 27 bits: first primitive of a leaf
constexpr uint32_t ENTRY_INDEX_MASK = 0x3FFFFFFFu; // 30 bits: left child of an inner node
constexpr int MAX_LEAF_PRIMS = 8;				   // 3 bits of (count-1) in a leaf entry

// Stack entries per lane kept in LDS; deeper entries go to private memory.  Closest-hit rays stack deeper than occlusion
// rays (which leave at their first hit), and less LDS per workgroup means more resident waves for the latter: swept
// on MI355X, closest 16/12/8 -> 1787 / 1831 / 1808 Msamples/s, occlusion kernels alone 8 -> -9 % time.
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 12
#endif
#ifndef RT_LDS_STACK_ANY
#define RT_LDS_STACK_ANY 8
#endif
constexpr int LDS_STACK = RT_LDS_STACK;			// closest-hit kernels
constexpr int LDS_STACK_ANY = RT_LDS_STACK_ANY; // occlusion kernels
constexpr int LDS_STACK_MAX = LDS_STACK > LDS_STACK_ANY ? LDS_STACK : LDS_STACK_ANY;
constexpr int LDS_STACK_MIN = LDS_STACK < LDS_STACK_ANY ? LDS_STACK : LDS_STACK_ANY;
// Further entries in private memory (touched only by deep or pathological rays).  Capacity = LDS part + SPILL_STACK must
// cover the worst case of the trees rfwhip_update() accepts: a 4-wide node pushes up to 3 entries and spans >= 2 BVH2 levels
// when it does, so a path needs <= 1.5 entries per BVH2 level; the exact need of every tree (bvh::stack_need4) is checked
// against STACK_CAPACITY on the host, and a dropped entry is counted (WaveCounters::stack_overflow -> rfwhip_wait fails).
constexpr int SPILL_STACK = 64;
constexpr int STACK_CAPACITY = LDS_STACK_MIN + SPILL_STACK; // entries every traversal kernel can hold per ray

RT_FN uint32_t make_entry(int left_first, int count, bool tlas)
{
	const uint32_t t = tlas ? ENTRY_TLAS : 0u;
	if (count >= 0)
		return ENTRY_LEAF | t | ((uint32_t)(count - 1) << 27) | ((uint32_t)left_first & ENTRY_FIRST_MASK);
	return t | ((uint32_t)left_first & ENTRY_INDEX_MASK);
}

// One BVH2 node, byte-compatible with rfw::bvh::BVHNode (32 B).
struct alignas(16) Node
{
	float bmin[3];
	float bmax[3];
	int left_first;
	int count;
};

// Traversal node: a 4-wide node collapsed from the BVH2 (128 bytes = eight 16-byte loads).  The bounce waves are bound
// by the chain of dependent node fetches (~1 us each out of the Infinity Cache): a 4-wide node halves the
// chain.  Child boxes are stored SoA (one float4 per plane, lane k = child k); an unused slot has a degenerate far-away box
// (lo = hi = 1e34: the slab test's tmax > tmin never holds) and entry ENTRY_EMPTY.  src[k] is the BVH2 node (BLAS-relative) the child box was copied from — refit refreshes the boxes
// from the refitted BVH2.
constexpr uint32_t ENTRY_EMPTY = 0xFFFFFFFCu;
struct alignas(16) Node4
{
	float lo[3][4];
	float hi[3][4];
	uint32_t entry[4]; // ready-made stack entries (absolute indices on the device)
	uint32_t src[4];
};
static_assert(sizeof(Node4) == 128, "4-wide node");

// What the rays actually fetch: the 4-wide node COMPRESSED to 64 bytes = four 16-byte rows (two nodes per 128-byte line).
// The traversal kernels are bound by the number of divergent 16-byte lane-loads a CU's vector L1 can serve (one per clock,
// DESIGN.md §4), so a node costs what its rows cost: 4 instead of the 7 rows a float node needs.  Child boxes are quantised
// to 8 bits per plane in the frame of the node's own box (Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs
// through compressed wide BVHs", 2017): plane = org + q * 2^e per axis, q rounded OUTWARD, so every compressed box contains
// the float box it came from — the set of triangles a ray reaches, and therefore every hit, is unchanged.
//   row 0: org.x org.y org.z | scale.x            (scale = 2^e as a float: no decode in the loop)
//   row 1: entry[4]
//   row 2: qlo.x[4] qlo.y[4] qlo.z[4] qhi.x[4]   (one byte per child, byte k = child k)
//   row 3: qhi.y[4] qhi.z[4] | scale.y scale.z
// An unused slot has an inverted box (qlo = 255, qhi = 0) and ENTRY_EMPTY: never hit.
struct alignas(16) Node4c
{
	float org[3];
	float scale_x;
	uint32_t entry[4];
	uint32_t qlo[3];
	uint32_t qhi[3];
	float scale_y, scale_z;
};
static_assert(sizeof(Node4c) == 64, "compressed 4-wide node");

// The same node once more with FLOAT planes — exactly the planes the compressed node decodes to (plane = fma(q, scale, org)):
// what the wave-uniform ("packet") traversal of the coherent waves fetches through SCALAR loads (kernels.hip: trace_packet).
// A wave that walks one node for all its lanes has the node in SGPRs; byte -> float conversions of wave-uniform data would
// cost a full VALU issue each, so they are done once per node and update (k_expand4) instead of once per visit.  Row order:
// which of a child's two planes per axis is the entry plane depends only on the sign of the ray direction, so the kernel
// reads "near" and "far" rows by choosing the row offset per wave (lo rows at 0 / 16 / 32, hi rows at 48 / 64 / 80).
// An unused slot: lo = +1e30, hi = -1e30 (never hit).
struct alignas(16) Node4f
{
	float lo[3][4];
	float hi[3][4];
	uint32_t entry[4];
	uint32_t pad[4];
};
static_assert(sizeof(Node4f) == 128, "float 4-wide node");

// Quantise the (up to four) child boxes lo[axis][child] .. hi[axis][child] into n (entries untouched).  Shared by the host
// (upload) and the device (after a refit), so both produce the same bytes.
RT_FN void pack_boxes4c(Node4c &n, const float lo[3][4], const float hi[3][4], const bool valid[4])
{
	for (int a = 0; a < 3; a++)
	{
		float mn = 3.0e38f, mx = -3.0e38f;
		for (int k = 0; k < 4; k++)
			if (valid[k])
				mn = lo[a][k] < mn ? lo[a][k] : mn, mx = hi[a][k] > mx ? hi[a][k] : mx;
		if (!(mx >= mn)) // no valid child
			mn = 0.0f, mx = 0.0f;
		const float ext = mx - mn;
		int e = -100;
		if (ext > 0.0f)
		{
			(void)frexpf(ext * (1.0f / 254.0f), &e); // ext / 254 = m * 2^e with m in [0.5, 1)  =>  254 * 2^e >= ext
			e = e < -100 ? -100 : (e > 120 ? 120 : e);
		}
		const float scale = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
		n.org[a] = mn;
		(a == 0 ? n.scale_x : (a == 1 ? n.scale_y : n.scale_z)) = scale;
		uint32_t ql4 = 0u, qh4 = 0u;
		for (int k = 0; k < 4; k++)
		{
			uint32_t ql = 255u, qh = 0u; // inverted: never hit
			if (valid[k])
			{
				float fl = floorf((lo[a][k] - mn) * inv), fh = ceilf((hi[a][k] - mn) * inv);
				fl = fl < 0.0f ? 0.0f : (fl > 255.0f ? 255.0f : fl), fh = fh < 0.0f ? 0.0f : (fh > 255.0f ? 255.0f : fh);
				// outward, whatever the rounding of the two lines above did
				while (fl > 0.0f && fmaf(fl, scale, mn) > lo[a][k])
					fl -= 1.0f;
				while (fh < 255.0f && fmaf(fh, scale, mn) < hi[a][k])
					fh += 1.0f;
				ql = (uint32_t)fl, qh = (uint32_t)fh;
			}
			ql4 |= ql << (8 * k), qh4 |= qh << (8 * k);
		}
		n.qlo[a] = ql4, n.qhi[a] = qh4;
	}
}

// Per-instance record (set_instance): inverse transform for rays, normal matrix for shading, BLAS location.
struct alignas(16) Instance
{
	float inv[12];		 // rows 0..2 of M^-1 (row-major 3x4)
	float nrm[12];		 // normal matrix, 3 columns padded to float4 (column-major like glm::mat3)
	uint32_t root_entry; // stack entry of the BLAS root (absolute indices, like every entry on the device)
	uint32_t node_base;	 // first node of the BLAS in SceneView::nodes
	uint32_t tri_base;	 // first leaf-ordered triangle of the BLAS in SceneView::tri_verts
	uint32_t shade_base; // first mesh-ordered shading record in SceneView::tri_shade
};

// Shading record per triangle (mesh order): 64 B = ONE 64-byte memory sector, never straddling two (round 6: the 96-byte record
// of rounds 1-5 always cost the gather two sectors = 128 B of HBM traffic for 96 B used; the texture coordinates, which only a
// textured material reads, moved into a record of their own).
struct alignas(16) TriShade
{
	f4 n0; // vN0.xyz, Nx
	f4 n1; // vN1.xyz, Ny
	f4 n2; // vN2.xyz, Nz
	f4 ex; // area, LOD, bits(lightTriIdx), bits(material)
};
static_assert(sizeof(TriShade) == 64, "shading record = one memory sector");
// Texture coordinates per triangle (mesh order), 32 B: read for hits on materials with a texture only.
struct alignas(16) TriUV
{
	f4 tu; // u0,u1,u2, 0
	f4 tv; // v0,v1,v2, 0
};
static_assert(sizeof(TriUV) == 32, "texture-coordinate record");

struct TexDesc
{
	uint32_t type, width, height, texelCount;
	uint32_t offset; // first texel in tex_u32 (UINT) or tex_f4 (FLOAT4)
	uint32_t pad[3];
};

// 192-byte material exactly as handed over (structs.h:85-127).
struct alignas(16) MaterialRec
{
	uint16_t diffuse[3];
	uint16_t transmittance[3];
	uint32_t flags;
	uint32_t parameters[4];
	struct
	{
		int16_t width, height;
		uint16_t uscale, vscale, uoffs, voffs;
		uint32_t addr;
	} map[10];
};
static_assert(sizeof(MaterialRec) == 192, "material layout");

struct AreaLight
{
	float position[3], energy, normal[3], area, radiance[3];
	int dummy0;
	float vertex0[3];
	int triIdx;
	float vertex1[3];
	int instIdx;
	float vertex2[3];
	int dummy1;
};
struct PointLight
{
	float position[3], energy, radiance[3];
	int dummy;
};
struct SpotLight
{
	float position[3], cosInner, radiance[3], cosOuter, direction[3], energy;
};
struct DirectionalLight
{
	float direction[3], energy, radiance[3];
	int dummy;
};

struct SceneView
{
	const Node *nodes;		 // ONE array: all BLAS nodes, then the TLAS nodes.  Node::left_first holds the node's
							 // ready-made stack entry with ABSOLUTE indices (node index into this array, leaf-ordered
							 // triangle index into tri_verts): a traversal step is base + 32-bit offset, no per-lane
							 // base pointers
	const Node4c *nodes4;	 // traversal form: all BLAS 4-wide nodes (compressed), then the TLAS's (absolute entries)
	const Node4f *nodes4f;	 // the same nodes with float planes (scalar-fetched by the packet traversal of the coherent waves)
	const f4 *tri_verts;	 // 3 per leaf-ordered triangle
	const TriShade *tri_shade;
	const TriUV *tri_uv;		 // same index as tri_shade
	const uint32_t *tlas_prims; // instance index per TLAS leaf slot
	const Instance *instances;
	uint32_t tlas_root_entry;
	uint32_t instance_count;
	const MaterialRec *materials;
	uint32_t material_count;
	const TexDesc *textures;
	uint32_t texture_count;
	const uint32_t *tex_u32;
	const f4 *tex_f4;
	const f4 *sky; // rgb + pad per texel
	uint32_t sky_w, sky_h;
	const AreaLight *area;
	const PointLight *point;
	const SpotLight *spot;
	const DirectionalLight *dir;
	uint32_t n_area, n_point, n_spot, n_dir;
};

// Camera as the kernels need it (EmbreeRT/src/Ray.cpp:3-14: right = p2-p1, up = p3-p1).
struct CamView
{
	f3 pos, p1, right, up;
	float aperture, spread_angle, clamp_value;
	const uint32_t *blue_noise; // 5 x 65536 table of the blue-noise sampler (tools.h:163-181) or null: hash RNG
};
constexpr uint32_t BLUE_NOISE_WORDS = 5u * 65536u;

// Division of a 31-bit index by a per-frame constant without the ~30-instruction udiv sequence (every path maps its slot to a
// pixel in the primary and in every shade kernel): Granlund-Montgomery, m = ceil(2^(31 + L) / d) with L = ceil(log2 d) fits 32
// bits and floor(n * m / 2^(31 + L)) == n / d for every n < 2^31 (RFWHIP_KAT_FASTDIV checks it on both builds).
struct FastDiv
{
	uint32_t m, s; // m == 0: the divisor is 1
};
RT_FN FastDiv make_fastdiv(uint32_t d)
{
	FastDiv f;
	f.m = 0, f.s = 0;
	if (d <= 1u)
		return f;
	uint32_t L = 0;
	while (L < 32u && (1ull << L) < (unsigned long long)d)
		L++;
	f.m = (uint32_t)(((1ull << (31u + L)) + d - 1u) / d);
	f.s = L - 1u;
	return f;
}
RT_FN uint32_t fast_div(uint32_t n, const FastDiv f)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return f.m ? __umulhi(n, f.m) >> f.s : n;
#else
	return f.m ? (uint32_t)(((unsigned long long)n * f.m) >> 32) >> f.s : n;
#endif
}

// Which image rows this rank owns, and how path slots map to pixels.
struct FrameView
{
	uint32_t W, H;		   // full image
	float inv_w, inv_h;	   // 1.0f / W, 1.0f / H — divided once on the host (correctly rounded, like the v_div sequence every ray generation paid for)
	uint32_t local_rows;   // padded rows on this rank (multiple of STRIP_ROWS)
	uint32_t tiles_x;	   // ceil(W / 8)
	uint32_t slots;		   // tiles_x*8 * local_rows  = path slots per sample
	uint32_t rank, world;
	uint32_t spp;		   // samples in this batch
	uint32_t sample_base;  // index of the first sample of the batch
	uint32_t probe_pixel;  // y*W + x
	uint32_t sgroup_log2;  // log2 of the sample group g (rt_core.h: slot layout): a wave's 64 slots = 64/g pixels x g samples
	uint32_t shadow_bins;  // b > 0: the shadow rays of the primary vertices carry the chosen light's bin in b bits above their path slot
						   // (bits 31 - b .. 30 of the slot word; the sub-batch's slots fit below; b = SHADOW_BIN_BITS = 4 up to 2^27
						   // slots, 3 / 2 / 1 for larger ones): k_shadow_packet sorts a run's rays by it
	FastDiv div_tiles_x;   // n / tiles_x
	FastDiv div_group;	   // n / (slots << sgroup_log2): which sample group a slot belongs to
};

// A pixel's index within its 8x8 tile <-> its coordinates in the tile: Z order (bits x0 y0 x1 y1 x2 y2), so that 2^k consecutive
// pixels are a square or a 2:1 block instead of a strip of a row — what a wave of 8 samples x 8 pixels, a run of the packet
// kernel or a queue block of a tile's bounce rays then share is a neighbourhood (8 spp per step: 3.85 -> 3.76 ms; nothing else
// moves).
RT_FN uint32_t tile_pix(uint32_t x, uint32_t y) // x, y in 0..7
{
	return (x & 1u) | ((y & 1u) << 1) | ((x & 2u) << 1) | ((y & 2u) << 2) | ((x & 4u) << 2) | ((y & 4u) << 3);
}
RT_FN uint32_t tile_pix_x(uint32_t pix) { return (pix & 1u) | ((pix >> 1) & 2u) | ((pix >> 2) & 4u); }
RT_FN uint32_t tile_pix_y(uint32_t pix) { return ((pix >> 1) & 1u) | ((pix >> 2) & 2u) | ((pix >> 3) & 4u); }

// Slot layout (rt_core.h, "pixel <-> path-slot mapping"): slot of sample s (within the batch) of pixel `pix` (0..63, tile_pix) of tile `tile`
RT_FN unsigned long long pixel_to_slot(const FrameView &fr, uint32_t tile, uint32_t pix, uint32_t s)
{
	const uint32_t gl = fr.sgroup_log2;
	const uint32_t sgroup = s >> gl, si = s & ((1u << gl) - 1u);
	return ((unsigned long long)sgroup * fr.slots << gl) + ((unsigned long long)tile << (6u + gl)) + (pix << gl) + si;
}

// Device counters, zeroed per render call.  ext[d] = number of paths entering depth d (ext[0] is set by the host),
// shadow[d] = shadow rays emitted by the shade stage of depth d.
constexpr int MAX_DEPTH_SLOTS = 16;
constexpr int WORK_QUEUES = 3 * MAX_DEPTH_SLOTS + 4; // one per traversal launch of a render call
// The extension / shadow queues are filled in BLOCKS: a wave of the shade kernel reserves QUEUE_BLOCK slots with one atomic
// and hands them out wave-locally (a device-scope atomic on one address completes about every 7 ns on this part: one per
// shade call and queue was 1 M atomics = 7 ms per 32-spp launch — the whole shade kernel).  What a wave has left of its last
// block when the kernel ends is filled with VOID entries (slot bits of the origin record all ones, extension and shadow rays alike), which
// the consumers skip; the hit record of a void extension ray carries HIT_VOID so that the next shade stage skips it too.
// ext_n / shadow_n are the queue lengths including void entries (what the consumers iterate over), ext / shadow the rays.
#ifndef RT_QUEUE_BLOCK
#define RT_QUEUE_BLOCK 256u
#endif
constexpr uint32_t QUEUE_BLOCK = RT_QUEUE_BLOCK;
constexpr uint32_t SHADOW_BIN_BITS = 4u, SHADOW_BINS = 1u << SHADOW_BIN_BITS; // at most: lights 0..14 have bins of their own, the others share the last
RT_FN uint32_t shadow_slot_bits(uint32_t bin_bits) { return 31u - bin_bits; }		 // (bit 31 stays clear: a slot word is never RAY_VOID)
constexpr uint32_t RAY_VOID = 0xFFFFFFFFu; // org.w / sh_org.w of a void queue entry (slots are < 2^31)
constexpr int HIT_VOID = -2;			   // hit.prim of a void entry (-1: miss)
constexpr int HIT_MISS_SHADED = -3;	   // primary wave, packet form: a miss whose sky term is already in its slot (read back as -1)
struct WaveCounters
{
	uint32_t ext[MAX_DEPTH_SLOTS];
	uint32_t shadow[MAX_DEPTH_SLOTS];
	uint32_t ext_n[MAX_DEPTH_SLOTS];
	uint32_t shadow_n[MAX_DEPTH_SLOTS];
	uint32_t work[WORK_QUEUES][8]; // per launch, per XCD: head of the chunk queue the persistent workgroups pull from
	unsigned long long rays_extend, rays_shadow, inner_extend, tris_extend, inner_shadow, tris_shadow, shaded, samples;
	unsigned long long lds_extend, lds_shadow; // node visits served by the LDS top-of-tree cache
	uint32_t probe_inst, probe_prim;
	float probe_dist;
	uint32_t probe_valid;
	uint32_t stack_overflow; // traversal-stack entries dropped (must stay 0: rfwhip_update bounds the trees; rfwhip_wait fails otherwise)
	uint32_t ext_timed;		 // extend-stage launches folded into ext_ticks so far
	uint32_t pad_[2];
	// Device-side clock of the extend stage (what a kernel trace reports as the kernel's duration): first workgroup in /
	// last workgroup out of the launch of depth d, in ticks of the constant 100 MHz counter (s_memrealtime); folded into
	// ext_ticks when the counters are re-armed for the next call, or when the host reads them.
	unsigned long long t_first[MAX_DEPTH_SLOTS], t_last[MAX_DEPTH_SLOTS];
	unsigned long long ext_ticks;
	// k_trace_fused: workgroup time (100 MHz ticks, summed over the launch's workgroups, never reset: the host takes differences) spent
	// on the extension rays of depth d [0] and on the shadow rays of depth d - 1 [1] — the shares the host splits the launch's
	// duration by, so that RenderStats::shadowTime is the shadow rays' although they share a kernel with the extension rays
	unsigned long long fused_ticks[MAX_DEPTH_SLOTS][2];
	// k_shadow_packet: runs sorted and light bins that occurred in them, summed over the launch (never reset: the host takes
	// differences) — bins per run is what tells a scene whose first vertices agree about their lights from one where they do not
	unsigned long long sp_runs, sp_bins;
};

// The wavefront state in HBM.
struct WaveView
{
	f4 *org[2];	 // origin.xyz, bits(slot << 1 | specular)
	f4 *dir[2];	 // direction.xyz, bits(packed normal of the previous vertex)
	f4 *thr[2];	 // throughput.rgb, pending bsdf pdf
	f4 *hit;	 // t, u, v, bits(prim)       (depth >= 1)
	int *hit_inst;
	f4 *hit0;	 // same, primary wave (kept for read_primary_hits / probe)
	int *hit0_inst;
	// one byte per 64-slot group of the primary wave, written by the packet kernel: 1 = every path of the group is finished (all
	// missed: the sky term is in `rad`), the shade kernel's scan passes the group by without reading its records.  Null: no flags
	unsigned char *hit0_done;
	f4 *sh_org;	 // shadow ray origin.xyz, bits(slot)
	f4 *sh_dir;	 // direction.xyz, tmax
	f4 *sh_rad;	 // contribution.rgb
	f4 *rad;	 // per-slot radiance of the batch (rgb, alpha): written by the shade stages
	f4 *rad_nee; // per-slot radiance added by the connection waves (they overlap the next depth's extend / shade stages, so
				 // they must not read-modify-write the same words); null: connections add into `rad`
	f4 *acc;	 // per local pixel accumulator (row-major local_rows x W)
	const uint32_t *packet_rng; // parity integrator: xor128 state per (sample, packet), 4 x u32
	WaveCounters *counters;
};

} // namespace rt
