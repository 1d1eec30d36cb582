// rfwhip_api.cpp — the C ABI of include/rfwhip.h: scene residency in HBM, BVH construction, wave scheduling.
//
// Host-side responsibilities (the kernels are in kernels.hip):
//   * copy every borrowed host buffer into device memory inside the call (context.h ownership rules),
//   * build one BVH2 per mesh (bvh_build.cpp), concatenate all meshes into the flat arrays of rt::SceneView,
//     build the top-level BVH over instances in update(), refit instead of rebuild when a mesh keeps its counts,
//   * enqueue the wavefront stages of a frame on the context's HIP stream without any host read-back between
//     bounces (contrast CUDART/src/Context.cpp:98,145), gather stage timings with hipEvents.
#include "rfwhip.h"

#include "bvh_build.h"
#include "internal.h"
#include "kernels.h"
#include "rt_types.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#if !defined(RFWHIP_HOST_EMULATION)
#include <hip/hip_runtime.h>
#endif

using rt::f4;

// =================================================================================================================
// error reporting
// =================================================================================================================
static thread_local char g_error[1024] = "";
static int set_error(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
	return code;
}
int rfwhip_internal_set_error(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
	return code;
}
extern "C" const char *rfwhip_last_error(void) { return g_error; }
extern "C" const char *rfwhip_version(void)
{
#if defined(RFWHIP_HOST_EMULATION)
	return "rfwhip 0.1 (HOST EMULATION BUILD — tests only)";
#else
	return "rfwhip 0.1 (gfx950 HIP)";
#endif
}

// =================================================================================================================
// device memory / stream abstraction
// =================================================================================================================
namespace dm
{
#if !defined(RFWHIP_HOST_EMULATION)
#define DM_CHECK(x)                                                                                      \
	do                                                                                                   \
	{                                                                                                    \
		hipError_t e_ = (x);                                                                             \
		if (e_ != hipSuccess)                                                                            \
			return set_error(RFWHIP_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

static int init(int device, int *cus)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return set_error(RFWHIP_ERR_NO_DEVICE, "no HIP device visible: the rendercore has no CPU path");
	if (device < 0 || device >= n)
		return set_error(RFWHIP_ERR_NO_DEVICE, "device ordinal %d out of range (%d devices)", device, n);
	DM_CHECK(hipSetDevice(device));
	hipDeviceProp_t prop;
	DM_CHECK(hipGetDeviceProperties(&prop, device));
	*cus = prop.multiProcessorCount;
	return 0;
}
static int use(int device)
{
	DM_CHECK(hipSetDevice(device));
	return 0;
}
static int alloc(void **p, size_t bytes)
{
	DM_CHECK(hipMalloc(p, bytes ? bytes : 16));
	return 0;
}
static void release(void *p)
{
	if (p)
		(void)hipFree(p);
}
static void mem_info(size_t *free_b, size_t *total_b)
{
	if (hipMemGetInfo(free_b, total_b) != hipSuccess)
		*free_b = *total_b = ~(size_t)0;
}
static int h2d(void *d, const void *h, size_t n, void *s)
{
	if (n)
		DM_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
	return 0;
}
static int d2h(void *h, const void *d, size_t n, void *s)
{
	if (n)
	{
		DM_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s));
		DM_CHECK(hipStreamSynchronize((hipStream_t)s));
	}
	return 0;
}
static int d2d(void *dst, const void *src, size_t n, void *s)
{
	if (n)
		DM_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)s));
	return 0;
}
static int zero(void *d, size_t n, void *s)
{
	if (n)
		DM_CHECK(hipMemsetAsync(d, 0, n, (hipStream_t)s));
	return 0;
}
static int sync(void *s)
{
	DM_CHECK(hipStreamSynchronize((hipStream_t)s));
	return 0;
}
static int stream_create(void **s)
{
	hipStream_t st;
	DM_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	*s = st;
	return 0;
}
static void stream_destroy(void *s)
{
	if (s)
		(void)hipStreamDestroy((hipStream_t)s);
}
static int last_launch_error()
{
	DM_CHECK(hipGetLastError());
	return 0;
}
typedef hipEvent_t event_t;
static int event_create(event_t *e)
{
	DM_CHECK(hipEventCreate(e));
	return 0;
}
static void event_destroy(event_t e) { (void)hipEventDestroy(e); }
static int event_record(event_t e, void *s)
{
	DM_CHECK(hipEventRecord(e, (hipStream_t)s));
	return 0;
}
static float event_ms(event_t a, event_t b)
{
	float ms = 0;
	if (hipEventElapsedTime(&ms, a, b) != hipSuccess)
		return 0.0f;
	return ms;
}
static int stream_wait_event(void *s, event_t e)
{
	DM_CHECK(hipStreamWaitEvent((hipStream_t)s, e, 0));
	return 0;
}
static int event_sync(event_t e)
{
	DM_CHECK(hipEventSynchronize(e));
	return 0;
}
#else
// ---- host emulation (tests/emu): plain heap memory, "streams" are immediate ----
static int init(int, int *cus)
{
	*cus = 1;
	return 0;
}
static int use(int) { return 0; }
static int alloc(void **p, size_t bytes)
{
	*p = calloc(bytes ? bytes : 16, 1);
	return *p ? 0 : set_error(RFWHIP_ERR_HIP, "out of memory");
}
static void release(void *p) { free(p); }
static void mem_info(size_t *free_b, size_t *total_b) { *free_b = *total_b = ~(size_t)0; }
static int h2d(void *d, const void *h, size_t n, void *)
{
	memcpy(d, h, n);
	return 0;
}
static int d2h(void *h, const void *d, size_t n, void *)
{
	memcpy(h, d, n);
	return 0;
}
static int d2d(void *dst, const void *src, size_t n, void *)
{
	memmove(dst, src, n);
	return 0;
}
static int zero(void *d, size_t n, void *)
{
	memset(d, 0, n);
	return 0;
}
static int sync(void *) { return 0; }
static int stream_create(void **s)
{
	*s = nullptr;
	return 0;
}
static void stream_destroy(void *) {}
static int last_launch_error() { return 0; }
typedef std::chrono::steady_clock::time_point event_t;
static int event_create(event_t *) { return 0; }
static void event_destroy(event_t) {}
static int event_record(event_t &e, void *)
{
	e = std::chrono::steady_clock::now();
	return 0;
}
static float event_ms(event_t a, event_t b) { return std::chrono::duration<float, std::milli>(b - a).count(); }
static int stream_wait_event(void *, event_t) { return 0; }
static int event_sync(event_t) { return 0; }
#endif
} // namespace dm

#define RF_TRY(x)            \
	do                       \
	{                        \
		const int rc_ = (x); \
		if (rc_)             \
			return rc_;      \
	} while (0)

struct DevBuf
{
	void *p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes)
	{
		if (bytes <= cap && p)
			return 0;
		dm::release(p);
		p = nullptr, cap = 0;
		// grow with headroom so animation / resize do not reallocate every call
		const size_t want = bytes + bytes / 8 + 256;
		RF_TRY(dm::alloc(&p, want));
		cap = want;
		return 0;
	}
	// exactly `bytes` (the path-state buffers: tens of GB, no headroom); an existing larger allocation is kept
	int ensure_exact(size_t bytes)
	{
		if (bytes <= cap && p)
			return 0;
		dm::release(p);
		p = nullptr, cap = 0;
		if (dm::alloc(&p, bytes + 256))
		{
			p = nullptr;
			return RFWHIP_ERR_HIP;
		}
		cap = bytes + 256;
		return 0;
	}
	void free_()
	{
		dm::release(p);
		p = nullptr, cap = 0;
	}
	template <typename T> T *as() const { return (T *)p; }
};

// =================================================================================================================
// context
// =================================================================================================================
struct MeshRec
{
	bool used = false;
	size_t vertexCount = 0, triCount = 0;
	bool indexed = false;
	bvh::Result bvh;				   // host copy of the topology as built
	std::vector<rt::Node4> n4;		   // 4-wide traversal nodes collapsed from it (relative entries)
	uint32_t n4_base = 0;
	std::vector<f4> leaf_verts;		   // 3 per leaf slot (host staging for (re)upload)
	std::vector<rt::TriShade> shade;   // mesh order
	std::vector<rt::TriUV> uv;		   // mesh order (texture coordinates: rt_types.h)
	float bounds_min[3] = {0, 0, 0}, bounds_max[3] = {0, 0, 0};
	DevBuf d_verts, d_indices;		   // raw vertices / indices (kept for refit)
	DevBuf d_parents, d_flags;		   // refit helpers
	uint32_t node_base = 0, tri_base = 0, shade_base = 0;
	// built on the device (builder=device): no host copy of the tree; the mesh-local device arrays below are what
	// rfwhip_update() places (device-to-device copies + an entry rebase)
	bool device_built = false;
	bool built = false; // the last (re)build of this mesh succeeded: the counts below describe its tree
	DevBuf d_b_nodes, d_b_nodes4, d_b_src, d_b_tri_verts;
	uint32_t node_count2 = 0, n4_count = 0; // BVH2 nodes / 4-wide nodes of the mesh, whoever built them
	int stack_need = 0;		   // worst-case traversal-stack entries of the 4-wide tree (bvh::stack_need4)
	uint32_t max_material = 0; // largest material id any triangle refers to
	bool resident = false; // placed in the global arrays by the last update()
	bool dirty = true;	   // host staging newer than the global arrays
	bool refit_pending = false;
	uint32_t generation = 0; // counts the (re)builds of this mesh (the world tree's cache key)
	uint32_t refits = 0;	 // same-topology rfwhip_set_mesh calls since the last build: an animated mesh (never written into the world tree)
	// device skinning (rfwhip_set_mesh_skin / rfwhip_pose_mesh)
	bool skinned = false, posed = false;
	DevBuf d_base_verts, d_base_normals, d_joints, d_weights, d_vnormals, d_joint_mats;
	uint32_t joint_count = 0;
	// device morph targets (rfwhip_set_mesh_morph / rfwhip_morph_mesh); shares d_base_verts / d_base_normals / d_vnormals
	bool morphed = false;
	DevBuf d_tgt_pos, d_tgt_nrm, d_morph_weights;
	uint32_t target_count = 0;
};

struct InstRec
{
	bool used = false;
	size_t mesh = 0;
	float transform[16];
	float normal[9];
	// world-tree membership (prepare_world): an instance whose matrix changed in one of the last few updates is being animated
	// through its matrix and keeps the two-level walk — writing it out in world space would rebuild the world tree per frame
	float prev_transform[16];
	size_t prev_mesh = 0;
	uint32_t prev_generation = 0;
	bool has_prev = false;
	uint32_t moving = 0; // updates left before it may (re)join the world tree
	uint32_t moves = 0;	 // times it has started to move out of stillness: the waiting time doubles with each (an object that moves now
						 // and then would otherwise cost two rebuilds of the world tree — seconds for millions of triangles — per episode)
};

// The WORLD TREE (rfwhip_update): the triangles of the static instances written out in world space under ONE tree.
struct WorldRec
{
	bool valid = false;
	std::vector<uint32_t> key;		   // what it was built from: (instance, mesh, mesh generation, transform) per member
	std::vector<uint8_t> member;	   // per instance: its triangles are in the tree
	std::vector<uint8_t> mesh_all;	   // per mesh: EVERY live instance of it is a member (no ray walks the mesh's own tree)
	std::vector<uint8_t> mesh_member;  // per mesh: at least one of its instances is (an instance that moves keeps the two-level walk of
									   // the same mesh: see fill_params for what that means for the LDS top-of-tree choice)
	std::vector<rt::Node4> n4;		   // relative entries, like MeshRec::n4
	std::vector<f4> leaf_verts;		   // 3 per leaf slot: v0.w = primitive id (mesh order), v1.w = instance index
	uint32_t root_first = 0, root_count = 0; // the root when it is a leaf (n4 empty)
	size_t tris = 0;
	int stack_need = 0;
	float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
	uint32_t n4_base = 0, tri_base = 0;
	bool resident = false; // placed in the scene-wide arrays by the last layout
};

enum KernelFamily
{
	KF_GENERATE = 0,
	KF_EXTEND = 1,
	KF_SHADE = 2,
	KF_CONNECT = 3,
	KF_FINALIZE = 4,
	KF_REFIT = 5,
	KF_COUNT = 6
};

struct TimedSpan
{
	dm::event_t a, b;
	bool fused = false; // a k_trace_fused launch: extension rays of `depth` + shadow rays of depth - 1 (split by the device's tick sums)
	int family;
	int depth; // wave depth for extend spans, -1 otherwise
};

struct rfwhip_context
{
	int device = 0, rank = 0, world = 1, cus = 256;
	void *stream = nullptr; // main stream: scene uploads, refits, prologue, resolve, presents
	static constexpr int MAX_SUB = 8;
	static constexpr int MAX_RING = 4;
	// every sub-batch of a render call runs on its own stream, its connection (shadow) waves on a second one beside it
	void *sub_stream[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	void *conn_stream[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	DevBuf d_counters_sub[MAX_SUB]; // [0] is d_counters' alias slot (unused), 1.. are the extra sub-batches' counters
	dm::event_t ev_prologue, ev_sub_done[MAX_SUB];
	dm::event_t ev_resolve[MAX_RING];								// resolve of the last call that used radiance buffer set 0 / 1
	dm::event_t ev_shade[MAX_SUB][rt::MAX_DEPTH_SLOTS];		// shade stage of depth d enqueued (its connections may start)
	dm::event_t ev_conn[MAX_SUB][rt::MAX_DEPTH_SLOTS];		// connection wave of depth d enqueued
	dm::event_t ev_conn_last[MAX_SUB];						// everything of the call on the connection stream
	bool conn_used[MAX_SUB] = {false, false, false, false, false, false, false, false};
	bool resolve_recorded[MAX_RING] = {false, false, false, false};
	uint32_t call_slot = 0;	  // which set of the ring the next render call uses
	int ring = 3;			  // single-sub-batch calls rotate through this many sets of wave buffers / streams / counters
							  // (3: with the main stream that is four streams = the HIP runtime's four hardware queues; a
							  // fourth set shares a queue with another chain: 1-spp frames 1.43 vs 1.67 ms)
	int ring_active = 0;	  // ring size of the calls in flight (2 for calls cut into sub-batches: radiance double-buffered)
	size_t paths_active = 0;  // path slots per call of the calls in flight
	int subs_active = 0;	  // ... and their sub-batch count
	dm::event_t ev_present_in, ev_present_out; // hand-off to / from a caller's stream (rfwhip_*_stream)
	bool present_pending = false;
	bool events_ready = false;
	size_t last_wave_off = 0; // where the most recent call's first sub-batch keeps its per-path records
	int subs_last = 1, subs_first = 0; // sub-batch slots of the most recent render call: subs_first .. subs_first + subs_last - 1
	bool cleaned = false;
	uint32_t W = 0, H = 0;

	// settings
	int integrator = 0; // 0 parity, 1 pt
	int spp = 1;
	int max_depth = 2;
	int jitter = 0; // 0 xor128, 1 center
	int stage_timing = 0;
	int count_traversal = 0;
	int builder = 0;	// 0 host (binned SAH), 1 device (Morton / Karras, lbvh.hip)
	DevBuf d_lbvh_scratch;
	int sampler = 0;	// 0 hash RNG, 1 blue noise
	DevBuf d_blue_noise;
	bool have_blue_noise = false;
	int lds_nodes = -1; // -1: as many as the kernels hold (rtk::max_lds_nodes())
	int refill = 15; // persistent lanes on — bit 0: extension waves, bit 1: shadow waves (off: the one-ray-per-lane kernels, kept as
					// a cross-check of the persistent-lane ones), bit 3: the pt primary wave in packet form (wave-uniform traversal,
					// kernels.hip: trace_packet; off: one ray per lane); bit 2 selected round 2's persistent-lane primary kernel, which
					// round 5 removed — accepted and ignored
	bool packet_ok = false; // the scene's trees fit the packet kernel's stack and its 32-bit node offsets
	int streams = 4; // sub-batches of one render call that run concurrently on their own HIP streams
	long long sub_batch_paths = 50000000; // a render call is cut into sub-batches only if each gets at least this many path slots
	int flat_instances = 1; // identity-transform instances of singly used meshes are linked into the top-level tree directly, and
							// static instances that are used several times or transformed are written out in world space (world tree)
	bool depth_stats_valid = false; // c->stats counts paths of the CURRENT scene (depth_items)
	int shadow_packets = -1;		// the connection wave of the primary vertices in packet form (kernels.hip: k_shadow_packet) where it applies:
									// 1 always, 0 never, -1 (default) while the runs it sorts hold few light bins (shadow_bins_per_run)
	int group_flags = 1;				// the packet form of the pt primary wave flags the 64-slot groups it finishes; the shade scan passes them by
	int shadow_side = 1;				// ... on the sub-batch's connection stream, beside the extension wave of depth 1 (0: on the sub-batch's own
									// stream, in front of it — what a per-stage timing wants)
	bool shadow_packets_auto_on = true; // -1: what the last waited frame's bins per run said (reset to true by rfwhip_update)
	double shadow_bins_per_run = 0.0;
	unsigned long long sp_seen[MAX_SUB][2] = {};
	unsigned long long fused_ticks_seen[MAX_SUB][rt::MAX_DEPTH_SLOTS][2] = {}; // WaveCounters::fused_ticks at the last rfwhip_wait, per counter set
	long long flatten_bytes = 1ll << 28; // ... as long as the world-space copy stays below this many bytes (256 MiB = 2.4 M triangles:
										 // the tree is built on the host inside rfwhip_update, ~0.2 s per million triangles on 16 cores)
	WorldRec wtree;
	int sample_group = 64; // slot layout: up to this many samples of a pixel share a wave (rt_core.h; the largest power of two
						   // <= this that divides every sub-batch of the call is used; 1 = a wave is one 8x8 tile of one sample;
						   // 64 = a wave is ONE pixel: primary wave 3.67 instead of 4.01 ms per 32 spp, depth-0 shadow wave -7 %)
	uint32_t sgroup_last = 0; // log2 of the group the most recent render call used
	int fuse = 1;	  // extension rays of depth d + 1 and shadow rays of depth d in one launch (kernels.hip: k_trace_fused)
	int overlap = -1; // connection waves beside the next depth's stages on a second stream: 0 off, 1 on, -1 by launch size

	// scene (host side)
	std::vector<MeshRec> meshes;
	std::vector<InstRec> instances;
	rfwhip_light_count lc = {0, 0, 0, 0};
	bool scene_dirty = true;

	// scene (device side)
	DevBuf d_nodes4f; // float form of d_nodes4 (rt::Node4f), refreshed at the end of every rfwhip_update that may need it
	bool nodes4f_current = false; // ... which is when the packet form of the primary wave can run (packet_ok, refill bit 3)
	size_t nodes4_live = 0;		  // 4-wide nodes in use: all mesh trees + the top-level tree
	DevBuf d_nodes, d_nodes4, d_nodes4_src, d_tri_verts, d_tri_shade, d_tri_uv, d_tlas_prims, d_instances;
	size_t blas_nodes4 = 0, node4_capacity = 0; // d_nodes4 = [all BLAS 4-wide nodes | TLAS 4-wide nodes | spare]
	DevBuf d_materials, d_textures, d_tex_u32, d_tex_f4, d_sky, d_area, d_point, d_spot, d_dir;
	uint32_t material_count = 0, texture_count = 0, sky_w = 0, sky_h = 0;
	bool textured = false; // some material carries a map
	uint32_t tlas_root_entry = 0, instance_count = 0;
	rt::SceneView sv;

	// wave state
	// shadow-ray buffers: two sets (by depth parity) so that shade(d + 1) can write while connect(d) still reads;
	// radiance: two sets (by call parity) x {shade, connections}, so that a call can start while the previous one resolves
	DevBuf d_org[2], d_dir2[2], d_thr[2], d_hit, d_hit_inst, d_hit0, d_hit0_inst, d_hit0_done, d_sh_org[2], d_sh_dir[2], d_sh_rad[2],
		d_rad[2], d_rad_nee[2], d_acc, d_counters, d_packet_rng, d_jump_table, d_present;
	uint32_t samples_done = 0;
	size_t wave_capacity = 0; // path slots the wave buffers can hold
	size_t rad_capacity[2] = {0, 0}; // ... and the two radiance sets
	rt::FrameView fr;
	bool jump_table_uploaded = false;

	// xor128 stream position (EmbreeRT's m_Rng persists across frames, Context.h:78)
	uint32_t rng_state[4] = {123456789u, 362436069u, 521288629u, 88675123u};
	std::vector<uint32_t> jump_table; // 64 x 128 x 4

	uint32_t probe_x = 0, probe_y = 0;
	uint32_t probe_inst = 0, probe_prim = 0;
	float probe_dist = 0;

	// timing
	std::vector<TimedSpan> spans;
	std::vector<dm::event_t> event_pool;
	size_t events_used = 0;
	float kernel_ms[KF_COUNT] = {0, 0, 0, 0, 0, 0};
	uint32_t kernel_launches[KF_COUNT] = {0, 0, 0, 0, 0, 0};
	rfwhip_render_stats stats;
	std::chrono::steady_clock::time_point render_t0;
	bool render_pending = false;
	rfwhip_counters totals;
};

// =================================================================================================================
// small math
// =================================================================================================================
static void mat4_inverse(const float *m, float *inv)
{
	float t[16];
	t[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
	t[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
	t[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
	t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
	t[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
	t[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
	t[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
	t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
	t[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
	t[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
	t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
	t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
	t[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
	t[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
	t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
	t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
	float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
	det = 1.0f / det;
	for (int i = 0; i < 16; i++)
		inv[i] = t[i] * det;
}

// xor128 is GF(2)-linear: jump_table[k] = M^(2^k), 128 columns of 4 words each.
static uint32_t xor128_step(uint32_t s[4])
{
	const uint32_t t = s[0] ^ (s[0] << 11);
	s[0] = s[1], s[1] = s[2], s[2] = s[3];
	s[3] = s[3] ^ (s[3] >> 19) ^ (t ^ (t >> 8));
	return s[3];
}
static void gf2_apply(const uint32_t *m, uint32_t s[4])
{
	uint32_t r[4] = {0, 0, 0, 0};
	for (int w = 0; w < 4; w++)
		for (int b = 0; b < 32; b++)
			if ((s[w] >> b) & 1u)
			{
				const uint32_t *c = m + 4 * (w * 32 + b);
				r[0] ^= c[0], r[1] ^= c[1], r[2] ^= c[2], r[3] ^= c[3];
			}
	memcpy(s, r, 16);
}
static void build_jump_table(std::vector<uint32_t> &tab)
{
	tab.assign(64 * 512, 0u);
	for (int i = 0; i < 128; i++)
	{
		uint32_t e[4] = {0, 0, 0, 0};
		e[i / 32] = 1u << (i % 32);
		xor128_step(e);
		memcpy(&tab[4 * i], e, 16);
	}
	for (int k = 1; k < 64; k++)
		for (int i = 0; i < 128; i++)
		{
			uint32_t c[4];
			memcpy(c, &tab[512 * (k - 1) + 4 * i], 16);
			gf2_apply(&tab[512 * (k - 1)], c);
			memcpy(&tab[512 * k + 4 * i], c, 16);
		}
}
static void xor128_jump(const std::vector<uint32_t> &tab, uint32_t s[4], unsigned long long draws)
{
	for (int k = 0; k < 64; k++)
		if ((draws >> k) & 1ull)
			gf2_apply(&tab[512 * k], s);
}

static uint32_t total_light_count(const rfwhip_context *c)
{
	return c->lc.areaLightCount + c->lc.pointLightCount + c->lc.spotLightCount + c->lc.directionalLightCount;
}

static void set_sample_group(rt::FrameView &fr, uint32_t sgroup_log2)
{
	fr.sgroup_log2 = sgroup_log2;
	fr.div_group = rt::make_fastdiv((uint32_t)std::min<unsigned long long>((unsigned long long)fr.slots << sgroup_log2, 0x7FFFFFFFull));
}
static uint32_t local_rows_of(const rfwhip_context *c)
{
	const uint32_t strips = (c->H + rt::STRIP_ROWS - 1) / rt::STRIP_ROWS;
	return ((strips + c->world - 1) / c->world) * rt::STRIP_ROWS;
}

static int sync_all(rfwhip_context *c);

// =================================================================================================================
// lifetime
// =================================================================================================================
extern "C" int rfwhip_create(int device_ordinal, int rank, int world, rfwhip_context **out)
{
	if (!out || world < 1 || rank < 0 || rank >= world)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_create: bad arguments (rank %d, world %d)", rank, world);
	int cus = 0;
	RF_TRY(dm::init(device_ordinal, &cus));
	rfwhip_context *c = new rfwhip_context();
	c->device = device_ordinal, c->rank = rank, c->world = world, c->cus = cus;
	rtk::set_device_cus(cus);
	memset(&c->stats, 0, sizeof(c->stats));
	memset(&c->totals, 0, sizeof(c->totals));
	memset(&c->sv, 0, sizeof(c->sv));
	memset(&c->fr, 0, sizeof(c->fr));
	if (dm::stream_create(&c->stream))
	{
		delete c;
		return RFWHIP_ERR_HIP;
	}
	build_jump_table(c->jump_table);
	if (c->d_counters.ensure(sizeof(rt::WaveCounters)) || dm::zero(c->d_counters.p, sizeof(rt::WaveCounters), c->stream))
	{
		delete c;
		return RFWHIP_ERR_HIP;
	}
	*out = c;
	return RFWHIP_OK;
}

static void free_all(rfwhip_context *c)
{
	for (auto &m : c->meshes)
	{
		DevBuf *mb[] = {&m.d_verts, &m.d_indices, &m.d_parents, &m.d_flags, &m.d_base_verts, &m.d_base_normals, &m.d_joints,
						&m.d_weights, &m.d_vnormals, &m.d_joint_mats, &m.d_tgt_pos, &m.d_tgt_nrm, &m.d_morph_weights,
						&m.d_b_nodes, &m.d_b_nodes4, &m.d_b_src, &m.d_b_tri_verts};
		for (DevBuf *b : mb)
			b->free_();
	}
	c->d_lbvh_scratch.free_(), c->d_blue_noise.free_();
	c->have_blue_noise = false;
	DevBuf *bufs[] = {&c->d_nodes4, &c->d_nodes4f, &c->d_nodes4_src, &c->d_nodes, &c->d_tri_verts, &c->d_tri_shade, &c->d_tri_uv, &c->d_tlas_prims, &c->d_instances,
					  &c->d_materials, &c->d_textures, &c->d_tex_u32, &c->d_tex_f4, &c->d_sky, &c->d_area, &c->d_point,
					  &c->d_spot, &c->d_dir, &c->d_org[0], &c->d_org[1], &c->d_dir2[0], &c->d_dir2[1], &c->d_thr[0],
					  &c->d_thr[1], &c->d_hit, &c->d_hit_inst, &c->d_hit0, &c->d_hit0_inst, &c->d_hit0_done, &c->d_sh_org[0], &c->d_sh_org[1],
					  &c->d_sh_dir[0], &c->d_sh_dir[1], &c->d_sh_rad[0], &c->d_sh_rad[1], &c->d_rad[0], &c->d_rad[1],
					  &c->d_rad_nee[0], &c->d_rad_nee[1], &c->d_acc, &c->d_counters, &c->d_packet_rng, &c->d_jump_table,
					  &c->d_present};
	for (DevBuf *b : bufs)
		b->free_();
	for (auto &e : c->event_pool)
		dm::event_destroy(e);
	c->event_pool.clear();
	for (int i = 0; i < rfwhip_context::MAX_SUB; i++)
		c->d_counters_sub[i].free_();
	if (c->events_ready)
	{
		dm::event_destroy(c->ev_prologue);
		for (int r = 0; r < rfwhip_context::MAX_RING; r++)
			dm::event_destroy(c->ev_resolve[r]);
		dm::event_destroy(c->ev_present_in), dm::event_destroy(c->ev_present_out);
		for (int i = 0; i < rfwhip_context::MAX_SUB; i++)
		{
			dm::event_destroy(c->ev_sub_done[i]), dm::event_destroy(c->ev_conn_last[i]);
			for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
				dm::event_destroy(c->ev_shade[i][d]), dm::event_destroy(c->ev_conn[i][d]);
		}
		c->events_ready = false, c->present_pending = false;
	}
	for (int i = 0; i < rfwhip_context::MAX_SUB; i++)
	{
		dm::stream_destroy(c->sub_stream[i]), dm::stream_destroy(c->conn_stream[i]);
		c->sub_stream[i] = nullptr, c->conn_stream[i] = nullptr;
		c->conn_used[i] = false;
	}
	for (int r = 0; r < rfwhip_context::MAX_RING; r++)
		c->resolve_recorded[r] = false;
	c->ring_active = 0, c->call_slot = 0;
	c->wave_capacity = 0, c->rad_capacity[0] = c->rad_capacity[1] = 0;
	c->blas_nodes4 = 0, c->node4_capacity = 0;
}

extern "C" int rfwhip_cleanup(rfwhip_context *c)
{
	if (!c)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null context");
	if (c->cleaned)
		return RFWHIP_OK; // the reference calls cleanup() twice on unload
	dm::use(c->device);
	if (c->stream)
		(void)sync_all(c); // every stream of the context and a present still pending on a caller's stream
	free_all(c);
	c->meshes.clear(), c->instances.clear(), c->wtree = WorldRec();
	c->cleaned = true;
	return RFWHIP_OK;
}

extern "C" void rfwhip_destroy(rfwhip_context *c)
{
	if (!c)
		return;
	rfwhip_cleanup(c);
	dm::stream_destroy(c->stream);
	delete c;
}

#define CTX_ENTER(c)                                                                      \
	if (!(c))                                                                             \
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null context");                    \
	if ((c)->cleaned)                                                                     \
		return set_error(RFWHIP_ERR_STATE, "context already cleaned up");                 \
	RF_TRY(dm::use((c)->device))

extern "C" int rfwhip_init(rfwhip_context *c, uint32_t width, uint32_t height)
{
	CTX_ENTER(c);
	if (!width || !height || width > 65536 || height > 65536)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_init: bad target size %ux%u", width, height);
	RF_TRY(sync_all(c));
	c->W = width, c->H = height;
	const uint32_t lr = local_rows_of(c);
	RF_TRY(c->d_acc.ensure((size_t)lr * width * sizeof(f4)));
	RF_TRY(dm::zero(c->d_acc.p, (size_t)lr * width * sizeof(f4), c->stream));
	c->samples_done = 0;
	c->fr.W = width, c->fr.H = height, c->fr.local_rows = lr;
	c->fr.inv_w = 1.0f / (float)width, c->fr.inv_h = 1.0f / (float)height;
	c->fr.tiles_x = (width + rt::TILE - 1) / rt::TILE;
	c->fr.div_tiles_x = rt::make_fastdiv(c->fr.tiles_x);
	c->fr.slots = c->fr.tiles_x * rt::TILE * lr;
	set_sample_group(c->fr, 0);
	c->fr.rank = (uint32_t)c->rank, c->fr.world = (uint32_t)c->world;
	return RFWHIP_OK;
}

// =================================================================================================================
// scene
// =================================================================================================================
extern "C" int rfwhip_set_sky(rfwhip_context *c, const float *rgb, size_t width, size_t height)
{
	CTX_ENTER(c);
	if (!rgb || !width || !height)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_sky: empty sky");
	std::vector<f4> px(width * height);
	for (size_t i = 0; i < width * height; i++)
		px[i] = f4{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 0.0f};
	RF_TRY(sync_all(c));
	RF_TRY(c->d_sky.ensure(px.size() * sizeof(f4)));
	RF_TRY(dm::h2d(c->d_sky.p, px.data(), px.size() * sizeof(f4), c->stream));
	RF_TRY(dm::sync(c->stream));
	c->sky_w = (uint32_t)width, c->sky_h = (uint32_t)height;
	c->scene_dirty = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_set_blue_noise(rfwhip_context *c, const uint32_t *table, size_t words)
{
	CTX_ENTER(c);
	if (!table || words < rt::BLUE_NOISE_WORDS)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_blue_noise: the table has 5 x 65536 words");
	RF_TRY(sync_all(c));
	RF_TRY(c->d_blue_noise.ensure(rt::BLUE_NOISE_WORDS * sizeof(uint32_t)));
	RF_TRY(dm::h2d(c->d_blue_noise.p, table, rt::BLUE_NOISE_WORDS * sizeof(uint32_t), c->stream));
	RF_TRY(dm::sync(c->stream));
	c->have_blue_noise = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_set_textures(rfwhip_context *c, const rfwhip_texture *tex, size_t count)
{
	CTX_ENTER(c);
	if (count && !tex)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_textures: null array");
	std::vector<rt::TexDesc> desc(count);
	std::vector<uint32_t> u32;
	std::vector<f4> f4s;
	for (size_t i = 0; i < count; i++)
	{
		rt::TexDesc &d = desc[i];
		memset(&d, 0, sizeof(d));
		d.type = tex[i].type, d.width = tex[i].width, d.height = tex[i].height, d.texelCount = tex[i].texelCount;
		if (!tex[i].data || !tex[i].texelCount)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_textures: texture %zu has no data", i);
		if (tex[i].type == RFWHIP_TEX_UINT)
		{
			d.offset = (uint32_t)u32.size();
			const uint32_t *src = (const uint32_t *)tex[i].data;
			u32.insert(u32.end(), src, src + tex[i].texelCount);
		}
		else if (tex[i].type == RFWHIP_TEX_FLOAT4)
		{
			d.offset = (uint32_t)f4s.size();
			const f4 *src = (const f4 *)tex[i].data;
			f4s.insert(f4s.end(), src, src + tex[i].texelCount);
		}
		else
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_textures: texture %zu has unknown type %u", i, tex[i].type);
	}
	RF_TRY(sync_all(c));
	RF_TRY(c->d_textures.ensure(desc.size() * sizeof(rt::TexDesc)));
	RF_TRY(c->d_tex_u32.ensure(u32.size() * 4));
	RF_TRY(c->d_tex_f4.ensure(f4s.size() * sizeof(f4)));
	RF_TRY(dm::h2d(c->d_textures.p, desc.data(), desc.size() * sizeof(rt::TexDesc), c->stream));
	RF_TRY(dm::h2d(c->d_tex_u32.p, u32.data(), u32.size() * 4, c->stream));
	RF_TRY(dm::h2d(c->d_tex_f4.p, f4s.data(), f4s.size() * sizeof(f4), c->stream));
	RF_TRY(dm::sync(c->stream));
	c->texture_count = (uint32_t)count;
	c->scene_dirty = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_set_materials(rfwhip_context *c, const rfwhip_material *materials,
									const rfwhip_material_tex_ids *tex_ids, size_t count)
{
	CTX_ENTER(c);
	if (count && !materials)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_materials: null array");
	// Every map descriptor's address becomes the index of its texture in the set_textures array.  rfw::system leaves
	// only texaddr0 filled — with the id of the LAST map it packed (material_list.cpp:385-455 write texaddr0 for every
	// slot) — and hands the per-slot texture ids over in MaterialTexIds; a backend resolves them per slot
	// (CUDART/src/Context.cpp:171-190: texture[0..2] diffuse layers, [3..5] normal maps, [6] specularity,
	// [7] roughness, [9] colour mask, [10] alpha mask).  Without ids the descriptors are taken as they are.
	std::vector<rfwhip_material> mats(materials, materials + count);
	if (tex_ids)
	{
		static const int slot_of_id[11] = {0, 1, 2, 3, 4, 5, 6, 7, -1, 8, 9};
		for (size_t i = 0; i < count; i++)
			for (int k = 0; k < 11; k++)
				if (slot_of_id[k] >= 0 && tex_ids[i].texture[k] != -1)
					mats[i].map[slot_of_id[k]].addr = (uint32_t)tex_ids[i].texture[k];
	}
	RF_TRY(sync_all(c));
	RF_TRY(c->d_materials.ensure(count * sizeof(rfwhip_material)));
	RF_TRY(dm::h2d(c->d_materials.p, mats.data(), count * sizeof(rfwhip_material), c->stream));
	RF_TRY(dm::sync(c->stream));
	c->material_count = (uint32_t)count;
	// does any material use a map?  (bits 2..5, 7..10: diffuse / normal / specularity / roughness maps and their extra layers)
	c->textured = false;
	for (size_t i = 0; i < count; i++)
		c->textured = c->textured || (mats[i].flags & 0x7BCu) != 0;
	c->scene_dirty = true;
	return RFWHIP_OK;
}

static void fill_shade_records(MeshRec &m, const rfwhip_triangle *tris)
{
	m.shade.resize(m.triCount);
	m.uv.resize(m.triCount);
	m.max_material = 0;
	for (size_t i = 0; i < m.triCount; i++)
	{
		const rfwhip_triangle &t = tris[i];
		rt::TriShade &s = m.shade[i];
		s.n0 = f4{t.vN0[0], t.vN0[1], t.vN0[2], t.Nx};
		s.n1 = f4{t.vN1[0], t.vN1[1], t.vN1[2], t.Ny};
		s.n2 = f4{t.vN2[0], t.vN2[1], t.vN2[2], t.Nz};
		float lt, mt;
		memcpy(&lt, &t.lightTriIdx, 4), memcpy(&mt, &t.material, 4);
		s.ex = f4{t.area, t.LOD, lt, mt};
		m.uv[i].tu = f4{t.u0, t.u1, t.u2, 0.0f};
		m.uv[i].tv = f4{t.v0, t.v1, t.v2, 0.0f};
		m.max_material = std::max(m.max_material, t.material);
	}
}

static inline void tri_indices(const rfwhip_mesh *mesh, size_t i, uint32_t &a, uint32_t &b, uint32_t &cidx)
{
	if (mesh->indices)
		a = mesh->indices[3 * i], b = mesh->indices[3 * i + 1], cidx = mesh->indices[3 * i + 2];
	else
		a = (uint32_t)(3 * i), b = a + 1, cidx = a + 2;
}

#ifndef RT_MAX_LEAF
#define RT_MAX_LEAF 4
#endif
constexpr int BLAS_MAX_LEAF = RT_MAX_LEAF; // triangles per leaf (<= rt::MAX_LEAF_PRIMS = 8)
// Traversal-stack budget (rt::STACK_CAPACITY entries per ray in every traversal kernel): a 4-wide node pushes at most 3
// entries and spans at least 2 BVH2 levels when it does, so a root-to-leaf path needs <= 1.5 entries per BVH2 level.
// The depth limits make the budgets hold by construction for the host builder (32 levels -> <= 48 entries, as deep as the
// reference's MAX_DEPTH, bvh_node.h:56-81; 14 levels -> <= 21 entries for the TLAS), and the exact need of every tree
// (bvh::stack_need4) is checked against them: BLAS + TLAS + 1 sentinel <= STACK_CAPACITY.
constexpr int BLAS_DEPTH_LIMIT = 32;
constexpr int TLAS_DEPTH_LIMIT = 14;
constexpr int BLAS_STACK_BUDGET = 48;
constexpr int TLAS_STACK_BUDGET = rt::STACK_CAPACITY - 1 - BLAS_STACK_BUDGET;
static_assert(TLAS_STACK_BUDGET >= (3 * TLAS_DEPTH_LIMIT) / 2 && BLAS_STACK_BUDGET >= (3 * BLAS_DEPTH_LIMIT) / 2, "traversal stack too small for the builders' depth limits");

extern "C" int rfwhip_set_mesh(rfwhip_context *c, size_t index, const rfwhip_mesh *mesh)
{
	CTX_ENTER(c);
	if (!mesh || !mesh->vertices || !mesh->triangles || !mesh->vertexCount || !mesh->triangleCount)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_mesh: empty mesh %zu", index);
	if (!mesh->indices && mesh->vertexCount < 3 * mesh->triangleCount)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_mesh: non-indexed mesh %zu needs 3 vertices per triangle", index);
	if (mesh->triangleCount > rt::ENTRY_FIRST_MASK)
		return set_error(RFWHIP_ERR_UNSUPPORTED, "rfwhip_set_mesh: more than 2^27 triangles in one mesh");
	if (mesh->indices)
		for (size_t i = 0; i < 3 * mesh->triangleCount; i++)
			if (mesh->indices[i] >= mesh->vertexCount)
				return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_mesh: index %u out of range in mesh %zu", mesh->indices[i], index);
	if (index >= c->meshes.size())
		c->meshes.resize(index + 1);
	MeshRec &m = c->meshes[index];
	const bool same_topology = m.used && m.resident && !m.dirty && m.vertexCount == mesh->vertexCount &&
							   m.triCount == mesh->triangleCount && m.indexed == (mesh->indices != nullptr);
	RF_TRY(sync_all(c));
	// vertices / indices always go to the device (the refit kernels read them there)
	RF_TRY(m.d_verts.ensure(mesh->vertexCount * sizeof(f4)));
	RF_TRY(dm::h2d(m.d_verts.p, mesh->vertices, mesh->vertexCount * sizeof(f4), c->stream));
	if (mesh->indices)
	{
		RF_TRY(m.d_indices.ensure(mesh->triangleCount * 12));
		RF_TRY(dm::h2d(m.d_indices.p, mesh->indices, mesh->triangleCount * 12, c->stream));
	}
	m.vertexCount = mesh->vertexCount, m.triCount = mesh->triangleCount, m.indexed = mesh->indices != nullptr;
	m.used = true;
	m.posed = false; // host vertices again; skinning data (if any) stays valid while the counts stay
	if (m.skinned && !same_topology)
		m.skinned = false;
	if (m.morphed && !same_topology)
		m.morphed = false;
	fill_shade_records(m, mesh->triangles);
	// mesh bounds (for the instance boxes of the TLAS)
	for (int a = 0; a < 3; a++)
		m.bounds_min[a] = 1e34f, m.bounds_max[a] = -1e34f;
	const f4 *V = (const f4 *)mesh->vertices;
	if (same_topology)
	{
		m.refits++;
		// animated mesh: same counts => refit on the device (EmbreeRT/src/Mesh.cpp:33-35, top_level_bvh.cpp:26)
		for (size_t i = 0; i < mesh->triangleCount; i++)
		{
			uint32_t ia, ib, ic;
			tri_indices(mesh, i, ia, ib, ic);
			const uint32_t id[3] = {ia, ib, ic};
			for (int k = 0; k < 3; k++)
			{
				const f4 &p = V[id[k]];
				m.bounds_min[0] = std::min(m.bounds_min[0], p.x), m.bounds_max[0] = std::max(m.bounds_max[0], p.x);
				m.bounds_min[1] = std::min(m.bounds_min[1], p.y), m.bounds_max[1] = std::max(m.bounds_max[1], p.y);
				m.bounds_min[2] = std::min(m.bounds_min[2], p.z), m.bounds_max[2] = std::max(m.bounds_max[2], p.z);
			}
		}
		for (int a = 0; a < 3; a++)
			m.bounds_min[a] -= 2e-5f, m.bounds_max[a] += 2e-5f;
		RF_TRY(dm::h2d(c->d_tri_shade.as<rt::TriShade>() + m.shade_base, m.shade.data(), m.shade.size() * sizeof(rt::TriShade), c->stream));
		RF_TRY(dm::h2d(c->d_tri_uv.as<rt::TriUV>() + m.shade_base, m.uv.data(), m.uv.size() * sizeof(rt::TriUV), c->stream));
		dm::event_t ea, eb;
		const bool timed = c->stage_timing != 0;
		if (timed)
		{
			dm::event_create(&ea), dm::event_create(&eb);
			dm::event_record(ea, c->stream);
		}
		rtk::launch_refit(c->d_nodes.as<rt::Node>(), m.node_base, m.d_parents.as<int>(), m.node_count2,
						  c->d_tri_verts.as<f4>(), m.tri_base, m.d_verts.as<f4>(),
						  m.indexed ? m.d_indices.as<uint32_t>() : nullptr, (uint32_t)m.triCount, m.d_flags.as<uint32_t>(),
						  c->stream);
		rtk::launch_refresh4(c->d_nodes4.as<rt::Node4c>() + m.n4_base, c->d_nodes4_src.as<uint32_t>() + 4ull * m.n4_base,
							 m.n4_count, c->d_nodes.as<rt::Node>() + m.node_base, c->stream);
		RF_TRY(dm::last_launch_error());
		if (timed)
			dm::event_record(eb, c->stream);
		RF_TRY(dm::sync(c->stream));
		if (timed)
		{
			c->kernel_ms[KF_REFIT] += dm::event_ms(ea, eb);
			c->kernel_launches[KF_REFIT] += 3;
			c->stats.animationTime = dm::event_ms(ea, eb);
			dm::event_destroy(ea), dm::event_destroy(eb);
		}
		c->scene_dirty = true; // instance boxes change: the TLAS is rebuilt in update()
		return RFWHIP_OK;
	}
	// (re)build.  From here on the record no longer describes what sits in the scene-wide arrays: whatever fails below, the
	// next set_mesh must not take the refit path and the next update must place the mesh again.
	m.dirty = true, m.resident = false;
	c->scene_dirty = true;
	// ... and until the build below has succeeded it describes NO tree (an error return leaves a mesh rfwhip_update() refuses,
	// not the counts of the previous build beside freed or half-written arrays)
	m.built = false, m.device_built = false, m.node_count2 = m.n4_count = 0, m.stack_need = 0;
	m.generation++, m.refits = 0;
	const size_t n = mesh->triangleCount;
	bool device_built = false;
	if (c->builder == 1 && n > (size_t)BLAS_MAX_LEAF)
	{
		// Construction on the device, end to end (lbvh.hip): Morton order, locally-ordered clustering, 4-wide collapse with
		// quantisation, depth-first triangle order — all into this mesh's own device arrays.  Only the counts, the stack
		// need and the root box come back; rfwhip_update() places the mesh with device-to-device copies.
		const size_t n2 = 2 * n;
		RF_TRY(c->d_lbvh_scratch.ensure(rtk::device_build_scratch_bytes((uint32_t)n)));
		RF_TRY(m.d_b_nodes.ensure(n2 * sizeof(rt::Node)));
		RF_TRY(m.d_parents.ensure(n2 * sizeof(int)));
		RF_TRY(m.d_flags.ensure(n2 * sizeof(uint32_t)));
		RF_TRY(m.d_b_nodes4.ensure(n * sizeof(rt::Node4c)));
		RF_TRY(m.d_b_src.ensure(4 * n * sizeof(uint32_t)));
		RF_TRY(m.d_b_tri_verts.ensure(3 * n * sizeof(f4)));
		const auto t0 = std::chrono::steady_clock::now();
		rtk::DeviceBuildResult res;
		const int rc = rtk::launch_device_build(m.d_verts.as<f4>(), m.indexed ? m.d_indices.as<uint32_t>() : nullptr, (uint32_t)n,
												c->d_lbvh_scratch.p, c->d_lbvh_scratch.cap, m.d_b_nodes.as<rt::Node>(),
												m.d_parents.as<int>(), m.d_flags.as<uint32_t>(), m.d_b_nodes4.as<rt::Node4c>(),
												m.d_b_src.as<uint32_t>(), m.d_b_tri_verts.as<f4>(), &res, c->stream);
		if (rc > 1 && rc != 5) // (5: the clustering did not converge within its pass budget — the host builder takes the mesh)
			return set_error(RFWHIP_ERR_HIP, "rfwhip_set_mesh: device BVH build failed (%d)", rc);
		RF_TRY(dm::last_launch_error());
		if (c->stage_timing)
		{
			c->kernel_ms[KF_REFIT] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
			c->kernel_launches[KF_REFIT] += 1;
		}
		// a tree too deep for the traversal stack falls back to the host builder
		device_built = rc == 0 && (int)res.stack_need <= BLAS_STACK_BUDGET;
		if (device_built)
		{
			m.node_count2 = res.node_count, m.n4_count = res.node4_count, m.stack_need = (int)res.stack_need;
			for (int a = 0; a < 3; a++)
				m.bounds_min[a] = res.bmin[a], m.bounds_max[a] = res.bmax[a];
			m.bvh = bvh::Result(), m.n4.clear(), m.leaf_verts.clear();
		}
	}
	if (device_built)
	{
		RF_TRY(dm::sync(c->stream));
		m.device_built = true, m.built = true;
		return RFWHIP_OK;
	}
	if (!device_built)
	{
		std::vector<float> bmin(3 * n), bmax(3 * n);
		for (size_t i = 0; i < n; i++)
		{
			uint32_t ia, ib, ic;
			tri_indices(mesh, i, ia, ib, ic);
			const f4 &p0 = V[ia], &p1 = V[ib], &p2 = V[ic];
			// per-triangle box grown by 1e-5 (bvh_tree.cpp:407-412)
			bmin[3 * i + 0] = std::min(p0.x, std::min(p1.x, p2.x)) - 1e-5f, bmax[3 * i + 0] = std::max(p0.x, std::max(p1.x, p2.x)) + 1e-5f;
			bmin[3 * i + 1] = std::min(p0.y, std::min(p1.y, p2.y)) - 1e-5f, bmax[3 * i + 1] = std::max(p0.y, std::max(p1.y, p2.y)) + 1e-5f;
			bmin[3 * i + 2] = std::min(p0.z, std::min(p1.z, p2.z)) - 1e-5f, bmax[3 * i + 2] = std::max(p0.z, std::max(p1.z, p2.z)) + 1e-5f;
		}
		bvh::build(bmin.data(), bmax.data(), n, BLAS_MAX_LEAF, BLAS_DEPTH_LIMIT, m.bvh);
	}
	for (int a = 0; a < 3; a++)
		m.bounds_min[a] = m.bvh.nodes[0].bmin[a], m.bounds_max[a] = m.bvh.nodes[0].bmax[a];
	bvh::collapse4(m.bvh, false, m.n4);
	m.node_count2 = (uint32_t)m.bvh.nodes.size(), m.n4_count = (uint32_t)m.n4.size();
	m.stack_need = bvh::stack_need4(m.n4);
	if (m.stack_need > BLAS_STACK_BUDGET)
		return set_error(RFWHIP_ERR_UNSUPPORTED, "rfwhip_set_mesh: the BVH of mesh %zu needs %d traversal-stack entries (budget %d)",
						 index, m.stack_need, BLAS_STACK_BUDGET);
	m.leaf_verts.resize(3 * n);
	for (size_t s = 0; s < n; s++)
	{
		const uint32_t prim = m.bvh.order[s];
		uint32_t ia, ib, ic;
		tri_indices(mesh, prim, ia, ib, ic);
		float pw;
		memcpy(&pw, &prim, 4);
		m.leaf_verts[3 * s + 0] = f4{V[ia].x, V[ia].y, V[ia].z, pw};
		m.leaf_verts[3 * s + 1] = f4{V[ib].x, V[ib].y, V[ib].z, 1.0f};
		m.leaf_verts[3 * s + 2] = f4{V[ic].x, V[ic].y, V[ic].z, rt::TRI_EPS}; // (w: the triangle's determinant threshold, rt::tri_test)
	}
	RF_TRY(m.d_parents.ensure(m.bvh.parents.size() * sizeof(int)));
	RF_TRY(dm::h2d(m.d_parents.p, m.bvh.parents.data(), m.bvh.parents.size() * sizeof(int), c->stream));
	RF_TRY(m.d_flags.ensure(m.bvh.nodes.size() * sizeof(uint32_t)));
	RF_TRY(dm::sync(c->stream));
	m.built = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_set_instance(rfwhip_context *c, size_t index, size_t mesh_index, const float *transform16,
								   const float *normal_matrix9)
{
	CTX_ENTER(c);
	if (!transform16 || !normal_matrix9)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_instance: null matrix");
	if (mesh_index >= c->meshes.size() || !c->meshes[mesh_index].used)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_instance: instance %zu refers to unknown mesh %zu", index, mesh_index);
	if (index >= c->instances.size())
		c->instances.resize(index + 1);
	InstRec &in = c->instances[index];
	in.used = true, in.mesh = mesh_index;
	memcpy(in.transform, transform16, 64), memcpy(in.normal, normal_matrix9, 36);
	c->scene_dirty = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_set_lights(rfwhip_context *c, rfwhip_light_count n, const rfwhip_area_light *area,
								 const rfwhip_point_light *point, const rfwhip_spot_light *spot,
								 const rfwhip_directional_light *directional)
{
	CTX_ENTER(c);
	if ((n.areaLightCount && !area) || (n.pointLightCount && !point) || (n.spotLightCount && !spot) ||
		(n.directionalLightCount && !directional))
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_lights: count > 0 with a null array");
	RF_TRY(sync_all(c));
	RF_TRY(c->d_area.ensure(n.areaLightCount * sizeof(rfwhip_area_light)));
	RF_TRY(c->d_point.ensure(n.pointLightCount * sizeof(rfwhip_point_light)));
	RF_TRY(c->d_spot.ensure(n.spotLightCount * sizeof(rfwhip_spot_light)));
	RF_TRY(c->d_dir.ensure(n.directionalLightCount * sizeof(rfwhip_directional_light)));
	RF_TRY(dm::h2d(c->d_area.p, area, n.areaLightCount * sizeof(rfwhip_area_light), c->stream));
	RF_TRY(dm::h2d(c->d_point.p, point, n.pointLightCount * sizeof(rfwhip_point_light), c->stream));
	RF_TRY(dm::h2d(c->d_spot.p, spot, n.spotLightCount * sizeof(rfwhip_spot_light), c->stream));
	RF_TRY(dm::h2d(c->d_dir.p, directional, n.directionalLightCount * sizeof(rfwhip_directional_light), c->stream));
	RF_TRY(dm::sync(c->stream));
	c->lc = n;
	c->scene_dirty = true;
	return RFWHIP_OK;
}

// ---- device skinning (not part of the RenderContext interface: rfw::system skins on the host,
// geometry/gltf/mesh.cpp:18-125, and hands the result to set_mesh) ---------------------------------------------------
extern "C" int rfwhip_set_mesh_skin(rfwhip_context *c, size_t index, const uint32_t *joints4, const float *weights4,
									const float *base_normals4, size_t vertex_count)
{
	CTX_ENTER(c);
	if (index >= c->meshes.size() || !c->meshes[index].used)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_set_mesh_skin: mesh %zu has not been set", index);
	MeshRec &m = c->meshes[index];
	if (!joints4 || !weights4 || !base_normals4 || vertex_count != m.vertexCount)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_mesh_skin: need joints, weights and normals for the mesh's %zu vertices", m.vertexCount);
	if (m.posed)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_set_mesh_skin: mesh %zu is posed; set_mesh the bind pose first", index);
	RF_TRY(sync_all(c));
	const size_t n = vertex_count;
	RF_TRY(m.d_base_verts.ensure(n * sizeof(f4)));
	RF_TRY(m.d_base_normals.ensure(n * sizeof(f4)));
	RF_TRY(m.d_vnormals.ensure(n * sizeof(f4)));
	RF_TRY(m.d_joints.ensure(n * 16));
	RF_TRY(m.d_weights.ensure(n * sizeof(f4)));
	RF_TRY(dm::d2d(m.d_base_verts.p, m.d_verts.p, n * sizeof(f4), c->stream)); // the vertices of the last set_mesh = bind pose
	RF_TRY(dm::h2d(m.d_base_normals.p, base_normals4, n * sizeof(f4), c->stream));
	RF_TRY(dm::h2d(m.d_joints.p, joints4, n * 16, c->stream));
	RF_TRY(dm::h2d(m.d_weights.p, weights4, n * sizeof(f4), c->stream));
	RF_TRY(dm::sync(c->stream));
	m.skinned = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_pose_mesh(rfwhip_context *c, size_t index, const float *joint_matrices16, size_t joint_count)
{
	CTX_ENTER(c);
	if (index >= c->meshes.size() || !c->meshes[index].used || !c->meshes[index].skinned)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_pose_mesh: mesh %zu has no skin (rfwhip_set_mesh_skin)", index);
	MeshRec &m = c->meshes[index];
	if (!joint_matrices16 || !joint_count)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_pose_mesh: no joint matrices");
	if (!m.resident || m.dirty)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_pose_mesh: mesh %zu is not resident yet (rfwhip_update first)", index);
	RF_TRY(sync_all(c));
	RF_TRY(m.d_joint_mats.ensure(joint_count * 64));
	RF_TRY(dm::h2d(m.d_joint_mats.p, joint_matrices16, joint_count * 64, c->stream));
	m.joint_count = (uint32_t)joint_count;
	dm::event_t ea, eb;
	const bool timed = c->stage_timing != 0;
	if (timed)
	{
		dm::event_create(&ea), dm::event_create(&eb);
		dm::event_record(ea, c->stream);
	}
	rtk::launch_skin_vertices(m.d_verts.as<f4>(), m.d_vnormals.as<f4>(), m.d_base_verts.as<f4>(), m.d_base_normals.as<f4>(),
							  m.d_joints.as<uint32_t>(), m.d_weights.as<f4>(), m.d_joint_mats.as<float>(), m.joint_count,
							  (uint32_t)m.vertexCount, c->stream);
	rtk::launch_skin_shade(c->d_tri_shade.as<rt::TriShade>() + m.shade_base, m.d_verts.as<f4>(), m.d_vnormals.as<f4>(),
						   m.indexed ? m.d_indices.as<uint32_t>() : nullptr, (uint32_t)m.triCount, c->stream);
	rtk::launch_refit(c->d_nodes.as<rt::Node>(), m.node_base, m.d_parents.as<int>(), m.node_count2,
					  c->d_tri_verts.as<f4>(), m.tri_base, m.d_verts.as<f4>(), m.indexed ? m.d_indices.as<uint32_t>() : nullptr,
					  (uint32_t)m.triCount, m.d_flags.as<uint32_t>(), c->stream);
	rtk::launch_refresh4(c->d_nodes4.as<rt::Node4c>() + m.n4_base, c->d_nodes4_src.as<uint32_t>() + 4ull * m.n4_base,
						 m.n4_count, c->d_nodes.as<rt::Node>() + m.node_base, c->stream);
	RF_TRY(dm::last_launch_error());
	if (timed)
		dm::event_record(eb, c->stream);
	// the instance boxes of the TLAS come from the mesh bounds = the refitted root (its two children's union)
	rt::Node root[2];
	const rt::Node *dn = c->d_nodes.as<rt::Node>() + m.node_base;
	RF_TRY(dm::d2h(&root[0], dn, sizeof(rt::Node), c->stream));
	RF_TRY(dm::sync(c->stream));
	for (int a = 0; a < 3; a++)
		m.bounds_min[a] = root[0].bmin[a] - 2e-5f, m.bounds_max[a] = root[0].bmax[a] + 2e-5f;
	if (timed)
	{
		c->kernel_ms[KF_REFIT] += dm::event_ms(ea, eb);
		c->kernel_launches[KF_REFIT] += 5;
		c->stats.animationTime = dm::event_ms(ea, eb);
		dm::event_destroy(ea), dm::event_destroy(eb);
	}
	m.posed = true;
	c->scene_dirty = true; // instance boxes change: the TLAS is rebuilt in update()
	return RFWHIP_OK;
}

// float 4-wide node of the collapse -> the compressed node the rays fetch (entries as they are)
static rt::Node4c compress4(const rt::Node4 &nd)
{
	rt::Node4c out;
	bool valid[4];
	for (int k = 0; k < 4; k++)
		valid[k] = nd.entry[k] != rt::ENTRY_EMPTY, out.entry[k] = nd.entry[k];
	rt::pack_boxes4c(out, nd.lo, nd.hi, valid);
	return out;
}

// ---- device morph targets (geometry/gltf/mesh.cpp:127-147 on the device) ---------------------------------------
extern "C" int rfwhip_set_mesh_morph(rfwhip_context *c, size_t index, const float *base_normals4, const float *target_positions4,
									 const float *target_normals4, size_t target_count, size_t vertex_count)
{
	CTX_ENTER(c);
	if (index >= c->meshes.size() || !c->meshes[index].used)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_set_mesh_morph: mesh %zu has not been set", index);
	MeshRec &m = c->meshes[index];
	if (!base_normals4 || !target_positions4 || !target_normals4 || !target_count || vertex_count != m.vertexCount)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_set_mesh_morph: need base normals and >= 1 target for the mesh's %zu vertices", m.vertexCount);
	if (m.posed)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_set_mesh_morph: mesh %zu is posed; set_mesh the base pose first", index);
	RF_TRY(sync_all(c));
	const size_t n = vertex_count;
	RF_TRY(m.d_base_verts.ensure(n * sizeof(f4)));
	RF_TRY(m.d_base_normals.ensure(n * sizeof(f4)));
	RF_TRY(m.d_vnormals.ensure(n * sizeof(f4)));
	RF_TRY(m.d_tgt_pos.ensure(target_count * n * sizeof(f4)));
	RF_TRY(m.d_tgt_nrm.ensure(target_count * n * sizeof(f4)));
	RF_TRY(m.d_morph_weights.ensure(target_count * sizeof(float)));
	RF_TRY(dm::d2d(m.d_base_verts.p, m.d_verts.p, n * sizeof(f4), c->stream)); // the vertices of the last set_mesh = base pose
	RF_TRY(dm::h2d(m.d_base_normals.p, base_normals4, n * sizeof(f4), c->stream));
	RF_TRY(dm::h2d(m.d_tgt_pos.p, target_positions4, target_count * n * sizeof(f4), c->stream));
	RF_TRY(dm::h2d(m.d_tgt_nrm.p, target_normals4, target_count * n * sizeof(f4), c->stream));
	RF_TRY(dm::sync(c->stream));
	m.morphed = true, m.skinned = false;
	m.target_count = (uint32_t)target_count;
	return RFWHIP_OK;
}

extern "C" int rfwhip_morph_mesh(rfwhip_context *c, size_t index, const float *weights, size_t weight_count)
{
	CTX_ENTER(c);
	if (index >= c->meshes.size() || !c->meshes[index].used || !c->meshes[index].morphed)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_morph_mesh: mesh %zu has no morph targets (rfwhip_set_mesh_morph)", index);
	MeshRec &m = c->meshes[index];
	if (!weights || weight_count != m.target_count)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_morph_mesh: mesh %zu has %u targets, got %zu weights", index, m.target_count, weight_count);
	if (!m.resident || m.dirty)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_morph_mesh: mesh %zu is not resident yet (rfwhip_update first)", index);
	RF_TRY(sync_all(c));
	RF_TRY(dm::h2d(m.d_morph_weights.p, weights, weight_count * sizeof(float), c->stream));
	dm::event_t ea, eb;
	const bool timed = c->stage_timing != 0;
	if (timed)
	{
		dm::event_create(&ea), dm::event_create(&eb);
		dm::event_record(ea, c->stream);
	}
	rtk::launch_morph_vertices(m.d_verts.as<f4>(), m.d_vnormals.as<f4>(), m.d_base_verts.as<f4>(), m.d_base_normals.as<f4>(),
							   m.d_tgt_pos.as<f4>(), m.d_tgt_nrm.as<f4>(), m.d_morph_weights.as<float>(), m.target_count,
							   (uint32_t)m.vertexCount, c->stream);
	rtk::launch_skin_shade(c->d_tri_shade.as<rt::TriShade>() + m.shade_base, m.d_verts.as<f4>(), m.d_vnormals.as<f4>(),
						   m.indexed ? m.d_indices.as<uint32_t>() : nullptr, (uint32_t)m.triCount, c->stream);
	rtk::launch_refit(c->d_nodes.as<rt::Node>(), m.node_base, m.d_parents.as<int>(), m.node_count2,
					  c->d_tri_verts.as<f4>(), m.tri_base, m.d_verts.as<f4>(), m.indexed ? m.d_indices.as<uint32_t>() : nullptr,
					  (uint32_t)m.triCount, m.d_flags.as<uint32_t>(), c->stream);
	rtk::launch_refresh4(c->d_nodes4.as<rt::Node4c>() + m.n4_base, c->d_nodes4_src.as<uint32_t>() + 4ull * m.n4_base,
						 m.n4_count, c->d_nodes.as<rt::Node>() + m.node_base, c->stream);
	RF_TRY(dm::last_launch_error());
	if (timed)
		dm::event_record(eb, c->stream);
	rt::Node root;
	RF_TRY(dm::d2h(&root, c->d_nodes.as<rt::Node>() + m.node_base, sizeof(rt::Node), c->stream));
	RF_TRY(dm::sync(c->stream));
	for (int a = 0; a < 3; a++)
		m.bounds_min[a] = root.bmin[a] - 2e-5f, m.bounds_max[a] = root.bmax[a] + 2e-5f;
	if (timed)
	{
		c->kernel_ms[KF_REFIT] += dm::event_ms(ea, eb);
		c->kernel_launches[KF_REFIT] += 5;
		c->stats.animationTime = dm::event_ms(ea, eb);
		dm::event_destroy(ea), dm::event_destroy(eb);
	}
	m.posed = true;
	c->scene_dirty = true; // instance boxes change: the TLAS is rebuilt in update()
	return RFWHIP_OK;
}

// The float copy of the traversal nodes (rt::Node4f) — written only when its one reader, the packet form of the pt primary wave,
// can run: at the end of an update, or by the first render call after `refill` bit 3 was switched on.
static int ensure_nodes4f(rfwhip_context *c)
{
	if (c->nodes4f_current || !c->packet_ok || !(c->refill & 8) || !c->nodes4_live)
		return 0;
	RF_TRY(c->d_nodes4f.ensure(c->node4_capacity * sizeof(rt::Node4f)));
	rtk::launch_expand4(c->d_nodes4.as<rt::Node4c>(), c->d_nodes4f.as<rt::Node4f>(), (uint32_t)c->nodes4_live, c->stream);
	RF_TRY(dm::last_launch_error());
	c->sv.nodes4f = c->d_nodes4f.as<rt::Node4f>();
	c->nodes4f_current = true;
	return 0;
}

// ---- the WORLD TREE -------------------------------------------------------------------------------------------------------------
// The reference walks two levels for every ray: top-level leaf -> instance -> the ray in object space -> mesh tree -> back
// (top_level_bvh.cpp:104-168).  A static instance needs none of that at run time: its triangles can be written out in WORLD space
// once, per update, and then ALL static geometry hangs under ONE tree built over all of it — no instance switch (ray transform,
// three reciprocals, a leaf phase in and one out per instance a ray's box test passes), no top-level tree over boxes that contain
// each other (a room around its columns), one top-of-tree cache in LDS instead of several trees' tops.  Semantics kept: the hit
// is the same triangle of the same instance at the same t (directions are not renormalised in the two-level walk, so t is shared;
// here the triangle simply moved instead of the ray); primitive ids stay mesh-relative (v0.w) and the instance index travels in
// v1.w, so shading still goes through the instance's record (normal matrix, shading records in object space).  For an instance
// whose matrix is the identity the world-space vertices ARE the object-space ones and nothing a ray computes changes; for a
// transformed one the triangle test sees M p instead of M^-1 o: the same numbers up to rounding.
// Members: every instance of a mesh that was built on the host and is not animated (skinned, morphed, posed, or re-set with the same
// topology since its build) whose own matrix has not changed in the last eight updates, as long as the copy stays below `flatten_bytes`.  A single member that is an identity instance of a
// singly used mesh has its mesh tree linked into the top-level tree instead (flat instances, below): same effect, no copy.
// Animated meshes and meshes built on the device keep the two-level walk.  The tree is rebuilt only when its key — members, their
// meshes' build generations, their matrices — changes (an animated scene's per-frame update leaves it alone).
constexpr uint32_t WORLD_ID = 0xFFFFFFFFu;
static int prepare_world(rfwhip_context *c, bool &changed)
{
	WorldRec &w = c->wtree;
	changed = false;
	std::vector<uint8_t> member(c->instances.size(), 0), mesh_member(c->meshes.size(), 0);
	std::vector<uint32_t> key;
	size_t tris = 0;
	bool worth = false;
	if (c->flat_instances != 0)
	{
		std::vector<uint32_t> uses(c->meshes.size(), 0u);
		for (const InstRec &in : c->instances)
			if (in.used && in.mesh < c->meshes.size() && c->meshes[in.mesh].used)
				uses[in.mesh]++;
		static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
		for (size_t i = 0; i < c->instances.size(); i++)
		{
			InstRec &in = c->instances[i];
			if (!in.used || in.mesh >= c->meshes.size() || !c->meshes[in.mesh].used)
				continue;
			// (another mesh, or the same one rebuilt: a new object in this slot, not a moving one)
			const bool same_object = in.has_prev && in.prev_mesh == in.mesh && in.prev_generation == c->meshes[in.mesh].generation;
			if (!same_object)
				in.moving = 0u, in.moves = 0u;
			else if (memcmp(in.prev_transform, in.transform, sizeof(in.transform)) != 0)
			{
				// stable for eight updates before it rejoins: one rebuild when an animation starts, one after it ends — and twice as
				// long after every further episode (8, 16, 32 ... 8192 updates): an instance that keeps starting and stopping ends up
				// outside the tree for good instead of paying two synchronous host rebuilds per episode (round 5's advisor)
				if (!in.moving)
					in.moves = std::min(in.moves + 1u, 11u);
				in.moving = 8u << (in.moves - 1u);
			}
			else if (in.moving)
				in.moving--;
			memcpy(in.prev_transform, in.transform, sizeof(in.transform));
			in.prev_mesh = in.mesh, in.prev_generation = c->meshes[in.mesh].generation, in.has_prev = true;
			if (in.moving)
				continue;
			const MeshRec &m = c->meshes[in.mesh];
			if (!m.built || m.device_built || m.skinned || m.morphed || m.posed || m.refits != 0u || m.leaf_verts.size() != 3 * m.triCount || !m.triCount)
				continue;
			member[i] = 1, mesh_member[in.mesh] = 1, tris += m.triCount;
			if (memcmp(in.transform, ident, sizeof(ident)) != 0 || uses[in.mesh] > 1u)
				worth = true;
			key.push_back((uint32_t)i), key.push_back((uint32_t)in.mesh), key.push_back(m.generation);
			for (int k = 0; k < 16; k++)
			{
				uint32_t b;
				memcpy(&b, &in.transform[k], 4);
				key.push_back(b);
			}
		}
	}
	if (key.size() > 19) // (two members or more: one tree over all of them beats a top-level tree over theirs even when every one of
		worth = true;	 // them is an identity instance — terrain + light quads of the bench scene: +0.9 %)
	const unsigned long long bytes = (unsigned long long)tris * (3 * sizeof(f4) + sizeof(rt::Node4c)); // (at most one 4-wide node per triangle)
	if (!worth || !tris || bytes > (unsigned long long)c->flatten_bytes)
	{
		std::fill(member.begin(), member.end(), 0), std::fill(mesh_member.begin(), mesh_member.end(), 0);
		key.clear(), tris = 0;
	}
	if (key == w.key && w.member.size() == member.size())
		return 0; // as built (or, if that build was refused, as refused)
	changed = true;
	w = WorldRec();
	w.key = key;
	w.member.assign(member.size(), 0), w.mesh_member.assign(mesh_member.size(), 0);
	if (key.empty())
		return 0;
	// the members' triangles in world space (leaf order of their meshes: any order does), with what a hit record needs
	std::vector<f4> verts(3 * tris);
	std::vector<float> bmin(3 * tris), bmax(3 * tris);
	size_t t = 0;
	for (size_t i = 0; i < c->instances.size(); i++)
	{
		if (!member[i])
			continue;
		const InstRec &in = c->instances[i];
		const MeshRec &m = c->meshes[in.mesh];
		const float *M = in.transform; // column-major 4x4, like the instance boxes below
		const uint32_t ii = (uint32_t)i;
		float iw;
		memcpy(&iw, &ii, 4);
		// the determinant threshold of the reference's triangle test, stated in object space (rt::tri_test): a = e1 . (d x e2) of the
		// world-space triangle is det(M) times the object-space one
		const double det3 = (double)M[0] * ((double)M[5] * M[10] - (double)M[9] * M[6]) - (double)M[4] * ((double)M[1] * M[10] - (double)M[9] * M[2]) +
							(double)M[8] * ((double)M[1] * M[6] - (double)M[5] * M[2]);
		const float tri_eps = std::max((float)((double)rt::TRI_EPS * fabs(det3)), 1e-37f); // (a singular matrix: every triangle is rejected)
		for (size_t s = 0; s < m.triCount; s++, t++)
		{
			for (int k = 0; k < 3; k++)
			{
				const f4 &p = m.leaf_verts[3 * s + k];
				f4 q;
				q.x = M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12];
				q.y = M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13];
				q.z = M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14];
				q.w = k == 0 ? p.w : (k == 1 ? iw : tri_eps); // v0.w: primitive id (mesh order), v1.w: instance, v2.w: determinant threshold
				verts[3 * t + k] = q;
			}
			const f4 &a = verts[3 * t], &b = verts[3 * t + 1], &d = verts[3 * t + 2];
			// per-triangle box grown by 1e-5 (bvh_tree.cpp:407-412), as rfwhip_set_mesh does
			bmin[3 * t + 0] = std::min(a.x, std::min(b.x, d.x)) - 1e-5f, bmax[3 * t + 0] = std::max(a.x, std::max(b.x, d.x)) + 1e-5f;
			bmin[3 * t + 1] = std::min(a.y, std::min(b.y, d.y)) - 1e-5f, bmax[3 * t + 1] = std::max(a.y, std::max(b.y, d.y)) + 1e-5f;
			bmin[3 * t + 2] = std::min(a.z, std::min(b.z, d.z)) - 1e-5f, bmax[3 * t + 2] = std::max(a.z, std::max(b.z, d.z)) + 1e-5f;
		}
	}
	bvh::Result tree;
	bvh::build(bmin.data(), bmax.data(), tris, BLAS_MAX_LEAF, BLAS_DEPTH_LIMIT, tree);
	const bool inner = bvh::collapse4(tree, false, w.n4);
	w.stack_need = bvh::stack_need4(w.n4);
	if (w.stack_need > BLAS_STACK_BUDGET)
	{
		// (a tree this deep is refused like a mesh's would be; the instances keep the two-level walk.  The key stays: no retry per update)
		w.n4.clear(), w.stack_need = 0;
		return 0;
	}
	if (!inner)
		w.root_first = (uint32_t)tree.nodes[0].left_first, w.root_count = (uint32_t)tree.nodes[0].count;
	w.leaf_verts.resize(3 * tris);
	for (size_t s = 0; s < tris; s++)
		for (int k = 0; k < 3; k++)
			w.leaf_verts[3 * s + k] = verts[3 * (size_t)tree.order[s] + k];
	for (int a = 0; a < 3; a++)
		w.bmin[a] = tree.nodes[0].bmin[a], w.bmax[a] = tree.nodes[0].bmax[a];
	w.tris = tris, w.member = member, w.mesh_member = mesh_member, w.valid = true;
	w.mesh_all = mesh_member;
	for (size_t i = 0; i < c->instances.size(); i++)
		if (c->instances[i].used && c->instances[i].mesh < w.mesh_all.size() && !member[i])
			w.mesh_all[c->instances[i].mesh] = 0;
	return 0;
}

extern "C" int rfwhip_update(rfwhip_context *c)
{
	CTX_ENTER(c);
	RF_TRY(sync_all(c));
	// the shade kernels index the material table with the triangles' ids unchecked
	for (size_t i = 0; i < c->meshes.size(); i++)
		if (c->meshes[i].used && c->meshes[i].max_material >= c->material_count)
			return set_error(RFWHIP_ERR_STATE, "rfwhip_update: mesh %zu refers to material %u, but %u materials are set",
							 i, c->meshes[i].max_material, c->material_count);
	for (size_t i = 0; i < c->meshes.size(); i++)
		if (c->meshes[i].used && !c->meshes[i].built)
			return set_error(RFWHIP_ERR_STATE, "rfwhip_update: the last rfwhip_set_mesh of mesh %zu failed; set it again", i);
	// ---- place meshes in the global arrays (only when some mesh was rebuilt) ----
	bool relayout = false;
	size_t live_instances = 0;
	for (auto &in : c->instances)
		if (in.used)
			live_instances++;
	const size_t tlas_reserve = live_instances + 3; // a 4-wide tree over n single-instance leaves (+ the world tree's) has < n inner nodes
	for (auto &m : c->meshes)
		if (m.used && m.dirty)
			relayout = true;
	bool world_changed = false;
	RF_TRY(prepare_world(c, world_changed));
	if (world_changed)
		relayout = true;
	WorldRec &world = c->wtree;
	if (c->blas_nodes4 + tlas_reserve > c->node4_capacity)
		relayout = true; // the TLAS lives behind the BLAS nodes in the same array: grow it together
	if (relayout)
	{
		size_t nodes = 0, tris = 0, nodes4 = 0;
		for (auto &m : c->meshes)
			if (m.used)
			{
				m.node_base = (uint32_t)nodes, m.tri_base = (uint32_t)tris, m.shade_base = (uint32_t)tris;
				m.n4_base = (uint32_t)nodes4;
				nodes += m.node_count2, tris += m.triCount, nodes4 += m.n4_count;
			}
		// the world tree's nodes and world-space triangles follow the meshes' (shading records stay per mesh, in object space)
		const size_t mesh_tris = tris, mesh_nodes4 = nodes4;
		world.resident = false;
		if (world.valid)
		{
			world.n4_base = (uint32_t)nodes4, world.tri_base = (uint32_t)tris;
			nodes4 += world.n4.size(), tris += world.tris;
		}
		if (tris > rt::ENTRY_FIRST_MASK)
			return set_error(RFWHIP_ERR_UNSUPPORTED, "more than 2^27 triangles in the scene");
		c->blas_nodes4 = nodes4;
		c->node4_capacity = nodes4 + 2 * tlas_reserve + 64;
		if (c->node4_capacity >= (size_t(1) << 26))
			return set_error(RFWHIP_ERR_UNSUPPORTED, "more than 2^26 4-wide nodes (the traversal addresses them by 32-bit byte offsets)");
		RF_TRY(c->d_nodes.ensure(nodes * sizeof(rt::Node)));
		RF_TRY(c->d_nodes4.ensure(c->node4_capacity * sizeof(rt::Node4c)));
		RF_TRY(c->d_nodes4_src.ensure(4 * mesh_nodes4 * sizeof(uint32_t)));
		RF_TRY(c->d_tri_verts.ensure(3 * tris * sizeof(f4)));
		RF_TRY(c->d_tri_shade.ensure(mesh_tris * sizeof(rt::TriShade)));
		RF_TRY(c->d_tri_uv.ensure(mesh_tris * sizeof(rt::TriUV)));
		// One mesh at a time.  Device form everywhere: left_first / entries carry ready-made stack entries (rt::make_entry)
		// with ABSOLUTE indices — node index into the scene-wide arrays, leaf-ordered triangle index into tri_verts.
		std::vector<rt::Node> nodes2;
		std::vector<rt::Node4c> nodes4c;
		std::vector<uint32_t> src;
		for (auto &m : c->meshes)
		{
			if (!m.used)
				continue;
			rt::Node *dn = c->d_nodes.as<rt::Node>() + m.node_base;
			rt::Node4c *dn4 = c->d_nodes4.as<rt::Node4c>() + m.n4_base;
			uint32_t *dsrc = c->d_nodes4_src.as<uint32_t>() + 4ull * m.n4_base;
			f4 *dtv = c->d_tri_verts.as<f4>() + 3ull * m.tri_base;
			if (m.device_built)
			{
				// built on the device in mesh-local arrays: copy and rebase there, nothing touches the host
				RF_TRY(dm::d2d(dn, m.d_b_nodes.p, (size_t)m.node_count2 * sizeof(rt::Node), c->stream));
				RF_TRY(dm::d2d(dn4, m.d_b_nodes4.p, (size_t)m.n4_count * sizeof(rt::Node4c), c->stream));
				RF_TRY(dm::d2d(dsrc, m.d_b_src.p, 4ull * m.n4_count * sizeof(uint32_t), c->stream));
				RF_TRY(dm::d2d(dtv, m.d_b_tri_verts.p, 3ull * m.triCount * sizeof(f4), c->stream));
				rtk::launch_rebase(dn, m.node_count2, m.node_base, dn4, m.n4_count, m.n4_base, m.tri_base, c->stream);
				RF_TRY(dm::last_launch_error());
			}
			else
			{
				nodes2 = m.bvh.nodes;
				for (rt::Node &nd : nodes2)
				{
					if (nd.count > 0)
						nd.left_first = (int)rt::make_entry(nd.left_first + (int)m.tri_base, nd.count, false);
					else if (nd.count < 0)
						nd.left_first = (int)rt::make_entry(nd.left_first + (int)m.node_base, nd.count, false);
				}
				nodes4c.resize(m.n4.size()), src.resize(4 * m.n4.size());
				for (size_t k = 0; k < m.n4.size(); k++)
				{
					rt::Node4 nd = m.n4[k];
					for (int j = 0; j < 4; j++)
					{
						const uint32_t e = nd.entry[j];
						if (e == rt::ENTRY_EMPTY)
							continue;
						if (e & rt::ENTRY_LEAF)
							nd.entry[j] = (e & ~rt::ENTRY_FIRST_MASK) | (((e & rt::ENTRY_FIRST_MASK) + m.tri_base) & rt::ENTRY_FIRST_MASK);
						else
							nd.entry[j] = e + m.n4_base;
					}
					nodes4c[k] = compress4(nd); // what the rays fetch: compressed (rt::pack_boxes4c)
					memcpy(&src[4 * k], nd.src, 16); // BVH2 node (BLAS-relative) behind each child box, for the refit
				}
				RF_TRY(dm::h2d(dn, nodes2.data(), nodes2.size() * sizeof(rt::Node), c->stream));
				RF_TRY(dm::h2d(dn4, nodes4c.data(), nodes4c.size() * sizeof(rt::Node4c), c->stream));
				RF_TRY(dm::h2d(dsrc, src.data(), src.size() * sizeof(uint32_t), c->stream));
				RF_TRY(dm::h2d(dtv, m.leaf_verts.data(), m.leaf_verts.size() * sizeof(f4), c->stream));
				RF_TRY(dm::sync(c->stream)); // the staging vectors are reused by the next mesh
			}
			RF_TRY(dm::h2d(c->d_tri_shade.as<rt::TriShade>() + m.shade_base, m.shade.data(), m.shade.size() * sizeof(rt::TriShade), c->stream));
		RF_TRY(dm::h2d(c->d_tri_uv.as<rt::TriUV>() + m.shade_base, m.uv.data(), m.uv.size() * sizeof(rt::TriUV), c->stream));
		}
		if (world.valid)
		{
			nodes4c.resize(world.n4.size());
			for (size_t k = 0; k < world.n4.size(); k++)
			{
				rt::Node4 nd = world.n4[k];
				for (int j = 0; j < 4; j++)
				{
					const uint32_t e = nd.entry[j];
					if (e == rt::ENTRY_EMPTY)
						continue;
					if (e & rt::ENTRY_LEAF)
						nd.entry[j] = (e & ~rt::ENTRY_FIRST_MASK) | (((e & rt::ENTRY_FIRST_MASK) + world.tri_base) & rt::ENTRY_FIRST_MASK);
					else
						nd.entry[j] = e + world.n4_base;
				}
				nodes4c[k] = compress4(nd);
			}
			RF_TRY(dm::h2d(c->d_nodes4.as<rt::Node4c>() + world.n4_base, nodes4c.data(), nodes4c.size() * sizeof(rt::Node4c), c->stream));
			RF_TRY(dm::h2d(c->d_tri_verts.as<f4>() + 3ull * world.tri_base, world.leaf_verts.data(), world.leaf_verts.size() * sizeof(f4), c->stream));
			world.resident = true;
		}
		RF_TRY(dm::sync(c->stream));
		// meshes that had been refit since their build are re-refit from their device vertices after the move
		for (auto &m : c->meshes)
		{
			if (!m.used)
				continue;
			if (m.resident && !m.dirty)
			{
				rtk::launch_refit(c->d_nodes.as<rt::Node>(), m.node_base, m.d_parents.as<int>(), m.node_count2,
								  c->d_tri_verts.as<f4>(), m.tri_base, m.d_verts.as<f4>(),
								  m.indexed ? m.d_indices.as<uint32_t>() : nullptr, (uint32_t)m.triCount,
								  m.d_flags.as<uint32_t>(), c->stream);
				rtk::launch_refresh4(c->d_nodes4.as<rt::Node4c>() + m.n4_base, c->d_nodes4_src.as<uint32_t>() + 4ull * m.n4_base,
									 m.n4_count, c->d_nodes.as<rt::Node>() + m.node_base, c->stream);
				if (m.posed) // the host copy of the shading records is the bind pose
					rtk::launch_skin_shade(c->d_tri_shade.as<rt::TriShade>() + m.shade_base, m.d_verts.as<f4>(),
										   m.d_vnormals.as<f4>(), m.indexed ? m.d_indices.as<uint32_t>() : nullptr,
										   (uint32_t)m.triCount, c->stream);
				RF_TRY(dm::last_launch_error());
			}
			m.resident = true, m.dirty = false;
		}
		RF_TRY(dm::sync(c->stream));
	}
	// ---- instances + TLAS ----
	std::vector<rt::Instance> inst(c->instances.size());
	std::vector<float> bmin, bmax;
	std::vector<uint32_t> live;
	for (size_t i = 0; i < c->instances.size(); i++)
	{
		rt::Instance &d = inst[i];
		memset(&d, 0, sizeof(d));
		const InstRec &in = c->instances[i];
		if (!in.used || in.mesh >= c->meshes.size() || !c->meshes[in.mesh].used)
			continue;
		const MeshRec &m = c->meshes[in.mesh];
		float inv[16];
		mat4_inverse(in.transform, inv);
		for (int r = 0; r < 3; r++)
			for (int col = 0; col < 4; col++)
				d.inv[4 * r + col] = inv[4 * col + r];
		for (int col = 0; col < 3; col++)
			for (int r = 0; r < 3; r++)
				d.nrm[4 * col + r] = in.normal[3 * col + r];
		d.node_base = m.node_base, d.tri_base = m.tri_base, d.shade_base = m.shade_base;
		d.root_entry = (m.device_built || m.bvh.nodes[0].count < 0)
						   ? rt::make_entry((int)m.n4_base, -1, false)
						   : rt::make_entry(m.bvh.nodes[0].left_first + (int)m.tri_base, m.bvh.nodes[0].count, false);
		float lo[3] = {1e34f, 1e34f, 1e34f}, hi[3] = {-1e34f, -1e34f, -1e34f};
		for (int k = 0; k < 8; k++)
		{
			const float p[3] = {k & 1 ? m.bounds_max[0] : m.bounds_min[0], k & 2 ? m.bounds_max[1] : m.bounds_min[1],
								k & 4 ? m.bounds_max[2] : m.bounds_min[2]};
			for (int r = 0; r < 3; r++)
			{
				const float w = in.transform[r] * p[0] + in.transform[4 + r] * p[1] + in.transform[8 + r] * p[2] + in.transform[12 + r];
				lo[r] = std::min(lo[r], w), hi[r] = std::max(hi[r], w);
			}
		}
		if (world.valid && world.member[i])
			continue; // its triangles hang in the world tree: no top-level leaf of its own
		for (int r = 0; r < 3; r++)
		{
			const float pad = 1e-4f + 1e-5f * std::max(std::fabs(lo[r]), std::fabs(hi[r]));
			bmin.push_back(lo[r] - pad), bmax.push_back(hi[r] + pad);
		}
		live.push_back((uint32_t)i);
	}
	if (world.valid)
	{
		for (int r = 0; r < 3; r++)
			bmin.push_back(world.bmin[r]), bmax.push_back(world.bmax[r]);
		live.push_back(WORLD_ID); // ONE top-level leaf for all of it, replaced by the entry of the world tree's root below
	}
	const uint32_t world_root_entry = !world.valid		 ? 0u
									  : !world.n4.empty() ? rt::make_entry((int)world.n4_base, -1, false)
														  : rt::make_entry((int)(world.root_first + world.tri_base), (int)world.root_count, false);
	bvh::Result tl;
	bvh::build(bmin.data(), bmax.data(), live.size(), 1, TLAS_DEPTH_LIMIT, tl);
	std::vector<uint32_t> tprims(std::max<size_t>(1, live.size()), 0u);
	for (size_t k = 0; k < live.size(); k++)
		tprims[k] = live[tl.order[k]];
	std::vector<rt::Node4> tl4;
	const bool tl_inner = bvh::collapse4(tl, true, tl4);
	{
		const int tlas_need = bvh::stack_need4(tl4);
		int blas_need = world.valid ? world.stack_need : 0;
		for (uint32_t i : live)
			if (i != WORLD_ID)
				blas_need = std::max(blas_need, c->meshes[c->instances[i].mesh].stack_need);
		// the packet form of the primary wave keeps ONE stack of PACKET_STACK entries per wave (kernels.hip: PacketStack)
		c->packet_ok = tlas_need + 1 + blas_need <= (int)rtk::PACKET_STACK;
		if (tlas_need + 1 + blas_need > rt::STACK_CAPACITY)
			return set_error(RFWHIP_ERR_UNSUPPORTED, "rfwhip_update: top-level tree over %zu instances needs %d traversal-stack "
							 "entries + 1 + %d for the deepest mesh (capacity %d)", live.size(), tlas_need, blas_need, rt::STACK_CAPACITY);
	}
	if (c->blas_nodes4 + tl4.size() > c->node4_capacity)
		return set_error(RFWHIP_ERR_STATE, "internal: TLAS does not fit behind the BLAS nodes");
	const uint32_t tlas_base = (uint32_t)c->blas_nodes4;
	// FLAT instances: an instance with the identity transform whose mesh no other instance uses is linked into the top-level
	// tree directly — its top-level leaf becomes the entry of its mesh's root, so a ray walks from the top-level nodes into the
	// mesh's nodes without the instance switch (ray transform and three divisions on the way in, the sentinel and the same on
	// the way out — and, in the wave kernels, a wait for the wave's next leaf phase each time).  The world-space ray IS the
	// object-space ray there (1 * x + 0 * y + 0 * z + 0 is exact), so nothing a ray computes changes; the triangles carry the
	// instance index the hit record needs (rtk::launch_stamp_instance).  Static world geometry is typically instanced this way.
	std::vector<uint8_t> flat(c->instances.size(), 0);
	{
		std::vector<uint32_t> uses(c->meshes.size(), 0u);
		for (uint32_t i : live)
			if (i != WORLD_ID)
				uses[c->instances[i].mesh]++;
		static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
		for (uint32_t i : live)
		{
			if (i == WORLD_ID)
				continue;
			bool id = uses[c->instances[i].mesh] == 1u && c->flat_instances != 0;
			for (int k = 0; k < 12 && id; k++)
				id = inst[i].inv[k] == ident[k];
			flat[i] = id ? 1 : 0;
		}
	}
	auto flat_entry = [&](uint32_t e) -> uint32_t { // a top-level leaf entry -> the mesh root's entry when its instance is flat
		if (e == rt::ENTRY_EMPTY || !(e & rt::ENTRY_LEAF))
			return e;
		const uint32_t ii = tprims[e & rt::ENTRY_FIRST_MASK];
		if (ii == WORLD_ID)
			return world_root_entry;
		return flat[ii] ? inst[ii].root_entry : e;
	};
	for (rt::Node4 &nd : tl4)
		for (int j = 0; j < 4; j++)
		{
			if (nd.entry[j] != rt::ENTRY_EMPTY && !(nd.entry[j] & rt::ENTRY_LEAF))
				nd.entry[j] += tlas_base; // inner: absolute node index; leaves index tlas_prims
			else
				nd.entry[j] = flat_entry(nd.entry[j]);
		}
	for (uint32_t i : live)
		if (i != WORLD_ID && flat[i])
		{
			const MeshRec &m = c->meshes[c->instances[i].mesh];
			rtk::launch_stamp_instance(c->d_tri_verts.as<f4>() + 3ull * m.tri_base, (uint32_t)m.triCount, i, c->stream);
		}
	RF_TRY(dm::last_launch_error());
	RF_TRY(c->d_instances.ensure(std::max<size_t>(1, inst.size()) * sizeof(rt::Instance)));
	RF_TRY(c->d_tlas_prims.ensure(tprims.size() * 4));
	RF_TRY(dm::h2d(c->d_instances.p, inst.data(), inst.size() * sizeof(rt::Instance), c->stream));
	std::vector<rt::Node4c> tl4c(tl4.size());
	for (size_t k = 0; k < tl4.size(); k++)
		tl4c[k] = compress4(tl4[k]);
	RF_TRY(dm::h2d(c->d_nodes4.as<rt::Node4c>() + tlas_base, tl4c.data(), tl4c.size() * sizeof(rt::Node4c), c->stream));
	std::vector<uint32_t> tprims_dev = tprims; // (the world tree's top-level slot is never looked up: its leaf entry was replaced above)
	for (uint32_t &tp : tprims_dev)
		if (tp == WORLD_ID)
			tp = 0u;
	RF_TRY(dm::h2d(c->d_tlas_prims.p, tprims_dev.data(), tprims_dev.size() * 4, c->stream));
	// the float form of every traversal node (mesh trees as rebuilt / refitted above + the top-level tree): one streaming pass
	// (only when the packet form of the primary wave — its one reader — can run: the table is twice the compressed one's size)
	c->packet_ok = c->packet_ok && (tlas_base + tl4c.size()) * sizeof(rt::Node4f) < (1ull << 32);
	c->nodes4_live = tlas_base + tl4c.size(), c->nodes4f_current = false;
	RF_TRY(ensure_nodes4f(c));
	RF_TRY(dm::last_launch_error());
	RF_TRY(dm::sync(c->stream));
	c->instance_count = (uint32_t)live.size();
	c->tlas_root_entry = live.empty() ? 0u
						 : tl_inner	   ? rt::make_entry((int)tlas_base, -1, true)
									   : flat_entry(rt::make_entry(tl.nodes[0].left_first, tl.nodes[0].count, true));

	rt::SceneView &sv = c->sv;
	sv.nodes4 = c->d_nodes4.as<rt::Node4c>(), sv.nodes4f = c->d_nodes4f.as<rt::Node4f>();
	sv.nodes = c->d_nodes.as<rt::Node>(), sv.tri_verts = c->d_tri_verts.as<f4>();
	sv.tri_shade = c->d_tri_shade.as<rt::TriShade>();
	sv.tri_uv = c->d_tri_uv.as<rt::TriUV>();
	sv.tlas_prims = c->d_tlas_prims.as<uint32_t>();
	sv.instances = c->d_instances.as<rt::Instance>();
	sv.tlas_root_entry = c->tlas_root_entry, sv.instance_count = c->instance_count;
	sv.materials = c->d_materials.as<rt::MaterialRec>(), sv.material_count = c->material_count;
	sv.textures = c->d_textures.as<rt::TexDesc>(), sv.texture_count = c->texture_count;
	sv.tex_u32 = c->d_tex_u32.as<uint32_t>(), sv.tex_f4 = c->d_tex_f4.as<f4>();
	sv.sky = c->d_sky.as<f4>(), sv.sky_w = c->sky_w, sv.sky_h = c->sky_h;
	sv.area = c->d_area.as<rt::AreaLight>(), sv.point = c->d_point.as<rt::PointLight>();
	sv.spot = c->d_spot.as<rt::SpotLight>(), sv.dir = c->d_dir.as<rt::DirectionalLight>();
	sv.n_area = c->lc.areaLightCount, sv.n_point = c->lc.pointLightCount, sv.n_spot = c->lc.spotLightCount;
	sv.n_dir = c->lc.directionalLightCount;
	c->scene_dirty = false;
	c->depth_stats_valid = false;
	c->shadow_packets_auto_on = true; // (another scene: the packet form of the depth-0 connection wave gets its chance again)
	return RFWHIP_OK;
}

// =================================================================================================================
// camera
// =================================================================================================================
extern "C" void rfwhip_camera_get_view(const rfwhip_camera *cam, rfwhip_camera_view *view)
{
	// Camera::get_view, Camera.cpp:74-88 + calculate_matrix :109-115
	const float dx = cam->direction[0], dy = cam->direction[1], dz = cam->direction[2];
	// x = normalize(cross(z, (0,1,0)))
	float rx = dy * 0.0f - 1.0f * dz, ry = dz * 0.0f - 0.0f * dx, rz = dx * 1.0f - 0.0f * dy;
	const float rl = 1.0f / sqrtf(rx * rx + ry * ry + rz * rz);
	rx *= rl, ry *= rl, rz *= rl;
	// y = cross(x, z)
	const float ux = ry * dz - dy * rz, uy = rz * dx - dz * rx, uz = rx * dy - dx * ry;
	const float pi = 3.14159265358979323846f;
	view->spreadAngle = (cam->FOV * pi / 180) / (float)cam->pixelCount[1];
	const float screenSize = tanf(cam->FOV / 2.0f / (180.0f / pi));
	const float cx = cam->position[0] + cam->focalDistance * dx, cy = cam->position[1] + cam->focalDistance * dy,
				cz = cam->position[2] + cam->focalDistance * dz;
	const float hx = ((screenSize * rx) * cam->focalDistance) * cam->aspectRatio,
				hy = ((screenSize * ry) * cam->focalDistance) * cam->aspectRatio,
				hz = ((screenSize * rz) * cam->focalDistance) * cam->aspectRatio;
	const float sv = screenSize * cam->focalDistance;
	const float vx = sv * ux, vy = sv * uy, vz = sv * uz;
	view->pos[0] = cam->position[0], view->pos[1] = cam->position[1], view->pos[2] = cam->position[2];
	view->p1[0] = (cx - hx) + vx, view->p1[1] = (cy - hy) + vy, view->p1[2] = (cz - hz) + vz;
	view->p2[0] = (cx + hx) + vx, view->p2[1] = (cy + hy) + vy, view->p2[2] = (cz + hz) + vz;
	view->p3[0] = (cx - hx) - vx, view->p3[1] = (cy - hy) - vy, view->p3[2] = (cz - hz) - vz;
	view->aperture = cam->aperture;
}

// =================================================================================================================
// render
// =================================================================================================================
static int sync_all(rfwhip_context *c);
// Path state per slot of the per-ray buffers: origin / direction / throughput x 2 depth parities, hit + instance x 2 (primary
// wave kept apart), shadow origin / direction / contribution x 2 depth parities = 232 B; radiance: shade + connections = 32 B
// per radiance slot.  rad_slots[k]: slots of radiance set k (a ring call uses set 0 only, as `ring` slices; a call cut into
// sub-batches double-buffers sets 0 / 1).
constexpr size_t WAVE_SLOT_BYTES = 2 * 3 * sizeof(f4) + 2 * (sizeof(f4) + 4) + 2 * 3 * sizeof(f4);
constexpr size_t RAD_SLOT_BYTES = 2 * sizeof(f4);
constexpr size_t HIT0_DONE_PAD = 512; // bytes beyond one per 64 slots in WaveView::hit0_done: every slice in flight begins on a byte of its own
constexpr double SHADOW_PACKET_MAX_BINS_PER_RUN = 8.0; // (shadow_packets = -1; measured: DESIGN.md §4 "Round 6")
static size_t wave_bytes_held(const rfwhip_context *c);
static int ensure_wave_buffers(rfwhip_context *c, size_t paths, size_t rad_slots0, size_t rad_slots1)
{
	const size_t rad_slots[2] = {rad_slots0, rad_slots1};
	if (paths <= c->wave_capacity && rad_slots0 <= c->rad_capacity[0] && rad_slots1 <= c->rad_capacity[1])
		return 0;
	RF_TRY(sync_all(c));
	for (int r = 0; r < rfwhip_context::MAX_RING; r++) // (everything was synchronised above)
		c->resolve_recorded[r] = false;
	// (a failed allocation leaves the capacities at 0: the next call lays everything out again)
	c->wave_capacity = 0, c->rad_capacity[0] = c->rad_capacity[1] = 0;
	const size_t b16 = paths * sizeof(f4);
	int rc = 0;
	for (int k = 0; k < 2 && !rc; k++)
		rc = c->d_org[k].ensure_exact(b16) || c->d_dir2[k].ensure_exact(b16) || c->d_thr[k].ensure_exact(b16) ||
			 c->d_sh_org[k].ensure_exact(b16) || c->d_sh_dir[k].ensure_exact(b16) || c->d_sh_rad[k].ensure_exact(b16) ||
			 c->d_rad[k].ensure_exact(rad_slots[k] * sizeof(f4)) || c->d_rad_nee[k].ensure_exact(rad_slots[k] * sizeof(f4));
	rc = rc || c->d_hit.ensure_exact(b16) || c->d_hit_inst.ensure_exact(paths * 4) || c->d_hit0.ensure_exact(b16) ||
		 c->d_hit0_inst.ensure_exact(paths * 4) || c->d_hit0_done.ensure_exact((paths >> 6) + HIT0_DONE_PAD);
	if (rc)
		return set_error(RFWHIP_ERR_HIP, "out of device memory for the path state: %zu slots x %zu B + %zu radiance slots x %zu B "
										 "(%.1f GB; lower spp, or ring / streams)", paths, WAVE_SLOT_BYTES, rad_slots0 + rad_slots1,
						 RAD_SLOT_BYTES, (paths * WAVE_SLOT_BYTES + (rad_slots0 + rad_slots1) * RAD_SLOT_BYTES) * 1e-9);
	c->wave_capacity = paths, c->rad_capacity[0] = rad_slots0, c->rad_capacity[1] = rad_slots1;
	return 0;
}
static size_t wave_bytes_held(const rfwhip_context *c)
{
	size_t n = c->d_hit.cap + c->d_hit_inst.cap + c->d_hit0.cap + c->d_hit0_inst.cap + c->d_hit0_done.cap;
	for (int k = 0; k < 2; k++)
		n += c->d_org[k].cap + c->d_dir2[k].cap + c->d_thr[k].cap + c->d_sh_org[k].cap + c->d_sh_dir[k].cap + c->d_sh_rad[k].cap +
			 c->d_rad[k].cap + c->d_rad_nee[k].cap;
	return n;
}

static dm::event_t *next_event(rfwhip_context *c)
{
	if (c->events_used == c->event_pool.size())
	{
		dm::event_t e;
		dm::event_create(&e);
		c->event_pool.push_back(e);
	}
	return &c->event_pool[c->events_used++];
}

struct StageTimer
{
	rfwhip_context *c;
	size_t ia = 0, ib = 0;
	bool on;
	void *stream;
	StageTimer(rfwhip_context *ctx, int family, int depth, void *st = nullptr)
		: c(ctx), on(ctx->stage_timing != 0), stream(st ? st : ctx->stream)
	{
		if (!on)
			return;
		next_event(c), ia = c->events_used - 1;
		next_event(c), ib = c->events_used - 1;
		dm::event_record(c->event_pool[ia], stream);
		fam = family, dep = depth;
	}
	int fam = 0, dep = 0;
	bool fused = false;
	void stop(int launches = 1)
	{
		if (!on)
			return;
		dm::event_record(c->event_pool[ib], stream);
		TimedSpan s;
		s.a = c->event_pool[ia], s.b = c->event_pool[ib], s.family = fam, s.depth = dep, s.fused = fused;
		c->spans.push_back(s);
		c->kernel_launches[fam] += (uint32_t)launches;
	}
};

static void fill_params(rfwhip_context *c, const rfwhip_camera *cam, rtk::Params &p)
{
	memset(&p, 0, sizeof(p));
	p.sc = c->sv;
	rt::WaveView &wv = p.wv;
	for (int k = 0; k < 2; k++)
		wv.org[k] = c->d_org[k].as<f4>(), wv.dir[k] = c->d_dir2[k].as<f4>(), wv.thr[k] = c->d_thr[k].as<f4>();
	wv.hit = c->d_hit.as<f4>(), wv.hit_inst = c->d_hit_inst.as<int>();
	wv.hit0 = c->d_hit0.as<f4>(), wv.hit0_inst = c->d_hit0_inst.as<int>(), wv.hit0_done = nullptr;
	wv.sh_org = c->d_sh_org[0].as<f4>(), wv.sh_dir = c->d_sh_dir[0].as<f4>(), wv.sh_rad = c->d_sh_rad[0].as<f4>();
	wv.rad = c->d_rad[0].as<f4>(), wv.rad_nee = nullptr, wv.acc = c->d_acc.as<f4>();
	wv.packet_rng = c->d_packet_rng.as<uint32_t>();
	wv.counters = c->d_counters.as<rt::WaveCounters>();
	p.cam.blue_noise = nullptr;
	if (cam)
	{
		rfwhip_camera_view v;
		rfwhip_camera_get_view(cam, &v);
		p.cam.pos = rt::f3{v.pos[0], v.pos[1], v.pos[2]};
		p.cam.p1 = rt::f3{v.p1[0], v.p1[1], v.p1[2]};
		p.cam.right = rt::f3{v.p2[0] - v.p1[0], v.p2[1] - v.p1[1], v.p2[2] - v.p1[2]};
		p.cam.up = rt::f3{v.p3[0] - v.p1[0], v.p3[1] - v.p1[1], v.p3[2] - v.p1[2]};
		p.cam.aperture = v.aperture, p.cam.spread_angle = v.spreadAngle, p.cam.clamp_value = cam->clampValue;
		p.cam.blue_noise = (c->sampler == 1 && c->have_blue_noise) ? c->d_blue_noise.as<uint32_t>() : nullptr;
	}
	p.fr = c->fr;
	p.fr.spp = (uint32_t)c->spp;
	p.fr.sample_base = c->samples_done;
	p.fr.probe_pixel = c->probe_y * c->W + c->probe_x;
	p.max_depth = (uint32_t)c->max_depth;
	p.parity_no_jitter = c->jitter == 1;
	// LDS top-of-tree cache: the first nodes (breadth-first top, bvh::collapse4) of the BLAS with the most nodes
	p.lds_first = 0, p.lds_count = 0;
	{
		// (... among the trees the rays walk: the meshes whose instances went into the world tree are not among them; that tree is)
		uint32_t big_base = 0, big_count = 0;
		for (size_t mi = 0; mi < c->meshes.size(); mi++)
		{
			const MeshRec &m = c->meshes[mi];
			const bool in_world = c->wtree.valid && mi < c->wtree.mesh_all.size() && c->wtree.mesh_all[mi];
			if (m.used && !in_world && m.n4_count > big_count)
				big_base = (uint32_t)m.n4_base, big_count = m.n4_count;
		}
		if (c->wtree.valid && c->wtree.resident && c->wtree.n4.size() > big_count)
			big_base = c->wtree.n4_base, big_count = (uint32_t)c->wtree.n4.size();
		const uint32_t cap = rtk::max_lds_nodes();
		const uint32_t want = c->lds_nodes < 0 ? cap : std::min<uint32_t>((uint32_t)c->lds_nodes, cap);
		if (big_count && want)
			p.lds_first = big_base, p.lds_count = std::min<uint32_t>(want, big_count);
	}
	p.refill = (uint32_t)c->refill & ((c->packet_ok && c->nodes4f_current) ? 15u : 7u);
	p.textured = c->textured ? 1u : 0u;
}

static int sync_all(rfwhip_context *c)
{
	for (int i = 0; i < rfwhip_context::MAX_SUB; i++)
	{
		if (c->sub_stream[i])
			RF_TRY(dm::sync(c->sub_stream[i]));
		if (c->conn_stream[i])
			RF_TRY(dm::sync(c->conn_stream[i]));
	}
	RF_TRY(dm::sync(c->stream));
	if (c->present_pending)
	{
		// a present enqueued on the CALLER's stream (rfwhip_read_local_framebuffer_stream) still reads the accumulator:
		// resize, cleanup and the scene setters must not free or rewrite anything under it
		RF_TRY(dm::event_sync(c->ev_present_out));
		c->present_pending = false;
	}
	return 0;
}

static int ensure_sub_batches(rfwhip_context *c, int subs)
{
	if (!c->events_ready)
	{
		RF_TRY(dm::event_create(&c->ev_prologue));
		for (int r = 0; r < rfwhip_context::MAX_RING; r++)
			RF_TRY(dm::event_create(&c->ev_resolve[r]));
		RF_TRY(dm::event_create(&c->ev_present_in));
		RF_TRY(dm::event_create(&c->ev_present_out));
		for (int i = 0; i < rfwhip_context::MAX_SUB; i++)
		{
			RF_TRY(dm::event_create(&c->ev_sub_done[i]));
			RF_TRY(dm::event_create(&c->ev_conn_last[i]));
			for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
			{
				RF_TRY(dm::event_create(&c->ev_shade[i][d]));
				RF_TRY(dm::event_create(&c->ev_conn[i][d]));
			}
		}
		c->events_ready = true;
	}
	for (int i = 0; i < subs; i++)
	{
		if (!c->sub_stream[i])
			RF_TRY(dm::stream_create(&c->sub_stream[i]));
		if (!c->conn_stream[i])
			RF_TRY(dm::stream_create(&c->conn_stream[i]));
		if (i > 0 && !c->d_counters_sub[i].p)
		{
			RF_TRY(c->d_counters_sub[i].ensure(sizeof(rt::WaveCounters)));
			RF_TRY(dm::zero(c->d_counters_sub[i].p, sizeof(rt::WaveCounters), c->stream));
			RF_TRY(dm::sync(c->stream));
		}
	}
	return 0;
}

// How many items the launches of depth d of a pt call should be SIZED for (grids only: every kernel reads its real count on the
// device and works through whatever it finds).  The host never reads a counter between bounces, so every launch used to get a grid
// for the primary count — for the deeper waves of a small call a chip-wide persistent grid whose waves find a few dozen rays each.
// Paths per primary at depth d: from the counts of the last frame this context waited for (rfwhip_get_stats; same scene, any spp),
// with a quarter of headroom; before there is one — and after every rfwhip_update(), until a frame of the new scene has been
// waited for — a half per depth.  shade = false: the traversal launch of depth d — extension rays of depth d and the shadow rays
// of depth d - 1 beside them; true: the shade launch of depth d.
static uint32_t depth_items(const rfwhip_context *c, uint32_t n, int d, bool shade)
{
	if (d <= 0)
		return n;
	const rfwhip_render_stats &st = c->stats;
	double ext = 1.0, prev = 1.0, shadow_per_path = 0.8;
	if (st.primaryCount > 0 && c->depth_stats_valid) // (after rfwhip_update the counts of the last frame describe another scene)
	{
		const double p = (double)st.primaryCount, r1 = (double)st.secondaryCount / p, rdeep = (double)st.deepCount / p;
		// (deepCount adds up every depth >= 2: taken as the bound for each of them)
		ext = d == 1 ? r1 : rdeep, prev = d == 1 ? 1.0 : (d == 2 ? r1 : rdeep);
		shadow_per_path = std::min(1.0, (double)st.shadowCount / std::max(1.0, p * (0.7 + r1 + rdeep))) + 0.1;
	}
	else
		for (int k = 0; k < d; k++)
			prev = ext, ext *= 0.5;
	const double want = shade ? ext : std::max(ext, prev * shadow_per_path);
	const double items = std::min(1.0, 1.25 * want + 0.02) * (double)n;
	return (uint32_t)std::max(1.0, items);
}

// One render call = `spp` samples per pixel.  The samples are cut into up to `streams` sub-batches that run the whole
// wavefront pipeline concurrently on their own HIP streams (own slices of the path buffers, own device counters):
// every traversal kernel ends with a tail in which a few long rays keep a handful of waves busy — a chain of ~150
// dependent node fetches at ~1 us each out of the Infinity Cache — and the other sub-batches' kernels fill the CUs
// during that tail.  The accumulator is touched by one resolve kernel over all sub-batches at the end (stream 0).
extern "C" int rfwhip_render(rfwhip_context *c, const rfwhip_camera *cam, int status)
{
	CTX_ENTER(c);
	if (!cam)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_render: null camera");
	if (!c->W || !c->H)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_render before rfwhip_init");
	if (c->scene_dirty)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_render: scene changed since the last rfwhip_update()");
	if (c->sampler == 1 && !c->have_blue_noise)
		return set_error(RFWHIP_ERR_STATE, "sampler=bluenoise needs rfwhip_set_blue_noise() first");
	if (c->max_depth + 2 > rt::MAX_DEPTH_SLOTS)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "max_depth %d too large", c->max_depth);
	const size_t paths = (size_t)c->fr.slots * (size_t)c->spp;
	if (paths >= (1ull << 31))
		return set_error(RFWHIP_ERR_UNSUPPORTED, "spp batch too large: %zu path slots (limit 2^31)", paths);
	if (c->integrator == 1 && !c->nodes4f_current && c->packet_ok && (c->refill & 8))
	{
		// (`refill` bit 3 came on after the last update: the packet form's float node table is written now, once)
		RF_TRY(ensure_nodes4f(c));
		RF_TRY(dm::sync(c->stream));
	}
	// Sub-batches: a call is cut into up to `streams` concurrent sub-batches only when that leaves every one at least
	// `sub_batch_paths` path slots and there are four of them: since the bounce / shadow / primary kernels keep their lanes
	// filled themselves, ONE sub-batch whose calls alternate between two sets of wave buffers / streams / counters (so that
	// consecutive calls overlap each other's kernel tails, connection waves on a side stream) is as fast or faster up to
	// ~200 M path slots (MI355X, 1080p terrain, Msamples/s as one sub-batch on the ring / cut into four — 16 spp: 2635 / 2366,
	// 32 spp: 2693 / 2515, 64 spp: 2741 / 2665; beyond that the ring's four sets of path state no longer fit comfortably).
	const long long want = (long long)paths / c->sub_batch_paths;
	int subs = (int)std::min<long long>(std::min(std::min(c->streams, (int)rfwhip_context::MAX_SUB), c->spp), want);
	if (subs < std::min(4, c->streams))
		subs = 1;
	const bool alternate = subs == 1;
	// sample groups of the slot layout (rt_core.h): the largest power of two <= sample_group at which every sub-batch begins
	uint32_t sgroup_log2 = 0;
	while ((2 << sgroup_log2) <= c->sample_group)
	{
		const int g = 2 << sgroup_log2;
		bool ok = c->spp % g == 0;
		for (int k = 1; k < subs && ok; k++)
			ok = (int)((long long)c->spp * k / subs) % g == 0;
		if (!ok)
			break;
		sgroup_log2++;
	}
	// ring of buffer sets: a single-sub-batch call uses set (call number mod ring) of everything — up to `ring` calls are
	// in flight, each a full-size launch chain; a call cut into sub-batches double-buffers its radiance only.  The ring is as
	// deep as the setting allows and the device's free memory holds (264 B of path state per slot and ring entry: 1080p at
	// 64 spp = 133 M slots = 35 GB per entry); when it does not fit, 2 and then 1 entries are tried before the call fails.
	// the extension / shadow queues are filled in blocks (rt_types.h: QUEUE_BLOCK): every sub-batch's slice of the per-ray
	// buffers has room for the void entries of its waves' last blocks behind the rays
	const size_t pad = rtk::queue_pad((uint32_t)std::min<size_t>(paths, 0xFFFFFFFFu));
	int ring = alternate ? c->ring : 2;
	for (;;)
	{
		const size_t wave_slots = alternate ? (size_t)ring * (paths + pad) : paths + (size_t)subs * pad;
		const size_t rad0 = alternate ? (size_t)ring * paths : paths, rad1 = alternate ? 0 : paths;
		const bool fits_as_is = wave_slots <= c->wave_capacity && rad0 <= c->rad_capacity[0] && rad1 <= c->rad_capacity[1];
		if (alternate && ring > 1 && !fits_as_is)
		{
			size_t free_b = 0, total_b = 0;
			dm::mem_info(&free_b, &total_b);
			const double avail = 0.94 * ((double)free_b + (double)wave_bytes_held(c));
			if ((double)wave_slots * WAVE_SLOT_BYTES + (double)(rad0 + rad1) * RAD_SLOT_BYTES > avail)
			{
				ring = ring > 2 ? 2 : 1;
				continue;
			}
		}
		if (ring != c->ring_active || paths != c->paths_active || subs != c->subs_active)
		{
			// the calls in flight lay their records out for another ring / batch size
			RF_TRY(sync_all(c));
			for (int r = 0; r < rfwhip_context::MAX_RING; r++)
				c->resolve_recorded[r] = false;
			c->ring_active = ring, c->paths_active = paths, c->subs_active = subs, c->call_slot = 0;
		}
		const int rc = ensure_wave_buffers(c, wave_slots, rad0, rad1);
		if (rc && alternate && ring > 1) // (the estimate was too optimistic: another process, fragmentation)
		{
			ring = ring > 2 ? 2 : 1;
			continue;
		}
		RF_TRY(rc);
		break;
	}
	RF_TRY(ensure_sub_batches(c, alternate ? ring : subs));
	const bool pipelined = c->render_pending; // the caller enqueues calls without waiting for them in between
	if (!c->render_pending)
	{
		c->render_t0 = std::chrono::steady_clock::now();
		c->render_pending = true;
	}
	void *s0 = c->stream;
	// ---- prologue on stream 0 ----
	if (c->present_pending)
	{
		// a present enqueued on the caller's stream (rfwhip_read_local_framebuffer_stream) still reads the accumulator
		RF_TRY(dm::stream_wait_event(s0, c->ev_present_out));
		c->present_pending = false;
	}
	if (status == RFWHIP_RESET)
	{
		RF_TRY(dm::zero(c->d_acc.p, (size_t)c->fr.local_rows * c->W * sizeof(f4), s0));
		c->samples_done = 0;
	}
	const uint32_t packets = (c->W / 4u) * (c->H / 2u);
	if (c->integrator == 0 && c->jitter == 0)
	{
		// per-packet xor128 states for the whole batch (global packet order => image independent of world)
		StageTimer tg(c, KF_GENERATE, -1);
		if (!c->jump_table_uploaded)
		{
			RF_TRY(c->d_jump_table.ensure(c->jump_table.size() * 4));
			RF_TRY(dm::h2d(c->d_jump_table.p, c->jump_table.data(), c->jump_table.size() * 4, s0));
			c->jump_table_uploaded = true;
		}
		if ((size_t)std::max(1u, packets) * c->spp * 16 > c->d_packet_rng.cap)
			RF_TRY(sync_all(c));
		RF_TRY(c->d_packet_rng.ensure((size_t)std::max(1u, packets) * c->spp * 16));
		if (packets)
			rtk::launch_rng_states(c->d_packet_rng.as<uint32_t>(), c->rng_state, c->d_jump_table.as<uint32_t>(), packets,
								   (uint32_t)c->spp, s0);
		tg.stop();
		xor128_jump(c->jump_table, c->rng_state, (unsigned long long)packets * 32ull * (unsigned long long)c->spp);
	}
	const bool rng_prologue = c->integrator == 0 && c->jitter == 0;
	if (rng_prologue)
		RF_TRY(dm::event_record(c->ev_prologue, s0));
	const bool count = c->count_traversal != 0;
	const uint32_t row_group = std::max(1u, (c->fr.tiles_x << sgroup_log2) / 4u); // primary wave: one row of 8x8 tiles per XCD group
	const uint32_t par = c->call_slot;							 // buffer set of this call
	const uint32_t prev = (par + (uint32_t)ring - 1u) % (uint32_t)ring; // ... and of the previous one
	const bool connect = c->integrator == 1 && total_light_count(c) > 0 && c->max_depth > 0;
	// Connection waves on a second stream hide the kernel tails of a call that runs alone (1 spp frames, wait after every
	// call: 1.56 -> 1.43 ms).  When the caller pipelines its calls, the ring of buffer sets already keeps up to four launch
	// chains in flight and the extra kernels only evict each other's working sets (1 spp: 1.17 ms without, 1.38 ms with the
	// side stream; 16 spp: 13.2 / 14.1 ms), and the same holds beside other sub-batches (4 sub-batches, 128 spp: -9 %).
	// extension rays of depth d + 1 and shadow rays of depth d in one launch (both in persistent-lane form): one tail per depth
	const bool fused = connect && c->fuse != 0 && (c->refill & 3) == 3 && c->overlap != 1;
	const bool side = connect && !fused && (c->overlap == 1 || (c->overlap < 0 && subs == 1 && !pipelined));
	rtk::Params base;
	fill_params(c, cam, base);
	set_sample_group(base.fr, sgroup_log2);
	c->sgroup_last = sgroup_log2;
	base.wv.rad = alternate ? c->d_rad[0].as<f4>() + paths * par : c->d_rad[par].as<f4>();
	// the connections always add into their own buffer, on a side stream or not: the image is then bit-identical whichever way
	// a call is scheduled (e.g. the ranks of a strip split against the single-rank image)
	base.wv.rad_nee = !connect ? nullptr : (alternate ? c->d_rad_nee[0].as<f4>() + paths * par : c->d_rad_nee[par].as<f4>());
	// ---- sub-batches: each on its own stream; nothing here waits for the previous call's resolve ----
	const int first_slot = alternate ? (int)par : 0;
	for (int k = 0; k < subs; k++)
	{
		const int i = first_slot + k; // which set of streams / counters / buffer slices
		void *s = c->sub_stream[i], *sc = side ? c->conn_stream[i] : c->sub_stream[i];
		const uint32_t s_begin = (uint32_t)((long long)c->spp * k / subs), s_end = (uint32_t)((long long)c->spp * (k + 1) / subs);
		const uint32_t spp_i = s_end - s_begin;
		if (!spp_i)
			continue;
		if (rng_prologue)
			RF_TRY(dm::stream_wait_event(s, c->ev_prologue));
		if (c->resolve_recorded[par]) // the resolve of the call `ring` calls ago read this radiance set
			RF_TRY(dm::stream_wait_event(s, c->ev_resolve[par]));
		// Several sub-batches start together, behind the previous call's resolve: they render the same pixels, and while they
		// run in step their kernels share BVH nodes in the L2s (letting them drift apart costs 2-4 %)
		if (subs > 1 && c->resolve_recorded[prev])
			RF_TRY(dm::stream_wait_event(s, c->ev_resolve[prev]));
		if (c->conn_used[i]) // the previous call's connection waves still use this sub-batch's counters and shadow buffers
			RF_TRY(dm::stream_wait_event(s, c->ev_conn_last[i]));
		rtk::Params p = base;
		const size_t off_rad = (size_t)c->fr.slots * s_begin; // this sub-batch's slice of the radiance buffers of the call
		const size_t off = alternate ? (paths + pad) * par : off_rad + (size_t)k * pad; // and of every per-ray buffer (padded)
		for (int q = 0; q < 2; q++)
			p.wv.org[q] += off, p.wv.dir[q] += off, p.wv.thr[q] += off;
		p.wv.hit += off, p.wv.hit_inst += off, p.wv.hit0 += off, p.wv.hit0_inst += off;
		p.wv.rad += off_rad;
		if (p.wv.rad_nee)
			p.wv.rad_nee += off_rad;
		if (p.wv.packet_rng)
			p.wv.packet_rng += (size_t)s_begin * packets * 4;
		if (i > 0)
			p.wv.counters = c->d_counters_sub[i].as<rt::WaveCounters>();
		p.fr.spp = spp_i;
		p.fr.sample_base = c->samples_done + s_begin;
		const uint32_t n = c->fr.slots * spp_i;
		// packet form of the depth-0 connection wave: where the packet traversal can run (float node table, trees within its stack),
		// a wave's shadow rays are neighbours (sample groups of >= 8) and the batch's slots leave room for the light's bin
		p.fr.shadow_bins = 0u;
		if (c->integrator == 1 && connect && (c->shadow_packets > 0 || (c->shadow_packets < 0 && c->shadow_packets_auto_on)) && (p.refill & 8u) &&
			sgroup_log2 >= 3u && !side)
			for (uint32_t b = rt::SHADOW_BIN_BITS; b >= 1u && !p.fr.shadow_bins; b--) // (as many bin bits as the batch's slots leave room for)
				if ((unsigned long long)n <= (1ull << rt::shadow_slot_bits(b)))
					p.fr.shadow_bins = b;
		// flags of the 64-slot groups the packet form of the primary wave finishes (this slice's own bytes: slices begin anywhere)
		p.wv.hit0_done = c->integrator == 1 && c->group_flags && rtk::primary_packet_form(p, n) && (size_t)i < HIT0_DONE_PAD / 2
							 ? c->d_hit0_done.as<unsigned char>() + (off >> 6) + (size_t)i
							 : nullptr;
		rtk::launch_init_counters(p.wv.counters, n, s);
		uint32_t queue = 0; // every traversal launch pulls from its own chunk queue
		bool conn_now = false;
		bool sp_side_pending = false; // k_shadow_packet of this sub-batch runs on its connection stream and nobody has waited for it yet
		if (c->integrator == 0)
		{
			p.depth = 0, p.group = row_group, p.queue = queue++;
			StageTimer te(c, KF_EXTEND, 0, s);
			rtk::launch_extend(p, rtk::GEN_PARITY, count, n, s);
			te.stop();
			p.queue = queue++;
			StageTimer ts(c, KF_SHADE, -1, s);
			rtk::launch_shade_parity(p, count, n, s);
			ts.stop();
		}
		else
		{
			rtk::Params pa = p; // fused: the connection wave of the previous depth, launched together with this depth's extension wave
			bool pa_pending = false;
			for (int d = 0; d <= c->max_depth; d++)
			{
				p.depth = (uint32_t)d;
				p.group = d == 0 ? row_group : 16u;
				p.queue = queue++;
				// shadow-ray buffers of this depth's parity: shade(d) writes them, connect(d) reads them
				p.wv.sh_org = c->d_sh_org[d & 1].as<f4>() + off, p.wv.sh_dir = c->d_sh_dir[d & 1].as<f4>() + off;
				p.wv.sh_rad = c->d_sh_rad[d & 1].as<f4>() + off;
				StageTimer te(c, KF_EXTEND, d, s);
				// (grid sizes of the deeper launches: from the paths EXPECTED there, not from the primary count — depth_items)
				const uint32_t n_ext = depth_items(c, n, d, false), n_shade = depth_items(c, n, d, true);
				te.fused = pa_pending;
				if (pa_pending && sp_side_pending) // (the depth-0 connection wave on the side stream zeroes slots this launch's shadow rays add into)
				{
					RF_TRY(dm::stream_wait_event(s, c->ev_conn[i][0]));
					sp_side_pending = false;
				}
				if (pa_pending)
					rtk::launch_trace_fused(p, pa, count, n_ext, s); // extension rays of depth d + shadow rays of depth d - 1
				else
					rtk::launch_extend(p, d == 0 ? rtk::GEN_PT : rtk::GEN_BUFFER, count, d == 0 ? n : n_ext, s);
				pa_pending = false;
				te.stop();
				if (side && d >= 2 && d < c->max_depth) // connect(d - 2) read the buffers shade(d) is about to write
					RF_TRY(dm::stream_wait_event(s, c->ev_conn[i][d - 2]));
				StageTimer ts(c, KF_SHADE, -1, s);
				rtk::launch_shade_pt(p, n_shade, s);
				ts.stop();
				// The connection wave of depth d runs beside extend / shade of depth d + 1 on its own stream (it adds into
				// rad_nee, they into rad).  The connections of the last shade call are never traced
				// (CUDART/src/Context.cpp:109-120): the shade kernel does not emit them, and no wave is launched for them.
				if (connect && d < c->max_depth)
				{
					if (d == 0 && p.fr.shadow_bins)
					{
						// the connections of the primary vertices, sorted by light, as packets (their own launch: the extension wave of
						// depth 1 then runs without them)
						p.group = 16u, p.queue = queue++;
						void *const sp_stream = c->shadow_side ? c->conn_stream[i] : s;
						if (c->shadow_side)
						{
							// beside the extension wave of depth 1, on the sub-batch's connection stream; the connection wave of depth 1
							// (which adds into the slots this one zeroes) waits for it below
							RF_TRY(dm::event_record(c->ev_shade[i][0], s));
							RF_TRY(dm::stream_wait_event(sp_stream, c->ev_shade[i][0]));
						}
						StageTimer tc(c, KF_CONNECT, -1, sp_stream);
						rtk::launch_shadow_packets(p, count, depth_items(c, n, 1, false), sp_stream);
						tc.stop();
						if (c->shadow_side)
						{
							RF_TRY(dm::event_record(c->ev_conn[i][0], sp_stream));
							sp_side_pending = true;
						}
						continue;
					}
					if (fused)
					{
						pa = p, pa.group = 16u, pa.queue = queue++;
						pa_pending = true; // (d < max_depth: the next depth's extension wave takes it along)
						continue;
					}
					if (side)
					{
						RF_TRY(dm::event_record(c->ev_shade[i][d], s));
						RF_TRY(dm::stream_wait_event(sc, c->ev_shade[i][d]));
					}
					p.group = 16u, p.queue = queue++;
					StageTimer tc(c, KF_CONNECT, -1, sc);
					rtk::launch_connect(p, count, depth_items(c, n, d + 1, false), sc);
					tc.stop();
					if (side)
					{
						RF_TRY(dm::event_record(c->ev_conn[i][d], sc));
						conn_now = true;
					}
				}
			}
		}
		if (sp_side_pending) // (nothing deeper waited for it: the sub-batch is done when it is)
			RF_TRY(dm::stream_wait_event(s, c->ev_conn[i][0]));
		RF_TRY(dm::event_record(c->ev_sub_done[i], s));
		if (conn_now)
			RF_TRY(dm::event_record(c->ev_conn_last[i], sc));
		c->conn_used[i] = conn_now;
	}
	// ---- epilogue on the main stream: one resolve over every sample of the call ----
	for (int i = first_slot; i < first_slot + subs; i++)
	{
		RF_TRY(dm::stream_wait_event(s0, c->ev_sub_done[i]));
		if (c->conn_used[i])
			RF_TRY(dm::stream_wait_event(s0, c->ev_conn_last[i]));
	}
	{
		StageTimer tf(c, KF_FINALIZE, -1);
		rtk::launch_resolve(base, s0);
		tf.stop();
	}
	RF_TRY(dm::event_record(c->ev_resolve[par], s0));
	c->resolve_recorded[par] = true;
	c->call_slot = (par + 1u) % (uint32_t)ring;
	RF_TRY(dm::last_launch_error());
	c->subs_last = subs, c->subs_first = first_slot;
	c->last_wave_off = alternate ? (paths + pad) * par : 0;
	c->samples_done += (uint32_t)c->spp;
	c->totals.samples += (uint64_t)c->W * c->H * (uint64_t)c->spp / (uint64_t)c->world;
	return RFWHIP_OK;
}

extern "C" int rfwhip_wait(rfwhip_context *c)
{
	CTX_ENTER(c);
	// every render call ends with its resolve on the main stream, behind the events of all its sub-batch and connection
	// streams: the main stream alone tells when the enqueued frames are done (one synchronisation instead of up to 17)
	RF_TRY(dm::sync(c->stream));
	if (c->present_pending)
	{
		RF_TRY(dm::event_sync(c->ev_present_out));
		c->present_pending = false;
	}
	// wave counters of the last frame, summed over its sub-batches.  A sub-batch's connection wave of depth d ran only
	// if its depth d + 1 had extension rays (k_connect / k_trace_stream: connection_count)
	rt::WaveCounters wc;
	RF_TRY(dm::d2h(&wc, c->subs_first == 0 ? c->d_counters.p : c->d_counters_sub[c->subs_first].p, sizeof(wc), c->stream));
	uint32_t stack_overflow = wc.stack_overflow;
	const rt::WaveCounters wc0 = wc; // (as read: the sums below fold the other sub-batches' counts into wc)
	std::vector<rt::WaveCounters> more;
	for (int d = 0; d + 1 < rt::MAX_DEPTH_SLOTS; d++)
		if (!wc.ext[d + 1])
			wc.shadow[d] = 0;
	for (int i = c->subs_first + 1; i < c->subs_first + c->subs_last; i++)
	{
		rt::WaveCounters w2;
		RF_TRY(dm::d2h(&w2, c->d_counters_sub[i].p, sizeof(w2), c->stream));
		more.push_back(w2);
		stack_overflow += w2.stack_overflow;
		for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
		{
			wc.ext[d] += w2.ext[d];
			if (d + 1 < rt::MAX_DEPTH_SLOTS && w2.ext[d + 1])
				wc.shadow[d] += w2.shadow[d];
		}
	}
	// k_trace_fused: the shadow rays' share of each depth's launch, from the workgroups' tick sums since the last wait (over the
	// counter sets of the last frame's sub-batches: the frames of a pipelined series render the same scene)
	double shadow_share[rt::MAX_DEPTH_SLOTS];
	{
		unsigned long long sums[rt::MAX_DEPTH_SLOTS][2] = {};
		auto fold = [&](int slot, const rt::WaveCounters &w) {
			for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
				for (int k = 0; k < 2; k++)
				{
					sums[d][k] += w.fused_ticks[d][k] - c->fused_ticks_seen[slot][d][k];
					c->fused_ticks_seen[slot][d][k] = w.fused_ticks[d][k];
				}
		};
		fold(c->subs_first, wc0);
		for (size_t k = 0; k < more.size(); k++)
			fold(c->subs_first + 1 + (int)k, more[k]);
		// k_shadow_packet: light bins per sorted run of the frames since the last wait.  Many bins per run = the first vertices of
		// neighbouring pixels do not agree about their lights: packets of 64 then mix many directions.  Measured: terrain 3.6 bins per
		// run, + 8 %; atrium (an interior lit from all sides) 5.5, + 0.6 %; the threshold, half of the 16 bins, is beyond what was measured
		unsigned long long runs = 0, bins = 0;
		auto fold_sp = [&](int slot, const rt::WaveCounters &w) {
			runs += w.sp_runs - c->sp_seen[slot][0], bins += w.sp_bins - c->sp_seen[slot][1];
			c->sp_seen[slot][0] = w.sp_runs, c->sp_seen[slot][1] = w.sp_bins;
		};
		fold_sp(c->subs_first, wc0);
		for (size_t k = 0; k < more.size(); k++)
			fold_sp(c->subs_first + 1 + (int)k, more[k]);
		if (runs)
		{
			c->shadow_bins_per_run = (double)bins / (double)runs;
			c->shadow_packets_auto_on = c->shadow_bins_per_run <= SHADOW_PACKET_MAX_BINS_PER_RUN;
		}
		for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
		{
			const double tot = (double)sums[d][0] + (double)sums[d][1];
			shadow_share[d] = tot > 0.0 ? (double)sums[d][1] / tot : 0.5;
		}
	}
	if (wc.probe_valid)
		c->probe_inst = wc.probe_inst, c->probe_prim = wc.probe_prim, c->probe_dist = wc.probe_dist;
	rfwhip_render_stats &st = c->stats;
	c->depth_stats_valid = true;
	const float anim = st.animationTime;
	memset(&st, 0, sizeof(st));
	st.animationTime = anim;
	{
		// ext[0] counts path slots (8x8 tiles, padded at the image border); primary RAYS exist for real pixels only
		uint32_t owned_rows = 0;
		for (uint32_t y = 0; y < c->H; y++)
			owned_rows += rt::strip_owner(y / rt::STRIP_ROWS, (uint32_t)c->world) == (uint32_t)c->rank;
		const uint32_t samples = c->fr.slots ? wc.ext[0] / c->fr.slots : 0u;
		st.primaryCount = owned_rows * c->W * samples;
	}
	st.secondaryCount = wc.ext[1];
	for (int d = 2; d < rt::MAX_DEPTH_SLOTS; d++)
		st.deepCount += wc.ext[d];
	for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
		st.shadowCount += wc.shadow[d];
	for (const TimedSpan &sp : c->spans)
	{
		float ms = dm::event_ms(sp.a, sp.b);
		if (sp.fused && sp.depth >= 0 && sp.depth < rt::MAX_DEPTH_SLOTS)
		{
			// (one launch, two stages: the shadow rays' share goes where a host of the reference looks for it — shadowTime, context.h:63)
			const float shadow_ms = ms * (float)shadow_share[sp.depth];
			st.shadowTime += shadow_ms, c->kernel_ms[KF_CONNECT] += shadow_ms;
			ms -= shadow_ms;
		}
		c->kernel_ms[sp.family] += ms;
		switch (sp.family)
		{
		case KF_EXTEND:
			if (sp.depth == 0)
				st.primaryTime += ms;
			else if (sp.depth == 1)
				st.secondaryTime += ms;
			else
				st.deepTime += ms;
			break;
		case KF_GENERATE:
			st.primaryTime += ms;
			break;
		case KF_SHADE:
			st.shadeTime += ms;
			break;
		case KF_CONNECT:
			st.shadowTime += ms;
			break;
		case KF_FINALIZE:
			st.finalizeTime += ms;
			break;
		default:
			break;
		}
	}
	c->spans.clear();
	c->events_used = 0;
	if (c->render_pending)
	{
		st.renderTime = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c->render_t0).count();
		c->render_pending = false;
	}
	// (reported last: the bookkeeping above is complete, so the context stays usable after the error)
	if (stack_overflow)
		return set_error(RFWHIP_ERR_STATE, "traversal stack overflow: %u entries dropped in the last frame (the image is wrong)", stack_overflow);
	return RFWHIP_OK;
}

// =================================================================================================================
// present
// =================================================================================================================
extern "C" uint32_t rfwhip_local_rows(const rfwhip_context *c) { return c ? local_rows_of(c) : 0u; }

static int present(rfwhip_context *c, f4 *dst_device, int full)
{
	rtk::Params p;
	fill_params(c, nullptr, p);
	const float scale = c->samples_done ? 1.0f / (float)c->samples_done : 0.0f;
	rtk::launch_present(p, dst_device, scale, full, c->stream);
	RF_TRY(dm::last_launch_error());
	return 0;
}

extern "C" int rfwhip_read_framebuffer_device(rfwhip_context *c, void *rgba_device)
{
	CTX_ENTER(c);
	if (!rgba_device)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null destination");
	if (c->world != 1)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_read_framebuffer*: this rank owns 1/%d of the image; gather local "
											"framebuffers and call rfwhip_deinterleave_device", c->world);
	if (!c->W)
		return set_error(RFWHIP_ERR_STATE, "no render target");
	RF_TRY(present(c, (f4 *)rgba_device, 1));
	return dm::sync(c->stream);
}

extern "C" int rfwhip_read_framebuffer(rfwhip_context *c, float *rgba_host)
{
	CTX_ENTER(c);
	if (!rgba_host)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null destination");
	const size_t bytes = (size_t)c->W * c->H * sizeof(f4);
	RF_TRY(c->d_present.ensure(bytes));
	RF_TRY(dm::zero(c->d_present.p, bytes, c->stream));
	RF_TRY(rfwhip_read_framebuffer_device(c, c->d_present.p));
	return dm::d2h(rgba_host, c->d_present.p, bytes, c->stream);
}

extern "C" int rfwhip_read_local_framebuffer_device(rfwhip_context *c, void *rgba_device)
{
	CTX_ENTER(c);
	if (!rgba_device)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null destination");
	if (!c->W)
		return set_error(RFWHIP_ERR_STATE, "no render target");
	RF_TRY(present(c, (f4 *)rgba_device, 0));
	return dm::sync(c->stream);
}

// Stream-ordered variants: nothing blocks the host.  The present runs on the CALLER's stream after everything this
// context has enqueued so far; the context's next render waits (on the device) until that present has read the
// accumulator.  With torch.distributed the caller passes torch's current stream, so the RCCL gather that follows is
// ordered behind the present by plain stream order, and the next frame's kernels overlap the gather.
extern "C" int rfwhip_read_local_framebuffer_stream(rfwhip_context *c, void *rgba_device, void *hip_stream)
{
	CTX_ENTER(c);
	if (!rgba_device)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null destination");
	if (!c->W)
		return set_error(RFWHIP_ERR_STATE, "no render target");
	RF_TRY(ensure_sub_batches(c, 1)); // creates the hand-off events
	RF_TRY(dm::event_record(c->ev_present_in, c->stream));
	RF_TRY(dm::stream_wait_event(hip_stream, c->ev_present_in));
	rtk::Params p;
	fill_params(c, nullptr, p);
	const float scale = c->samples_done ? 1.0f / (float)c->samples_done : 0.0f;
	rtk::launch_present(p, (f4 *)rgba_device, scale, 0, hip_stream);
	RF_TRY(dm::last_launch_error());
	RF_TRY(dm::event_record(c->ev_present_out, hip_stream));
	c->present_pending = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_deinterleave_stream(rfwhip_context *c, const void *gathered_device, void *rgba_device, void *hip_stream)
{
	CTX_ENTER(c);
	if (!gathered_device || !rgba_device)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null buffer");
	rtk::launch_deinterleave((const f4 *)gathered_device, (f4 *)rgba_device, c->W, c->H, local_rows_of(c), (uint32_t)c->world, hip_stream);
	return dm::last_launch_error();
}

extern "C" int rfwhip_get_placement(rfwhip_context *c, int *device_ordinal, int *rank, int *world)
{
	CTX_ENTER(c);
	if (device_ordinal)
		*device_ordinal = c->device;
	if (rank)
		*rank = c->rank;
	if (world)
		*world = c->world;
	return RFWHIP_OK;
}

extern "C" int rfwhip_get_target_size(rfwhip_context *c, uint32_t *width, uint32_t *height)
{
	CTX_ENTER(c);
	if (width)
		*width = c->W;
	if (height)
		*height = c->H;
	return RFWHIP_OK;
}

extern "C" int rfwhip_deinterleave_device(rfwhip_context *c, const void *gathered_device, void *rgba_device)
{
	CTX_ENTER(c);
	if (!gathered_device || !rgba_device)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null buffer");
	rtk::launch_deinterleave((const f4 *)gathered_device, (f4 *)rgba_device, c->W, c->H, local_rows_of(c), (uint32_t)c->world, c->stream);
	RF_TRY(dm::last_launch_error());
	return dm::sync(c->stream);
}

// =================================================================================================================
// probe / stats / settings / hooks
// =================================================================================================================
extern "C" int rfwhip_set_probe_index(rfwhip_context *c, uint32_t x, uint32_t y)
{
	CTX_ENTER(c);
	c->probe_x = x, c->probe_y = y;
	return RFWHIP_OK;
}
extern "C" int rfwhip_get_probe_results(rfwhip_context *c, uint32_t *inst, uint32_t *prim, float *dist)
{
	CTX_ENTER(c);
	if (inst)
		*inst = c->probe_inst;
	if (prim)
		*prim = c->probe_prim;
	if (dist)
		*dist = c->probe_dist;
	return RFWHIP_OK;
}
extern "C" int rfwhip_get_stats(rfwhip_context *c, rfwhip_render_stats *stats)
{
	CTX_ENTER(c);
	if (!stats)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null stats");
	*stats = c->stats;
	return RFWHIP_OK;
}

static const char *const k_setting_keys[] = {"integrator", "spp", "max_depth", "jitter", "stage_timing", "count_traversal", "lds_nodes", "refill", "streams", "sampler", "builder", "overlap", "sub_batch_paths", "ring", "sample_group", "flat_instances", "flatten_bytes", "fuse", "shadow_packets", "shadow_side", "group_flags"};

extern "C" int rfwhip_set_setting(rfwhip_context *c, const char *key, const char *value)
{
	CTX_ENTER(c);
	if (!key || !value)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null key/value");
	const std::string k(key), v(value);
	if (k == "integrator")
	{
		if (v == "parity")
			c->integrator = 0;
		else if (v == "pt")
			c->integrator = 1;
		else
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "integrator must be \"parity\" or \"pt\", got \"%s\"", value);
	}
	else if (k == "spp")
	{
		const int n = atoi(value);
		if (n < 1 || n > 4096)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "spp must be in [1, 4096]");
		c->spp = n;
	}
	else if (k == "max_depth")
	{
		const int n = atoi(value);
		if (n < 0 || n + 2 > rt::MAX_DEPTH_SLOTS)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "max_depth must be in [0, %d]", rt::MAX_DEPTH_SLOTS - 2);
		c->max_depth = n;
	}
	else if (k == "jitter")
	{
		if (v == "xor128")
			c->jitter = 0;
		else if (v == "center")
			c->jitter = 1;
		else
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "jitter must be \"xor128\" or \"center\"");
	}
	else if (k == "builder")
	{
		if (!strcmp(value, "host"))
			c->builder = 0;
		else if (!strcmp(value, "device"))
			c->builder = 1;
		else
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "builder must be \"host\" or \"device\"");
	}
	else if (k == "sampler")
	{
		if (!strcmp(value, "hash"))
			c->sampler = 0;
		else if (!strcmp(value, "bluenoise"))
			c->sampler = 1;
		else
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "sampler must be \"hash\" or \"bluenoise\"");
	}
	else if (k == "stage_timing")
		c->stage_timing = atoi(value) != 0;
	else if (k == "count_traversal")
		c->count_traversal = atoi(value) != 0;
	else if (k == "lds_nodes")
		c->lds_nodes = std::max(-1, atoi(value));
	else if (k == "refill")
		c->refill = atoi(value) & 15;
	else if (k == "arm")
	{
		// (self-arming primary kernels: round 4's switch, removed with the variant in round 5 — accepted and ignored, like refill bit 2)
	}
	else if (k == "fuse")
		c->fuse = atoi(value) != 0;
	else if (k == "shadow_side")
		c->shadow_side = atoi(value) != 0;
	else if (k == "group_flags")
		c->group_flags = atoi(value) != 0;
	else if (k == "shadow_packets")
	{
		const int v = atoi(value);
		c->shadow_packets = v < 0 ? -1 : (v != 0);
		c->shadow_packets_auto_on = true;
	}
	else if (k == "flatten_bytes")
	{
		c->flatten_bytes = std::max(0ll, atoll(value));
		c->scene_dirty = true;
	}
	else if (k == "sub_batch_paths")
	{
		const long long n = atoll(value);
		if (n < 1)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "sub_batch_paths must be >= 1");
		c->sub_batch_paths = n;
	}
	else if (k == "ring")
	{
		const int n = atoi(value);
		if (n < 1 || n > rfwhip_context::MAX_RING)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "ring must be in [1, %d]", (int)rfwhip_context::MAX_RING);
		c->ring = n;
	}
	else if (k == "flat_instances")
	{
		c->flat_instances = atoi(value) != 0;
		c->scene_dirty = true; // takes effect with the next rfwhip_update()
	}
	else if (k == "sample_group")
	{
		const int n = atoi(value);
		if (n < 1 || n > 64 || (n & (n - 1)))
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "sample_group must be a power of two in [1, 64]");
		c->sample_group = n;
	}
	else if (k == "overlap")
	{
		const int n = atoi(value);
		if (n < -1 || n > 1)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "overlap must be -1 (by launch size), 0 or 1");
		c->overlap = n;
	}
	else if (k == "streams")
	{
		const int n = atoi(value);
		if (n < 1 || n > rfwhip_context::MAX_SUB)
			return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "streams must be in [1, %d]", (int)rfwhip_context::MAX_SUB);
		c->streams = n;
	}
	else
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "unknown setting \"%s\"", key);
	return RFWHIP_OK;
}

extern "C" int rfwhip_get_setting(rfwhip_context *c, const char *key, char *value, size_t cap)
{
	CTX_ENTER(c);
	if (!key || !value || !cap)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null key/value");
	const std::string k(key);
	if (k == "integrator")
		snprintf(value, cap, "%s", c->integrator ? "pt" : "parity");
	else if (k == "spp")
		snprintf(value, cap, "%d", c->spp);
	else if (k == "max_depth")
		snprintf(value, cap, "%d", c->max_depth);
	else if (k == "jitter")
		snprintf(value, cap, "%s", c->jitter ? "center" : "xor128");
	else if (k == "builder")
		snprintf(value, cap, "%s", c->builder ? "device" : "host");
	else if (k == "sampler")
		snprintf(value, cap, "%s", c->sampler ? "bluenoise" : "hash");
	else if (k == "stage_timing")
		snprintf(value, cap, "%d", c->stage_timing);
	else if (k == "count_traversal")
		snprintf(value, cap, "%d", c->count_traversal);
	else if (k == "lds_nodes")
		snprintf(value, cap, "%d", c->lds_nodes);
	else if (k == "refill")
		snprintf(value, cap, "%d", c->refill);
	else if (k == "arm")
		snprintf(value, cap, "0"); // (retired: see rfwhip_set_setting)
	else if (k == "fuse")
		snprintf(value, cap, "%d", c->fuse);
	else if (k == "shadow_packets")
		snprintf(value, cap, "%d", c->shadow_packets);
	else if (k == "shadow_side")
		snprintf(value, cap, "%d", c->shadow_side);
	else if (k == "group_flags")
		snprintf(value, cap, "%d", c->group_flags);
	else if (k == "shadow_packets_on") // (read-only: would the next large pt call take the packet form of the depth-0 connection wave?)
		snprintf(value, cap, "%d", (c->shadow_packets > 0 || (c->shadow_packets < 0 && c->shadow_packets_auto_on)) && c->packet_ok && c->nodes4f_current && (c->refill & 8) ? 1 : 0);
	else if (k == "shadow_bins_per_run") // (read-only: light bins per sorted run of the last waited frames, see shadow_packets)
		snprintf(value, cap, "%.3f", c->shadow_bins_per_run);
	else if (k == "streams")
		snprintf(value, cap, "%d", c->streams);
	else if (k == "flatten_bytes")
		snprintf(value, cap, "%lld", c->flatten_bytes);
	else if (k == "sub_batch_paths")
		snprintf(value, cap, "%lld", c->sub_batch_paths);
	else if (k == "ring")
		snprintf(value, cap, "%d", c->ring);
	else if (k == "overlap")
		snprintf(value, cap, "%d", c->overlap);
	else if (k == "sample_group")
		snprintf(value, cap, "%d", c->sample_group);
	else if (k == "flat_instances")
		snprintf(value, cap, "%d", c->flat_instances);
	// read-only: which kernel variants the render calls launch (for hosts that label their measurements: bench.py)
	else if (k == "textured") // some material carries a texture / normal map: k_shade_pt<true>
		snprintf(value, cap, "%d", c->textured ? 1 : 0);
	else if (k == "packet") // the pt primary wave runs in packet form (k_primary_packet) for sample groups >= 2 or large launches
		snprintf(value, cap, "%d", (c->packet_ok && (c->refill & 8)) ? 1 : 0);
	else if (k == "world_tree") // triangles in the world tree of the last rfwhip_update (0: none)
		snprintf(value, cap, "%zu", c->wtree.valid ? c->wtree.tris : (size_t)0);
	else
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "unknown setting \"%s\"", key);
	return RFWHIP_OK;
}

extern "C" int rfwhip_get_settings(rfwhip_context *c, const char **keys, size_t cap)
{
	(void)c;
	const size_t n = sizeof(k_setting_keys) / sizeof(k_setting_keys[0]);
	for (size_t i = 0; i < n && i < cap && keys; i++)
		keys[i] = k_setting_keys[i];
	return (int)n;
}

extern "C" int rfwhip_get_counters(rfwhip_context *c, rfwhip_counters *out, int reset)
{
	CTX_ENTER(c);
	if (!out)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null counters");
	RF_TRY(sync_all(c));
	memset(out, 0, sizeof(*out));
	for (int i = 0; i < rfwhip_context::MAX_SUB; i++)
	{
		void *buf = i == 0 ? c->d_counters.p : c->d_counters_sub[i].p;
		if (!buf)
			continue;
		rt::WaveCounters wc;
		RF_TRY(dm::d2h(&wc, buf, sizeof(wc), c->stream));
		out->rays_extend += wc.rays_extend, out->rays_shadow += wc.rays_shadow;
		out->inner_extend += wc.inner_extend, out->tris_extend += wc.tris_extend;
		out->inner_shadow += wc.inner_shadow, out->tris_shadow += wc.tris_shadow;
		out->shaded += wc.shaded;
		out->lds_extend += wc.lds_extend, out->lds_shadow += wc.lds_shadow;
		// the device-side clock of the extend stage: launches already folded + the ones of the most recent call
		out->extend_ticks += wc.ext_ticks, out->extend_launches_timed += wc.ext_timed;
		for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
			if (wc.t_last[d] > wc.t_first[d])
				out->extend_ticks += wc.t_last[d] - wc.t_first[d], out->extend_launches_timed++;
		if (reset)
		{
			wc.rays_extend = wc.rays_shadow = wc.inner_extend = wc.tris_extend = wc.inner_shadow = wc.tris_shadow = wc.shaded = 0;
			wc.lds_extend = wc.lds_shadow = 0;
			wc.ext_ticks = 0, wc.ext_timed = 0;
			for (int d = 0; d < rt::MAX_DEPTH_SLOTS; d++)
				wc.t_first[d] = ~0ull, wc.t_last[d] = 0ull;
			RF_TRY(dm::h2d(buf, &wc, sizeof(wc), c->stream));
			RF_TRY(dm::sync(c->stream));
		}
	}
	out->samples = c->totals.samples;
	if (reset)
		c->totals.samples = 0;
	return RFWHIP_OK;
}

extern "C" int rfwhip_get_kernel_time(rfwhip_context *c, int which, float *ms, uint32_t *launches, int reset)
{
	CTX_ENTER(c);
	if (which < 0 || which >= KF_COUNT)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "kernel family %d out of range", which);
	if (ms)
		*ms = c->kernel_ms[which];
	if (launches)
		*launches = c->kernel_launches[which];
	if (reset)
		c->kernel_ms[which] = 0.0f, c->kernel_launches[which] = 0;
	return RFWHIP_OK;
}

extern "C" int rfwhip_read_primary_hits(rfwhip_context *c, float *t, int32_t *prim, int32_t *inst, float *u, float *v)
{
	CTX_ENTER(c);
	if (!c->W || c->wave_capacity == 0)
		return set_error(RFWHIP_ERR_STATE, "no frame rendered yet");
	RF_TRY(sync_all(c));
	// the first sample of the most recent call's first sub-batch lives in that sub-batch's first sample group
	rt::FrameView fr = c->fr;
	set_sample_group(fr, c->sgroup_last);
	const size_t slots = (size_t)c->fr.slots << fr.sgroup_log2;
	std::vector<f4> h(slots);
	std::vector<int> hi(slots);
	RF_TRY(dm::d2h(h.data(), c->d_hit0.as<f4>() + c->last_wave_off, slots * sizeof(f4), c->stream));
	RF_TRY(dm::d2h(hi.data(), c->d_hit0_inst.as<int>() + c->last_wave_off, slots * 4, c->stream));
	const size_t n = (size_t)c->W * c->H;
	for (size_t i = 0; i < n; i++)
	{
		if (t)
			t[i] = 1e34f;
		if (prim)
			prim[i] = -1;
		if (inst)
			inst[i] = -1;
		if (u)
			u[i] = 0;
		if (v)
			v[i] = 0;
	}
	const bool parity = c->integrator == 0;
	for (uint32_t yl = 0; yl < c->fr.local_rows; yl++)
	{
		const uint32_t y = rt::strip_of_local(yl / rt::STRIP_ROWS, (uint32_t)c->rank, (uint32_t)c->world) * rt::STRIP_ROWS + yl % rt::STRIP_ROWS;
		if (y >= c->H)
			continue;
		for (uint32_t x = 0; x < c->W; x++)
		{
			if (parity && (x >= (c->W / 4u) * 4u || y >= (c->H / 2u) * 2u))
				continue;
			const uint32_t tile = (yl / rt::TILE) * c->fr.tiles_x + x / rt::TILE;
			const size_t slot = (size_t)rt::pixel_to_slot(fr, tile, rt::tile_pix(x % rt::TILE, yl % rt::TILE), 0u);
			const size_t o = (size_t)y * c->W + x;
			int pr;
			memcpy(&pr, &h[slot].w, 4);
			if (pr == rt::HIT_MISS_SHADED) // (a miss the primary kernel shaded itself: kernels.hip, primary_finish_item)
				pr = -1;
			if (t)
				t[o] = h[slot].x;
			if (u)
				u[o] = h[slot].y;
			if (v)
				v[o] = h[slot].z;
			if (prim)
				prim[o] = pr;
			if (inst)
				inst[o] = pr >= 0 ? hi[slot] : -1;
		}
	}
	return RFWHIP_OK;
}

extern "C" int rfwhip_trace_rays(rfwhip_context *c, size_t n, const float *org, const float *dir, float t_min, float t_max,
								 float *t, int32_t *prim, int32_t *inst, float *u, float *v)
{
	CTX_ENTER(c);
	if (n && (!org || !dir))
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_trace_rays: null rays");
	if (c->scene_dirty)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_trace_rays: scene changed since the last rfwhip_update()");
	if (n == 0)
		return RFWHIP_OK;
	if (n >= (1ull << 31))
		return set_error(RFWHIP_ERR_UNSUPPORTED, "rfwhip_trace_rays: too many rays");
	RF_TRY(sync_all(c)); // the wave buffers are shared with the sub-batch streams of a render in flight
	RF_TRY(ensure_wave_buffers(c, std::max(n, c->wave_capacity), c->rad_capacity[0], c->rad_capacity[1]));
	std::vector<f4> o4(n), d4(n);
	for (size_t i = 0; i < n; i++)
	{
		o4[i] = f4{org[3 * i], org[3 * i + 1], org[3 * i + 2], t_min};
		d4[i] = f4{dir[3 * i], dir[3 * i + 1], dir[3 * i + 2], t_max};
	}
	void *s = c->stream;
	// the generic-ray wave uses the odd buffers at depth 1: org/dir carry (origin, t_min) and (direction, t_max)
	RF_TRY(dm::h2d(c->d_org[1].p, o4.data(), n * sizeof(f4), s));
	RF_TRY(dm::h2d(c->d_dir2[1].p, d4.data(), n * sizeof(f4), s));
	rtk::Params p;
	fill_params(c, nullptr, p);
	rtk::launch_init_counters(p.wv.counters, 0, s);
	rtk::launch_set_ext_count(p.wv.counters, 1, (uint32_t)n, s);
	p.depth = 1, p.queue = 0, p.group = 16;
	rtk::launch_extend(p, rtk::GEN_RANGED, c->count_traversal != 0, (uint32_t)n, s);
	RF_TRY(dm::last_launch_error());
	std::vector<f4> h(n);
	std::vector<int> hi(n);
	RF_TRY(dm::d2h(h.data(), c->d_hit.p, n * sizeof(f4), s));
	RF_TRY(dm::d2h(hi.data(), c->d_hit_inst.p, n * 4, s));
	{
		rt::WaveCounters wc;
		RF_TRY(dm::d2h(&wc, c->d_counters.p, sizeof(wc), s));
		if (wc.stack_overflow)
			return set_error(RFWHIP_ERR_STATE, "traversal stack overflow: %u entries dropped", wc.stack_overflow);
	}
	for (size_t i = 0; i < n; i++)
	{
		int pr;
		memcpy(&pr, &h[i].w, 4);
		if (t)
			t[i] = h[i].x;
		if (u)
			u[i] = h[i].y;
		if (v)
			v[i] = h[i].z;
		if (prim)
			prim[i] = pr;
		if (inst)
			inst[i] = pr >= 0 ? hi[i] : -1;
	}
	return RFWHIP_OK;
}

static_assert(RFWHIP_KAT_IN == 24 && RFWHIP_KAT_OUT == 8, "kat_item's record layout (kernels.hip)");
static_assert(RFWHIP_STRIP_ROWS == (int)rt::STRIP_ROWS, "rfwhip.h: rfwhip_row_owner() restates rt::strip_owner()");
extern "C" int rfwhip_kat(rfwhip_context *c, int function, size_t n, const float *in, float *out)
{
	CTX_ENTER(c);
	if (n && (!in || !out))
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_kat: null records");
	if (function < 0 || function > RFWHIP_KAT_TEX_WRAP)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_kat: unknown function %d", function);
	if ((function == RFWHIP_KAT_POINT_ON_LIGHT || function == RFWHIP_KAT_LIGHT_PICK_PROB) && c->scene_dirty)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_kat: the light functions use the lights of the last rfwhip_update()");
	if (function == RFWHIP_KAT_BLUE_NOISE && !c->have_blue_noise)
		return set_error(RFWHIP_ERR_STATE, "rfwhip_kat: no blue-noise table (rfwhip_set_blue_noise)");
	if (n == 0)
		return RFWHIP_OK;
	if (n >= (1ull << 31))
		return set_error(RFWHIP_ERR_UNSUPPORTED, "rfwhip_kat: too many records");
	RF_TRY(sync_all(c));
	DevBuf d_in, d_out;
	int rc = d_in.ensure(n * RFWHIP_KAT_IN * sizeof(float));
	if (!rc)
		rc = d_out.ensure(n * RFWHIP_KAT_OUT * sizeof(float));
	if (!rc)
		rc = dm::h2d(d_in.p, in, n * RFWHIP_KAT_IN * sizeof(float), c->stream);
	if (!rc)
	{
		rtk::Params p;
		fill_params(c, nullptr, p);
		p.cam.blue_noise = c->have_blue_noise ? c->d_blue_noise.as<uint32_t>() : nullptr;
		rtk::launch_kat(p, function, d_in.as<float>(), d_out.as<float>(), (uint32_t)n, c->stream);
		rc = dm::last_launch_error();
	}
	if (!rc)
		rc = dm::d2h(out, d_out.p, n * RFWHIP_KAT_OUT * sizeof(float), c->stream);
	d_in.free_(), d_out.free_();
	return rc;
}

extern "C" int rfwhip_get_bvh(rfwhip_context *c, size_t mesh_index, rfwhip_bvh_node *nodes, size_t node_cap,
							  uint32_t *prim_indices, size_t prim_cap, size_t *node_count, size_t *prim_count)
{
	CTX_ENTER(c);
	if (mesh_index >= c->meshes.size() || !c->meshes[mesh_index].used)
		return set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_get_bvh: no mesh %zu", mesh_index);
	MeshRec &m = c->meshes[mesh_index];
	if (m.device_built)
	{
		// no host copy exists: fetched on demand (this is a debugging / test hook, not part of the build path)
		if (node_count)
			*node_count = m.node_count2;
		if (prim_count)
			*prim_count = m.triCount;
		RF_TRY(sync_all(c));
		const uint32_t nb = m.resident ? m.node_base : 0u, tb = m.resident ? m.tri_base : 0u;
		if (nodes && node_cap)
		{
			const size_t n = std::min<size_t>(node_cap, m.node_count2);
			const rt::Node *src = m.resident ? c->d_nodes.as<rt::Node>() + m.node_base : m.d_b_nodes.as<rt::Node>();
			RF_TRY(dm::d2h(nodes, src, n * sizeof(rt::Node), c->stream));
			for (size_t k = 0; k < n; k++) // device entries -> the reference layout (bvh_node.h:23-28), mesh-local
			{
				if (nodes[k].count > 0)
					nodes[k].left_first = (int32_t)(((uint32_t)nodes[k].left_first & rt::ENTRY_FIRST_MASK) - tb);
				else if (nodes[k].count < 0)
					nodes[k].left_first = (int32_t)(((uint32_t)nodes[k].left_first & rt::ENTRY_INDEX_MASK) - nb);
			}
		}
		if (prim_indices && prim_cap)
		{
			std::vector<f4> tv(3 * m.triCount);
			const f4 *src = m.resident ? c->d_tri_verts.as<f4>() + 3ull * m.tri_base : m.d_b_tri_verts.as<f4>();
			RF_TRY(dm::d2h(tv.data(), src, tv.size() * sizeof(f4), c->stream));
			for (size_t k = 0; k < std::min<size_t>(prim_cap, m.triCount); k++)
				memcpy(&prim_indices[k], &tv[3 * k].w, 4);
		}
		return RFWHIP_OK;
	}
	if (node_count)
		*node_count = m.bvh.nodes.size();
	if (prim_count)
		*prim_count = m.bvh.order.size();
	if (nodes && node_cap)
	{
		const size_t n = std::min(node_cap, m.bvh.nodes.size());
		if (m.resident)
		{
			RF_TRY(sync_all(c));
			RF_TRY(dm::d2h(nodes, c->d_nodes.as<rt::Node>() + m.node_base, n * sizeof(rt::Node), c->stream));
			for (size_t k = 0; k < n; k++) // device nodes carry packed entries: hand out the reference layout
				nodes[k].left_first = m.bvh.nodes[k].left_first;
		}
		else
			memcpy(nodes, m.bvh.nodes.data(), n * sizeof(rt::Node));
	}
	if (prim_indices && prim_cap)
		memcpy(prim_indices, m.bvh.order.data(), std::min(prim_cap, m.bvh.order.size()) * 4);
	return RFWHIP_OK;
}
