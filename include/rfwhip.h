/*
 * rfwhip.h — C ABI of the MI355X-native (gfx950, HIP) wavefront rendercore for MeirBon/rendering-fw.
 *
 * This is the drop-in boundary: a `rfw::RenderContext` plugin (RFW/system/context/rfw/context/context.h:74-111)
 * forwards each virtual call to the entry point listed beside it below; `rendering-fw_amd/csrc/plugin/HipRT.cpp`
 * is that plugin, and INTEGRATION.md shows the binding.  Only PODs from rfwhip_abi.h, plain pointers and sizes
 * cross this boundary; errors are int codes + rfwhip_last_error() (the reference throws std::runtime_error across
 * the .so boundary, context.h:84-91 — the plugin shim re-throws).  Every pointer argument is borrowed for the call
 * only: the core uploads into HBM inside the call and never dereferences host memory later (contrast
 * EmbreeRT/src/Mesh.cpp:29,46 which keeps shared buffers).
 *
 * All calls for one context must come from one thread at a time (RFW/system/src/rfw/app.cpp:15-16).
 */
#ifndef RFWHIP_H
#define RFWHIP_H

#include "rfwhip_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RFWHIP_API __attribute__((visibility("default")))

typedef struct rfwhip_context rfwhip_context;

enum rfwhip_status
{
	RFWHIP_OK = 0,
	RFWHIP_ERR_INVALID_ARGUMENT = 1,
	RFWHIP_ERR_NO_DEVICE = 2, /* no HIP device / kernel image not loadable: the core never falls back to a CPU path */
	RFWHIP_ERR_HIP = 3,
	RFWHIP_ERR_STATE = 4,
	RFWHIP_ERR_UNSUPPORTED = 5
};

/* Thread-local description of the last failure of any rfwhip_* call on this thread. */
RFWHIP_API const char *rfwhip_last_error(void);
RFWHIP_API const char *rfwhip_version(void);

/* ---- lifetime ------------------------------------------------------------------------------------------------
 * createRenderContext / destroyRenderContext   (RFW/system/context/rfw/context/export.h:8-15)
 * rank/world: this context renders the 8-row strips it owns.  Strips are dealt to the ranks in periods of
 * `world`, forwards in even periods and backwards in odd ones (0 1 .. w-1 | w-1 .. 1 0 | ...): strip s belongs to rank
 * (s / world) odd ? world - 1 - s % world : s % world — NOT plain s % world (rows get dearer down an image of terrain
 * under sky, and the serpentine cancels that gradient).  A host that gathers local framebuffers itself de-interleaves
 * with rfwhip_deinterleave_*; rfwhip_group_* / rfwhip_comm_* below do the whole gather.  world = 1 renders everything. */
#define RFWHIP_STRIP_ROWS 8
/* the rank (of `world`) that owns image row y: the rule above, for hosts that route per-pixel queries (the probe) themselves */
static inline int rfwhip_row_owner(int y, int world)
{
	const int strip = y / RFWHIP_STRIP_ROWS, k = strip / world, pos = strip % world;
	return (k & 1) ? world - 1 - pos : pos;
}
RFWHIP_API int rfwhip_create(int device_ordinal, int rank, int world, rfwhip_context **out);
/* RenderContext::cleanup() (context.h:93).  Idempotent: the reference calls it twice on unload
 * (system.cpp:160-178 + EmbreeRT/src/Context.cpp:30). */
RFWHIP_API int rfwhip_cleanup(rfwhip_context *ctx);
RFWHIP_API void rfwhip_destroy(rfwhip_context *ctx);

/* RenderContext::init(GLuint*, uint width, uint height) (context.h:88) with the headless BUFFER target
 * (RenderTarget::BUFFER, context.h:27-34); may be called again on resize (app.cpp:44-59). */
RFWHIP_API int rfwhip_init(rfwhip_context *ctx, uint32_t width, uint32_t height);

/* ---- scene synchronisation, in the order rfw::system::synchronize issues them (system.cpp:247-433) ----------- */
/* set_sky(const std::vector<glm::vec3>&, size_t w, size_t h)                               context.h:100 */
RFWHIP_API int rfwhip_set_sky(rfwhip_context *ctx, const float *rgb, size_t width, size_t height);
/* set_textures(const std::vector<TextureData>&)                                            context.h:97 */
RFWHIP_API int rfwhip_set_textures(rfwhip_context *ctx, const rfwhip_texture *textures, size_t count);
/* set_materials(const std::vector<DeviceMaterial>&, const std::vector<MaterialTexIds>&)    context.h:95-96 */
RFWHIP_API int rfwhip_set_materials(rfwhip_context *ctx, const rfwhip_material *materials,
									const rfwhip_material_tex_ids *tex_ids, size_t count);
/* set_mesh(size_t index, const Mesh&): same vertexCount as before => refit, else rebuild    context.h:98,
 * EmbreeRT/src/Mesh.cpp:33-35, bvh/src/top_level_bvh.cpp:26 */
RFWHIP_API int rfwhip_set_mesh(rfwhip_context *ctx, size_t index, const rfwhip_mesh *mesh);
/* set_instance(size_t i, size_t meshIdx, const mat4& transform, const mat3& inverse_transform)
 * transform: column-major 4x4 object->world; normal_matrix: column-major 3x3 inverse-transpose
 * (system.cpp:347).                                                                          context.h:99 */
RFWHIP_API int rfwhip_set_instance(rfwhip_context *ctx, size_t index, size_t mesh_index, const float *transform16,
								   const float *normal_matrix9);
/* set_lights(LightCount, area*, point*, spot*, directional*)                                context.h:101-103 */
RFWHIP_API int rfwhip_set_lights(rfwhip_context *ctx, rfwhip_light_count count, const rfwhip_area_light *area,
								 const rfwhip_point_light *point, const rfwhip_spot_light *spot,
								 const rfwhip_directional_light *directional);
/* The 5 x 65536-word table of the reference's blue-noise sampler (createBlueNoiseBuffer(), blue_noise.h:8204, uploaded
 * by CUDART/src/Context.cpp:43-46).  The table is data of the reference tree and is NOT part of this library: the
 * plugin shim hands it over when it is built there.  With a table and sampler=bluenoise the pt integrator draws the
 * primary-ray jitter / lens sample from blueNoiseSampler (Kernels.cu:391-394) instead of the hash RNG. */
RFWHIP_API int rfwhip_set_blue_noise(rfwhip_context *ctx, const uint32_t *table, size_t words);
/* ---- device skinning (extension: rfw::system skins on the host, geometry/gltf/mesh.cpp:18-125, and re-sends the mesh
 * through set_mesh; these two calls keep the bind pose on the device and take the CPU out of the animation loop) ----
 * set_mesh_skin: per vertex of mesh `index` (as last set with rfwhip_set_mesh = bind pose) four joint indices, four
 * weights and the bind-pose vertex normal (xyz, w ignored).
 * pose_mesh: joint_count column-major 4x4 joint matrices -> vertex = sum_k w_k M[j_k] * base, normal =
 * normalize(base_normal * inverse(that matrix)) (mesh.cpp:35-44), triangles' vertex/face normals (update_triangles,
 * mesh.cpp:428-485), then the same device refit as a same-count rfwhip_set_mesh.  rfwhip_update() afterwards. */
RFWHIP_API int rfwhip_set_mesh_skin(rfwhip_context *ctx, size_t mesh_index, const uint32_t *joints4, const float *weights4,
									const float *base_normals4, size_t vertex_count);
RFWHIP_API int rfwhip_pose_mesh(rfwhip_context *ctx, size_t mesh_index, const float *joint_matrices16, size_t joint_count);
/* ---- device morph targets (extension, same idea: SceneMesh::set_pose(weights), geometry/gltf/mesh.cpp:127-147, runs on the
 * host in the reference).  set_mesh_morph: for mesh `index` (last rfwhip_set_mesh = base pose) the base vertex normals and
 * target_count displacement sets, each vertex_count float4 positions and float4 normals (w ignored), target-major.
 * morph_mesh: vertex = base + sum_j weights[j] * target_j for positions and normals (normals NOT renormalised, as there),
 * then update_triangles (mesh.cpp:428-485) and the device refit.  rfwhip_update() afterwards. */
RFWHIP_API int rfwhip_set_mesh_morph(rfwhip_context *ctx, size_t mesh_index, const float *base_normals4,
									 const float *target_positions4, const float *target_normals4, size_t target_count,
									 size_t vertex_count);
RFWHIP_API int rfwhip_morph_mesh(rfwhip_context *ctx, size_t mesh_index, const float *weights, size_t weight_count);
/* update(): once after a batch of set_* — builds the TLAS, uploads descriptors              context.h:108 */
RFWHIP_API int rfwhip_update(rfwhip_context *ctx);

/* ---- per-frame ---------------------------------------------------------------------------------------------- */
/* Camera::get_view() (Camera.cpp:74-88) as a free function; pure host arithmetic. */
RFWHIP_API void rfwhip_camera_get_view(const rfwhip_camera *camera, rfwhip_camera_view *view);
/* render_frame(const Camera&, RenderStatus) (context.h:94).  Enqueues `spp` samples per pixel on the context's
 * HIP stream and returns without a host sync; RESET clears the accumulator and the sample index, CONVERGE
 * accumulates (context.h:19-23, CUDART/src/Context.cpp:75-80). */
RFWHIP_API int rfwhip_render(rfwhip_context *ctx, const rfwhip_camera *camera, int status);
/* Block until every enqueued render has finished (the reference's render_frame ends with glFinish /
 * cudaDeviceSynchronize: the plugin shim calls rfwhip_render + rfwhip_wait). Resolves stage timings. */
RFWHIP_API int rfwhip_wait(rfwhip_context *ctx);

/* Present: accumulator / samples -> float4 RGBA rows, row 0 = top image row.
 * Full image on this rank (world == 1), host or device destination of width*height*4 floats. */
RFWHIP_API int rfwhip_read_framebuffer(rfwhip_context *ctx, float *rgba_host);
RFWHIP_API int rfwhip_read_framebuffer_device(rfwhip_context *ctx, void *rgba_device);
/* Multi-GPU: this rank's strips, compacted, padded to rfwhip_local_rows() rows (same on every rank). */
RFWHIP_API uint32_t rfwhip_local_rows(const rfwhip_context *ctx);
RFWHIP_API int rfwhip_read_local_framebuffer_device(rfwhip_context *ctx, void *rgba_device);
/* Root-side inverse of the strip interleave: gathered = [world][local_rows][width] float4 -> [height][width]. */
RFWHIP_API int rfwhip_deinterleave_device(rfwhip_context *ctx, const void *gathered_device, void *rgba_device);

/* Stream-ordered forms of the two calls above: enqueue only, no host synchronisation.  hip_stream is a hipStream_t of the
 * caller (e.g. torch's current stream); the present is ordered behind everything this context has enqueued and the
 * context's next rfwhip_render waits on the device until the present has read the accumulator. */
RFWHIP_API int rfwhip_read_local_framebuffer_stream(rfwhip_context *ctx, void *rgba_device, void *hip_stream);
RFWHIP_API int rfwhip_deinterleave_stream(rfwhip_context *ctx, const void *gathered_device, void *rgba_device,
										  void *hip_stream);

/* Where a context runs and what it renders into (for hosts that move its strips themselves). */
RFWHIP_API int rfwhip_get_placement(rfwhip_context *ctx, int *device_ordinal, int *rank, int *world);
RFWHIP_API int rfwhip_get_target_size(rfwhip_context *ctx, uint32_t *width, uint32_t *height);

/* ---- multi-GPU below this ABI (no counterpart in the reference, which renders on one device) ------------
 * The frame is split into interleaved 8-row strips over `world` devices of one node, each device holds the whole scene,
 * and per presented frame the rank-local strips are gathered ONCE into the root's (rank 0's) HBM and de-interleaved
 * there.  Transport: RCCL point-to-point over xGMI (ncclSend on every rank, world - 1 ncclRecv on the root, one
 * ncclGroup), or peer copies (hipMemcpyPeerAsync).  Everything below is enqueue-only unless it says it waits.
 *
 * rfwhip_group_*: ONE process, ONE thread drives n devices — the reference's host model (RFW/system/src/rfw/app.cpp:3-26).
 *   create   n contexts, context i = rank i of world n on devices[i]; a device may be listed more than once only with the
 *            peer transport (tests on one GPU).  AUTO = RCCL when the devices are distinct and librccl.so opens.
 *   context  the i-th context, for the scene calls: the host repeats every rfwhip_set_* per context (each device keeps
 *            its own copy of the scene), then calls rfwhip_group_update.
 *   init / update / set_setting / render / wait   the context call of the same name on every rank.
 *   gather   present on every rank -> transfer -> de-interleave into the group's full image on the root's device.
 *   read_framebuffer   gather + wait + copy to the host: width * height float4.
 *   framebuffer_device the root-side image (valid after a gather has completed) and the device it lives on. */
enum rfwhip_transport
{
	RFWHIP_TRANSPORT_AUTO = 0,
	RFWHIP_TRANSPORT_RCCL = 1,
	RFWHIP_TRANSPORT_PEER = 2
};
typedef struct rfwhip_group rfwhip_group;
RFWHIP_API int rfwhip_group_create(const int *devices, int n, int transport, rfwhip_group **out);
RFWHIP_API void rfwhip_group_destroy(rfwhip_group *group);
RFWHIP_API int rfwhip_group_size(const rfwhip_group *group);
RFWHIP_API int rfwhip_group_transport(const rfwhip_group *group);
RFWHIP_API rfwhip_context *rfwhip_group_context(rfwhip_group *group, int rank);
RFWHIP_API int rfwhip_group_init(rfwhip_group *group, uint32_t width, uint32_t height);
RFWHIP_API int rfwhip_group_update(rfwhip_group *group);
RFWHIP_API int rfwhip_group_set_setting(rfwhip_group *group, const char *key, const char *value);
RFWHIP_API int rfwhip_group_render(rfwhip_group *group, const rfwhip_camera *camera, int status);
RFWHIP_API int rfwhip_group_gather(rfwhip_group *group);
RFWHIP_API int rfwhip_group_wait(rfwhip_group *group);
RFWHIP_API int rfwhip_group_read_framebuffer(rfwhip_group *group, float *rgba_host);
RFWHIP_API int rfwhip_group_framebuffer_device(rfwhip_group *group, void **rgba_device, int *device_ordinal);
/* Pipelined presentation (frames in flight): present_async = gather + asynchronous copy of the image into one of
 * RFWHIP_PRESENT_SLOTS pinned host buffers, enqueue only; present_wait blocks until that slot's copy has landed and hands out
 * the buffer (valid until the slot is presented into again).  A host that keeps n <= RFWHIP_PRESENT_SLOTS frames in flight —
 * render(k), present_async(k % n), present_wait((k + 1) % n) from frame n - 1 on — shows frame k - n + 1 while frames up to
 * k render: a frame's launch chain is ten dependent kernels of tails, and it takes about four chains in flight to fill the
 * device (1080p, 1 spp, image on the host every frame: 2.76 ms per frame with render + wait + read-back, 2.14 ms with two
 * frames in flight, 1.70 ms with four). */
#define RFWHIP_PRESENT_SLOTS 4
RFWHIP_API int rfwhip_group_present_async(rfwhip_group *group, int slot);
RFWHIP_API int rfwhip_group_present_wait(rfwhip_group *group, int slot, const float **rgba_host);

/* rfwhip_comm_*: one process per device (e.g. under torch.distributed.run).  Rank 0 calls rfwhip_comm_unique_id and
 * hands the RFWHIP_COMM_ID_BYTES bytes to the other ranks by any means (a file, MPI, a torch broadcast); then EVERY rank
 * calls rfwhip_comm_create with its context (rank / world as given to rfwhip_create) — a collective call, like
 * ncclCommInitRank.  rfwhip_comm_gather is collective too: every rank presents and sends, the root receives and
 * de-interleaves into rgba_device (width * height float4 on its device; ignored on the other ranks; NULL = an internal
 * buffer).  The transport is RCCL; nothing but the id travels outside this library.  A world of ONE needs no id (the gather is a
 * copy); given one all the same, rfwhip_comm_create builds a real one-rank RCCL communicator and the gather sends the strips to
 * itself through it (ncclSend + ncclRecv on rank 0): the library's RCCL calls exercised on a box with a single device. */
#define RFWHIP_COMM_ID_BYTES 128
typedef struct rfwhip_comm rfwhip_comm;
RFWHIP_API int rfwhip_comm_unique_id(void *id, size_t cap);
RFWHIP_API int rfwhip_comm_create(rfwhip_context *ctx, const void *id, rfwhip_comm **out);
RFWHIP_API void rfwhip_comm_destroy(rfwhip_comm *comm);
RFWHIP_API int rfwhip_comm_gather(rfwhip_comm *comm, void *rgba_device);
RFWHIP_API int rfwhip_comm_wait(rfwhip_comm *comm);

/* get_probe_results / set_probe_index                                                       context.h:104,109 */
RFWHIP_API int rfwhip_set_probe_index(rfwhip_context *ctx, uint32_t x, uint32_t y);
RFWHIP_API int rfwhip_get_probe_results(rfwhip_context *ctx, uint32_t *instance_index, uint32_t *primitive_index,
										float *distance);
/* get_stats()                                                                               context.h:110 */
RFWHIP_API int rfwhip_get_stats(rfwhip_context *ctx, rfwhip_render_stats *stats);

/* get_settings / set_setting (context.h:106-107). Keys:
 *   integrator   = "parity" (restates EmbreeRT/src/Context.cpp:104-300) | "pt" (CUDART/src/Kernels.cu:571-794)
 *   spp          = samples per pixel enqueued by one rfwhip_render call (default 1)
 *   max_depth    = MAX_PATH_LENGTH of the pt integrator (settings.h:5, default 2)
 *   jitter       = "xor128" (EmbreeRT: rfw::utils::xor128 stream) | "center" (r0=r1=0.5) — parity integrator only
 *   builder      = "host" (parallel binned SAH on the CPU, the default) | "device" (on the GPU, lbvh.hip: 63-bit Morton
 *                  order, parallel locally-ordered clustering (PLOC), device collapse to 4-wide nodes; applies to the next
 *                  rfwhip_set_mesh that (re)builds)
 *   sampler      = "hash" (WangHash + xorshift32, tools.h:218-235; default) | "bluenoise" (needs rfwhip_set_blue_noise)
 *                  — pt integrator: primary rays (dimensions 0-3, Kernels.cu:391-394) and, for the first 256 samples, the
 *                  light sample of next-event estimation (dimensions 4-5, Kernels.cu:712-719)
 *   stage_timing = "0"|"1": bracket every stage with hipEvents (fills RenderStats like the reference's timers)
 *   count_traversal = "0"|"1": instrumented traversal (popped inner nodes / triangle tests), for the roofline
 *   lds_nodes    = top-of-tree 4-wide nodes of the largest mesh BVH that every traversal workgroup keeps in LDS
 *                  (-1 = as many as the kernels were built for, the default; 0 disables)
 *   streams      = sub-batches of one render call that run concurrently on their own HIP streams (1..8, default 4)
 *   overlap      = "1": the connection (shadow) wave of depth d runs on a second stream beside extend / shade of depth d + 1
 *                  (hides kernel tails when launches are small); "0": in order on the sub-batch's stream; "-1" (default):
 *                  chosen by the size of the render call
 *   sub_batch_paths = a render call's spp are cut into concurrent sub-batches only if each gets at least this many path
 *                  slots and there are four of them (default 50000000: 1080p from 128 spp per call).  A call below the
 *                  threshold stays ONE sub-batch and rotates through the ring (below) — whose depth in turn depends on the
 *                  device's FREE memory (232 B of path state + 32 B of radiance per slot and ring entry: 1080p at 64 spp =
 *                  35 GB per entry; the ring falls back 3 -> 2 -> 1 before the call fails), so what this setting does to
 *                  throughput depends on how much HBM the rest of the process holds
 *   sample_group = slot layout: up to this many samples of a pixel sit side by side in one wave (power of two <= 64,
 *                  default 64 = a wave is one pixel; the largest such group that divides every sub-batch of a call is
 *                  used: 32 for a 128-spp call cut into four sub-batches; 1 = a wave is one 8x8 tile of one sample).
 *                  Changes which path sits where, never the image
 *   flat_instances = "1" (default): an instance with the identity transform whose mesh no other instance uses is linked
 *                  into the top-level tree directly (rays reach its triangles without an instance switch; hit records,
 *                  images and counters are unchanged), and static instances that are transformed or share their mesh are
 *                  written out in world space under one tree (the WORLD TREE, flatten_bytes); "0": every instance behind
 *                  a top-level leaf, the reference's two-level walk.  Takes effect with the next rfwhip_update()
 *   flatten_bytes = default 268435456 (2.4 M triangles): the world tree is built as long as the world-space copy of its members'
 *                  triangles (every instance of a host-built mesh that is not animated) stays below this many bytes; "0":
 *                  never.  The tree is built on the host INSIDE rfwhip_update() whenever its members change (~0.2 s per million
 *                  member triangles on 16 cores): raise the budget for large static scenes, lower it where updates must stay
 *                  short.  The hit is the same triangle of the same instance at the same t as the two-level walk's up to
 *                  rounding (M p is tested instead of M^-1 o; the triangle test's determinant threshold is scaled by |det M|,
 *                  so a triangle is rejected as degenerate exactly when the reference's object-space test rejects it).
 *                  Animated meshes always keep the two-level walk; an instance whose matrix changes leaves the tree and
 *                  rejoins after 8 updates without a change — 16, 32 ... after every further episode
 *   arm          = retired (round 4's self-arming primary kernels): accepted and ignored
 *   ring         = render calls that are ONE sub-batch rotate through this many sets of wave buffers / streams / counters,
 *                  so that up to `ring` consecutive calls are in flight (1..4, default 3: three chains + the main stream
 *                  are the HIP runtime's four hardware queues; a host that keeps four frames in flight with
 *                  rfwhip_group_present_async sets 4)
 *   refill       = bit mask, default 15: bit 0 the extension (bounce) waves, bit 1 the shadow waves keep persistent lanes (a
 *                  lane that finishes its ray pulls the next one from the wave's run of the launch's queue; off: the
 *                  one-ray-per-lane kernels, kept as a cross-check); bit 3: the pt primary wave in PACKET form — a wave walks
 *                  the tree once for the rays of its 64 slots (scalar node fetches, one stack per wave; used when the samples
 *                  of a pixel sit side by side, sample_group >= 2, or the launch is large, and only for scenes whose trees fit
 *                  its 61-entry stack; hit records are those of the per-lane kernels; off: one ray per lane).  Bit 2 selected
 *                  a persistent-lane primary kernel until round 5 and is ignored
 *   fuse         = "1" (default): the extension rays of depth d + 1 and the shadow rays of depth d share ONE launch (both
 *                  queues are complete when the shade stage of depth d has finished; one kernel tail per depth instead of
 *                  two: 1-spp frames 1.34 -> 1.21 ms); "0": a launch each.  Never changes the image
 *   shadow_packets = "-1" (default) | "1" | "0": the connection wave of the PRIMARY vertices in packet form — their shadow rays
 *                  carry the chosen light's bin in the top bits of their slot word, a wave sorts runs of 256 rays by it and walks
 *                  the tree once per 64 rays (wave-uniform occlusion traversal) instead of once per lane; applies where the pt
 *                  primary wave can run in packet form and the samples of a pixel sit side by side (sample_group >= 8); 16 bins up
 *                  to 2^27 path slots per sub-batch, 8 / 4 / 2 for larger ones.  "-1": while the sorted runs of the last waited frame hold at most
 *                  8 light bins on average (read-only key "shadow_bins_per_run"; "shadow_packets_on" says what the next call
 *                  will do; measured: 3.6 bins per run + 8 %, 5.5 bins + 0.6 %).  Never changes the image
 *   group_flags  = "1" (default): the packet form of the pt primary wave flags the 64-slot groups it has finished itself (no hit:
 *                  sky terms written) in a byte each, and the shade kernel's scan passes them by without reading their records
 *                  (a quarter of the terrain's groups).  "0": every record is read.  Never changes the image
 *   shadow_side  = "1" (default): that wave runs on the sub-batch's connection stream, beside the extension wave of depth 1;
 *                  "0": on the sub-batch's own stream, in front of it (per-stage timings)
 *   rfwhip_get_setting also answers read-only keys: "textured" (the textured shade kernel variant is in use), "packet" (the
 *   pt primary wave can run in packet form), "world_tree" (triangles in the world tree of the last update; 0: none),
 *   "shadow_bins_per_run", "shadow_packets_on".
 * Returns the number of keys; fills up to cap pointers with static strings. */
RFWHIP_API int rfwhip_set_setting(rfwhip_context *ctx, const char *key, const char *value);
RFWHIP_API int rfwhip_get_setting(rfwhip_context *ctx, const char *key, char *value, size_t cap);
RFWHIP_API int rfwhip_get_settings(rfwhip_context *ctx, const char **keys, size_t cap);

/* ---- measurement hooks (bench / tests; not part of the reference interface) ---------------------------------- */
typedef struct rfwhip_counters
{
	uint64_t rays_extend;	 /* closest-hit rays traced since the last reset (primary + extension) */
	uint64_t rays_shadow;	 /* any-hit rays traced */
	uint64_t inner_extend;	 /* popped 4-wide inner nodes (64 bytes = 4 rows of 16 B each), closest-hit rays */
	uint64_t tris_extend;	 /* triangle tests, closest-hit rays */
	uint64_t inner_shadow;
	uint64_t tris_shadow;
	uint64_t shaded;		 /* shade-kernel invocations with a hit */
	uint64_t samples;		 /* pixel samples started */
	uint64_t lds_extend;	 /* of inner_extend: visits served by the LDS top-of-tree cache (no vector-L1 lane-loads) */
	uint64_t lds_shadow;	 /* of inner_shadow: the same */
	uint64_t extend_ticks;	 /* extend-stage kernels, first workgroup in to last workgroup out, summed: ticks of the 100 MHz device clock */
	uint64_t extend_launches_timed; /* launches in extend_ticks */
} rfwhip_counters;
RFWHIP_API int rfwhip_get_counters(rfwhip_context *ctx, rfwhip_counters *out, int reset);

/* Accumulated hipEvent time (ms) and launch count per kernel family since the last reset; requires
 * stage_timing=1.  which: 0 generate, 1 extend, 2 shade, 3 connect, 4 finalize, 5 refit. */
RFWHIP_API int rfwhip_get_kernel_time(rfwhip_context *ctx, int which, float *ms, uint32_t *launches, int reset);

/* Raw closest-hit records of the most recent primary wave (parity tests): per pixel of this rank's local image
 * t (1e34 = miss), primID, instID, u, v. Any pointer may be NULL. */
RFWHIP_API int rfwhip_read_primary_hits(rfwhip_context *ctx, float *t, int32_t *prim, int32_t *inst, float *u,
										float *v);

/* Trace n arbitrary world-space rays (org/dir: n x 3 floats, closest hit in (t_min, t_max)) through the resident
 * scene with the extend kernel; any output pointer may be NULL.  t = t_max on a miss. */
RFWHIP_API int rfwhip_trace_rays(rfwhip_context *ctx, size_t n, const float *org, const float *dir, float t_min,
								 float t_max, float *t, int32_t *prim, int32_t *inst, float *u, float *v);

/* Known-answer hook: one of the path tracer's DEVICE functions (rt_core.h: BSDF, light sampling, packing, samplers — the
 * very functions the shade kernel calls) evaluated by a kernel on n records; functions and record layout: RFWHIP_KAT_* in
 * rfwhip_abi.h.  in: n x RFWHIP_KAT_IN floats, out: n x RFWHIP_KAT_OUT floats (host pointers).  The light functions use
 * the lights of the last rfwhip_update(), BLUE_NOISE the table of rfwhip_set_blue_noise. */
RFWHIP_API int rfwhip_kat(rfwhip_context *ctx, int function, size_t n, const float *in, float *out);

/* BVH of mesh `index` as built on the device side (bvh_node.h layout) + its primitive order. */
RFWHIP_API int rfwhip_get_bvh(rfwhip_context *ctx, size_t mesh_index, rfwhip_bvh_node *nodes, size_t node_cap,
							  uint32_t *prim_indices, size_t prim_cap, size_t *node_count, size_t *prim_count);

#ifdef __cplusplus
}
#endif
#endif /* RFWHIP_H */
