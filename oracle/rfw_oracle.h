/*
 * rfw_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the algorithms on the north-star hot path of MeirBon/rendering-fw, used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under rendering-fw_amd/
 * may include, link or call it.
 *
 * PARITY UNPINNED against the reference's own binaries: the reference ships no tests, golden images or known-answer
 * vectors, and neither its Embree rendercore (Embree 3, TBB, glm, GLEW, a GL context), its CUDA rendercore, its BSDF
 * headers (glm) nor its in-tree BVH (a Rust crate fetched from the network) can be compiled in this image
 * (SURVEY.md §0.3, §8c).  What pins this restatement within that limit:
 *   - the one reference file that does compile here: blue_noise.h, built by `make -C oracle ref` into
 *     oracle/_ref/libbluenoise.so — the sampler is checked against the REAL tables (crc32s of SURVEY §2) and sampler
 *     outputs / a render using them are committed under tests/golden/,
 *   - an independent numpy float32 restatement of BOTH integrators written from the reference text
 *     (tests/golden/make_golden.py, make_golden_pt.py): known-answer tables for the BSDF / light-sampling / packing
 *     functions and brute-force renders with per-depth wave counts, committed under tests/golden/,
 *   - known answers derived by hand from the reference's integer arithmetic (xor128.h:20-27, tools.h:218-235),
 *   - its own brute-force (no-BVH) mode.
 *
 * The interface mirrors include/rfwhip.h one-to-one (prefix rfwo_ instead of rfwhip_), so the parity tests drive
 * both with the same scene description.
 */
#ifndef RFW_ORACLE_H
#define RFW_ORACLE_H

#include "../include/rfwhip_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rfwo_context rfwo_context;

const char *rfwo_last_error(void);
int rfwo_create(int device_ordinal, int rank, int world, rfwo_context **out);
int rfwo_cleanup(rfwo_context *ctx);
void rfwo_destroy(rfwo_context *ctx);
int rfwo_init(rfwo_context *ctx, uint32_t width, uint32_t height);
int rfwo_set_sky(rfwo_context *ctx, const float *rgb, size_t width, size_t height);
int rfwo_set_textures(rfwo_context *ctx, const rfwhip_texture *textures, size_t count);
int rfwo_set_materials(rfwo_context *ctx, const rfwhip_material *materials, const rfwhip_material_tex_ids *ids,
					   size_t count);
int rfwo_set_mesh(rfwo_context *ctx, size_t index, const rfwhip_mesh *mesh);
int rfwo_set_instance(rfwo_context *ctx, size_t index, size_t mesh_index, const float *transform16,
					  const float *normal_matrix9);
int rfwo_set_lights(rfwo_context *ctx, rfwhip_light_count count, const rfwhip_area_light *area,
					const rfwhip_point_light *point, const rfwhip_spot_light *spot,
					const rfwhip_directional_light *directional);
int rfwo_set_blue_noise(rfwo_context *ctx, const uint32_t *table, size_t words);
int rfwo_update(rfwo_context *ctx);
void rfwo_camera_get_view(const rfwhip_camera *camera, rfwhip_camera_view *view);
int rfwo_render(rfwo_context *ctx, const rfwhip_camera *camera, int status);
int rfwo_wait(rfwo_context *ctx);
int rfwo_read_framebuffer(rfwo_context *ctx, float *rgba);
uint32_t rfwo_local_rows(const rfwo_context *ctx);
int rfwo_read_local_framebuffer(rfwo_context *ctx, float *rgba);
int rfwo_set_probe_index(rfwo_context *ctx, uint32_t x, uint32_t y);
int rfwo_get_probe_results(rfwo_context *ctx, uint32_t *inst, uint32_t *prim, float *dist);
int rfwo_get_stats(rfwo_context *ctx, rfwhip_render_stats *stats);
/* keys as rfwhip_set_setting (integrator, spp, max_depth, jitter, sampler; the launch-shape keys are accepted and
 * ignored); extra keys: bvh = "1" | "0" (0 = brute force over all triangles), threads = N,
 * arith = "product" (default) | "reference": the triangle test, pt primary ray and sky lookup in the shapes the product fixes
 * (rounded(), fmaf, total order on (t, instance, prim)) or as the reference's text shapes them (plain products, strict t > tt).
 * NOTE: `arith` is PROCESS-WIDE although it is set through a context — the arithmetic helpers take no context; contexts
 * (and threads) of one process share it, so a test that switches it restores "product" in a finally block. */
int rfwo_set_setting(rfwo_context *ctx, const char *key, const char *value);
int rfwo_read_primary_hits(rfwo_context *ctx, float *t, int32_t *prim, int32_t *inst, float *u, float *v);
int rfwo_get_bvh(rfwo_context *ctx, size_t mesh_index, rfwhip_bvh_node *nodes, size_t node_cap, uint32_t *prims,
				 size_t prim_cap, size_t *node_count, size_t *prim_count);
int rfwo_trace_rays(rfwo_context *ctx, size_t n, const float *org, const float *dir, float t_min, float t_max, float *t,
					int32_t *prim, int32_t *inst, float *u, float *v);
/* traversal statistics of everything traced since the last reset: rays/inner/tris for closest + shadow */
int rfwo_get_counters(rfwo_context *ctx, uint64_t out[8], int reset);

/* SceneMesh::set_pose (geometry/gltf/mesh.cpp:31-45) on plain arrays: n vertices, base positions / normals as float4,
 * joints as 4 uint32 per vertex, mats = column-major 4x4 per joint.  out_n.w = 0. */
void rfwo_skin_vertices(const float *base_v4, const float *base_n4, const uint32_t *joints4, const float *weights4,
						const float *mats16, uint32_t joint_count, size_t n, float *out_v4, float *out_n4);

/* ---- known-answer hooks ---- */
uint32_t rfwo_xor128_next(uint32_t state[4]);				  /* utils/xor128.h:20-27 */
float rfwo_rng_rand(uint32_t state[4]);						  /* utils/rng.h:14 */
void rfwo_xor128_jump(uint32_t state[4], uint64_t draws);	  /* = calling rfwo_xor128_next `draws` times */
/* bsdf/tools.h:163-181 on a 5 x 65536-word table */
float rfwo_blue_noise_sample(const uint32_t *table, int x, int y, int sample_idx, int dim);
uint32_t rfwo_wang_hash(uint32_t s);						  /* bsdf/tools.h:218-225 */
uint32_t rfwo_random_int(uint32_t *s);						  /* bsdf/tools.h:227-233 */
float rfwo_random_float(uint32_t *s);						  /* bsdf/tools.h:235 */
float rfwo_half_to_float(uint16_t h);
/* bvh/src/bvh_tree.cpp:166-196; returns 1 and updates *t,*u,*v on an accepted hit */
int rfwo_intersect_triangle(const float org[3], const float dir[3], float t_min, float *t, const float p0[3],
							const float p1[3], const float p2[3], float *u, float *v);
/* bvh/src/aabb.cpp:39-77 */
int rfwo_intersect_aabb(const float bmin[3], const float bmax[3], const float org[3], const float inv_dir[3],
						float t, float *tmin, float *tmax);
float rfwo_triangle_area(const float v0[3], const float v1[3], const float v2[3]); /* context.cpp:6-15 */
uint32_t rfwo_pack_normal(const float n[3]);									   /* bsdf/tools.h:10-21 */
void rfwo_unpack_normal(uint32_t p, float n[3]);								   /* bsdf/tools.h:22-29 */
/* bsdf/disney.h:266-280 on raw parameters: returns bsdf rgb + pdf (evaluate) */
void rfwo_evaluate_bsdf(const float color[3], const uint32_t params[4], const float iN[3], const float wo[3],
						const float wi[3], float out_rgb[3], float *pdf);
void rfwo_sample_bsdf(const float color[3], const float absorption[3], const uint32_t params[4], const float iN[3],
					  const float wo[3], float t, int backfacing, float r3, float r4, float out_rgb[3],
					  float wi[3], float *pdf);

/* one of the path tracer's functions on n records; functions and record layout: RFWHIP_KAT_* in include/rfwhip_abi.h */
int rfwo_kat(rfwo_context *ctx, int function, size_t n, const float *in, float *out);

#ifdef __cplusplus
}
#endif
#endif
