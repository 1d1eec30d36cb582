/*
 * rfw_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  PARITY UNPINNED against the reference's binaries —
 * what pins it instead: rfw_oracle.h.
 *
 * Restates, in plain C, what the reference computes on the north-star path:
 *   camera view              RFW/system/context/rfw/context/Camera.cpp:74-115
 *   xor128 / rand            RFW/system/utils/src/rfw/utils/xor128.h:20-27, rng.h:14
 *   primary rays             RFW/backends/EmbreeRT/src/Ray.cpp:3-47 (scalar form), :176-384 (packet draw order)
 *   BVH2 build               RFW/system/bvh/include/bvh/bvh_node.h:56-81,136-233 (binned SAH, 10 planes/axis)
 *   BVH2 traversal           bvh_node.h:317-448, slab test RFW/system/bvh/src/aabb.cpp:39-77
 *   Möller–Trumbore          RFW/system/bvh/src/bvh_tree.cpp:166-196
 *   two-level instancing     RFW/system/bvh/src/top_level_bvh.cpp:104-191
 *   parity integrator        RFW/backends/EmbreeRT/src/Context.cpp:104-300, 417-476
 *   path-tracing integrator  RFW/backends/CUDART/src/Kernels.cu:383-426, 571-794; lights.h; getShadingData.h;
 *                            RFW/system/context/rfw/bsdf/{disney.h,tools.h,compat.h}
 * Documented deviations are listed in oracle/README.md.
 */
#include "rfw_oracle.h"
#include "rfw_oracle_math.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define STRIP 8
#define GEO_EPS 1e-5f
#define TRI_EPS 1e-6f

static __thread char g_err[512];
const char *rfwo_last_error(void) { return g_err; }
static int fail(const char *msg)
{
	snprintf(g_err, sizeof(g_err), "%s", msg);
	return 1;
}

typedef struct
{
	float bmin[3], bmax[3];
} oaabb;

typedef struct
{
	float *verts; /* vec4 per vertex */
	rfwhip_triangle *tris;
	uint32_t *indices;
	size_t vertexCount, triCount;
	/* per triangle */
	v3 *p0, *p1, *p2;
	oaabb *aabbs;
	v3 *centroids;
	/* BVH2 */
	rfwhip_bvh_node *nodes;
	uint32_t *prims;
	int nodeCount;
	int used;
} omesh;

typedef struct
{
	int used;
	size_t mesh;
	float transform[16], inverse[16], normal[9];
	oaabb world;
} oinstance;

typedef struct
{
	uint32_t type, width, height, texelCount;
	void *data;
	size_t bytes;
} otexture;

struct rfwo_context
{
	int rank, world;
	uint32_t W, H;
	float *acc; /* W*H*4 sums */
	uint32_t samples;
	/* primary hit records of the last sample */
	float *hit_t, *hit_u, *hit_v;
	int32_t *hit_prim, *hit_inst;

	omesh *meshes;
	size_t meshCount;
	oinstance *instances;
	size_t instanceCount;
	rfwhip_material *materials;
	size_t materialCount;
	otexture *textures;
	size_t textureCount;
	float *sky;
	size_t skyW, skyH;
	rfwhip_light_count lc;
	rfwhip_area_light *area;
	rfwhip_point_light *point;
	rfwhip_spot_light *spot;
	rfwhip_directional_light *dir;

	/* settings */
	int integrator; /* 0 parity, 1 pt */
	int spp;
	int max_depth;
	int jitter; /* 0 xor128, 1 center */
	int sampler;			 /* 0 hash RNG, 1 blue noise */
	uint32_t *blue_noise; /* 5 x 65536 words or NULL */
	float *pend;		  /* pt: unoccluded connection contributions per pixel and depth, W*H*pend_depths*3 */
	size_t pend_cap;
	int use_bvh;
	int threads;

	uint32_t rng[4];
	uint32_t probe_x, probe_y;
	uint32_t probe_inst, probe_prim;
	float probe_dist;
	rfwhip_render_stats stats;
	uint64_t cnt[8];
};

/* =============================================================================================================
 * small helpers
 * ========================================================================================================== */
static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void aabb_reset(oaabb *b)
{
	for (int i = 0; i < 3; i++)
		b->bmin[i] = 1e34f, b->bmax[i] = -1e34f;
}
static void aabb_grow_p(oaabb *b, v3 p)
{
	const float q[3] = {p.x, p.y, p.z};
	for (int i = 0; i < 3; i++)
	{
		b->bmin[i] = fminf(b->bmin[i], q[i]);
		b->bmax[i] = fmaxf(b->bmax[i], q[i]);
	}
}
static void aabb_grow(oaabb *b, const oaabb *o)
{
	for (int i = 0; i < 3; i++)
	{
		b->bmin[i] = fminf(b->bmin[i], o->bmin[i]);
		b->bmax[i] = fmaxf(b->bmax[i], o->bmax[i]);
	}
}
static void aabb_offset(oaabb *b, float o)
{
	for (int i = 0; i < 3; i++)
		b->bmin[i] -= o, b->bmax[i] += o;
}
/* aabb.cpp:273-277 */
static float aabb_area(const oaabb *b)
{
	const float e0 = b->bmax[0] - b->bmin[0], e1 = b->bmax[1] - b->bmin[1], e2 = b->bmax[2] - b->bmin[2];
	return fmaxf(0.0f, e0 * e1 + e0 * e2 + e1 * e2);
}

/* general 4x4 inverse, column-major, cofactor expansion in fp32 */
static void mat4_inverse(const float *m, float *inv)
{
	float t[16];
	t[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
		   m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
	t[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
		   m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
	t[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
		   m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
	t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
			m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
	t[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
		   m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
	t[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
		   m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
	t[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
		   m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
	t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
			m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
	t[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
		   m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
	t[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
		   m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
	t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
			m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
	t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
			m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
	t[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
		   m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
	t[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
		   m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
	t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
			m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
	t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
			m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
	float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
	det = 1.0f / det;
	for (int i = 0; i < 16; i++)
		inv[i] = t[i] * det;
}

/* =============================================================================================================
 * known-answer hooks
 * ========================================================================================================== */
uint32_t rfwo_xor128_next(uint32_t s[4]) { return xor128_next(s); }
float rfwo_rng_rand(uint32_t s[4]) { return xor128_rand(s); }
uint32_t rfwo_wang_hash(uint32_t s) { return wang_hash(s); }
uint32_t rfwo_random_int(uint32_t *s) { return random_int(s); }
float rfwo_random_float(uint32_t *s) { return random_float(s); }
float rfwo_half_to_float(uint16_t h) { return half_to_float(h); }
uint32_t rfwo_pack_normal(const float n[3]) { return pack_normal(v3p(n)); }
void rfwo_unpack_normal(uint32_t p, float n[3])
{
	const v3 r = unpack_normal(p);
	n[0] = r.x, n[1] = r.y, n[2] = r.z;
}

/* xor128 is linear over GF(2): state' = M * state with a 128x128 bit matrix.  Jump = product of M^(2^k). */
typedef struct
{
	uint32_t col[128][4];
} gf2m;
static void gf2_apply(const gf2m *m, const uint32_t in[4], uint32_t out[4])
{
	uint32_t r[4] = {0, 0, 0, 0};
	for (int w = 0; w < 4; w++)
		for (int b = 0; b < 32; b++)
			if ((in[w] >> b) & 1u)
			{
				const uint32_t *c = m->col[w * 32 + b];
				r[0] ^= c[0], r[1] ^= c[1], r[2] ^= c[2], r[3] ^= c[3];
			}
	memcpy(out, r, 16);
}
void rfwo_xor128_jump(uint32_t state[4], uint64_t draws)
{
	static gf2m *pw = NULL; /* pw[k] = M^(2^k) */
	if (!pw)
	{
		pw = (gf2m *)malloc(sizeof(gf2m) * 64);
		for (int i = 0; i < 128; i++)
		{
			uint32_t e[4] = {0, 0, 0, 0};
			e[i / 32] = 1u << (i % 32);
			xor128_next(e);
			memcpy(pw[0].col[i], e, 16);
		}
		for (int k = 1; k < 64; k++)
			for (int i = 0; i < 128; i++)
				gf2_apply(&pw[k - 1], pw[k - 1].col[i], pw[k].col[i]);
	}
	for (int k = 0; k < 64; k++)
		if ((draws >> k) & 1ull)
			gf2_apply(&pw[k], state, state);
}

/* bvh_tree.cpp:166-196.  u,v are the weights of p1 and p2 (Embree convention used by Context.cpp:210-211). */
/* Setting "arith" = "reference": the triangle test, the pt primary ray and the pt sky lookup as the REFERENCE's text shapes them —
 * plain products and sums whose contraction into fmas is left to the compiler (the reference builds with -ffast-math -mavx2), strict
 * `t > tt` with no order on equal distances (bvh_tree.cpp:166-196) — instead of the fixed shapes the product states (default,
 * "product": rounded(), fmaf, total order on (t, prim)).  Round 4's advisor: with the oracle following the product's arithmetic, HIP
 * against oracle no longer measures fidelity to the upstream behaviour for these paths; tests/test_parity_gpu.py compares the product
 * with THIS form under round 3's statistical bounds.  Process-wide (the functions below take no context). */
static int g_ref_arith = 0;
static inline int tri_test_ref(v3 org, v3 dir, float t_min, float *t, v3 p0, v3 p1, v3 p2, float *u_out, float *v_out)
{
	const v3 e1 = vsub(p1, p0), e2 = vsub(p2, p0);
	const v3 h = vcross(dir, e2);
	const float a = vdot(e1, h);
	if (a > -TRI_EPS && a < TRI_EPS)
		return 0;
	const float f = 1.f / a;
	const v3 s = vsub(org, p0);
	const float u = f * vdot(s, h);
	if (u < 0.0f || u > 1.0f)
		return 0;
	const v3 q = vcross(s, e1);
	const float v = f * vdot(dir, q);
	if (v < 0.0f || u + v > 1.0f)
		return 0;
	const float tt = f * vdot(e2, q);
	if (tt > t_min && *t > tt)
	{
		*t = tt;
		*u_out = u;
		*v_out = v;
		return 1;
	}
	return 0;
}
/* tie != 0 (closest-hit queries): of two triangles hit at bit-identical distance the lower (instance, primitive id) wins — a total
 * order on (t, instance, prim), so that the hit does not depend on the order a tree happens to present the triangles in (the
 * reference keeps the first it reaches; the product serves the same rays from several traversals and trees: csrc/rt_core.h,
 * tri_test).  Instances are visited in rising order here, so a candidate replaces an equally distant hit only when that hit is of
 * the SAME instance (tie == 1) and has the higher primitive id; tie == 2: the hit so far is of an earlier instance and stays. */
static inline int tri_test_tie(v3 org, v3 dir, float t_min, float *t, v3 p0, v3 p1, v3 p2, float *u_out, float *v_out, int tie, uint32_t prim,
							   uint32_t cur_prim)
{
	if (g_ref_arith)
		return tri_test_ref(org, dir, t_min, t, p0, p1, p2, u_out, v_out);
	/* (fixed-shape arithmetic: rfw_oracle_math.h, rounded()) */
	const v3 e1 = vsub(p1, p0), e2 = vsub(p2, p0);
	const v3 h = vcross_r(dir, e2);
	const float a = vdot_r(e1, h);
	if (a > -TRI_EPS && a < TRI_EPS)
		return 0;
	const float f = 1.f / a;
	const v3 s = vsub(org, p0);
	const float u = rounded(f * vdot_r(s, h));
	if (u < 0.0f || u > 1.0f)
		return 0;
	const v3 q = vcross_r(s, e1);
	const float v = rounded(f * vdot_r(dir, q));
	if (v < 0.0f || u + v > 1.0f)
		return 0;
	const float tt = rounded(f * vdot_r(e2, q));
	if (tt > t_min && (*t > tt || (tie == 1 && *t == tt && prim < cur_prim)))
	{
		*t = tt;
		*u_out = u;
		*v_out = v;
		return 1;
	}
	return 0;
}
static inline int tri_test(v3 org, v3 dir, float t_min, float *t, v3 p0, v3 p1, v3 p2, float *u_out, float *v_out)
{
	return tri_test_tie(org, dir, t_min, t, p0, p1, p2, u_out, v_out, 0, 0u, 0u);
}
int rfwo_intersect_triangle(const float org[3], const float dir[3], float t_min, float *t, const float p0[3],
							const float p1[3], const float p2[3], float *u, float *v)
{
	return tri_test(v3p(org), v3p(dir), t_min, t, v3p(p0), v3p(p1), v3p(p2), u, v);
}

/* aabb.cpp:39-77: hit iff tmax > tmin && tmin < t */
static inline int slab_test(const float *bmin, const float *bmax, v3 org, v3 idir, float t, float *tmin_o,
							float *tmax_o)
{
	const float tx1 = (bmin[0] - org.x) * idir.x, tx2 = (bmax[0] - org.x) * idir.x;
	const float ty1 = (bmin[1] - org.y) * idir.y, ty2 = (bmax[1] - org.y) * idir.y;
	const float tz1 = (bmin[2] - org.z) * idir.z, tz2 = (bmax[2] - org.z) * idir.z;
	const float tmin = fmaxf(fminf(tx1, tx2), fmaxf(fminf(ty1, ty2), fminf(tz1, tz2)));
	const float tmax = fminf(fmaxf(tx1, tx2), fminf(fmaxf(ty1, ty2), fmaxf(tz1, tz2)));
	*tmin_o = tmin;
	*tmax_o = tmax;
	return tmax > tmin && tmin < t;
}
int rfwo_intersect_aabb(const float bmin[3], const float bmax[3], const float org[3], const float inv_dir[3],
						float t, float *tmin, float *tmax)
{
	return slab_test(bmin, bmax, v3p(org), v3p(inv_dir), t, tmin, tmax);
}

/* context.cpp:6-15 — Heron */
float rfwo_triangle_area(const float v0[3], const float v1[3], const float v2[3])
{
	const float a = vlen(vsub(v3p(v1), v3p(v0)));
	const float b = vlen(vsub(v3p(v2), v3p(v1)));
	const float c = vlen(vsub(v3p(v0), v3p(v2)));
	const float s = (a + b + c) * 0.5f;
	return sqrtf(s * (s - a) * (s - b) * (s - c));
}

void rfwo_evaluate_bsdf(const float color[3], const uint32_t params[4], const float iN[3], const float wo[3],
						const float wi[3], float out_rgb[3], float *pdf)
{
	oshading sd;
	sd.color = v3p(color);
	sd.absorption = V3(0, 0, 0);
	memcpy(sd.p, params, 16);
	const v3 r = bsdf_eval(&sd, v3p(iN), v3p(wo), v3p(wi), 0.0f, 0);
	*pdf = bsdf_pdf(&sd, v3p(iN), v3p(wo), v3p(wi));
	out_rgb[0] = r.x, out_rgb[1] = r.y, out_rgb[2] = r.z;
}
void rfwo_sample_bsdf(const float color[3], const float absorption[3], const uint32_t params[4], const float iN[3],
					  const float wo[3], float t, int backfacing, float r3, float r4, float out_rgb[3],
					  float wi[3], float *pdf)
{
	oshading sd;
	sd.color = v3p(color);
	sd.absorption = v3p(absorption);
	memcpy(sd.p, params, 16);
	v3 T, B, R = V3(0, 0, 1);
	create_tangent_space(v3p(iN), &T, &B);
	float p = 0.0f;
	bsdf_sample(&sd, T, B, v3p(iN), v3p(wo), &R, &p, r3, r4);
	const v3 r = bsdf_eval(&sd, v3p(iN), v3p(wo), R, t, backfacing);
	out_rgb[0] = r.x, out_rgb[1] = r.y, out_rgb[2] = r.z;
	wi[0] = R.x, wi[1] = R.y, wi[2] = R.z;
	*pdf = p;
}

/* =============================================================================================================
 * camera — Camera.cpp:74-88, 109-115
 * ========================================================================================================== */
void rfwo_camera_get_view(const rfwhip_camera *c, rfwhip_camera_view *view)
{
	const v3 z = v3p(c->direction);
	const v3 x = vnorm(vcross(z, V3(0.0f, 1.0f, 0.0f)));
	const v3 y = vcross(x, z);
	const v3 pos = v3p(c->position);
	const float pi = 3.14159265358979323846f;
	view->spreadAngle = (c->FOV * pi / 180) / (float)c->pixelCount[1];
	const float screenSize = tanf(c->FOV / 2.0f / (180.0f / pi));
	const v3 center = vadd(pos, vscale(z, c->focalDistance));
	/* screenSize * right * focalDistance * aspectRatio  and  screenSize * focalDistance * up */
	const v3 h = vscale(vscale(vscale(x, screenSize), c->focalDistance), c->aspectRatio);
	const v3 v = vscale(y, screenSize * c->focalDistance);
	const v3 p1 = vadd(vsub(center, h), v), p2 = vadd(vadd(center, h), v), p3 = vsub(vsub(center, h), v);
	view->pos[0] = pos.x, view->pos[1] = pos.y, view->pos[2] = pos.z;
	view->p1[0] = p1.x, view->p1[1] = p1.y, view->p1[2] = p1.z;
	view->p2[0] = p2.x, view->p2[1] = p2.y, view->p2[2] = p2.z;
	view->p3[0] = p3.x, view->p3[1] = p3.y, view->p3[2] = p3.z;
	view->aperture = c->aperture;
}

/* =============================================================================================================
 * BVH2 build — bvh_node.h:56-81 (subdivide, MAX_PRIMITIVES 3, MAX_DEPTH 32), :136-233 (partition<9>)
 * ========================================================================================================== */
static float cen_axis(const v3 *c, int axis) { return axis == 0 ? c->x : (axis == 1 ? c->y : c->z); }

static int bvh_partition(omesh *m, int nodeIdx, int *poolPtr)
{
	rfwhip_bvh_node *node = &m->nodes[nodeIdx];
	const int lFirst = node->left_first, count = node->count;
	int lCount = 0, rFirst = lFirst, rCount = count;
	float lowest = 1e34f, best_split = 0;
	int bestAxis = 0;
	oaabb bestL, bestR, nb;
	aabb_reset(&bestL), aabb_reset(&bestR);
	memcpy(nb.bmin, node->bmin, 12), memcpy(nb.bmax, node->bmax, 12);
	const float parent_cost = aabb_area(&nb) * (float)count;
	const float lengths[3] = {nb.bmax[0] - nb.bmin[0], nb.bmax[1] - nb.bmin[1], nb.bmax[2] - nb.bmin[2]};
	const float bin_size = 1.0f / (float)(9 + 2);
	for (int axis = 0; axis < 3; axis++)
		for (int i = 1; i < 9 + 2; i++)
		{
			const float split = nb.bmin[axis] + lengths[axis] * ((float)i * bin_size);
			int lc = 0, rc = 0;
			oaabb lb, rb;
			aabb_reset(&lb), aabb_reset(&rb);
			for (int k = 0; k < count; k++)
			{
				const uint32_t p = m->prims[lFirst + k];
				if (cen_axis(&m->centroids[p], axis) <= split)
					aabb_grow(&lb, &m->aabbs[p]), lc++;
				else
					aabb_grow(&rb, &m->aabbs[p]), rc++;
			}
			const float cost = aabb_area(&lb) * (float)lc + aabb_area(&rb) * (float)rc;
			if (lowest > cost)
				lowest = cost, best_split = split, bestAxis = axis, bestL = lb, bestR = rb;
		}
	if (parent_cost < lowest)
		return 0;
	for (int k = 0; k < count; k++)
	{
		const uint32_t p = m->prims[lFirst + k];
		if (cen_axis(&m->centroids[p], bestAxis) <= best_split)
		{
			const uint32_t tmp = m->prims[lFirst + k];
			m->prims[lFirst + k] = m->prims[lFirst + lCount];
			m->prims[lFirst + lCount] = tmp;
			lCount++, rFirst++, rCount--;
		}
	}
	const int left = *poolPtr;
	*poolPtr += 2;
	aabb_offset(&bestL, 1e-5f), aabb_offset(&bestR, 1e-5f);
	memcpy(m->nodes[left].bmin, bestL.bmin, 12), memcpy(m->nodes[left].bmax, bestL.bmax, 12);
	m->nodes[left].left_first = lFirst, m->nodes[left].count = lCount;
	memcpy(m->nodes[left + 1].bmin, bestR.bmin, 12), memcpy(m->nodes[left + 1].bmax, bestR.bmax, 12);
	m->nodes[left + 1].left_first = rFirst, m->nodes[left + 1].count = rCount;
	node = &m->nodes[nodeIdx];
	node->left_first = left;
	node->count = -1;
	return 1;
}

static void bvh_subdivide(omesh *m, int nodeIdx, int depth, int *poolPtr)
{
	depth++;
	if (m->nodes[nodeIdx].count < 3 || depth >= 32)
		return;
	if (!bvh_partition(m, nodeIdx, poolPtr))
		return;
	const int left = m->nodes[nodeIdx].left_first;
	if (m->nodes[left].count > 0)
		bvh_subdivide(m, left, depth, poolPtr);
	if (m->nodes[left + 1].count > 0)
		bvh_subdivide(m, left + 1, depth, poolPtr);
}

static void mesh_free(omesh *m)
{
	free(m->verts), free(m->tris), free(m->indices), free(m->p0), free(m->p1), free(m->p2), free(m->aabbs);
	free(m->centroids), free(m->nodes), free(m->prims);
	memset(m, 0, sizeof(*m));
}

/* bvh_tree.cpp:388-452 (per-triangle AABB grown by 1e-5) + root bounds + subdivide */
static void mesh_build(omesh *m)
{
	const size_t n = m->triCount;
	free(m->p0), free(m->p1), free(m->p2), free(m->aabbs), free(m->centroids), free(m->nodes), free(m->prims);
	m->p0 = (v3 *)malloc(sizeof(v3) * (n + 1)), m->p1 = (v3 *)malloc(sizeof(v3) * (n + 1));
	m->p2 = (v3 *)malloc(sizeof(v3) * (n + 1));
	m->aabbs = (oaabb *)malloc(sizeof(oaabb) * (n + 1));
	m->centroids = (v3 *)malloc(sizeof(v3) * (n + 1));
	m->nodes = (rfwhip_bvh_node *)calloc(2 * n + 2, sizeof(rfwhip_bvh_node));
	m->prims = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
	oaabb root;
	aabb_reset(&root);
	for (size_t i = 0; i < n; i++)
	{
		uint32_t i0 = (uint32_t)(3 * i), i1 = i0 + 1, i2 = i0 + 2;
		if (m->indices)
			i0 = m->indices[3 * i], i1 = m->indices[3 * i + 1], i2 = m->indices[3 * i + 2];
		m->p0[i] = v3p(&m->verts[4 * i0]), m->p1[i] = v3p(&m->verts[4 * i1]), m->p2[i] = v3p(&m->verts[4 * i2]);
		aabb_reset(&m->aabbs[i]);
		aabb_grow_p(&m->aabbs[i], m->p0[i]), aabb_grow_p(&m->aabbs[i], m->p1[i]);
		aabb_grow_p(&m->aabbs[i], m->p2[i]);
		aabb_offset(&m->aabbs[i], 1e-5f);
		const oaabb *b = &m->aabbs[i];
		m->centroids[i] = V3((b->bmin[0] + b->bmax[0]) * 0.5f, (b->bmin[1] + b->bmax[1]) * 0.5f,
							 (b->bmin[2] + b->bmax[2]) * 0.5f);
		m->prims[i] = (uint32_t)i;
		aabb_grow(&root, b);
	}
	memcpy(m->nodes[0].bmin, root.bmin, 12), memcpy(m->nodes[0].bmax, root.bmax, 12);
	m->nodes[0].left_first = 0, m->nodes[0].count = (int)n;
	int pool = 2;
	bvh_subdivide(m, 0, 0, &pool);
	m->nodeCount = pool;
}

/* =============================================================================================================
 * traversal — bvh_node.h:317-448 (32-entry stack; both children tested with the current t)
 * ========================================================================================================== */
typedef struct
{
	uint64_t inner, tris;
} tstat;

static int blas_closest(const omesh *m, v3 o, v3 d, float t_min, float *t, int *prim, float *u, float *v, tstat *st, int same_inst)
{
	int valid = 0;
	int todo[64];
	int sp = 0;
	const v3 idir = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	todo[0] = 0;
	while (sp >= 0)
	{
		const rfwhip_bvh_node *node = &m->nodes[todo[sp--]];
		if (node->count > -1)
		{
			for (int i = 0; i < node->count; i++)
			{
				const uint32_t p = m->prims[node->left_first + i];
				st->tris++;
				if (tri_test_tie(o, d, t_min, t, m->p0[p], m->p1[p], m->p2[p], u, v, same_inst ? 1 : 2, p, (uint32_t)*prim))
					same_inst = 1, valid = 1, *prim = (int)p;
			}
		}
		else
		{
			float n1, f1, n2, f2;
			const int l = node->left_first;
			st->inner++;
			const int hl = slab_test(m->nodes[l].bmin, m->nodes[l].bmax, o, idir, *t, &n1, &f1);
			const int hr = slab_test(m->nodes[l + 1].bmin, m->nodes[l + 1].bmax, o, idir, *t, &n2, &f2);
			if (hl && hr)
			{
				if (n1 < n2)
					todo[++sp] = l, todo[++sp] = l + 1;
				else
					todo[++sp] = l + 1, todo[++sp] = l;
			}
			else if (hl)
				todo[++sp] = l;
			else if (hr)
				todo[++sp] = l + 1;
		}
	}
	return valid;
}

static int blas_any(const omesh *m, v3 o, v3 d, float t_min, float t_max, tstat *st)
{
	int todo[64];
	int sp = 0;
	const v3 idir = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	todo[0] = 0;
	while (sp >= 0)
	{
		const rfwhip_bvh_node *node = &m->nodes[todo[sp--]];
		if (node->count > -1)
		{
			for (int i = 0; i < node->count; i++)
			{
				const uint32_t p = m->prims[node->left_first + i];
				float tt = t_max, u, v;
				st->tris++;
				if (tri_test(o, d, t_min, &tt, m->p0[p], m->p1[p], m->p2[p], &u, &v))
					return 1;
			}
		}
		else
		{
			float n1, f1, n2, f2;
			const int l = node->left_first;
			st->inner++;
			const int hl = slab_test(m->nodes[l].bmin, m->nodes[l].bmax, o, idir, t_max, &n1, &f1);
			const int hr = slab_test(m->nodes[l + 1].bmin, m->nodes[l + 1].bmax, o, idir, t_max, &n2, &f2);
			if (hl && hr)
			{
				if (n1 < n2)
					todo[++sp] = l, todo[++sp] = l + 1;
				else
					todo[++sp] = l + 1, todo[++sp] = l;
			}
			else if (hl)
				todo[++sp] = l;
			else if (hr)
				todo[++sp] = l + 1;
		}
	}
	return 0;
}

/* top_level_bvh.cpp:104-168: o' = M^-1 (o,1), d' = M^-1 (d,0), not renormalised => t is shared.  The instance
 * level is a linear loop over world AABBs here (the closest hit does not depend on the TLAS shape). */
static int scene_closest(const rfwo_context *c, v3 o, v3 d, float t_min, float *t, int *inst, int *prim, float *u,
						 float *v, tstat *st)
{
	int hit = 0;
	const v3 idir = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	for (size_t i = 0; i < c->instanceCount; i++)
	{
		const oinstance *in = &c->instances[i];
		if (!in->used || in->mesh >= c->meshCount || !c->meshes[in->mesh].used || !c->meshes[in->mesh].triCount)
			continue;
		const omesh *m = &c->meshes[in->mesh];
		const v3 lo = m4_mul(in->inverse, o, 1.0f), ld = m4_mul(in->inverse, d, 0.0f);
		if (c->use_bvh)
		{
			float a, b;
			if (!slab_test(in->world.bmin, in->world.bmax, o, idir, *t, &a, &b))
				continue;
			if (blas_closest(m, lo, ld, t_min, t, prim, u, v, st, 0)) /* (the hit so far is of an earlier instance) */
				hit = 1, *inst = (int)i;
		}
		else
			for (size_t p = 0; p < m->triCount; p++)
			{
				st->tris++;
				if (tri_test_tie(lo, ld, t_min, t, m->p0[p], m->p1[p], m->p2[p], u, v, (hit && *inst == (int)i) ? 1 : 2, (uint32_t)p, (uint32_t)*prim))
					hit = 1, *inst = (int)i, *prim = (int)p;
			}
	}
	return hit;
}

static int scene_occluded(const rfwo_context *c, v3 o, v3 d, float t_min, float t_max, tstat *st)
{
	const v3 idir = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	for (size_t i = 0; i < c->instanceCount; i++)
	{
		const oinstance *in = &c->instances[i];
		if (!in->used || in->mesh >= c->meshCount || !c->meshes[in->mesh].used || !c->meshes[in->mesh].triCount)
			continue;
		const omesh *m = &c->meshes[in->mesh];
		const v3 lo = m4_mul(in->inverse, o, 1.0f), ld = m4_mul(in->inverse, d, 0.0f);
		if (c->use_bvh)
		{
			float a, b;
			if (!slab_test(in->world.bmin, in->world.bmax, o, idir, t_max, &a, &b))
				continue;
			if (blas_any(m, lo, ld, t_min, t_max, st))
				return 1;
		}
		else
			for (size_t p = 0; p < m->triCount; p++)
			{
				float tt = t_max, u, v;
				st->tris++;
				if (tri_test(lo, ld, t_min, &tt, m->p0[p], m->p1[p], m->p2[p], &u, &v))
					return 1;
			}
	}
	return 0;
}

/* =============================================================================================================
 * context plumbing
 * ========================================================================================================== */
int rfwo_create(int device_ordinal, int rank, int world, rfwo_context **out)
{
	(void)device_ordinal;
	if (!out || world < 1 || rank < 0 || rank >= world)
		return fail("rfwo_create: bad arguments");
	rfwo_context *c = (rfwo_context *)calloc(1, sizeof(*c));
	c->rank = rank, c->world = world;
	c->integrator = 0, c->spp = 1, c->max_depth = 2, c->jitter = 0, c->use_bvh = 1, c->threads = 0;
	c->rng[0] = 123456789u, c->rng[1] = 362436069u, c->rng[2] = 521288629u, c->rng[3] = 88675123u;
	*out = c;
	return 0;
}
int rfwo_cleanup(rfwo_context *c)
{
	(void)c;
	return 0;
}
void rfwo_destroy(rfwo_context *c)
{
	if (!c)
		return;
	for (size_t i = 0; i < c->meshCount; i++)
		mesh_free(&c->meshes[i]);
	for (size_t i = 0; i < c->textureCount; i++)
		free(c->textures[i].data);
	free(c->meshes), free(c->instances), free(c->materials), free(c->textures), free(c->sky), free(c->area);
	free(c->point), free(c->spot), free(c->dir), free(c->acc), free(c->hit_t), free(c->hit_u), free(c->hit_v);
	free(c->hit_prim), free(c->hit_inst), free(c->blue_noise), free(c->pend);
	free(c);
}
int rfwo_init(rfwo_context *c, uint32_t w, uint32_t h)
{
	if (!c || !w || !h)
		return fail("rfwo_init: bad size");
	c->W = w, c->H = h;
	const size_t n = (size_t)w * h;
	free(c->acc), free(c->hit_t), free(c->hit_u), free(c->hit_v), free(c->hit_prim), free(c->hit_inst);
	c->acc = (float *)calloc(n * 4, sizeof(float));
	c->hit_t = (float *)calloc(n, 4), c->hit_u = (float *)calloc(n, 4), c->hit_v = (float *)calloc(n, 4);
	c->hit_prim = (int32_t *)calloc(n, 4), c->hit_inst = (int32_t *)calloc(n, 4);
	c->samples = 0;
	return 0;
}
int rfwo_set_sky(rfwo_context *c, const float *rgb, size_t w, size_t h)
{
	free(c->sky);
	c->sky = (float *)malloc(w * h * 12);
	memcpy(c->sky, rgb, w * h * 12);
	c->skyW = w, c->skyH = h;
	return 0;
}
int rfwo_set_textures(rfwo_context *c, const rfwhip_texture *t, size_t count)
{
	for (size_t i = 0; i < c->textureCount; i++)
		free(c->textures[i].data);
	free(c->textures);
	c->textures = (otexture *)calloc(count ? count : 1, sizeof(otexture));
	c->textureCount = count;
	for (size_t i = 0; i < count; i++)
	{
		otexture *o = &c->textures[i];
		o->type = t[i].type, o->width = t[i].width, o->height = t[i].height, o->texelCount = t[i].texelCount;
		o->bytes = (size_t)t[i].texelCount * (t[i].type == RFWHIP_TEX_FLOAT4 ? 16 : 4);
		o->data = malloc(o->bytes ? o->bytes : 4);
		memcpy(o->data, t[i].data, o->bytes);
	}
	return 0;
}
int rfwo_set_materials(rfwo_context *c, const rfwhip_material *m, const rfwhip_material_tex_ids *ids, size_t count)
{
	free(c->materials);
	c->materials = (rfwhip_material *)malloc(sizeof(rfwhip_material) * (count ? count : 1));
	memcpy(c->materials, m, sizeof(rfwhip_material) * count);
	if (ids)
	{
		/* per-slot texture ids as a backend resolves them (CUDART/src/Context.cpp:171-190) */
		static const int slot_of_id[11] = {0, 1, 2, 3, 4, 5, 6, 7, -1, 8, 9};
		for (size_t i = 0; i < count; i++)
			for (int k = 0; k < 11; k++)
				if (slot_of_id[k] >= 0 && ids[i].texture[k] != -1)
					c->materials[i].map[slot_of_id[k]].addr = (uint32_t)ids[i].texture[k];
	}
	c->materialCount = count;
	return 0;
}
int rfwo_set_mesh(rfwo_context *c, size_t index, const rfwhip_mesh *mesh)
{
	if (!mesh || !mesh->vertices || !mesh->triangles)
		return fail("rfwo_set_mesh: null mesh data");
	if (index >= c->meshCount)
	{
		c->meshes = (omesh *)realloc(c->meshes, sizeof(omesh) * (index + 1));
		memset(&c->meshes[c->meshCount], 0, sizeof(omesh) * (index + 1 - c->meshCount));
		c->meshCount = index + 1;
	}
	omesh *m = &c->meshes[index];
	free(m->verts), free(m->tris), free(m->indices);
	m->vertexCount = mesh->vertexCount, m->triCount = mesh->triangleCount;
	m->verts = (float *)malloc(16 * mesh->vertexCount + 16);
	memcpy(m->verts, mesh->vertices, 16 * mesh->vertexCount);
	m->tris = (rfwhip_triangle *)malloc(160 * mesh->triangleCount + 160);
	memcpy(m->tris, mesh->triangles, 160 * mesh->triangleCount);
	m->indices = NULL;
	if (mesh->indices)
	{
		m->indices = (uint32_t *)malloc(12 * mesh->triangleCount + 12);
		memcpy(m->indices, mesh->indices, 12 * mesh->triangleCount);
	}
	m->used = 1;
	mesh_build(m); /* the oracle always rebuilds: a refit BVH returns the same closest hits */
	return 0;
}
int rfwo_set_instance(rfwo_context *c, size_t i, size_t mesh, const float *t16, const float *n9)
{
	if (i >= c->instanceCount)
	{
		c->instances = (oinstance *)realloc(c->instances, sizeof(oinstance) * (i + 1));
		memset(&c->instances[c->instanceCount], 0, sizeof(oinstance) * (i + 1 - c->instanceCount));
		c->instanceCount = i + 1;
	}
	oinstance *in = &c->instances[i];
	in->used = 1, in->mesh = mesh;
	memcpy(in->transform, t16, 64), memcpy(in->normal, n9, 36);
	mat4_inverse(in->transform, in->inverse);
	return 0;
}
int rfwo_set_lights(rfwo_context *c, rfwhip_light_count n, const rfwhip_area_light *a, const rfwhip_point_light *p,
					const rfwhip_spot_light *s, const rfwhip_directional_light *d)
{
	free(c->area), free(c->point), free(c->spot), free(c->dir);
	c->lc = n;
	c->area = (rfwhip_area_light *)malloc(96 * (n.areaLightCount + 1));
	c->point = (rfwhip_point_light *)malloc(32 * (n.pointLightCount + 1));
	c->spot = (rfwhip_spot_light *)malloc(48 * (n.spotLightCount + 1));
	c->dir = (rfwhip_directional_light *)malloc(32 * (n.directionalLightCount + 1));
	if (n.areaLightCount)
		memcpy(c->area, a, 96 * n.areaLightCount);
	if (n.pointLightCount)
		memcpy(c->point, p, 32 * n.pointLightCount);
	if (n.spotLightCount)
		memcpy(c->spot, s, 48 * n.spotLightCount);
	if (n.directionalLightCount)
		memcpy(c->dir, d, 32 * n.directionalLightCount);
	return 0;
}
int rfwo_update(rfwo_context *c)
{
	for (size_t i = 0; i < c->instanceCount; i++)
	{
		oinstance *in = &c->instances[i];
		if (!in->used || in->mesh >= c->meshCount || !c->meshes[in->mesh].used || !c->meshes[in->mesh].triCount)
			continue;
		const rfwhip_bvh_node *r = &c->meshes[in->mesh].nodes[0];
		aabb_reset(&in->world);
		for (int k = 0; k < 8; k++)
		{
			const v3 p = V3(k & 1 ? r->bmax[0] : r->bmin[0], k & 2 ? r->bmax[1] : r->bmin[1],
							k & 4 ? r->bmax[2] : r->bmin[2]);
			aabb_grow_p(&in->world, m4_mul(in->transform, p, 1.0f));
		}
		aabb_offset(&in->world, 1e-4f);
	}
	return 0;
}
int rfwo_set_setting(rfwo_context *c, const char *key, const char *val)
{
	if (!strcmp(key, "integrator"))
	{
		if (!strcmp(val, "parity"))
			c->integrator = 0;
		else if (!strcmp(val, "pt"))
			c->integrator = 1;
		else
			return fail("integrator must be parity|pt");
	}
	else if (!strcmp(key, "spp"))
		c->spp = atoi(val) > 0 ? atoi(val) : 1;
	else if (!strcmp(key, "max_depth"))
		c->max_depth = atoi(val);
	else if (!strcmp(key, "jitter"))
	{
		if (!strcmp(val, "xor128"))
			c->jitter = 0;
		else if (!strcmp(val, "center"))
			c->jitter = 1;
		else
			return fail("jitter must be xor128|center");
	}
	else if (!strcmp(key, "sampler"))
	{
		if (!strcmp(val, "hash"))
			c->sampler = 0;
		else if (!strcmp(val, "bluenoise"))
			c->sampler = 1;
		else
			return fail("sampler must be hash|bluenoise");
	}
	else if (!strcmp(key, "bvh"))
		c->use_bvh = atoi(val) != 0;
	else if (!strcmp(key, "arith"))
	{
		if (!strcmp(val, "product"))
			g_ref_arith = 0;
		else if (!strcmp(val, "reference"))
			g_ref_arith = 1;
		else
			return fail("arith must be product|reference");
	}
	else if (!strcmp(key, "threads"))
		c->threads = atoi(val);
	else if (!strcmp(key, "stage_timing") || !strcmp(key, "count_traversal") || !strcmp(key, "lds_nodes") || !strcmp(key, "refill") || !strcmp(key, "streams") || !strcmp(key, "builder"))
		return 0;
	else
		return fail("unknown setting");
	return 0;
}
/* ---- host skinning: geometry/gltf/mesh.cpp:31-45 (4x4 blend, general 4x4 inverse, row vector times inverse) ---- */
static int invert4(const float m[16], float inv[16])
{
	/* cofactor expansion of a column-major 4x4 */
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
	const float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
	if (det == 0.0f)
		return 0;
	for (int i = 0; i < 16; i++)
		inv[i] = t[i] / det;
	return 1;
}
void rfwo_skin_vertices(const float *base_v4, const float *base_n4, const uint32_t *joints4, const float *weights4,
						const float *mats16, uint32_t joint_count, size_t n, float *out_v4, float *out_n4)
{
	for (size_t i = 0; i < n; i++)
	{
		float m[16] = {0};
		for (int k = 0; k < 4; k++)
		{
			uint32_t j = joints4[4 * i + k];
			if (j >= joint_count)
				j = 0;
			for (int e = 0; e < 16; e++)
				m[e] += mats16[16 * j + e] * weights4[4 * i + k];
		}
		const float *b = base_v4 + 4 * i;
		for (int r = 0; r < 4; r++)
			out_v4[4 * i + r] = m[r] * b[0] + m[4 + r] * b[1] + m[8 + r] * b[2] + m[12 + r] * b[3];
		float inv[16];
		if (!invert4(m, inv))
			memset(inv, 0, sizeof(inv));
		/* row vector times matrix: result[c] = sum_r n[r] * inv(r, c), element (r, c) = inv[c * 4 + r] */
		const float *nb = base_n4 + 4 * i;
		float r3[3];
		for (int c = 0; c < 3; c++)
			r3[c] = nb[0] * inv[c * 4 + 0] + nb[1] * inv[c * 4 + 1] + nb[2] * inv[c * 4 + 2] + 0.0f * inv[c * 4 + 3];
		const float len = sqrtf(r3[0] * r3[0] + r3[1] * r3[1] + r3[2] * r3[2]);
		out_n4[4 * i + 0] = r3[0] / len, out_n4[4 * i + 1] = r3[1] / len, out_n4[4 * i + 2] = r3[2] / len, out_n4[4 * i + 3] = 0.0f;
	}
}

#define BLUE_NOISE_WORDS (5u * 65536u)
int rfwo_set_blue_noise(rfwo_context *c, const uint32_t *table, size_t words)
{
	if (!table || words < BLUE_NOISE_WORDS)
		return fail("rfwo_set_blue_noise: the table has 5 x 65536 words");
	free(c->blue_noise);
	c->blue_noise = (uint32_t *)malloc(BLUE_NOISE_WORDS * 4);
	memcpy(c->blue_noise, table, BLUE_NOISE_WORDS * 4);
	return 0;
}
/* blueNoiseSampler, bsdf/tools.h:163-181 / CUDART/src/Kernels.cu:205-223; table layout blue_noise.h:8204 */
float rfwo_blue_noise_sample(const uint32_t *table, int x, int y, int sampleIdx, int dim)
{
	x &= 127, y &= 127, sampleIdx &= 255, dim &= 255;
	uint32_t ri = (uint32_t)dim + (uint32_t)(x + y * 128) * 8u + 65536u * 3u;
	if (ri >= BLUE_NOISE_WORDS)
		ri = BLUE_NOISE_WORDS - 1u;
	const int ranked = (sampleIdx ^ (int)table[ri]) & 255;
	int value = (int)table[dim + ranked * 256];
	value ^= (int)table[(dim & 7) + (x + y * 128) * 8 + 65536];
	return (0.5f + (float)value) * (1.0f / 256.0f);
}
int rfwo_set_probe_index(rfwo_context *c, uint32_t x, uint32_t y)
{
	c->probe_x = x, c->probe_y = y;
	return 0;
}
int rfwo_get_probe_results(rfwo_context *c, uint32_t *inst, uint32_t *prim, float *dist)
{
	*inst = c->probe_inst, *prim = c->probe_prim, *dist = c->probe_dist;
	return 0;
}
int rfwo_get_stats(rfwo_context *c, rfwhip_render_stats *s)
{
	*s = c->stats;
	return 0;
}
int rfwo_get_counters(rfwo_context *c, uint64_t out[8], int reset)
{
	memcpy(out, c->cnt, sizeof(c->cnt));
	if (reset)
		memset(c->cnt, 0, sizeof(c->cnt));
	return 0;
}
int rfwo_wait(rfwo_context *c)
{
	(void)c;
	return 0;
}
/* strips are dealt to the ranks forwards in even periods of `world` strips and backwards in odd ones (product: rt::strip_owner) */
static int owns_row(const rfwo_context *c, uint32_t y)
{
	const uint32_t strip = y / STRIP, w = (uint32_t)c->world, k = strip / w, pos = strip % w;
	return (int)((k & 1u) ? w - 1u - pos : pos) == c->rank;
}
uint32_t rfwo_local_rows(const rfwo_context *c)
{
	const uint32_t strips = (c->H + STRIP - 1) / STRIP;
	return ((strips + c->world - 1) / c->world) * STRIP;
}
int rfwo_read_framebuffer(rfwo_context *c, float *rgba)
{
	const size_t n = (size_t)c->W * c->H * 4;
	const float s = c->samples ? 1.0f / (float)c->samples : 0.0f;
	for (size_t i = 0; i < n; i++)
		rgba[i] = c->acc[i] * s;
	return 0;
}
int rfwo_read_local_framebuffer(rfwo_context *c, float *rgba)
{
	const uint32_t rows = rfwo_local_rows(c);
	memset(rgba, 0, (size_t)rows * c->W * 16);
	const float s = c->samples ? 1.0f / (float)c->samples : 0.0f;
	for (uint32_t y = 0; y < c->H; y++)
	{
		if (!owns_row(c, y))
			continue;
		const uint32_t ly = (y / STRIP / c->world) * STRIP + (y % STRIP);
		for (uint32_t i = 0; i < c->W * 4; i++)
			rgba[(size_t)ly * c->W * 4 + i] = c->acc[(size_t)y * c->W * 4 + i] * s;
	}
	return 0;
}
int rfwo_read_primary_hits(rfwo_context *c, float *t, int32_t *prim, int32_t *inst, float *u, float *v)
{
	const size_t n = (size_t)c->W * c->H;
	if (t)
		memcpy(t, c->hit_t, n * 4);
	if (prim)
		memcpy(prim, c->hit_prim, n * 4);
	if (inst)
		memcpy(inst, c->hit_inst, n * 4);
	if (u)
		memcpy(u, c->hit_u, n * 4);
	if (v)
		memcpy(v, c->hit_v, n * 4);
	return 0;
}
int rfwo_get_bvh(rfwo_context *c, size_t mi, rfwhip_bvh_node *nodes, size_t node_cap, uint32_t *prims,
				 size_t prim_cap, size_t *node_count, size_t *prim_count)
{
	if (mi >= c->meshCount || !c->meshes[mi].used)
		return fail("rfwo_get_bvh: no such mesh");
	const omesh *m = &c->meshes[mi];
	if (node_count)
		*node_count = (size_t)m->nodeCount;
	if (prim_count)
		*prim_count = m->triCount;
	if (nodes)
		memcpy(nodes, m->nodes, sizeof(rfwhip_bvh_node) * (node_cap < (size_t)m->nodeCount ? node_cap : (size_t)m->nodeCount));
	if (prims)
		memcpy(prims, m->prims, 4 * (prim_cap < m->triCount ? prim_cap : m->triCount));
	return 0;
}

int rfwo_trace_rays(rfwo_context *c, size_t n, const float *org, const float *dir, float t_min, float t_max, float *t,
					int32_t *prim, int32_t *inst, float *u, float *v)
{
	uint64_t tinner = 0, ttris = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : tinner, ttris)
	for (long i = 0; i < (long)n; i++)
	{
		float tt = t_max, uu = 0, vv = 0;
		int ii = -1, pp = -1;
		tstat st = {0, 0};
		const int hit = scene_closest(c, v3p(org + 3 * i), v3p(dir + 3 * i), t_min, &tt, &ii, &pp, &uu, &vv, &st);
		tinner += st.inner, ttris += st.tris;
		if (t)
			t[i] = tt;
		if (u)
			u[i] = uu;
		if (v)
			v[i] = vv;
		if (prim)
			prim[i] = hit ? pp : -1;
		if (inst)
			inst[i] = hit ? ii : -1;
	}
	c->cnt[0] += n, c->cnt[2] += tinner, c->cnt[3] += ttris;
	return 0;
}

/* =============================================================================================================
 * shading helpers shared by both integrators
 * ========================================================================================================== */
static inline v3 mat_color(const rfwhip_material *m)
{
	return V3(half_to_float(m->diffuse[0]), half_to_float(m->diffuse[1]), half_to_float(m->diffuse[2]));
}
static inline int mat_flag(const rfwhip_material *m, int f) { return (m->flags >> f) & 1u; }

/* =============================================================================================================
 * PARITY INTEGRATOR — EmbreeRT/src/Context.cpp:104-300, retrieve_material :417-476
 * ========================================================================================================== */
typedef struct
{
	v3 pos, right, up, p1;
	float aperture;
} camparams;

/* Ray.cpp:16-47 — scalar form.  The AVX packet form actually executed (Ray.cpp:176-384) differs only when
 * aperture != 0; its summation order for the pixel point and the normalisation is used here. */
static void parity_ray(const camparams *cp, uint32_t W, uint32_t H, int x, int y, float r0, float r1, float r2,
					   float r3, v3 *O, v3 *D)
{
	v3 org = cp->pos;
	if (cp->aperture != 0.0f)
	{
		const float blade = (float)(int)(r0 * 9);
		r2 = (r2 - blade * (1.0f / 9.0f)) * 9.0f;
		const float piOver4point5 = 3.14159265359f / 4.5f;
		const float x1 = cosf(blade * piOver4point5), y1 = sinf(blade * piOver4point5);
		const float x2 = cosf((blade + 1.0f) * piOver4point5), y2 = sinf((blade + 1.0f) * piOver4point5);
		if ((r2 + r3) > 1.0f)
			r2 = 1.0f - r2, r3 = 1.0f - r3;
		const float xr = x1 * r2 + x2 * r3, yr = y1 * r2 + y2 * r3;
		org = vadd(cp->pos, vscale(vadd(vscale(cp->right, xr), vscale(cp->up, yr)), cp->aperture));
	}
	const float u = ((float)x + r0) * (1.0f / (float)W);
	const float v = ((float)y + r1) * (1.0f / (float)H);
	const v3 pix = vadd(cp->p1, vadd(vscale(cp->right, u), vscale(cp->up, v)));
	const v3 d = vsub(pix, org);
	float l2 = d.x * d.x;
	l2 = d.y * d.y + l2;
	l2 = d.z * d.z + l2;
	const float inv = 1.0f / sqrtf(l2);
	*O = org;
	*D = vscale(d, inv);
}

static v3 parity_shade(rfwo_context *c, v3 O, v3 D, float t, int inst, int prim, float u, float v, float *alpha,
					   tstat *st, uint64_t *nshadow)
{
	const oinstance *in = &c->instances[inst];
	const omesh *mesh = &c->meshes[in->mesh];
	const rfwhip_triangle *tri = &mesh->tris[prim];
	const v3 bary = V3(1.0f - u - v, u, v);
	const v3 p = vadd(O, vscale(D, t));
	const rfwhip_material *mat = &c->materials[tri->material];
	/* retrieve_material */
	const v3 iNl =
		vadd(vadd(vscale(v3p(tri->vN0), bary.x), vscale(v3p(tri->vN1), bary.y)), vscale(v3p(tri->vN2), bary.z));
	const v3 iN = vnorm(m3_mul(in->normal, iNl));
	v3 color = mat_color(mat);
	if (mat_flag(mat, RFWHIP_MAT_HAS_DIFFUSE_MAP))
	{
		const float tu = bary.x * tri->u0 + bary.y * tri->u1 + bary.z * tri->u2;
		const float tv = bary.x * tri->v0 + bary.y * tri->v1 + bary.z * tri->v2;
		const rfwhip_map_desc *md = &mat->map[0];
		const float uu = (tu + half_to_float(md->uoffs)) * half_to_float(md->uscale);
		const float vv = (tv + half_to_float(md->voffs)) * half_to_float(md->vscale);
		float tx = fmodf(uu, 1.0f), ty = fmodf(vv, 1.0f);
		if (tx < 0.f)
			tx = 1.f + tx;
		if (ty < 0.f)
			ty = 1.f + ty;
		if (md->addr < c->textureCount)
		{
			const otexture *tex = &c->textures[md->addr];
			const uint32_t ix = f2u_sat(tx * (float)(tex->width - 1)), iy = f2u_sat(ty * (float)(tex->height - 1));
			const int id = (int)(iy * tex->width + ix);
			if (tex->type == RFWHIP_TEX_FLOAT4)
			{
				const float *px = (const float *)tex->data + 4 * (size_t)id;
				color = vmul(color, V3(px[0], px[1], px[2]));
			}
			/* Context.cpp:458-472: the FLOAT4 case has no break and falls through into the UINT decode */
			const uint32_t tc = ((const uint32_t *)tex->data)[id];
			const float sc = 1.0f / 256.0f;
			color = vmul(vscale(color, sc), V3((float)(tc & 0xFFu), (float)((tc >> 8) & 0xFFu), (float)((tc >> 16) & 0xFFu)));
		}
	}
	*alpha = 1.0f;
	if (color.x > 1.0f || color.y > 1.0f || color.z > 1.0f)
		return color;
	v3 contrib = V3(0.1f, 0.1f, 0.1f);
	for (uint32_t i = 0; i < c->lc.areaLightCount; i++)
	{
		const rfwhip_area_light *l = &c->area[i];
		v3 L = vsub(v3p(l->position), p);
		const float sq = vdot(L, L), dist = sqrtf(sq);
		L = V3(L.x / dist, L.y / dist, L.z / dist);
		const float NdotL = vdot(iN, L), LNdotL = -vdot(v3p(l->normal), L);
		if (NdotL <= 0 || LNdotL <= 0)
			continue;
		(*nshadow)++;
		if (!scene_occluded(c, p, L, 1e-4f, dist, st))
		{
			/* l.radiance * l.area / sq_dist * NdotL * LNdotL */
			v3 r = vscale(v3p(l->radiance), l->area);
			r = V3(r.x / sq, r.y / sq, r.z / sq);
			contrib = vadd(contrib, vscale(vscale(r, NdotL), LNdotL));
		}
	}
	for (uint32_t i = 0; i < c->lc.pointLightCount; i++)
	{
		const rfwhip_point_light *l = &c->point[i];
		v3 L = vsub(v3p(l->position), p);
		const float sq = vdot(L, L), dist = sqrtf(sq);
		L = V3(L.x / dist, L.y / dist, L.z / dist);
		const float NdotL = vdot(iN, L);
		if (NdotL <= 0)
			continue;
		(*nshadow)++;
		if (!scene_occluded(c, p, L, 1e-4f, dist, st))
		{
			const v3 r = V3(l->radiance[0] / sq, l->radiance[1] / sq, l->radiance[2] / sq);
			contrib = vadd(contrib, vscale(r, NdotL));
		}
	}
	return vmul(color, contrib);
}

static v3 parity_sky(const rfwo_context *c, v3 D)
{
	if (!c->sky || !c->skyW || !c->skyH)
		return V3(0, 0, 0);
	const float inv_pi = 0.318309886183790671538f;
	const float ux = 0.5f * (1.0f + atan2f(D.x, -D.z) * inv_pi);
	const float uy = acosf(fclamp(D.y, -1.0f, 1.0f)) * inv_pi;
	uint32_t px = f2u_sat(ux * (float)(c->skyW - 1)), py = f2u_sat(uy * (float)(c->skyH - 1));
	if (px >= c->skyW)
		px = (uint32_t)c->skyW - 1;
	if (py >= c->skyH)
		py = (uint32_t)c->skyH - 1;
	return v3p(&c->sky[3 * (py * c->skyW + px)]);
}

static void render_parity_sample(rfwo_context *c, const camparams *cp)
{
	const uint32_t W = c->W, H = c->H;
	const int npx = (int)(W / 4), npy = (int)(H / 2);
	uint32_t(*rows)[4] = NULL;
	if (c->jitter == 0)
	{
		rows = (uint32_t(*)[4])malloc(sizeof(uint32_t[4]) * (npy + 1));
		uint32_t s[4];
		memcpy(s, c->rng, 16);
		for (int r = 0; r < npy; r++)
		{
			memcpy(rows[r], s, 16);
			rfwo_xor128_jump(s, (uint64_t)npx * 32);
		}
		memcpy(c->rng, s, 16);
	}
	const int probe_id = (int)(c->probe_y * W + c->probe_x);
	uint64_t tinner = 0, ttris = 0, sinner = 0, stris = 0, nsh = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : tinner, ttris, sinner, stris, nsh)
	for (int yl = 0; yl < npy; yl++)
	{
		if (!owns_row(c, (uint32_t)yl * 2))
			continue;
		uint32_t s[4] = {0, 0, 0, 0};
		if (rows)
			memcpy(s, rows[yl], 16);
		for (int xl = 0; xl < npx; xl++)
		{
			/* Ray.cpp:213-216: 8 x r0, 8 x r1, 8 x r2, 8 x r3; lane 0 drawn first within each group */
			float r[4][8];
			for (int g = 0; g < 4; g++)
				for (int j = 0; j < 8; j++)
					r[g][j] = rows ? xor128_rand(s) : 0.5f;
			for (int j = 0; j < 8; j++)
			{
				const int x = xl * 4 + (j & 3), y = yl * 2 + (j >> 2);
				const int pid = y * (int)W + x;
				v3 O, D;
				parity_ray(cp, W, H, x, y, r[0][j], r[1][j], r[2][j], r[3][j], &O, &D);
				float t = 1e34f, u = 0, v = 0;
				int inst = -1, prim = -1;
				tstat st = {0, 0}, ss = {0, 0};
				const int hit = scene_closest(c, O, D, 1e-5f, &t, &inst, &prim, &u, &v, &st);
				tinner += st.inner, ttris += st.tris;
				c->hit_t[pid] = t, c->hit_u[pid] = u, c->hit_v[pid] = v;
				c->hit_prim[pid] = hit ? prim : -1, c->hit_inst[pid] = hit ? inst : -1;
				v3 col;
				float alpha = 0.0f;
				if (!hit)
					col = parity_sky(c, D);
				else
				{
					if (pid == probe_id)
						c->probe_dist = t, c->probe_inst = (uint32_t)inst, c->probe_prim = (uint32_t)prim;
					uint64_t n = 0;
					col = parity_shade(c, O, D, t, inst, prim, u, v, &alpha, &ss, &n);
					sinner += ss.inner, stris += ss.tris, nsh += n;
				}
				float *a = &c->acc[(size_t)pid * 4];
				a[0] += col.x, a[1] += col.y, a[2] += col.z, a[3] += alpha;
			}
		}
	}
	free(rows);
	c->cnt[2] += tinner, c->cnt[3] += ttris, c->cnt[4] += sinner, c->cnt[5] += stris, c->cnt[1] += nsh;
	c->stats.shadowCount += (uint32_t)nsh;
}

/* =============================================================================================================
 * PATH-TRACING INTEGRATOR — CUDART/src/Kernels.cu:383-426 (generate), :571-794 (shade), lights.h, getShadingData.h
 * ========================================================================================================== */
static v3 pt_sky(const rfwo_context *c, v3 D)
{
	if (!c->sky || !c->skyW || !c->skyH)
		return V3(0, 0, 0);
	const float inv_pi = 0.318309886183790671538f;
	const float turns = rounded(atan2f(D.x, -D.z) * inv_pi); /* (the product rounded on its own: csrc/rt_core.h, pt_sky) */
	const uint32_t u = g_ref_arith ? f2u_sat((float)c->skyW * 0.5f * (1.0f + atan2f(D.x, -D.z) * inv_pi)) /* Kernels.cu:596-598 as written */
								   : f2u_sat((float)c->skyW * 0.5f * (1.0f + turns));
	const uint32_t v = f2u_sat((float)c->skyH * acosf(fclamp(D.y, -1.0f, 1.0f)) * inv_pi);
	const uint64_t idx = (uint64_t)u + (uint64_t)v * c->skyW;
	if (idx < (uint64_t)c->skyW * c->skyH)
		return v3p(&c->sky[3 * idx]);
	return V3(0, 0, 0);
}

/* lights.h:17-76 */
static float pot_area(const rfwo_context *c, int idx, v3 O, v3 N, v3 I, v3 bary)
{
	const rfwhip_area_light *l = &c->area[idx];
	v3 L = I;
	if (bary.x >= 0)
		L = vadd(vadd(vscale(v3p(l->vertex0), bary.x), vscale(v3p(l->vertex1), bary.y)), vscale(v3p(l->vertex2), bary.z));
	L = vsub(L, O);
	const float att = 1.0f / vdot(L, L);
	L = vnorm(L);
	const float LNdotL = fmaxf(0.0f, -vdot(v3p(l->normal), L));
	const float NdotL = fmaxf(0.0f, vdot(N, L));
	return l->energy * LNdotL * NdotL * att;
}
static float pot_point(const rfwo_context *c, int idx, v3 I, v3 N)
{
	const rfwhip_point_light *l = &c->point[idx];
	const v3 L = vsub(v3p(l->position), I);
	const float NdotL = fmaxf(0.0f, vdot(N, L));
	const float att = 1.0f / vdot(L, L);
	return l->energy * NdotL * att;
}
static float pot_spot(const rfwo_context *c, int idx, v3 I, v3 N)
{
	const rfwhip_spot_light *l = &c->spot[idx];
	v3 L = vsub(v3p(l->position), I);
	const float att = 1.0f / vdot(L, L);
	L = vnorm(L);
	const float d = (fmaxf(0.0f, -vdot(L, v3p(l->direction))) - l->cosOuter) / (l->cosInner - l->cosOuter);
	const float NdotL = fmaxf(0.0f, vdot(N, L));
	const float LNdotL = fmaxf(0.0f, fminf(1.0f, d));
	return l->energy * LNdotL * NdotL * att;
}
static float pot_dir(const rfwo_context *c, int idx, v3 N)
{
	const rfwhip_directional_light *l = &c->dir[idx];
	return l->energy * fmaxf(0.0f, -vdot(v3p(l->direction), N));
}
static uint32_t total_lights(const rfwo_context *c)
{
	return c->lc.areaLightCount + c->lc.pointLightCount + c->lc.spotLightCount + c->lc.directionalLightCount;
}
/* potential of light k in the fixed order area, point, spot, directional */
static float pot_any(const rfwo_context *c, uint32_t k, v3 I, v3 N, v3 bary)
{
	if (k < c->lc.areaLightCount)
		return pot_area(c, (int)k, I, N, V3(0, 0, 0), bary);
	k -= c->lc.areaLightCount;
	if (k < c->lc.pointLightCount)
		return pot_point(c, (int)k, I, N);
	k -= c->lc.pointLightCount;
	if (k < c->lc.spotLightCount)
		return pot_spot(c, (int)k, I, N);
	k -= c->lc.spotLightCount;
	return pot_dir(c, (int)k, N);
}
/* lights.h:83-116 (IS_LIGHTS 1) */
static float light_pick_prob(const rfwo_context *c, int idx, v3 O, v3 N, v3 I)
{
	float sum = 0, mine = 0;
	for (uint32_t i = 0; i < c->lc.areaLightCount; i++)
	{
		const float p = pot_area(c, (int)i, O, N, I, V3(-1, -1, -1));
		if ((int)i == idx)
			mine = p;
		sum += p;
	}
	for (uint32_t i = 0; i < c->lc.pointLightCount; i++)
		sum += pot_point(c, (int)i, O, N);
	for (uint32_t i = 0; i < c->lc.spotLightCount; i++)
		sum += pot_spot(c, (int)i, O, N);
	for (uint32_t i = 0; i < c->lc.directionalLightCount; i++)
		sum += pot_dir(c, (int)i, N);
	if (sum <= 0)
		return 0;
	return mine / sum;
}
/* lights.h:119-157 */
static v3 random_barycentrics(float r0)
{
	const uint32_t uf = f2u_sat(r0 * 4294967295.0f);
	float Ax = 1.f, Ay = 0.f, Bx = 0.f, By = 1.f, Cx = 0.f, Cy = 0.f;
	for (int i = 0; i < 16; ++i)
	{
		const int d = (int)((uf >> (2 * (15 - i))) & 0x3u);
		float Anx, Any, Bnx, Bny, Cnx, Cny;
		switch (d)
		{
		case 0:
			Anx = (Bx + Cx) * 0.5f, Any = (By + Cy) * 0.5f;
			Bnx = (Ax + Cx) * 0.5f, Bny = (Ay + Cy) * 0.5f;
			Cnx = (Ax + Bx) * 0.5f, Cny = (Ay + By) * 0.5f;
			break;
		case 1:
			Anx = Ax, Any = Ay;
			Bnx = (Ax + Bx) * 0.5f, Bny = (Ay + By) * 0.5f;
			Cnx = (Ax + Cx) * 0.5f, Cny = (Ay + Cy) * 0.5f;
			break;
		case 2:
			Anx = (Bx + Ax) * 0.5f, Any = (By + Ay) * 0.5f;
			Bnx = Bx, Bny = By;
			Cnx = (Bx + Cx) * 0.5f, Cny = (By + Cy) * 0.5f;
			break;
		default:
			Anx = (Cx + Ax) * 0.5f, Any = (Cy + Ay) * 0.5f;
			Bnx = (Cx + Bx) * 0.5f, Bny = (Cy + By) * 0.5f;
			Cnx = Cx, Cny = Cy;
			break;
		}
		Ax = Anx, Ay = Any, Bx = Bnx, By = Bny, Cx = Cnx, Cy = Cny;
	}
	const float rx = (Ax + Bx + Cx) * 0.3333333f, ry = (Ay + By + Cy) * 0.3333333f;
	return V3(rx, ry, 1.0f - rx - ry);
}
/* lights.h:159-265 (IS_LIGHTS 1).  The reference keeps the potentials in a MAX_IS_LIGHTS array; they are
 * recomputed in a second pass here so any number of lights is defined behaviour. */
static v3 random_point_on_light(const rfwo_context *c, float r0, float r1, v3 I, v3 N, float *pickProb,
								float *lightPdf, v3 *lightColor)
{
	const uint32_t lights = total_lights(c);
	const v3 bary = random_barycentrics(r0);
	float sum = 0;
	for (uint32_t k = 0; k < lights; k++)
		sum += pot_any(c, k, I, N, bary);
	if (sum <= 0)
	{
		*lightPdf = 0;
		return V3(1, 1, 1);
	}
	r1 *= sum;
	float total = 0, chosen = 0;
	int lightIdx = 0;
	float first = 0;
	for (uint32_t k = 0; k < lights; k++)
	{
		const float p = pot_any(c, k, I, N, bary);
		if (k == 0)
			first = p;
		total += p;
		if (total >= r1)
		{
			lightIdx = (int)k, chosen = p;
			break;
		}
		if (k == lights - 1)
			lightIdx = 0, chosen = first; /* loop fell through: lightIdx stays 0 in the reference */
	}
	*pickProb = chosen / sum;
	uint32_t li = (uint32_t)lightIdx;
	if (li < c->lc.areaLightCount)
	{
		const rfwhip_area_light *l = &c->area[li];
		*lightColor = v3p(l->radiance);
		const v3 LN = v3p(l->normal);
		const v3 P = vadd(vadd(vscale(v3p(l->vertex0), bary.x), vscale(v3p(l->vertex1), bary.y)),
						  vscale(v3p(l->vertex2), bary.z));
		v3 L = vsub(I, P);
		const float sqDist = vdot(L, L);
		L = vnorm(L);
		const float LNdotL = vdot(L, LN);
		const float reciSolidAngle = sqDist / (l->area * LNdotL);
		const float energy = vlen(v3p(l->radiance)); /* DeviceAreaLight::getEnergy, device_structs.h:115 */
		*lightPdf = (LNdotL > 0 && vdot(L, N) < 0) ? (reciSolidAngle * (1.0f / energy)) : 0;
		return P;
	}
	li -= c->lc.areaLightCount;
	if (li < c->lc.pointLightCount)
	{
		const rfwhip_point_light *l = &c->point[li];
		const v3 pos = v3p(l->position);
		*lightColor = v3p(l->radiance);
		const v3 L = vsub(I, pos);
		const float sqDist = vdot(L, L);
		*lightPdf = vdot(L, N) < 0 ? (sqDist / l->energy) : 0;
		return pos;
	}
	li -= c->lc.pointLightCount;
	if (li < c->lc.spotLightCount)
	{
		const rfwhip_spot_light *l = &c->spot[li];
		const v3 P = v3p(l->position);
		v3 L = vsub(I, P);
		const float sqDist = vdot(L, L);
		L = vnorm(L);
		const float d = fmaxf(0.0f, vdot(L, v3p(l->direction)) - l->cosOuter) / (l->cosInner - l->cosOuter);
		const float LNdotL = fminf(1.0f, d);
		*lightPdf = (LNdotL > 0 && vdot(L, N) < 0) ? (sqDist / (LNdotL * l->energy)) : 0;
		*lightColor = v3p(l->radiance);
		return P;
	}
	li -= c->lc.spotLightCount;
	const rfwhip_directional_light *l = &c->dir[li];
	const v3 L = v3p(l->direction);
	*lightColor = v3p(l->radiance);
	const float NdotL = vdot(L, N);
	*lightPdf = NdotL < 0 ? (1.0f / l->energy) : 0;
	return vsub(I, vscale(L, 1000.0f));
}

/* getShadingData.h:25-60 — bilinear fetch from a flat texel array, UINT (RGBA8, x1/256) or FLOAT4 */
static void fetch_texel(const otexture *tex, float tu, float tv, size_t o, int w, int h, float out[4])
{
	const float tcx = (fmaxf(tu + 1000, 0.0f) * w) - 0.5f, tcy = (fmaxf(tv + 1000, 0.0f) * h) - 0.5f;
	const int iu = (int)tcx % w, iv = (int)tcy % h;
	const float fu = tcx - floorf(tcx), fv = tcy - floorf(tcy);
	const float w0 = (1 - fu) * (1 - fv), w1 = fu * (1 - fv), w2 = (1 - fu) * fv, w3 = 1 - (w0 + w1 + w2);
	const int iu1 = (iu + 1) % w, iv1 = (iv + 1) % h;
	const size_t id[4] = {o + iu + (size_t)iv * w, o + iu1 + (size_t)iv * w, o + iu + (size_t)iv1 * w,
						  o + iu1 + (size_t)iv1 * w};
	const float wt[4] = {w0, w1, w2, w3};
	out[0] = out[1] = out[2] = out[3] = 0;
	for (int k = 0; k < 4; k++)
	{
		float p[4];
		size_t i = id[k];
		if (i >= tex->texelCount)
			i = tex->texelCount - 1;
		if (tex->type == RFWHIP_TEX_UINT)
		{
			const uint32_t t = ((const uint32_t *)tex->data)[i];
			const float r = 1.0f / 256.0f;
			p[0] = (float)(t & 255u) * r, p[1] = (float)((t >> 8) & 255u) * r, p[2] = (float)((t >> 16) & 255u) * r;
			p[3] = (float)(t >> 24) * r;
		}
		else
			memcpy(p, (const float *)tex->data + 4 * i, 16);
		for (int q = 0; q < 4; q++)
			out[q] += p[q] * wt[k];
	}
}
/* getShadingData.h:61-98 — MIPLEVELCOUNT 5; a texture that carries fewer texels than the chain needs is sampled at
 * level 0 only (texelCount tells). */
static void fetch_trilinear(const otexture *tex, float lambda, float tu, float tv, int width, int height,
							float out[4])
{
	size_t chain = 0;
	{
		int w = width, h = height;
		for (int i = 0; i < 5; i++)
			chain += (size_t)w * h, w >>= 1, h >>= 1;
	}
	const int has_mips = tex->texelCount >= chain;
	/* getShadingData.h:66-67: level0 = min(4, (int)lambda), level1 = min(4, level0 + 1) — NOT clamped at 0: for
	 * lambda <= -1 both loops below run zero times and both levels are the base level */
	int level0 = (int)lambda;
	if (level0 > 4)
		level0 = 4;
	int level1 = level0 + 1 > 4 ? 4 : level0 + 1;
	if (!has_mips)
		level0 = 0;
	if (!has_mips)
		level1 = 0;
	const float f = lambda - floorf(lambda);
	size_t o0 = 0, o1 = 0;
	int w0 = width, h0 = height, w1 = width, h1 = height;
	for (int i = 0; i < level0; i++)
		o0 += (size_t)w0 * h0, w0 >>= 1, h0 >>= 1;
	for (int i = 0; i < level1; i++)
		o1 += (size_t)w1 * h1, w1 >>= 1, h1 >>= 1;
	float p0[4], p1[4];
	fetch_texel(tex, tu, tv, o0, w0 > 0 ? w0 : 1, h0 > 0 ? h0 : 1, p0);
	fetch_texel(tex, tu, tv, o1, w1 > 0 ? w1 : 1, h1 > 0 ? h1 : 1, p1);
	for (int q = 0; q < 4; q++)
		out[q] = (1.0f - f) * p0[q] + f * p1[q];
}

static void layer_trilinear(const rfwo_context *c, const rfwhip_map_desc *md, float lambda, float tu, float tv,
							float out[4])
{
	fetch_trilinear(&c->textures[md->addr], lambda, half_to_float(md->uscale) * (half_to_float(md->uoffs) + tu),
					half_to_float(md->vscale) * (half_to_float(md->voffs) + tv), md->width, md->height, out);
}
static v3 layer_normal(const rfwo_context *c, const rfwhip_map_desc *md, float tu, float tv)
{
	float p[4];
	fetch_texel(&c->textures[md->addr], half_to_float(md->uscale) * (half_to_float(md->uoffs) + tu),
				half_to_float(md->vscale) * (half_to_float(md->voffs) + tv), 0, md->width > 0 ? md->width : 1,
				md->height > 0 ? md->height : 1, p);
	return vscale(vsub(V3(p[0], p[1], p[2]), V3(0.5f, 0.5f, 0.5f)), 2.0f);
}

#define PT_MAX_DEPTHS 16
typedef struct
{
	v3 sum;					   /* sky + emissive terms: added unconditionally */
	v3 pend[PT_MAX_DEPTHS];	   /* unoccluded connection emitted by the shade call of depth d (added if that wave is launched) */
	uint32_t ext_d[PT_MAX_DEPTHS + 1]; /* rays of this path in the extension wave of depth d */
	uint32_t shadow_d[PT_MAX_DEPTHS];  /* connections emitted at depth d */
	tstat st, ss_d[PT_MAX_DEPTHS];
	int probe_hit, probe_inst, probe_prim;
	float probe_t;
	float pt, pu, pv;
	int pprim, pinst;
} ptresult;

static void pt_path(rfwo_context *c, const rfwhip_camera_view *view, float clampValue, uint32_t pixel,
					uint32_t sampleIdx, ptresult *res)
{
	const uint32_t W = c->W, H = c->H;
	/* ---- generatePrimaryRay, Kernels.cu:383-426, RNG branch (BLUENOISE off) ---- */
	uint32_t seed = wang_hash(pixel * 16789u + sampleIdx * 1791u);
	const int sx = (int)(pixel % W), sy = (int)(pixel / W);
	float r0, r1, r2, r3;
	if (c->sampler == 1 && c->blue_noise) /* Kernels.cu:391-394 */
	{
		r0 = rfwo_blue_noise_sample(c->blue_noise, sx, sy, (int)sampleIdx, 0);
		r1 = rfwo_blue_noise_sample(c->blue_noise, sx, sy, (int)sampleIdx, 1);
		r2 = rfwo_blue_noise_sample(c->blue_noise, sx, sy, (int)sampleIdx, 2);
		r3 = rfwo_blue_noise_sample(c->blue_noise, sx, sy, (int)sampleIdx, 3);
	}
	else
	{
		r0 = random_float(&seed), r1 = random_float(&seed);
		r2 = random_float(&seed), r3 = random_float(&seed);
	}
	const float blade = (float)(int)(r0 * 9);
	r2 = (r2 - blade * (1.0f / 9.0f)) * 9.0f;
	const float piOver4point5 = 3.14159265359f / 4.5f;
	/* __sincosf(x, &x1, &y1): x1 = sin, y1 = cos */
	const float x1 = sinf(blade * piOver4point5), y1 = cosf(blade * piOver4point5);
	const float x2 = sinf((blade + 1.0f) * piOver4point5), y2 = cosf((blade + 1.0f) * piOver4point5);
	if ((r2 + r3) > 1.0f)
		r2 = 1.0f - r2, r3 = 1.0f - r3;
	/* (fixed-shape arithmetic from here on: rfw_oracle_math.h, rounded()) */
	const float xr = fmaf(x2, r3, rounded(x1 * r2)), yr = fmaf(y2, r3, rounded(y1 * r2));
	const v3 right = vsub(v3p(view->p2), v3p(view->p1)), up = vsub(v3p(view->p3), v3p(view->p1));
	v3 O = v3p(view->pos);
	if (view->aperture != 0.0f) /* (with aperture 0 the offset is exactly zero) */
		O = vmadd2_r(v3p(view->pos), right, rounded(xr * view->aperture), up, rounded(yr * view->aperture));
	const float uu = rounded(((float)sx + r0) * (1.0f / (float)W)), vv = rounded(((float)sy + r1) * (1.0f / (float)H));
	v3 D = vnorm_r(vsub(vmadd2_r(v3p(view->p1), right, uu, up, vv), O));
	if (g_ref_arith) /* Kernels.cu:413-424 as written: plain products and sums */
	{
		const float xr2 = x1 * r2 + x2 * r3, yr2 = y1 * r2 + y2 * r3;
		O = vadd(v3p(view->pos), vscale(vadd(vscale(right, xr2), vscale(up, yr2)), view->aperture));
		const float u2 = ((float)sx + r0) * (1.0f / (float)W), v2 = ((float)sy + r1) * (1.0f / (float)H);
		D = vnorm(vsub(vadd(vadd(v3p(view->p1), vscale(right, u2)), vscale(up, v2)), O));
	}

	v3 T = V3(1, 1, 1);
	float bsdfPdf = 1.0f;
	uint32_t flags = 1; /* IS_SPECULAR */
	uint32_t packedN = 0;
	const uint32_t nlights = total_lights(c);

	for (uint32_t pathLength = 0;; pathLength++)
	{
		float t = 1e34f, bu = 0, bv = 0;
		int inst = -1, prim = -1;
		res->ext_d[pathLength]++;
		const int hit = scene_closest(c, O, D, 1e-5f, &t, &inst, &prim, &bu, &bv, &res->st);
		if (pathLength == 0)
			res->pt = t, res->pu = bu, res->pv = bv, res->pprim = hit ? prim : -1, res->pinst = hit ? inst : -1;
		if (!hit)
		{
			v3 contribution = vmul(vscale(T, 1.0f / bsdfPdf), pt_sky(c, D));
			if (v3_any_nan(contribution))
				return;
			contribution = clamp_intensity(contribution, clampValue);
			res->sum = vadd(res->sum, contribution);
			return;
		}
		const v3 I = vadd(O, vscale(D, t));
		const oinstance *in = &c->instances[inst];
		const rfwhip_triangle *tri = &c->meshes[in->mesh].tris[prim];
		const rfwhip_material *mat = &c->materials[tri->material];
		/* ---- getShadingData (getShadingData.h:100-217); u,v,w there are the weights of v0,v1,v2 ---- */
		const float bw0 = 1.0f - bu - bv, bw1 = bu, bw2 = bv;
		oshading sd;
		sd.color = mat_color(mat);
		sd.absorption = V3(half_to_float(mat->transmittance[0]), half_to_float(mat->transmittance[1]),
						   half_to_float(mat->transmittance[2]));
		memcpy(sd.p, mat->parameters, 16);
		v3 N = V3(tri->Nx, tri->Ny, tri->Nz), iN = N;
		if (mat_flag(mat, RFWHIP_MAT_HAS_SMOOTH_NORMALS))
			iN = vnorm(vadd(vadd(vscale(v3p(tri->vN0), bw0), vscale(v3p(tri->vN1), bw1)), vscale(v3p(tri->vN2), bw2)));
		N = vnorm(m3_mul(in->normal, N));
		iN = vnorm(m3_mul(in->normal, iN));
		v3 Tg, Bt;
		create_tangent_space(iN, &Tg, &Bt);
		int alpha_skip = 0;
		if (mat_flag(mat, RFWHIP_MAT_HAS_DIFFUSE_MAP) && mat->map[0].addr < c->textureCount)
		{
			const float tu = bw0 * tri->u0 + bw1 * tri->u1 + bw2 * tri->u2;
			const float tv = bw0 * tri->v0 + bw1 * tri->v1 + bw2 * tri->v2;
			const float coneWidth = view->spreadAngle * t;
			const float lambda = tri->LOD + log2f(coneWidth * (1.0f / fabsf(vdot(vscale(D, -1.0f), N))));
			float texel[4];
			layer_trilinear(c, &mat->map[0], lambda, tu, tv, texel);
			if (mat_flag(mat, RFWHIP_MAT_HAS_ALPHA) && texel[3] < 0.5f)
				alpha_skip = 1; /* getShadingData.h:145-149 */
			else
			{
				sd.color = vmul(sd.color, V3(texel[0], texel[1], texel[2]));
				/* additive second and third layers (getShadingData.h:153-166) */
				if (mat_flag(mat, RFWHIP_MAT_HAS_2ND_DIFFUSE_MAP) && mat->map[1].addr < c->textureCount)
				{
					float l1[4];
					layer_trilinear(c, &mat->map[1], lambda, tu, tv, l1);
					sd.color = vadd(sd.color, V3(l1[0], l1[1], l1[2]));
				}
				if (mat_flag(mat, RFWHIP_MAT_HAS_3RD_DIFFUSE_MAP) && mat->map[2].addr < c->textureCount)
				{
					float l2[4];
					layer_trilinear(c, &mat->map[2], lambda, tu, tv, l2);
					sd.color = vadd(sd.color, V3(l2[0], l2[1], l2[2]));
				}
				/* normal maps at level 0 (getShadingData.h:169-200); layer 3 reads layer 2's descriptor (:189-196) */
				if (mat_flag(mat, RFWHIP_MAT_HAS_NORMAL_MAP) && mat->map[3].addr < c->textureCount)
				{
					v3 sn = layer_normal(c, &mat->map[3], tu, tv);
					if (mat_flag(mat, RFWHIP_MAT_HAS_2ND_NORMAL_MAP) && mat->map[4].addr < c->textureCount)
						sn = vadd(sn, layer_normal(c, &mat->map[4], tu, tv));
					if (mat_flag(mat, RFWHIP_MAT_HAS_3RD_NORMAL_MAP) && mat->map[4].addr < c->textureCount)
						sn = vadd(sn, layer_normal(c, &mat->map[4], tu, tv));
					sn = vnorm(sn);
					/* tangentToWorld (tools.h:214) with the frame of the unperturbed normal */
					iN = vnorm(vadd(vadd(vscale(Tg, sn.x), vscale(Bt, sn.y)), vscale(iN, sn.z)));
				}
				/* getShadingData.h:150 and :206 both multiply the colour by the texel */
				sd.color = vmul(sd.color, V3(texel[0], texel[1], texel[2]));
			}
		}
		if (pathLength == 0 && pixel == c->probe_y * W + c->probe_x)
			res->probe_hit = 1, res->probe_inst = inst, res->probe_prim = prim, res->probe_t = t;

		/* ---- alpha pass-through: Kernels.cu:633-647 (the path continues behind the surface, state untouched) ---- */
		if (alpha_skip)
		{
			if (pathLength >= (uint32_t)c->max_depth || v3_any_nan(T))
				return;
			O = vadd(I, vscale(D, GEO_EPS));
			continue;
		}
		/* ---- emissive: Kernels.cu:650-692 ---- */
		if (sd.color.x > 1.0f || sd.color.y > 1.0f || sd.color.z > 1.0f)
		{
			const float DdotNL = -vdot(D, N);
			v3 contribution = V3(0, 0, 0);
			if (DdotNL > 0)
			{
				if (pathLength == 0)
					contribution = sd.color;
				else if (flags & 1u)
					contribution = vscale(vmul(T, sd.color), 1.0f / bsdfPdf);
				else
				{
					const v3 lastN = unpack_normal(packedN);
					const float lightPdf = (t * t) / (-vdot(D, N) * tri->area); /* lights.h:78-81 */
					/* the reference reads the material id here (device_structs.h:37,40); the light index meant
					 * is Triangle::lightTriIdx */
					const float pickProb = tri->lightTriIdx >= 0 && (uint32_t)tri->lightTriIdx < c->lc.areaLightCount
											   ? light_pick_prob(c, tri->lightTriIdx, O, lastN, I)
											   : 0.0f;
					if ((bsdfPdf + lightPdf * pickProb) <= 0)
						return;
					contribution = vscale(vmul(T, sd.color), 1.0f / (bsdfPdf + lightPdf * pickProb));
				}
			}
			if (v3_any_nan(contribution))
				contribution = V3(0, 0, 0);
			contribution = clamp_intensity(contribution, clampValue);
			res->sum = vadd(res->sum, contribution);
			return;
		}
		if (SD_ROUGHNESS(&sd) < 0.01f)
			flags |= 1u;
		else
			flags &= ~1u;
		seed = wang_hash(pixel * 16789u + sampleIdx * 1791u + pathLength * 720898027u);
		const float flip = (vdot(D, N) > 0) ? -1.0f : 1.0f;
		N = vscale(N, flip);
		iN = vscale(iN, flip);
		T = vscale(T, 1.0f / bsdfPdf);

		/* ---- next-event estimation: Kernels.cu:702-755 ---- */
		if ((flags & 1u) == 0 && nlights > 0)
		{
			v3 lightColor = V3(0, 0, 0);
			float pickProb = 0, lightPdf = 0;
			float q0, q1;
			if (c->sampler == 1 && c->blue_noise && sampleIdx < 256) /* BLUENOISE, Kernels.cu:712-719: seed not advanced */
			{
				q0 = rfwo_blue_noise_sample(c->blue_noise, sx, sy, (int)sampleIdx, 4);
				q1 = rfwo_blue_noise_sample(c->blue_noise, sx, sy, (int)sampleIdx, 5);
			}
			else
				q0 = random_float(&seed), q1 = random_float(&seed);
			v3 L = vsub(random_point_on_light(c, q0, q1, I, iN, &pickProb, &lightPdf, &lightColor), I);
			const float dist = vlen(L);
			L = vscale(L, 1.0f / dist);
			const float NdotL = vdot(L, iN);
			if (NdotL > 0 && lightPdf > 0)
			{
				const v3 wo = vscale(D, -1.0f);
				const v3 bs = bsdf_eval(&sd, iN, wo, L, 0.0f, 0);
				const float shadowPdf = bsdf_pdf(&sd, iN, wo, L);
				if (shadowPdf > 0)
				{
					v3 contribution =
						vscale(vmul(vmul(T, bs), lightColor), NdotL / (shadowPdf + lightPdf * pickProb));
					contribution = clamp_intensity(contribution, clampValue);
					/* The connections of a shade call are traced at the top of the NEXT iteration of the host loop
					 * (CUDART/src/Context.cpp:109-120): those of the last shade call (pathLength == MAX_PATH_LENGTH)
					 * never are.  Whether the wave of an earlier depth is launched (activePaths > 0) is known only
					 * after the whole frame: render_pt_sample decides. */
					if (!v3_any_nan(contribution) && pathLength < (uint32_t)c->max_depth)
					{
						res->shadow_d[pathLength]++;
						const v3 so = vadd(I, vscale(N, 1e-5f)); /* SafeOrigin, tools.h:119-123 */
						if (!scene_occluded(c, so, L, GEO_EPS, dist - 2.0f * GEO_EPS, &res->ss_d[pathLength]))
							res->pend[pathLength] = contribution;
					}
				}
			}
		}
		if (pathLength >= (uint32_t)c->max_depth)
			return;
		/* ---- SampleBSDF: r3 then r4 from the same seed (disney.h:274-280) ---- */
		v3 R = V3(0, 0, 1);
		float newPdf = 0.0f;
		const float q3 = random_float(&seed), q4 = random_float(&seed);
		const v3 wo = vscale(D, -1.0f);
		bsdf_sample(&sd, Tg, Bt, iN, wo, &R, &newPdf, q3, q4);
		const v3 bs = bsdf_eval(&sd, iN, wo, R, t, flip < 0);
		{
			const float surv = survival_probability(T);
			T = vscale(vmul(V3(T.x / surv, T.y / surv, T.z / surv), bs), fabsf(vdot(iN, R)));
		}
		if (newPdf < 1e-6f || isnan(newPdf) || T.x < 0.0f || T.y < 0.0f || T.z < 0.0f)
			return;
		O = vadd(I, vscale(N, 1e-5f));
		D = R;
		packedN = pack_normal(iN);
		bsdfPdf = newPdf;
	}
}

/* Known-answer hook (rfw_oracle.h): one of the path tracer's functions on n records, same record layout as the
 * product's rfwhip_kat (include/rfwhip.h). */
int rfwo_kat(rfwo_context *c, int function, size_t n, const float *in, float *out)
{
	if (!c || !in || !out)
		return fail("rfwo_kat: null argument");
	for (size_t i = 0; i < n; i++)
	{
		const float *r = in + i * RFWHIP_KAT_IN;
		float *o = out + i * RFWHIP_KAT_OUT;
		uint32_t ub[RFWHIP_KAT_IN];
		memcpy(ub, r, sizeof(ub));
		for (int k = 0; k < RFWHIP_KAT_OUT; k++)
			o[k] = 0.0f;
		oshading sd;
		sd.color = V3(r[0], r[1], r[2]), sd.absorption = V3(r[3], r[4], r[5]);
		sd.p[0] = ub[6], sd.p[1] = ub[7], sd.p[2] = ub[8], sd.p[3] = 0;
		const v3 N = V3(r[9], r[10], r[11]), wo = V3(r[12], r[13], r[14]), wi = V3(r[15], r[16], r[17]);
		switch (function)
		{
		case RFWHIP_KAT_BSDF_EVAL:
		{
			const v3 e = bsdf_eval(&sd, N, wo, wi, r[18], ub[19] != 0);
			o[0] = e.x, o[1] = e.y, o[2] = e.z;
			break;
		}
		case RFWHIP_KAT_BSDF_PDF:
			o[0] = bsdf_pdf(&sd, N, wo, wi);
			break;
		case RFWHIP_KAT_BSDF_SAMPLE:
		{
			v3 T, B, R = V3(0, 0, 1);
			float pdf = 0.0f;
			create_tangent_space(N, &T, &B);
			bsdf_sample(&sd, T, B, N, wo, &R, &pdf, r[20], r[21]);
			o[0] = R.x, o[1] = R.y, o[2] = R.z, o[3] = pdf;
			break;
		}
		case RFWHIP_KAT_TANGENT_SPACE:
		{
			v3 T, B;
			create_tangent_space(N, &T, &B);
			o[0] = T.x, o[1] = T.y, o[2] = T.z, o[3] = B.x, o[4] = B.y, o[5] = B.z;
			break;
		}
		case RFWHIP_KAT_PACK_NORMAL:
		{
			const uint32_t pk = pack_normal(N);
			const v3 u = unpack_normal(pk);
			memcpy(&o[0], &pk, 4);
			o[1] = u.x, o[2] = u.y, o[3] = u.z;
			break;
		}
		case RFWHIP_KAT_RANDOM_BARYCENTRICS:
		{
			const v3 b = random_barycentrics(r[20]);
			o[0] = b.x, o[1] = b.y, o[2] = b.z;
			break;
		}
		case RFWHIP_KAT_POINT_ON_LIGHT:
		{
			float pick = 0, pdf = 0;
			v3 col = V3(0, 0, 0);
			const v3 P = random_point_on_light(c, r[6], r[7], V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), &pick, &pdf, &col);
			o[0] = P.x, o[1] = P.y, o[2] = P.z, o[3] = pick, o[4] = pdf, o[5] = col.x, o[6] = col.y, o[7] = col.z;
			break;
		}
		case RFWHIP_KAT_LIGHT_PICK_PROB:
			o[0] = light_pick_prob(c, (int)ub[8], V3(r[9], r[10], r[11]), V3(r[3], r[4], r[5]), V3(r[0], r[1], r[2]));
			break;
		case RFWHIP_KAT_BLUE_NOISE:
			if (!c->blue_noise)
				return fail("rfwo_kat: no blue-noise table");
			o[0] = rfwo_blue_noise_sample(c->blue_noise, (int)ub[0], (int)ub[1], (int)ub[2], (int)ub[3]);
			break;
		case RFWHIP_KAT_HASH:
		{
			uint32_t st = wang_hash(ub[0]);
			memcpy(&o[0], &st, 4);
			o[1] = random_float(&st);
			memcpy(&o[2], &st, 4);
			break;
		}
		default:
			return fail("rfwo_kat: unknown function");
		}
	}
	return 0;
}

static void render_pt_sample(rfwo_context *c, const rfwhip_camera_view *view, float clampValue, uint32_t sampleIdx)
{
	const uint32_t W = c->W, H = c->H;
	const int nd = c->max_depth > 0 ? c->max_depth : 1; /* depths whose connections can be traced: 0 .. max_depth-1 */
	const size_t need = (size_t)W * H * (size_t)nd * 3;
	if (need > c->pend_cap)
	{
		free(c->pend);
		c->pend = (float *)malloc(need * sizeof(float));
		c->pend_cap = need;
	}
	uint64_t ext_d[PT_MAX_DEPTHS + 1], shadow_d[PT_MAX_DEPTHS], sinner_d[PT_MAX_DEPTHS], stris_d[PT_MAX_DEPTHS];
	memset(ext_d, 0, sizeof(ext_d)), memset(shadow_d, 0, sizeof(shadow_d));
	memset(sinner_d, 0, sizeof(sinner_d)), memset(stris_d, 0, sizeof(stris_d));
	uint64_t tinner = 0, ttris = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tinner, ttris, ext_d[:PT_MAX_DEPTHS + 1], shadow_d[:PT_MAX_DEPTHS], sinner_d[:PT_MAX_DEPTHS], stris_d[:PT_MAX_DEPTHS])
	for (int y = 0; y < (int)H; y++)
	{
		if (!owns_row(c, (uint32_t)y))
			continue;
		for (uint32_t x = 0; x < W; x++)
		{
			const uint32_t pixel = (uint32_t)y * W + x;
			ptresult r;
			memset(&r, 0, sizeof(r));
			pt_path(c, view, clampValue, pixel, sampleIdx, &r);
			float *a = &c->acc[(size_t)pixel * 4];
			a[0] += r.sum.x, a[1] += r.sum.y, a[2] += r.sum.z, a[3] += 1.0f;
			for (int d = 0; d < nd; d++)
			{
				float *pd = &c->pend[((size_t)pixel * nd + d) * 3];
				pd[0] = r.pend[d].x, pd[1] = r.pend[d].y, pd[2] = r.pend[d].z;
			}
			c->hit_t[pixel] = r.pt, c->hit_u[pixel] = r.pu, c->hit_v[pixel] = r.pv;
			c->hit_prim[pixel] = r.pprim, c->hit_inst[pixel] = r.pinst;
			if (r.probe_hit && sampleIdx == 0)
				c->probe_inst = (uint32_t)r.probe_inst, c->probe_prim = (uint32_t)r.probe_prim, c->probe_dist = r.probe_t;
			tinner += r.st.inner, ttris += r.st.tris;
			for (int d = 0; d < PT_MAX_DEPTHS; d++)
				ext_d[d] += r.ext_d[d], shadow_d[d] += r.shadow_d[d], sinner_d[d] += r.ss_d[d].inner, stris_d[d] += r.ss_d[d].tris;
			ext_d[PT_MAX_DEPTHS] += r.ext_d[PT_MAX_DEPTHS];
		}
	}
	/* CUDART/src/Context.cpp:109: while (activePaths > 0 && pathLength < MAX_PATH_LENGTH) { trace the connections of the
	 * previous shade call; extend; shade }.  The connection wave of depth d runs iff depth d + 1 has extension rays. */
	uint64_t ext = 0, shadow = 0, sinner = 0, stris = 0;
	for (int d = 0; d <= PT_MAX_DEPTHS; d++)
		ext += ext_d[d];
	for (int d = 0; d < c->max_depth && d < PT_MAX_DEPTHS; d++)
	{
		if (ext_d[d + 1] == 0)
			continue;
		shadow += shadow_d[d], sinner += sinner_d[d], stris += stris_d[d];
#pragma omp parallel for schedule(static)
		for (int y = 0; y < (int)H; y++)
		{
			if (!owns_row(c, (uint32_t)y))
				continue;
			for (uint32_t x = 0; x < W; x++)
			{
				const size_t pixel = (size_t)y * W + x;
				const float *pd = &c->pend[(pixel * nd + d) * 3];
				float *a = &c->acc[pixel * 4];
				a[0] += pd[0], a[1] += pd[1], a[2] += pd[2];
			}
		}
	}
	c->cnt[0] += ext, c->cnt[1] += shadow, c->cnt[2] += tinner, c->cnt[3] += ttris, c->cnt[4] += sinner;
	c->cnt[5] += stris;
	c->stats.shadowCount += (uint32_t)shadow;
	c->stats.secondaryCount += (uint32_t)ext_d[1];
	for (int d = 2; d <= PT_MAX_DEPTHS; d++)
		c->stats.deepCount += (uint32_t)ext_d[d];
}

/* =============================================================================================================
 * render
 * ========================================================================================================== */
int rfwo_render(rfwo_context *c, const rfwhip_camera *camera, int status)
{
	if (!c || !camera || !c->acc)
		return fail("rfwo_render: context not initialised");
#ifdef _OPENMP
	if (c->threads > 0)
		omp_set_num_threads(c->threads);
#endif
	const double t0 = now_ms();
	if (status == RFWHIP_RESET)
	{
		memset(c->acc, 0, (size_t)c->W * c->H * 16);
		c->samples = 0;
	}
	memset(&c->stats, 0, sizeof(c->stats));
	rfwhip_camera_view view;
	rfwo_camera_get_view(camera, &view);
	camparams cp;
	cp.pos = v3p(view.pos), cp.p1 = v3p(view.p1);
	cp.right = vsub(v3p(view.p2), v3p(view.p1)), cp.up = vsub(v3p(view.p3), v3p(view.p1));
	cp.aperture = view.aperture;
	for (int s = 0; s < c->spp; s++)
	{
		if (c->integrator == 0)
		{
			render_parity_sample(c, &cp);
			c->cnt[0] += (uint64_t)(c->W / 4) * 4 * (c->H / 2) * 2;
		}
		else
			render_pt_sample(c, &view, camera->clampValue, c->samples);
		c->samples++;
		c->cnt[7] += (uint64_t)c->W * c->H;
	}
	c->stats.primaryCount = c->W * c->H * (uint32_t)c->spp;
	c->stats.primaryTime = (float)(now_ms() - t0);
	c->stats.renderTime = c->stats.primaryTime;
	return 0;
}
