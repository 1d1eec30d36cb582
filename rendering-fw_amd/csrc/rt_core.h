// rt_core.h — the per-ray / per-path arithmetic of the HIP rendercore: primary-ray generation, two-level BVH2
// traversal with Möller–Trumbore, the parity integrator's shade loop and the wavefront path tracer's shade step
// (next-event estimation + Disney BSDF).  Pure functions over SceneView/WaveView pointers; the kernels in
// kernels.hip add the thread mapping, the LDS traversal stack and the wave-level compaction around them.
//
// Behavioural spec (reference file:line under /root/reference):
//   primary rays (parity)     RFW/backends/EmbreeRT/src/Ray.cpp:16-47, :176-384
//   primary rays (pt)         RFW/backends/CUDART/src/Kernels.cu:383-426
//   slab test                 RFW/system/bvh/src/aabb.cpp:39-77
//   BVH2 traversal            RFW/system/bvh/include/bvh/bvh_node.h:317-448, CUDART/src/CUDAIntersect.h:199-268
//   Möller–Trumbore           RFW/system/bvh/src/bvh_tree.cpp:166-196, CUDAIntersect.h:48-94
//   instancing                RFW/system/bvh/src/top_level_bvh.cpp:104-191, Kernels.cu:267-301
//   parity shade              RFW/backends/EmbreeRT/src/Context.cpp:179-282, :417-476
//   pt shade                  RFW/backends/CUDART/src/Kernels.cu:571-794, lights.h, getShadingData.h
//   BSDF                      RFW/system/context/rfw/bsdf/disney.h, tools.h, compat.h:47-74
#pragma once
#include "rt_types.h"
#include <math.h>

namespace rt
{

// RT_DIAG_SHADE_CLOCK (development builds, tools/dev/variant.sh): where a wave of the shade kernel spends its cycles.  RT_TICK(k)
// adds the s_memtime ticks since the previous tick to phase k of the wave's table in LDS, which the wave adds to a global one when
// it leaves (a load's latency lands in the phase that first uses its result).
#if defined(RT_DIAG_SHADE_CLOCK)
struct ClkProbe
{
	unsigned long long last;
	unsigned long long *acc;
};
RT_FN void clk_tick(ClkProbe &c, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned long long t = __builtin_readcyclecounter();
	const unsigned long long m = __ballot(1);
	if (m && (int)__lane_id() == __ffsll((long long)m) - 1)
		c.acc[k] += t - c.last, c.acc[16 + k] += 1ull; // (the wave's own table in LDS; flushed once when the kernel ends)
	c.last = __builtin_readcyclecounter();
#endif
}
#define RT_TICK(K) do { if (clk) clk_tick(*clk, K); } while (0)
#define RT_CLK_PARAM , ClkProbe *clk = nullptr
#define RT_CLK_ARG , clk
#else
#define RT_TICK(K)
#define RT_CLK_PARAM
#define RT_CLK_ARG
#endif

// ---------------------------------------------------------------------------------------------------------------
// division and square root of the SHADING arithmetic (BSDF, light sampling, normals: everything behind the hit record)
// ---------------------------------------------------------------------------------------------------------------
// The shipped device build takes the hardware's 1-ulp v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 there: a correctly rounded fp32
// division is 10 instructions around the same v_rcp_f32 (two v_div_scale, four fmas, v_div_fmas, v_div_fixup — ~46 issue
// cycles against 18), a correctly rounded square root a dozen around v_sqrt_f32, and the plain shade kernel held 106 + 50 of
// them.  The host build and the validation build (RT_STRICT_MATH) keep IEEE division and square root, like the oracle; ray
// generation, traversal and the triangle test are not written with these (rounded(), safe_rcp).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_STRICT_MATH)
RT_FN float m_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_FN float m_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
RT_FN float m_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_FN float m_rsqrtf(float x) { return __builtin_amdgcn_rsqf(x); }
#else
RT_FN float m_rcp(float x) { return 1.0f / x; }
RT_FN float m_div(float a, float b) { return a / b; }
RT_FN float m_sqrtf(float x) { return sqrtf(x); }
RT_FN float m_rsqrtf(float x) { return 1.0f / sqrtf(x); }
#endif

// A record of a table whose index is the same in every lane of the wave (the loops over all lights): on the device it is read
// through the constant address space, from which the compiler selects scalar loads — one s_load per 16 / 32 / 64 bytes into
// SGPRs, waited for with lgkmcnt — instead of a global_load per lane with the same address and a wait for EVERY load the wave has
// in flight.  (The light loop of a shaded hit was ten dependent round trips of that kind: 46 % of the shade kernel's wave time,
// tools/dev: RT_DIAG_SHADE_CLOCK.)  A lane-varying index still works (the compiler falls back to vector loads).
template <typename T> RT_FN T uniform_record(const T *table, uint32_t idx)
{
#if defined(__HIP_DEVICE_COMPILE__)
	static_assert(sizeof(T) % 4 == 0, "records are whole dwords");
	const __attribute__((address_space(4))) uint32_t *src = (const __attribute__((address_space(4))) uint32_t *)(table + idx);
	T r;
	uint32_t *dst = (uint32_t *)&r;
#pragma unroll
	for (uint32_t i = 0; i < sizeof(T) / 4; i++)
		dst[i] = src[i];
	return r;
#else
	return table[idx];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// small vector algebra
// ---------------------------------------------------------------------------------------------------------------
RT_FN f3 mk3(float x, float y, float z)
{
	f3 r;
	r.x = x, r.y = y, r.z = z;
	return r;
}
RT_FN f4 mk4(float x, float y, float z, float w)
{
	f4 r;
	r.x = x, r.y = y, r.z = z, r.w = w;
	return r;
}
RT_FN f3 xyz(const f4 &a) { return mk3(a.x, a.y, a.z); }
RT_FN f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RT_FN f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_FN f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_FN f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_FN f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_FN float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_FN f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
RT_FN float length(f3 a) { return m_sqrtf(dot(a, a)); }
RT_FN f3 normalize(f3 a) { return a * m_rsqrtf(dot(a, a)); }
// a / s per component: one reciprocal in the shipped device build, three divisions where the arithmetic is IEEE
RT_FN f3 m_div3(f3 a, float s)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_STRICT_MATH)
	return a * m_rcp(s);
#else
	return mk3(a.x / s, a.y / s, a.z / s);
#endif
}
RT_FN f3 normalize_ieee(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); } // (the parity integrator, geometry updates)
RT_FN f3 lerp3(f3 a, f3 b, float t) { return a + (b - a) * t; }
RT_FN float lerp1(float a, float b, float t) { return a + t * (b - a); }
RT_FN float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
RT_FN bool any_nan(f3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }
RT_FN float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// Arithmetic with a FIXED shape.  A compiler is free to contract a * b + c into one fma wherever it sees the pattern, and sees
// it differently in every context a function is inlined into: the same ray against the same triangle then gives hit records
// that differ in the last bit between two kernels (the packet form of the primary wave and the per-lane form; the wave-wide
// triangle phase and a lane's own loop), between the device and the host emulation, and between those and the oracle.  The
// functions whose results travel between kernels — the triangle test, the primary ray, the instance transform — therefore say
// where they fuse: fmaf() where a fused multiply-add is meant, rounded() around every product that must be rounded on its own
// (an empty asm hides it from the contraction pass of hipcc, g++ and gcc alike; it emits nothing).  oracle/rfw_oracle.c states
// the same shapes, so that all three agree to the bit wherever the inputs do (the device's 1-ulp v_rcp_f32 aside).
RT_FN float rounded(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	asm("" : "+v"(x));
#elif defined(__x86_64__)
	asm("" : "+x"(x));
#else
	volatile float y = x;
	x = y;
#endif
	return x;
}
RT_FN float dot_r(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, rounded(a.x * b.x))); }
RT_FN f3 cross_r(f3 a, f3 b)
{
	return mk3(fmaf(a.y, b.z, -rounded(b.y * a.z)), fmaf(a.z, b.x, -rounded(b.z * a.x)), fmaf(a.x, b.y, -rounded(b.x * a.y)));
}
// p + a * s + b * t, each component two fused multiply-adds onto p
RT_FN f3 madd2_r(f3 p, f3 a, float s, f3 b, float t)
{
	return mk3(fmaf(b.x, t, fmaf(a.x, s, p.x)), fmaf(b.y, t, fmaf(a.y, s, p.y)), fmaf(b.z, t, fmaf(a.z, s, p.z)));
}
RT_FN f3 normalize_r(f3 a)
{
	const float inv = 1.0f / sqrtf(dot_r(a, a));
	return mk3(rounded(a.x * inv), rounded(a.y * inv), rounded(a.z * inv));
}
// one row of a 3x4 transform applied to a point (w = 1) / a direction (w = 0)
RT_FN float row_point_r(const float *r, f3 p) { return fmaf(r[2], p.z, fmaf(r[1], p.y, rounded(r[0] * p.x))) + r[3]; }
RT_FN float row_dir_r(const float *r, f3 d) { return fmaf(r[2], d.z, fmaf(r[1], d.y, rounded(r[0] * d.x))); }

RT_FN uint32_t f2u_sat(float f)
{
	if (!(f > 0.0f))
		return 0u;
	if (f >= 4294967296.0f)
		return 0xFFFFFFFFu;
	return (uint32_t)f;
}
RT_FN uint32_t fbits(float f)
{
	union
	{
		float f;
		uint32_t u;
	} c;
	c.f = f;
	return c.u;
}
RT_FN float ubits(uint32_t u)
{
	union
	{
		float f;
		uint32_t u;
	} c;
	c.u = u;
	return c.f;
}

// The transcendental functions the path tracer calls.  Shipped build: the platform's (device math library / v_sin_f32 and
// v_cos_f32 in sincos_turns / glibc on the host).  RT_STRICT_MATH (the validation build, rt_strict_math.h): one float
// implementation shared by device and host, so that the two agree to the bit.
} // namespace rt
#if defined(RT_STRICT_MATH)
#include "rt_strict_math.h"
#endif
namespace rt
{
#if defined(RT_STRICT_MATH)
RT_FN float m_sinf(float x) { return strict::sin_(x); }
RT_FN float m_cosf(float x) { return strict::cos_(x); }
RT_FN float m_expf(float x) { return strict::exp_(x); }
RT_FN float m_logf(float x) { return strict::log_(x); }
RT_FN float m_log2f(float x) { return strict::log2_(x); }
RT_FN float m_atan2f(float y, float x) { return strict::atan2_(y, x); }
RT_FN float m_acosf(float x) { return strict::acos_(x); }
#else
RT_FN float m_sinf(float x) { return sinf(x); }
RT_FN float m_cosf(float x) { return cosf(x); }
RT_FN float m_expf(float x) { return expf(x); }
RT_FN float m_logf(float x) { return logf(x); }
RT_FN float m_log2f(float x) { return log2f(x); }
RT_FN float m_atan2f(float y, float x) { return atan2f(y, x); }
RT_FN float m_acosf(float x) { return acosf(x); }
#endif

RT_FN float half_to_float(uint16_t h)
{
#if defined(__HIP_DEVICE_COMPILE__)
	// v_cvt_f32_f16 (exact, f16 denormals are enabled in the kernels' mode register).  The branchy form below put every
	// half field of a material into its own basic block: six dependent load-wait round trips per shaded hit.
	return (float)__builtin_bit_cast(_Float16, h);
#endif
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	const uint32_t exp = (h >> 10) & 0x1Fu;
	uint32_t man = h & 0x3FFu;
	uint32_t bits;
	if (exp == 0)
	{
		if (man == 0)
			bits = sign;
		else
		{
			int e = -1;
			do
			{
				man <<= 1;
				e++;
			} while (!(man & 0x400u));
			bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
		}
	}
	else if (exp == 31)
		bits = sign | 0x7F800000u | (man << 13);
	else
		bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
	return ubits(bits);
}

// ---------------------------------------------------------------------------------------------------------------
// RNGs: utils/xor128.h:20-27, utils/rng.h:14, bsdf/tools.h:218-235
// ---------------------------------------------------------------------------------------------------------------
RT_FN uint32_t xor128_next(uint32_t s[4])
{
	const uint32_t t = s[0] ^ (s[0] << 11);
	s[0] = s[1];
	s[1] = s[2];
	s[2] = s[3];
	s[3] = s[3] ^ (s[3] >> 19) ^ (t ^ (t >> 8));
	return s[3];
}
RT_FN float u32_to_unit(uint32_t v) { return (float)v * 2.3283064365387e-10f; }
RT_FN uint32_t wang_hash(uint32_t s)
{
	s = (s ^ 61u) ^ (s >> 16);
	s *= 9u;
	s = s ^ (s >> 4);
	s *= 0x27d4eb2du;
	s = s ^ (s >> 15);
	return s;
}
RT_FN float random_float(uint32_t &s)
{
	s ^= s << 13;
	s ^= s >> 17;
	s ^= s << 5;
	return (float)s * 2.3283064365387e-10f;
}

// ---------------------------------------------------------------------------------------------------------------
// pixel <-> path-slot mapping.  Within one sample, pixels run over 8x8 tiles, tiles row-major over the rank's compacted local
// image.  Samples of a batch are laid out in GROUPS of g = 2^sgroup_log2 (g <= 64, every sub-batch of a call is a multiple
// of g): one group of one tile is 64 g consecutive slots, pixel-major —
//     slot = sgroup * (g * slots) + tile * 64 g + pix * g + si,        sample = sgroup * g + si, pix = 0..63 in Z order (tile_pix),
// so a wave (64 consecutive slots) holds the g samples of 64 / g neighbouring pixels.  g = 1 is the plain layout (slot =
// s * slots + tile * 64 + pix: a wave = one 8x8 tile of one sample).  With g = 32 a wave is 2 neighbouring pixels x 32
// samples: its primary rays differ by sub-pixel jitter only, walk the same nodes and reach the same leaves in step — the
// two phases of the while-while traversal stay converged (the same holds for the shadow rays their first vertices emit).
// The layout only decides which path sits where; a pixel's samples are still summed in sample order (resolve_item), so the
// image does not depend on g.
// ---------------------------------------------------------------------------------------------------------------
struct PixelRef
{
	uint32_t x, y;		// global pixel
	uint32_t local;		// local row-major index (ylocal * W + x)
	uint32_t sample;	// sample index within the batch
	bool valid;
};
RT_FN uint32_t local_to_global_row(const FrameView &fr, uint32_t yl)
{
	return strip_of_local(yl / STRIP_ROWS, fr.rank, fr.world) * STRIP_ROWS + (yl % STRIP_ROWS);
}
RT_FN PixelRef slot_to_pixel(const FrameView &fr, uint32_t slot)
{
	PixelRef p;
	const uint32_t gl = fr.sgroup_log2;
	const uint32_t per_group = fr.slots << gl;
	const uint32_t sgroup = fast_div(slot, fr.div_group); // (slots are < 2^31: rt_types.h)
	const uint32_t rem = slot - sgroup * per_group;
	const uint32_t tile = rem >> (6u + gl), pix = (rem >> gl) & 63u;
	p.sample = (sgroup << gl) + (rem & ((1u << gl) - 1u));
	const uint32_t ty = fast_div(tile, fr.div_tiles_x), tx = tile - ty * fr.tiles_x;
	p.x = tx * TILE + tile_pix_x(pix);
	const uint32_t yl = ty * TILE + tile_pix_y(pix);
	p.y = local_to_global_row(fr, yl);
	p.local = yl * fr.W + p.x;
	p.valid = p.x < fr.W && p.y < fr.H;
	return p;
}

// ---------------------------------------------------------------------------------------------------------------
// primary rays
// ---------------------------------------------------------------------------------------------------------------
// EmbreeRT: u = (x + r0)/W, v = (y + r1)/H, point = p1 + (u*right + v*up), dir = (point - org) * (1/sqrt(len2)) with
// len2 accumulated x, y, z (Ray.cpp:318-373).  The lens sample follows the scalar form Ray.cpp:16-47.
RT_FN void parity_primary_ray(const CamView &cam, const FrameView &fr, uint32_t x, uint32_t y, float r0, float r1,
							  float r2, float r3, f3 &O, f3 &D)
{
	f3 org = cam.pos;
	if (cam.aperture != 0.0f)
	{
		const float blade = (float)(int)(r0 * 9);
		r2 = (r2 - blade * (1.0f / 9.0f)) * 9.0f;
		const float piOver4point5 = 3.14159265359f / 4.5f;
		const float x1 = m_cosf(blade * piOver4point5), y1 = m_sinf(blade * piOver4point5);
		const float x2 = m_cosf((blade + 1.0f) * piOver4point5), y2 = m_sinf((blade + 1.0f) * piOver4point5);
		if ((r2 + r3) > 1.0f)
			r2 = 1.0f - r2, r3 = 1.0f - r3;
		const float xr = x1 * r2 + x2 * r3, yr = y1 * r2 + y2 * r3;
		org = cam.pos + (cam.right * xr + cam.up * yr) * cam.aperture;
	}
	const float u = ((float)x + r0) * fr.inv_w; // (1.0f / (float)W, Ray.cpp:318-373)
	const float v = ((float)y + r1) * fr.inv_h;
	const f3 pix = cam.p1 + (cam.right * u + cam.up * v);
	const f3 d = pix - org;
	float l2 = d.x * d.x;
	l2 = d.y * d.y + l2;
	l2 = d.z * d.z + l2;
	const float inv = 1.0f / sqrtf(l2);
	O = org;
	D = d * inv;
}

// blueNoiseSampler (bsdf/tools.h:163-181, CUDART/src/Kernels.cu:205-223): Sobol' sequence value of (sample, dimension),
// ranked and scrambled per pixel of a 128 x 128 tile.  table = [sobol 256 x 256 | scrambling tile | ranking tile]
// (blue_noise.h:8204: sobol at 0, scrambling at 65536, ranking at 3 * 65536).
RT_FN float blue_noise_sample(const uint32_t *table, int x, int y, int sampleIdx, int dim)
{
	x &= 127, y &= 127, sampleIdx &= 255, dim &= 255;
	uint32_t ri = (uint32_t)dim + (uint32_t)(x + y * 128) * 8u + 65536u * 3u;
	if (ri >= BLUE_NOISE_WORDS) // dimensions >= 8 of the last pixels index past the table in the reference
		ri = BLUE_NOISE_WORDS - 1u;
	const int ranked = (sampleIdx ^ (int)table[ri]) & 255;
	int value = (int)table[dim + ranked * 256];
	value ^= (int)table[(dim & 7) + (x + y * 128) * 8 + 65536];
	return (0.5f + (float)value) * (1.0f / 256.0f);
}

// CUDART generatePrimaryRay (Kernels.cu:383-426): r0..r3 from the blue-noise sampler when a table was handed over
// (what the reference runs), else its hash-RNG branch: seed = WangHash(pixel*16789 + sample*1791), four RandomFloat.
RT_FN void pt_primary_ray(const CamView &cam, const FrameView &fr, uint32_t x, uint32_t y, uint32_t sampleIdx,
						  f3 &O, f3 &D)
{
	const uint32_t pixel = y * fr.W + x;
	float r0, r1, r2, r3;
	if (cam.blue_noise)
	{
		r0 = blue_noise_sample(cam.blue_noise, (int)x, (int)y, (int)sampleIdx, 0);
		r1 = blue_noise_sample(cam.blue_noise, (int)x, (int)y, (int)sampleIdx, 1);
		r2 = blue_noise_sample(cam.blue_noise, (int)x, (int)y, (int)sampleIdx, 2);
		r3 = blue_noise_sample(cam.blue_noise, (int)x, (int)y, (int)sampleIdx, 3);
	}
	else
	{
		uint32_t seed = wang_hash(pixel * 16789u + sampleIdx * 1791u);
		r0 = random_float(seed), r1 = random_float(seed);
		r2 = random_float(seed), r3 = random_float(seed);
	}
	O = cam.pos;
	if (cam.aperture != 0.0f) // with aperture 0 the lens offset is exactly zero: skip the four trig evaluations
	{
		const float blade = (float)(int)(r0 * 9);
		r2 = (r2 - blade * (1.0f / 9.0f)) * 9.0f;
		const float piOver4point5 = 3.14159265359f / 4.5f;
		// __sincosf(a, &x1, &y1): x1 = sin, y1 = cos (Kernels.cu:407-408)
		const float x1 = m_sinf(blade * piOver4point5), y1 = m_cosf(blade * piOver4point5);
		const float x2 = m_sinf((blade + 1.0f) * piOver4point5), y2 = m_cosf((blade + 1.0f) * piOver4point5);
		if ((r2 + r3) > 1.0f)
			r2 = 1.0f - r2, r3 = 1.0f - r3;
		// (fixed-shape arithmetic from here on: see rounded())
		const float xr = fmaf(x2, r3, rounded(x1 * r2)), yr = fmaf(y2, r3, rounded(y1 * r2));
		O = madd2_r(cam.pos, cam.right, rounded(xr * cam.aperture), cam.up, rounded(yr * cam.aperture));
	}
	const float u = rounded(((float)x + r0) * fr.inv_w), v = rounded(((float)y + r1) * fr.inv_h); // (x (1.0f / (float)W): Kernels.cu:418-419)
	D = normalize_r(madd2_r(cam.p1, cam.right, u, cam.up, v) - O);
}

// ---------------------------------------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------------------------------------
struct Hit
{
	float t, u, v;
	int prim, inst;
};
struct TStat
{
	uint32_t inner, tris;
	uint32_t lds; // of `inner`: node visits served by the LDS top-of-tree cache (no vector-L1 traffic)
};

// LDS stack layout: stack[entry][thread of the workgroup] (bank = thread % 32: conflict-free); the stride = the kernel's
// workgroup size travels in TravStack::stride (a compile-time constant after inlining).  Host emulation: stride 1.
constexpr uint32_t ENTRY_DONE = 0xFFFFFFFEu;

// The top of the largest BLAS, kept in LDS by every traversal workgroup: the four rows of the compressed nodes
// top_first .. top_first + top_count - 1.  A divergent 16-byte load costs the CU's vector L1 one cycle per lane
// (profiles/micro/gather_micro.hip: 9.6 TB/s chip-wide whatever the table size) and that rate bounds the traversal
// kernels; the same rows from LDS cost 4 cycles per 64 lanes, and every ray walks through these nodes.
#ifndef RT_LDS_NODES
#define RT_LDS_NODES 128 // about four levels (swept 0 / 21 / 85 / 128 / 170 / 341 against the LDS stack depth)
#endif
constexpr uint32_t MAX_LDS_NODES = RT_LDS_NODES;
constexpr uint32_t TOP_ROWS = 4; // a compressed node is exactly four rows (the emulation's `top` aliases the node table)

struct TravStack
{
	uint32_t *lds;	 // this lane's column of the workgroup's LDS stack
	uint32_t *spill; // SPILL_STACK private entries behind it (a plain local array of the caller: keeping it out of
					 // the traversal state lets that state live in registers)
	const f4 *top;	 // staged top-of-tree rows (TOP_ROWS per node)
	uint32_t top_first, top_count;
	uint32_t *overflow; // WaveCounters::stack_overflow
	uint32_t stride;	// entries between two stack levels of one lane (= workgroup size on the device)
};

// 1/d for the slab test.  A direction component of exactly 0 (it happens: jitter r0 == 1.0f puts a ray on the image's
// centre line) would give inf, and inf * 0 = NaN in the fma form below makes the slab test ignore that axis — the ray
// then visits every box along it.  Such components are replaced by +-1e-30, i.e. a finite 1e30.
RT_FN float safe_rcp(float d)
{
	return 1.0f / (fabsf(d) > 1e-30f ? d : copysignf(1e-30f, d));
}

RT_FN float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_STRICT_MATH)
	return __builtin_amdgcn_rcpf(x); // v_rcp_f32, 1 ulp
#else
	return 1.0f / x;
#endif
}

// RT_FAST_ID: 1/d of the slab test by v_rcp_f32 (1 ulp) instead of a correctly rounded division (ten instructions, three times
// per ray and change of space).  1/d only feeds the box tests — which boxes a ray enters, never what it hits there: the
// triangle test takes o and d — and the boxes carry a margin of 2e-5 plus the outward rounding of their quantisation.
#ifndef RT_FAST_ID
#define RT_FAST_ID 1
#endif
RT_FN float slab_rcp(float d)
{
	const float c = fabsf(d) > 1e-30f ? d : copysignf(1e-30f, d);
#if RT_FAST_ID
	return fast_rcp(c);
#else
	return 1.0f / c;
#endif
}

// RT_NORM_T: the slab test in NORMALISED distances s = t * k with k = (1 - 2^-16) / hit.t (k = 2^-100 while the ray has no hit
// and no bound): the interval a box must meet, [0, hit.t], becomes [0, 1] — exactly what the `clamp` output modifier of
// v_max3_f32 / v_min3_f32 clamps to for free.  With near' = clamp(max3(..)), far' = clamp(min3(..)) the reference's three
// conditions (aabb.cpp:39-77: tmax > tmin && tmin < t, plus tmax >= 0 here) are ONE compare, near' < far': per child
// max3, min3, one compare, one select instead of max3, min3, three compares, one select (all of them 4-clock instructions).
// k multiplies 1/d and o/d of the ray (ids, oids), so the node step itself is unchanged; when a closest-hit ray finds a nearer
// hit the six values are rescaled by hit.t_old / hit.t_new (a handful of instructions, once or twice per ray).  The factor
// 1 - 2^-16 keeps the far clamp conservative against the rounding of k and of the rescaling: a box is culled by distance only
// when its entry lies beyond hit.t * (1 + 1.5e-5).  (A box that ENDS exactly at the ray's origin, tmax == 0, was entered before
// and is not now; nothing in it can be hit: t > t_min = 1e-5.)
#ifndef RT_NORM_T
#define RT_NORM_T 1
#endif

RT_FN float norm_k(float t)
{
	// t >= 1e30: "no bound" (the integrators' 1e34): a power of two, so the scaling is exact; k <= 1e3 keeps o/d * k finite
	// for the shortest shadow rays (their far clamp is then merely looser)
	const float tt = fmaxf(t, 1e-3f);
	return t >= 1e30f ? 7.8886090522101181e-31f : 0.99998474f * fast_rcp(tt);
}
// max3 / min3 clamped to [0, 1]: one instruction each on the device (output modifier)
RT_FN float max3_clamp01(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	float r;
	asm("v_max3_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
#else
	return fminf(fmaxf(fmaxf(fmaxf(a, b), c), 0.0f), 1.0f);
#endif
}
RT_FN float min3_clamp01(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	float r;
	asm("v_min3_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
#else
	return fminf(fmaxf(fminf(fminf(a, b), c), 0.0f), 1.0f);
#endif
}

// Slab test (aabb.cpp:39-77) in fma form: t = b * (1/d) - o * (1/d).  a = bmin.xyz, bmax.x ; b = bmax.y, bmax.z, ...
// The reference accepts tmax > tmin && tmin < t; boxes entirely behind the origin (tmax < 0) cannot contain an
// accepted hit (t > t_min >= 0) and are culled as well.
RT_FN bool slab(const f4 &a, const f4 &b, f3 id, f3 oid, float t, float &tnear)
{
	const float tx1 = fmaf(a.x, id.x, -oid.x), tx2 = fmaf(a.w, id.x, -oid.x);
	const float ty1 = fmaf(a.y, id.y, -oid.y), ty2 = fmaf(b.x, id.y, -oid.y);
	const float tz1 = fmaf(a.z, id.z, -oid.z), tz2 = fmaf(b.y, id.z, -oid.z);
	const float tmin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
	const float tmax = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
	tnear = tmin;
	return tmax > tmin && tmin < t && tmax >= 0.0f;
}

// The four 16-byte rows of a compressed 4-wide node (rt::Node4c).
//  * device, global table (PIN): the four loads are pinned by one empty asm that names every row, so they are issued back
//    to back (one round trip; none is sunk into the branch that uses it, none is waited for while others are still to be
//    issued).  The LDS copy of the top of the tree needs no pin.
struct Node4cRows
{
	f4 r0, r1, r2, r3;
};
template <bool PIN> RT_FN Node4cRows load_rows(const char *base, uint32_t nb)
{
	Node4cRows r;
#if defined(__HIP_DEVICE_COMPILE__)
	typedef float v4f __attribute__((ext_vector_type(4)));
	v4f a = *(const v4f *)(base + nb), b = *(const v4f *)(base + (nb + 16u));
	v4f c = *(const v4f *)(base + (nb + 32u)), d = *(const v4f *)(base + (nb + 48u));
	if (PIN)
		asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#define RT_F4(V) mk4(V.x, V.y, V.z, V.w)
	r.r0 = RT_F4(a), r.r1 = RT_F4(b), r.r2 = RT_F4(c), r.r3 = RT_F4(d);
#undef RT_F4
#else
	r.r0 = *(const f4 *)(base + nb), r.r1 = *(const f4 *)(base + (nb + 16u));
	r.r2 = *(const f4 *)(base + (nb + 32u)), r.r3 = *(const f4 *)(base + (nb + 48u));
#endif
	return r;
}
// byte k of a dword as float: one v_cvt_f32_ubyteK each
RT_FN float ub0(uint32_t x) { return (float)(x & 255u); }
RT_FN float ub1(uint32_t x) { return (float)((x >> 8) & 255u); }
RT_FN float ub2(uint32_t x) { return (float)((x >> 16) & 255u); }
RT_FN float ub3(uint32_t x) { return (float)(x >> 24); }

// Möller–Trumbore with the reference's rejections: |a| < 1e-6, u outside [0,1], v < 0, u+v > 1, t <= t_min, t >= t.
// TIE (closest-hit queries): of two triangles hit at BIT-IDENTICAL distance the one with the lower primitive id is the hit.  The
// reference keeps whichever its traversal reaches first (strict t > tt, bvh_tree.cpp:166-196) — an answer that depends on the
// shape of its tree; here several traversals serve the same rays (one ray per lane, persistent lanes, the packet form of the
// primary wave; host-built and device-built trees), and a ray exactly on an edge shared by two triangles must not get a
// different triangle — normal, material — from each of them: with a total order on (t, prim) the hit record is a function of
// the ray and the scene alone.  (About one primary ray in a million on the terrain; occlusion queries have no such question.)
template <bool TIE = false>
RT_FN bool tri_test(f3 o, f3 d, float t_min, float &t, f3 p0, f3 p1, f3 p2, float &uo, float &vo, float eps = TRI_EPS, uint32_t prim = 0u,
					uint32_t cur_prim = 0u, uint32_t inst = 0u, uint32_t cur_inst = 0u)
{
	// (fixed-shape arithmetic: see rounded())
	const f3 e1 = p1 - p0, e2 = p2 - p0;
	const f3 h = cross_r(d, e2);
	const float a = dot_r(e1, h);
	// bvh_tree.cpp:174: |a| < 1e-6 rejects the triangle — in the space the reference tests it in, the instance's OBJECT space.  A
	// triangle of the world tree (written out in world space, M p instead of M^-1 o) has a_world = det(M) a_object, so its threshold
	// travels with it: eps = w of its third vertex = 1e-6 |det M| (1e-6 for every other triangle).  Round 5's advisor: with the
	// constant, a finely tessellated mesh instanced at scale 0.002 lost 696 of 697 primary hits to the world tree.
	if (a > -eps && a < eps)
		return false;
	const float f = fast_rcp(a);
	const f3 s = o - p0;
	const float u = rounded(f * dot_r(s, h));
	if (u < 0.0f || u > 1.0f)
		return false;
	const f3 q = cross_r(s, e1);
	const float v = rounded(f * dot_r(d, q));
	if (v < 0.0f || u + v > 1.0f)
		return false;
	const float tt = rounded(f * dot_r(e2, q));
	bool nearer = t > tt;
	// total order on (t, instance, primitive): primitive ids are mesh-relative, and the world tree puts the triangles of many
	// instances (of one mesh, too) into one leaf space
	if (TIE)
		nearer = nearer || (t == tt && (inst < cur_inst || (inst == cur_inst && prim < cur_prim)));
	if (tt > t_min && nearer)
	{
		t = tt, uo = u, vo = v;
		return true;
	}
	return false;
}

// Two-level traversal, "while-while" form for wave64: every lane first descends through inner nodes until it holds
// a leaf (lanes that already found one wait), then the wave processes leaves together.  With incoherent rays an
// if-inner/if-leaf loop would execute the triangle code in nearly every iteration for a handful of lanes.
//
// Node::left_first holds the ready-made stack entry of the node on the device (see make_entry): inner node = index
// of its left child (children are adjacent), leaf = first/count packed, ENTRY_TLAS marks top-level entries,
// ENTRY_SENTINEL on the stack marks "leave the instance".
// ANY = true: occlusion query, finishes on the first accepted hit in (t_min, t_max).
//
// The traversal is a resumable per-lane state machine (begin / descend / visit) so that the persistent kernels can
// hand a finished lane a new ray while its neighbours are still busy; trace() below runs it to completion.
// (Speculative traversal — a lane takes its first triangle leaf in hand and walks on — was built in round 3, is bit-identical and
// loses; round 5 took it out of the sources: DESIGN_LOG.md, "variants removed".)
// WORLD = false: the lane does not keep the world-space ray beside the ray of the current space — the caller hands it to
// visit() on the two occasions it is needed (entering and leaving an instance), e.g. by reading the ray record again (the
// persistent-lane kernels: six registers less per lane, and instance switches are rare once static geometry is linked flat).
template <bool WORLD> struct TraverserWorld
{
	f3 O, D; // world-space ray
};
template <> struct TraverserWorld<false>
{
};
template <bool ANY, bool COUNT, bool WORLD = true>
struct Traverser : TraverserWorld<WORLD>
{
	f3 o, d, id, oid; // ray in the current space (world, or the object space of cur_inst), 1/d, o/d
	bool neg_x, neg_y, neg_z; // direction signs: which of a child's two planes per axis is the entry plane
	int cur_inst;
	int sp;
	uint32_t cur; // entry in hand; ENTRY_DONE when the lane has no work
	float t_min;
	Hit hit;

	// the ray in the space about to be traversed.  Which of a child's two planes per axis is the entry plane depends
	// only on the direction's sign, so it is resolved once here into load offsets (lo row or hi row of the Node4)
	// instead of a min/max pair per plane in the loop.
	RT_FN void enter_space(f3 o_, f3 d_)
	{
		o = o_, d = d_;
		id = mk3(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
#if RT_NORM_T
		id = id * norm_k(hit.t); // (id, oid hold the NORMALISED 1/d and o/d: see RT_NORM_T)
#endif
		oid = o * id;
		neg_x = id.x < 0.0f, neg_y = id.y < 0.0f, neg_z = id.z < 0.0f;
	}
	// RT_NORM_T: the closest hit moved from t_old to hit.t — rescale the normalised 1/d and o/d
	RT_FN void renormalise(float t_old)
	{
#if RT_NORM_T
		// (the last factor, 1 - 2^-21, outweighs the rounding of the line: however often a ray's hit moves, k only drifts DOWN —
		// towards a looser far clamp — never past (1 - 2^-16) / hit.t)
		const float r = norm_k(hit.t) * fast_rcp(norm_k(t_old)) * 0.99999952f;
		id = id * r, oid = oid * r;
#endif
	}

	RT_FN void begin(const SceneView &sc, f3 O_, f3 D_, float t_min_, float t_max)
	{
		if constexpr (WORLD)
			this->O = O_, this->D = D_;
		t_min = t_min_;
		hit.t = t_max, hit.u = 0.0f, hit.v = 0.0f, hit.prim = -1, hit.inst = -1;
		cur_inst = -1, sp = 0;
		enter_space(O_, D_);
		cur = sc.instance_count ? sc.tlas_root_entry : ENTRY_DONE;
	}
	// the ray in the object space of an instance (3x4 inverse, direction NOT renormalised so that t is shared:
	// top_level_bvh.cpp:104-168)
	RT_FN void enter_instance(const Instance &in, f3 O, f3 D)
	{
		enter_space(mk3(row_point_r(in.inv, O), row_point_r(in.inv + 4, O), row_point_r(in.inv + 8, O)),
					mk3(row_dir_r(in.inv, D), row_dir_r(in.inv + 4, D), row_dir_r(in.inv + 8, D)));
	}
	RT_FN bool done() const { return cur == ENTRY_DONE; }
	// a triangle leaf (not a top-level leaf, the sentinel or ENTRY_DONE, which all carry the ENTRY_TLAS bit)
	static RT_FN bool tri_leaf(uint32_t e) { return (e & (ENTRY_LEAF | ENTRY_TLAS)) == ENTRY_LEAF; }
	// has this lane nothing left to do in the node phase?
	RT_FN bool parked() const { return (cur & ENTRY_LEAF) != 0u; }

	static constexpr int LDS_DEPTH = ANY ? LDS_STACK_ANY : LDS_STACK;
	static constexpr float NODE_INF = 3.0e38f;
	RT_FN void push(const TravStack stk, uint32_t e)
	{
		if (sp < LDS_DEPTH)
			stk.lds[sp * stk.stride] = e;
		else if (sp < LDS_DEPTH + SPILL_STACK)
			stk.spill[sp - LDS_DEPTH] = e;
		else // cannot happen for trees rfwhip_update() accepted; counted so that it can never go unnoticed
		{
#if defined(__HIP_DEVICE_COMPILE__)
			atomicAdd(stk.overflow, 1u);
#else
			(*stk.overflow)++;
#endif
		}
		sp++;
	}
	// next entry from the stack; ENTRY_SENTINEL comes back like a leaf and is resolved in visit().  The common case
	// (stack within its LDS part) is branch-free: an unconditional read of the clamped slot and two selects.
	RT_FN uint32_t pop(const TravStack stk)
	{
		if (sp > LDS_DEPTH)
		{
			sp--;
			return sp < LDS_DEPTH + SPILL_STACK ? stk.spill[sp - LDS_DEPTH] : ENTRY_DONE; // builders bound the depth
		}
		const int s = sp > 0 ? sp - 1 : 0;
		const uint32_t e = stk.lds[s * stk.stride];
		const uint32_t r = sp > 0 ? e : ENTRY_DONE;
		sp = s;
		return r;
	}
	// Up to three entries at once, in this order, each only if its flag is set.  While the stack stays within its LDS
	// part the three stores are unconditional — an unwanted entry lands on the slot the next wanted one overwrites, or
	// just above the new top — so the usual case costs one branch instead of nine.
	RT_FN void push3(const TravStack stk, uint32_t ea, bool fa, uint32_t eb, bool fb, uint32_t ec, bool fc)
	{
		if (sp + 3 <= LDS_DEPTH)
		{
			const int na = fa ? 1 : 0, nb = fb ? 1 : 0, nc = fc ? 1 : 0;
			stk.lds[sp * stk.stride] = ea;
			stk.lds[(sp + na) * stk.stride] = eb;
			stk.lds[(sp + na + nb) * stk.stride] = ec;
			sp += na + nb + nc;
		}
		else
		{
			if (fa)
				push(stk, ea);
			if (fb)
				push(stk, eb);
			if (fc)
				push(stk, ec);
		}
	}

	// One 4-wide node: fetch its four rows (LDS copy of the top of the tree, or the table), slab-test the four children
	// (aabb.cpp:39-77 in fma form) and — closest-hit rays — order them by entry distance.  Out: t[k] = entry distance or INF
	// (a miss, an empty slot = an inverted box), e[k] = the children's stack entries; closest-hit: nearest first.
	RT_FN void node_step(const SceneView &sc, const TravStack stk, TStat &st, uint32_t entry, float &t0, float &t1, float &t2, float &t3,
						 uint32_t &e0, uint32_t &e1, uint32_t &e2, uint32_t &e3)
	{
		const uint32_t idx = entry & ENTRY_INDEX_MASK;
		const uint32_t rel = idx - stk.top_first;
		Node4cRows rows;
		if (rel < stk.top_count)
		{
			rows = load_rows<false>((const char *)stk.top, rel * (TOP_ROWS * 16u));
			if (COUNT)
				st.lds++;
		}
		else // byte offset of the node in the table (tables stay below 4 GiB)
			rows = load_rows<true>((const char *)sc.nodes4, idx << 6);
		if (COUNT)
			st.inner++;
		// plane = org + q * 2^e  =>  distance = q * (2^e / d) + (org / d - o / d): three scales and three offsets per node,
		// then one v_cvt_f32_ubyte + one fma per plane.  Which of a child's two planes per axis is the entry plane depends
		// only on the sign of the direction: resolved per node by swapping the lo / hi dwords of the axis.
		const float Ax = rows.r0.w * id.x, Ay = rows.r3.z * id.y, Az = rows.r3.w * id.z;
		const float Bx = fmaf(rows.r0.x, id.x, -oid.x), By = fmaf(rows.r0.y, id.y, -oid.y), Bz = fmaf(rows.r0.z, id.z, -oid.z);
		const uint32_t lox = fbits(rows.r2.x), loy = fbits(rows.r2.y), loz = fbits(rows.r2.z);
		const uint32_t hix = fbits(rows.r2.w), hiy = fbits(rows.r3.x), hiz = fbits(rows.r3.y);
		const uint32_t nxq = neg_x ? hix : lox, fxq = neg_x ? lox : hix;
		const uint32_t nyq = neg_y ? hiy : loy, fyq = neg_y ? loy : hiy;
		const uint32_t nzq = neg_z ? hiz : loz, fzq = neg_z ? loz : hiz;
		const float INF = NODE_INF;
#if RT_NORM_T
#define RT_SLAB4(UB, OUT)                                                                                   \
	{                                                                                                       \
		const float tmin = max3_clamp01(fmaf(UB(nxq), Ax, Bx), fmaf(UB(nyq), Ay, By), fmaf(UB(nzq), Az, Bz)); \
		const float tmax = min3_clamp01(fmaf(UB(fxq), Ax, Bx), fmaf(UB(fyq), Ay, By), fmaf(UB(fzq), Az, Bz)); \
		OUT = tmax > tmin ? tmin : INF;                                                                     \
	}
#else
#define RT_SLAB4(UB, OUT)                                                                                   \
	{                                                                                                       \
		const float tmin = fmaxf(fmaxf(fmaf(UB(nxq), Ax, Bx), fmaf(UB(nyq), Ay, By)), fmaf(UB(nzq), Az, Bz)); \
		const float tmax = fminf(fminf(fmaf(UB(fxq), Ax, Bx), fmaf(UB(fyq), Ay, By)), fmaf(UB(fzq), Az, Bz)); \
		OUT = (tmax > tmin && tmin < hit.t && tmax >= 0.0f) ? tmin : INF;                                   \
	}
#endif
		RT_SLAB4(ub0, t0)
		RT_SLAB4(ub1, t1)
		RT_SLAB4(ub2, t2)
		RT_SLAB4(ub3, t3)
#undef RT_SLAB4
		e0 = fbits(rows.r1.x), e1 = fbits(rows.r1.y), e2 = fbits(rows.r1.z), e3 = fbits(rows.r1.w);
		if (!ANY)
		{
			// order the four children by entry distance (5-comparator network), nearest first
#define RT_CSWAP(TA, EA, TB, EB)              \
	{                                         \
		const bool sw = TB < TA;              \
		const float tt = sw ? TB : TA;        \
		const uint32_t ee = sw ? EB : EA;     \
		TB = sw ? TA : TB, EB = sw ? EA : EB; \
		TA = tt, EA = ee;                     \
	}
			RT_CSWAP(t0, e0, t1, e1)
			RT_CSWAP(t2, e2, t3, e3)
			RT_CSWAP(t0, e0, t2, e2)
			RT_CSWAP(t1, e1, t3, e3)
			RT_CSWAP(t1, e1, t2, e2)
#undef RT_CSWAP
		}
	}
	// What a node step leaves behind: the children that were hit (t < limit = NODE_INF) go on the stack, the next entry comes
	// back.  closest-hit: t sorted, nearest first; occlusion: any order, the first child hit is next.
	RT_FN uint32_t take_children(const TravStack stk, float limit, float t0, float t1, float t2, float t3, uint32_t e0, uint32_t e1,
								 uint32_t e2, uint32_t e3)
	{
		if (!ANY)
		{
			if (t0 < limit)
			{
				// far children first, so the nearest of them is popped first
				push3(stk, e3, t3 < limit, e2, t2 < limit, e1, t1 < limit);
				return e0;
			}
			return pop(stk);
		}
		const bool h0 = t0 < limit, h1 = t1 < limit, h2 = t2 < limit, h3 = t3 < limit;
		const bool have = h0 || h1 || h2 || h3;
		const uint32_t next = h0 ? e0 : (h1 ? e1 : (h2 ? e2 : e3));
		if (have)
			push3(stk, e1, h1 && h0, e2, h2 && (h0 || h1), e3, h3 && (h0 || h1 || h2));
		return have ? next : pop(stk);
	}

	// phase 1: walk 4-wide inner nodes until this lane holds a leaf entry (or ENTRY_DONE / ENTRY_SENTINEL)
	// VOTE < 64 (wave kernels only): the wave leaves the node phase as soon as VOTE / 64 of its lanes with a ray hold a leaf,
	// so that those lanes do not sit idle while the others finish a long descent (lanes still on an inner node skip visit()).
#ifndef RT_VOTE_RELATIVE
#define RT_VOTE_RELATIVE 1
#endif
	template <int VOTE = 64> RT_FN void descend(const SceneView &sc, const TravStack stk, TStat &st)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		const int nwork = VOTE < 64 ? __popcll(__ballot(!done())) : 0;
#endif
		for (;;)
		{
			if (cur & ENTRY_LEAF)
				break;
			float t0, t1, t2, t3;
			uint32_t e0, e1, e2, e3;
			node_step(sc, stk, st, cur, t0, t1, t2, t3, e0, e1, e2, e3);
			cur = take_children(stk, NODE_INF, t0, t1, t2, t3, e0, e1, e2, e3);
#if defined(__HIP_DEVICE_COMPILE__)
			if (VOTE < 64)
			{
				// only the lanes still inside this loop execute the ballot: the lanes with work that are NOT counted here
				// are the ones already waiting with a leaf (or with a finished ray to retire)
				const int inner = __popcll(__ballot(!parked()));
#if RT_VOTE_RELATIVE
				if ((nwork - inner) * 64 >= nwork * VOTE) // VOTE / 64 of the lanes that HAVE a ray (idle lanes do not count)
					break;
#else
				if (nwork - inner >= VOTE)
					break;
#endif
			}
#endif
		}
	}

	// phase 2: the leaf in hand — enter an instance (top-level leaf) or test the triangles, then fetch the next entry
	RT_FN void visit(const SceneView &sc, const TravStack stk, TStat &st)
	{
		static_assert(WORLD, "a traverser without the world-space ray is told it: visit(sc, stk, st, world)");
		visit(sc, stk, st, [this](f3 &O_, f3 &D_) { O_ = this->O, D_ = this->D; });
	}
	// world(O, D): the world-space ray of this lane (asked for when an instance is entered or left)
	template <typename F> RT_FN void visit(const SceneView &sc, const TravStack stk, TStat &st, F world)
	{
		const uint32_t leaf = cur;
		if (cur == ENTRY_DONE || !(cur & ENTRY_LEAF)) // (an inner node in hand: the wave left descend<VOTE>() early)
			return;
		else if (cur == ENTRY_SENTINEL)
		{
			// leaving an instance: back to the world-space ray
			f3 O, D;
			world(O, D);
			enter_space(O, D);
			cur_inst = -1;
			cur = pop(stk);
			return;
		}
		else if (cur & ENTRY_TLAS)
		{
			// top-level leaf: exactly one instance (the TLAS builder never merges)
			const uint32_t ii = sc.tlas_prims[cur & ENTRY_FIRST_MASK];
			const Instance &in = sc.instances[ii];
			push(stk, ENTRY_SENTINEL);
			f3 O, D;
			world(O, D);
			enter_instance(in, O, D);
			cur_inst = (int)ii;
			cur = in.root_entry;
			return;
		}
		const uint32_t first = leaf & ENTRY_FIRST_MASK;
		const uint32_t count = ((leaf >> 27) & 7u) + 1u;
#if RT_NORM_T
		const float t_before = hit.t;
#endif
		for (uint32_t i = 0; i < count; i++)
		{
			const f4 *tv = sc.tri_verts + 3u * (first + i);
			const f4 v0 = tv[0], v1 = tv[1], v2 = tv[2];
			if (COUNT)
				st.tris++;
			// (outside every instance: the triangle belongs to an instance that was linked into the top-level tree directly,
			// and carries its index — rfwhip_update, "flat" instances)
			const int tri_inst = cur_inst >= 0 ? cur_inst : (int)fbits(v1.w);
			if (tri_test<!ANY>(o, d, t_min, hit.t, xyz(v0), xyz(v1), xyz(v2), hit.u, hit.v, v2.w, fbits(v0.w), (uint32_t)hit.prim, (uint32_t)tri_inst,
							   (uint32_t)hit.inst))
			{
				hit.prim = (int)fbits(v0.w);
				hit.inst = tri_inst;
				if (ANY)
				{
					cur = ENTRY_DONE;
					return;
				}
			}
		}
#if RT_NORM_T
		if (!ANY && hit.t != t_before)
			renormalise(t_before);
#endif
		cur = pop(stk);
	}
};

template <bool ANY, bool COUNT>
RT_FN bool trace(const SceneView &sc, f3 O, f3 D, float t_min, float t_max, Hit &hit, const TravStack stk, TStat &st)
{
	Traverser<ANY, COUNT> T;
	T.begin(sc, O, D, t_min, t_max);
	while (!T.done())
	{
		T.descend(sc, stk, st);
		T.visit(sc, stk, st);
	}
	hit = T.hit;
	return hit.prim >= 0;
}

// ---------------------------------------------------------------------------------------------------------------
// materials / textures
// ---------------------------------------------------------------------------------------------------------------
RT_FN f3 material_color(const MaterialRec &m)
{
	return mk3(half_to_float(m.diffuse[0]), half_to_float(m.diffuse[1]), half_to_float(m.diffuse[2]));
}
RT_FN bool mat_flag(uint32_t flags, int bit) { return ((flags >> bit) & 1u) != 0; }
enum
{
	MF_DIFFUSE_MAP = 2,
	MF_NORMAL_MAP = 3,
	MF_2ND_NORMAL_MAP = 7,
	MF_3RD_NORMAL_MAP = 8,
	MF_2ND_DIFFUSE_MAP = 9,
	MF_3RD_DIFFUSE_MAP = 10,
	MF_SMOOTH_NORMALS = 11,
	MF_ALPHA = 12
};

RT_FN f3 mul_normal(const Instance &in, f3 n)
{
	return mk3(in.nrm[0] * n.x + in.nrm[4] * n.y + in.nrm[8] * n.z, in.nrm[1] * n.x + in.nrm[5] * n.y + in.nrm[9] * n.z,
			   in.nrm[2] * n.x + in.nrm[6] * n.y + in.nrm[10] * n.z);
}

RT_FN f4 decode_rgba8(uint32_t t)
{
	const float r = 1.0f / 256.0f;
	return mk4((float)(t & 255u) * r, (float)((t >> 8) & 255u) * r, (float)((t >> 16) & 255u) * r, (float)(t >> 24) * r);
}
RT_FN f4 load_texel(const SceneView &sc, const TexDesc &td, uint32_t i)
{
	if (i >= td.texelCount)
		i = td.texelCount - 1;
	if (td.type == 1u)
		return decode_rgba8(sc.tex_u32[td.offset + i]);
	return sc.tex_f4[td.offset + i];
}
// x % w for x >= 0, w >= 1 — the wrap of a texel coordinate (the coordinate is clamped at 0 before it is scaled, so it never is
// negative).  The compiler's 32-bit signed remainder is ~30 instructions, and a trilinear fetch needs eight: on the device the
// quotient comes from v_rcp_f32 instead — exact integers below 2^22 on both sides, so the estimate is off by at most one, which
// the two corrections take back; the remainder itself is integer arithmetic.  Same value as `%` for every input.
RT_FN int tex_wrap(int x, int w)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_STRICT_MATH)
	if (x < (1 << 22) && w < (1 << 22))
	{
		const uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)w));
		int r = x - (int)(q * (uint32_t)w);
		r += r < 0 ? w : 0;
		r -= r >= w ? w : 0;
		return r;
	}
#endif
	return x % w;
}
// getShadingData.h:25-60 — bilinear, wrap
RT_FN f4 fetch_texel(const SceneView &sc, const TexDesc &td, float tu, float tv, uint32_t o, int w, int h)
{
	const float tcx = (fmaxf(tu + 1000, 0.0f) * w) - 0.5f, tcy = (fmaxf(tv + 1000, 0.0f) * h) - 0.5f;
	const int iu = tex_wrap((int)tcx, w), iv = tex_wrap((int)tcy, h);
	const float fu = tcx - floorf(tcx), fv = tcy - floorf(tcy);
	const float w0 = (1 - fu) * (1 - fv), w1 = fu * (1 - fv), w2 = (1 - fu) * fv, w3 = 1 - (w0 + w1 + w2);
	const int iu1 = iu + 1 == w ? 0 : iu + 1, iv1 = iv + 1 == h ? 0 : iv + 1; // (iu + 1) % w with iu < w
	const f4 p0 = load_texel(sc, td, o + iu + (uint32_t)iv * w), p1 = load_texel(sc, td, o + iu1 + (uint32_t)iv * w);
	const f4 p2 = load_texel(sc, td, o + iu + (uint32_t)iv1 * w), p3 = load_texel(sc, td, o + iu1 + (uint32_t)iv1 * w);
	f4 r;
	r.x = 0.0f + p0.x * w0 + p1.x * w1 + p2.x * w2 + p3.x * w3;
	r.y = 0.0f + p0.y * w0 + p1.y * w1 + p2.y * w2 + p3.y * w3;
	r.z = 0.0f + p0.z * w0 + p1.z * w1 + p2.z * w2 + p3.z * w3;
	r.w = 0.0f + p0.w * w0 + p1.w * w1 + p2.w * w2 + p3.w * w3;
	return r;
}
// getShadingData.h:61-98 — MIPLEVELCOUNT 5; a texture without the appended chain is read at level 0 only
RT_FN f4 fetch_trilinear(const SceneView &sc, const TexDesc &td, float lambda, float tu, float tv, int width, int height)
{
	uint32_t chain = 0;
	{
		int w = width, h = height;
		for (int i = 0; i < 5; i++)
			chain += (uint32_t)w * h, w >>= 1, h >>= 1;
	}
	const bool has_mips = td.texelCount >= chain;
	// getShadingData.h:66-67: level0 = min(4, (int)lambda), level1 = min(4, level0 + 1) — NOT clamped at 0: for
	// lambda <= -1 both loops below run zero times and both levels are the base level
	int level0 = (int)lambda;
	if (level0 > 4)
		level0 = 4;
	int level1 = level0 + 1 > 4 ? 4 : level0 + 1;
	if (!has_mips)
		level0 = 0;
	if (!has_mips)
		level1 = 0;
	const float f = lambda - floorf(lambda);
	uint32_t o0 = 0, o1 = 0;
	int w0 = width, h0 = height, w1 = width, h1 = height;
	for (int i = 0; i < level0; i++)
		o0 += (uint32_t)w0 * h0, w0 >>= 1, h0 >>= 1;
	for (int i = 0; i < level1; i++)
		o1 += (uint32_t)w1 * h1, w1 >>= 1, h1 >>= 1;
	const f4 p0 = fetch_texel(sc, td, tu, tv, o0, w0 > 0 ? w0 : 1, h0 > 0 ? h0 : 1);
	const f4 p1 = fetch_texel(sc, td, tu, tv, o1, w1 > 0 ? w1 : 1, h1 > 0 ? h1 : 1);
	return mk4((1.0f - f) * p0.x + f * p1.x, (1.0f - f) * p0.y + f * p1.y, (1.0f - f) * p0.z + f * p1.z,
			   (1.0f - f) * p0.w + f * p1.w);
}

// ---------------------------------------------------------------------------------------------------------------
// sky
// ---------------------------------------------------------------------------------------------------------------
#define RT_INV_PI 0.318309886183790671538f
// EmbreeRT/src/Context.cpp:187-196: nearest texel at uv * (size - 1)
RT_FN f3 parity_sky(const SceneView &sc, f3 D)
{
	if (!sc.sky_w || !sc.sky_h)
		return mk3(0, 0, 0);
	const float ux = 0.5f * (1.0f + m_atan2f(D.x, -D.z) * RT_INV_PI);
	const float uy = m_acosf(clampf(D.y, -1.0f, 1.0f)) * RT_INV_PI;
	uint32_t px = f2u_sat(ux * (float)(sc.sky_w - 1)), py = f2u_sat(uy * (float)(sc.sky_h - 1));
	if (px >= sc.sky_w)
		px = sc.sky_w - 1;
	if (py >= sc.sky_h)
		py = sc.sky_h - 1;
	return xyz(sc.sky[py * sc.sky_w + px]);
}
// CUDART/src/Kernels.cu:593-600: texel at uv * size, black when out of range
RT_FN f3 pt_sky(const SceneView &sc, f3 D)
{
	if (!sc.sky_w || !sc.sky_h)
		return mk3(0, 0, 0);
	// (fixed shape, see rounded(): the product is rounded before the sum — a miss is shaded by the shade kernel or, for the packet
	// form of the primary wave, by that kernel itself, and one of two million pixels got another texel from each)
	const float turns = rounded(m_atan2f(D.x, -D.z) * RT_INV_PI);
	const uint32_t u = f2u_sat((float)sc.sky_w * 0.5f * (1.0f + turns));
	const uint32_t v = f2u_sat((float)sc.sky_h * m_acosf(clampf(D.y, -1.0f, 1.0f)) * RT_INV_PI);
	const unsigned long long idx = (unsigned long long)u + (unsigned long long)v * sc.sky_w;
	if (idx < (unsigned long long)sc.sky_w * sc.sky_h)
		return xyz(sc.sky[idx]);
	return mk3(0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// PARITY INTEGRATOR shade: EmbreeRT/src/Context.cpp:198-281 + retrieve_material :417-476
// ---------------------------------------------------------------------------------------------------------------
template <bool COUNT>
RT_FN f4 parity_shade(const SceneView &sc, f3 O, f3 D, const Hit &h, TravStack &stk, TStat &st, uint32_t &nshadow)
{
	const Instance &in = sc.instances[h.inst];
	const TriShade &ts = sc.tri_shade[in.shade_base + (uint32_t)h.prim];
	const f3 bary = mk3(1.0f - h.u - h.v, h.u, h.v);
	const f3 p = O + D * h.t;
	const f4 n0 = ts.n0, n1 = ts.n1, n2 = ts.n2;
	const MaterialRec &mat = sc.materials[fbits(ts.ex.w)];
	const f3 iNl = (xyz(n0) * bary.x + xyz(n1) * bary.y) + xyz(n2) * bary.z;
	const f3 iN = normalize_ieee(mul_normal(in, iNl));
	f3 color = material_color(mat);
	const uint32_t mflags = mat.flags;
	if (mat_flag(mflags, MF_DIFFUSE_MAP))
	{
		const TriUV &uv = sc.tri_uv[in.shade_base + (uint32_t)h.prim];
		const f4 tu4 = uv.tu, tv4 = uv.tv;
		const float tu = bary.x * tu4.x + bary.y * tu4.y + bary.z * tu4.z;
		const float tv = bary.x * tv4.x + bary.y * tv4.y + bary.z * tv4.z;
		const float uu = (tu + half_to_float(mat.map[0].uoffs)) * half_to_float(mat.map[0].uscale);
		const float vv = (tv + half_to_float(mat.map[0].voffs)) * half_to_float(mat.map[0].vscale);
		float tx = fmodf(uu, 1.0f), ty = fmodf(vv, 1.0f);
		if (tx < 0.f)
			tx = 1.f + tx;
		if (ty < 0.f)
			ty = 1.f + ty;
		const uint32_t ti = mat.map[0].addr;
		if (ti < sc.texture_count)
		{
			const TexDesc td = sc.textures[ti];
			const uint32_t ix = f2u_sat(tx * (float)(td.width - 1)), iy = f2u_sat(ty * (float)(td.height - 1));
			const uint32_t id = iy * td.width + ix;
			uint32_t tc;
			if (td.type == 0u)
			{
				const f4 px = sc.tex_f4[td.offset + id];
				color = color * xyz(px);
				// Context.cpp:458-472: no break — the float4 data is then decoded once more as packed RGBA8
				const float *raw = (const float *)(sc.tex_f4 + td.offset);
				tc = fbits(raw[id]);
			}
			else
				tc = sc.tex_u32[td.offset + id];
			const float s = 1.0f / 256.0f;
			color = (color * s) * mk3((float)(tc & 0xFFu), (float)((tc >> 8) & 0xFFu), (float)((tc >> 16) & 0xFFu));
		}
	}
	if (color.x > 1.0f || color.y > 1.0f || color.z > 1.0f)
		return mk4(color.x, color.y, color.z, 1.0f);
	f3 contrib = mk3(0.1f, 0.1f, 0.1f);
	Hit sh;
	for (uint32_t i = 0; i < sc.n_area; i++)
	{
		const AreaLight l = uniform_record(sc.area, i);
		f3 L = ld3(l.position) - p;
		const float sq = dot(L, L), dist = sqrtf(sq);
		L = mk3(L.x / dist, L.y / dist, L.z / dist);
		const float NdotL = dot(iN, L), LNdotL = -dot(ld3(l.normal), L);
		if (NdotL <= 0 || LNdotL <= 0)
			continue;
		nshadow++;
		if (!trace<true, COUNT>(sc, p, L, 1e-4f, dist, sh, stk, st))
		{
			f3 r = ld3(l.radiance) * l.area;
			r = mk3(r.x / sq, r.y / sq, r.z / sq);
			contrib = contrib + (r * NdotL) * LNdotL;
		}
	}
	for (uint32_t i = 0; i < sc.n_point; i++)
	{
		const PointLight l = uniform_record(sc.point, i);
		f3 L = ld3(l.position) - p;
		const float sq = dot(L, L), dist = sqrtf(sq);
		L = mk3(L.x / dist, L.y / dist, L.z / dist);
		const float NdotL = dot(iN, L);
		if (NdotL <= 0)
			continue;
		nshadow++;
		if (!trace<true, COUNT>(sc, p, L, 1e-4f, dist, sh, stk, st))
		{
			const f3 r = mk3(l.radiance[0] / sq, l.radiance[1] / sq, l.radiance[2] / sq);
			contrib = contrib + r * NdotL;
		}
	}
	const f3 c = color * contrib;
	return mk4(c.x, c.y, c.z, 1.0f);
}

// ---------------------------------------------------------------------------------------------------------------
// Disney BSDF (bsdf/disney.h) on the 16 packed 8-bit parameters (bsdf/compat.h:47-74)
// ---------------------------------------------------------------------------------------------------------------
struct Shading
{
	f3 color, absorption;
	uint32_t p0, p1, p2;
};
RT_FN float chan(uint32_t v, int shift) { return (float)((v >> shift) & 255u) * (1.0f / 255.0f); }
RT_FN float sd_metallic(const Shading &s) { return chan(s.p0, 0); }
RT_FN float sd_subsurface(const Shading &s) { return chan(s.p0, 8); }
RT_FN float sd_specular(const Shading &s) { return chan(s.p0, 16); }
RT_FN float sd_roughness(const Shading &s) { return fmaxf(0.001f, chan(s.p0, 24)); }
RT_FN float sd_spectint(const Shading &s) { return chan(s.p1, 0); }
RT_FN float sd_clearcoat(const Shading &s) { return chan(s.p2, 0); }
RT_FN float sd_clearcoatgloss(const Shading &s) { return chan(s.p2, 8); }
RT_FN float sd_transmission(const Shading &s) { return chan(s.p2, 16); }
RT_FN float sd_eta(const Shading &s) { return chan(s.p2, 24); }

#define RT_INVPI 0.318309886183790671537767526745028724f
#define RT_PI 3.14159265358979323846264338327950288f
#define RT_INV2PI 0.159154943091895335768883763372514362f
#define RT_TWOPI 6.28318530717958647692528676655900576f

RT_FN float sqr(float x) { return x * x; }
RT_FN float schlick_fresnel(float u)
{
	const float m = clampf(1.0f - u, 0.0f, 1.0f);
	return (m * m) * (m * m) * m;
}
RT_FN float gtr1(float NDotH, float a)
{
	if (a >= 1.0f)
		return RT_INVPI;
	const float a2 = a * a;
	const float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
	return m_div(a2 - 1.0f, RT_PI * m_logf(a2) * t);
}
RT_FN float gtr2(float NDotH, float a)
{
	const float a2 = a * a;
	const float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
	return m_div(a2, RT_PI * t * t);
}
RT_FN float smith_ggx(float NDotv, float alphaG)
{
	const float a = alphaG * alphaG;
	const float b = NDotv * NDotv;
	return m_rcp(NDotv + m_sqrtf(a + b - a * b));
}
RT_FN float fresnel_fr(float VDotN, float eio)
{
	const float SinThetaT2 = sqr(eio) * (1.0f - VDotN * VDotN);
	if (SinThetaT2 > 1.0f)
		return 1.0f;
	const float LDotN = m_sqrtf(1.0f - SinThetaT2);
	const float eta = m_rcp(eio);
	const float r1 = m_div(VDotN - eta * LDotN, VDotN + eta * LDotN);
	const float r2 = m_div(LDotN - eta * VDotN, LDotN + eta * VDotN);
	return 0.5f * (sqr(r1) + sqr(r2));
}
RT_FN f3 safe_normalize(f3 a)
{
	const float ls = dot(a, a);
	if (ls > 0.0f)
		return a * m_rsqrtf(ls);
	return mk3(0, 0, 0);
}
RT_FN bool refract_dir(f3 wi, f3 n, float eta, f3 &wt)
{
	const float cosThetaI = dot(n, wi);
	const float sin2ThetaI = fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI);
	const float sin2ThetaT = eta * eta * sin2ThetaI;
	if (sin2ThetaT >= 1.0f)
		return false;
	const float cosThetaT = m_sqrtf(1.0f - sin2ThetaT);
	wt = (wi * -1.0f) * eta + n * (eta * cosThetaI - cosThetaT);
	return true;
}
// disney.h:83-101
RT_FN float bsdf_pdf(const Shading &sd, f3 N, f3 wo, f3 wi)
{
	float bsdfPdf = 0.0f, brdfPdf;
	if (dot(wi, N) <= 0.0f)
		brdfPdf = RT_INV2PI * sd_subsurface(sd) * 0.5f;
	else
	{
		const float F = fresnel_fr(dot(N, wo), sd_eta(sd));
		const f3 halfway = safe_normalize(wi + wo);
		const float cosThetaHalf = fabsf(dot(halfway, N));
		const float pdfHalf = gtr2(cosThetaHalf, sd_roughness(sd)) * cosThetaHalf;
		const float pdfSpec = m_div(0.25f * pdfHalf, fmaxf(1.e-6f, dot(wi, halfway)));
		const float pdfDiff = fabsf(dot(wi, N)) * RT_INVPI * (1.0f - sd_subsurface(sd));
		bsdfPdf = pdfSpec * F;
		brdfPdf = lerp1(pdfDiff, pdfSpec, 0.5f);
	}
	return lerp1(brdfPdf, bsdfPdf, sd_transmission(sd));
}
// disney.h:104-185
RT_FN f3 bsdf_eval(const Shading &sd, f3 N, f3 wo, f3 wi, float t, bool backfacing)
{
	const float NDotL = dot(N, wi);
	const float NDotV = dot(N, wo);
	const f3 H = normalize(wi + wo);
	const float NDotH = dot(N, H);
	const float LDotH = dot(wi, H);
	const f3 Cdlin = sd.color;
	const float Cdlum = .3f * Cdlin.x + .6f * Cdlin.y + .1f * Cdlin.z;
	const f3 Ctint = Cdlum > 0.0f ? Cdlin * m_rcp(Cdlum) : mk3(1, 1, 1);
	const float METALLIC = sd_metallic(sd), TRANSMISSION = sd_transmission(sd), SUBSURFACE = sd_subsurface(sd);
	const float ROUGHNESS = sd_roughness(sd), ETA = sd_eta(sd);
	const f3 Cspec0 = lerp3(lerp3(mk3(1, 1, 1), Ctint, sd_spectint(sd)) * (sd_specular(sd) * .08f), Cdlin, METALLIC);
	f3 bsdf = mk3(0, 0, 0), brdf = mk3(0, 0, 0);
	if (TRANSMISSION > 0.0f)
	{
		if (NDotL <= 0)
		{
			const float F = fresnel_fr(NDotV, ETA);
			const float s = m_div(1.0f - F, fabsf(NDotL)) * (1.0f - METALLIC) * TRANSMISSION;
			bsdf = mk3(s, s, s);
		}
		else
		{
			const float a = ROUGHNESS;
			const float Ds = gtr2(NDotH, a);
			const float FH = fresnel_fr(LDotH, ETA);
			const f3 Fs = lerp3(Cspec0, mk3(1, 1, 1), FH);
			const float Gs = smith_ggx(NDotV, a) * smith_ggx(NDotL, a);
			bsdf = Fs * (Gs * Ds);
		}
	}
	if (TRANSMISSION < 1.0f)
	{
		if (NDotL <= 0)
		{
			if (SUBSURFACE > 0.0f)
			{
				const f3 s = mk3(m_sqrtf(sd.color.x), m_sqrtf(sd.color.y), m_sqrtf(sd.color.z));
				const float FL = schlick_fresnel(fabsf(NDotL)), FV = schlick_fresnel(NDotV);
				const float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
				brdf = (((s * RT_INVPI) * SUBSURFACE) * Fd) * (1.0f - METALLIC);
			}
		}
		else
		{
			const float a = ROUGHNESS;
			const float Ds = gtr2(NDotH, a);
			const float FH = schlick_fresnel(LDotH);
			const f3 Fs = lerp3(Cspec0, mk3(1, 1, 1), FH);
			const float Gs = smith_ggx(NDotV, a) * smith_ggx(NDotL, a);
			const float FL = schlick_fresnel(NDotL), FV = schlick_fresnel(NDotV);
			const float Fd90 = 0.5f + 2.0f * LDotH * LDotH * a;
			const float Fd = lerp1(1.0f, Fd90, FL) * lerp1(1.0f, Fd90, FV);
			const f3 diff = ((Cdlin * (RT_INVPI * Fd)) * (1.0f - METALLIC)) * (1.0f - SUBSURFACE);
			const f3 spec = (Fs * Gs) * Ds;
			float cc = 0.0f; // CLEARCOAT * Gr * Fc * Dr: all factors are finite, so a zero clearcoat contributes exactly 0
			if (sd_clearcoat(sd) > 0.0f)
			{
				const float Dr = gtr1(NDotH, lerp1(.1f, .001f, sd_clearcoatgloss(sd)));
				const float Fc = lerp1(.04f, 1.0f, FH);
				const float Gr = smith_ggx(NDotL, .25f) * smith_ggx(NDotV, .25f);
				cc = sd_clearcoat(sd) * Gr * Fc * Dr;
			}
			brdf = (diff + spec) + mk3(cc, cc, cc);
		}
	}
	const f3 fin = lerp3(brdf, bsdf, TRANSMISSION);
	if (backfacing)
		return fin * mk3(m_expf(-sd.absorption.x * t), m_expf(-sd.absorption.y * t), m_expf(-sd.absorption.z * t));
	return fin;
}
// sin / cos of 2*pi*frac.  On the GPU v_sin_f32 / v_cos_f32 take their argument in turns — one instruction each instead
// of a range-reduced polynomial; |error| ~1e-6, well inside the path tracer's tolerance.
RT_FN void sincos_turns(float frac, float &s, float &c)
{
#if defined(RT_STRICT_MATH)
	strict::sincos_turns(frac, s, c);
#elif defined(__HIP_DEVICE_COMPILE__)
	s = __builtin_amdgcn_sinf(frac), c = __builtin_amdgcn_cosf(frac);
#else
	s = m_sinf(frac * RT_TWOPI), c = m_cosf(frac * RT_TWOPI);
#endif
}
RT_FN f3 reflect_dir(f3 I, f3 N) { return I - N * (dot(N, I) * 2.0f); }
RT_FN f3 diffuse_reflection_uniform(float r0, float r1)
{
	const float term2 = m_sqrtf(1.0f - r1 * r1);
	float sn, cs;
	sincos_turns(r0, sn, cs);
	return mk3(cs * term2, sn * term2, r1);
}
RT_FN f3 diffuse_reflection_cos_weighted(float r0, float r1)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_STRICT_MATH)
	const float term2 = m_sqrtf(1.0f - r1); // (1 - r1 is exact above one half; below it the float and the double form differ in the last place)
#else
	const float term2 = (float)sqrt(1.0 - (double)r1); // tools.h:113 computes this term in double
#endif
	float sn, cs;
	sincos_turns(r0, sn, cs);
	return normalize(mk3(cs * term2, sn * term2, m_sqrtf(r1)));
}
RT_FN f3 ggx_halfway(f3 T, f3 B, f3 N, f3 wo, float rough, float r1, float r2)
{
	const float cosThetaHalf = m_sqrtf(m_div(1.0f - r2, 1.0f + (sqr(rough) - 1.0f) * r2));
	const float sinThetaHalf = m_sqrtf(fmaxf(0.0f, 1.0f - sqr(cosThetaHalf)));
	float sinPhiHalf, cosPhiHalf;
	sincos_turns(r1, sinPhiHalf, cosPhiHalf);
	f3 halfway = (T * (sinThetaHalf * cosPhiHalf) + B * (sinThetaHalf * sinPhiHalf)) + N * cosThetaHalf;
	if (dot(halfway, wo) <= 0.0f)
		halfway = halfway * -1.0f;
	return halfway;
}
// disney.h:188-262; pdf keeps its incoming value (0) on the Fresnel-reflection branch like the reference
RT_FN void bsdf_sample(const Shading &sd, f3 T, f3 B, f3 N, f3 wo, f3 &wi, float &pdf, float r3, float r4)
{
	const float transmission = sd_transmission(sd);
	const float ROUGHNESS = sd_roughness(sd);
	if (r3 < transmission)
	{
		const float F = fresnel_fr(dot(N, wo), sd_eta(sd));
		if (r4 < F)
		{
			const float r1 = m_div(r3, transmission);
			const float r2 = m_div(r4, F);
			wi = reflect_dir(wo * -1.0f, ggx_halfway(T, B, N, wo, ROUGHNESS, r1, r2));
		}
		else
		{
			pdf = 0;
			if (refract_dir(wo, N, sd_eta(sd), wi))
				pdf = (1.0f - F) * transmission;
		}
		return;
	}
	const float r1 = m_div(r3 - transmission, 1 - transmission);
	if (r4 < 0.5f)
	{
		const float r2 = r4 * 2;
		const float subsurface = sd_subsurface(sd);
		f3 d;
		if (r2 < subsurface)
		{
			const float r5 = m_div(r2, subsurface);
			d = diffuse_reflection_uniform(r1, r5);
			d.z *= -1.0f;
		}
		else
		{
			const float r5 = m_div(r2 - subsurface, 1.0f - subsurface);
			d = diffuse_reflection_cos_weighted(r1, r5);
		}
		wi = (T * d.x + B * d.y) + N * d.z;
	}
	else
	{
		const float r2 = (r4 - 0.5f) * 2.0f;
		wi = reflect_dir(wo * -1.0f, ggx_halfway(T, B, N, wo, ROUGHNESS, r1, r2));
	}
	pdf = bsdf_pdf(sd, N, wo, wi);
}

// bsdf/tools.h
RT_FN uint32_t pack_normal(f3 N)
{
	const float f = m_div(65535.0f, fmaxf(m_sqrtf(8.0f * N.z + 8.0f), 0.0001f));
	return f2u_sat(N.x * f + 32767.0f) + (f2u_sat(N.y * f + 32767.0f) << 16);
}
RT_FN f3 unpack_normal(uint32_t p)
{
	float nx = (float)(p & 65535u) * (2.0f / 65535.0f), ny = (float)(p >> 16) * (2.0f / 65535.0f);
	nx += -1.0f, ny += -1.0f;
	const float nz0 = 1.0f, nw = -1.0f;
	float l = nx * -nx + ny * -ny + nz0 * -nw;
	const float nz = l;
	l = m_sqrtf(l);
	nx *= l, ny *= l;
	return mk3(nx * 2.0f, ny * 2.0f, nz * 2.0f - 1.0f);
}
RT_FN float survival_probability(f3 d) { return fminf(1.0f, fmaxf(fmaxf(d.x, d.y), d.z)); }
RT_FN f3 clamp_intensity(f3 v, float clampValue)
{
	const float m = fmaxf(v.x, fmaxf(v.y, v.z));
	if (m > clampValue)
		return v * m_div(clampValue, m);
	return v;
}
RT_FN void create_tangent_space(f3 N, f3 &T, f3 &B)
{
	const float s = signf(N.z);
	const float a = -m_rcp(s + N.z);
	const float b = N.x * N.y * a;
	T = mk3(1.0f + s * N.x * N.x * a, s * b, -s * N.x);
	B = mk3(b, s + N.y * N.y * a, -N.y);
}

// ---------------------------------------------------------------------------------------------------------------
// light sampling: CUDART/src/lights.h
// ---------------------------------------------------------------------------------------------------------------
RT_FN float pot_area(const SceneView &sc, uint32_t idx, f3 O, f3 N, f3 I, f3 bary)
{
	const AreaLight l = uniform_record(sc.area, idx);
	f3 L = I;
	if (bary.x >= 0)
		L = (ld3(l.vertex0) * bary.x + ld3(l.vertex1) * bary.y) + ld3(l.vertex2) * bary.z;
	L = L - O;
	const float att = m_rcp(dot(L, L));
	L = normalize(L);
	const float LNdotL = fmaxf(0.0f, -dot(ld3(l.normal), L));
	const float NdotL = fmaxf(0.0f, dot(N, L));
	return l.energy * LNdotL * NdotL * att;
}
RT_FN float pot_point(const SceneView &sc, uint32_t idx, f3 I, f3 N)
{
	const PointLight l = uniform_record(sc.point, idx);
	const f3 L = ld3(l.position) - I;
	const float NdotL = fmaxf(0.0f, dot(N, L));
	const float att = m_rcp(dot(L, L));
	return l.energy * NdotL * att;
}
RT_FN float pot_spot(const SceneView &sc, uint32_t idx, f3 I, f3 N)
{
	const SpotLight l = uniform_record(sc.spot, idx);
	f3 L = ld3(l.position) - I;
	const float att = m_rcp(dot(L, L));
	L = normalize(L);
	const float d = m_div(fmaxf(0.0f, -dot(L, ld3(l.direction))) - l.cosOuter, l.cosInner - l.cosOuter);
	const float NdotL = fmaxf(0.0f, dot(N, L));
	const float LNdotL = fmaxf(0.0f, fminf(1.0f, d));
	return l.energy * LNdotL * NdotL * att;
}
RT_FN float pot_dir(const SceneView &sc, uint32_t idx, f3 N)
{
	const DirectionalLight l = uniform_record(sc.dir, idx);
	return l.energy * fmaxf(0.0f, -dot(ld3(l.direction), N));
}
RT_FN uint32_t total_lights(const SceneView &sc) { return sc.n_area + sc.n_point + sc.n_spot + sc.n_dir; }
RT_FN float pot_any(const SceneView &sc, uint32_t k, f3 I, f3 N, f3 bary)
{
	if (k < sc.n_area)
		return pot_area(sc, k, I, N, mk3(0, 0, 0), bary);
	k -= sc.n_area;
	if (k < sc.n_point)
		return pot_point(sc, k, I, N);
	k -= sc.n_point;
	if (k < sc.n_spot)
		return pot_spot(sc, k, I, N);
	k -= sc.n_spot;
	return pot_dir(sc, k, N);
}
// lights.h:83-116
RT_FN float light_pick_prob(const SceneView &sc, int idx, f3 O, f3 N, f3 I)
{
	float sum = 0, mine = 0;
	for (uint32_t i = 0; i < sc.n_area; i++)
	{
		const float p = pot_area(sc, i, O, N, I, mk3(-1, -1, -1));
		if ((int)i == idx)
			mine = p;
		sum += p;
	}
	for (uint32_t i = 0; i < sc.n_point; i++)
		sum += pot_point(sc, i, O, N);
	for (uint32_t i = 0; i < sc.n_spot; i++)
		sum += pot_spot(sc, i, O, N);
	for (uint32_t i = 0; i < sc.n_dir; i++)
		sum += pot_dir(sc, i, N);
	if (sum <= 0)
		return 0;
	return m_div(mine, sum);
}
// lights.h:119-157: sixteen rounds of "pick one of the four sub-triangles" (two bits of r0 each, most significant first),
// then the centroid.  The reference walks the three corners through a 16-iteration loop of four-way branches — ~600
// instructions per call on a wave whose lanes all take different branches, a quarter of the shade kernel.  Every quantity in
// that loop is a multiple of 2^-16 in [0, 1] (midpoints of such numbers, halved: exact in float), so the result can be
// computed in integers, bit for bit the same.  In the frame of corner A with edges u = B - A, v = C - A one round halves
// both edges and moves A by (alpha u + beta v) / 2, where d = 1: (0, 0), d = 2: (1, 0), d = 3: (0, 1), d = 0: (1, 1) and the
// frame flips sign (the middle sub-triangle is the parent turned upside down).  Hence
//     A16 = A0 + (U u0 + V v0) / 2^16,   U = sum_i s_i alpha_i 2^(15 - i),  V = sum_i s_i beta_i 2^(15 - i),
// s_i = (-1)^(number of rounds j < i with d_j = 0), A0 = (1, 0), u0 = (-1, 1), v0 = (-1, 0), and the sum of the three final
// corners is 3 A16 + s_16 (u0 + v0) / 2^16.  The one rounding of the reference — (Ax + Bx + Cx) * 0.3333333f on an exact
// sum — is the one rounding here.  (tests/test_oracle_kat.py compares with the oracle's loop form, bit for bit.)
RT_FN uint32_t even_bits(uint32_t x) // the 16 bits at even positions, packed
{
	x &= 0x55555555u;
	x = (x ^ (x >> 1)) & 0x33333333u;
	x = (x ^ (x >> 2)) & 0x0F0F0F0Fu;
	x = (x ^ (x >> 4)) & 0x00FF00FFu;
	x = (x ^ (x >> 8)) & 0x0000FFFFu;
	return x;
}
RT_FN f3 random_barycentrics(float r0)
{
#if defined(__clang__)
#pragma clang fp contract(off) // 1 - rx - ry with rx, ry rounded products, not fma(-sum, 1/3, 1): device == emulation == oracle
#endif
	const uint32_t uf = f2u_sat(r0 * 4294967295.0f);
	// bit 15 - i of hi / lo = high / low bit of round i's digit d_i
	const uint32_t hi = even_bits(uf >> 1), lo = even_bits(uf);
	const uint32_t zero = ~(hi | lo) & 0xFFFFu;		// d == 0
	const uint32_t alpha = ~lo & 0xFFFFu;			// d == 2 or d == 0
	const uint32_t beta = ~(hi ^ lo) & 0xFFFFu;		// d == 3 or d == 0
	// neg bit (15 - i) = parity of the zeros among rounds 0 .. i - 1 (an exclusive prefix XOR from the top bit down)
	uint32_t neg = zero >> 1;
	neg ^= neg >> 1, neg ^= neg >> 2, neg ^= neg >> 4, neg ^= neg >> 8;
	const int U = (int)(alpha & ~neg) - (int)(alpha & neg), V = (int)(beta & ~neg) - (int)(beta & neg);
#if defined(__HIP_DEVICE_COMPILE__)
	const int s16 = (__popc(zero) & 1) ? -1 : 1;
#else
	const int s16 = (__builtin_popcount(zero) & 1) ? -1 : 1;
#endif
	// sum of the corners, scaled by 2^16: x: 3 (65536 - U - V) - 2 s16, y: 3 U + s16  (both < 2^18: exact as floats)
	const float sx = (float)(3 * (65536 - U - V) - 2 * s16) * (1.0f / 65536.0f);
	const float sy = (float)(3 * U + s16) * (1.0f / 65536.0f);
	const float rx = sx * 0.3333333f, ry = sy * 0.3333333f;
	return mk3(rx, ry, 1.0f - rx - ry);
}
// lights.h:159-265 with importance sampling over the potential contribution of every light.  The potentials are
// recomputed in the selection pass instead of being kept in a MAX_IS_LIGHTS array, so any light count is valid.
constexpr uint32_t POT_CACHE = 16; // potentials of the first 16 lights are kept (in LDS on the GPU) between the passes
constexpr uint32_t POT_SLOTS = POT_CACHE + 1; // + one slot that takes the writes of every further light (a store without a branch)
#if defined(RT_DEVICE_BUILD)
constexpr int POT_STRIDE = 256;
#else
constexpr int POT_STRIDE = 1;
#endif
RT_FN f3 random_point_on_light(const SceneView &sc, float r0, float r1, f3 I, f3 N, float &pickProb, float &lightPdf,
							   f3 &lightColor, uint32_t &light, float *pot_cache RT_CLK_PARAM)
{
	light = 0u;
	const uint32_t lights = total_lights(sc);
	const f3 bary = random_barycentrics(r0);
	RT_TICK(9);
	// (one loop per kind of light, in the order of pot_any(): a loop without the four-way branch, whose records are read by
	// scalar loads and which the compiler can unroll)
	float sum = 0;
	uint32_t kk = 0;
#define RT_POT_LOOP(COUNT, EXPR)                      \
	for (uint32_t i = 0; i < (COUNT); i++, kk++)      \
	{                                                 \
		const float pk = EXPR;                        \
		pot_cache[(kk < POT_CACHE ? kk : POT_CACHE) * POT_STRIDE] = pk; \
		sum += pk;                                    \
	}
	RT_POT_LOOP(sc.n_area, pot_area(sc, i, I, N, mk3(0, 0, 0), bary))
	RT_POT_LOOP(sc.n_point, pot_point(sc, i, I, N))
	RT_POT_LOOP(sc.n_spot, pot_spot(sc, i, I, N))
	RT_POT_LOOP(sc.n_dir, pot_dir(sc, i, N))
#undef RT_POT_LOOP
	if (sum <= 0)
	{
		lightPdf = 0;
		return mk3(1, 1, 1);
	}
	RT_TICK(10);
	r1 *= sum;
	float total = 0, chosen = 0, first = 0;
	uint32_t li = 0;
	for (uint32_t k = 0; k < lights; k++)
	{
		const float p = k < POT_CACHE ? pot_cache[k * POT_STRIDE] : pot_any(sc, k, I, N, bary);
		if (k == 0)
			first = p;
		total += p;
		if (total >= r1)
		{
			li = k, chosen = p;
			break;
		}
		if (k == lights - 1)
			li = 0, chosen = first;
	}
	pickProb = m_div(chosen, sum);
	light = li;
	RT_TICK(11);
	if (li < sc.n_area)
	{
		const AreaLight &l = sc.area[li];
		lightColor = ld3(l.radiance);
		const f3 LN = ld3(l.normal);
		const f3 P = (ld3(l.vertex0) * bary.x + ld3(l.vertex1) * bary.y) + ld3(l.vertex2) * bary.z;
		f3 L = I - P;
		const float sqDist = dot(L, L);
		L = normalize(L);
		const float LNdotL = dot(L, LN);
		const float reciSolidAngle = m_div(sqDist, l.area * LNdotL);
		const float energy = length(ld3(l.radiance)); // DeviceAreaLight::getEnergy, device_structs.h:115
		lightPdf = (LNdotL > 0 && dot(L, N) < 0) ? (reciSolidAngle * m_rcp(energy)) : 0;
		return P;
	}
	li -= sc.n_area;
	if (li < sc.n_point)
	{
		const PointLight &l = sc.point[li];
		const f3 pos = ld3(l.position);
		lightColor = ld3(l.radiance);
		const f3 L = I - pos;
		const float sqDist = dot(L, L);
		lightPdf = dot(L, N) < 0 ? m_div(sqDist, l.energy) : 0;
		return pos;
	}
	li -= sc.n_point;
	if (li < sc.n_spot)
	{
		const SpotLight &l = sc.spot[li];
		const f3 P = ld3(l.position);
		f3 L = I - P;
		const float sqDist = dot(L, L);
		L = normalize(L);
		const float d = m_div(fmaxf(0.0f, dot(L, ld3(l.direction)) - l.cosOuter), l.cosInner - l.cosOuter);
		const float LNdotL = fminf(1.0f, d);
		lightPdf = (LNdotL > 0 && dot(L, N) < 0) ? m_div(sqDist, LNdotL * l.energy) : 0;
		lightColor = ld3(l.radiance);
		return P;
	}
	li -= sc.n_spot;
	const DirectionalLight &l = sc.dir[li];
	const f3 L = ld3(l.direction);
	lightColor = ld3(l.radiance);
	const float NdotL = dot(L, N);
	lightPdf = NdotL < 0 ? m_rcp(l.energy) : 0;
	return I - L * 1000.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// PATH-TRACING shade step for one path vertex: CUDART/src/Kernels.cu:571-794
// ---------------------------------------------------------------------------------------------------------------
struct PathIn
{
	f3 O, D, T;
	float bsdfPdf;
	uint32_t slot, flags, packedN;
	uint32_t depth; // pathLength
};
struct ShadeOut
{
	f3 radiance; // to add to the path's slot
	bool emit_shadow, emit_ext;
	f4 eo, ed, et; // extension ray: origin|slot<<1|flags, dir|packedN, throughput|pdf
};

// the surface at a hit: shading record, material, barycentric weights, geometric / shading normal in world space
struct Surface
{
	float area, lod;
	int ltri;
	uint32_t shade_idx; // index of the triangle's records in tri_shade / tri_uv
	const MaterialRec *mat;
	uint32_t mflags;
	float bw0, bw1, bw2;
	f3 N, iN;
};
RT_FN void pt_surface(const SceneView &sc, const Hit &h, Surface &sf)
{
	const Instance &inst = sc.instances[h.inst];
	sf.shade_idx = inst.shade_base + (uint32_t)h.prim;
	const TriShade &ts = sc.tri_shade[sf.shade_idx];
	const f4 n0 = ts.n0, n1 = ts.n1, n2 = ts.n2, ex = ts.ex;
	sf.area = ex.x, sf.lod = ex.y, sf.ltri = (int)fbits(ex.z);
	sf.mat = &sc.materials[fbits(ex.w)];
	sf.mflags = sf.mat->flags;
	// getShadingData.h:100-217 (its u,v,w weight vertex 0,1,2)
	sf.bw0 = 1.0f - h.u - h.v, sf.bw1 = h.u, sf.bw2 = h.v;
	f3 N = mk3(n0.w, n1.w, n2.w), iN = N;
	if (mat_flag(sf.mflags, MF_SMOOTH_NORMALS))
		iN = normalize((xyz(n0) * sf.bw0 + xyz(n1) * sf.bw1) + xyz(n2) * sf.bw2);
	sf.N = normalize(mul_normal(inst, N));
	sf.iN = normalize(mul_normal(inst, iN));
}
RT_FN bool pt_has_textures(const SceneView &sc, const Surface &sf)
{
	return mat_flag(sf.mflags, MF_DIFFUSE_MAP) && sf.mat->map[0].addr < sc.texture_count;
}
// the texture layers of a textured hit: color = the material's colour on entry; Tg / Bt = the tangent frame of the unperturbed
// shading normal (tools.h:204-211)
RT_FN void pt_textures(const SceneView &sc, const CamView &cam, f3 D, float t, const Surface &sf, f3 &color, f3 &iN, bool &alpha_skip)
{
	const MaterialRec &mat = *sf.mat;
	const uint32_t mflags = sf.mflags;
	const TriUV &uv = sc.tri_uv[sf.shade_idx];
	const f4 tu4 = uv.tu, tv4 = uv.tv;
	const float tu = sf.bw0 * tu4.x + sf.bw1 * tu4.y + sf.bw2 * tu4.z;
	const float tv = sf.bw0 * tv4.x + sf.bw1 * tv4.y + sf.bw2 * tv4.z;
	const float coneWidth = cam.spread_angle * t;
	const float lambda = sf.lod + m_log2f(coneWidth * m_rcp(fabsf(dot(D * -1.0f, sf.N))));
	// map slots: 0-2 diffuse layers, 3-5 normal-map layers (structs.h:98-115)
#define RT_LAYER(K) \
	fetch_trilinear(sc, sc.textures[mat.map[K].addr], lambda,                                          \
					half_to_float(mat.map[K].uscale) * (half_to_float(mat.map[K].uoffs) + tu),         \
					half_to_float(mat.map[K].vscale) * (half_to_float(mat.map[K].voffs) + tv), mat.map[K].width, \
					mat.map[K].height)
#define RT_NORMAL_LAYER(K) \
	((xyz(fetch_texel(sc, sc.textures[mat.map[K].addr],                                                \
					  half_to_float(mat.map[K].uscale) * (half_to_float(mat.map[K].uoffs) + tu),       \
					  half_to_float(mat.map[K].vscale) * (half_to_float(mat.map[K].voffs) + tv), 0u,   \
					  mat.map[K].width > 0 ? mat.map[K].width : 1, mat.map[K].height > 0 ? mat.map[K].height : 1)) - \
	  mk3(0.5f, 0.5f, 0.5f)) *                                                                         \
	 2.0f)
	const f4 texel = RT_LAYER(0);
	if (mat_flag(mflags, MF_ALPHA) && texel.w < 0.5f)
		alpha_skip = true; // getShadingData.h:145-149: nothing else of the surface is evaluated
	else
	{
		color = color * xyz(texel);
		// second and third layers are additive (getShadingData.h:153-166)
		if (mat_flag(mflags, MF_2ND_DIFFUSE_MAP) && mat.map[1].addr < sc.texture_count)
			color = color + xyz(RT_LAYER(1));
		if (mat_flag(mflags, MF_3RD_DIFFUSE_MAP) && mat.map[2].addr < sc.texture_count)
			color = color + xyz(RT_LAYER(2));
		// normal mapping, level 0 only (getShadingData.h:169-200); the third layer reads the descriptor of the
		// second one there (:189-196) — kept.  The tangent frame stays the one of the unperturbed normal.
		if (mat_flag(mflags, MF_NORMAL_MAP) && mat.map[3].addr < sc.texture_count)
		{
			f3 sn = RT_NORMAL_LAYER(3);
			if (mat_flag(mflags, MF_2ND_NORMAL_MAP) && mat.map[4].addr < sc.texture_count)
				sn = sn + RT_NORMAL_LAYER(4);
			if (mat_flag(mflags, MF_3RD_NORMAL_MAP) && mat.map[4].addr < sc.texture_count)
				sn = sn + RT_NORMAL_LAYER(4);
			sn = normalize(sn);
			f3 Tg, Bt;
			create_tangent_space(sf.iN, Tg, Bt);
			iN = normalize((Tg * sn.x + Bt * sn.y) + iN * sn.z); // tangentToWorld, tools.h:214
		}
		// getShadingData.h:150 and :206 both multiply by the texel
		color = color * xyz(texel);
	}
#undef RT_LAYER
#undef RT_NORMAL_LAYER
}
// One path vertex, in PHASES every lane of the calling wave passes through together (round 6: until then the function returned from
// half a dozen places and handed both of its rays back at the end — the shadow ray's twelve registers stayed live across the BSDF
// sampling, the tangent frame and the absorption across the light sampling):
//   1. surface: sky for a miss; shading record, material, texture layers; alpha pass-through and emitters end here;
//   2. next-event estimation (Kernels.cu:702-755) -> the shadow ray, handed to `sink.shadow()` AT ONCE — the kernel allocates its
//      queue slot and stores it there and then (a wave-wide ballot: every lane calls it, with emit = false when it has no ray);
//   3. BSDF sampling (Kernels.cu:758-793) -> the extension ray in `out`.  What only this phase needs is produced here: the tangent
//      frame (from the unperturbed shading normal: flip * flip = 1 takes the plain kernel back to it exactly), the absorption
//      (re-read from the material for a back-facing hit of a transmissive one), SafeOrigin (the same point as the shadow ray's).
// TEX = false: the scene has no material with a texture or normal map (the host knows: rfwhip_set_materials) — the texture
// layers, their descriptors and the level-of-detail arithmetic are compiled out, which frees a fifth of the registers.
template <bool TEX, class Sink>
RT_FN void pt_shade(const SceneView &sc, const CamView &cam, const FrameView &fr, uint32_t max_depth, bool active, const PathIn &in,
					const Hit &h, ShadeOut &out, float *pot_cache, Sink &sink RT_CLK_PARAM)
{
	out.radiance = mk3(0, 0, 0);
	out.emit_shadow = false, out.emit_ext = false;
	const f3 D = in.D;
	f3 T = in.T;
	RT_TICK(1);
	// state of a path that goes on from a surface (phases 2 and 3)
	bool regular = false;
	f3 I = mk3(0, 0, 0), N = mk3(0, 0, 1), iN = mk3(0, 0, 1), iN0 = mk3(0, 0, 1);
	Shading sd;
	sd.color = mk3(0, 0, 0), sd.absorption = mk3(0, 0, 0), sd.p0 = 0, sd.p1 = 0, sd.p2 = 0;
	const MaterialRec *matp = nullptr;
	uint32_t flags = in.flags, seed = 0;
	float flip = 1.0f;
	f4 so = mk4(0, 0, 0, 0), sdir = mk4(0, 0, 0, 0), se = mk4(0, 0, 0, 0);
	bool emit_shadow = false;
	if (!active)
	{
		// (a lane without a path: it only takes part in the wave-wide steps)
	}
	else if (h.prim < 0)
	{
		const f3 contribution = (T * m_rcp(in.bsdfPdf)) * pt_sky(sc, D);
		if (!any_nan(contribution))
			out.radiance = clamp_intensity(contribution, cam.clamp_value);
		RT_TICK(2);
	}
	else
	{
		I = in.O + D * h.t;
		Surface sf;
		pt_surface(sc, h, sf);
		RT_TICK(3);
		const MaterialRec &mat = *sf.mat;
		matp = sf.mat;
		sd.color = material_color(mat);
		sd.p0 = mat.parameters[0], sd.p1 = mat.parameters[1], sd.p2 = mat.parameters[2];
		N = sf.N, iN = sf.iN, iN0 = sf.iN;
		bool alpha_skip = false;
		if (TEX && pt_has_textures(sc, sf))
			pt_textures(sc, cam, D, h.t, sf, sd.color, iN, alpha_skip);
		// the path's pixel and sample: the key of its random numbers (and the probe pixel of the primary wave)
		const PixelRef pr = slot_to_pixel(fr, in.slot);
		const uint32_t pixel = pr.y * fr.W + pr.x, sampleIdx = fr.sample_base + pr.sample;
		if (in.depth == 0 && pr.sample == 0 && pixel == fr.probe_pixel)
			sink.probe(h);
		if (alpha_skip)
		{
			// alpha pass-through (Kernels.cu:633-647): the path continues behind the surface, state untouched
			if (in.depth < max_depth && !any_nan(T))
			{
				const f3 eo = I + D * 1e-5f;
				out.emit_ext = true;
				out.eo = mk4(eo.x, eo.y, eo.z, ubits((in.slot << 1) | (in.flags & 1u)));
				out.ed = mk4(D.x, D.y, D.z, ubits(in.packedN));
				out.et = mk4(T.x, T.y, T.z, in.bsdfPdf);
			}
		}
		else if (sd.color.x > 1.0f || sd.color.y > 1.0f || sd.color.z > 1.0f)
		{
			// emissive surface: Kernels.cu:650-692
			const float DdotNL = -dot(D, N);
			f3 contribution = mk3(0, 0, 0);
			bool drop = false;
			if (DdotNL > 0)
			{
				if (in.depth == 0)
					contribution = sd.color;
				else if (in.flags & 1u)
					contribution = (T * sd.color) * m_rcp(in.bsdfPdf);
				else
				{
					const f3 lastN = unpack_normal(in.packedN);
					const float lightPdf = m_div(h.t * h.t, -dot(D, N) * sf.area); // lights.h:78-81
					const int ltri = sf.ltri;
					// the reference reads the material id as light index (device_structs.h:37,40); lightTriIdx is meant
					const float pickProb =
						(ltri >= 0 && (uint32_t)ltri < sc.n_area) ? light_pick_prob(sc, ltri, in.O, lastN, I) : 0.0f;
					if ((in.bsdfPdf + lightPdf * pickProb) <= 0)
						drop = true;
					else
						contribution = (T * sd.color) * m_rcp(in.bsdfPdf + lightPdf * pickProb);
				}
			}
			if (!drop)
			{
				if (any_nan(contribution))
					contribution = mk3(0, 0, 0);
				out.radiance = clamp_intensity(contribution, cam.clamp_value);
			}
		}
		else
		{
			regular = true;
			RT_TICK(4);
			if (sd_roughness(sd) < 0.01f)
				flags |= 1u;
			else
				flags &= ~1u;
			seed = wang_hash(pixel * 16789u + sampleIdx * 1791u + in.depth * 720898027u);
			flip = (dot(D, N) > 0) ? -1.0f : 1.0f;
			N = N * flip;
			iN = iN * flip;
			T = T * m_rcp(in.bsdfPdf);
			// next-event estimation: Kernels.cu:702-755.  The connections of a shade call are traced by the NEXT iteration of the
			// reference's host loop (CUDART/src/Context.cpp:109-120), so those of the last call (depth == max_depth) never are:
			// they are not even computed here (the two random numbers they would consume are followed by no other draw).
			if ((flags & 1u) == 0 && total_lights(sc) > 0 && in.depth < max_depth)
			{
				const f3 wo = D * -1.0f;
				f3 lightColor = mk3(0, 0, 0);
				float pickProb = 0, lightPdf = 0;
				float q0, q1;
				if (cam.blue_noise && sampleIdx < 256u) // BLUENOISE (Kernels.cu:712-719): the hash seed is not advanced
				{
					q0 = blue_noise_sample(cam.blue_noise, (int)pr.x, (int)pr.y, (int)sampleIdx, 4);
					q1 = blue_noise_sample(cam.blue_noise, (int)pr.x, (int)pr.y, (int)sampleIdx, 5);
				}
				else
					q0 = random_float(seed), q1 = random_float(seed);
				uint32_t light = 0u;
				f3 L = random_point_on_light(sc, q0, q1, I, iN, pickProb, lightPdf, lightColor, light, pot_cache RT_CLK_ARG) - I;
				RT_TICK(12);
				const float dist = length(L);
				L = L * m_rcp(dist);
				const float NdotL = dot(L, iN);
				if (NdotL > 0 && lightPdf > 0)
				{
					const f3 bs = bsdf_eval(sd, iN, wo, L, 0.0f, false);
					const float shadowPdf = bsdf_pdf(sd, iN, wo, L);
					if (shadowPdf > 0)
					{
						f3 contribution = ((T * bs) * lightColor) * m_div(NdotL, shadowPdf + lightPdf * pickProb);
						contribution = clamp_intensity(contribution, cam.clamp_value);
						if (!any_nan(contribution))
						{
							const f3 o = I + N * 1e-5f; // SafeOrigin, tools.h:119-123
							emit_shadow = true;
							// (depth 0, fr.shadow_bins: the light's bin rides in the top bits of the slot word — what the packet form of the
							// connection wave sorts a run's rays by)
							const uint32_t last_bin = (1u << fr.shadow_bins) - 1u;
							const uint32_t bin = (fr.shadow_bins && in.depth == 0u) ? (light < last_bin ? light : last_bin) << shadow_slot_bits(fr.shadow_bins) : 0u;
							so = mk4(o.x, o.y, o.z, ubits(in.slot | bin));
							sdir = mk4(L.x, L.y, L.z, dist - 2.0f * 1e-5f);
							se = mk4(contribution.x, contribution.y, contribution.z, 0.0f);
						}
					}
				}
			}
			RT_TICK(5);
		}
	}
	// ---- every lane: the shadow ray leaves the registers here ----
	out.emit_shadow = emit_shadow;
	sink.shadow(emit_shadow, so, sdir, se);
	if (!regular || in.depth >= max_depth)
		return;
	// ---- phase 3: BSDF sampling ----
	{
		const f3 wo = D * -1.0f;
		f3 Tg, Bt;
		create_tangent_space(TEX ? iN0 : iN * flip, Tg, Bt);
		if (flip < 0) // (the only reader of the absorption: bsdf_eval's back-facing branch)
			sd.absorption = mk3(half_to_float(matp->transmittance[0]), half_to_float(matp->transmittance[1]),
								half_to_float(matp->transmittance[2]));
		f3 R = mk3(0, 0, 1);
		float newPdf = 0.0f;
		const float q3 = random_float(seed), q4 = random_float(seed);
		bsdf_sample(sd, Tg, Bt, iN, wo, R, newPdf, q3, q4);
		const f3 bs = bsdf_eval(sd, iN, wo, R, h.t, flip < 0);
		{
			// throughput * 1.0f / SurvivalProbability(throughput) * bsdf * abs(dot(iN, R))   (Kernels.cu:783)
			const float surv = survival_probability(T);
			T = (m_div3(T, surv) * bs) * fabsf(dot(iN, R));
		}
		if (newPdf < 1e-6f || (newPdf != newPdf) || T.x < 0.0f || T.y < 0.0f || T.z < 0.0f)
			return;
		const f3 eo = I + N * 1e-5f;
		out.emit_ext = true;
		out.eo = mk4(eo.x, eo.y, eo.z, ubits((in.slot << 1) | (flags & 1u)));
		out.ed = mk4(R.x, R.y, R.z, ubits(pack_normal(iN)));
		out.et = mk4(T.x, T.y, T.z, newPdf);
		RT_TICK(6);
	}
}

} // namespace rt
