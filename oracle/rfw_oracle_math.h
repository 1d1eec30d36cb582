/*
 * rfw_oracle_math.h — CPU ORACLE (test infrastructure, NOT product code).  Parity unpinned, see rfw_oracle.h.
 * Scalar fp32 helpers, the hash RNGs, the Disney BSDF and the light sampling of the reference, restated in C.
 */
#ifndef RFW_ORACLE_MATH_H
#define RFW_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct
{
	float x, y, z;
} v3;

static inline v3 V3(float x, float y, float z)
{
	v3 r = {x, y, z};
	return r;
}
static inline v3 v3p(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
static inline float vlen(v3 a) { return sqrtf(vdot(a, a)); }
/* Fixed-shape arithmetic for the results that are compared bit for bit across implementations (triangle test, pt primary ray):
 * fmaf() where a fused multiply-add is meant, rounded() around a product that is rounded on its own — an empty asm keeps the
 * compiler's fma contraction from deciding otherwise.  The product states the same shapes (csrc/rt_core.h: rounded()). */
static inline float rounded(float x)
{
#if defined(__x86_64__)
	__asm__("" : "+x"(x));
#else
	volatile float y = x;
	x = y;
#endif
	return x;
}
static inline float vdot_r(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, rounded(a.x * b.x))); }
static inline v3 vcross_r(v3 a, v3 b)
{
	return V3(fmaf(a.y, b.z, -rounded(b.y * a.z)), fmaf(a.z, b.x, -rounded(b.z * a.x)), fmaf(a.x, b.y, -rounded(b.x * a.y)));
}
static inline v3 vmadd2_r(v3 p, v3 a, float s, v3 b, float t)
{
	return V3(fmaf(b.x, t, fmaf(a.x, s, p.x)), fmaf(b.y, t, fmaf(a.y, s, p.y)), fmaf(b.z, t, fmaf(a.z, s, p.z)));
}
static inline v3 vnorm_r(v3 a)
{
	const float inv = 1.0f / sqrtf(vdot_r(a, a));
	return V3(rounded(a.x * inv), rounded(a.y * inv), rounded(a.z * inv));
}
/* glm::normalize = v * inversesqrt(dot(v, v)) */
static inline v3 vnorm(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
static inline v3 vlerp(v3 a, v3 b, float t) { return vadd(a, vscale(vsub(b, a), t)); }
static inline float flerp(float a, float b, float t) { return a + t * (b - a); }
static inline float fmaxf3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
static inline int v3_any_nan(v3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
static inline float fclamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

/* float -> uint32 with the saturating semantics of the GPU conversion (C leaves out-of-range undefined) */
static inline uint32_t f2u_sat(float f)
{
	if (!(f > 0.0f))
		return 0u;
	if (f >= 4294967296.0f)
		return 0xFFFFFFFFu;
	return (uint32_t)f;
}

/* column-major 4x4 * (v,w) */
static inline v3 m4_mul(const float *m, v3 v, float w)
{
	return V3(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * w,
			  m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * w);
}
/* column-major 3x3 * v */
static inline v3 m3_mul(const float *m, v3 v)
{
	return V3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
			  m[2] * v.x + m[5] * v.y + m[8] * v.z);
}

/* ---- RNGs -------------------------------------------------------------------------------------------------- */
/* RFW/system/utils/src/rfw/utils/xor128.h:20-27 */
static inline uint32_t xor128_next(uint32_t s[4])
{
	const uint32_t t = s[0] ^ (s[0] << 11);
	s[0] = s[1];
	s[1] = s[2];
	s[2] = s[3];
	s[3] = s[3] ^ (s[3] >> 19) ^ (t ^ (t >> 8));
	return s[3];
}
/* utils/rng.h:14 — can return exactly 1.0f */
static inline float xor128_rand(uint32_t s[4]) { return (float)xor128_next(s) * 2.3283064365387e-10f; }

/* RFW/system/context/rfw/bsdf/tools.h:218-235 */
static inline uint32_t wang_hash(uint32_t s)
{
	s = (s ^ 61u) ^ (s >> 16);
	s *= 9u;
	s = s ^ (s >> 4);
	s *= 0x27d4eb2du;
	s = s ^ (s >> 15);
	return s;
}
static inline uint32_t random_int(uint32_t *s)
{
	*s ^= *s << 13;
	*s ^= *s >> 17;
	*s ^= *s << 5;
	return *s;
}
static inline float random_float(uint32_t *s) { return (float)random_int(s) * 2.3283064365387e-10f; }

/* IEEE binary16 -> binary32 */
static inline float half_to_float(uint16_t h)
{
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1Fu;
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
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

/* ---- bsdf/tools.h ------------------------------------------------------------------------------------------- */
/* tools.h:10-21 (the "#if 1" branch) */
static inline uint32_t pack_normal(v3 N)
{
	const float f = 65535.0f / fmaxf(sqrtf(8.0f * N.z + 8.0f), 0.0001f);
	return f2u_sat(N.x * f + 32767.0f) + (f2u_sat(N.y * f + 32767.0f) << 16);
}
/* tools.h:22-29 */
static inline v3 unpack_normal(uint32_t p)
{
	float nx = (float)(p & 65535u) * (2.0f / 65535.0f), ny = (float)(p >> 16) * (2.0f / 65535.0f);
	nx += -1.0f, ny += -1.0f;
	const float nz0 = 1.0f, nw = -1.0f;
	float l = nx * -nx + ny * -ny + nz0 * -nw;
	const float nz = l;
	l = sqrtf(l);
	nx *= l, ny *= l;
	return V3(nx * 2.0f, ny * 2.0f, nz * 2.0f - 1.0f);
}
/* tools.h:86 */
static inline float survival_probability(v3 d) { return fminf(1.0f, fmaxf(fmaxf(d.x, d.y), d.z)); }
/* tools.h:184-192 */
static inline v3 clamp_intensity(v3 v, float clampValue)
{
	const float m = fmaxf(v.x, fmaxf(v.y, v.z));
	if (m > clampValue)
		return vscale(v, clampValue / m);
	return v;
}
static inline float fsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
/* tools.h:204-211 */
static inline void create_tangent_space(v3 N, v3 *T, v3 *B)
{
	const float s = fsign(N.z);
	const float a = -1.0f / (s + N.z);
	const float b = N.x * N.y * a;
	*T = V3(1.0f + s * N.x * N.x * a, s * b, -s * N.x);
	*B = V3(b, s + N.y * N.y * a, -N.y);
}
/* tools.h:102-108 */
static inline v3 diffuse_reflection_uniform(float r0, float r1)
{
	const float term1 = 6.28318530717958647692f * r0, term2 = sqrtf(1.0f - r1 * r1);
	return V3(cosf(term1) * term2, sinf(term1) * term2, r1);
}
/* tools.h:110-117 (term2 is computed in double in the reference: sqrt(1.0 - r1)) */
static inline v3 diffuse_reflection_cos_weighted(float r0, float r1)
{
	const float term1 = 6.28318530717958647692f * r0;
	const float term2 = (float)sqrt(1.0 - (double)r1);
	return vnorm(V3(cosf(term1) * term2, sinf(term1) * term2, sqrtf(r1)));
}

/* ---- Disney BSDF, RFW/system/context/rfw/bsdf/disney.h + compat.h:47-74 ------------------------------------- */
#define O_INVPI 0.318309886183790671537767526745028724f
#define O_PI 3.14159265358979323846264338327950288f
#define O_INV2PI 0.159154943091895335768883763372514362f
#define O_TWOPI 6.28318530717958647692528676655900576f

typedef struct
{
	v3 color;
	v3 absorption;
	uint32_t p[4];
} oshading;

static inline float sd_chan(uint32_t v, int shift) { return (float)((v >> shift) & 255u) * (1.0f / 255.0f); }
#define SD_METALLIC(sd) sd_chan((sd)->p[0], 0)
#define SD_SUBSURFACE(sd) sd_chan((sd)->p[0], 8)
#define SD_SPECULAR(sd) sd_chan((sd)->p[0], 16)
#define SD_ROUGHNESS(sd) fmaxf(0.001f, sd_chan((sd)->p[0], 24))
#define SD_SPECTINT(sd) sd_chan((sd)->p[1], 0)
#define SD_CLEARCOAT(sd) sd_chan((sd)->p[2], 0)
#define SD_CLEARCOATGLOSS(sd) sd_chan((sd)->p[2], 8)
#define SD_TRANSMISSION(sd) sd_chan((sd)->p[2], 16)
#define SD_ETA(sd) sd_chan((sd)->p[2], 24)

static inline float d_sqr(float x) { return x * x; }
/* disney.h:32-36 */
static inline float schlick_fresnel(float u)
{
	const float m = fclamp(1.0f - u, 0.0f, 1.0f);
	return (m * m) * (m * m) * m;
}
/* disney.h:38-45 */
static inline float gtr1(float NDotH, float a)
{
	if (a >= 1.0f)
		return O_INVPI;
	const float a2 = a * a;
	const float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
	return (a2 - 1.0f) / (O_PI * logf(a2) * t);
}
/* disney.h:47-52 */
static inline float gtr2(float NDotH, float a)
{
	const float a2 = a * a;
	const float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
	return a2 / (O_PI * t * t);
}
/* disney.h:54-59 */
static inline float smith_ggx(float NDotv, float alphaG)
{
	const float a = alphaG * alphaG;
	const float b = NDotv * NDotv;
	return 1.0f / (NDotv + sqrtf(a + b - a * b));
}
/* disney.h:61-72 */
static inline float fresnel_fr(float VDotN, float eio)
{
	const float SinThetaT2 = d_sqr(eio) * (1.0f - VDotN * VDotN);
	if (SinThetaT2 > 1.0f)
		return 1.0f;
	const float LDotN = sqrtf(1.0f - SinThetaT2);
	const float eta = 1.0f / eio;
	const float r1 = (VDotN - eta * LDotN) / (VDotN + eta * LDotN);
	const float r2 = (LDotN - eta * VDotN) / (LDotN + eta * VDotN);
	return 0.5f * (d_sqr(r1) + d_sqr(r2));
}
/* disney.h:74-81 */
static inline v3 safe_normalize(v3 a)
{
	const float ls = vdot(a, a);
	if (ls > 0.0f)
		return vscale(a, 1.0f / sqrtf(ls));
	return V3(0, 0, 0);
}
/* disney.h:19-30 */
static inline int refract_dir(v3 wi, v3 n, float eta, v3 *wt)
{
	const float cosThetaI = vdot(n, wi);
	const float sin2ThetaI = fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI);
	const float sin2ThetaT = eta * eta * sin2ThetaI;
	if (sin2ThetaT >= 1.0f)
		return 0;
	const float cosThetaT = sqrtf(1.0f - sin2ThetaT);
	*wt = vadd(vscale(vscale(wi, -1.0f), eta), vscale(n, eta * cosThetaI - cosThetaT));
	return 1;
}
/* disney.h:83-101 */
static inline float bsdf_pdf(const oshading *sd, v3 N, v3 wo, v3 wi)
{
	float bsdfPdf = 0.0f, brdfPdf;
	if (vdot(wi, N) <= 0.0f)
		brdfPdf = O_INV2PI * SD_SUBSURFACE(sd) * 0.5f;
	else
	{
		const float F = fresnel_fr(vdot(N, wo), SD_ETA(sd));
		const v3 halfway = safe_normalize(vadd(wi, wo));
		const float cosThetaHalf = fabsf(vdot(halfway, N));
		const float pdfHalf = gtr2(cosThetaHalf, SD_ROUGHNESS(sd)) * cosThetaHalf;
		const float pdfSpec = 0.25f * pdfHalf / fmaxf(1.e-6f, vdot(wi, halfway));
		const float pdfDiff = fabsf(vdot(wi, N)) * O_INVPI * (1.0f - SD_SUBSURFACE(sd));
		bsdfPdf = pdfSpec * F;
		brdfPdf = flerp(pdfDiff, pdfSpec, 0.5f);
	}
	return flerp(brdfPdf, bsdfPdf, SD_TRANSMISSION(sd));
}
/* disney.h:104-185 */
static inline v3 bsdf_eval(const oshading *sd, v3 N, v3 wo, v3 wi, float t, int backfacing)
{
	const float NDotL = vdot(N, wi);
	const float NDotV = vdot(N, wo);
	const v3 H = vnorm(vadd(wi, wo));
	const float NDotH = vdot(N, H);
	const float LDotH = vdot(wi, H);
	const v3 Cdlin = sd->color;
	const float Cdlum = .3f * Cdlin.x + .6f * Cdlin.y + .1f * Cdlin.z;
	const v3 Ctint = Cdlum > 0.0f ? vscale(Cdlin, 1.0f / Cdlum) : V3(1, 1, 1);
	const float METALLIC = SD_METALLIC(sd), TRANSMISSION = SD_TRANSMISSION(sd), SUBSURFACE = SD_SUBSURFACE(sd);
	const float ROUGHNESS = SD_ROUGHNESS(sd), ETA = SD_ETA(sd);
	const v3 Cspec0 =
		vlerp(vscale(vlerp(V3(1, 1, 1), Ctint, SD_SPECTINT(sd)), SD_SPECULAR(sd) * .08f), Cdlin, METALLIC);
	v3 bsdf = V3(0, 0, 0), brdf = V3(0, 0, 0);
	if (TRANSMISSION > 0.0f)
	{
		if (NDotL <= 0)
		{
			const float F = fresnel_fr(NDotV, ETA);
			const float s = (1.0f - F) / fabsf(NDotL) * (1.0f - METALLIC) * TRANSMISSION;
			bsdf = V3(s, s, s);
		}
		else
		{
			const float a = ROUGHNESS;
			const float Ds = gtr2(NDotH, a);
			const float FH = fresnel_fr(LDotH, ETA);
			const v3 Fs = vlerp(Cspec0, V3(1, 1, 1), FH);
			const float Gs = smith_ggx(NDotV, a) * smith_ggx(NDotL, a);
			bsdf = vscale(Fs, Gs * Ds);
		}
	}
	if (TRANSMISSION < 1.0f)
	{
		if (NDotL <= 0)
		{
			if (SUBSURFACE > 0.0f)
			{
				const v3 s = V3(sqrtf(sd->color.x), sqrtf(sd->color.y), sqrtf(sd->color.z));
				const float FL = schlick_fresnel(fabsf(NDotL)), FV = schlick_fresnel(NDotV);
				const float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
				brdf = vscale(vscale(vscale(vscale(s, O_INVPI), SUBSURFACE), Fd), 1.0f - METALLIC);
			}
		}
		else
		{
			const float a = ROUGHNESS;
			const float Ds = gtr2(NDotH, a);
			const float FH = schlick_fresnel(LDotH);
			const v3 Fs = vlerp(Cspec0, V3(1, 1, 1), FH);
			const float Gs = smith_ggx(NDotV, a) * smith_ggx(NDotL, a);
			const float FL = schlick_fresnel(NDotL), FV = schlick_fresnel(NDotV);
			const float Fd90 = 0.5f + 2.0f * LDotH * LDotH * a;
			const float Fd = flerp(1.0f, Fd90, FL) * flerp(1.0f, Fd90, FV);
			const float Dr = gtr1(NDotH, flerp(.1f, .001f, SD_CLEARCOATGLOSS(sd)));
			const float Fc = flerp(.04f, 1.0f, FH);
			const float Gr = smith_ggx(NDotL, .25f) * smith_ggx(NDotV, .25f);
			const v3 diff = vscale(vscale(vscale(Cdlin, O_INVPI * Fd), 1.0f - METALLIC), 1.0f - SUBSURFACE);
			const v3 spec = vscale(vscale(Fs, Gs), Ds);
			const float cc = SD_CLEARCOAT(sd) * Gr * Fc * Dr;
			brdf = vadd(vadd(diff, spec), V3(cc, cc, cc));
		}
	}
	const v3 final = vlerp(brdf, bsdf, TRANSMISSION);
	if (backfacing)
		return vmul(final, V3(expf(-sd->absorption.x * t), expf(-sd->absorption.y * t), expf(-sd->absorption.z * t)));
	return final;
}
static inline v3 reflect_dir(v3 I, v3 N) { return vsub(I, vscale(N, vdot(N, I) * 2.0f)); }
/* disney.h:188-262.  Leaves *pdf untouched in the refraction-failed and reflected-by-Fresnel branches exactly
 * as the reference does (callers initialise it to 0). */
static inline void bsdf_sample(const oshading *sd, v3 T, v3 B, v3 N, v3 wo, v3 *wi, float *pdf, float r3, float r4)
{
	const float transmission = SD_TRANSMISSION(sd);
	const float ROUGHNESS = SD_ROUGHNESS(sd);
	if (r3 < transmission)
	{
		const float F = fresnel_fr(vdot(N, wo), SD_ETA(sd));
		if (r4 < F)
		{
			const float r1 = r3 / transmission;
			const float r2 = r4 / F;
			const float cosThetaHalf = sqrtf((1.0f - r2) / (1.0f + (d_sqr(ROUGHNESS) - 1.0f) * r2));
			const float sinThetaHalf = sqrtf(fmaxf(0.0f, 1.0f - d_sqr(cosThetaHalf)));
			const float sinPhiHalf = sinf(r1 * O_TWOPI);
			const float cosPhiHalf = cosf(r1 * O_TWOPI);
			v3 halfway = vadd(vadd(vscale(T, sinThetaHalf * cosPhiHalf), vscale(B, sinThetaHalf * sinPhiHalf)),
							  vscale(N, cosThetaHalf));
			if (vdot(halfway, wo) <= 0.0f)
				halfway = vscale(halfway, -1.0f);
			*wi = reflect_dir(vscale(wo, -1.0f), halfway);
		}
		else
		{
			*pdf = 0;
			if (refract_dir(wo, N, SD_ETA(sd), wi))
				*pdf = (1.0f - F) * transmission;
		}
		return;
	}
	const float r1 = (r3 - transmission) / (1 - transmission);
	if (r4 < 0.5f)
	{
		const float r2 = r4 * 2;
		const float subsurface = SD_SUBSURFACE(sd);
		v3 d;
		if (r2 < subsurface)
		{
			const float r5 = r2 / subsurface;
			d = diffuse_reflection_uniform(r1, r5);
			d.z *= -1.0f;
		}
		else
		{
			const float r5 = (r2 - subsurface) / (1.0f - subsurface);
			d = diffuse_reflection_cos_weighted(r1, r5);
		}
		*wi = vadd(vadd(vscale(T, d.x), vscale(B, d.y)), vscale(N, d.z));
	}
	else
	{
		const float r2 = (r4 - 0.5f) * 2.0f;
		const float cosThetaHalf = sqrtf((1.0f - r2) / (1.0f + (d_sqr(ROUGHNESS) - 1.0f) * r2));
		const float sinThetaHalf = sqrtf(fmaxf(0.0f, 1.0f - d_sqr(cosThetaHalf)));
		const float sinPhiHalf = sinf(r1 * O_TWOPI);
		const float cosPhiHalf = cosf(r1 * O_TWOPI);
		v3 halfway = vadd(vadd(vscale(T, sinThetaHalf * cosPhiHalf), vscale(B, sinThetaHalf * sinPhiHalf)),
						  vscale(N, cosThetaHalf));
		if (vdot(halfway, wo) <= 0.0f)
			halfway = vscale(halfway, -1.0f);
		*wi = reflect_dir(vscale(wo, -1.0f), halfway);
	}
	*pdf = bsdf_pdf(sd, N, wo, *wi);
}

#endif
