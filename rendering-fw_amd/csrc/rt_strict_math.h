// rt_strict_math.h — RT_STRICT_MATH: the transcendental functions of the path tracer as plain float arithmetic.
//
// The shipped kernels evaluate sin / cos of turn fractions with v_sin_f32 / v_cos_f32, the reciprocal of the triangle
// determinant with v_rcp_f32 and everything else through the device's math library; the host emulation of the same sources
// (tests/emu) uses glibc.  Each differs from the other in the last bit here and there, a path tracer turns a last bit into a
// different random decision, and so "emulation == HIP" had been a statistical statement (2 % of the bench scene's pixels
// differ).  The VALIDATION build (build.py: build_strict(); never shipped, loaded by tests/test_strict_gpu.py only) replaces
// every such function by the ones below — argument reduction and a polynomial in explicit fmaf() steps, no library call, no
// hardware approximation, fma contraction off for the whole translation unit — so that device and host execute the same
// IEEE operations in the same order and must agree to the bit: whatever differs between the shipped HIP image and the
// emulation is then PROVEN to be arithmetic mode, not a difference between two programs.  Accuracy is that of a careful
// float implementation (about 1-2 ulp), not correctly rounded; the oracle keeps glibc, so emulation-strict vs oracle counts
// the decisions that the choice of libm flips.
#pragma once

namespace rt
{
namespace strict
{
RT_FN float poly_sin(float a) // |a| <= pi/4
{
	const float z = rounded(a * a);
	float p = 2.7557319e-6f;
	p = fmaf(p, z, -1.9841270e-4f);
	p = fmaf(p, z, 8.3333333e-3f);
	p = fmaf(p, z, -1.6666667e-1f);
	return fmaf(rounded(a * z), p, a);
}
RT_FN float poly_cos(float a) // |a| <= pi/4
{
	const float z = rounded(a * a);
	float p = 2.4801587e-5f;
	p = fmaf(p, z, -1.3888889e-3f);
	p = fmaf(p, z, 4.1666667e-2f);
	p = fmaf(p, z, -0.5f);
	return fmaf(p, z, 1.0f);
}
RT_FN void quadrant(int k, float a, float &s, float &c) // sin / cos of k quarter turns + a
{
	const float ps = poly_sin(a), pc = poly_cos(a);
	k &= 3;
	s = k == 0 ? ps : (k == 1 ? pc : (k == 2 ? -ps : -pc));
	c = k == 0 ? pc : (k == 1 ? -ps : (k == 2 ? -pc : ps));
}
RT_FN void sincos_turns(float frac, float &s, float &c)
{
	const float t = rounded(frac * 4.0f);
	const float kf = floorf(t + 0.5f);
	quadrant((int)kf, rounded((t - kf) * 1.57079632679f), s, c);
}
RT_FN void sincos_rad(float x, float &s, float &c) // |x| small (lens blades: 0 .. 2 pi)
{
	const float kf = floorf(fmaf(x, 0.636619772f, 0.5f));
	float r = fmaf(-kf, 1.57079625129699707031f, x); // pi/2 in two parts
	r = fmaf(-kf, 7.54978941586159635335e-08f, r);
	quadrant((int)kf, r, s, c);
}
RT_FN float sin_(float x)
{
	float s, c;
	sincos_rad(x, s, c);
	return s;
}
RT_FN float cos_(float x)
{
	float s, c;
	sincos_rad(x, s, c);
	return c;
}
RT_FN float exp_(float x)
{
	if (!(x > -87.0f)) // (also NaN) results below the normal range are of no consequence here: 0
		return x != x ? x : 0.0f;
	if (x > 88.0f)
		return 3.0e38f * 3.0e38f;
	const float kf = floorf(fmaf(x, 1.44269504089f, 0.5f));
	float r = fmaf(-kf, 0.693145751953125f, x); // ln 2 in two parts
	r = fmaf(-kf, 1.42860682030941723212e-6f, r);
	float p = 1.9841270e-4f;
	p = fmaf(p, r, 1.3888889e-3f);
	p = fmaf(p, r, 8.3333333e-3f);
	p = fmaf(p, r, 4.1666667e-2f);
	p = fmaf(p, r, 1.6666667e-1f);
	p = fmaf(p, r, 0.5f);
	p = fmaf(p, r, 1.0f);
	p = fmaf(p, r, 1.0f);
	// 2^k by its bit pattern (k in -126 .. 127 after the range checks): exact, no library scaling function
	return rounded(p * ubits((uint32_t)((int)kf + 127) << 23));
}
RT_FN float log_(float x) // x > 0, finite, normal
{
	const uint32_t b = fbits(x);
	int e = (int)(b >> 23) - 126;
	float m = ubits((b & 0x007FFFFFu) | 0x3F000000u); // [0.5, 1)
	if (m < 0.70710678f)
		m = m + m, e -= 1;
	const float f = m - 1.0f;
	const float s = f / (2.0f + f);
	const float z = rounded(s * s);
	float p = 0.18181818f;
	p = fmaf(p, z, 0.22222222f);
	p = fmaf(p, z, 0.28571429f);
	p = fmaf(p, z, 0.4f);
	p = fmaf(p, z, 0.66666667f);
	const float lm = fmaf(rounded(s * z), p, s + s);
	return fmaf((float)e, 0.69314718056f, lm);
}
RT_FN float log2_(float x) { return rounded(log_(x) * 1.44269504089f); }
RT_FN float atan_small(float x) // |x| <= tan(pi/8)
{
	const float z = rounded(x * x);
	float p = 0.058823529f; // 1/17
	p = fmaf(p, z, -0.066666667f);
	p = fmaf(p, z, 0.076923077f);
	p = fmaf(p, z, -0.090909091f);
	p = fmaf(p, z, 0.11111111f);
	p = fmaf(p, z, -0.14285714f);
	p = fmaf(p, z, 0.2f);
	p = fmaf(p, z, -0.33333333f);
	return fmaf(rounded(x * z), p, x);
}
RT_FN float atan2_(float y, float x)
{
	const float ax = fabsf(x), ay = fabsf(y);
	const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
	const float a = mx > 0.0f ? mn / mx : 0.0f;
	float r = a > 0.41421356f ? 0.78539816339f + atan_small((a - 1.0f) / (a + 1.0f)) : atan_small(a);
	if (ay > ax)
		r = 1.57079632679f - r;
	if (x < 0.0f || (x == 0.0f && fbits(x) != 0u)) // (x < 0 or -0)
		r = 3.14159265359f - r;
	return copysignf(r, y);
}
RT_FN float acos_(float x) { return atan2_(sqrtf(rounded((1.0f - x) * (1.0f + x))), x); } // x in [-1, 1]
} // namespace strict
} // namespace rt
