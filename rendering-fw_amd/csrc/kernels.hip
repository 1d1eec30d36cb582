// kernels.hip — the wavefront stages of the HIP rendercore for gfx950 (MI355X, CDNA4).
//
//   extend        closest-hit traversal of one wave of rays: k_primary_packet walks the tree once per wave; k_trace_stream<false> keeps 64 traversals in
//                 flight per wave and refill finished lanes themselves (stream_rays); k_extend is the one-ray-per-lane form
//                 (parity primaries, small pt launches, rfwhip_trace_rays)
//   shade_parity  EmbreeRT-equivalent direct-lighting integrator (shadow rays traced inline, fixed light order)
//   shade_pt      path-tracing shade: per-wave hit queues, emits the shadow-ray and extension-ray queues in blocks
//   connect       any-hit traversal of the shadow queue (k_trace_stream<true> / k_connect; the primary vertices' connections as packets
//                 sorted by light: k_shadow_packet), adds unoccluded contributions
//   resolve/present/deinterleave   accumulate the batch, scale by 1/samples, undo the multi-GPU strip interleave
//   rng_states    xor128 jump-ahead: per-packet generator states for the parity integrator's jitter stream
//   refit / skin / morph   bottom-up BVH refit after a same-topology set_mesh; vertex posing on the device
//
// CDNA4 specifics:
//   * the traversal stack lives in LDS as stack[entry][thread] (bank = thread % 32: conflict-free), 12 entries per lane
//     in the closest-hit kernels and 8 in the occlusion kernels, with a 64-entry private-memory spill behind it that
//     ordinary rays never reach (this array is the "scratch" the kernel statistics show); the top of the largest BLAS
//     (128 compressed 4-wide nodes, 8 KiB) is staged in LDS by every workgroup;
//   * a wave takes runs of up to 512 consecutive rays from a launch's queue with one atomic; lanes refill from the run once
//     32 are idle; the node phase of a wave ends as soon as 32 lanes hold a leaf (leaf vote);
//   * queue slots are allocated with __ballot + mbcnt prefixes out of blocks of 256 slots a wave reserves with ONE atomic
//     (the reference issues one atomicAdd per surviving thread, CUDART/src/Kernels.cu:640,747,788; same-address atomics
//     complete about every 7 ns on this part);
//   * persistent grids of 8..12 (traversal) / ..16 (shade) workgroups per CU; the one-ray-per-lane kernels pull 256-ray
//     chunks from per-XCD queues: chunks are dealt to the 8 XCDs in groups of one tile row, workgroup b (which runs on XCD
//     b % 8) pulls consecutive chunks of its XCD's sequence, so neighbouring tiles meet in the same 4 MiB L2;
//   * wave counts come from device-side counters; the host never reads a counter between bounces
//     (contrast CUDART/src/Context.cpp:98,145).
// No MFMA: the path is divergent pointer chasing — VALU issue at ~half-full waves and the vector L1's lane-load rate bound
// the per-lane traversal kernels, VALU issue and the CU's scalar unit (one scalar instruction per CU and clock: 4.5 clocks of a
// SIMD's turn each, tools/dev/micro/inst_rate5.hip) the packet kernels, HBM traffic the shade kernel.
#include "kernels.h"
#include "rt_core.h"

#include <string.h>
#include <algorithm>

using namespace rt;

namespace rtk
{

constexpr int BLOCK = 256;
constexpr int KAT_IN = 24, KAT_OUT = 8; // = RFWHIP_KAT_IN / RFWHIP_KAT_OUT (static_assert in rfwhip_api.cpp)
// minimum waves per SIMD the register allocator must leave room for in the traversal kernels.  Swept on MI355X with
// the final kernels: 4 / 5 / 6 / 7 / 8 -> 1863 / 1910 / 1923 / 1864 / 1863 Msamples/s (the 85-register budget of 6 makes
// the compiler schedule the node loop tighter; the kernels need 36-52 registers either way)
#ifndef RT_TRAVERSAL_WAVES
#define RT_TRAVERSAL_WAVES 6
#endif
// the persistent-lane kernels of the incoherent waves: with the 64-byte nodes their LDS is 20 / 16 KiB per workgroup, so
// eight workgroups fit a CU; bound to 64 registers they run 8 waves per SIMD instead of 7 (65 registers): +3 % (and every
// step down in resident waves costs: 6 / 5 / 4 / 3 workgroups per CU -> 2065 / 1899 / 1802 / 1596 Msamples/s)
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 8
#endif
#ifndef RT_TRACE_VGPRS
#define RT_TRACE_VGPRS 64 // (launch bounds alone let the allocator settle one register above the 8-wave budget)
#endif
#ifndef RT_PRIMARY_WAVES
#define RT_PRIMARY_WAVES 7 // k_extend: the kernel that also generates the primary rays (6 / 7 / 8 -> 6.95 / 6.65 / 7.42 ms per launch)
#endif
#ifndef RT_ANY_WAVES
#define RT_ANY_WAVES 8 // occlusion kernels (fewer registers, less LDS)
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4 // the textured shade kernel: 128 registers
#endif
// the shade kernel's waves queue their misses as well as their hits (1), or shade every chunk's misses in place (0).
// (Two queued misses per lane through a routine of their own — the loads of both in flight together, a miss being four dependent
// round trips and a few dozen instructions — was built, is bit-identical, and loses: shade alone 8.57 -> 9.05 ms per sub-batch,
// 4610 -> 4505 Msamples/s; the second code path costs the hit path registers.)
#ifndef RT_MISS_QUEUE
#define RT_MISS_QUEUE 1
#endif
// (Round 5: the scan's primitive ids asked for one chunk ahead with global_load_lds_dword — straight into the wave's LDS, no register
// in flight — so that a scan is not a memory round trip with nothing beside it (13 % of a wave's time by the clock): bit-identical,
// and slower, 8.22 -> 8.63 ms per sub-batch — the plain kernel went from 10 to 18 spilled registers.  Like every change to this
// kernel's body since round 4.)
#ifndef RT_SHADE_WAVES_PLAIN
#define RT_SHADE_WAVES_PLAIN 5 // the shade kernel of scenes without textures: 111 registers unbounded; 4 / 5 / 6 waves ->
								// 2492 / 2589 / 2584 Msamples/s (96 registers + a few spilled dwords at 5)
#endif

// ================================================================================================================
// per-thread context: traversal stack + compaction + statistics.  Device and host-emulation flavours.
// ================================================================================================================
#if defined(RT_DEVICE_BUILD)

struct Ctx
{
	TravStack stk;
	float *pot = nullptr; // this lane's column of the light-potential cache (shade kernel)

	// Block-wise slot allocation of the shade kernel's two output queues (rt_types.h: QUEUE_BLOCK).  Wave-uniform state.
	struct OutQueue
	{
		uint32_t pos = 0, end = 0, rays = 0;
	};
	OutQueue q_ext, q_shadow;
	uint32_t q_block = QUEUE_BLOCK;
	__device__ __forceinline__ uint32_t alloc(OutQueue &q, bool flag, uint32_t *queue_len)
	{
		const unsigned long long mask = __ballot(flag);
		const uint32_t n = (uint32_t)__popcll(mask);
		if (n == 0u)
			return 0u;
		const uint32_t room = q.end - q.pos; // (q_block == 0: exact mode — pos == end always, every call reserves its n slots)
		uint32_t fresh = 0;
		if (n > room) // (wave-uniform) the outputs that do not fit the rest of this block start a new one
		{
			if (__lane_id() == 0u)
				fresh = atomicAdd(queue_len, q_block ? q_block : n);
			fresh = (uint32_t)__builtin_amdgcn_readfirstlane((int)fresh);
		}
		const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
		const uint32_t slot = prefix < room ? q.pos + prefix : fresh + (prefix - room);
		if (n > room)
			q.pos = fresh + (n - room), q.end = fresh + (q_block ? q_block : n);
		else
			q.pos += n;
		q.rays += n;
		return slot;
	}
	__device__ __forceinline__ void add64(unsigned long long *dst, uint32_t v)
	{
		// wave reduction, then one atomic per wave
		for (int o = 32; o > 0; o >>= 1)
			v += __shfl_down(v, o);
		if (__lane_id() == 0 && v)
			atomicAdd(dst, (unsigned long long)v);
	}
};

#else

struct Ctx
{
	TravStack stk;
	uint32_t lds[LDS_STACK_MAX], spill[SPILL_STACK];
	float potbuf[POT_SLOTS];
	float *pot;
	uint32_t overflow_sink = 0;
	Ctx() { stk.lds = lds, stk.spill = spill, pot = potbuf, stk.top = nullptr, stk.top_first = 0, stk.top_count = 0, stk.overflow = &overflow_sink, stk.stride = 1; }
	explicit Ctx(const Params &p) : Ctx()
	{
		stk.overflow = &p.wv.counters->stack_overflow;
		// emulation: the "LDS" rows alias the node table (TOP_ROWS = 8), the range check is the device's
		stk.top = (const f4 *)(p.sc.nodes4 + p.lds_first), stk.top_first = p.lds_first, stk.top_count = p.lds_count;
	}
	// (emulation: one slot at a time, no blocks, no void entries — queue length == ray count)
	struct OutQueue
	{
		uint32_t rays = 0;
	};
	OutQueue q_ext, q_shadow;
	uint32_t alloc(OutQueue &q, bool flag, uint32_t *queue_len)
	{
		if (!flag)
			return 0u;
		q.rays++;
		return (*queue_len)++;
	}
	void add64(unsigned long long *dst, uint32_t v) { *dst += v; }
};

#endif

// ================================================================================================================
// work items (one ray / path / pixel each) — shared by the device kernels and the host emulation
// ================================================================================================================
// The end of primary ray `idx` of the pt integrator where the wave is converged when its rays end: the packet kernel (device:
// k_primary_packet, emulation: packet_emu::primary) and the one-ray-per-lane kernel of small launches (extend_item<GEN_PT>).
// A hit goes to the shade stage: direction record + hit record.  A MISS is finished here (RT_PRIMARY_MISS): the wave is converged,
// all 64 samples of a sky pixel miss together, and what the shade stage would do for it — the sky along the ray into the slot
// (pt_shade, h.prim < 0 at depth 0: throughput 1, pdf 1; shade_pt_item: alpha -1 = the path ends, no connection record) — needs
// nothing the kernel does not hold.  The shade kernel then neither queues the path nor reads its direction (27 % of the bench
// scene's primaries), and no direction record is written for it; its hit record says HIT_MISS_SHADED (read back as a miss).
#ifndef RT_PRIMARY_INITS_RAD
#define RT_PRIMARY_INITS_RAD 1
#endif
#ifndef RT_PRIMARY_MISS
#define RT_PRIMARY_MISS 1
#endif
// Round 6: the primary wave writes NO ray record at all (0) — the shade kernel regenerates a primary ray from its pixel and sample
// (pt_primary_ray has a fixed arithmetic shape: the same bits in every kernel) instead of reading 16 bytes the primary kernel wrote
// for it: ~60 instructions in a kernel whose VALUs are busy 0.45 of the time against 32 bytes of HBM traffic per primary hit in the
// kernel that is bound by it.  1: direction record per hit (+ origin record behind a lens), rounds 1-5.
#ifndef RT_PRIMARY_RAY_RECORD
#define RT_PRIMARY_RAY_RECORD 0
#endif
RT_FN void primary_finish_item(const Params &q, uint32_t idx, f3 D, const Hit &h)
{
	int prim = h.prim;
	if (RT_PRIMARY_MISS && prim < 0)
	{
		f3 radiance = mk3(0, 0, 0);
		const f3 contribution = (mk3(1, 1, 1) * m_rcp(1.0f)) * pt_sky(q.sc, D);
		if (!any_nan(contribution))
			radiance = clamp_intensity(contribution, q.cam.clamp_value);
		q.wv.rad[idx] = mk4(radiance.x, radiance.y, radiance.z, q.wv.rad_nee ? -1.0f : 1.0f);
		prim = HIT_MISS_SHADED;
	}
	else if (RT_PRIMARY_RAY_RECORD)
		q.wv.dir[0][idx] = mk4(D.x, D.y, D.z, 0.0f);
	q.wv.hit0[idx] = mk4(h.t, h.u, h.v, ubits((uint32_t)prim));
	q.wv.hit0_inst[idx] = h.inst;
	// a hit's slot starts as what most hits leave there — no radiance, a path that goes on — from here, where the stores of a
	// wave are neighbours and the memory pipes idle; the shade kernel writes the slot of a path that adds light or ends only
	// (after the hit record's stores: its registers are free then — in front of them the packet kernel spilled four)
	if (RT_PRIMARY_INITS_RAD && prim >= 0)
	{
		float zero = 0.0f, one = 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("" : "+v"(zero), "+v"(one)); // (made here: hoisted out of the packet loop, the constant record was spilled and reloaded per packet)
#endif
		q.wv.rad[idx] = mk4(zero, zero, zero, one);
	}
}

template <int GEN, bool COUNT>
RT_FN void extend_item(const Params &p, uint32_t i, bool active, Ctx &ctx)
{
	f3 O = mk3(0, 0, 0), D = mk3(0, 0, 1);
	const uint32_t b = p.depth & 1u;
	float t_min = 1e-5f, t_max = 1e34f; // every wave of the integrators uses the reference's interval (Kernels.cu:455,478)
	if (GEN == GEN_BUFFER || GEN == GEN_RANGED)
	{
		if (active)
		{
			const f4 o4 = p.wv.org[b][i], d4 = p.wv.dir[b][i];
			O = xyz(o4), D = xyz(d4);
			if (GEN == GEN_RANGED)
				t_min = o4.w, t_max = d4.w;
			else if (fbits(o4.w) == RAY_VOID) // the unfilled rest of a wave's last queue block
			{
				p.wv.hit[i] = mk4(0, 0, 0, ubits((uint32_t)HIT_VOID));
				active = false;
			}
		}
	}
	else
	{
		const PixelRef pr = slot_to_pixel(p.fr, i);
		active = active && pr.valid;
		if (GEN == GEN_PT)
		{
			if (active)
				pt_primary_ray(p.cam, p.fr, pr.x, pr.y, p.fr.sample_base + pr.sample, O, D);
		}
		else
		{
			// EmbreeRT renders whole 4x2 packets only (Context.cpp:137-139): remainder columns/rows are never written
			const uint32_t npx = p.fr.W / 4u, npy = p.fr.H / 2u;
			active = active && pr.x < npx * 4u && pr.y < npy * 2u;
			if (active)
			{
				float r0 = 0.5f, r1 = 0.5f, r2 = 0.5f, r3 = 0.5f;
				if (!p.parity_no_jitter)
				{
					// Ray.cpp:213-216 — 8 x r0, 8 x r1, 8 x r2, 8 x r3 per packet; lane j of the packet takes draw j
					const uint32_t packet = (pr.y / 2u) * npx + pr.x / 4u;
					const uint32_t lane = (pr.y & 1u) * 4u + (pr.x & 3u);
					const uint32_t *sp = p.wv.packet_rng + 4ull * ((unsigned long long)pr.sample * npx * npy + packet);
					uint32_t s[4] = {sp[0], sp[1], sp[2], sp[3]};
					const bool lens = p.cam.aperture != 0.0f;
					const uint32_t draws = lens ? 32u : 16u;
					for (uint32_t k = 0; k < draws; k++)
					{
						const float r = u32_to_unit(xor128_next(s));
						if (k == lane)
							r0 = r;
						else if (k == 8u + lane)
							r1 = r;
						else if (k == 16u + lane)
							r2 = r;
						else if (k == 24u + lane)
							r3 = r;
					}
				}
				parity_primary_ray(p.cam, p.fr, pr.x, pr.y, r0, r1, r2, r3, O, D);
			}
		}
		if (active)
		{
			// (pt, pinhole camera: every primary ray starts at the camera position and entry i of the primary wave is path slot
			// i — the shade kernel needs no origin record: 16 bytes less written here and read there per primary ray)
			if (GEN != GEN_PT || (RT_PRIMARY_RAY_RECORD && p.cam.aperture != 0.0f))
				p.wv.org[0][i] = mk4(O.x, O.y, O.z, ubits((i << 1) | 1u));
			if (GEN != GEN_PT) // (pt: written with the hit record, or not at all for a miss — primary_finish_item)
				p.wv.dir[0][i] = mk4(D.x, D.y, D.z, 0.0f);
		}
	}
	Hit h;
	h.t = 1e34f, h.u = 0, h.v = 0, h.prim = -1, h.inst = -1;
	TStat st;
	st.inner = 0, st.tris = 0, st.lds = 0;
	if (active)
	{
		trace<false, COUNT>(p.sc, O, D, t_min, t_max, h, ctx.stk, st);
		if (GEN == GEN_PT)
			primary_finish_item(p, i, D, h);
		else
		{
			f4 *hb = p.depth == 0 ? p.wv.hit0 : p.wv.hit;
			int *ib = p.depth == 0 ? p.wv.hit0_inst : p.wv.hit_inst;
			hb[i] = mk4(h.t, h.u, h.v, ubits((uint32_t)h.prim));
			ib[i] = h.inst;
		}
	}
	if (COUNT)
	{
		ctx.add64(&p.wv.counters->inner_extend, st.inner);
		ctx.add64(&p.wv.counters->tris_extend, st.tris);
		ctx.add64(&p.wv.counters->lds_extend, st.lds);
		ctx.add64(&p.wv.counters->rays_extend, active ? 1u : 0u);
	}
}

template <bool COUNT>
RT_FN void shade_parity_item(const Params &p, uint32_t i, bool active, Ctx &ctx)
{
	const PixelRef pr = slot_to_pixel(p.fr, i);
	active = active && pr.valid && pr.x < (p.fr.W / 4u) * 4u && pr.y < (p.fr.H / 2u) * 2u;
	TStat st;
	st.inner = 0, st.tris = 0, st.lds = 0;
	uint32_t nshadow = 0;
	f4 out = mk4(0, 0, 0, 0);
	if (active)
	{
		const f4 o4 = p.wv.org[0][i], d4 = p.wv.dir[0][i], h4 = p.wv.hit0[i];
		Hit h;
		h.t = h4.x, h.u = h4.y, h.v = h4.z, h.prim = (int)fbits(h4.w), h.inst = p.wv.hit0_inst[i];
		if (h.prim < 0)
		{
			const f3 s = parity_sky(p.sc, xyz(d4));
			out = mk4(s.x, s.y, s.z, 0.0f);
		}
		else
		{
			if (pr.sample == 0 && pr.y * p.fr.W + pr.x == p.fr.probe_pixel)
			{
				WaveCounters *c = p.wv.counters;
				c->probe_inst = (uint32_t)h.inst, c->probe_prim = (uint32_t)h.prim, c->probe_dist = h.t, c->probe_valid = 1u;
			}
			out = parity_shade<COUNT>(p.sc, xyz(o4), xyz(d4), h, ctx.stk, st, nshadow);
		}
	}
	if (pr.valid)
		p.wv.rad[i] = out;
	if (COUNT)
	{
		ctx.add64(&p.wv.counters->inner_shadow, st.inner);
		ctx.add64(&p.wv.counters->tris_shadow, st.tris);
		ctx.add64(&p.wv.counters->lds_shadow, st.lds);
		ctx.add64(&p.wv.counters->rays_shadow, nshadow);
		ctx.add64(&p.wv.counters->shaded, (active && out.w > 0.0f) ? 1u : 0u);
	}
}

#if defined(RT_DIAG_SHADE_CLOCK)
__device__ unsigned long long g_shade_clk[32];
#define RT_ITEM_CLK , ClkProbe *clk
#define RT_ITEM_TICK(K) clk_tick(*clk, K)
#else
#define RT_ITEM_CLK
#define RT_ITEM_TICK(K)
#endif
// What pt_shade hands out while it runs (rt_core.h): the shadow ray — queue slot and stores at once, so that its registers are free
// for the BSDF sampling — and the probe pixel's hit.
struct ShadeSink
{
	const Params &p;
	Ctx &ctx;
	uint32_t slot;
	RT_FN void probe(const Hit &h) const
	{
		WaveCounters *c = p.wv.counters;
		c->probe_inst = (uint32_t)h.inst, c->probe_prim = (uint32_t)h.prim, c->probe_dist = h.t, c->probe_valid = 1u;
	}
	RT_FN void shadow(bool emit, const f4 &so, const f4 &sd, const f4 &se)
	{
		const uint32_t si = ctx.alloc(ctx.q_shadow, emit, &p.wv.counters->shadow_n[p.depth]);
		if (emit)
		{
			p.wv.sh_org[si] = so;
			p.wv.sh_dir[si] = sd;
			// Depth 0 with a connection buffer: the slot's connection term starts as what the shadow ray of this vertex would add if
			// the light is visible — 0 + e: the bits an accumulation onto a zeroed slot produces — and the connection wave of depth 0
			// only ZEROES it for an occluded light (connect_finish): that wave then retires a ray without reading anything (the
			// contribution record and the wait for it were 8 % of a shadow wave's time), and no contribution record is written here.
			if (p.depth != 0 || !p.wv.rad_nee)
				p.wv.sh_rad[si] = se;
			else
				p.wv.rad_nee[slot] = mk4(0.0f + se.x, 0.0f + se.y, 0.0f + se.z, 0.0f);
		}
	}
};

// One entry of the wave of depth p.depth.  `active` = the lane holds a path (the device kernel's scan has checked that the entry is
// a path's: not void, at depth 0 a real pixel's slot); h4 / hi = its hit record (the device kernel's scan has read them already).
template <bool TEX> RT_FN void shade_pt_item(const Params &p, uint32_t i, bool active, const f4 &h4, int hi, Ctx &ctx RT_ITEM_CLK)
{
	RT_ITEM_TICK(0);
	const uint32_t b = p.depth & 1u, nb = b ^ 1u;
	ShadeOut out;
	PathIn in;
	in.O = mk3(0, 0, 0), in.D = mk3(0, 0, 1), in.T = mk3(1, 1, 1), in.bsdfPdf = 1.0f;
	in.slot = 0, in.flags = 0, in.packedN = 0, in.depth = p.depth;
	Hit h;
	h.t = h4.x, h.u = h4.y, h.v = h4.z, h.prim = (int)fbits(h4.w), h.inst = hi;
	if (active && p.depth == 0 && !RT_PRIMARY_RAY_RECORD)
	{
		// (the primary wave: entry i IS path slot i, and its ray is a function of pixel and sample — regenerated, not read)
		const PixelRef pr = slot_to_pixel(p.fr, i);
		pt_primary_ray(p.cam, p.fr, pr.x, pr.y, p.fr.sample_base + pr.sample, in.O, in.D);
		in.slot = i, in.flags = 1u, in.packedN = 0u;
	}
	else if (active)
	{
		// (depth 0 behind a pinhole camera: no origin record was written — the origin is the camera, the slot is the entry)
		const f4 o4 = (p.depth == 0 && p.cam.aperture == 0.0f) ? mk4(p.cam.pos.x, p.cam.pos.y, p.cam.pos.z, ubits((i << 1) | 1u)) : p.wv.org[b][i];
		const f4 d4 = p.wv.dir[b][i];
		in.O = xyz(o4), in.D = xyz(d4);
		const uint32_t ow = fbits(o4.w);
		in.slot = ow >> 1, in.flags = ow & 1u, in.packedN = fbits(d4.w);
		if (p.depth != 0)
		{
			const f4 t4 = p.wv.thr[b][i];
			in.T = xyz(t4), in.bsdfPdf = t4.w;
		}
	}
	const uint32_t slot = in.slot;
	ShadeSink sink{p, ctx, slot};
	pt_shade<TEX>(p.sc, p.cam, p.fr, p.max_depth, active, in, h, out, ctx.pot, sink RT_CLK_ARG);
	if (active)
	{
		// depth 0 initialises the slot (no clear pass); later depths accumulate.  One path per slot => no race.
		if (p.depth == 0)
		{
			// A path that ends here — a sky miss, an emitter, nothing to continue with — never gets a connection term: its slot of
			// rad_nee is neither initialised nor read; alpha -1 tells the resolve (|alpha| is the alpha: the pt integrator's is 1).
			// 16 bytes less written here and 16 less read there for every such path (two in five on the terrain).
			const bool ends = p.wv.rad_nee && !out.emit_shadow && !out.emit_ext;
			// (a hit without light of its own that goes on: primary_finish_item has written exactly that)
			if (!RT_PRIMARY_INITS_RAD || h.prim < 0 || ends || out.radiance.x != 0.0f || out.radiance.y != 0.0f || out.radiance.z != 0.0f)
				p.wv.rad[slot] = mk4(out.radiance.x, out.radiance.y, out.radiance.z, ends ? -1.0f : 1.0f);
			// (a path with a shadow ray: ShadeSink::shadow has stored its connection term)  Paths that emit no shadow ray but go on
			// start theirs at zero.
			if (p.wv.rad_nee && out.emit_ext && !out.emit_shadow)
				p.wv.rad_nee[slot] = mk4(0, 0, 0, 0);
		}
		else if (out.radiance.x != 0.0f || out.radiance.y != 0.0f || out.radiance.z != 0.0f)
		{
			// (the same additions as three fire-and-forget global_atomic_add_f32 — no wait for the slot's old value — measured: the
			// L2's atomic units are the slower path by far, shade alone 8.55 -> 10.0 ms per sub-batch)
			f4 r = p.wv.rad[slot];
			r.x += out.radiance.x, r.y += out.radiance.y, r.z += out.radiance.z;
			p.wv.rad[slot] = r;
		}
	}
	RT_ITEM_TICK(7);
	const uint32_t ei = ctx.alloc(ctx.q_ext, out.emit_ext, &p.wv.counters->ext_n[p.depth + 1]);
	if (out.emit_ext)
	{
		p.wv.org[nb][ei] = out.eo;
		p.wv.dir[nb][ei] = out.ed;
		p.wv.thr[nb][ei] = out.et;
	}
	RT_ITEM_TICK(8);
}

// The end of shadow ray i of path slot `slot`.  Depth 0 with a connection buffer: the slot already holds the term of a visible
// light (shade_pt_item); an occluded one zeroes it — nothing is read.  Later depths accumulate.
RT_FN void connect_finish(const Params &p, uint32_t i, uint32_t slot, bool visible)
{
	if (p.depth == 0 && p.wv.rad_nee)
	{
		if (!visible)
			p.wv.rad_nee[slot] = mk4(0, 0, 0, 0);
	}
	else if (visible)
	{
		const f4 e4 = p.wv.sh_rad[i];
		f4 *const dst = p.wv.rad_nee ? p.wv.rad_nee : p.wv.rad;
		f4 r = dst[slot];
		r.x += e4.x, r.y += e4.y, r.z += e4.z;
		dst[slot] = r;
	}
}
// Depth 0 and no path went on (connection_count() == 0: the reference's host loop traces no connections then): the slots the
// shade kernel left to the connection wave still have to start at zero.
// the path slot of a shadow ray's slot word (depth 0 with FrameView::shadow_bins: the light's bin sits above it)
RT_FN uint32_t shadow_slot(const Params &p, uint32_t word)
{
	return (p.depth == 0 && p.fr.shadow_bins) ? word & ((1u << shadow_slot_bits(p.fr.shadow_bins)) - 1u) : word;
}
RT_FN void connect_skip_item(const Params &p, uint32_t i)
{
	const f4 o4 = p.wv.sh_org[i];
	if (fbits(o4.w) != RAY_VOID)
		p.wv.rad_nee[shadow_slot(p, fbits(o4.w))] = mk4(0, 0, 0, 0);
}

template <bool COUNT>
RT_FN void connect_item(const Params &p, uint32_t i, bool active, Ctx &ctx)
{
	TStat st;
	st.inner = 0, st.tris = 0, st.lds = 0;
	if (active)
	{
		const f4 o4 = p.wv.sh_org[i], d4 = p.wv.sh_dir[i];
		Hit h;
		if (fbits(o4.w) == RAY_VOID) // void entry (a real shadow ray may carry tmax < 0: it is traced, hits nothing, and counts)
			active = false;
		else
			connect_finish(p, i, shadow_slot(p, fbits(o4.w)), !trace<true, COUNT>(p.sc, xyz(o4), xyz(d4), 1e-5f, d4.w, h, ctx.stk, st));
	}
	if (COUNT)
	{
		ctx.add64(&p.wv.counters->inner_shadow, st.inner);
		ctx.add64(&p.wv.counters->tris_shadow, st.tris);
		ctx.add64(&p.wv.counters->lds_shadow, st.lds);
		ctx.add64(&p.wv.counters->rays_shadow, active ? 1u : 0u);
	}
}

// rfwhip_kat: one of the shade kernel's functions on one record (rfwhip_abi.h: RFWHIP_KAT_*)
RT_FN void kat_item(const Params &p, int function, const float *in, float *out, uint32_t i, float *pot_cache)
{
	const float *r = in + (size_t)i * KAT_IN;
	float *o = out + (size_t)i * KAT_OUT;
	for (int k = 0; k < KAT_OUT; k++)
		o[k] = 0.0f;
	Shading sd;
	sd.color = mk3(r[0], r[1], r[2]), sd.absorption = mk3(r[3], r[4], r[5]);
	sd.p0 = fbits(r[6]), sd.p1 = fbits(r[7]), sd.p2 = fbits(r[8]);
	const f3 N = mk3(r[9], r[10], r[11]), wo = mk3(r[12], r[13], r[14]), wi = mk3(r[15], r[16], r[17]);
	switch (function)
	{
	case 0: // BSDFEval
	{
		const f3 e = bsdf_eval(sd, N, wo, wi, r[18], fbits(r[19]) != 0u);
		o[0] = e.x, o[1] = e.y, o[2] = e.z;
		break;
	}
	case 1: // BSDFPdf
		o[0] = bsdf_pdf(sd, N, wo, wi);
		break;
	case 2: // BSDFSample
	{
		f3 T, B, R = mk3(0, 0, 1);
		float pdf = 0.0f;
		create_tangent_space(N, T, B);
		bsdf_sample(sd, T, B, N, wo, R, pdf, r[20], r[21]);
		o[0] = R.x, o[1] = R.y, o[2] = R.z, o[3] = pdf;
		break;
	}
	case 3: // createTangentSpace
	{
		f3 T, B;
		create_tangent_space(N, T, B);
		o[0] = T.x, o[1] = T.y, o[2] = T.z, o[3] = B.x, o[4] = B.y, o[5] = B.z;
		break;
	}
	case 4: // PackNormal / UnpackNormal
	{
		const uint32_t pk = pack_normal(N);
		const f3 u = unpack_normal(pk);
		o[0] = ubits(pk), o[1] = u.x, o[2] = u.y, o[3] = u.z;
		break;
	}
	case 5: // RandomBarycentrics
	{
		const f3 b = random_barycentrics(r[20]);
		o[0] = b.x, o[1] = b.y, o[2] = b.z;
		break;
	}
	case 6: // RandomPointOnLight
	{
		float pick = 0, pdf = 0;
		f3 col = mk3(0, 0, 0);
		uint32_t chosen_light = 0u;
		const f3 P = random_point_on_light(p.sc, r[6], r[7], mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), pick, pdf, col, chosen_light, pot_cache);
		o[0] = P.x, o[1] = P.y, o[2] = P.z, o[3] = pick, o[4] = pdf, o[5] = col.x, o[6] = col.y, o[7] = col.z;
		break;
	}
	case 7: // LightPickProb
		o[0] = light_pick_prob(p.sc, (int)fbits(r[8]), mk3(r[9], r[10], r[11]), mk3(r[3], r[4], r[5]), mk3(r[0], r[1], r[2]));
		break;
	case 8: // blueNoiseSampler
		o[0] = p.cam.blue_noise ? blue_noise_sample(p.cam.blue_noise, (int)fbits(r[0]), (int)fbits(r[1]), (int)fbits(r[2]), (int)fbits(r[3])) : -1.0f;
		break;
	case 9: // WangHash, RandomFloat
	{
		uint32_t st = wang_hash(fbits(r[0]));
		o[0] = ubits(st);
		o[1] = random_float(st);
		o[2] = ubits(st);
		break;
	}
	case 10: // half -> float (material colours, uv scales): eight bit patterns per record
		for (int k = 0; k < KAT_OUT; k++)
			o[k] = half_to_float((uint16_t)fbits(r[k]));
		break;
	case 11: // fast_div (rt_types.h): four (n, d) pairs of 31-bit integers per record -> n / d
		for (int k = 0; k < 4; k++)
			o[k] = ubits(fast_div(fbits(r[2 * k]), make_fastdiv(fbits(r[2 * k + 1]))));
		break;
	case 12: // tex_wrap (rt_core.h): four (x, w) pairs per record, 0 <= x, 1 <= w -> x % w
		for (int k = 0; k < 4; k++)
			o[k] = ubits((uint32_t)tex_wrap((int)fbits(r[2 * k]), (int)fbits(r[2 * k + 1])));
		break;
	default:
		break;
	}
}

// CUDART/src/Context.cpp:109: the connections of shade call d are traced only if depth d + 1 has extension rays
// (`while (activePaths > 0 && ...)`); evaluated on the device, per wavefront batch — no host read-back.
RT_FN uint32_t connection_count(const WaveCounters *c, uint32_t depth)
{
	return c->ext[depth + 1] ? c->shadow_n[depth] : 0u; // (queue length: includes the void entries of unfinished blocks)
}

// Re-arms the per-call part of the counters; thread t of nt takes a strided share (one thread doing the ~500 dependent
// stores alone took 0.6-1.4 ms on a busy chip — at the head of every launch chain).
RT_FN void init_counters_item(WaveCounters *c, uint32_t primary_count, uint32_t t, uint32_t nt)
{
	for (uint32_t d = t; d < (uint32_t)MAX_DEPTH_SLOTS; d += nt)
	{
		c->ext[d] = c->ext_n[d] = d == 0u ? primary_count : 0u, c->shadow[d] = c->shadow_n[d] = 0u;
		if (c->t_last[d] > c->t_first[d]) // fold the previous call's extend-stage clock
		{
#if defined(__HIP_DEVICE_COMPILE__)
			atomicAdd(&c->ext_ticks, c->t_last[d] - c->t_first[d]);
			atomicAdd(&c->ext_timed, 1u);
#else
			c->ext_ticks += c->t_last[d] - c->t_first[d], c->ext_timed++;
#endif
		}
		c->t_first[d] = ~0ull, c->t_last[d] = 0ull;
	}
	uint32_t *const work = &c->work[0][0];
	for (uint32_t q = t; q < (uint32_t)WORK_QUEUES * 8u; q += nt)
		work[q] = 0u;
	if (t == 0u)
		c->probe_valid = 0u, c->stack_overflow = 0u;
}

// One pixel: its samples in sample order whatever the slot layout (rt_core.h: sample groups) — the image is independent of
// it.  The g samples of a group are consecutive 16-byte records; they are fetched eight at a time (one 128-byte line per
// lane and batch, the loads issued back to back) so that a line is consumed while it is in flight, not re-fetched.
RT_FN void resolve_item(const Params &p, uint32_t li)
{
	const uint32_t x = li % p.fr.W, yl = li / p.fr.W;
	if (local_to_global_row(p.fr, yl) >= p.fr.H)
		return;
	const uint32_t tile = (yl / TILE) * p.fr.tiles_x + x / TILE, pix = tile_pix(x % TILE, yl % TILE);
	f4 a = p.wv.acc[li];
	const uint32_t g = 1u << p.fr.sgroup_log2;
	for (uint32_t s0 = 0; s0 < p.fr.spp; s0 += g)
	{
		const unsigned long long base = pixel_to_slot(p.fr, tile, pix, s0);
		const f4 *const r = p.wv.rad + base, *const q = p.wv.rad_nee ? p.wv.rad_nee + base : nullptr;
		uint32_t i = 0;
		// (alpha < 0: a path that ended at depth 0 — no connection record was written for it, shade_pt_item)
		for (; i + 8u <= g; i += 8u)
		{
			f4 rv[8], qv[8];
			for (int k = 0; k < 8; k++)
				rv[k] = r[i + k];
			for (int k = 0; k < 8; k++)
				qv[k] = (q && !(rv[k].w < 0.0f)) ? q[i + k] : mk4(0, 0, 0, 0);
			for (int k = 0; k < 8; k++)
			{
				a.x += rv[k].x, a.y += rv[k].y, a.z += rv[k].z, a.w += fabsf(rv[k].w);
				if (q && !(rv[k].w < 0.0f))
					a.x += qv[k].x, a.y += qv[k].y, a.z += qv[k].z;
			}
		}
		for (; i < g; i++)
		{
			const f4 rv = r[i];
			a.x += rv.x, a.y += rv.y, a.z += rv.z, a.w += fabsf(rv.w);
			if (q && !(rv.w < 0.0f))
			{
				const f4 qv = q[i];
				a.x += qv.x, a.y += qv.y, a.z += qv.z;
			}
		}
	}
	p.wv.acc[li] = a;
}

RT_FN void present_item(const Params &p, f4 *out, float scale, int full, uint32_t li)
{
	const uint32_t x = li % p.fr.W, yl = li / p.fr.W;
	const uint32_t y = local_to_global_row(p.fr, yl);
	const f4 a = p.wv.acc[li];
	const f4 v = mk4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
	if (full)
	{
		if (y < p.fr.H)
			out[y * p.fr.W + x] = v;
	}
	else
		out[li] = y < p.fr.H ? v : mk4(0, 0, 0, 0);
}

RT_FN void deinterleave_item(const f4 *gathered, f4 *out, uint32_t W, uint32_t H, uint32_t local_rows, uint32_t world,
							 uint32_t gi)
{
	const uint32_t x = gi % W, y = gi / W;
	if (y >= H)
		return;
	const uint32_t strip = y / STRIP_ROWS, rank = strip_owner(strip, world);
	const uint32_t yl = (strip / world) * STRIP_ROWS + y % STRIP_ROWS;
	out[gi] = gathered[((unsigned long long)rank * local_rows + yl) * W + x];
}

// xor128 jump-ahead: jump_table[k] = M^(2^k) as 128 columns x 4 words
RT_FN void gf2_apply(const uint32_t *m, uint32_t s[4])
{
	uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
	for (int w = 0; w < 4; w++)
		for (int bit = 0; bit < 32; bit++)
			if ((s[w] >> bit) & 1u)
			{
				const uint32_t *c = m + 4 * (w * 32 + bit);
				r0 ^= c[0], r1 ^= c[1], r2 ^= c[2], r3 ^= c[3];
			}
	s[0] = r0, s[1] = r1, s[2] = r2, s[3] = r3;
}
constexpr uint32_t RNG_RUN = 32; // packets per rng_states thread
RT_FN void rng_states_item(uint32_t *states, const uint32_t base[4], const uint32_t *table, uint32_t total, uint32_t r)
{
	const unsigned long long start = (unsigned long long)r * RNG_RUN;
	if (start >= total)
		return;
	uint32_t s[4] = {base[0], base[1], base[2], base[3]};
	const unsigned long long draws = start * 32ull;
	for (int k = 0; k < 64; k++)
		if ((draws >> k) & 1ull)
			gf2_apply(table + 512ull * k, s);
	for (uint32_t j = 0; j < RNG_RUN && start + j < total; j++)
	{
		uint32_t *o = states + 4ull * (start + j);
		o[0] = s[0], o[1] = s[1], o[2] = s[2], o[3] = s[3];
		for (int d = 0; d < 32; d++)
			xor128_next(s);
	}
}

// ---- device skinning ------------------------------------------------------------------------------
// SceneMesh::set_pose (geometry/gltf/mesh.cpp:31-45): skinMatrix = sum_k w_k * jointMatrix[j_k]; vertex = skinMatrix *
// base; normal = normalize(baseNormal * inverse(skinMatrix)) — a row vector times the inverse, i.e. the
// inverse-transpose of the upper 3x3 applied to the normal.
RT_FN void skin_vertex_item(f4 *verts, f4 *vnormals, const f4 *base_verts, const f4 *base_normals, const uint32_t *joints4,
							const f4 *weights4, const float *mats, uint32_t joint_count, uint32_t i)
{
	const f4 w4 = weights4[i];
	const float w[4] = {w4.x, w4.y, w4.z, w4.w};
	float m[16];
	for (int e = 0; e < 16; e++)
		m[e] = 0.0f;
	for (int k = 0; k < 4; k++)
	{
		uint32_t j = joints4[4u * i + k];
		if (j >= joint_count)
			j = 0;
		const float *mj = mats + 16u * j;
		for (int e = 0; e < 16; e++)
			m[e] += mj[e] * w[k];
	}
	const f4 b = base_verts[i];
	// column-major: element (row r, column c) = m[c * 4 + r]
	verts[i] = mk4(m[0] * b.x + m[4] * b.y + m[8] * b.z + m[12] * b.w, m[1] * b.x + m[5] * b.y + m[9] * b.z + m[13] * b.w,
				   m[2] * b.x + m[6] * b.y + m[10] * b.z + m[14] * b.w, m[3] * b.x + m[7] * b.y + m[11] * b.z + m[15] * b.w);
	// inverse-transpose of the upper 3x3 = cofactor matrix / determinant
	const float a00 = m[0], a01 = m[4], a02 = m[8], a10 = m[1], a11 = m[5], a12 = m[9], a20 = m[2], a21 = m[6], a22 = m[10];
	const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
	const float c10 = a02 * a21 - a01 * a22, c11 = a00 * a22 - a02 * a20, c12 = a01 * a20 - a00 * a21;
	const float c20 = a01 * a12 - a02 * a11, c21 = a02 * a10 - a00 * a12, c22 = a00 * a11 - a01 * a10;
	const float det = a00 * c00 + a01 * c01 + a02 * c02;
	const float id = 1.0f / det;
	const f4 n = base_normals[i];
	const f3 r = mk3((c00 * n.x + c01 * n.y + c02 * n.z) * id, (c10 * n.x + c11 * n.y + c12 * n.z) * id,
					 (c20 * n.x + c21 * n.y + c22 * n.z) * id);
	const f3 rn = r * (1.0f / length(r));
	vnormals[i] = mk4(rn.x, rn.y, rn.z, 0.0f);
}
// SceneMesh::set_pose(weights) (geometry/gltf/mesh.cpp:127-147): base + sum_j w_j * target_j for positions (w stays 1) and
// normals (not renormalised there either).  targets: [target][vertex] float4 positions, then the same for normals.
RT_FN void morph_vertex_item(f4 *verts, f4 *vnormals, const f4 *base_verts, const f4 *base_normals, const f4 *tgt_pos,
							 const f4 *tgt_nrm, const float *weights, uint32_t target_count, uint32_t vertex_count, uint32_t i)
{
	const f4 b = base_verts[i], bn = base_normals[i];
	f3 p = xyz(b), n = xyz(bn);
	for (uint32_t j = 0; j < target_count; j++)
	{
		const float w = weights[j];
		p = p + xyz(tgt_pos[(size_t)j * vertex_count + i]) * w;
		n = n + xyz(tgt_nrm[(size_t)j * vertex_count + i]) * w;
	}
	verts[i] = mk4(p.x, p.y, p.z, 1.0f);
	vnormals[i] = mk4(n.x, n.y, n.z, 0.0f);
}
// SceneMesh::update_triangles (mesh.cpp:428-485) on the shading record: vN0..2 and N = normalize(cross(v1-v0, v2-v0))
RT_FN void skin_shade_item(TriShade *shade, const f4 *verts, const f4 *vnormals, const uint32_t *indices, uint32_t i)
{
	uint32_t a, b, c;
	if (indices)
		a = indices[3u * i], b = indices[3u * i + 1u], c = indices[3u * i + 2u];
	else
		a = 3u * i, b = a + 1u, c = a + 2u;
	const f3 v0 = xyz(verts[a]), v1 = xyz(verts[b]), v2 = xyz(verts[c]);
	const f3 N = normalize_ieee(cross(v1 - v0, v2 - v0));
	const f4 n0 = vnormals[a], n1 = vnormals[b], n2 = vnormals[c];
	TriShade &t = shade[i];
	t.n0 = mk4(n0.x, n0.y, n0.z, N.x);
	t.n1 = mk4(n1.x, n1.y, n1.z, N.y);
	t.n2 = mk4(n2.x, n2.y, n2.z, N.z);
}

// after a refit: re-quantise the child boxes of one compressed node from the refitted BVH2 boxes (src4: the BVH2 node, BLAS-
// relative, each child box came from; 0xFFFFFFFF = unused slot)
RT_FN void refresh4_item(Node4c *nodes4, const uint32_t *src4, const Node *nodes2, uint32_t i)
{
	float lo[3][4], hi[3][4];
	bool valid[4];
	for (int k = 0; k < 4; k++)
	{
		const uint32_t src = src4[4u * i + k];
		valid[k] = src != 0xFFFFFFFFu;
		for (int a = 0; a < 3; a++)
			lo[a][k] = valid[k] ? nodes2[src].bmin[a] : 0.0f, hi[a][k] = valid[k] ? nodes2[src].bmax[a] : 0.0f;
	}
	pack_boxes4c(nodes4[i], lo, hi, valid);
}

// The float form of one compressed node (rt_types.h: Node4f): exactly the planes the per-lane traversal decodes,
// plane = q * scale + org, for the packet traversal's scalar fetches.  Runs over the whole table at the end of rfwhip_update.
RT_FN void expand4_item(const Node4c *nodes4, Node4f *out, uint32_t i)
{
	const Node4c n = nodes4[i];
	const float scale[3] = {n.scale_x, n.scale_y, n.scale_z};
	Node4f f;
	for (int k = 0; k < 4; k++)
	{
		bool empty = n.entry[k] == ENTRY_EMPTY;
		for (int a = 0; a < 3; a++)
			empty = empty || ((n.qlo[a] >> (8 * k)) & 255u) > ((n.qhi[a] >> (8 * k)) & 255u);
		for (int a = 0; a < 3; a++)
		{
			const float ql = (float)((n.qlo[a] >> (8 * k)) & 255u), qh = (float)((n.qhi[a] >> (8 * k)) & 255u);
			f.lo[a][k] = empty ? 1e30f : fmaf(ql, scale[a], n.org[a]);
			f.hi[a][k] = empty ? -1e30f : fmaf(qh, scale[a], n.org[a]);
		}
		f.entry[k] = n.entry[k], f.pad[k] = 0u;
	}
	out[i] = f;
}

// refit, pass 1: rewrite the leaf-ordered triangle vertices from the new mesh vertices
RT_FN void refit_tris_item(f4 *tri_verts, const f4 *verts, const uint32_t *indices, uint32_t slot)
{
	const uint32_t prim = fbits(tri_verts[3ull * slot].w);
	uint32_t i0 = 3u * prim, i1 = i0 + 1u, i2 = i0 + 2u;
	if (indices)
		i0 = indices[3ull * prim], i1 = indices[3ull * prim + 1], i2 = indices[3ull * prim + 2];
	const f4 a = verts[i0], b = verts[i1], c = verts[i2];
	tri_verts[3ull * slot] = mk4(a.x, a.y, a.z, ubits(prim));
	tri_verts[3ull * slot + 1] = mk4(b.x, b.y, b.z, 1.0f);
	tri_verts[3ull * slot + 2] = mk4(c.x, c.y, c.z, TRI_EPS); // (w: the triangle's determinant threshold, tri_test)
}
RT_FN void leaf_bounds(const Node &n, const f4 *tri_verts, float mn[3], float mx[3])
{
	mn[0] = mn[1] = mn[2] = 1e34f, mx[0] = mx[1] = mx[2] = -1e34f;
	const uint32_t first = (uint32_t)n.left_first & ENTRY_FIRST_MASK; // device nodes carry packed entries
	for (int k = 0; k < n.count; k++)
		for (int v = 0; v < 3; v++)
		{
			const f4 q = tri_verts[3ull * (first + (uint32_t)k) + v];
			mn[0] = fminf(mn[0], q.x), mn[1] = fminf(mn[1], q.y), mn[2] = fminf(mn[2], q.z);
			mx[0] = fmaxf(mx[0], q.x), mx[1] = fmaxf(mx[1], q.y), mx[2] = fmaxf(mx[2], q.z);
		}
	// per-triangle boxes are grown by 1e-5 (bvh_tree.cpp:412), node boxes once more (bvh_node.h:218-219)
	for (int a = 0; a < 3; a++)
		mn[a] -= 2e-5f, mx[a] += 2e-5f;
}

// ================================================================================================================
#if defined(RT_DEVICE_BUILD)
// ================================================================================================================

static int g_cus = 256;
void set_device_cus(int cus) { g_cus = cus > 0 ? cus : 256; }
uint32_t max_lds_nodes() { return MAX_LDS_NODES; }

// XCD-aware dynamic chunk queue of a persistent grid.  Chunk = 256 consecutive items.  Chunks are dealt to the 8
// XCDs in groups of `group` consecutive chunks (for the primary wave: one row of 8x8 tiles), so each XCD owns every
// 8th tile row — balanced whatever the image content — and the workgroups of one XCD (workgroup b runs on XCD b % 8)
// pull consecutive chunks of that XCD's sequence from one device-scope counter: at any time they work on
// neighbouring tiles whose BVH nodes and triangles meet in that XCD's 4 MiB L2.  The mapping only affects speed.
struct ChunkQueue
{
	uint32_t xcd, nchunks, group;
	uint32_t *head;
	__device__ __forceinline__ ChunkQueue(const Params &p, uint32_t count)
	{
		nchunks = (count + BLOCK - 1) / BLOCK;
		xcd = blockIdx.x & 7u;
		group = p.group ? p.group : 16u;
		head = &p.wv.counters->work[p.queue][xcd];
	}
	// workgroup-uniform; false when this XCD's share is exhausted
	__device__ __forceinline__ bool next(uint32_t &c)
	{
		__shared__ uint32_t s_q;
		__syncthreads();
		if (threadIdx.x == 0)
			s_q = atomicAdd(head, 1u);
		__syncthreads();
		const uint32_t q = s_q;
		const uint32_t g = q / group, w = q - g * group;
		const uint32_t base = (g * 8u + xcd) * group;
		c = base + w;
		return base < nchunks;
	}
};

// Workgroup size of the persistent-lane kernels of the incoherent waves (k_trace_stream).  Their LDS = stack + ONE copy of the
// top-of-tree cache per workgroup: a larger workgroup shares that copy among more waves, so more waves fit a CU
// (256 threads: 26 KiB -> 6 workgroups = 24 waves per CU; 1024 threads: 62 KiB -> 2 workgroups = 32 waves per CU).
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 256
#endif
constexpr int TRACE_BLOCK = RT_TRACE_BLOCK;
#define RT_STACK_DECL_N(DEPTH, NTHREADS)                                                     \
	__shared__ uint32_t s_stack[(DEPTH) * (NTHREADS)];                                      \
	__shared__ f4 s_top[MAX_LDS_NODES * TOP_ROWS];                                          \
	uint32_t spill_[SPILL_STACK];                                                           \
	Ctx ctx;                                                                                \
	ctx.stk.lds = s_stack + threadIdx.x;                                                    \
	ctx.stk.stride = (NTHREADS);                                                            \
	ctx.stk.spill = spill_;                                                                 \
	stage_top<NTHREADS>(p, s_top);                                                          \
	ctx.stk.top = s_top, ctx.stk.top_first = p.lds_first, ctx.stk.top_count = p.lds_count;            \
	ctx.stk.overflow = &p.wv.counters->stack_overflow;
#define RT_STACK_DECL_(DEPTH) RT_STACK_DECL_N(DEPTH, BLOCK)
#define RT_STACK_DECL_CLOSEST RT_STACK_DECL_(LDS_STACK)
#define RT_STACK_DECL_ANY RT_STACK_DECL_(LDS_STACK_ANY)

// every workgroup copies the top-of-tree rows into its LDS once (the grids are persistent)
template <int NTHREADS> __device__ __forceinline__ void stage_top(const Params &p, f4 *s_top)
{
	const f4 *src = (const f4 *)(p.sc.nodes4 + p.lds_first);
	const uint32_t rows = p.lds_count * TOP_ROWS;
	for (uint32_t i = threadIdx.x; i < rows; i += NTHREADS)
		s_top[i] = src[i];
	__syncthreads();
}

__device__ __forceinline__ uint32_t wave_prefix(unsigned long long mask)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// first workgroup in / last workgroup out of an extend-stage launch (WaveCounters::t_first / t_last)
__device__ __forceinline__ void clock_in(WaveCounters *wc, uint32_t depth)
{
	if (threadIdx.x == 0)
		atomicMin(&wc->t_first[depth], (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void clock_out(WaveCounters *wc, uint32_t depth)
{
	if (threadIdx.x == 0)
		atomicMax(&wc->t_last[depth], (unsigned long long)wall_clock64());
}

template <int GEN, bool COUNT>
__global__ void __launch_bounds__(BLOCK, RT_PRIMARY_WAVES) k_extend(const Params p, const uint32_t fixed_count)
{
	clock_in(p.wv.counters, p.depth);
	RT_STACK_DECL_CLOSEST
	const uint32_t count = (GEN == GEN_BUFFER || GEN == GEN_RANGED) ? p.wv.counters->ext_n[p.depth] : fixed_count;
	ChunkQueue w(p, count);
	uint32_t c;
	while (w.next(c))
	{
		const uint32_t i = c * BLOCK + threadIdx.x;
		if (c < w.nchunks)
			extend_item<GEN, COUNT>(p, i, i < count, ctx);
	}
	clock_out(p.wv.counters, p.depth);
}

// ----------------------------------------------------------------------------------------------------------------
// Persistent-lane traversal for the incoherent waves (extension rays of depth >= 1, shadow rays).
// A wave keeps 64 traversals in flight; as soon as RT_REFILL_IDLE lanes have finished their ray, those lanes pull the
// next rays from the launch's queue (one wave-aggregated atomicAdd) while the others continue — the wave's SIMD lanes
// stay busy although rays of one wave need very different numbers of steps (measured lane utilisation of the
// one-ray-per-lane form on the bounce waves: 18-30 %).
// ----------------------------------------------------------------------------------------------------------------
// Three knobs, swept together on the MI355X (terrain_1002k, 128 spp per step; Msamples/s):
//   refill threshold (idle lanes)   leaf vote   run length      result
//   52 / 56                         64          per refill      2208   (round 1: every refill = one atomic on the queue head)
//   32                              64          per refill      2153   (eager refills alone lose: the queue atomic stalls the wave)
//   52 / 56                         32          per refill      2367
//   32                              32          512             2514
//   24 / 40 / 48                    32          512             2520 / 2451 / 2425
//   32                              24 / 48     512             2503 / 2266
//   32                              32          1024 / 2048     2512 / 2493
#ifndef RT_REFILL_IDLE_EXT
#define RT_REFILL_IDLE_EXT 32 // extension rays: idle lanes pull new rays once 32 of the wave's 64 lanes are idle
#endif
#ifndef RT_REFILL_IDLE_ANY
#define RT_REFILL_IDLE_ANY 32 // shadow rays
#endif

// the node phase of a wave ends early once this many of its lanes hold a leaf (64: only when none is on an inner node)
#ifndef RT_LEAF_VOTE_EXT
#define RT_LEAF_VOTE_EXT 40 // (of 64: the share of the wave's lanes WITH a ray that must hold a leaf, rt_core.h RT_VOTE_RELATIVE)
#endif
// a single-sample primary launch (sample groups of 1: a wave is an 8x8 tile of different pixels) takes the packet form only when
// it is large; small ones keep the one-ray-per-lane kernel with its tile-row-to-XCD dealing (launch_extend)
#ifndef RT_PRIMARY_PACKET_MIN
#define RT_PRIMARY_PACKET_MIN (16u << 20)
#endif
// rays a wave takes from the launch's queue per atomic (0: one atomic per refill, exactly the idle lanes).  Round 5, per kernel
// (MI355X, serialised ms per 64-spp sub-batch at 128 / 256 / 512 / 1024): extension rays of depth 1 6.09 / 5.96 / 6.02 / 6.27, of
// depth 2 1.72 / 1.72 / 1.75 / 1.76, shadow rays 9.24 / 7.49 / 7.72 / 7.88 — short runs spread a queue's neighbourhoods over more
// waves while their nodes are still in the L2s, too short ones pay in atomics.  The packet kernel's runs are its own
// (RT_PACKET_CHUNK).
#ifndef RT_STREAM_CHUNK
#define RT_STREAM_CHUNK 256
#endif
#ifndef RT_STREAM_CHUNK_ANY
#define RT_STREAM_CHUNK_ANY RT_STREAM_CHUNK
#endif
#ifndef RT_LEAF_VOTE_ANY
#define RT_LEAF_VOTE_ANY 40
#endif
#ifndef RT_REFILL_PIN
#define RT_REFILL_PIN 1
#endif




// MODE: where a lane's next ray comes from — the extension-ray buffers of this depth or the shadow-ray buffers.  (The pt
// integrator's primary wave had a persistent-lane form of its own, k_primary_stream, until the packet form of round 4 replaced
// it; round 5 took it out of the sources: DESIGN_LOG.md, "variants removed".)
enum
{
	STREAM_EXT = 0,
	STREAM_ANY = 1
};

// RT_DIAG_TRACE_CLOCK (development builds): a traversal wave's cycles by phase of its loop — refill, node phase, leaf phase,
// retirement — kept in scalar registers (s_memtime), added to a global table when the wave leaves.
#if defined(RT_DIAG_TRACE_CLOCK)
__device__ unsigned long long g_trace_clk[3][8];
#define RT_TRACE_TICK(K)                                           \
	{                                                              \
		const unsigned long long t_ = __builtin_readcyclecounter(); \
		clk_acc[K] += t_ - clk_last, clk_n[K]++;                    \
		clk_last = __builtin_readcyclecounter();                    \
	}
#else
#define RT_TRACE_TICK(K)
#endif
template <int MODE, bool COUNT> __device__ __forceinline__ void stream_rays(const Params &p, const uint32_t count, Ctx &ctx)
{
	constexpr bool ANY = MODE == STREAM_ANY;
#if defined(RT_DIAG_TRACE_CLOCK)
	unsigned long long clk_acc[4] = {0, 0, 0, 0}, clk_n[4] = {0, 0, 0, 0}, clk_last = __builtin_readcyclecounter();
#endif
	WaveCounters *const wc = p.wv.counters;
	uint32_t *const head = &wc->work[p.queue][0];
	const uint32_t b = p.depth & 1u;
	const f4 *const ray_o = ANY ? p.wv.sh_org : p.wv.org[b];
	const f4 *const ray_d = ANY ? p.wv.sh_dir : p.wv.dir[b];
	// (the queue-fed waves re-read a ray's record on the rare instance switch instead of keeping the world-space ray)
	constexpr bool WORLD = false;
	Traverser<ANY, COUNT, WORLD> T;
	T.cur = ENTRY_DONE;
	TStat st;
	st.inner = 0, st.tris = 0, st.lds = 0;
	uint32_t nrays = 0;
	const uint32_t lane = __lane_id();
	bool has_ray = false, exhausted = false;
	uint32_t ray = 0, slot = 0;
#if RT_STREAM_CHUNK
	uint32_t q_next = 0, q_end = 0; // wave-uniform: the rest of the run this wave owns
	// run length: RT_STREAM_CHUNK for big launches, down to 64 when the launch has fewer than ~4 runs per wave
	uint32_t run = count / (gridDim.x * (blockDim.x / 64u) * 4u);
	constexpr uint32_t RUN_MAX = MODE == STREAM_ANY ? (uint32_t)RT_STREAM_CHUNK_ANY : (uint32_t)RT_STREAM_CHUNK;
	run = run > RUN_MAX ? RUN_MAX : (run < 64u ? 64u : run);
#endif
	uint32_t REFILL = MODE == STREAM_ANY ? RT_REFILL_IDLE_ANY : RT_REFILL_IDLE_EXT;
	// Waves of near-identical rays — the slot layout's sample groups put >= 8 samples of a pixel side by side (rt_core.h), so a
	// wave of the shadow rays the first vertices emit walks the same nodes in step — are refilled only as a whole: a partial
	// refill mixes rays at different stages into the wave and the two phases of the traversal fall out of step again (MI355X,
	// 32 samples per group: depth-0 shadow wave 6.40 -> 5.63 ms per 32-spp launch; the vote threshold stays: 48 / 64 lose 2-7 %)
#ifndef RT_REFILL_ANY0
#define RT_REFILL_ANY0 48u
#endif
	if (MODE == STREAM_ANY && p.depth == 0 && p.fr.sgroup_log2 >= 3u)
		REFILL = RT_REFILL_ANY0;
	constexpr int VOTE = MODE == STREAM_ANY ? RT_LEAF_VOTE_ANY : RT_LEAF_VOTE_EXT;
	for (;;)
	{
		const unsigned long long idle_mask = __ballot(!has_ray);
		const uint32_t nidle = (uint32_t)__popcll(idle_mask);
		if (!exhausted && nidle >= REFILL)
		{
#if RT_STREAM_CHUNK
			// the wave owns a run of consecutive rays at a time: one queue atomic per run, refills in between are wave-local
			if (q_next == q_end)
			{
				uint32_t g = 0;
				if (lane == 0)
					g = atomicAdd(head, run);
				g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
				q_next = g < count ? g : count;
				q_end = g + run < count ? g + run : count;
			}
			const uint32_t base = q_next;
			const uint32_t take = nidle < q_end - q_next ? nidle : q_end - q_next;
			q_next += take;
			const uint32_t limit = base + take;
#else
			const uint32_t leader = (uint32_t)__ffsll((long long)idle_mask) - 1u;
			uint32_t base = 0;
			if (lane == leader)
				base = atomicAdd(head, nidle);
			base = __shfl(base, (int)leader);
			const uint32_t limit = count;
#endif
			if (!has_ray)
			{
				const uint32_t idx = base + wave_prefix(idle_mask);
				if (idx < limit)
				{
					{
#if RT_REFILL_PIN
						// both records of the ray in ONE round trip: left alone, the compiler fetches the slot word first, tests it
						// for RAY_VOID and only then asks for the rest — two dependent waits per refill for the whole wave
						typedef float v4f_ __attribute__((ext_vector_type(4)));
						v4f_ oa = *(const v4f_ *)(ray_o + idx), da = *(const v4f_ *)(ray_d + idx);
						asm volatile("" : "+v"(oa), "+v"(da));
						const f4 o4 = mk4(oa.x, oa.y, oa.z, oa.w), d4 = mk4(da.x, da.y, da.z, da.w);
#else
						const f4 o4 = ray_o[idx], d4 = ray_d[idx];
#endif
						// void entries (the unfilled rest of a shade wave's last queue block) are skipped
						if (fbits(o4.w) == RAY_VOID)
						{
							if (!ANY)
								p.wv.hit[idx] = mk4(0, 0, 0, ubits((uint32_t)HIT_VOID));
						}
						else
						{
							// shadow rays: (epsilon, dist - 2 epsilon) (Kernels.cu:750, :486); extension rays: (1e-5, 1e34)
							T.begin(p.sc, xyz(o4), xyz(d4), 1e-5f, ANY ? d4.w : 1e34f);
							has_ray = true, ray = idx, slot = ANY ? shadow_slot(p, fbits(o4.w)) : fbits(o4.w), nrays++;
						}
					}
				}
			}
#if RT_STREAM_CHUNK
			exhausted = q_next >= count;
#else
			exhausted = base + nidle >= count;
#endif
		}
		if (__ballot(has_ray) == 0ull)
		{
			if (exhausted)
				break;
			continue; // (primary slots outside the image leave their lanes idle: take the next ones)
		}
		RT_TRACE_TICK(0)
		T.template descend<VOTE>(p.sc, ctx.stk, st);
		RT_TRACE_TICK(1)
		const auto world = [&](f3 &O, f3 &D) {
			const f4 o4 = ray_o[ray], d4 = ray_d[ray];
			O = xyz(o4), D = xyz(d4);
		};
		if (has_ray)
			T.visit(p.sc, ctx.stk, st, world);
		RT_TRACE_TICK(2)
		if (has_ray)
		{
			if (T.done())
			{
				if (ANY)
					connect_finish(p, ray, slot, T.hit.prim < 0);
				else
				{
					p.wv.hit[ray] = mk4(T.hit.t, T.hit.u, T.hit.v, ubits((uint32_t)T.hit.prim));
					p.wv.hit_inst[ray] = T.hit.inst;
				}
				has_ray = false;
			}
		}
		RT_TRACE_TICK(3)
	}
#if defined(RT_DIAG_TRACE_CLOCK)
	if (lane == 0u)
		for (int k = 0; k < 4; k++)
			atomicAdd(&g_trace_clk[MODE == STREAM_ANY ? 1 : 0][k], clk_acc[k]), atomicAdd(&g_trace_clk[MODE == STREAM_ANY ? 1 : 0][4 + k], clk_n[k]);
#endif
	if (COUNT)
	{
		ctx.add64(ANY ? &wc->inner_shadow : &wc->inner_extend, st.inner);
		ctx.add64(ANY ? &wc->tris_shadow : &wc->tris_extend, st.tris);
		ctx.add64(ANY ? &wc->lds_shadow : &wc->lds_extend, st.lds);
		ctx.add64(ANY ? &wc->rays_shadow : &wc->rays_extend, nrays);
	}
}

template <bool ANY, bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK, ANY ? RT_ANY_WAVES : RT_TRACE_WAVES) __attribute__((amdgpu_num_vgpr(RT_TRACE_VGPRS))) k_trace_stream(const Params p)
{
	const uint32_t count = ANY ? connection_count(p.wv.counters, p.depth) : p.wv.counters->ext_n[p.depth];
	if (count == 0u)
	{
		if (ANY && p.depth == 0 && p.wv.rad_nee)
			for (uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x, n = p.wv.counters->shadow_n[0]; i < n; i += gridDim.x * TRACE_BLOCK)
				connect_skip_item(p, i);
		return;
	}
	if (!ANY)
		clock_in(p.wv.counters, p.depth);
	RT_STACK_DECL_N(ANY ? LDS_STACK_ANY : LDS_STACK, TRACE_BLOCK)
	stream_rays<ANY ? STREAM_ANY : STREAM_EXT, COUNT>(p, count, ctx);
	if (!ANY)
		clock_out(p.wv.counters, p.depth);
}

// Extension rays of depth d + 1 and shadow rays of depth d in ONE launch: both queues are complete when the shade kernel of
// depth d has finished, and as two launches each ends in its own tail — a few long rays keep a handful of waves busy while
// the rest of the chip waits for the kernel boundary.  A wave walks the extension queue first (the next shade kernel waits
// for those hits) and moves on to the shadow queue when the extension queue has run dry, so the waves that finish early fill
// the other queue's work instead of idling: one tail per depth instead of two, and 4 launches less per frame of depth 2.
// Registers and LDS are those of the closest-hit kernel (the occlusion traversal uses the first 8 levels of its stack).
template <bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK, RT_TRACE_WAVES) __attribute__((amdgpu_num_vgpr(RT_TRACE_VGPRS))) k_trace_fused(const Params pe, const Params pa)
{
	const uint32_t count_e = pe.wv.counters->ext_n[pe.depth];
	const uint32_t count_a = connection_count(pa.wv.counters, pa.depth);
	if (count_a == 0u && pa.depth == 0 && pa.wv.rad_nee)
		for (uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x, n = pa.wv.counters->shadow_n[0]; i < n; i += gridDim.x * TRACE_BLOCK)
			connect_skip_item(pa, i);
	if (count_e == 0u && count_a == 0u)
		return;
	const Params &p = pe; // (the stack declaration reads the LDS node range from `p`)
	RT_STACK_DECL_N(LDS_STACK, TRACE_BLOCK)
	// (whose time is it?  every workgroup adds what it spent on either kind of ray to WaveCounters::fused_ticks: two atomics per
	// workgroup and launch; the host splits the launch's duration by the sums — rfwhip_wait)
	const unsigned long long t0 = threadIdx.x == 0 ? (unsigned long long)wall_clock64() : 0ull;
	if (count_e) // (the device clock of the extend stage covers extension rays only: no near-empty span for a launch without any)
	{
		clock_in(pe.wv.counters, pe.depth);
		stream_rays<STREAM_EXT, COUNT>(pe, count_e, ctx);
		clock_out(pe.wv.counters, pe.depth);
	}
	const unsigned long long t1 = threadIdx.x == 0 ? (unsigned long long)wall_clock64() : 0ull;
	if (count_a)
		stream_rays<STREAM_ANY, COUNT>(pa, count_a, ctx);
	if (threadIdx.x == 0)
	{
		const unsigned long long t2 = (unsigned long long)wall_clock64();
		if (t1 > t0)
			atomicAdd(&pe.wv.counters->fused_ticks[pe.depth][0], t1 - t0);
		if (t2 > t1)
			atomicAdd(&pe.wv.counters->fused_ticks[pe.depth][1], t2 - t1);
	}
}

// ----------------------------------------------------------------------------------------------------------------
// Wave-uniform ("packet") closest-hit traversal for waves of near-identical rays: the pt primary wave, where the slot
// layout's sample groups make a wave the g samples of 64 / g neighbouring pixels (g = 64: ONE pixel; g = 1: one 8x8 tile).
// Such a wave walks the same nodes anyway, so it walks them ONCE: one node index for the wave, the node's float planes
// (rt::Node4f) and a leaf's triangles fetched through SCALAR loads into SGPRs, one stack for the wave (a VGPR used as a
// 64-entry array: v_writelane / v_readlane), the children ordered on the scalar unit by the entry distance of the first lane
// that hits each.  What stays per lane is the arithmetic that differs per lane: six fmas + max3 / min3 + one compare per
// child, and the triangle test.  Against the per-lane form (Traverser::node_step: 24 byte -> float conversions, the frame of
// the node, 6 selects for the direction signs, a 5-comparator sort of (distance, entry) pairs, LDS pushes and pops: ~117 VALU
// instructions per node step in a kernel that is bound by VALU issue) a node step is ~50 VALU instructions, with about as
// many scalar ones beside them; the wave pays for every node ANY of its lanes visits, which for the rays of one pixel is
// hardly more than one ray's.
// Every ray still gets exactly its closest hit: a lane tests every triangle of every leaf the wave reaches with the same
// tri_test() on its own ray — a superset of the leaves its own traversal would reach — and culls with its own hit distance.
// (Which of two triangles hit at bit-identical distance is reported can depend on the order leaves are reached in, as it
// does between any two traversal orders.)  Direction signs pick the near / far plane rows per wave (row offsets, no selects);
// a wave whose lanes disagree about a sign on some axis takes min / max per plane pair instead (MIXED).  Instances: the
// transform is wave-uniform, every lane transforms its ray; the sentinel restores the world-space ray.
// ----------------------------------------------------------------------------------------------------------------
#define RT_CONST_AS __attribute__((address_space(4)))
typedef float pk_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t pk_v4u __attribute__((ext_vector_type(4)));
// a 16-byte load whose address is wave-uniform: from the constant address space the compiler selects s_load_dwordx4
__device__ __forceinline__ pk_v4f sload4(const void *base, uint32_t off)
{
	return *(const RT_CONST_AS pk_v4f *)((const char *)base + off);
}
__device__ __forceinline__ pk_v4u sload4u(const void *base, uint32_t off)
{
	return *(const RT_CONST_AS pk_v4u *)((const char *)base + off);
}
__device__ __forceinline__ uint32_t sload1u(const void *base, uint32_t off)
{
	return *(const RT_CONST_AS uint32_t *)((const char *)base + off);
}

// v_writelane_b32 (this clang has no __builtin for it; the intrinsic is reached by its name)
extern "C" __device__ int rt_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// 1 when k is not all ones, else 0 — on the scalar unit (written as a compare, the compiler keeps the wave-uniform bool in a lane
// mask and adds it up on the VALU)
__device__ __forceinline__ uint32_t packet_flag(uint32_t k)
{
	uint32_t r;
	asm("s_cmp_lg_u32 %1, -1\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(k) : "scc");
	return r;
}
// The six plane rows of a float node and its entries: seven scalar loads with the node's byte offset as the SGPR offset of
// wave-constant row pointers (no address arithmetic; the compiler adds base and offset as 64-bit integers first).  The wait is
// part of the block — the compiler does not track loads it did not issue.
struct PacketRows
{
	pk_v4f nx, ny, nz, fx, fy, fz;
	pk_v4u ent;
};
__device__ __forceinline__ void packet_load_rows(PacketRows &r, const char *const rn[3], const char *const rf[3], const char *nodes, uint32_t nb)
{
	asm volatile("s_load_dwordx4 %0, %7, %14\n\t"
				 "s_load_dwordx4 %1, %8, %14\n\t"
				 "s_load_dwordx4 %2, %9, %14\n\t"
				 "s_load_dwordx4 %3, %10, %14\n\t"
				 "s_load_dwordx4 %4, %11, %14\n\t"
				 "s_load_dwordx4 %5, %12, %14\n\t"
				 "s_load_dwordx4 %6, %13, %14 offset:0x60\n\t"
				 "s_waitcnt lgkmcnt(0)"
				 : "=&s"(r.nx), "=&s"(r.ny), "=&s"(r.nz), "=&s"(r.fx), "=&s"(r.fy), "=&s"(r.fz), "=&s"(r.ent)
				 : "s"(rn[0]), "s"(rn[1]), "s"(rn[2]), "s"(rf[0]), "s"(rf[1]), "s"(rf[2]), "s"(nodes), "s"(nb)
				 : "memory");
}

struct PacketStack
{
	// ONE stack for the wave: entry j lives in lane j of a VGPR.  Lane 0 holds ENTRY_DONE for good, so a pop never has to ask
	// whether the stack is empty; 63 entries above it (rfwhip_update computes every tree's worst-case need and the host launches
	// the packet kernel only for scenes within PACKET_STACK).  All arithmetic on sp is integer min / max / add: a wave-uniform
	// bool would live in a lane mask and drag the selects onto the VALU.
	int s0;
	uint32_t sp; // wave-uniform: number of entries including the sentinel
	__device__ __forceinline__ PacketStack() : s0((int)ENTRY_DONE), sp(1u) {}
	__device__ __forceinline__ void push(uint32_t e)
	{
		s0 = rt_writelane((int)e, (int)sp, s0);
		sp++;
	}
	__device__ __forceinline__ uint32_t pop()
	{
		sp--;
		return (uint32_t)__builtin_amdgcn_readlane(s0, (int)sp);
	}
	// up to three entries, each only if its count is 1: the three writes are unconditional (an unwanted entry lands on the
	// slot the next wanted one overwrites, or above the new top)
	__device__ __forceinline__ void push3(uint32_t ea, uint32_t na, uint32_t eb, uint32_t nb, uint32_t ec, uint32_t nc)
	{
		s0 = rt_writelane((int)ea, (int)sp, s0);
		s0 = rt_writelane((int)eb, (int)(sp + na), s0);
		s0 = rt_writelane((int)ec, (int)(sp + na + nb), s0);
		sp += na + nb + nc;
	}
	// the same with the KEYS of the three entries (all ones: not wanted) and the stack pointer in m0: a compare sets SCC, an add
	// with carry takes it — 7 scalar instructions instead of 12 (round 6, late: the packet kernels are bound by the scalar unit
	// as much as by the VALUs — one scalar instruction per SIMD every 4.5 clocks, tools/dev/micro/inst_rate5.hip).  One scalar
	// instruction sits between every write of m0 and the v_writelane that selects its lane with it.
	__device__ __forceinline__ void push3_keys(uint32_t ea, uint32_t ka, uint32_t eb, uint32_t kb, uint32_t ec, uint32_t kc)
	{
		asm volatile("s_mov_b32 m0, %1\n\t"
					 "s_cmp_lg_u32 %3, -1\n\t"
					 "v_writelane_b32 %0, %2, m0\n\t"
					 "s_addc_u32 m0, m0, 0\n\t"
					 "s_cmp_lg_u32 %5, -1\n\t"
					 "v_writelane_b32 %0, %4, m0\n\t"
					 "s_addc_u32 m0, m0, 0\n\t"
					 "s_cmp_lg_u32 %7, -1\n\t"
					 "v_writelane_b32 %0, %6, m0\n\t"
					 "s_addc_u32 %1, m0, 0"
					 : "+v"(s0), "+s"(sp)
					 : "s"(ea), "s"(ka), "s"(eb), "s"(kb), "s"(ec), "s"(kc)
					 : "scc", "m0");
	}
	// ... and with the lane MASKS of the three entries (nobody enters: not wanted): the occlusion packets keep no keys
	__device__ __forceinline__ void push3_masks(uint32_t ea, unsigned long long ma, uint32_t eb, unsigned long long mb, uint32_t ec, unsigned long long mc)
	{
		asm volatile("s_mov_b32 m0, %1\n\t"
					 "s_cmp_lg_u64 %3, 0\n\t"
					 "v_writelane_b32 %0, %2, m0\n\t"
					 "s_addc_u32 m0, m0, 0\n\t"
					 "s_cmp_lg_u64 %5, 0\n\t"
					 "v_writelane_b32 %0, %4, m0\n\t"
					 "s_addc_u32 m0, m0, 0\n\t"
					 "s_cmp_lg_u64 %7, 0\n\t"
					 "v_writelane_b32 %0, %6, m0\n\t"
					 "s_addc_u32 %1, m0, 0"
					 : "+v"(s0), "+s"(sp)
					 : "s"(ea), "s"(ma), "s"(eb), "s"(mb), "s"(ec), "s"(mc)
					 : "scc", "m0");
	}
};

struct PacketSpace
{
	f3 o, d, id, noid;	  // the lane's ray in the current space, 1/d, -o/d
	const char *row_n[3]; // wave-uniform: the node table offset to the near / far plane row of a Node4f per axis — a row is
	const char *row_f[3]; // fetched as s_load_dwordx4 dst, row, node_byte_offset with no address arithmetic
	uint32_t mixed;		  // 1: the lanes disagree about a direction sign on some axis (an integer in an SGPR: a wave-uniform bool lives in a lane mask)
	// t_hit: the lane's hit distance — 1/d and -o/d are NORMALISED by k = norm_k(t_hit) (rt_core.h, RT_NORM_T: the interval a box
	// must meet becomes [0, 1], which the clamp modifier of v_max3 / v_min3 folds into the slab test)
	// off: this lane holds no ray and the caller does not mask its ballots — 1/d = -o/d = 0: every slab distance is 0, entry == exit,
	// and no box is entered (0 < 0)
	__device__ __forceinline__ void enter(f3 o_, f3 d_, float t_hit, unsigned long long act, const char *nodes, bool off = false)
	{
		o = o_, d = d_;
		id = mk3(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
#if RT_NORM_T
		id = id * norm_k(t_hit);
#endif
		noid = mk3(-(o.x * id.x), -(o.y * id.y), -(o.z * id.z));
		const unsigned long long mx = __ballot(id.x < 0.0f) & act, my = __ballot(id.y < 0.0f) & act, mz = __ballot(id.z < 0.0f) & act;
		if (off)
			id = mk3(0, 0, 0), noid = mk3(0, 0, 0);
		mixed = (uint32_t)__builtin_amdgcn_readfirstlane(((mx != 0ull && mx != act) || (my != 0ull && my != act) || (mz != 0ull && mz != act)) ? 1 : 0);
		row_n[0] = nodes + (mx ? 48u : 0u), row_f[0] = nodes + (mx ? 0u : 48u);
		row_n[1] = nodes + (my ? 64u : 16u), row_f[1] = nodes + (my ? 16u : 64u);
		row_n[2] = nodes + (mz ? 80u : 32u), row_f[2] = nodes + (mz ? 32u : 80u);
	}
};

// sort key of a child on the scalar unit: the entry distance of the wave's reference lane (bits of a float >= 0: they order
// like unsigned integers); a child no lane enters sorts last
__device__ __forceinline__ uint32_t packet_key(float tk_lane, unsigned long long m, int ref_lane)
{
	const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)fbits(tk_lane), ref_lane);
	return m ? bits : 0xFFFFFFFFu;
}


// ANY (round 6: the shadow rays of the primary vertices, sorted by light): occlusion query — hit.t holds the ray's length on entry, a
// lane that finds a triangle inside (t_min, length) has hit.prim >= 0 and leaves the packet; the wave stops when none is left.
template <bool COUNT, bool ANY = false>
__device__ __forceinline__ void trace_packet(const SceneView &sc, const bool active, const f3 O, const f3 D, const float t_min, Hit &hit, TStat &st)
{
	unsigned long long act = __ballot(active);
	// a lane without a ray — or, ANY, one that has found its occluder — never enters a box: its ray is degenerate (PacketSpace::enter,
	// `off`), no ballot is masked; and it never takes a hit (t > tt fails)
	bool off = !active;
	hit.t = active ? hit.t : -3.0e38f;
	if (act == 0ull)
		return;
	int ref_lane = __ffsll((long long)act) - 1;
	PacketSpace sp;
	const char *const nodes = (const char *)sc.nodes4f; // (the table stays below 4 GiB: 32-bit byte offsets)
	sp.enter(O, D, hit.t, act, nodes, off);
	PacketStack stk;
	int cur_inst = -1;
	uint32_t cur = sc.instance_count ? sc.tlas_root_entry : ENTRY_DONE;
	if (COUNT && active && cur != ENTRY_DONE) // (the root: popped by every ray)
	{
		if (!(cur & ENTRY_LEAF))
			st.inner++;
		else if (!(cur & ENTRY_TLAS))
			st.tris += ((cur >> 27) & 7u) + 1u;
	}
	for (;;)
	{
		while (!(cur & ENTRY_LEAF))
		{
			const uint32_t nb = (cur & ENTRY_INDEX_MASK) << 7;
			float tk[4];
			pk_v4u ent;
			unsigned long long m[4];
			// per child: tmin = max3 of the near-plane distances, clamped at 0; tmax = min3 of the far-plane distances, capped by the
			// lane's hit distance; the lane enters the child when tmin < tmax.  (Traverser::node_step accepts tmax > tmin && tmin <
			// hit.t && tmax >= 0; a box whose far side passes through the origin exactly, tmax == 0, holds nothing beyond t_min.)
			if (!sp.mixed)
			{
				PacketRows r;
				packet_load_rows(r, sp.row_n, sp.row_f, nodes, nb);
				ent = r.ent;
#pragma unroll
				for (int k = 0; k < 4; k++)
				{
#if RT_NORM_T
					tk[k] = max3_clamp01(fmaf(r.nx[k], sp.id.x, sp.noid.x), fmaf(r.ny[k], sp.id.y, sp.noid.y), fmaf(r.nz[k], sp.id.z, sp.noid.z));
					const float tmax = min3_clamp01(fmaf(r.fx[k], sp.id.x, sp.noid.x), fmaf(r.fy[k], sp.id.y, sp.noid.y), fmaf(r.fz[k], sp.id.z, sp.noid.z));
					m[k] = __ballot(tk[k] < tmax);
#else
					const float tmin = fmaxf(fmaxf(fmaf(r.nx[k], sp.id.x, sp.noid.x), fmaf(r.ny[k], sp.id.y, sp.noid.y)), fmaf(r.nz[k], sp.id.z, sp.noid.z));
					const float tmax = fminf(fminf(fmaf(r.fx[k], sp.id.x, sp.noid.x), fmaf(r.fy[k], sp.id.y, sp.noid.y)), fmaf(r.fz[k], sp.id.z, sp.noid.z));
					tk[k] = fmaxf(tmin, 0.0f);
					m[k] = __ballot(tk[k] < fminf(tmax, hit.t));
#endif
				}
			}
			else
			{
				ent = sload4u(nodes, nb + 96u);
				const pk_v4f lx = sload4(nodes, nb), ly = sload4(nodes, nb + 16u), lz = sload4(nodes, nb + 32u);
				const pk_v4f hx = sload4(nodes, nb + 48u), hy = sload4(nodes, nb + 64u), hz = sload4(nodes, nb + 80u);
#pragma unroll
				for (int k = 0; k < 4; k++)
				{
					const float ax = fmaf(lx[k], sp.id.x, sp.noid.x), bx = fmaf(hx[k], sp.id.x, sp.noid.x);
					const float ay = fmaf(ly[k], sp.id.y, sp.noid.y), by = fmaf(hy[k], sp.id.y, sp.noid.y);
					const float az = fmaf(lz[k], sp.id.z, sp.noid.z), bz = fmaf(hz[k], sp.id.z, sp.noid.z);
					// (min / max per plane pair turn the inverted box of an unused slot into a huge one: asked for by its entry)
#if RT_NORM_T
					tk[k] = max3_clamp01(fminf(ax, bx), fminf(ay, by), fminf(az, bz));
					const float tmax = min3_clamp01(fmaxf(ax, bx), fmaxf(ay, by), fmaxf(az, bz));
					m[k] = ent[k] != ENTRY_EMPTY ? __ballot(tk[k] < tmax) : 0ull;
#else
					const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
					const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
					tk[k] = fmaxf(tmin, 0.0f);
					m[k] = ent[k] != ENTRY_EMPTY ? __ballot(tk[k] < fminf(tmax, hit.t)) : 0ull;
#endif
				}
			}
			if (COUNT)
			{
				// statistics per RAY, as its own traversal of this tree would count them: a lane counts the children IT enters (an inner
				// node popped later, or the triangles of a leaf tested later), not every node the wave walks for its neighbours
				const unsigned long long me = 1ull << __lane_id();
				for (int k = 0; k < 4; k++)
					if (m[k] & me)
					{
						const uint32_t e = ent[k];
						if (!(e & ENTRY_LEAF))
							st.inner++;
						else if (!(e & ENTRY_TLAS))
							st.tris += ((e >> 27) & 7u) + 1u;
					}
			}
			// the nearest child some lane enters comes next, the other entered children go on the stack (scalar unit)
			// (ANY: an occlusion ray that reaches its light walks every node along it whatever the order, and most do — the children
			// are taken as they come: no entry distances fetched from the reference lane, no comparators)
			uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
			if (!ANY)
				k0 = packet_key(tk[0], m[0], ref_lane), k1 = packet_key(tk[1], m[1], ref_lane), k2 = packet_key(tk[2], m[2], ref_lane), k3 = packet_key(tk[3], m[3], ref_lane);
			uint32_t e0 = ent[0], e1 = ent[1], e2 = ent[2], e3 = ent[3];
#define RT_PSWAP(KA, EA, KB, EB)                          \
	{                                                     \
		const bool sw = KB < KA;                          \
		const uint32_t kl = sw ? KB : KA, el = sw ? EB : EA; \
		KB = sw ? KA : KB, EB = sw ? EA : EB;             \
		KA = kl, EA = el;                                 \
	}
			if (!ANY)
			{
				RT_PSWAP(k0, e0, k1, e1)
				RT_PSWAP(k2, e2, k3, e3)
				RT_PSWAP(k0, e0, k2, e2)
			}
#undef RT_PSWAP
			// (three of the five comparators: the nearest entered child first, the others in no particular order — all five measure
			// the same, 5.18 against 4.93 ms per primary wave; (key, entry) as one 64-bit scalar each and s_cselect_b64: the compiler
			// splits the selects again.)  The pop below reads what was just written
			if (ANY)
				stk.push3_masks(e3, m[3], e2, m[2], e1, m[1]);
			else
				stk.push3_keys(e3, k3, e2, k2, e1, k1);
			if (ANY ? m[0] == 0ull : k0 == 0xFFFFFFFFu)
				e0 = stk.pop();
			cur = e0;
		}
		if (cur == ENTRY_DONE)
			break;
		if (cur == ENTRY_SENTINEL)
		{
			sp.enter(O, D, hit.t, act, nodes, off); // leaving an instance: back to the world-space ray
			cur_inst = -1;
			cur = stk.pop();
			continue;
		}
		if (cur & ENTRY_TLAS)
		{
			// top-level leaf: exactly one instance; its inverse transform is wave-uniform, every lane transforms its own ray
			// (direction NOT renormalised so that t is shared: top_level_bvh.cpp:104-168)
			const uint32_t ii = sload1u(sc.tlas_prims, (cur & ENTRY_FIRST_MASK) * 4u);
			const char *const ib = (const char *)(sc.instances + ii);
			const pk_v4f r0 = sload4(ib, 0u), r1 = sload4(ib, 16u), r2 = sload4(ib, 32u);
			const uint32_t root = sload1u(ib, (uint32_t)offsetof(Instance, root_entry));
			stk.push(ENTRY_SENTINEL);
			const float m0[4] = {r0[0], r0[1], r0[2], r0[3]}, m1[4] = {r1[0], r1[1], r1[2], r1[3]}, m2[4] = {r2[0], r2[1], r2[2], r2[3]};
			sp.enter(mk3(row_point_r(m0, O), row_point_r(m1, O), row_point_r(m2, O)), mk3(row_dir_r(m0, D), row_dir_r(m1, D), row_dir_r(m2, D)),
					 hit.t, act, nodes, off);
			cur_inst = (int)ii;
			cur = root;
			continue;
		}
		// a triangle leaf: every lane tests every triangle (Traverser::visit's test on its own ray)
		{
			const uint32_t first = cur & ENTRY_FIRST_MASK, count = ((cur >> 27) & 7u) + 1u;
			const char *const tb = (const char *)sc.tri_verts + (size_t)first * 48u;
#if RT_NORM_T
			const float t_before = hit.t;
#endif
			for (uint32_t i = 0; i < count; i++)
			{
				const pk_v4f v0 = sload4(tb, i * 48u), v1 = sload4(tb, i * 48u + 16u), v2 = sload4(tb, i * 48u + 32u);
				const int tri_inst = cur_inst >= 0 ? cur_inst : (int)fbits(v1[3]);
				if (tri_test<!ANY>(sp.o, sp.d, t_min, hit.t, mk3(v0[0], v0[1], v0[2]), mk3(v1[0], v1[1], v1[2]), mk3(v2[0], v2[1], v2[2]), hit.u, hit.v,
								   v2[3], fbits(v0[3]), (uint32_t)hit.prim, (uint32_t)tri_inst, (uint32_t)hit.inst))
				{
					hit.prim = (int)fbits(v0[3]);
					hit.inst = tri_inst;
					if (ANY)
						hit.t = -3.0e38f, off = true, sp.id = mk3(0, 0, 0), sp.noid = mk3(0, 0, 0); // (occluded: this lane takes no further hit and enters no further box)
				}
			}
			if (ANY)
			{
				// (... and enters no further box)
				act &= ~__ballot(hit.prim >= 0);
				if (act == 0ull)
					break;
				ref_lane = __ffsll((long long)act) - 1;
			}
#if RT_NORM_T
			if (!ANY && hit.t != t_before) // the lane's hit moved: rescale its normalised 1/d and -o/d (Traverser::renormalise)
			{
				const float r = norm_k(hit.t) * fast_rcp(norm_k(t_before)) * 0.99999952f;
				sp.id = sp.id * r, sp.noid = sp.noid * r;
			}
#endif
			cur = stk.pop();
		}
	}
}

// The pt primary wave in packet form: a wave takes runs of consecutive path slots from the launch's queue (one atomic per
// run), generates the 64 primary rays of one 64-slot group per lane and walks the tree once for all of them.  No LDS.
#ifndef RT_PACKET_WAVES
#define RT_PACKET_WAVES 8
#endif
// path slots a wave takes from the queue per atomic: a run is 64-slot groups of neighbouring pixels, and the wave that walks
// them one after the other finds the previous group's nodes and triangles in the scalar cache: 256 / 512 / 1024 / 2048 / 4096
// slots per run -> 6.27 / 4.90 / 4.49 / 4.65 / 5.59 ms per 64-spp primary wave (longer runs leave too few of them per wave).
#ifndef RT_PACKET_CHUNK
#define RT_PACKET_CHUNK 1024
#endif

// The kernel's Params read afresh from the kernarg segment (the first argument sits at offset 0).  The packet kernel asks for
// them once per stage of a group — generation, traversal, hit store: read once at kernel entry, the ~60 scalars of camera,
// frame and scene stay live across the node loop, which needs ~65 SGPRs itself, and spill into VGPR lanes (a v_readlane per
// use, on the VALU); a scalar load from the kernarg segment per stage costs the VALU nothing.
__device__ __forceinline__ const Params &fresh_params()
{
	auto ka = __builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(ka));
	return *(const Params *)ka;
}

template <bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK, RT_PACKET_WAVES) k_primary_packet(const Params p, const uint32_t count)
{
	clock_in(p.wv.counters, 0);
	uint32_t *const head = &p.wv.counters->work[p.queue][0];
	const uint32_t lane = __lane_id();
	uint32_t run = count / (gridDim.x * (blockDim.x / 64u) * 4u);
	run = run > (uint32_t)RT_PACKET_CHUNK ? (uint32_t)RT_PACKET_CHUNK : (run < 64u ? 64u : (run & ~63u));
	TStat st;
	st.inner = 0, st.tris = 0, st.lds = 0;
	uint32_t nrays = 0;
	for (;;)
	{
		uint32_t g = 0;
		if (lane == 0)
			g = atomicAdd(head, run);
		g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
		if (g >= count)
			break;
		const uint32_t end = g + run < count ? g + run : count;
		for (uint32_t base = g; base < end; base += 64u)
		{
			const uint32_t idx = base + lane;
			bool active = idx < end;
			f3 O = mk3(0, 0, 0), D = mk3(0, 0, 1);
			{
				const Params &q = fresh_params();
				if (active)
				{
					const PixelRef pr = slot_to_pixel(q.fr, idx);
					active = pr.valid;
					if (active)
					{
						pt_primary_ray(q.cam, q.fr, pr.x, pr.y, q.fr.sample_base + pr.sample, O, D);
						if (q.cam.aperture != 0.0f) // (pinhole: no origin record, extend_item)
							q.wv.org[0][idx] = mk4(O.x, O.y, O.z, ubits((idx << 1) | 1u));
					}
				}
			}
			Hit h;
			h.t = 1e34f, h.u = 0.0f, h.v = 0.0f, h.prim = -1, h.inst = -1;
			trace_packet<COUNT>(fresh_params().sc, active, O, D, 1e-5f, h, st);
			if (active)
			{
				primary_finish_item(fresh_params(), idx, D, h); // (direction + hit record, or the sky term of a miss)
				nrays++;
			}
			// a group without a hit is finished here (27 % of the terrain's): the shade kernel's scan passes it by
			if (RT_PRIMARY_MISS)
			{
				unsigned char *const done = fresh_params().wv.hit0_done;
				const bool none = __ballot(active && h.prim >= 0) == 0ull;
				if (done && lane == 0u)
					done[base >> 6] = none ? 1 : 0;
			}
		}
	}
	if (COUNT)
	{
		WaveCounters *const wc = p.wv.counters;
		Ctx ctx;
		ctx.add64(&wc->inner_extend, st.inner);
		ctx.add64(&wc->tris_extend, st.tris);
		ctx.add64(&wc->rays_extend, nrays);
	}
	clock_out(p.wv.counters, 0);
}

// ----------------------------------------------------------------------------------------------------------------
// The connection wave of the PRIMARY vertices in packet form (round 6; setting `shadow_packets`).  With 64 samples of a pixel side
// by side, a run of the depth-0 shadow queue holds the connections of four or five neighbouring pixels' first vertices: nearly one
// origin, and as many directions as those vertices chose lights — three or four that matter.  The shade kernel has written the
// chosen light's bin into the top bits of every ray's slot word (FrameView::shadow_bins); a wave takes a run of RT_SHADOW_RUN rays,
// sorts their queue indices by bin (stable: ballots and prefixes, the order in LDS), and walks the tree ONCE for
// every 64 rays of the sorted order — trace_packet<ANY>: scalar node fetches, one stack per wave — instead of once per lane.  Which
// rays share a packet changes nothing about a ray's answer (is anything inside (1e-5, length)?), so the image is the per-lane
// kernels' bit for bit.
// ----------------------------------------------------------------------------------------------------------------
#ifndef RT_SHADOW_RUN
#define RT_SHADOW_RUN 256u
#endif
template <bool COUNT>
__global__ void __launch_bounds__(TRACE_BLOCK, RT_PACKET_WAVES) k_shadow_packet(const Params p)
{
	static_assert(RT_SHADOW_RUN % 64u == 0u && RT_SHADOW_RUN <= 1024u, "run = whole packets");
	__shared__ uint32_t s_order[TRACE_BLOCK / 64][RT_SHADOW_RUN];
	const uint32_t count = connection_count(p.wv.counters, p.depth);
	if (count == 0u)
	{
		if (p.depth == 0 && p.wv.rad_nee)
			for (uint32_t i = blockIdx.x * TRACE_BLOCK + threadIdx.x, n = p.wv.counters->shadow_n[0]; i < n; i += gridDim.x * TRACE_BLOCK)
				connect_skip_item(p, i);
		return;
	}
	const uint32_t lane = __lane_id(), wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	uint32_t *const order = s_order[wave];
	const uint32_t slot_bits = shadow_slot_bits(p.fr.shadow_bins);
	uint32_t *const head = &p.wv.counters->work[p.queue][0];
	TStat st;
	st.inner = 0, st.tris = 0, st.lds = 0;
	uint32_t nrays = 0, nruns = 0, nbins = 0;
	for (;;)
	{
		uint32_t g = 0;
		if (lane == 0)
			g = atomicAdd(head, RT_SHADOW_RUN);
		g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
		if (g >= count)
			break;
		const uint32_t n = count - g < RT_SHADOW_RUN ? count - g : RT_SHADOW_RUN;
		// ---- the run's queue indices sorted by light bin (void entries left out), STABLE: within a bin the rays keep the order of
		// the queue — the order of the image — so that a packet is the rays of one or two neighbouring pixels towards one light,
		// not a sample of all the run's pixels.  One ballot + prefix per bin and 64 entries (64 of them per run: a few hundred
		// instructions against the thousands of a packet's traversal)
		uint32_t bin[RT_SHADOW_RUN / 64u];
#pragma unroll
		for (uint32_t j = 0; j < RT_SHADOW_RUN / 64u; j++)
		{
			const uint32_t e = j * 64u + lane;
			bin[j] = SHADOW_BINS; // (beyond the run, or void)
			if (e < n)
			{
				const uint32_t w = fbits(fresh_params().wv.sh_org[g + e].w);
				bin[j] = w == RAY_VOID ? SHADOW_BINS : w >> slot_bits;
			}
		}
		uint32_t valid = 0;
		for (uint32_t b = 0; b < SHADOW_BINS; b++)
		{
			const uint32_t before = valid;
#pragma unroll
			for (uint32_t j = 0; j < RT_SHADOW_RUN / 64u; j++)
			{
				const unsigned long long m = __ballot(bin[j] == b);
				if (bin[j] == b)
					order[valid + wave_prefix(m)] = g + j * 64u + lane;
				valid += (uint32_t)__popcll(m);
			}
			nbins += valid != before ? 1u : 0u;
		}
		nruns++;
		__builtin_amdgcn_wave_barrier();
		// ---- one packet per 64 rays of the sorted order
		for (uint32_t k = 0; k < valid; k += 64u)
		{
			const bool active = k + lane < valid;
			uint32_t idx = 0, slot = 0;
			f3 O = mk3(0, 0, 0), D = mk3(0, 0, 1);
			Hit h;
			h.t = 0.0f, h.u = 0.0f, h.v = 0.0f, h.prim = -1, h.inst = -1;
			if (active)
			{
				idx = order[k + lane];
				const Params &q = fresh_params(); // (the arguments from the kernarg segment again: nothing of them stays live across the walk)
				const f4 o4 = q.wv.sh_org[idx], d4 = q.wv.sh_dir[idx];
				O = xyz(o4), D = xyz(d4), h.t = d4.w, slot = shadow_slot(q, fbits(o4.w));
			}
			// (a ray of negative length is traced, hits nothing and counts: Kernels.cu:750 — the packet walks nothing for it)
			trace_packet<COUNT, true>(fresh_params().sc, active && h.t > 1e-5f, O, D, 1e-5f, h, st);
			if (active)
			{
				connect_finish(fresh_params(), idx, slot, h.prim < 0);
				nrays++;
			}
		}
	}
	if (lane == 0u && nruns)
		atomicAdd(&p.wv.counters->sp_runs, (unsigned long long)nruns), atomicAdd(&p.wv.counters->sp_bins, (unsigned long long)nbins);
	if (COUNT)
	{
		WaveCounters *const wc = p.wv.counters;
		Ctx ctx;
		ctx.add64(&wc->inner_shadow, st.inner);
		ctx.add64(&wc->tris_shadow, st.tris);
		ctx.add64(&wc->rays_shadow, nrays);
	}
}

template <bool COUNT>
__global__ void __launch_bounds__(BLOCK, RT_TRAVERSAL_WAVES) k_shade_parity(const Params p, const uint32_t count)
{
	RT_STACK_DECL_ANY
	ChunkQueue w(p, count);
	uint32_t c;
	while (w.next(c))
	{
		const uint32_t i = c * BLOCK + threadIdx.x;
		if (c < w.nchunks)
			shade_parity_item<COUNT>(p, i, i < count, ctx);
	}
}

template <bool TEX> __global__ void __launch_bounds__(BLOCK, TEX ? RT_SHADE_WAVES : RT_SHADE_WAVES_PLAIN) k_shade_pt(const Params p)
{
	static_assert(BLOCK == POT_STRIDE, "potential cache layout is pot[light][thread]");
	__shared__ float s_pot[POT_SLOTS * BLOCK];
	Ctx ctx;
	ctx.stk.lds = nullptr, ctx.stk.spill = nullptr, ctx.stk.top = nullptr, ctx.stk.top_first = 0, ctx.stk.top_count = 0;
	ctx.stk.overflow = nullptr, ctx.stk.stride = 0;
	ctx.pot = s_pot + threadIdx.x;
	const uint32_t count = p.wv.counters->ext_n[p.depth];
	// Hits and misses cost two orders of magnitude apart (sky lookup vs. BSDF + light sampling) and are mixed lane by lane
	// on the bounce waves.  Every WAVE keeps its own queues of hit paths and of misses in LDS: it walks 64-path chunks and appends
	// every entry to its queue; whenever 64 of a kind are queued it shades them as one full wave (hits first).
	// No workgroup barrier anywhere (round 1's per-256 compaction parked the waves without hits at a barrier, holding
	// their SIMD slots, while the others shaded), and the expensive path always runs with all lanes.  Which lane shades a
	// path does not affect its result.
	// Round 6: the queues are RINGS of 128 entries (nothing moves down when a wave of entries leaves), and a queued hit keeps the hit
	// record and instance the scan has read beside its entry — the scan's loads are coalesced and fetched the whole records' lines
	// anyway; the shading then starts at "instance -> shading record" instead of at a second, gathered read of both.
	constexpr uint32_t QN = 128u, QM = QN - 1u;
	__shared__ f4 s_qhit[BLOCK / 64][QN];
	__shared__ uint32_t s_qidx[BLOCK / 64][QN];
	__shared__ int s_qinst[BLOCK / 64][QN];
	__shared__ uint32_t s_qmiss[BLOCK / 64][QN];
	// (Round 6, built on this state, bit-identical, and the same speed to 1 %: (a) the scan's records asked for one scan ahead by LDS-DMA —
	// global_load_lds_dwordx4 / _dword into staging rows of the wave, no register in flight: 7.89 against 7.93 ms per sub-batch; (b) the
	// scan of the next chunk issued with the loads of the batch being shaded, one round trip for both, the hit's shading index looked up
	// by the scan so that the shading record is asked for with the path records: 7.88 against 7.94; (c) hits sorted by material on
	// textured scenes — plain variant for the hits on untextured materials, textured variant over a list of the others: atrium shade
	// 16.7 against 15.2 ms.  A wave's time between two items is vmcnt(0) — on gfx9 one counter for loads AND stores — and what the
	// kernel as a whole is bound by is not the length of a wave's chain of round trips.  DESIGN_LOG.md, round 6.)
	// The wave index as a scalar (readfirstlane): the four ring bases are SGPRs, not a VGPR each — with it the kernel spills nothing.
	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	f4 *const qhit = s_qhit[wave];
	uint32_t *const qidx = s_qidx[wave], *const qmiss = s_qmiss[wave];
	int *const qinst = s_qinst[wave];
	const f4 *const hits = p.depth == 0 ? p.wv.hit0 : p.wv.hit;
	const int *const insts = p.depth == 0 ? p.wv.hit0_inst : p.wv.hit_inst;
	const uint32_t nchunks = (count + 63u) / 64u;
	const uint32_t nwaves = gridDim.x * (BLOCK / 64u);
	// queue block: QUEUE_BLOCK slots for big launches; a launch so small that a wave would leave most of a block empty
	// reserves exactly what each call emits instead (no void entries; the atomics are few then)
	{
		const uint32_t per_wave = count / (nwaves * 4u);
		ctx.q_block = per_wave > QUEUE_BLOCK ? QUEUE_BLOCK : (per_wave < 64u ? 0u : per_wave);
	}
#if defined(RT_DIAG_SHADE_CLOCK)
	__shared__ unsigned long long s_clk[BLOCK / 64][32];
	if (lane < 32u)
		s_clk[wave][lane] = 0ull;
	ClkProbe clk;
	clk.last = __builtin_readcyclecounter(), clk.acc = s_clk[wave];
#endif
#ifndef RT_SHADE_RUN
#define RT_SHADE_RUN 32u
#endif
#ifndef RT_SHADE_FRESH_PARAMS
#define RT_SHADE_FRESH_PARAMS 1
#endif
	// a wave walks RUNS of consecutive chunks (fewer when the launch has less than four runs per wave)
	uint32_t srun = nchunks / (nwaves * 4u);
	srun = srun > RT_SHADE_RUN ? RT_SHADE_RUN : (srun ? srun : 1u);
	uint32_t c = (blockIdx.x * (BLOCK / 64u) + wave) * srun; // this wave's next chunk
	uint32_t c_left = srun;									 // chunks left of the wave's run
	uint32_t hq = 0, nq = 0;						// hit queue: first entry, entries (wave-uniform)
	uint32_t mq = 0, nm = 0;						// miss queue
	uint32_t nshaded = 0;							// hits shaded by this wave (statistics: the gathers of the roofline's byte count)
	// depth 0 behind the packet form of the primary wave: which 64-slot groups of the wave's run are finished already
	// (WaveView::hit0_done; one coalesced byte load and a ballot per run)
	static_assert(RT_SHADE_RUN <= 32u, "run_done is a 32-bit mask");
	const unsigned char *const done = p.depth == 0 ? p.wv.hit0_done : nullptr;
	uint32_t run_done = 0;
#pragma nounroll
	for (;;)
	{
		uint32_t idx = 0;
		bool act = false;
		f4 h4 = mk4(0, 0, 0, ubits((uint32_t)-1));
		int hi = -1;
		const bool drain = c >= nchunks;
		const bool take_hits = nq >= 64u || (drain && nq > 0u);
		if (take_hits)
		{
			const uint32_t n = nq < 64u ? nq : 64u, e = (hq + lane) & QM;
			nshaded += n;
			act = lane < n;
			idx = qidx[e], h4 = qhit[e], hi = qinst[e];
			hq = (hq + n) & QM, nq -= n;
		}
		else if (nm >= 64u || (drain && nm > 0u))
		{
			const uint32_t n = nm < 64u ? nm : 64u;
			act = lane < n;
			idx = qmiss[(mq + lane) & QM];
			mq = (mq + n) & QM, nm -= n;
		}
		else if (!drain)
		{
			if (done && c_left == srun) // a run begins
				run_done = (uint32_t)__ballot(lane < srun && c + lane < nchunks && done[c + lane] != 0);
			const bool skip = (run_done >> (srun - c_left)) & 1u;
			idx = c * 64u + lane;
			if (--c_left)
				c++;
			else
				c += (nwaves - 1u) * srun + 1u, c_left = srun;
			if (skip)
				continue;
			bool valid = idx < count;
			if (p.depth == 0 && valid)
				valid = slot_to_pixel(p.fr, idx).valid;
			if (valid)
				h4 = hits[idx], hi = insts[idx];
			const int prim = valid ? (int)fbits(h4.w) : HIT_VOID;
			const bool is_hit = prim >= 0;
			const unsigned long long m = __ballot(is_hit);
			if (is_hit)
			{
				const uint32_t e = (hq + nq + wave_prefix(m)) & QM;
				qidx[e] = idx, qhit[e] = h4, qinst[e] = hi;
			}
			nq += (uint32_t)__popcll(m);
			// a miss; HIT_VOID entries (unfilled queue slots) and HIT_MISS_SHADED ones (finished by the primary kernel) are nobody's
			// path.  The misses of a bounce wave are mixed lane by lane with its hits: they wait for a full wave too.
			const bool is_miss = prim == -1;
			const unsigned long long mm = __ballot(is_miss);
			if (is_miss)
				qmiss[(mq + nm + wave_prefix(mm)) & QM] = idx;
			nm += (uint32_t)__popcll(mm);
#if defined(RT_DIAG_SHADE_CLOCK)
			clk_tick(clk, 14); // (a scan: from the end of the last item or scan to here)
#endif
			continue;
		}
		else
			break;
		__builtin_amdgcn_wave_barrier();
#if defined(RT_DIAG_SHADE_CLOCK)
		shade_pt_item<TEX>(p, idx, act, h4, hi, ctx, &clk);
#else
#if RT_SHADE_FRESH_PARAMS
		// (the kernel's arguments read again from the kernarg segment — scalar loads — instead of staying live across the loop: with
		// ~150 wave-uniform words of scene, wave buffers, camera and frame the compiler parks them in VGPR lanes, a v_readlane per use)
		shade_pt_item<TEX>(fresh_params(), idx, act, h4, hi, ctx);
#else
		shade_pt_item<TEX>(p, idx, act, h4, hi, ctx);
#endif
#endif
	}
#if defined(RT_DIAG_SHADE_CLOCK)
	if (lane < 32u)
		atomicAdd(g_shade_clk + lane, s_clk[wave][lane]);
#endif
	// what is left of this wave's last queue blocks becomes void entries; the ray counts go to the statistics
	WaveCounters *const wc = p.wv.counters;
	const uint32_t nb = (p.depth & 1u) ^ 1u;
	for (uint32_t s = ctx.q_ext.pos + lane; s < ctx.q_ext.end; s += 64u)
		p.wv.org[nb][s] = mk4(0, 0, 0, ubits(RAY_VOID));
	for (uint32_t s = ctx.q_shadow.pos + lane; s < ctx.q_shadow.end; s += 64u)
		p.wv.sh_org[s] = mk4(0, 0, 0, ubits(RAY_VOID));
	if (lane == 0u)
	{
		if (ctx.q_ext.rays)
			atomicAdd(&wc->ext[p.depth + 1], ctx.q_ext.rays);
		if (ctx.q_shadow.rays)
			atomicAdd(&wc->shadow[p.depth], ctx.q_shadow.rays);
		if (nshaded)
			atomicAdd(&wc->shaded, (unsigned long long)nshaded);
	}
}

template <bool COUNT>
__global__ void __launch_bounds__(BLOCK, RT_ANY_WAVES) k_connect(const Params p)
{
	const uint32_t count = connection_count(p.wv.counters, p.depth);
	if (count == 0u)
	{
		if (p.depth == 0 && p.wv.rad_nee)
			for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x, n = p.wv.counters->shadow_n[0]; i < n; i += gridDim.x * BLOCK)
				connect_skip_item(p, i);
		return;
	}
	RT_STACK_DECL_ANY
	ChunkQueue w(p, count);
	uint32_t c;
	while (w.next(c))
	{
		const uint32_t i = c * BLOCK + threadIdx.x;
		if (c < w.nchunks)
			connect_item<COUNT>(p, i, i < count, ctx);
	}
}

__global__ void __launch_bounds__(BLOCK) k_resolve(const Params p)
{
	const uint32_t n = p.fr.W * p.fr.local_rows;
	for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
		resolve_item(p, i);
}

__global__ void __launch_bounds__(BLOCK) k_present(const Params p, f4 *out, float scale, int full)
{
	const uint32_t n = p.fr.W * p.fr.local_rows;
	for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
		present_item(p, out, scale, full, i);
}

__global__ void __launch_bounds__(BLOCK) k_deinterleave(const f4 *gathered, f4 *out, uint32_t W, uint32_t H,
														uint32_t local_rows, uint32_t world)
{
	const uint32_t n = W * H;
	for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
		deinterleave_item(gathered, out, W, H, local_rows, world, i);
}

__global__ void __launch_bounds__(BLOCK) k_kat(const Params p, int function, const float *in, float *out, uint32_t n)
{
	__shared__ float s_pot[POT_SLOTS * BLOCK];
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n)
		kat_item(p, function, in, out, i, s_pot + threadIdx.x);
}
void launch_kat(const Params &p, int function, const float *in, float *out, uint32_t n, stream_t s)
{
	if (n)
		hipLaunchKernelGGL(k_kat, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, p, function, in, out, n);
}

__global__ void k_init_counters(WaveCounters *c, uint32_t primary_count)
{
	if (blockIdx.x == 0)
		init_counters_item(c, primary_count, threadIdx.x, blockDim.x);
}
__global__ void k_set_ext_count(WaveCounters *c, uint32_t depth, uint32_t count)
{
	if (threadIdx.x == 0 && blockIdx.x == 0)
		c->ext[depth] = c->ext_n[depth] = count;
}

struct RngBase
{
	uint32_t s[4];
};
__global__ void __launch_bounds__(BLOCK) k_rng_states(uint32_t *states, RngBase base, const uint32_t *table, uint32_t total)
{
	const uint32_t r = blockIdx.x * BLOCK + threadIdx.x;
	rng_states_item(states, base.s, table, total, r);
}

__global__ void __launch_bounds__(BLOCK) k_refit_tris(f4 *tri_verts, const f4 *verts, const uint32_t *indices, uint32_t n)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n)
		refit_tris_item(tri_verts, verts, indices, i);
}

// pass 2: one thread per node; leaves recompute their box and walk up; the second thread to arrive at a parent
// (agent-scope counter + fences: the sibling's box was written by another CU) merges the two children.
__global__ void __launch_bounds__(BLOCK) k_refit_nodes(Node *all_nodes, uint32_t node_base, const int *parents,
													   uint32_t node_count, const f4 *tri_verts, uint32_t *flags)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= node_count)
		return;
	Node *nodes = all_nodes + node_base; // BLAS-relative view; entries inside are absolute
	const Node n = nodes[i];
	if (n.count < 0 || (i == 1u)) // inner node, or the unused slot next to the root
		return;
	float mn[3], mx[3];
	leaf_bounds(n, tri_verts, mn, mx);
	for (int a = 0; a < 3; a++)
		nodes[i].bmin[a] = mn[a], nodes[i].bmax[a] = mx[a];
	int cur = (int)i;
	for (;;)
	{
		const int parent = parents[cur];
		if (parent < 0)
			break;
		__threadfence(); // release our box
		if (atomicAdd(&flags[parent], 1u) == 0u)
			break;		 // first arrival: the sibling will continue
		__threadfence(); // acquire the sibling's box
		const int l = (int)(((uint32_t)nodes[parent].left_first & ENTRY_INDEX_MASK) - node_base);
		for (int a = 0; a < 3; a++)
		{
			const float lo0 = __hip_atomic_load(&nodes[l].bmin[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const float lo1 = __hip_atomic_load(&nodes[l + 1].bmin[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const float hi0 = __hip_atomic_load(&nodes[l].bmax[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const float hi1 = __hip_atomic_load(&nodes[l + 1].bmax[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			nodes[parent].bmin[a] = fminf(lo0, lo1);
			nodes[parent].bmax[a] = fmaxf(hi0, hi1);
		}
		cur = parent;
	}
}

__global__ void __launch_bounds__(BLOCK) k_stamp_instance(f4 *tri_verts, uint32_t tri_count, uint32_t instance)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < tri_count)
		tri_verts[3ull * i + 1].w = ubits(instance);
}
void launch_stamp_instance(f4 *tri_verts, uint32_t tri_count, uint32_t instance, stream_t s)
{
	if (tri_count)
		hipLaunchKernelGGL(k_stamp_instance, dim3((tri_count + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, tri_verts, tri_count, instance);
}

__global__ void __launch_bounds__(BLOCK) k_refresh4(Node4c *nodes4, const uint32_t *src4, uint32_t count4, const Node *nodes2)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < count4)
		refresh4_item(nodes4, src4, nodes2, i);
}

__global__ void __launch_bounds__(BLOCK) k_expand4(const Node4c *nodes4, Node4f *out, uint32_t count4)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < count4)
		expand4_item(nodes4, out, i);
}
void launch_expand4(const Node4c *nodes4, Node4f *out, uint32_t count4, stream_t s)
{
	if (count4)
		hipLaunchKernelGGL(k_expand4, dim3((count4 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, nodes4, out, count4);
}

// Grid of the persistent kernels: more workgroups than fit on the chip at once (4-7 per CU).  The queue-driven kernels
// do not care much (late workgroups find their queue empty), the shade kernel walks its chunks with a static stride
// and balances better the finer that stride is.  Swept on MI355X: 8 / 16 / 32 / 64 / 128 per CU -> 1820 / 1835 /
// 1857 / 1829 / 1780 Msamples/s (beyond 32 every extra workgroup still stages the LDS node cache and polls a queue).
// Persistent grids: workgroups per CU, upper bound.  Re-swept after the traversal kernels took the persistent-lane form (a
// CU holds 8 of their workgroups; more only queue up behind them): traversal 8 / 12 / 16 / 24 / 32 / 48 per CU -> 2673 /
// 2675 / 2659 / 2640 / 2626 / 2615 Msamples/s, shade kernel 8 / 12 / 16 / 24 / 32 / 64 -> 2592 / 2612 / 2633 / 2668 / 2589 / 2584.
// Round 3: 8 / 8.  A CU holds 8 workgroups of the traversal kernels; a grid of exactly that many leaves the workgroups of the
// NEXT chain's kernel nothing to queue behind, which is what a strip-split rank's smaller launches need (33 M paths per call,
// three calls in flight: 12 / 16 -> 8 / 8 per CU = 10.45 -> 9.78 ms per 128-spp step; the single-GPU bench is the same at
// 6 / 8 / 12 per CU, 4 loses on every traversal kernel).
#ifndef RT_GRID_BLOCKS_PER_CU
#define RT_GRID_BLOCKS_PER_CU 8u
#endif
// Round 4: the shade kernel's grid is what a CU HOLDS of it (5 workgroups of the plain variant at 96 registers, 4 of the textured
// one at 128).  Its waves walk their chunks with a static stride, so the 3 workgroups per CU that a grid of 8 left waiting ran as
// a second round on a chip 60 % full: shade alone on the chip 10.4 -> 9.5 ms per 64-spp sub-batch (8 / 5 / 4 / 10 per CU ->
// 10.37 / 9.51 / 9.87 / 9.60; the pipelined step does not notice: 4428-4442 / 4432 / 4400 / 4443 Msamples/s).
#ifndef RT_SHADE_BLOCKS_PER_CU
#define RT_SHADE_BLOCKS_PER_CU(TEX) ((TEX) ? (uint32_t)RT_SHADE_WAVES : (uint32_t)RT_SHADE_WAVES_PLAIN)
#endif
#ifndef RT_GRID_CHUNKS_PER_BLOCK
#define RT_GRID_CHUNKS_PER_BLOCK 32u // a workgroup should find about this many 256-item chunks to be worth launching
#endif
// Round 5: the SMALLEST grid of a queue-fed kernel (traversal, shade) is 2-3 workgroups per CU, not a full chip's worth: a 1-spp
// 1080p frame is ten launches over 2 M items or fewer, three frames' chains are in flight, and a wave of a full persistent grid then
// finds ~300 rays — five refills — before it idles through the launch's tail; with a quarter of the waves each keeps its lanes
// filled four times as long and the kernels of the other chains fill the CUs (pipelined 1-spp frames 1.195 -> 0.815 ms; a lower
// bound of 1 / 2 / 3 / 4 per CU: 1.04 / 0.82 / 0.81 / 0.96).  The streaming kernels (resolve, present, de-interleave) keep a full
// grid: they want every CU's memory pipeline (with the lower bound on them too, 8-spp steps lose 4.6 %).
#ifndef RT_GRID_LO_PER_CU
#define RT_GRID_LO_PER_CU 3u
#endif
static inline uint32_t persistent_grid(uint32_t items, uint32_t per_cu = RT_GRID_BLOCKS_PER_CU, uint32_t lo_per_cu = RT_GRID_LO_PER_CU)
{
	const uint32_t chunks = (items + BLOCK - 1) / BLOCK;
	const uint32_t lo = (uint32_t)g_cus * (per_cu < lo_per_cu ? per_cu : lo_per_cu), hi = (uint32_t)g_cus * per_cu;
	uint32_t blocks = chunks / RT_GRID_CHUNKS_PER_BLOCK;
	blocks = blocks < lo ? lo : (blocks > hi ? hi : blocks);
	if (blocks > chunks)
		blocks = chunks;
	blocks = (blocks + 7u) & ~7u; // multiple of 8 so every XCD gets the same number of blocks
	return blocks ? blocks : 8u;
}

void launch_init_counters(WaveCounters *c, uint32_t primary_count, stream_t s)
{
	hipLaunchKernelGGL(k_init_counters, dim3(1), dim3(256), 0, (hipStream_t)s, c, primary_count);
}

void launch_set_ext_count(WaveCounters *c, uint32_t depth, uint32_t count, stream_t s)
{
	hipLaunchKernelGGL(k_set_ext_count, dim3(1), dim3(64), 0, (hipStream_t)s, c, depth, count);
}

void launch_rng_states(uint32_t *states, const uint32_t base_state[4], const uint32_t *jump_table,
					   uint32_t packets_per_sample, uint32_t spp, stream_t s)
{
	const uint32_t total = packets_per_sample * spp;
	const uint32_t threads = (total + RNG_RUN - 1) / RNG_RUN;
	RngBase b;
	memcpy(b.s, base_state, 16);
	hipLaunchKernelGGL(k_rng_states, dim3((threads + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, states, b,
					   jump_table, total);
}

// does the pt primary wave of this launch run in packet form?  (the host asks too: only that form fills WaveView::hit0_done)
bool primary_packet_form(const Params &p, uint32_t max_items)
{
	return (p.refill & 8u) && (p.fr.sgroup_log2 >= 1u || max_items >= RT_PRIMARY_PACKET_MIN);
}

void launch_extend(const Params &p, int gen, bool count, uint32_t max_items, stream_t s)
{
	const dim3 g(persistent_grid(max_items)), b(BLOCK);
	hipStream_t st = (hipStream_t)s;
#define RT_EXT(G, C) hipLaunchKernelGGL((k_extend<G, C>), g, b, 0, st, p, max_items)
	if (gen == GEN_BUFFER && (p.refill & 1u))
	{
		const dim3 gt(std::max(8u, g.x * BLOCK / TRACE_BLOCK)), bt(TRACE_BLOCK);
		if (count)
			hipLaunchKernelGGL((k_trace_stream<false, true>), gt, bt, 0, st, p);
		else
			hipLaunchKernelGGL((k_trace_stream<false, false>), gt, bt, 0, st, p);
	}
	else if (gen == GEN_BUFFER)
	{
		if (count)
			RT_EXT(GEN_BUFFER, true);
		else
			RT_EXT(GEN_BUFFER, false);
	}
	else if (gen == GEN_RANGED)
	{
		if (count)
			RT_EXT(GEN_RANGED, true);
		else
			RT_EXT(GEN_RANGED, false);
	}
	else if (gen == GEN_PT && primary_packet_form(p, max_items))
	{
		// packet form of the primary wave: no LDS, one wave = one 64-slot group at a time.  MI355X, 1080p terrain, 64 spp per
		// launch, primary wave per-lane / packet by sample-group size: 1: 9.33 / 9.07 ms, 4: 9.08 / 6.60, 8: 8.87 / 6.39, 16: 8.79 /
		// 5.52, 32: 8.45 / 5.15, 64: 7.27 / 4.92 — a wave of one pixel's samples reaches the same few leaves, a wave that is an 8x8
		// tile of different pixels reaches 64 different ones and every lane tests them all.  Small launches of single samples (1-spp
		// frames: 0.37 against 0.57 ms) keep the one-ray-per-lane kernel with its tile-row-to-XCD dealing.
		const dim3 gt(std::max(8u, g.x * BLOCK / TRACE_BLOCK)), bt(TRACE_BLOCK);
		if (count)
			hipLaunchKernelGGL((k_primary_packet<true>), gt, bt, 0, st, p, max_items);
		else
			hipLaunchKernelGGL((k_primary_packet<false>), gt, bt, 0, st, p, max_items);
	}
	else if (gen == GEN_PT)
	{
		if (count)
			RT_EXT(GEN_PT, true);
		else
			RT_EXT(GEN_PT, false);
	}
	else
	{
		if (count)
			RT_EXT(GEN_PARITY, true);
		else
			RT_EXT(GEN_PARITY, false);
	}
#undef RT_EXT
}

void launch_shade_parity(const Params &p, bool count, uint32_t max_items, stream_t s)
{
	const dim3 g(persistent_grid(max_items)), b(BLOCK);
	if (count)
		hipLaunchKernelGGL((k_shade_parity<true>), g, b, 0, (hipStream_t)s, p, max_items);
	else
		hipLaunchKernelGGL((k_shade_parity<false>), g, b, 0, (hipStream_t)s, p, max_items);
}

// slots the void entries of one shade launch over max_items paths can take beyond the rays themselves (per queue)
uint32_t queue_pad(uint32_t max_items)
{
	return std::max(persistent_grid(max_items, RT_SHADE_BLOCKS_PER_CU(false)), persistent_grid(max_items, RT_SHADE_BLOCKS_PER_CU(true))) * (BLOCK / 64u) * QUEUE_BLOCK;
}

void launch_shade_pt(const Params &p, uint32_t max_items, stream_t s)
{
#if defined(RT_DIAG_SHADE_CLOCK)
	static int launches = 0;
	if (++launches % 96 == 0)
	{
		unsigned long long h[32];
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shade_clk), sizeof(h));
		double tot = 0;
		for (int k = 0; k < 16; k++)
			tot += (double)h[k];
		fprintf(stderr, "[shade clock] after %d launches:", launches);
		for (int k = 0; k < 15; k++)
			fprintf(stderr, " %d: %.1f%% (%.0f cyc x %llu)", k, 100.0 * (double)h[k] / tot, h[16 + k] ? (double)h[k] / (double)h[16 + k] : 0.0, h[16 + k]);
		fprintf(stderr, "\n");
	}
#endif
	if (p.textured)
		hipLaunchKernelGGL(k_shade_pt<true>, dim3(persistent_grid(max_items, RT_SHADE_BLOCKS_PER_CU(true))), dim3(BLOCK), 0, (hipStream_t)s, p);
	else
		hipLaunchKernelGGL(k_shade_pt<false>, dim3(persistent_grid(max_items, RT_SHADE_BLOCKS_PER_CU(false))), dim3(BLOCK), 0, (hipStream_t)s, p);
}

void launch_shadow_packets(const Params &p, bool count, uint32_t max_items, stream_t s)
{
	const dim3 g(persistent_grid(max_items));
	const dim3 gt(std::max(8u, g.x * BLOCK / TRACE_BLOCK)), bt(TRACE_BLOCK);
	if (count)
		hipLaunchKernelGGL((k_shadow_packet<true>), gt, bt, 0, (hipStream_t)s, p);
	else
		hipLaunchKernelGGL((k_shadow_packet<false>), gt, bt, 0, (hipStream_t)s, p);
}

void launch_connect(const Params &p, bool count, uint32_t max_items, stream_t s)
{
	const dim3 g(persistent_grid(max_items)), b(BLOCK);
	if (p.refill & 2u)
	{
		const dim3 gt(std::max(8u, g.x * BLOCK / TRACE_BLOCK)), bt(TRACE_BLOCK);
		if (count)
			hipLaunchKernelGGL((k_trace_stream<true, true>), gt, bt, 0, (hipStream_t)s, p);
		else
			hipLaunchKernelGGL((k_trace_stream<true, false>), gt, bt, 0, (hipStream_t)s, p);
	}
	else if (count)
		hipLaunchKernelGGL((k_connect<true>), g, b, 0, (hipStream_t)s, p);
	else
		hipLaunchKernelGGL((k_connect<false>), g, b, 0, (hipStream_t)s, p);
}

void launch_trace_fused(const Params &pe, const Params &pa, bool count, uint32_t max_items, stream_t s)
{
#if defined(RT_DIAG_TRACE_CLOCK)
	static int launches = 0;
	if (++launches % 64 == 0)
	{
		unsigned long long h[3][8];
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace_clk), sizeof(h));
		const char *names[3] = {"ext", "any", "primary"};
		for (int m = 0; m < 3; m++)
		{
			double tot = 0;
			for (int k = 0; k < 4; k++)
				tot += (double)h[m][k];
			if (tot == 0)
				continue;
			fprintf(stderr, "[trace clock %s] refill %.1f%% (%.0f cyc x %llu)  nodes %.1f%% (%.0f)  leaves %.1f%% (%.0f)  retire %.1f%% (%.0f)\n", names[m],
					100 * h[m][0] / tot, (double)h[m][0] / (double)h[m][4], h[m][4], 100 * h[m][1] / tot, (double)h[m][1] / (double)h[m][5],
					100 * h[m][2] / tot, (double)h[m][2] / (double)h[m][6], 100 * h[m][3] / tot, (double)h[m][3] / (double)h[m][7]);
		}
	}
#endif
	const dim3 g(persistent_grid(max_items));
	const dim3 gt(std::max(8u, g.x * BLOCK / TRACE_BLOCK)), bt(TRACE_BLOCK);
	if (count)
		hipLaunchKernelGGL((k_trace_fused<true>), gt, bt, 0, (hipStream_t)s, pe, pa);
	else
		hipLaunchKernelGGL((k_trace_fused<false>), gt, bt, 0, (hipStream_t)s, pe, pa);
}

void launch_resolve(const Params &p, stream_t s)
{
	hipLaunchKernelGGL(k_resolve, dim3(persistent_grid(p.fr.W * p.fr.local_rows, RT_GRID_BLOCKS_PER_CU, 8u)), dim3(BLOCK), 0, (hipStream_t)s, p);
}

void launch_present(const Params &p, f4 *out, float scale, int full, stream_t s)
{
	hipLaunchKernelGGL(k_present, dim3(persistent_grid(p.fr.W * p.fr.local_rows, RT_GRID_BLOCKS_PER_CU, 8u)), dim3(BLOCK), 0, (hipStream_t)s, p,
					   out, scale, full);
}

void launch_deinterleave(const f4 *gathered, f4 *out, uint32_t W, uint32_t H, uint32_t local_rows, uint32_t world,
						 stream_t s)
{
	hipLaunchKernelGGL(k_deinterleave, dim3(persistent_grid(W * H, RT_GRID_BLOCKS_PER_CU, 8u)), dim3(BLOCK), 0, (hipStream_t)s, gathered, out, W, H,
					   local_rows, world);
}

__global__ void __launch_bounds__(BLOCK) k_skin_vertices(f4 *verts, f4 *vnormals, const f4 *base_verts, const f4 *base_normals,
														 const uint32_t *joints4, const f4 *weights4, const float *mats,
														 uint32_t joint_count, uint32_t vertex_count)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < vertex_count)
		skin_vertex_item(verts, vnormals, base_verts, base_normals, joints4, weights4, mats, joint_count, i);
}
__global__ void __launch_bounds__(BLOCK) k_morph_vertices(f4 *verts, f4 *vnormals, const f4 *base_verts, const f4 *base_normals,
															  const f4 *tgt_pos, const f4 *tgt_nrm, const float *weights,
															  uint32_t target_count, uint32_t vertex_count)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < vertex_count)
		morph_vertex_item(verts, vnormals, base_verts, base_normals, tgt_pos, tgt_nrm, weights, target_count, vertex_count, i);
}
void launch_morph_vertices(f4 *verts, f4 *vnormals, const f4 *base_verts, const f4 *base_normals, const f4 *tgt_pos,
						   const f4 *tgt_nrm, const float *weights, uint32_t target_count, uint32_t vertex_count, stream_t s)
{
	if (vertex_count)
		hipLaunchKernelGGL(k_morph_vertices, dim3((vertex_count + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, verts,
						   vnormals, base_verts, base_normals, tgt_pos, tgt_nrm, weights, target_count, vertex_count);
}
__global__ void __launch_bounds__(BLOCK) k_skin_shade(TriShade *shade, const f4 *verts, const f4 *vnormals, const uint32_t *indices,
													  uint32_t tri_count)
{
	const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < tri_count)
		skin_shade_item(shade, verts, vnormals, indices, i);
}
void launch_skin_vertices(f4 *verts, f4 *vnormals, const f4 *base_verts, const f4 *base_normals, const uint32_t *joints4,
						  const f4 *weights4, const float *mats, uint32_t joint_count, uint32_t vertex_count, stream_t s)
{
	if (vertex_count)
		hipLaunchKernelGGL(k_skin_vertices, dim3((vertex_count + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, verts,
						   vnormals, base_verts, base_normals, joints4, weights4, mats, joint_count, vertex_count);
}
void launch_skin_shade(TriShade *shade, const f4 *verts, const f4 *vnormals, const uint32_t *indices, uint32_t tri_count, stream_t s)
{
	if (tri_count)
		hipLaunchKernelGGL(k_skin_shade, dim3((tri_count + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, shade, verts,
						   vnormals, indices, tri_count);
}
void launch_refresh4(Node4c *nodes4, const uint32_t *src4, uint32_t count4, const Node *blas_nodes2, stream_t s)
{
	if (count4)
		hipLaunchKernelGGL(k_refresh4, dim3((count4 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)s, nodes4, src4, count4, blas_nodes2);
}

void launch_refit(Node *nodes, uint32_t node_base, const int *parents, uint32_t node_count, f4 *tri_verts,
				  uint32_t tri_base, const f4 *verts, const uint32_t *indices, uint32_t tri_count, uint32_t *flags, stream_t s)
{
	hipStream_t st = (hipStream_t)s;
	(void)hipMemsetAsync(flags, 0, sizeof(uint32_t) * node_count, st);
	hipLaunchKernelGGL(k_refit_tris, dim3((tri_count + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, tri_verts + 3ull * tri_base,
					   verts, indices, tri_count);
	hipLaunchKernelGGL(k_refit_nodes, dim3((node_count + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, nodes, node_base, parents,
					   node_count, tri_verts, flags);
}

// ================================================================================================================
#else // RFWHIP_HOST_EMULATION: the same work items driven by plain loops (tests/_emu only — never shipped)
#include "kernels_emu.inc"
#endif

} // namespace rtk
