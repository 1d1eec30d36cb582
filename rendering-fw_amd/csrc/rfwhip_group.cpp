// rfwhip_group.cpp — multi-GPU below the C ABI: the strip split of SURVEY §8(e) driven by host C++.
//
// The path shards by pixels: every device holds the whole scene and renders the 8-row strips it owns
// (rt::strip_owner); per presented frame the rank-local strips are gathered ONCE into the root's staging image
// [world][local_rows][width] and de-interleaved there.  Two front ends over one implementation:
//
//   rfwhip_group_*   ONE process, ONE host thread, n devices — how the reference's host runs (RFW/system/src/rfw/app.cpp:3-26:
//                    one thread owns the render loop): n contexts, every call enqueues on all of them, nothing blocks the
//                    host between the render calls and the gather;
//   rfwhip_comm_*    one process per device (torch.distributed.run style): each process owns one context; the processes
//                    only exchange a 128-byte id out of band, the data path is this library's RCCL calls.
//
// Transport of the gather: RCCL point-to-point (ncclSend from every rank, ncclRecv x (world - 1) on the root inside one
// ncclGroup — on xGMI every peer has its own link to the root, so the n - 1 transfers run side by side), or peer copies
// (hipMemcpyPeerAsync pushed on the source device's stream; also what n contexts on ONE device use in the tests, where
// RCCL refuses duplicate devices).  Everything is stream-ordered: present -> transfer -> de-interleave run on one gather
// stream per device behind the frame's kernels, and the next frame's kernels overlap them.
//
// Built on the public C ABI of rfwhip.h only (no access to the context's internals); librccl is opened on first use.
#include "rfwhip.h"

#include "internal.h"

#include <stdint.h>
#include <string.h>
#include <vector>

#if !defined(RFWHIP_HOST_EMULATION)
#include <stdlib.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#else
#include <stdlib.h>
#endif

#define GR_TRY(x)            \
	do                       \
	{                        \
		const int rc_ = (x); \
		if (rc_)             \
			return rc_;      \
	} while (0)

namespace
{
constexpr size_t PIXEL_BYTES = 16; // float4

// ---- device layer: HIP, or heap memory in the host-emulation build of the tests ------------------------------------
#if !defined(RFWHIP_HOST_EMULATION)
#define GR_HIP(x)                                                                                                   \
	do                                                                                                              \
	{                                                                                                               \
		const hipError_t e_ = (x);                                                                                  \
		if (e_ != hipSuccess)                                                                                       \
			return rfwhip_internal_set_error(RFWHIP_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)
int dev_use(int device)
{
	GR_HIP(hipSetDevice(device));
	return 0;
}
int dev_alloc(void **p, size_t bytes)
{
	GR_HIP(hipMalloc(p, bytes ? bytes : 16));
	return 0;
}
void dev_free(void *p)
{
	if (p)
		(void)hipFree(p);
}
int stream_create(void **s)
{
	hipStream_t st;
	GR_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	*s = st;
	return 0;
}
void stream_destroy(void *s)
{
	if (s)
		(void)hipStreamDestroy((hipStream_t)s);
}
int stream_sync(void *s)
{
	GR_HIP(hipStreamSynchronize((hipStream_t)s));
	return 0;
}
typedef hipEvent_t event_t;
int event_create(event_t *e)
{
	GR_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
	return 0;
}
void event_destroy(event_t e) { (void)hipEventDestroy(e); }
int event_record(event_t e, void *s)
{
	GR_HIP(hipEventRecord(e, (hipStream_t)s));
	return 0;
}
int stream_wait(void *s, event_t e)
{
	GR_HIP(hipStreamWaitEvent((hipStream_t)s, e, 0));
	return 0;
}
// dst on dst_device <- src on src_device, enqueued on `s` (a stream of the source device: the copy is pushed)
int copy_async(void *dst, int dst_device, const void *src, int src_device, size_t bytes, void *s)
{
	if (dst_device == src_device)
		GR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
	else
		GR_HIP(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t)s));
	return 0;
}
int copy_to_host(void *dst, const void *src, size_t bytes, void *s)
{
	GR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
	GR_HIP(hipStreamSynchronize((hipStream_t)s));
	return 0;
}
int host_alloc(void **p, size_t bytes) // pinned: the device-to-host copy of a presented frame runs asynchronously
{
	GR_HIP(hipHostMalloc(p, bytes ? bytes : 16, hipHostMallocDefault));
	return 0;
}
void host_free(void *p)
{
	if (p)
		(void)hipHostFree(p);
}
int copy_to_host_async(void *dst, const void *src, size_t bytes, void *s)
{
	GR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
	return 0;
}
int event_sync(event_t e)
{
	GR_HIP(hipEventSynchronize(e));
	return 0;
}
int device_count()
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
void enable_peer(int a, int b) // best effort: without it hipMemcpyPeerAsync stages through the host
{
	int can = 0;
	if (a == b || hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can)
		return;
	if (hipSetDevice(a) == hipSuccess)
		(void)hipDeviceEnablePeerAccess(b, 0); // (hipErrorPeerAccessAlreadyEnabled is fine)
	(void)hipGetLastError();
}

// ---- RCCL, opened on first use (a single-GPU host never loads it) ----------------------------------------------------
struct Rccl
{
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool tried = false;
	bool load()
	{
		if (lib)
			return true;
		if (tried)
			return false;
		tried = true;
		// RFWHIP_RCCL_LIBRARY: another library with the same eight entry points (tests/stub/rccl_stub.cpp walks the RCCL
		// branch's enqueue order on one device; a site with its own RCCL build points here too)
		const char *override_lib = getenv("RFWHIP_RCCL_LIBRARY");
		const char *names[] = {override_lib ? override_lib : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *n : names)
		{
			if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
				break;
			if (override_lib) // the named library or nothing: no silent change of transport implementation
				break;
		}
		if (!lib)
			return false;
#define GR_SYM(F, N) F = (decltype(F))dlsym(lib, N)
		GR_SYM(GetUniqueId, "ncclGetUniqueId"), GR_SYM(CommInitRank, "ncclCommInitRank"), GR_SYM(CommDestroy, "ncclCommDestroy");
		GR_SYM(GroupStart, "ncclGroupStart"), GR_SYM(GroupEnd, "ncclGroupEnd"), GR_SYM(Send, "ncclSend"), GR_SYM(Recv, "ncclRecv");
		GR_SYM(GetErrorString, "ncclGetErrorString");
#undef GR_SYM
		if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv)
		{
			dlclose(lib), lib = nullptr;
			return false;
		}
		return true;
	}
	const char *err(ncclResult_t r) const { return GetErrorString ? GetErrorString(r) : "?"; }
};
Rccl g_rccl;
#define GR_NCCL(x)                                                                                                            \
	do                                                                                                                        \
	{                                                                                                                         \
		const ncclResult_t r_ = (x);                                                                                          \
		if (r_ != ncclSuccess)                                                                                                \
			return rfwhip_internal_set_error(RFWHIP_ERR_HIP, "RCCL: %s failed: %s (%s:%d)", #x, g_rccl.err(r_), __FILE__, __LINE__); \
	} while (0)
typedef ncclComm_t comm_t;
#else
int dev_use(int) { return 0; }
int dev_alloc(void **p, size_t bytes)
{
	*p = calloc(bytes ? bytes : 16, 1);
	return *p ? 0 : rfwhip_internal_set_error(RFWHIP_ERR_HIP, "out of memory");
}
void dev_free(void *p) { free(p); }
int stream_create(void **s)
{
	*s = nullptr;
	return 0;
}
void stream_destroy(void *) {}
int stream_sync(void *) { return 0; }
typedef int event_t;
int event_create(event_t *e)
{
	*e = 0;
	return 0;
}
void event_destroy(event_t) {}
int event_record(event_t, void *) { return 0; }
int stream_wait(void *, event_t) { return 0; }
int copy_async(void *dst, int, const void *src, int, size_t bytes, void *)
{
	memmove(dst, src, bytes);
	return 0;
}
int copy_to_host(void *dst, const void *src, size_t bytes, void *)
{
	memcpy(dst, src, bytes);
	return 0;
}
int host_alloc(void **p, size_t bytes) { return dev_alloc(p, bytes); }
void host_free(void *p) { free(p); }
int copy_to_host_async(void *dst, const void *src, size_t bytes, void *)
{
	memcpy(dst, src, bytes);
	return 0;
}
int event_sync(event_t) { return 0; }
int device_count() { return 1 << 20; }
void enable_peer(int, int) {}
typedef void *comm_t;
#endif

// One rank as this process sees it.
struct Endpoint
{
	rfwhip_context *ctx = nullptr;
	int device = 0, rank = 0;
	void *stream = nullptr;	  // the gather chain of this device: present -> transfer (-> de-interleave on the root)
	void *local_fb = nullptr; // [local_rows][width] float4: where this rank's present lands (ranks other than the root)
	comm_t comm = nullptr;
	event_t sent;			  // peer transport: the push of this rank's strips is enqueued up to here
	bool have_event = false;
};

} // namespace

// What both front ends share: the local endpoints of a world, and on the process that owns rank 0 the staging + full image.
struct rfwhip_group
{
	std::vector<Endpoint> ep; // local ranks (all of them for a group, one for a comm)
	int world = 1, transport = RFWHIP_TRANSPORT_PEER;
	bool loopback = false; // rfwhip_comm_create with an id and world 1: a REAL one-rank RCCL communicator; the gather sends the strips to
						   // itself (ncclSend + ncclRecv on rank 0 in one group) — the library's RCCL calls on a box with one device
	bool owns_contexts = false;
	uint32_t W = 0, H = 0, local_rows = 0;
	void *staging = nullptr, *full = nullptr; // on the root's device
	// peer transport: recorded on the root's stream once a frame's de-interleave has read the staging image; every rank's next
	// push waits for it (frames in flight: without it a fast rank overwrites its chunk while the root still waits for a slow one)
	event_t staging_read;
	bool staging_event = false, staging_read_valid = false;
	bool poisoned = false; // a call failed half-way through the ranks: sample counts / ring slots diverge, re-init needed
	int root_local = -1;					  // index of rank 0 in ep, -1 when another process owns it
	// pipelined presentation (rfwhip_group_present_async / _wait): two pinned host images and the events of their copies
	static constexpr int SLOTS = RFWHIP_PRESENT_SLOTS;
	void *host_img[SLOTS] = {};
	void *slot_img[SLOTS] = {};	 // the de-interleaved image of each slot on the root's device (the copy's source)
	void *copy_stream = nullptr; // the device-to-host copies run beside the gather chain, not inside it
	event_t host_ready[SLOTS], slot_done[SLOTS];
	bool host_events = false, host_pending[SLOTS] = {};
	size_t chunk_bytes() const { return (size_t)local_rows * W * PIXEL_BYTES; }
};
struct rfwhip_comm
{
	rfwhip_group g;
};

namespace
{

void release_buffers(rfwhip_group *g)
{
	for (auto &e : g->ep)
	{
		(void)dev_use(e.device);
		dev_free(e.local_fb), e.local_fb = nullptr;
	}
	if (g->root_local >= 0)
	{
		(void)dev_use(g->ep[g->root_local].device);
		dev_free(g->staging), dev_free(g->full);
	}
	g->staging = g->full = nullptr;
	g->staging_read_valid = false;
	for (int k = 0; k < rfwhip_group::SLOTS; k++)
	{
		host_free(g->host_img[k]), g->host_img[k] = nullptr, g->host_pending[k] = false;
		dev_free(g->slot_img[k]), g->slot_img[k] = nullptr;
	}
}

// (re)allocate the gather buffers for the contexts' current render target
int size_buffers(rfwhip_group *g, uint32_t W, uint32_t H)
{
	release_buffers(g);
	g->W = W, g->H = H;
	g->local_rows = g->ep.empty() ? 0 : rfwhip_local_rows(g->ep[0].ctx);
	for (auto &e : g->ep)
	{
		if (rfwhip_local_rows(e.ctx) != g->local_rows)
			return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "ranks disagree about the padded strip rows");
		GR_TRY(dev_use(e.device));
		if (e.rank != 0 || g->loopback)
			GR_TRY(dev_alloc(&e.local_fb, g->chunk_bytes()));
	}
	if (g->root_local >= 0)
	{
		GR_TRY(dev_use(g->ep[g->root_local].device));
		GR_TRY(dev_alloc(&g->staging, g->chunk_bytes() * (size_t)g->world));
		GR_TRY(dev_alloc(&g->full, (size_t)W * H * PIXEL_BYTES));
	}
	return 0;
}

int make_streams(rfwhip_group *g)
{
	for (auto &e : g->ep)
	{
		GR_TRY(dev_use(e.device));
		GR_TRY(stream_create(&e.stream));
		GR_TRY(event_create(&e.sent));
		e.have_event = true;
	}
	return 0;
}

#if !defined(RFWHIP_HOST_EMULATION)
int init_rccl(rfwhip_group *g, const ncclUniqueId &id)
{
	// every local rank joins the communicator; inside one ncclGroup so that a single thread can own several ranks
	GR_NCCL(g_rccl.GroupStart());
	const int rc_group = [&]() -> int {
		for (auto &e : g->ep)
		{
			GR_TRY(dev_use(e.device));
			GR_NCCL(g_rccl.CommInitRank(&e.comm, g->world, id, e.rank));
		}
		return 0;
	}();
	const ncclResult_t r_end = g_rccl.GroupEnd(); // (closed on every path)
	GR_TRY(rc_group);
	GR_NCCL(r_end);
	return 0;
}
#endif

// The gather of one presented frame, enqueue only.  full_out: where the root's de-interleaved image goes (null: the group's
// own buffer).
int gather(rfwhip_group *g, void *full_out)
{
	if (!g->W)
		return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "gather before init");
	const size_t chunk = g->chunk_bytes();
	Endpoint *root = g->root_local >= 0 ? &g->ep[g->root_local] : nullptr;
	// 1. every local rank presents its strips on its gather stream (ordered behind its frame by the context; the context's
	//    next render waits on the device until the present has read the accumulator)
	for (auto &e : g->ep)
	{
		void *dst = (e.rank == 0 && !g->loopback) ? g->staging : e.local_fb;
		if (rfwhip_read_local_framebuffer_stream(e.ctx, dst, e.stream))
			return RFWHIP_ERR_STATE; // (the context's message stands)
	}
#if !defined(RFWHIP_HOST_EMULATION)
	if (g->loopback && root)
	{
		// world 1 over RCCL: the one rank sends its strips to itself — same calls, same streams as the root's side of a real gather
		const size_t count = chunk / sizeof(float);
		GR_TRY(dev_use(root->device));
		GR_NCCL(g_rccl.GroupStart());
		const ncclResult_t r_send = g_rccl.Send(root->local_fb, count, ncclFloat, 0, root->comm, (hipStream_t)root->stream);
		const ncclResult_t r_recv = r_send == ncclSuccess ? g_rccl.Recv(g->staging, count, ncclFloat, 0, root->comm, (hipStream_t)root->stream) : r_send;
		const ncclResult_t r_end = g_rccl.GroupEnd();
		GR_NCCL(r_send);
		GR_NCCL(r_recv);
		GR_NCCL(r_end);
	}
#endif
	if (g->world > 1)
	{
#if !defined(RFWHIP_HOST_EMULATION)
		if (g->transport == RFWHIP_TRANSPORT_RCCL)
		{
			// 2a. one ncclGroup: a send per non-root rank, world - 1 receives on the root, each pair on its own xGMI link
			const size_t count = chunk / sizeof(float);
			GR_NCCL(g_rccl.GroupStart());
			// (whatever fails inside, the group is closed again: an open ncclGroup would swallow every later call)
			const int rc_group = [&]() -> int {
				for (auto &e : g->ep)
					if (e.rank != 0)
					{
						GR_TRY(dev_use(e.device));
						GR_NCCL(g_rccl.Send(e.local_fb, count, ncclFloat, 0, e.comm, (hipStream_t)e.stream));
					}
				if (root)
				{
					GR_TRY(dev_use(root->device));
					for (int r = 1; r < g->world; r++)
						GR_NCCL(g_rccl.Recv((char *)g->staging + (size_t)r * chunk, count, ncclFloat, r, root->comm, (hipStream_t)root->stream));
				}
				return 0;
			}();
			const ncclResult_t r_end = g_rccl.GroupEnd();
			GR_TRY(rc_group);
			GR_NCCL(r_end);
		}
		else
#endif
		{
			// 2b. peer copies, pushed by the source device behind its present; the root's stream waits for each of them
			if (!root)
				return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "the peer transport needs every rank in one process");
			for (auto &e : g->ep)
				if (e.rank != 0)
				{
					GR_TRY(dev_use(e.device));
					// (the root's de-interleave of the PREVIOUS frame must have read this rank's chunk of the staging image before the
					// push overwrites it: the event is recorded behind that de-interleave, step 3, and re-recorded only after this
					// wait has been enqueued — with frames in flight the push of frame k + 1 would otherwise race the read of frame k)
					if (g->staging_read_valid)
						GR_TRY(stream_wait(e.stream, g->staging_read));
					GR_TRY(copy_async((char *)g->staging + (size_t)e.rank * chunk, root->device, e.local_fb, e.device, chunk, e.stream));
					GR_TRY(event_record(e.sent, e.stream));
				}
			GR_TRY(dev_use(root->device));
			for (auto &e : g->ep)
				if (e.rank != 0)
					GR_TRY(stream_wait(root->stream, e.sent));
		}
	}
	// 3. the root undoes the strip interleave
	if (root)
	{
		if (g->world > 1)
		{
			if (rfwhip_deinterleave_stream(root->ctx, g->staging, full_out ? full_out : g->full, root->stream))
				return RFWHIP_ERR_STATE;
			if (g->transport != RFWHIP_TRANSPORT_RCCL) // (RCCL: the receives sit on the root's stream behind the de-interleave)
			{
				if (!g->staging_event)
				{
					GR_TRY(event_create(&g->staging_read));
					g->staging_event = true;
				}
				GR_TRY(event_record(g->staging_read, root->stream));
				g->staging_read_valid = true;
			}
		}
		else
			GR_TRY(copy_async(full_out ? full_out : g->full, root->device, g->staging, root->device, (size_t)g->W * g->H * PIXEL_BYTES, root->stream));
	}
	return 0;
}

int wait_all(rfwhip_group *g)
{
	// every rank is waited for, whatever one of them reports: the first error comes back afterwards (the contexts keep their
	// own messages; rfwhip_last_error() holds the last one set)
	int first = 0;
	for (auto &e : g->ep)
	{
		if (rfwhip_wait(e.ctx) && !first)
			first = RFWHIP_ERR_STATE;
		int rc = dev_use(e.device);
		if (!rc)
			rc = stream_sync(e.stream);
		if (rc && !first)
			first = rc;
	}
	if (first)
		return first;
	if (g->copy_stream && g->root_local >= 0)
	{
		GR_TRY(dev_use(g->ep[(size_t)g->root_local].device));
		GR_TRY(stream_sync(g->copy_stream));
	}
	return 0;
}

void destroy_group(rfwhip_group *g)
{
	for (auto &e : g->ep)
	{
		(void)dev_use(e.device);
		if (e.stream)
			(void)stream_sync(e.stream);
	}
	if (g->copy_stream && g->root_local >= 0)
	{
		(void)dev_use(g->ep[(size_t)g->root_local].device);
		(void)stream_sync(g->copy_stream);
	}
	release_buffers(g);
	for (auto &e : g->ep)
	{
		(void)dev_use(e.device);
#if !defined(RFWHIP_HOST_EMULATION)
		if (e.comm && g_rccl.lib)
			(void)g_rccl.CommDestroy(e.comm);
#endif
		if (e.have_event)
			event_destroy(e.sent);
		stream_destroy(e.stream);
		if (g->owns_contexts && e.ctx)
			rfwhip_destroy(e.ctx);
	}
	if (g->host_events)
	{
		for (int k = 0; k < rfwhip_group::SLOTS; k++)
			event_destroy(g->host_ready[k]), event_destroy(g->slot_done[k]);
		g->host_events = false;
	}
	if (g->staging_event)
		event_destroy(g->staging_read), g->staging_event = false;
	stream_destroy(g->copy_stream), g->copy_stream = nullptr;
	g->ep.clear();
}

int resolve_transport(int transport, bool distinct_devices, int *out)
{
	if (transport != RFWHIP_TRANSPORT_AUTO && transport != RFWHIP_TRANSPORT_RCCL && transport != RFWHIP_TRANSPORT_PEER)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "unknown transport %d", transport);
#if defined(RFWHIP_HOST_EMULATION)
	if (transport == RFWHIP_TRANSPORT_RCCL)
		return rfwhip_internal_set_error(RFWHIP_ERR_UNSUPPORTED, "the host-emulation build has no RCCL");
	*out = RFWHIP_TRANSPORT_PEER;
#else
	// (RFWHIP_RCCL_SHARED_DEVICE=1: for a stand-in library that can serve several ranks on one device — the stub of the tests)
	const char *shared = getenv("RFWHIP_RCCL_SHARED_DEVICE");
	if (shared && shared[0] == '1' && getenv("RFWHIP_RCCL_LIBRARY"))
		distinct_devices = true;
	if (transport == RFWHIP_TRANSPORT_RCCL && !distinct_devices)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "RCCL needs one device per rank (a device is listed twice)");
	if (transport == RFWHIP_TRANSPORT_RCCL && !g_rccl.load())
		return rfwhip_internal_set_error(RFWHIP_ERR_UNSUPPORTED, "librccl.so could not be opened: %s", dlerror());
	if (transport == RFWHIP_TRANSPORT_AUTO)
		transport = (distinct_devices && g_rccl.load()) ? RFWHIP_TRANSPORT_RCCL : RFWHIP_TRANSPORT_PEER;
	*out = transport;
#endif
	return 0;
}

} // namespace

// =================================================================================================================
// one process, n devices
// =================================================================================================================
extern "C" int rfwhip_group_create(const int *devices, int n, int transport, rfwhip_group **out)
{
	if (!devices || !out || n < 1 || n > 64)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_group_create: bad arguments (n = %d)", n);
	bool distinct = true;
	for (int i = 0; i < n; i++)
	{
		if (devices[i] < 0 || devices[i] >= device_count())
			return rfwhip_internal_set_error(RFWHIP_ERR_NO_DEVICE, "rfwhip_group_create: device ordinal %d out of range", devices[i]);
		for (int j = 0; j < i; j++)
			distinct = distinct && devices[i] != devices[j];
	}
	int tr = RFWHIP_TRANSPORT_PEER;
	if (n > 1)
		GR_TRY(resolve_transport(transport, distinct, &tr));
	rfwhip_group *g = new rfwhip_group();
	g->world = n, g->transport = tr, g->owns_contexts = true, g->root_local = 0;
	g->ep.resize((size_t)n);
	int rc = 0;
	for (int i = 0; i < n && !rc; i++)
	{
		g->ep[i].device = devices[i], g->ep[i].rank = i;
		rc = rfwhip_create(devices[i], i, n, &g->ep[i].ctx);
	}
	if (!rc)
		rc = make_streams(g);
#if !defined(RFWHIP_HOST_EMULATION)
	if (!rc && n > 1 && tr == RFWHIP_TRANSPORT_RCCL)
	{
		ncclUniqueId id;
		const ncclResult_t r = g_rccl.GetUniqueId(&id);
		rc = r == ncclSuccess ? init_rccl(g, id) : rfwhip_internal_set_error(RFWHIP_ERR_HIP, "RCCL: ncclGetUniqueId failed: %s", g_rccl.err(r));
	}
	if (!rc && tr == RFWHIP_TRANSPORT_PEER)
		for (int i = 1; i < n; i++)
			enable_peer(devices[i], devices[0]), enable_peer(devices[0], devices[i]);
#endif
	if (rc)
	{
		destroy_group(g);
		delete g;
		return rc;
	}
	*out = g;
	return RFWHIP_OK;
}

extern "C" void rfwhip_group_destroy(rfwhip_group *g)
{
	if (!g)
		return;
	destroy_group(g);
	delete g;
}

extern "C" int rfwhip_group_size(const rfwhip_group *g) { return g ? g->world : 0; }
extern "C" int rfwhip_group_transport(const rfwhip_group *g) { return g ? g->transport : 0; }

extern "C" rfwhip_context *rfwhip_group_context(rfwhip_group *g, int rank)
{
	return (g && rank >= 0 && rank < (int)g->ep.size()) ? g->ep[(size_t)rank].ctx : nullptr;
}

extern "C" int rfwhip_group_init(rfwhip_group *g, uint32_t width, uint32_t height)
{
	if (!g)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null group");
	(void)wait_all(g); // a gather in flight still uses the buffers (a poisoned group may report its old error here)
	for (auto &e : g->ep)
		if (rfwhip_init(e.ctx, width, height))
			return RFWHIP_ERR_STATE;
	g->poisoned = false; // every rank starts from sample 0 of a fresh target again
	return size_buffers(g, width, height);
}

extern "C" int rfwhip_group_update(rfwhip_group *g)
{
	if (!g)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null group");
	int first = 0; // every rank is updated; the first error comes back afterwards
	for (auto &e : g->ep)
	{
		const int rc = rfwhip_update(e.ctx);
		if (rc && !first)
			first = rc;
	}
	return first;
}

extern "C" int rfwhip_group_set_setting(rfwhip_group *g, const char *key, const char *value)
{
	if (!g)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null group");
	for (auto &e : g->ep)
	{
		const int rc = rfwhip_set_setting(e.ctx, key, value);
		if (rc)
			return rc;
	}
	return RFWHIP_OK;
}

extern "C" int rfwhip_group_render(rfwhip_group *g, const rfwhip_camera *camera, int status)
{
	if (!g)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null group");
	if (g->poisoned && status != RFWHIP_RESET)
		return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "rfwhip_group_render: an earlier call failed on some ranks only — the ranks' sample "
																 "counts differ; render with RFWHIP_RESET (or call rfwhip_group_init) first");
	int first = 0;
	for (auto &e : g->ep) // enqueue only: the n devices render side by side; every rank is asked even when one fails
	{
		const int rc = rfwhip_render(e.ctx, camera, status);
		if (rc && !first)
			first = rc;
	}
	// a RESET that every rank took puts them in step again; a failure on some ranks leaves their sample counts apart
	g->poisoned = first != 0;
	return first;
}

extern "C" int rfwhip_group_gather(rfwhip_group *g)
{
	if (!g)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null group");
	return gather(g, nullptr);
}

extern "C" int rfwhip_group_wait(rfwhip_group *g)
{
	if (!g)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null group");
	return wait_all(g);
}

extern "C" int rfwhip_group_read_framebuffer(rfwhip_group *g, float *rgba_host)
{
	if (!g || !rgba_host)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null argument");
	GR_TRY(gather(g, nullptr));
	GR_TRY(wait_all(g));
	Endpoint &root = g->ep[(size_t)g->root_local];
	GR_TRY(dev_use(root.device));
	return copy_to_host(rgba_host, g->full, (size_t)g->W * g->H * PIXEL_BYTES, root.stream);
}

// Pipelined presentation: frame k's image travels to the host while the next frames render.  present_async enqueues gather +
// device-to-host copy into pinned host image `slot` (0 .. RFWHIP_PRESENT_SLOTS - 1) and returns; present_wait blocks until
// that copy has landed.  With n slots a host keeps n frames in flight: render(k), present_async(k % n), present_wait((k + 1) % n).
extern "C" int rfwhip_group_present_async(rfwhip_group *g, int slot)
{
	if (!g || slot < 0 || slot >= rfwhip_group::SLOTS)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_group_present_async: bad arguments");
	if (g->root_local < 0 || !g->full)
		return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "no render target");
	Endpoint &root = g->ep[(size_t)g->root_local];
	GR_TRY(dev_use(root.device));
	const size_t bytes = (size_t)g->W * g->H * PIXEL_BYTES;
	if (!g->host_events)
	{
		for (int k = 0; k < rfwhip_group::SLOTS; k++)
		{
			GR_TRY(event_create(&g->host_ready[k]));
			GR_TRY(event_create(&g->slot_done[k]));
		}
		GR_TRY(stream_create(&g->copy_stream));
		g->host_events = true;
	}
	if (!g->host_img[slot])
		GR_TRY(host_alloc(&g->host_img[slot], bytes));
	if (!g->slot_img[slot])
		GR_TRY(dev_alloc(&g->slot_img[slot], bytes));
	// the slot's device image is rewritten only after its previous copy to the host has read it
	if (g->host_pending[slot])
		GR_TRY(stream_wait(root.stream, g->host_ready[slot]));
	GR_TRY(gather(g, g->slot_img[slot]));
	GR_TRY(dev_use(root.device));
	GR_TRY(event_record(g->slot_done[slot], root.stream));
	// ... and the copy rides its own stream: the next frame's present / transfer / de-interleave do not queue behind 33 MB of PCIe
	GR_TRY(stream_wait(g->copy_stream, g->slot_done[slot]));
	GR_TRY(copy_to_host_async(g->host_img[slot], g->slot_img[slot], bytes, g->copy_stream));
	GR_TRY(event_record(g->host_ready[slot], g->copy_stream));
	g->host_pending[slot] = true;
	return RFWHIP_OK;
}

extern "C" int rfwhip_group_present_wait(rfwhip_group *g, int slot, const float **rgba_host)
{
	if (!g || slot < 0 || slot >= rfwhip_group::SLOTS || !rgba_host)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_group_present_wait: bad arguments");
	if (!g->host_img[slot])
		return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "rfwhip_group_present_wait: nothing was presented into slot %d", slot);
	if (g->host_pending[slot])
	{
		GR_TRY(dev_use(g->ep[(size_t)g->root_local].device));
		GR_TRY(event_sync(g->host_ready[slot]));
		g->host_pending[slot] = false;
	}
	*rgba_host = (const float *)g->host_img[slot];
	return RFWHIP_OK;
}

extern "C" int rfwhip_group_framebuffer_device(rfwhip_group *g, void **rgba_device, int *device_ordinal)
{
	if (!g || !rgba_device)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null argument");
	if (!g->full)
		return rfwhip_internal_set_error(RFWHIP_ERR_STATE, "no render target");
	*rgba_device = g->full;
	if (device_ordinal)
		*device_ordinal = g->ep[(size_t)g->root_local].device;
	return RFWHIP_OK;
}

// =================================================================================================================
// one process per device
// =================================================================================================================
extern "C" int rfwhip_comm_unique_id(void *id, size_t cap)
{
	if (!id || cap < RFWHIP_COMM_ID_BYTES)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_comm_unique_id: the id takes %d bytes", RFWHIP_COMM_ID_BYTES);
#if defined(RFWHIP_HOST_EMULATION)
	return rfwhip_internal_set_error(RFWHIP_ERR_UNSUPPORTED, "the host-emulation build has no RCCL");
#else
	static_assert(sizeof(ncclUniqueId) == RFWHIP_COMM_ID_BYTES, "id size");
	if (!g_rccl.load())
		return rfwhip_internal_set_error(RFWHIP_ERR_UNSUPPORTED, "librccl.so could not be opened: %s", dlerror());
	ncclUniqueId u;
	GR_NCCL(g_rccl.GetUniqueId(&u));
	memcpy(id, &u, sizeof(u));
	return RFWHIP_OK;
#endif
}

extern "C" int rfwhip_comm_create(rfwhip_context *ctx, const void *id, rfwhip_comm **out)
{
	if (!ctx || !out)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null argument");
	int device = 0, rank = 0, world = 1;
	GR_TRY(rfwhip_get_placement(ctx, &device, &rank, &world));
	if (world > 1 && !id)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "rfwhip_comm_create: world %d needs the id of rfwhip_comm_unique_id", world);
	rfwhip_comm *c = new rfwhip_comm();
	rfwhip_group *g = &c->g;
	g->world = world, g->owns_contexts = false, g->root_local = rank == 0 ? 0 : -1;
	g->transport = world > 1 ? RFWHIP_TRANSPORT_RCCL : RFWHIP_TRANSPORT_PEER;
#if !defined(RFWHIP_HOST_EMULATION)
	if (world == 1 && id) // (an id for a world of one: the caller wants the RCCL path, e.g. to see it work before a multi-GPU run)
		g->transport = RFWHIP_TRANSPORT_RCCL, g->loopback = true;
#endif
	g->ep.resize(1);
	g->ep[0].ctx = ctx, g->ep[0].device = device, g->ep[0].rank = rank;
	int rc = make_streams(g);
#if !defined(RFWHIP_HOST_EMULATION)
	if (!rc && (world > 1 || g->loopback))
	{
		if (!g_rccl.load())
			rc = rfwhip_internal_set_error(RFWHIP_ERR_UNSUPPORTED, "librccl.so could not be opened: %s", dlerror());
		else
		{
			ncclUniqueId u;
			memcpy(&u, id, sizeof(u));
			rc = init_rccl(g, u);
		}
	}
#else
	if (!rc && world > 1)
		rc = rfwhip_internal_set_error(RFWHIP_ERR_UNSUPPORTED, "the host-emulation build has no RCCL");
#endif
	if (rc)
	{
		destroy_group(g);
		delete c;
		return rc;
	}
	*out = c;
	return RFWHIP_OK;
}

extern "C" void rfwhip_comm_destroy(rfwhip_comm *c)
{
	if (!c)
		return;
	destroy_group(&c->g);
	delete c;
}

extern "C" int rfwhip_comm_gather(rfwhip_comm *c, void *rgba_device)
{
	if (!c)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null comm");
	rfwhip_group *g = &c->g;
	uint32_t W = 0, H = 0;
	GR_TRY(rfwhip_get_target_size(g->ep[0].ctx, &W, &H));
	if (W != g->W || H != g->H || rfwhip_local_rows(g->ep[0].ctx) != g->local_rows)
	{
		GR_TRY(wait_all(g));
		GR_TRY(size_buffers(g, W, H));
	}
	return gather(g, rgba_device);
}

extern "C" int rfwhip_comm_wait(rfwhip_comm *c)
{
	if (!c)
		return rfwhip_internal_set_error(RFWHIP_ERR_INVALID_ARGUMENT, "null comm");
	return wait_all(&c->g);
}
