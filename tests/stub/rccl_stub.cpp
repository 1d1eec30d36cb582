// rccl_stub.cpp — TEST INFRASTRUCTURE: a stand-in for librccl.so that lets the RCCL branch of csrc/rfwhip_group.cpp run on ONE
// device (RFWHIP_RCCL_LIBRARY=<this library> RFWHIP_RCCL_SHARED_DEVICE=1; tests/test_group.py).  It implements the eight entry
// points the product binds with the semantics the product relies on — point-to-point operations posted inside a group are
// matched by (source rank, destination rank) when the group ends; a receive completes on the receiver's stream after the
// sender's stream has reached its send; the sender's stream does not pass its send before the data has been taken — and
// writes one line per call (rank, peer, bytes, stream) to $RFWHIP_RCCL_STUB_LOG, so that a test can check the ENQUEUE ORDER the
// real library would see: one group per presented frame, a send per non-root rank on that rank's gather stream, world - 1
// receives on the root's.  Never shipped, never loaded by the product unless the two variables above say so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace
{
struct StubComm
{
	int rank, nranks;
};
struct Op
{
	bool send;
	void *buf;
	size_t bytes;
	int rank, peer;
	hipStream_t stream;
};
std::vector<Op> g_ops;
int g_depth = 0;
FILE *g_log = nullptr;
void logf(const char *fmt, ...)
{
	if (!g_log)
	{
		const char *p = getenv("RFWHIP_RCCL_STUB_LOG");
		g_log = p ? fopen(p, "a") : nullptr;
		if (!g_log)
			return;
	}
	va_list ap;
	va_start(ap, fmt);
	vfprintf(g_log, fmt, ap);
	va_end(ap);
	fflush(g_log);
}
size_t type_bytes(ncclDataType_t t) { return t == ncclFloat ? 4 : (t == ncclUint8 || t == ncclInt8 ? 1 : 4); }
ncclResult_t flush()
{
	// match every receive with the send of its peer
	for (const Op &r : g_ops)
	{
		if (r.send)
			continue;
		const Op *s = nullptr;
		for (const Op &c : g_ops)
			if (c.send && c.rank == r.peer && c.peer == r.rank)
				s = &c;
		if (!s || s->bytes != r.bytes)
		{
			logf("error unmatched recv rank %d peer %d bytes %zu\n", r.rank, r.peer, r.bytes);
			g_ops.clear();
			return ncclInvalidUsage;
		}
		hipEvent_t sent, taken;
		if (hipEventCreateWithFlags(&sent, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&taken, hipEventDisableTiming) != hipSuccess)
			return ncclUnhandledCudaError;
		(void)hipEventRecord(sent, s->stream);			 // the sender's stream has reached its send
		(void)hipStreamWaitEvent(r.stream, sent, 0);	 // ... before the receiver's stream copies
		(void)hipMemcpyAsync(r.buf, s->buf, r.bytes, hipMemcpyDeviceToDevice, r.stream);
		(void)hipEventRecord(taken, r.stream);
		(void)hipStreamWaitEvent(s->stream, taken, 0);	 // the send completes when the data has been taken
		(void)hipEventDestroy(sent), (void)hipEventDestroy(taken); // (destruction is deferred until the events complete)
	}
	for (const Op &c : g_ops)
		if (c.send)
		{
			bool matched = false;
			for (const Op &r : g_ops)
				matched = matched || (!r.send && r.rank == c.peer && r.peer == c.rank);
			if (!matched)
			{
				logf("error unmatched send rank %d peer %d\n", c.rank, c.peer);
				g_ops.clear();
				return ncclInvalidUsage;
			}
		}
	g_ops.clear();
	return ncclSuccess;
}
} // namespace

extern "C"
{
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	memset(id, 0, sizeof(*id));
	memcpy(id, "rfwhip-stub", 11);
	logf("unique_id\n");
	return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank)
{
	StubComm *c = new StubComm{rank, nranks};
	*comm = (ncclComm_t)c;
	logf("comm_init rank %d of %d in_group %d\n", rank, nranks, g_depth);
	return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	delete (StubComm *)comm;
	return ncclSuccess;
}
ncclResult_t ncclGroupStart()
{
	g_depth++;
	logf("group_start\n");
	return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
	logf("group_end ops %zu\n", g_ops.size());
	if (--g_depth > 0)
		return ncclSuccess;
	return flush();
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	const StubComm *c = (const StubComm *)comm;
	logf("send rank %d peer %d bytes %zu stream %p\n", c->rank, peer, count * type_bytes(type), (void *)stream);
	g_ops.push_back(Op{true, (void *)buf, count * type_bytes(type), c->rank, peer, stream});
	return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	const StubComm *c = (const StubComm *)comm;
	logf("recv rank %d peer %d bytes %zu stream %p\n", c->rank, peer, count * type_bytes(type), (void *)stream);
	g_ops.push_back(Op{false, buf, count * type_bytes(type), c->rank, peer, stream});
	return g_depth ? ncclSuccess : flush();
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "rccl stub error"; }
}
