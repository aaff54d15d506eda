/* tests/hipemu/include/rccl/rccl.h — TEST INFRASTRUCTURE: the RCCL calls of kmc_hip_allreduce_stats and kmc_hip_process_bin_multi over host memory
 * ("devices" are buffers of one process): an all-reduce issued per device between ncclGroupStart/End is summed when the group ends (uint64 sum only);
 * ncclSend / ncclRecv of a group are matched per (sender, receiver) pair in issue order and copied when the group ends — an unmatched or
 * size-mismatched pair fails the group, as a real one would hang. */
#ifndef KMC_TESTS_HIPEMU_RCCL_H
#define KMC_TESTS_HIPEMU_RCCL_H
#include <cstdint>
#include <cstring>
#include <vector>
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
typedef struct hipemuComm *ncclComm_t;
enum ncclDataType_t { ncclUint8 = 1, ncclUint64 = 5 };
enum ncclRedOp_t { ncclSum = 0 };
namespace hipemu {
struct PendingReduce {
	const void *send;
	void *recv;
	size_t count;
};
inline std::vector<PendingReduce> g_pending;
struct PendingP2P {
	int from, to; /* ranks */
	const void *src;
	void *dst;
	size_t bytes;
	bool done;
};
inline std::vector<PendingP2P> g_sends, g_recvs;
inline bool flush_p2p()
{
	bool ok = true;
	for (auto &r : g_recvs) {
		bool found = false;
		for (auto &s : g_sends)
			if (!s.done && s.from == r.from && s.to == r.to) {
				ok = ok && s.bytes == r.bytes;
				if (s.bytes == r.bytes)
					memcpy(r.dst, s.src, s.bytes);
				s.done = found = true;
				break;
			}
		ok = ok && found;
	}
	for (auto &s : g_sends)
		ok = ok && s.done;
	g_sends.clear();
	g_recvs.clear();
	return ok;
}
inline int g_group_depth = 0;
inline void flush_reduces()
{
	if (g_pending.empty())
		return;
	std::vector<uint64_t> sum(g_pending[0].count, 0);
	for (auto &p : g_pending)
		for (size_t i = 0; i < p.count; ++i)
			sum[i] += static_cast<const uint64_t *>(p.send)[i];
	for (auto &p : g_pending)
		for (size_t i = 0; i < p.count; ++i)
			static_cast<uint64_t *>(p.recv)[i] = sum[i];
	g_pending.clear();
}
} // namespace hipemu
static inline const char *ncclGetErrorString(ncclResult_t) { return "error (emulated)"; }
static inline ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *)
{
	for (int i = 0; i < n; ++i)
		comms[i] = reinterpret_cast<ncclComm_t>(new int(i)); /* the communicator IS its rank */
	return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c)
{
	delete reinterpret_cast<int *>(c);
	return ncclSuccess;
}
static inline ncclResult_t ncclGroupStart()
{
	++hipemu::g_group_depth;
	return ncclSuccess;
}
static inline ncclResult_t ncclGroupEnd()
{
	if (--hipemu::g_group_depth == 0) {
		hipemu::flush_reduces();
		if (!hipemu::flush_p2p())
			return 1;
	}
	return ncclSuccess;
}
template <typename S> static inline ncclResult_t ncclSend(const void *src, size_t count, ncclDataType_t t, int peer, ncclComm_t c, S)
{
	if (t != ncclUint8 || hipemu::g_group_depth == 0)
		return 1; /* a send outside a group would block on a real communicator of one process */
	hipemu::g_sends.push_back({*reinterpret_cast<int *>(c), peer, src, nullptr, count, false});
	return ncclSuccess;
}
template <typename S> static inline ncclResult_t ncclRecv(void *dst, size_t count, ncclDataType_t t, int peer, ncclComm_t c, S)
{
	if (t != ncclUint8 || hipemu::g_group_depth == 0)
		return 1;
	hipemu::g_recvs.push_back({peer, *reinterpret_cast<int *>(c), nullptr, dst, count, false});
	return ncclSuccess;
}
template <typename S> static inline ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t, S)
{
	hipemu::g_pending.push_back({send, recv, count});
	if (hipemu::g_group_depth == 0)
		hipemu::flush_reduces();
	return ncclSuccess;
}
#endif
