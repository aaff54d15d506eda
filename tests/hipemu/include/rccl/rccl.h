/* tests/hipemu/include/rccl/rccl.h — TEST INFRASTRUCTURE: the four RCCL calls of kmc_hip_allreduce_stats over host memory ("devices" are
 * buffers of one process): an all-reduce issued per device between ncclGroupStart/End is summed when the group ends. uint64 sum only. */
#ifndef KMC_TESTS_HIPEMU_RCCL_H
#define KMC_TESTS_HIPEMU_RCCL_H
#include <cstdint>
#include <vector>
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
typedef struct hipemuComm *ncclComm_t;
enum ncclDataType_t { ncclUint64 = 5 };
enum ncclRedOp_t { ncclSum = 0 };
namespace hipemu {
struct PendingReduce {
	const void *send;
	void *recv;
	size_t count;
};
inline std::vector<PendingReduce> g_pending;
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
		comms[i] = reinterpret_cast<ncclComm_t>(new char);
	return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c)
{
	delete reinterpret_cast<char *>(c);
	return ncclSuccess;
}
static inline ncclResult_t ncclGroupStart()
{
	++hipemu::g_group_depth;
	return ncclSuccess;
}
static inline ncclResult_t ncclGroupEnd()
{
	if (--hipemu::g_group_depth == 0)
		hipemu::flush_reduces();
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
