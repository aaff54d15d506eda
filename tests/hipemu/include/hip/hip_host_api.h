/*
 * tests/hipemu/include/hip/hip_host_api.h — TEST INFRASTRUCTURE: the part of the HIP RUNTIME API kmc_amd/csrc/kmc_hip.hip uses, as
 * synchronous host functions over plain memory ("device" pointers are host pointers, streams do nothing, events are time stamps), so that the
 * product's HOST library — buffer management, launch sequences, error handling, the whole C-ABI — can be compiled with g++ and run on a box
 * without a GPU (tests/emu.py build_hostlib: the source is taken as it is, only `kernel<<<grid, block, lds, stream>>>(args)` is rewritten to
 * HIPEMU_LAUNCH). Included by hip_runtime.h when HIPEMU_HOST_API is defined. $HIPEMU_DEVICES = number of "devices" (default 1).
 */
#ifndef KMC_TESTS_HIPEMU_HOST_API_H
#define KMC_TESTS_HIPEMU_HOST_API_H

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct hipemuStream *hipStream_t;
struct hipemuEvent {
	std::chrono::steady_clock::time_point t;
};
typedef hipemuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostRegisterPortable = 1, hipEventDisableTiming = 2, hipEventBlockingSync = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hipemu {
inline std::recursive_mutex g_launch_mtx; /* emulated "LDS" is static storage: one launch at a time, whatever host thread issues it */
}
#define HIPEMU_LAUNCH(kernel, grid, block, lds, stream, ...)                                                            \
	do {                                                                                                               \
		std::lock_guard<std::recursive_mutex> hipemu_lck(hipemu::g_launch_mtx);                                        \
		hipemu::launch(grid, block, (size_t)(lds), [&] { kernel(__VA_ARGS__); });                                     \
	} while (0)
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) HIPEMU_LAUNCH(kernel, grid, block, lds, stream, __VA_ARGS__)

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (emulated)" : "error (emulated)"); }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n)
{
	const char *e = getenv("HIPEMU_DEVICES");
	*n = e ? atoi(e) : 1;
	return hipSuccess;
}
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b)
{
	*total_b = (size_t)288 << 30;
	*free_b = (size_t)280 << 30;
	return hipSuccess;
}
static inline hipError_t hipGetDevice(int *d)
{
	*d = 0;
	return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t bytes)
{
	/* 0xCD: device memory is not zero-initialised */
	if (posix_memalign(p, 256, bytes + 256))
		return hipErrorOutOfMemory;
	memset(*p, 0xCD, bytes + 256);
	return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc((void **)p, bytes); }
static inline hipError_t hipFree(void *p)
{
	free(p);
	return hipSuccess;
}
static inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned = 0) { return hipMalloc(p, bytes); }
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
/* every caller buffer counts as ordinary memory: the emulated runs take the library's pinned-staging path */
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
	memmove(d, s, n);
	return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { return hipMemcpy(d, s, n, hipMemcpyDeviceToDevice); }
static inline hipError_t hipMemset(void *d, int v, size_t n)
{
	memset(d, v, n);
	return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
	*s = reinterpret_cast<hipStream_t>(new char);
	return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t s)
{
	delete reinterpret_cast<char *>(s);
	return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e)
{
	*e = new hipemuEvent;
	return hipSuccess;
}
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e)
{
	delete e;
	return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
	e->t = std::chrono::steady_clock::now();
	return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
	*ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
	return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

#endif
