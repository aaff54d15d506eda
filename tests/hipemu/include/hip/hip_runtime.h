/*
 * tests/hipemu/include/hip/hip_runtime.h — TEST INFRASTRUCTURE, not product code.
 *
 * A host-side emulation of the small subset of the HIP device language that kmc_amd/csrc/kernels.hip.h uses, so that the
 * kernel SOURCE (unchanged) can be executed on the CPU by the `-m "not gpu"` tests (tests/test_kernels_emulated.py) and
 * checked against the oracle where no GPU exists. It is put in front of the real <hip/hip_runtime.h> by the include path of
 * tests/hipemu/Makefile only; nothing under kmc_amd/ or bench.py links or loads it, and libkmc_hip.so is never built from it.
 *
 * Model: one OS thread per GPU thread of a workgroup; workgroups run one after the other in blockIdx order (so a decoupled
 * look-back always finds its predecessors finished); __shared__ is `static` (one workgroup at a time); __syncthreads is a
 * barrier over those threads; wave64 cross-lane operations (ballot, shuffles, readfirstlane) exchange values through a per-wave slot
 * array bracketed by per-wave barriers, which requires what the hardware code needs anyway: every lane of a wave executes
 * the same sequence of cross-lane operations. Atomics map to the GCC __atomic builtins (x86 is stronger than the GPU's
 * memory model: ordering bugs do not show here, missing barriers and indexing bugs do — run under -fsanitize=thread to
 * see races on "LDS").
 */
#ifndef KMC_TESTS_HIPEMU_H
#define KMC_TESTS_HIPEMU_H

#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <type_traits>
#include <vector>

#define KMC_HIPEMU 1

struct dim3 {
	unsigned x, y, z;
	dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint4 {
	unsigned x, y, z, w;
};
struct ulonglong2 {
	unsigned long long x, y;
};
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline ulonglong2 make_ulonglong2(unsigned long long a, unsigned long long b) { return ulonglong2{a, b}; }

namespace hipemu {
/* sense-reversing barrier that yields while it waits: a workgroup is hundreds of OS threads on a handful of cores, and a futex
 * round trip per cross-lane operation (pthread_barrier) is several times slower than handing the core on */
struct Barrier {
	unsigned count = 0;
	unsigned waiting = 0, generation = 0;
	void init(unsigned n)
	{
		count = n;
		waiting = 0;
		generation = 0;
	}
	void wait()
	{
		const unsigned gen = __atomic_load_n(&generation, __ATOMIC_ACQUIRE);
		if (__atomic_add_fetch(&waiting, 1, __ATOMIC_ACQ_REL) == count) {
			__atomic_store_n(&waiting, 0, __ATOMIC_RELAXED);
			__atomic_add_fetch(&generation, 1, __ATOMIC_RELEASE);
		} else {
			while (__atomic_load_n(&generation, __ATOMIC_ACQUIRE) == gen)
				sched_yield();
		}
	}
};
struct Wave {
	Barrier bar;
	unsigned long long slot[64];
};
struct Launch {
	dim3 grid, block;
	Barrier block_bar;
	std::vector<Wave> waves;
	unsigned char *dyn = nullptr;
};
inline Launch *g_launch = nullptr;
inline thread_local dim3 t_threadIdx, t_blockIdx;
inline thread_local Wave *t_wave = nullptr;
inline unsigned lane_id() { return t_threadIdx.x & 63u; }

inline void wave_exchange(unsigned long long v, unsigned long long (&all)[64])
{
	Wave &w = *t_wave;
	w.slot[lane_id()] = v;
	w.bar.wait();
	memcpy(all, w.slot, sizeof all);
	w.bar.wait();
}
template <typename T> inline unsigned long long to_bits(T v)
{
	static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits");
	unsigned long long b = 0;
	memcpy(&b, &v, sizeof(T));
	return b;
}
template <typename T> inline T from_bits(unsigned long long b)
{
	T v;
	memcpy(&v, &b, sizeof(T));
	return v;
}

/* run `body()` once per GPU thread of every workgroup; `dyn_bytes` of dynamic LDS */
template <class F> void launch(dim3 grid, dim3 block, size_t dyn_bytes, F body)
{
	if (block.x % 64 != 0 || block.y != 1 || block.z != 1 || grid.z != 1)
		abort();
	Launch L;
	L.grid = grid;
	L.block = block;
	L.block_bar.init(block.x);
	L.waves = std::vector<Wave>(block.x / 64);
	for (auto &w : L.waves)
		w.bar.init(64);
	void *dyn = nullptr;
	if (posix_memalign(&dyn, 256, dyn_bytes + 256))
		abort();
	memset(dyn, 0xA5, dyn_bytes + 256); /* LDS is not zero-initialised on the GPU either */
	L.dyn = static_cast<unsigned char *>(dyn);
	g_launch = &L;
	std::vector<std::thread> th;
	th.reserve(block.x);
	for (unsigned t = 0; t < block.x; ++t)
		th.emplace_back([&, t] {
			t_threadIdx = dim3(t);
			t_wave = &L.waves[t / 64];
			for (unsigned by = 0; by < grid.y; ++by)
				for (unsigned b = 0; b < grid.x; ++b) {
					t_blockIdx = dim3(b, by);
					body();
					L.block_bar.wait(); /* static "LDS" is reused by the next workgroup */
				}
		});
	for (auto &x : th)
		x.join();
	g_launch = nullptr;
	/* the 256 bytes behind what the launch asked for are a guard (on the GPU a store past the dynamic LDS is dropped and a load returns 0 — silently): a kernel
	 * that wrote there laid out more LDS than its launch requests */
	for (size_t i = 0; i < 256; ++i)
		if (static_cast<unsigned char *>(dyn)[dyn_bytes + i] != 0xA5) {
			fprintf(stderr, "hipemu: a kernel stored %zu bytes past its %zu bytes of dynamic LDS\n", i + 1, dyn_bytes);
			abort();
		}
	free(dyn);
}
} // namespace hipemu

#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::g_launch->block)
#define gridDim (hipemu::g_launch->grid)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
/* the four constructs of kernels.hip.h that have no host spelling (see the defaults there) */
#define KMC_DYN_LDS(type, name) type *name = reinterpret_cast<type *>(hipemu::g_launch->dyn)
#define KMC_LAUNDER(x) ((void)(x))
#define KMC_WAIT_VMEM() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define KMC_WAVE_LOCKSTEP() hipemu::t_wave->bar.wait()

static inline void __syncthreads() { hipemu::g_launch->block_bar.wait(); }

/* ---- wave64 cross-lane operations */
static inline unsigned long long __ballot(int pred)
{
	unsigned long long all[64], m = 0;
	hipemu::wave_exchange(pred ? 1ull : 0ull, all);
	for (int l = 0; l < 64; ++l)
		m |= (all[l] & 1ull) << l;
	return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(pred) == ~0ull; }
template <typename T> static inline T __shfl_up(T v, unsigned o)
{
	unsigned long long all[64];
	hipemu::wave_exchange(hipemu::to_bits(v), all);
	const unsigned l = hipemu::lane_id();
	return l >= o ? hipemu::from_bits<T>(all[l - o]) : v;
}
template <typename T> static inline T __shfl_down(T v, unsigned o)
{
	unsigned long long all[64];
	hipemu::wave_exchange(hipemu::to_bits(v), all);
	const unsigned l = hipemu::lane_id();
	return l + o < 64 ? hipemu::from_bits<T>(all[l + o]) : v;
}
template <typename T> static inline T __shfl(T v, int src)
{
	unsigned long long all[64];
	hipemu::wave_exchange(hipemu::to_bits(v), all);
	return hipemu::from_bits<T>(all[src & 63]);
}
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned s) { return (unsigned)((((unsigned long long)hi << 32) | lo) >> (s & 31)); } /* v_alignbit_b32 */
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0); } /* every lane is active where the kernels use it */
static inline unsigned long long __builtin_amdgcn_uicmp(unsigned a, unsigned b, int cond)
{
	if (cond != 33) /* ICMP_NE is the only condition the kernels use */
		abort();
	return __ballot(a != b);
}
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned acc)
{
	const unsigned l = hipemu::lane_id();
	const unsigned lt = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u);
	return acc + (unsigned)__builtin_popcount(mask & lt);
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned acc)
{
	const unsigned l = hipemu::lane_id();
	const unsigned lt = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
	return acc + (unsigned)__builtin_popcount(mask & lt);
}

/* ---- scalar helpers */
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __builtin_amdgcn_sbfe(int v, unsigned off, unsigned width)
{
	const unsigned x = ((unsigned)v >> off) & ((width >= 32 ? 0u : (1u << width)) - 1u);
	const unsigned sign = 1u << (width - 1);
	return (int)((x ^ sign) - sign);
}
static inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned width) { return (v >> off) & ((width >= 32 ? 0u : (1u << width)) - 1u); }
static inline unsigned __builtin_amdgcn_bitop3_b32(unsigned a, unsigned b, unsigned c, unsigned tt)
{
	unsigned r = 0;
	for (int i = 0; i < 32; ++i) {
		const unsigned idx = (((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u);
		r |= ((tt >> idx) & 1u) << i;
	}
	return r;
}
static inline void __builtin_amdgcn_s_sleep(int) { sched_yield(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline unsigned long long wall_clock64() { return 0; }
#if !defined(__clang__)
static inline unsigned long long __builtin_readcyclecounter() { return 0; }
#endif

/* ---- atomics (scopes are ignored: one coherent host memory) */
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v)
{
	__atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
	return cmp; /* the value found */
}
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T> static inline T atomicMax(T *p, T v)
{
	T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
	}
	return old;
}
template <typename T> static inline T atomicMin(T *p, T v)
{
	T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
	}
	return old;
}

#ifdef HIPEMU_HOST_API
#include "hip_host_api.h" /* the runtime API of the product's host library, over plain memory (tests/emu.py build_hostlib) */
#endif

#endif
