/*
 * kmc_amd/host/host_pool.h — pinned host buffers shared by the plug-ins and the engine loader (no kmc_core types: the loader is compiled without the
 * reference's headers).
 */
#ifndef KMC_AMD_HOST_POOL_H
#define KMC_AMD_HOST_POOL_H

#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <iterator>

/* Pinned host buffers for bin images, shared by the reader plug-in (reads a bin file straight into one) and the worker plug-in (hands it to the engine, gives
 * it back): the image then reaches the GPU by DMA from where the reader put it — no page of the arena is touched for it, and the library has nothing to stage
 * (1.7 of the 2.3 GB a 2 Gbp run moves; summed over the workers the staging copies were 1 s of CPU). ONE slab of pinned memory, allocated by the engine's loader
 * when the library comes up — on its background thread, during KMC's stage 1 ($KMC_HIP_PINNED_POOL_MB, default 2048 since round 6; allocating pinned buffers one by one while
 * stage 2 runs serialised the readers behind the runtime: reader wall 0.17 -> 0.42 s) — and cut first-fit. No slab (the oracle engines of the tests), or no room:
 * get() returns NULL and the image goes to the arena as in the reference. */
struct KmcHostPool {
	std::mutex m;
	std::condition_variable freed; /* put() */
	char *slab = nullptr;
	size_t slab_bytes = 0;
	std::map<size_t, size_t> free_ranges; /* offset -> length, coalesced */
	std::map<size_t, size_t> taken;       /* offset -> length */
	void (*free_fn)(void *) = nullptr;
	static KmcHostPool &inst()
	{
		static KmcHostPool p;
		return p;
	}
	static size_t wanted_bytes()
	{
		const char *e = getenv("KMC_HIP_PINNED_POOL_MB");
		return (size_t)(e ? strtoull(e, nullptr, 10) : 2048) << 20; /* (16 workers x up to 4 waiting bins + the readers' look-ahead want ~1 GB of 13 MB bins at 8 Gbp, 1.2+ GB of 50 MB bins at 30 Gbp: with 1 GB some images go through the arena and a staging copy; a larger hipHostMalloc slab costs 0.25 s per GB when the process exits) */
	}
	void adopt(void *p, size_t bytes, void (*release)(void *))
	{
		std::lock_guard<std::mutex> lck(m);
		slab = (char *)p;
		slab_bytes = bytes;
		free_fn = release;
		free_ranges.clear();
		taken.clear();
		free_ranges[0] = bytes;
	}
	void *get(size_t bytes)
	{
		std::lock_guard<std::mutex> lck(m);
		return get_locked(bytes);
	}
	void *get_locked(size_t bytes) /* caller holds m */
	{
		if (!slab || !bytes)
			return nullptr;
		const size_t need = (bytes + 4095) & ~(size_t)4095;
		for (auto it = free_ranges.begin(); it != free_ranges.end(); ++it) {
			if (it->second < need)
				continue;
			const size_t off = it->first, len = it->second;
			free_ranges.erase(it);
			if (len > need)
				free_ranges[off + need] = len - need;
			taken[off] = need;
			return slab + off;
		}
		return nullptr;
	}
	/* get(), but a reader that finds the pool exhausted WAITS for a worker to give a buffer back (up to `ms` milliseconds; round 6): at 30 Gbp the readers ran 170 bins
	 * ahead of the workers, the pool was gone after the first 20, and every image behind them went through the arena — a 50 MB staging copy by the worker plus the
	 * pinned staging buffer's own allocation, 16.7 of the workers' 56.7 engine-seconds (profiles/r06/e2e_sweep_30gbp_session_w.jsonl). Workers never wait for the
	 * reader while they hold buffers, so a waiting reader always gets one; the timeout is the fallback to the arena, as before. */
	void *get_wait(size_t bytes, int ms)
	{
		std::unique_lock<std::mutex> lck(m);
		if (void *p = get_locked(bytes))
			return p;
		/* only for images of which the slab holds 16 at least: with fewer, the other reader threads — which hold theirs while they wait for their turn to push behind
		 * this one — could own all of it, and nobody who could give a buffer back would have one */
		if (!slab || !bytes || ((bytes + 4095) & ~(size_t)4095) > slab_bytes / 16 || ms <= 0)
			return nullptr;
		const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
		while (freed.wait_until(lck, deadline) != std::cv_status::timeout)
			if (void *p = get_locked(bytes))
				return p;
		return get_locked(bytes);
	}
	/* true if `p` was one of the pool's (and is now free again) */
	bool put(void *p)
	{
		struct Wake {
			KmcHostPool *h;
			~Wake() { h->freed.notify_all(); }
		} wake{this};
		std::lock_guard<std::mutex> lck(m);
		if (!slab || (char *)p < slab || (char *)p >= slab + slab_bytes)
			return false;
		size_t off = (size_t)((char *)p - slab);
		auto it = taken.find(off);
		if (it == taken.end())
			return false;
		size_t len = it->second;
		taken.erase(it);
		auto nx = free_ranges.lower_bound(off);
		if (nx != free_ranges.end() && nx->first == off + len) {
			len += nx->second;
			nx = free_ranges.erase(nx);
		}
		if (nx != free_ranges.begin()) {
			auto pv = std::prev(nx);
			if (pv->first + pv->second == off) {
				off = pv->first;
				len += pv->second;
				free_ranges.erase(pv);
			}
		}
		free_ranges[off] = len;
		return true;
	}
	/* No destructor work: this object dies during static destruction, and hipHostFree through a dlopen'd library at that point runs after (or during) the HIP
	 * runtime's own atexit teardown unless the pool happened to be constructed after the library was loaded. The slab lives as long as the process;
	 * the kernel returns pinned memory at exit. */
	~KmcHostPool() {}
};

#endif
