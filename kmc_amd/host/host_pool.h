/*
 * kmc_amd/host/host_pool.h — pinned host buffers shared by the plug-ins and the engine loader (no kmc_core types: the loader is compiled without the
 * reference's headers).
 */
#ifndef KMC_AMD_HOST_POOL_H
#define KMC_AMD_HOST_POOL_H

#include <cstddef>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

/* Pinned host buffers for bin images, shared by the reader plug-in (reads a bin file straight into one) and the worker plug-in (hands it to the engine, gives
 * it back): the image then reaches the GPU by DMA from where the reader put it — no page of the arena is touched for it, and the library has nothing to stage
 * (1.7 of the 2.3 GB a 2 Gbp run moves; summed over the workers the staging copies were 1 s of CPU). The engine's loader supplies the allocator
 * (kmc_hip_host_alloc / _free); without one — the oracle engines of the tests — get() returns NULL and the image goes to the arena as in the reference. Buffers
 * are kept and reused (first fit); the pool stops growing at $KMC_HIP_PINNED_POOL_MB (default 1024) and then says NULL as well. */
struct KmcHostPool {
	std::mutex m;
	void *(*alloc_fn)(size_t) = nullptr;
	void (*free_fn)(void *) = nullptr;
	std::vector<std::pair<void *, size_t>> free_list; /* (buffer, capacity) */
	std::map<void *, size_t> owned;                    /* every buffer of the pool -> capacity */
	size_t total = 0, limit = 0;
	static KmcHostPool &inst()
	{
		static KmcHostPool p;
		return p;
	}
	void *get(size_t bytes)
	{
		std::lock_guard<std::mutex> lck(m);
		if (!alloc_fn || !bytes)
			return nullptr;
		if (!limit) {
			const char *e = getenv("KMC_HIP_PINNED_POOL_MB");
			limit = (size_t)(e ? strtoull(e, nullptr, 10) : 1024) << 20;
			if (!limit)
				limit = 1; /* "0": the pool is off */
		}
		size_t best = free_list.size();
		for (size_t i = 0; i < free_list.size(); ++i)
			if (free_list[i].second >= bytes && (best == free_list.size() || free_list[i].second < free_list[best].second))
				best = i;
		if (best < free_list.size()) {
			void *p = free_list[best].first;
			free_list.erase(free_list.begin() + (ptrdiff_t)best);
			return p;
		}
		const size_t cap = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
		if (total + cap > limit)
			return nullptr;
		void *p = alloc_fn(cap);
		if (!p)
			return nullptr;
		owned[p] = cap;
		total += cap;
		return p;
	}
	/* true if `p` was one of the pool's (and is now free again) */
	bool put(void *p)
	{
		std::lock_guard<std::mutex> lck(m);
		auto it = owned.find(p);
		if (it == owned.end())
			return false;
		free_list.emplace_back(p, it->second);
		return true;
	}
	~KmcHostPool()
	{
		if (free_fn)
			for (auto &e : owned)
				free_fn(e.first);
	}
};

#endif
