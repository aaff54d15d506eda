/*
 * kmc_amd/host/kmc_order.h — state shared by the two stage-2 plug-ins (kb_reader_plugin.h, kb_sorter_plugin.h) of one run:
 * the emission order of bins and, for KMC_HIP_VERBOSE=1, where the stage-2 threads spent their time.
 *
 * Ordered hand-off to the completer. The reference's database bytes depend on the order bins reach kq
 * (kb_completer.cpp:131-221): with several CPU sorters that order is a race, which is why reference
 * runs are only reproducible with -sr1. Here every bin gets a sequence number when GetNext hands it to a worker (taken
 * under a mutex; bins reach the bin queue in CBinDesc's sorted order, queues.h:499-571, from the reference's reader and
 * from the reader plug-in alike) and is pushed to kq strictly in that sequence, however many workers / GPUs / stream
 * slots finish out of order: the DB equals the reference's -sr1 bytes for ANY -sr.
 * Since round 6 a worker does not WAIT for its turn: it deposits the finished bin in a reorder buffer and goes for the next one; whoever deposits the bin the
 * completer is waiting for pushes it and every consecutive bin behind it (16 workers spent a third of their wall time waiting for their turn on 8 Gbp:
 * 3.3 of 9.6 s summed, profiles/r06/e2e_sweep_8gbp_session_n.jsonl).
 */
#ifndef KMC_AMD_ORDER_H
#define KMC_AMD_ORDER_H

#include <atomic>
#include <chrono>
#include <list>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <utility>
#include <mutex>
#include <vector>

#include <sys/mman.h>
#include <thread>
#include <algorithm>
#include "host_pool.h"
#include "defs.h"
#include "params.h"
#include "critical_error_handler.h"

/* KMC_HIP_VERBOSE=1: wall-clock marks of one run (reader / workers / completer), printed at exit as seconds since the first mark —
 * where between its start and its end does "2nd stage" spend its time? */
struct KmcTimeline {
	std::mutex m;
	std::vector<std::pair<const char *, long long>> marks;
	bool on = getenv("KMC_HIP_VERBOSE") != nullptr;
	static KmcTimeline &inst()
	{
		static KmcTimeline t;
		return t;
	}
	static void mark(const char *what)
	{
		KmcTimeline &t = inst();
		if (!t.on)
			return;
		const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
		std::lock_guard<std::mutex> lck(t.m);
		t.marks.emplace_back(what, now);
	}
	/* first / last of many events of one kind (the 100+ splitter workers of stage 1) */
	static void mark_first_last(const char *first, const char *last)
	{
		KmcTimeline &t = inst();
		if (!t.on)
			return;
		const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
		std::lock_guard<std::mutex> lck(t.m);
		bool have_first = false, have_last = false;
		for (auto &e : t.marks) {
			have_first = have_first || e.first == first;
			if (e.first == last) {
				e.second = now;
				have_last = true;
			}
		}
		if (first && !have_first)
			t.marks.emplace_back(first, now);
		if (last && !have_last)
			t.marks.emplace_back(last, now);
	}
	~KmcTimeline()
	{
		if (!on || marks.empty())
			return;
		mark("process exit (static destructors)");
		fprintf(stderr, "[kmc_hip timeline] (first mark at %.3f s of the steady clock)", marks[0].second * 1e-9);
		for (auto &e : marks)
			fprintf(stderr, " %s %.3f |", e.first, (e.second - marks[0].second) * 1e-9);
		fprintf(stderr, "\n");
	}
};

/* The arena of CMemoryBins as the reader plug-in found it (one anonymous mapping; kb_reader_plugin.h advise_arena_once). The reference releases it at the end of
 * stage 2 on a thread it joins before the "2nd stage" timer stops (kmc.h:1602-1605): on the GPU boxes of this pool unmapping the 2.3 GB a 2 Gbp run has touched
 * takes 0.15-0.19 s — 40 % of the stage once the sort runs on the GPU (main-thread samples, profiles/r04/e2e_teardown.txt) — with 4 KB pages and with huge pages
 * alike. The pages can go earlier and in parallel: when the completer plug-in has written the last bin, nothing in the arena is needed any more, and 16 threads
 * of madvise(MADV_DONTNEED) take 0.02 s (tools/ubench_munmap.c); the reference's own release then finds nothing to tear down. */
struct KmcArena {
	std::atomic<uintptr_t> lo{0}, hi{0};     /* the arena's allocator block [lo, hi), page-granular; 0 = not found: nothing is advised or zapped */
	std::atomic<uintptr_t> last_inside{0};   /* a pointer CMemoryBins handed out most recently: the buffer may have been re-allocated since the first one (queues.h:1196-1207) */
	std::atomic<uint64_t> arena_bytes{0};
	static KmcArena &inst()
	{
		static KmcArena a;
		return a;
	}
	/* The block malloc() gave CMemoryBins::prepare (queues.h:1130: total_size + ALIGNMENT bytes — far beyond the mmap threshold, so glibc serves it with a
	 * mapping of its own and a 16-byte chunk header at the mapping's first bytes: [prev_size = 0][size | IS_MMAPPED]). The kernel MERGES adjacent anonymous
	 * mappings into one VMA, so the VMA around `inside` is usually NOT the arena alone (a neighbour's pages zapped = silent corruption); the range is
	 * therefore taken from the allocator's own header: the VMA is walked chunk by chunk from its start, every step checked (mmapped flag, whole pages, inside the
	 * VMA), and the chunk that holds `inside` must have exactly the size CMemoryBins asked for, page-rounded. Anything else: not found, feature off. */
	static bool find_block(const void *inside, uint64_t bytes, uintptr_t &blo, uintptr_t &bhi)
	{
#ifndef __GLIBC__
		(void)inside, (void)bytes, (void)blo, (void)bhi;
		return false; /* the chunk-header layout below is glibc's */
#endif
		FILE *f = fopen("/proc/self/maps", "r");
		if (!f)
			return false;
		/* private read-write mappings, adjacent ones joined: madvise(MADV_HUGEPAGE) on a part of the block SPLITS its VMA in three, and neighbours of the same
		 * kind are merged into it — either way the allocator's headers decide, not the VMA bounds. Every start of a VMA at or below `inside` inside the joined
		 * region is tried as the start of a chain of chunks. */
		std::vector<std::pair<uintptr_t, uintptr_t>> region;
		char line[512], perms[8];
		const uintptr_t p = (uintptr_t)inside;
		bool past = false;
		while (fgets(line, sizeof line, f) && !past) {
			unsigned long long a = 0, b = 0, off = 0, ino = 0;
			char devno[16], path[8];
			path[0] = 0;
			if (sscanf(line, "%llx-%llx %7s %llx %15s %llu %7s", &a, &b, perms, &off, devno, &ino, path) < 6)
				continue;
			if (ino != 0 || path[0] == '/') /* a file-backed mapping is never the allocator's (and a read past its end would fault: ADVICE r5) */
				continue;
			const bool rw = perms[0] == 'r' && perms[1] == 'w' && perms[3] == 'p';
			if (!rw || (!region.empty() && region.back().second != (uintptr_t)a)) {
				if (!region.empty() && p >= region.front().first && p < region.back().second)
					past = true; /* the region that holds `inside` is complete */
				else
					region.clear();
				if (!rw || past)
					continue;
			}
			region.emplace_back((uintptr_t)a, (uintptr_t)b);
		}
		fclose(f);
		if (region.empty() || p < region.front().first || p >= region.back().second)
			return false;
		const uintptr_t end = region.back().second;
		const uint64_t page = 4096, want_lo = bytes + 256 + 8, want_hi = want_lo + 2 * page; /* request = total_size + ALIGNMENT, + the header, page-rounded */
		for (size_t i = region.size(); i-- > 0;) {
			if (region[i].first > p)
				continue;
			for (uintptr_t c = region[i].first; c + 16 <= end;) {
				const size_t prev = ((const size_t *)c)[0], sz = ((const size_t *)c)[1];
				const size_t len = sz & ~(size_t)7;
				if (prev != 0 || (sz & 7) != 2 || len < page || (len & (page - 1)) || len > end - c)
					break; /* not a chain of glibc's mmapped chunks from here: try the VMA start before it */
				if (p >= c + 16 && p < c + len) {
					if (len >= want_lo && len <= want_hi) {
						blo = c;
						bhi = c + len;
						return true;
					}
					break;
				}
				c += len;
			}
		}
		return false;
	}
	void note(const void *inside, uint64_t bytes)
	{
		last_inside.store((uintptr_t)inside);
		arena_bytes.store(bytes);
	}
	void zap_parallel(int n_threads = 16)
	{
		const char *e = getenv("KMC_HIP_ARENA_ZAP");
		const uintptr_t in = last_inside.load();
		if (!in || (e && atoi(e) == 0))
			return;
		uintptr_t blo = 0, bhi = 0; /* looked up NOW, from the latest pointer: CMemoryBins may have re-allocated its buffer since the reader advised the first one */
		const bool found = arena_bytes.load() >= (64ull << 20) && find_block((const void *)in, arena_bytes.load(), blo, bhi);
		if (getenv("KMC_HIP_VERBOSE"))
			fprintf(stderr, "[kmc_hip stage 2] arena zap: %s\n", found ? "block found, pages returned in parallel" : "the arena's block was not identified: nothing zapped");
		last_inside.store(0);
		if (!found)
			return;
		const uintptr_t a = blo + 4096, b = bhi; /* the first page holds the allocator's header of the block (free() reads it at the release) */
		const uintptr_t two_mb = (uintptr_t)2 << 20;
		const uintptr_t per = (((b - a) / (uintptr_t)n_threads) + two_mb - 1) & ~(two_mb - 1);
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t) {
			const uintptr_t s = a + (uintptr_t)t * per, f = std::min(b, s + per);
			if (s < f)
				th.emplace_back([s, f] { (void)madvise((void *)s, f - s, MADV_DONTNEED); });
		}
		for (auto &x : th)
			x.join();
	}
};

struct KmcOrderedEmit {
	std::mutex take_mtx, emit_mtx;
	uint64 next_take = 0, next_emit = 0;
	/* a finished bin, as kq->push wants it (queues.h:826) */
	struct Ready {
		int32 bin_id;
		uchar *out;
		std::list<std::pair<uint64, uint64>> packs;
		uchar *lut;
		uint64 lut_bytes, n_unique, n_cutoff_min, n_cutoff_max, n_total;
	};
	std::map<uint64, Ready> ready; /* by sequence number; under emit_mtx */
	/* hand a finished bin over; pushes it, and whatever is consecutive behind it, if it is the one the completer waits for */
	template <typename KQ> void deposit(KQ *kq, uint64 seq, Ready &&r)
	{
		const long long t0 = now_ns();
		std::lock_guard<std::mutex> lck(emit_mtx);
		ready.emplace(seq, std::move(r));
		while (!ready.empty() && ready.begin()->first == next_emit) {
			Ready &a = ready.begin()->second;
			kq->push(a.bin_id, a.out, a.packs, a.lut, a.lut_bytes, a.n_unique, a.n_cutoff_min, a.n_cutoff_max, a.n_total);
			ready.erase(ready.begin());
			++next_emit;
		}
		ns_push += now_ns() - t0;
	}
	/* KMC_HIP_VERBOSE=1: nanoseconds summed over threads */
	std::atomic<long long> ns_reader_init{0}, ns_reader_read{0}, ns_reader_wall{0}, ns_getnext{0}, ns_engine{0}, ns_turn{0}, ns_push{0},
	    ns_worker_wall{0};
	std::atomic<long long> n_bins{0}, n_workers_done{0}, n_readers{0}, n_group_calls{0};
	static long long now_ns()
	{
		return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
	}
	void report(int n_workers)
	{
		if (!getenv("KMC_HIP_VERBOSE"))
			return;
		fprintf(stderr,
		        "[kmc_hip stage 2] %lld bins, %d workers, %lld reader threads | reader: admit %.3f s, read %.3f s (summed), wall %.3f s | workers "
		        "(summed over threads): wait for a bin %.3f s, engine %.3f s, wait for the turn to push %.3f s, push %.3f s, wall %.3f s | %lld engine calls "
		        "with several bins\n",
		        n_bins.load(), n_workers, n_readers.load(), ns_reader_init.load() * 1e-9, ns_reader_read.load() * 1e-9, ns_reader_wall.load() * 1e-9,
		        ns_getnext.load() * 1e-9, ns_engine.load() * 1e-9, ns_turn.load() * 1e-9, ns_push.load() * 1e-9, ns_worker_wall.load() * 1e-9, n_group_calls.load());
	}

	static std::shared_ptr<KmcOrderedEmit> for_queue(CKmerQueue *kq)
	{
		static std::mutex m;
		static std::map<CKmerQueue *, std::weak_ptr<KmcOrderedEmit>> live;
		std::lock_guard<std::mutex> lck(m);
		auto sp = live[kq].lock();
		if (!sp) {
			sp = std::make_shared<KmcOrderedEmit>();
			live[kq] = sp;
		}
		return sp;
	}
};

#endif
