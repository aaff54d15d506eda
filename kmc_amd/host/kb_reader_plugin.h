/*
 * kmc_amd/host/kb_reader_plugin.h — the stage-2 bin reader of kmc_core with the file reads done by several threads
 * ("reader decoupling").
 *
 * Drop-in for the reference's kmc_core/kb_reader.h: it defines the same class template
 *     template <unsigned SIZE> class CWKmerBinReader        (reference: kb_reader.h:233-260)
 * with the same constructor and operator()(), so CKMC<SIZE>::ProcessStage2_impl (kmc.h:1586-1587) builds against it
 * unchanged; compiled in like the sorter plug-in: -D_KB_READER_H -include kb_reader_plugin.h (oracle/Makefile).
 *
 * Why: with the sort on a GPU, stage 2 is bound by the reference's ONE reader thread, which copies every bin image
 * (CMemDiskFile::Read: fread or a RAM-to-RAM copy) and deletes its file, bin after bin (kb_reader.h:116-219).
 * Here $KMC_HIP_READERS threads (default 8; 1 = the reference's behaviour) do that work for different bins at once.
 *
 * What is kept, per bin, exactly as kb_reader.h:116-219: bd->get_next_sort_bin -> bd->read -> buffer sizes (:130-165)
 * -> memory_bins->init (too-large bins go to tlbq in strict-memory mode) -> reserve mba_input_file -> file->Read ->
 * memory_bins->extend -> bq->push + sorters_manager->NotifyBQPush (also for empty bins) -> file Close/Remove ->
 * disk_logger->log_remove -> progress; at the end bq->mark_completed + NotifyQueueCompleted.
 *
 * Order and progress guarantees:
 *   - bins are ADMITTED to the arena (get_next_sort_bin + memory_bins->init) one at a time, in CBinDesc's sorted order, and
 *     PUSHED to the bin queue in that same order (a turnstile after the read), so the workers receive bins exactly as from
 *     the reference's reader: the database bytes stay those of the reference's -sr1 run, and the worker holding the next
 *     bin to emit always exists (ordered emission, kmc_order.h, cannot starve);
 *   - only a bin whose whole region was reserved by init() is read outside that critical section. When the arena is tight
 *     init() reserves just the file bytes and extend() must find or make room later (queues.h:1288-1340, :1395-1537): such a bin
 *     is read, extended and pushed INSIDE the critical section, i.e. exactly like the single reference reader, so a later bin
 *     can never take the space an earlier one is waiting for (the reference's progress argument carries over).
 */
#ifndef KMC_AMD_KB_READER_PLUGIN_H
#define KMC_AMD_KB_READER_PLUGIN_H

#include "defs.h"
#include "params.h"
#include "kmer.h"
#include "s_mapper.h"
#include "radix.h"
#include "percent_progress.h"
#include "critical_error_handler.h"
#include "exception_aware_thread.h"
#include <string>
#include <algorithm>
#include <numeric>
#include <array>
#include <vector>
#include <sstream>
#include <stdio.h>
#include <mutex>
#include <sys/mman.h>

#include "kmc_order.h"

template <unsigned SIZE> class CWKmerBinReader {
	CBinDesc *bd;
	CBinQueue *bq;
	CSortersManager *sorters_manager;
	CTooLargeBinsQueue *tlbq;
	CMemoryBins *memory_bins;
	CDiskLogger *disk_logger;
	std::shared_ptr<KmcOrderedEmit> order;

	uint32 cutoff_min, cutoff_max, counter_max, kmer_len, max_x;
	int32 lut_prefix_len;
	KMC::IPercentProgressObserver *percentProgressObserver;
#ifdef DEVELOP_MODE
	bool verbose_log;
#endif
	int n_readers;
	std::mutex admit_mtx, progress_mtx, push_mtx;
	CThrowingOnCancelConditionVariable push_cv;
	uint64 next_seq = 0, push_turn = 0;

	static int64 round_up_to_alignment(int64 x) { return (x + ALIGNMENT - 1) / ALIGNMENT * ALIGNMENT; }

	struct BinPlan {
		int32 bin_id;
		CMemDiskFile *file;
		string name;
		uint64 size, n_rec, n_plus_x_recs;
		uint32 rec_len;
		uint64 seq;
		int64 a_size, a_kxmers, a_out, a_counters, a_lut; /* aligned sizes handed to init/extend */
	};

	/* buffer sizes of one bin: kb_reader.h:130-165 */
	void plan(BinPlan &b)
	{
		uint64 input_kmer_size, kxmer_counter_size;
		uint32 kxmer_symbols;
		if (max_x) {
			input_kmer_size = b.n_plus_x_recs * sizeof(CKmer<SIZE>);
			kxmer_counter_size = b.n_plus_x_recs * sizeof(uint32);
			kxmer_symbols = kmer_len + max_x + 1;
		} else {
			input_kmer_size = b.n_rec * sizeof(CKmer<SIZE>);
			kxmer_counter_size = 0;
			kxmer_symbols = kmer_len;
		}
		uint64 max_out_recs = (b.n_rec + 1) / max(cutoff_min, 1u);
		uint64 counter_size = calc_counter_size(cutoff_max, counter_max);
		uint32 kmer_symbols = kmer_len - lut_prefix_len;
		uint64 kmer_bytes = kmer_symbols / 4;
		if (lut_prefix_len == 0)
			kmer_bytes = (kmer_symbols + 3) / 4;
		uint64 out_buffer_size = max_out_recs * (kmer_bytes + counter_size);
		b.rec_len = (kxmer_symbols + 3) / 4;
		uint64 lut_recs = lut_prefix_len ? 1ull << (2 * lut_prefix_len) : 0;
		b.a_size = round_up_to_alignment(b.size);
		b.a_kxmers = round_up_to_alignment(input_kmer_size);
		b.a_out = round_up_to_alignment(out_buffer_size);
		b.a_counters = round_up_to_alignment(kxmer_counter_size);
		b.a_lut = round_up_to_alignment(lut_recs * sizeof(uint64));
	}

	/* The arena of CMemoryBins (kmc.h:1511: one anonymous mapping of max_mem_stage2 bytes, -m) is touched once per bin image and output buffer and
	 * released at the end of stage 2 (kmc.h:1602-1605, joined before the "2nd stage" timer stops): with 4 KB pages the faults of the reader threads and,
	 * above all, the unmapping of the ~2.3 GB a 2 Gbp run has touched cost 0.17 s of the 0.39 s the stage takes once the sort is on the GPU (round 3's
	 * timeline: "files closed 0.220 | process exit 0.388"). Where transparent huge pages are in `madvise` mode (the GPU boxes of this pool) the mapping is
	 * put on 2 MB pages before its first touch: 512x fewer faults and page-table entries to tear down. The first reserved pointer locates the mapping
	 * (/proc/self/maps). KMC_HIP_ARENA_THP=0 switches it off. */
	static void advise_arena_once(const void *inside, uint64 arena_bytes)
	{
		KmcArena::inst().note(inside, arena_bytes); /* every call: the completer's zap looks the block up again from the LATEST pointer */
		static std::once_flag once;
		std::call_once(once, [inside, arena_bytes] {
			const char *e = getenv("KMC_HIP_ARENA_THP");
			const bool thp = !(e && atoi(e) == 0);
			uintptr_t blo = 0, bhi = 0;
			/* a small arena lives in the heap, next to everything else, and is left alone; a large one is the allocator's own mapping — identified by the
			 * allocator's chunk header, never by the bounds of the VMA (KmcArena::find_block: adjacent mappings are merged into one VMA) */
			const bool found = arena_bytes >= (64ull << 20) && KmcArena::find_block(inside, arena_bytes, blo, bhi);
			if (!found) {
				if (getenv("KMC_HIP_VERBOSE"))
					fprintf(stderr, "[kmc_hip stage 2] arena %.1f GB: its block was not identified (no huge-page advice, no parallel zap)\n", (double)arena_bytes / 1e9);
				return;
			}
			KmcArena::inst().lo.store(blo);
			KmcArena::inst().hi.store(bhi);
			const uintptr_t two_mb = (uintptr_t)2 << 20;
			const uintptr_t lo = (blo + two_mb - 1) & ~(two_mb - 1), hi = bhi & ~(two_mb - 1);
			if (hi > lo && thp) {
				const int rc = madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
				if (getenv("KMC_HIP_VERBOSE"))
					fprintf(stderr, "[kmc_hip stage 2] arena %.1f GB: madvise(MADV_HUGEPAGE) %s\n", (double)(bhi - blo) / 1e9, rc == 0 ? "ok" : "refused");
			}
		});
	}

	/* read (any order) -> wait for this bin's turn -> extend -> push (kb_reader.h:167-205) */
	void load_and_push(BinPlan &b)
	{
		uchar *data = nullptr, *pinned = nullptr;
		if (b.size > 0) {
			if (b.file == nullptr) {
				std::ostringstream ostr;
				ostr << "Error: Cannot open temporary file: " << b.name;
				CCriticalErrorHandler::Inst().HandleCriticalError(ostr.str());
			}
			b.file->Rewind();
			memory_bins->reserve(b.bin_id, data, CMemoryBins::mba_input_file);
			advise_arena_once(data, (uint64)memory_bins->GetTotalSize());
			/* a pinned buffer of the engine's, if it has any to give: the image never touches the arena (its space there stays reserved, as the protocol
			 * wants, but no page of it is faulted in); the worker plug-in returns the buffer */
			static const int pool_wait_ms = [] {
				const char *e = getenv("KMC_HIP_POOL_WAIT_MS"); /* 0: never wait for a pool buffer (round 5: straight to the arena) */
				return e ? atoi(e) : 2000;
			}();
			pinned = (uchar *)KmcHostPool::inst().get_wait(b.size, pool_wait_ms);
			if (pinned)
				data = pinned;
			const long long t0 = KmcOrderedEmit::now_ns();
			uint64 readed = b.file->Read(data, 1, b.size);
			order->ns_reader_read += KmcOrderedEmit::now_ns() - t0;
			if (readed != b.size) {
				std::ostringstream ostr;
				ostr << "Error: Corrupted file: " << b.name << "   " << "Real size : " << readed << "   " << "Should be : " << b.size;
				CCriticalErrorHandler::Inst().HandleCriticalError(ostr.str());
			}
		}
		std::unique_lock<std::mutex> lck(push_mtx);
		push_cv.wait(lck, [&] { return push_turn == b.seq; });
		memory_bins->extend(b.bin_id, b.rec_len, b.a_size, b.a_kxmers, b.a_out, b.a_counters, b.a_lut);
		if (b.size > 0) {
			memory_bins->reserve(b.bin_id, data, CMemoryBins::mba_input_file);
			bq->push(b.bin_id, pinned ? pinned : data, b.size, b.n_rec);
		} else {
			bq->push(b.bin_id, nullptr, 0, 0); /* empty bins are pushed too: every bin id must be processed */
		}
		sorters_manager->NotifyBQPush();
		++push_turn;
		lck.unlock();
		push_cv.notify_all();
	}

	void retire_file(BinPlan &b, CPercentProgress &percent_progress)
	{
		if (b.file) {
			b.file->Close();
#ifdef DEVELOP_MODE
			if (!verbose_log)
				b.file->Remove();
#else
			b.file->Remove();
#endif
		}
		disk_logger->log_remove(b.size);
		std::lock_guard<std::mutex> lck(progress_mtx);
		percent_progress.NotifyProgress(b.n_rec);
	}

	void reader_thread(CPercentProgress &percent_progress)
	{
		while (true) {
			BinPlan b;
			bool loaded = false;
			{
				std::lock_guard<std::mutex> lck(admit_mtx);
				const long long t0 = KmcOrderedEmit::now_ns();
				b.bin_id = bd->get_next_sort_bin();
				if (b.bin_id < 0)
					break;
				bd->read(b.bin_id, b.file, b.name, b.size, b.n_rec, b.n_plus_x_recs);
				plan(b);
				if (!memory_bins->init(b.bin_id, b.rec_len, b.a_size, b.a_kxmers, b.a_out, b.a_counters, b.a_lut)) {
					tlbq->insert(b.bin_id); /* strict-memory mode: handled after stage 2 (kb_reader.h:161-165) */
					continue;
				}
				b.seq = next_seq++;
				uchar *suffix = nullptr;
				memory_bins->reserve(b.bin_id, suffix, CMemoryBins::mba_suffix);
				order->ns_reader_init += KmcOrderedEmit::now_ns() - t0;
				if (suffix == nullptr || n_readers == 1) { /* tight arena: only the file bytes were reserved (queues.h:1301-1312) */
					load_and_push(b);
					loaded = true;
				}
			}
			if (!loaded)
				load_and_push(b);
			retire_file(b, percent_progress);
		}
	}

public:
	CWKmerBinReader(CKMCParams &Params, CKMCQueues &Queues)
	{
		bd = Queues.bd.get();
		bq = Queues.bq.get();
		sorters_manager = Queues.sorters_manager.get();
		tlbq = Queues.tlbq.get();
		disk_logger = Queues.disk_logger.get();
		memory_bins = Queues.memory_bins.get();
		order = KmcOrderedEmit::for_queue(Queues.kq.get());

		kmer_len = (uint32)Params.kmer_len;
		cutoff_min = Params.cutoff_min;
		cutoff_max = (uint32)Params.cutoff_max;
		counter_max = (uint32)Params.counter_max;
		max_x = Params.max_x;
		lut_prefix_len = Params.lut_prefix_len;
#ifdef DEVELOP_MODE
		verbose_log = Params.verbose_log;
#endif
		percentProgressObserver = Params.percentProgressObserver;
		n_readers = 8;
		if (const char *e = getenv("KMC_HIP_READERS"))
			n_readers = atoi(e);
		if (n_readers < 1)
			n_readers = 1;
		if (n_readers > 64)
			n_readers = 64;
	}

	void operator()()
	{
		KmcTimeline::mark("reader start");
		const long long t0 = KmcOrderedEmit::now_ns();
		CPercentProgress percent_progress("Stage 2: ", true, percentProgressObserver);
		percent_progress.SetMaxVal(bd->get_n_rec_sum());
		percent_progress.NotifyProgress(0);
		order->n_readers = n_readers;
		{
			std::vector<CExceptionAwareThread> helpers;
			for (int i = 1; i < n_readers; ++i)
				helpers.emplace_back([this, &percent_progress] { reader_thread(percent_progress); });
			try {
				reader_thread(percent_progress);
			} catch (...) {
				for (auto &t : helpers)
					t.join();
				throw;
			}
			for (auto &t : helpers)
				t.join();
		}
		bq->mark_completed();
		sorters_manager->NotifyQueueCompleted();
		fflush(stdout);
		order->ns_reader_wall += KmcOrderedEmit::now_ns() - t0;
		KmcTimeline::mark("reader done");
	}
};

#endif
