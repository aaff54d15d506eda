/*
 * kmc_amd/host/kb_sorter_plugin.h — the stage-2 worker of kmc_core, re-implemented over a bin engine.
 *
 * Drop-in for the reference's kmc_core/kb_sorter.h: it defines the same class template
 *     template <unsigned SIZE> class CWKmerBinSorter      (reference: kb_sorter.h:1298-1322)
 * with the same constructor, operator()() and GetDebugStats(), so CKMC<SIZE>::ProcessStage2_impl
 * (kmc.h:1576-1584, :1736-1741) builds against it unchanged. Instead of Expand/Sort/Compact on the CPU
 * (kb_sorter.h:223-231) each bin is handed to a KmcBinEngine (bin_engine.h): the HIP engine behind
 * include/kmc_hip.h (hip_loader.cpp). (Test infrastructure plugs its own engine in through the same interface:
 * oracle/oracle_engine.h, used only by the oracle-pinning build `kmc_oracle`.)
 *
 * How it is compiled in (oracle/Makefile): kmc_runner.cpp, the one translation unit that
 * instantiates CKMC<SIZE>, is compiled with
 *     -D_KB_SORTER_H  -include kmc_amd/host/kb_sorter_plugin.h  -I <kmc_core>
 * so the reference's own `#include "kb_sorter.h"` (kmc.h:30) is skipped by its include guard and this
 * definition is the one seen. No reference source is modified or copied.
 *
 * Protocol obligations kept (SURVEY.md §8b "Full boundary"):
 *   per bin : sorters_manager->GetNext (queues.h:2087) -> bd->read (queues.h:654) -> epd->pop (queues.h:390)
 *             -> reserve mba_suffix / mba_lut -> engine -> free mba_input_file, mba_input_array, mba_tmp_array,
 *             mba_kxmer_counters -> exactly one kq->push (queues.h:826), also for empty bins
 *             -> sorters_manager->ReturnThreads (queues.h:2130)
 *   per run : kq->mark_completed() once per worker (kb_sorter.h:236)
 *   errors  : CCriticalErrorHandler::Inst().HandleCriticalError(msg) (critical_error_handler.h)
 *   order   : bins are pushed in the order GetNext hands them out (KmcOrderedEmit below), so the .kmc_pre/.kmc_suf
 *             bytes equal the reference's -sr1 bytes for any number of workers (SURVEY.md §4 determinism finding).
 */
#ifndef KMC_AMD_KB_SORTER_PLUGIN_H
#define KMC_AMD_KB_SORTER_PLUGIN_H

/* the headers kb_sorter.h would have pulled in for the rest of kmc.h (kb_sorter.h:16-34) */
#include "defs.h"
#include "params.h"
#include "kmer.h"

/* small_sort.h is one of the replaced subsystems too. On non-Intel hosts ProcessStage2_impl calls
 * CSmallSort<SIZE>::Adjust(384) (kmc.h:1559), which benchmarks six CPU small-array sorters for 1.0-1.4 s (small_sort.h:68-103,
 * :154-172) — more than the whole GPU stage 2 of a 2 Gbp input — to tune a CPU radix sort the GPU worker never runs.
 * Its include guard is taken here, with the same interface: Adjust is a no-op, Sort (still reachable from
 * RadixSort::SmallSortDispatch, radix.h:37-41, if strict-memory mode sorts an oversized bin on the CPU) is std::sort
 * on CKmer's operator< (kmer.h:271-278), i.e. the same order. Define KMC_PLUGIN_KEEP_SMALL_SORT to keep the original. */
#if !defined(_SMALL_SORT_H) && !defined(KMC_PLUGIN_KEEP_SMALL_SORT)
#define _SMALL_SORT_H
#include <cstdint>
#include <chrono>
#include <stdlib.h>
#include <random>
#include <algorithm>
#include <vector>
#include <functional>
#include <array>
#include <string>
using namespace std; /* small_sort.h:26 — later reference headers rely on it */
template <unsigned SIZE> class CSmallSort {
public:
	static void Adjust(uint32 /*arr_size*/ = 384) {}
	static void Sort(CKmer<SIZE> *ptr, uint32 size) { std::sort(ptr, ptr + size); }
};
#endif

#include "raduls.h"
#include "radix.h"
#include "s_mapper.h"
#include <string>
#include <algorithm>
#include <numeric>
#include <array>
#include <vector>
#include <stdio.h>
#include <functional>
#include <cstddef>
#include <set>
#include <map>
#include <mutex>
#include <atomic>
#include <memory>
#include <sstream>
#include "kxmer_set.h"
#include "rev_byte.h"
#include "critical_error_handler.h"

#include "bin_engine.h"

#include "kmc_order.h" /* KmcOrderedEmit: emission order + KMC_HIP_VERBOSE statistics, shared with kb_reader_plugin.h */

template <unsigned SIZE> class CWKmerBinSorter {
	std::shared_ptr<KmcOrderedEmit> order;
	CBinDesc *bd;
	CBinQueue *bq;
	CExpanderPackDesc *epd;
	CKmerQueue *kq;
	CMemoryBins *memory_bins;
	CSortersManager *sorters_manager;

	kmc_hip_bin_params bp;
	uint32 max_x;
	uint64 sum_n_rec = 0, sum_n_plus_x_rec = 0;
	int worker_idx, n_workers;

	static std::atomic<int> &worker_counter()
	{
		static std::atomic<int> c{0};
		return c;
	}

public:
	CWKmerBinSorter(CKMCParams &Params, CKMCQueues &Queues, SortFunction<CKmer<SIZE>> /*sort_func: CPU sorter, unused*/)
	{
		bd = Queues.bd.get();
		bq = Queues.bq.get();
		epd = Queues.epd.get();
		kq = Queues.kq.get();
		memory_bins = Queues.memory_bins.get();
		sorters_manager = Queues.sorters_manager.get();
		order = KmcOrderedEmit::for_queue(kq);
		if (worker_counter().load() == 0)
			KmcTimeline::mark("first worker constructed");

		bp.kmer_len = Params.kmer_len;
		bp.both_strands = Params.both_strands ? 1 : 0;
		bp.cutoff_min = Params.cutoff_min;
		bp.without_output = Params.without_output ? 1 : 0;
		bp.cutoff_max = (uint64)Params.cutoff_max;
		bp.counter_max = (uint64)Params.counter_max;
		bp.lut_prefix_len = Params.lut_prefix_len;
		bp.output_type = Params.output_type == OutputType::KMC ? 0 : 1;
		max_x = Params.max_x;
		n_workers = Params.n_sorters;
		worker_idx = worker_counter()++ % (n_workers > 0 ? n_workers : 1);
	}

	~CWKmerBinSorter() { KmcTimeline::mark_first_last(nullptr, "worker objects destroyed (kmc.h:1735-1742)"); }

	void GetDebugStats(uint64 &_sum_n_recs, uint64 &_sum_n_plus_x_recs)
	{
		_sum_n_recs = sum_n_rec;
		_sum_n_plus_x_recs = sum_n_plus_x_rec;
	}

	void operator()()
	{
		std::unique_ptr<KmcBinEngine> engine(kmc_make_bin_engine(worker_idx, n_workers));
		if (!engine)
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: cannot create the stage-2 bin engine");

		const uint64 lut_recs = bp.lut_prefix_len ? 1ull << (2 * bp.lut_prefix_len) : 0;
		const uint64 out_rec_bytes = kmc_hip_out_rec_bytes_host(bp);
		/* a bin this worker holds between GetNext and kq->push */
		struct Taken {
			int32 bin_id = 0;
			uchar *data = nullptr;
			uint64 size = 0, n_rec = 0, seq = 0, tmp_size = 0, n_plus_x_recs = 0, out_capacity = 0;
			int n_sorting_threads = 0; /* 0: taken straight from the bin queue (below), nothing to hand back to the sorters manager */
			std::vector<uint64> pack_bytes;
			uchar *out_buffer = nullptr, *raw_lut = nullptr;
		};
		/* Bins that are ALREADY waiting when this worker comes for one are taken together (up to KMC_HIP_WORKER_GROUP, default 4, 1 = off) and handed
		 * to the engine in one call: on the device they share one sort (kmc_hip_process_bins_submit). Only bins the reader has finished are taken —
		 * CBinQueue::pop_if_any (queues.h:751) never waits — so a worker never holds bins while it waits for more, and the memory they occupy was
		 * reserved by the reader before it read them: nothing new can block. */
		static const size_t max_group = [] {
			const char *e = getenv("KMC_HIP_WORKER_GROUP");
			const int v = e ? atoi(e) : 4;
			return (size_t)(v < 1 ? 1 : (v > 16 ? 16 : v));
		}();

		/* ... and only while the worker holds few k-mers: a call with several LARGE bins uploads all of them before its first kernel starts, and through
		 * PCIe that costs more overlap than the shared sort saves (bench.py host_boundary.legs: 48 M k-mers per bin, one per call 29.2, four per call 23-27
		 * Gk-mers/s); small bins are bound by launches, and there the shared sort pays (DESIGN.md 4 "Groups of bins") */
		static const uint64 group_recs = [] {
			const char *e = getenv("KMC_HIP_WORKER_GROUP_KMERS");
			return e ? (uint64)strtoull(e, nullptr, 10) : (uint64)4 << 20;
		}();

		const long long t_start = KmcOrderedEmit::now_ns();
		std::vector<Taken> grp;
		while (true) {
			grp.clear();
			{
				const long long t0 = KmcOrderedEmit::now_ns();
				std::lock_guard<std::mutex> lck(order->take_mtx);
				Taken t;
				if (!sorters_manager->GetNext(t.bin_id, t.data, t.size, t.n_rec, t.n_sorting_threads))
					break;
				t.seq = order->next_take++;
				if (t.seq == 0)
					KmcTimeline::mark("first bin taken");
				grp.push_back(std::move(t));
				uint64 held = grp[0].n_rec;
				while (grp.size() < max_group && held < group_recs) {
					Taken e;
					if (!bq->pop_if_any(e.bin_id, e.data, e.size, e.n_rec))
						break;
					e.seq = order->next_take++;
					held += e.n_rec;
					grp.push_back(std::move(e));
				}
				order->ns_getnext += KmcOrderedEmit::now_ns() - t0;
			}
			for (Taken &t : grp) {
				CMemDiskFile *file;
				string desc;
				uint64 tmp_n_rec;
				bd->read(t.bin_id, file, desc, t.tmp_size, tmp_n_rec, t.n_plus_x_recs);
				sum_n_rec += t.n_rec;
				sum_n_plus_x_rec += t.n_plus_x_recs;
				if (const char *dd = getenv("KMC_HIP_BINDESC_DUMP")) { /* tests: what stage 1 recorded for this bin (CBinDesc, queues.h:643-680) */
					static std::mutex dump_mtx;
					std::lock_guard<std::mutex> lck(dump_mtx);
					if (FILE *f = fopen(dd, "a")) {
						fprintf(f, "%d %llu %llu %llu\n", (int)t.bin_id, (unsigned long long)t.tmp_size, (unsigned long long)tmp_n_rec, (unsigned long long)t.n_plus_x_recs);
						fclose(f);
					}
				}
				list<pair<uint64, uint64>> packs;
				epd->pop(t.bin_id, packs);
				for (auto &e : packs)
					t.pack_bytes.push_back(e.first);
				if (const char *dir = getenv("KMC_HIP_BIN_DUMP_DIR")) { /* measurement aid (bench.py e2e_large): the bins exactly as the reference's stage 1 handed them over */
					const std::string base = std::string(dir) + "/bin_" + std::to_string(t.bin_id);
					if (FILE *f = fopen((base + ".img").c_str(), "wb")) {
						if (t.tmp_size)
							fwrite(t.data, 1, t.tmp_size, f);
						fclose(f);
					}
					if (FILE *f = fopen((base + ".meta").c_str(), "w")) {
						fprintf(f, "%d %llu %llu %llu\n", (int)t.bin_id, (unsigned long long)t.tmp_size, (unsigned long long)t.n_rec, (unsigned long long)t.pack_bytes.size());
						for (uint64 pb : t.pack_bytes)
							fprintf(f, "%llu\n", (unsigned long long)pb);
						fclose(f);
					}
				}
				memory_bins->reserve(t.bin_id, t.out_buffer, CMemoryBins::mba_suffix);
				memory_bins->reserve(t.bin_id, t.raw_lut, CMemoryBins::mba_lut);
				/* capacity of mba_suffix exactly as the reader sized it (kb_reader.h:141-150) */
				const uint64 max_out_recs = (t.n_rec + 1) / max((uint32)bp.cutoff_min, 1u);
				t.out_capacity = max_out_recs * out_rec_bytes;
			}

			uint64 out_bytes[16] = {};
			uint64 stats[16][4] = {};
			const long long t1 = KmcOrderedEmit::now_ns();
			int rc;
			if (grp.size() == 1) {
				Taken &t = grp[0];
				rc = engine->process_bin(bp, t.data, t.tmp_size, t.n_rec, t.pack_bytes.data(), t.pack_bytes.size(), t.out_buffer, t.out_capacity, &out_bytes[0],
				                         (uint64 *)t.raw_lut, stats[0]);
			} else {
				kmc_hip_host_bin hb[16];
				for (size_t i = 0; i < grp.size(); ++i) {
					Taken &t = grp[i];
					hb[i].superkmers = t.data;
					hb[i].size = t.tmp_size;
					hb[i].n_rec = t.n_rec;
					hb[i].pack_bytes = t.pack_bytes.data();
					hb[i].n_packs = t.pack_bytes.size();
					hb[i].out_suffix = t.out_buffer;
					hb[i].out_capacity = t.out_capacity;
					hb[i].lut = (uint64_t *)t.raw_lut;
				}
				rc = engine->process_bins(bp, hb, (uint32_t)grp.size(), out_bytes, &stats[0][0]);
				++order->n_group_calls;
			}
			order->ns_engine += KmcOrderedEmit::now_ns() - t1;
			if (rc != 0) {
				std::ostringstream ostr;
				ostr << "Error: stage-2 bin engine failed on bin " << grp[0].bin_id;
				if (grp.size() > 1)
					ostr << " .. " << grp.back().bin_id << " (one call)";
				ostr << " (code " << rc << "): " << engine->last_error();
				CCriticalErrorHandler::Inst().HandleCriticalError(ostr.str());
			}

			for (size_t i = 0; i < grp.size(); ++i) {
				Taken &t = grp[i];
				/* the CPU sorter's working slots are never used here, but the region only recycles once every
				 * slot of the bin is released (queues.h:1556-1583) */
				(void)KmcHostPool::inst().put(t.data); /* the reader plug-in's pinned buffer, if the image came in one */
				memory_bins->free(t.bin_id, CMemoryBins::mba_input_file);
				memory_bins->free(t.bin_id, CMemoryBins::mba_input_array);
				memory_bins->free(t.bin_id, CMemoryBins::mba_tmp_array);
				if (max_x && t.n_plus_x_recs)
					memory_bins->free(t.bin_id, CMemoryBins::mba_kxmer_counters);

				/* data packs exactly as the reference emits them: one pack [0,out_bytes); none when output is off,
				 * and none for an empty bin on the k+x-mer path (kb_sorter.h:967-968,1105-1106,1269-1271) */
				list<pair<uint64, uint64>> data_packs;
				if (!bp.without_output && !(max_x && t.n_plus_x_recs == 0))
					data_packs.emplace_back(0, out_bytes[i]);
				KmcOrderedEmit::Ready rdy{t.bin_id, t.out_buffer, std::move(data_packs), t.raw_lut, lut_recs * sizeof(uint64), stats[i][0], stats[i][1], stats[i][2], stats[i][3]};
				order->deposit(kq, t.seq, std::move(rdy)); /* pushed in sequence, by this thread or by the one that deposits the bin in front of it (kmc_order.h) */
				++order->n_bins;
				if (t.n_sorting_threads)
					sorters_manager->ReturnThreads(t.n_sorting_threads, t.bin_id);
			}
		}
		order->ns_worker_wall += KmcOrderedEmit::now_ns() - t_start;
		if (++order->n_workers_done == n_workers) {
			KmcTimeline::mark("last worker done");
			order->report(n_workers);
		}
		kq->mark_completed();
	}

private:
	static uint64 kmc_hip_out_rec_bytes_host(const kmc_hip_bin_params &p)
	{
		uint32 sym = p.kmer_len - p.lut_prefix_len;
		uint64 kmer_bytes = p.lut_prefix_len ? sym / 4 : (sym + 3) / 4; /* kb_reader.h:146-149 */
		return kmer_bytes + calc_counter_size((int64)p.cutoff_max, (int64)p.counter_max);
	}
};

#endif
