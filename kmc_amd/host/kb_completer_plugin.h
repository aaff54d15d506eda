/*
 * kmc_amd/host/kb_completer_plugin.h — the stage-2 bin completer of kmc_core, decoupled from the database file writes
 * ("completer decoupling").
 *
 * Drop-in for the wrapper class of the reference's kmc_core/kb_completer.h:
 *     class CWKmerBinCompleter        (reference: kb_completer.h:74-86, kb_completer.cpp:340-370)
 * with the same constructor, operator()(bool first_stage), GetTotal() and InitStage2(), so CKMC<SIZE>::ProcessStage2_impl
 * (kmc.h:1589-1590, :1697, :1728) builds against it unchanged. Compiled in with -include kb_completer_plugin.h; the reference's
 * own CKmerBinCompleter / CWKmerBinCompleter are still compiled (from the reference's kb_completer.cpp, class names suffixed
 * _ref by the oracle/Makefile recipe) and are what runs for everything the fast path does not cover: strict-memory mode
 * (-sm: stage 2 continues with the big-bin merge), KFF output, runs without output, or KMC_HIP_COMPLETER=ref.
 *
 * Why: with the sort on a GPU and the bin files read by several threads, stage 2 of a 2 Gbp input (512 bins) takes ~0.2 s
 * up to the last kq->push — and the reference's single completer thread needs ~0.35 s for the same bins: per bin it copies the
 * suffix records into the page cache with fwrite (0.54 GB in all) and scans ALL 4^9+1 signatures to find the bin's own
 * (kb_completer.cpp:207-213: 134 M probes per run).
 * Here the thread that pops kq only does the order-dependent work — file offsets, the running LUT prefix sum
 * (kb_completer.cpp:187-199), signature -> LUT index through a bin -> signatures table built once — and hands the suffix data
 * to $KMC_HIP_WRITERS (default 8) threads that pwrite() it at its final offset and release the bin's mba_suffix slot.
 *
 * The files are byte-identical to the reference's (format: kb_completer.cpp:119-127, :141-199, :283-324):
 *   .kmc_suf  "KMCS" | records of every bin in kq order | "KMCS"
 *   .kmc_pre  "KMCP" | per bin: 4^p running record offsets (uint64) | total records (uint64) | sig_map[4^s + 1] (uint32) |
 *             header: k, mode 0, counter bytes, p, s, cutoff_min, cutoff_max (uint32 each), counted k-mers (uint64),
 *             both_strands ? 0 : 1 (1 byte), 27 zero bytes, 0x200 (uint32), header length (uint32) | "KMCP"
 */
#ifndef KMC_AMD_KB_COMPLETER_PLUGIN_H
#define KMC_AMD_KB_COMPLETER_PLUGIN_H

#define CKmerBinCompleter CKmerBinCompleter_ref
#define CWKmerBinCompleter CWKmerBinCompleter_ref
#include "kb_completer.h" /* the reference header, read-only: declares the *_ref classes (and CSmallKCompleter, untouched) */
#undef CKmerBinCompleter
#undef CWKmerBinCompleter

#include <condition_variable>
#include <deque>
#include <fcntl.h>
#include <sstream>
#include <unistd.h>
#include "critical_error_handler.h"
#include "exception_aware_thread.h"
#include "kmc_order.h"

class CWKmerBinCompleter {
	std::unique_ptr<CWKmerBinCompleter_ref> ref; /* non-null: the reference completer runs (modes listed in the header comment) */

	/* --- fast path --- */
	CKmerQueue *kq = nullptr;
	CSignatureMapper *s_mapper = nullptr;
	CMemoryBins *memory_bins = nullptr;
	std::string suf_name, pre_name;
	uint32 kmer_len = 0, signature_len = 0, lut_prefix_len = 0, cutoff_min = 0, cutoff_max = 0, counter_max = 0, n_bins = 0;
	bool both_strands = true;
	int fd_suf = -1;
	FILE *out_pre = nullptr;
	std::vector<uint32> sig_map;
	uint32 lut_pos = 0;
	uint64 suf_offset = 0, n_recs = 0;
	uint64 n_unique = 0, n_cutoff_min = 0, n_cutoff_max = 0, n_total = 0;

	struct Job {
		int32 bin_id;
		const uchar *data;
		uint64 len, offset;
		bool last; /* release the bin's mba_suffix slot after this write */
	};
	std::mutex jobs_mtx;
	std::condition_variable jobs_cv;
	std::deque<Job> jobs;
	bool jobs_done = false;

	void writer_thread()
	{
		while (true) {
			Job j;
			{
				std::unique_lock<std::mutex> lck(jobs_mtx);
				jobs_cv.wait(lck, [this] { return jobs_done || !jobs.empty(); });
				if (jobs.empty())
					return;
				j = jobs.front();
				jobs.pop_front();
			}
			uint64 done = 0;
			while (done < j.len) {
				ssize_t w = pwrite(fd_suf, j.data + done, j.len - done, (off_t)(j.offset + done));
				if (w <= 0) {
					std::ostringstream ostr;
					ostr << "Error: Cannot write to " << suf_name;
					CCriticalErrorHandler::Inst().HandleCriticalError(ostr.str());
				}
				done += (uint64)w;
			}
			if (j.last)
				memory_bins->free(j.bin_id, CMemoryBins::mba_suffix);
		}
	}

	static void put_le(FILE *f, uint64 x, uint32 bytes)
	{
		for (uint32 i = 0; i < bytes; ++i)
			putc((int)((x >> (8 * i)) & 0xFF), f);
	}

	void first_stage()
	{
		KmcTimeline::mark("completer start");
		fd_suf = open(suf_name.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0666);
		if (fd_suf < 0)
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: Cannot create " + suf_name);
		out_pre = fopen(pre_name.c_str(), "wb");
		if (!out_pre) {
			close(fd_suf);
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: Cannot create " + pre_name);
		}
		fwrite("KMCP", 1, 4, out_pre);
		if (pwrite(fd_suf, "KMCS", 4, 0) != 4)
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: Cannot write to " + suf_name);
		suf_offset = 4;

		/* bin -> its signatures, once (the reference scans all signatures for every bin) */
		const uint32 sig_map_size = (1u << (signature_len * 2)) + 1;
		sig_map.assign(sig_map_size, 0);
		std::vector<uint32> sig_start(n_bins + 1, 0), sig_list(sig_map_size);
		for (uint32 i = 0; i < sig_map_size; ++i) {
			const int32 b = s_mapper->get_bin_id(i);
			if (b >= 0 && (uint32)b < n_bins)
				++sig_start[b + 1];
		}
		for (uint32 b = 0; b < n_bins; ++b)
			sig_start[b + 1] += sig_start[b];
		{
			std::vector<uint32> fill(sig_start.begin(), sig_start.end() - 1);
			for (uint32 i = 0; i < sig_map_size; ++i) {
				const int32 b = s_mapper->get_bin_id(i);
				if (b >= 0 && (uint32)b < n_bins)
					sig_list[fill[b]++] = i;
			}
		}

		int n_writers = 8; /* round 6: with 4, the writers were 0.3-0.6 s behind the last push at 30 Gbp (7 GB of suffix records; profiles/r06/e2e_sweep_30gbp_session_y.jsonl: "2nd stage" 2.96 -> 2.34 s) */
		if (const char *e = getenv("KMC_HIP_WRITERS"))
			n_writers = atoi(e);
		n_writers = n_writers < 1 ? 1 : (n_writers > 32 ? 32 : n_writers);
		std::vector<CExceptionAwareThread> writers;
		for (int i = 0; i < n_writers; ++i)
			writers.emplace_back([this] { writer_thread(); });
		auto stop_writers = [&] {
			{
				std::lock_guard<std::mutex> lck(jobs_mtx);
				jobs_done = true;
			}
			jobs_cv.notify_all();
			for (auto &t : writers)
				t.join();
		};

		try {
			int32 bin_id = 0;
			uchar *data = nullptr, *lut = nullptr;
			list<pair<uint64, uint64>> data_packs;
			uint64 lut_size = 0, u = 0, cmin = 0, cmax = 0, tot = 0;
			while (!kq->empty()) {
				if (!kq->pop(bin_id, data, data_packs, lut, lut_size, u, cmin, cmax, tot))
					continue;
				/* suffix records: file offsets are assigned here, in kq order; the bytes are written by the pool */
				{
					std::lock_guard<std::mutex> lck(jobs_mtx);
					size_t left = data_packs.size();
					for (auto &e : data_packs) {
						jobs.push_back(Job{bin_id, data + e.first, e.second - e.first, suf_offset, --left == 0});
						suf_offset += e.second - e.first;
					}
				}
				if (data_packs.empty())
					memory_bins->free(bin_id, CMemoryBins::mba_suffix);
				else
					jobs_cv.notify_all();
				/* LUT: per-bin counts -> running offsets over the whole database (kb_completer.cpp:187-199) */
				const uint64 lut_recs = lut_size / sizeof(uint64);
				uint64 *ulut = (uint64 *)lut;
				for (uint64 i = 0; i < lut_recs; ++i) {
					const uint64 x = ulut[i];
					ulut[i] = n_recs;
					n_recs += x;
				}
				if (lut_recs)
					fwrite(lut, sizeof(uint64), lut_recs, out_pre);
				memory_bins->free(bin_id, CMemoryBins::mba_lut);
				n_unique += u;
				n_cutoff_min += cmin;
				n_cutoff_max += cmax;
				n_total += tot;
				if (bin_id >= 0 && (uint32)bin_id < n_bins)
					for (uint32 q = sig_start[bin_id]; q < sig_start[bin_id + 1]; ++q)
						sig_map[sig_list[q]] = lut_pos;
				++lut_pos;
			}
		} catch (...) {
			stop_writers();
			throw;
		}
		KmcTimeline::mark("completer: queue drained");
		stop_writers();
		KmcTimeline::mark("completer: writers joined");
		KmcArena::inst().zap_parallel(); /* every bin is written: the arena's pages go now, in parallel, instead of under the reference's release at the end of the stage */
		KmcTimeline::mark("completer: arena pages returned");
	}

	void second_stage()
	{
		if (pwrite(fd_suf, "KMCS", 4, (off_t)suf_offset) != 4 || close(fd_suf) != 0)
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: Cannot write to " + suf_name);
		fd_suf = -1;
		fwrite(&n_recs, 1, sizeof(uint64), out_pre);
		fwrite(sig_map.data(), sizeof(uint32), sig_map.size(), out_pre);
		const uint32 counter_size = (uint32)calc_counter_size(cutoff_max, counter_max);
		put_le(out_pre, kmer_len, 4);
		put_le(out_pre, 0, 4); /* mode 0: counting */
		put_le(out_pre, counter_size, 4);
		put_le(out_pre, lut_prefix_len, 4);
		put_le(out_pre, signature_len, 4);
		put_le(out_pre, cutoff_min, 4);
		put_le(out_pre, cutoff_max, 4);
		put_le(out_pre, n_unique - n_cutoff_min - n_cutoff_max, 8);
		put_le(out_pre, both_strands ? 0 : 1, 1);
		for (int i = 0; i < 27; ++i)
			put_le(out_pre, 0, 1);
		put_le(out_pre, 0x200, 4);
		put_le(out_pre, 7 * 4 + 8 + 1 + 27 + 4, 4); /* header length up to here */
		fwrite("KMCP", 1, 4, out_pre);
		if (fclose(out_pre) != 0)
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: Cannot write to " + pre_name);
		out_pre = nullptr;
		KmcTimeline::mark("completer: files closed");
	}

public:
	CWKmerBinCompleter(CKMCParams &Params, CKMCQueues &Queues)
	{
		const char *force = getenv("KMC_HIP_COMPLETER");
		if (Params.use_strict_mem || Params.without_output || Params.output_type != OutputType::KMC || (force && std::string(force) == "ref")) {
			ref = std::make_unique<CWKmerBinCompleter_ref>(Params, Queues);
			return;
		}
		kq = Queues.kq.get();
		s_mapper = Queues.s_mapper.get();
		memory_bins = Queues.memory_bins.get();
		suf_name = Params.output_file_name + ".kmc_suf";
		pre_name = Params.output_file_name + ".kmc_pre";
		kmer_len = (uint32)Params.kmer_len;
		signature_len = (uint32)Params.signature_len;
		lut_prefix_len = (uint32)Params.lut_prefix_len;
		cutoff_min = (uint32)Params.cutoff_min;
		cutoff_max = (uint32)Params.cutoff_max;
		counter_max = (uint32)Params.counter_max;
		both_strands = Params.both_strands;
		n_bins = (uint32)Params.n_bins;
	}

	void operator()(bool first_stage_flag)
	{
		if (ref) {
			(*ref)(first_stage_flag);
			return;
		}
		if (first_stage_flag)
			first_stage();
		else
			second_stage();
	}

	void GetTotal(uint64 &_n_unique, uint64 &_n_cutoff_min, uint64 &_n_cutoff_max, uint64 &_n_total)
	{
		if (ref) {
			ref->GetTotal(_n_unique, _n_cutoff_min, _n_cutoff_max, _n_total);
			return;
		}
		_n_unique = n_unique;
		_n_cutoff_min = n_cutoff_min;
		_n_cutoff_max = n_cutoff_max;
		_n_total = n_total;
	}

	void InitStage2(CKMCParams &Params, CKMCQueues &Queues)
	{
		if (ref)
			ref->InitStage2(Params, Queues); /* strict-memory mode only (kmc.h:1662) */
	}

	~CWKmerBinCompleter()
	{
		if (fd_suf >= 0)
			close(fd_suf);
		if (out_pre)
			fclose(out_pre);
	}
};

#endif
