/*
 * kmc_amd/host/kb_splitter_plugin.h — the stage-1 splitter worker of kmc_core with the splitting done by an engine
 * (docs/history/DESIGN_rounds_1_to_5.md §9): the drop-in boundary of a GPU stage 1.
 *
 * Drop-in for the reference's CWSplitter (splitter.h:100-113, splitter.cpp:814-867): the same class name, constructor, operator()(),
 * GetTotal() and destructor, so CKMC<SIZE>::ProcessStage1_impl (kmc.h:1274-1362) builds against it unchanged. Compiled in with
 * `-include kb_splitter_plugin.h` ahead of kmc_runner.cpp; the reference's own splitter.cpp is compiled with -DCWSplitter=CWSplitter_ref
 * (oracle/Makefile), which is what this header declares first: CSplitter stays the reference's for what is NOT this worker (stage 0's signature
 * statistics, the small-k path and histogram estimation use it directly, kmc.h:1100-1200); this worker itself never calls it (see FAILS CLOSED below).
 *
 * Protocol kept from CWSplitter::operator() + CSplitter::ProcessReads + CKmerBinCollector:
 *   pq->pop(part, size, read_type) -> [engine: text -> sequences -> super-k-mers -> bin records] -> pmm_fastq->free(part)
 *   per bin a buffer of Params.bin_part_size bytes from pmm_bins; records are appended whole; a buffer that cannot take the next piece is
 *   pushed: bpq->push(bin, buffer, used, bin_part_size, {one expander pack: (used, n_plus_x_recs)}) + bd->update(bin, used, n_recs,
 *   n_plus_x_recs, n_super_kmers) (kb_collector.cpp:88-106; the reference never closes a pack early: its super_kmer_no is not incremented
 *   in 3.2.4, kb_collector.cpp:36-41); at the end every non-empty buffer is pushed, then bpq->mark_completed().
 * Differences a downstream stage can see: the ORDER of records inside a bin (records of one part stay together; the reference interleaves
 * reads) — stage 2 sorts, the database is byte-identical — and buffers are reserved when a bin first gets data, not all n_bins up front.
 * The reference pushes a buffer when the next RECORD does not fit; here the unit is the part's piece for that bin (its three sums come
 * from the engine in one go), walked record by record only when a piece is larger than what an empty buffer holds.
 *
 * FAILS CLOSED (round 5): this worker has no path into the reference's CSplitter. What the engine does not cover stops the run through
 * CCriticalErrorHandler with a message that names it —
 *   per job : input other than FASTA / FASTQ (multi-line FASTA, BAM, KMC), homopolymer compression (-hc), histogram estimation while counting (--opt-out-size; -e alone runs the reference's estimate-only worker, not this one):
 *             "use kmc_hip" (the reference's stage 1 + this library's stage 2) is the answer the message gives;
 *   per part: KMC_SPLIT_UNCOVERED — MALFORMED text that CSplitter::GetSeq happens to tolerate (blank lines, a quality string of another length than its
 *             sequence, control characters, a lone '\r').
 * Parts the reader labelled ReadType::long_read (queues.h:40) and lines of mem_part_pmm_reads symbols or more go through the engine like any other.
 * Only a build with -DKMC_HIP_S1_REFERENCE_FALLBACK (no shipped binary has it; oracle/_ref/kmc_emu_s1_fb is the test build) AND $KMC_HIP_S1_FALLBACK=1
 * in the environment hands such jobs / parts to the reference (CWSplitter_ref / a CSplitter of this thread, our buffers pushed first) and says so on stderr;
 * nothing run that way is covered by this repo's parity claims.
 */
#ifndef KMC_AMD_KB_SPLITTER_PLUGIN_H
#define KMC_AMD_KB_SPLITTER_PLUGIN_H

#define CWSplitter CWSplitter_ref
#include "splitter.h" /* the reference header, read-only: CSplitter, the *_ref worker, the stats / small-k / estimate workers (untouched) */
#undef CWSplitter

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <list>
#include <memory>
#include <sstream>
#include <vector>

#include "critical_error_handler.h"
#include "split_engine.h"
#include "kmc_order.h" /* KmcTimeline: where does the wall clock go between the stages (KMC_HIP_VERBOSE=1) */

/* n_plus_x_recs of one record (kb_collector.cpp:83-100, kb_collector.h:72-118), from the PACKED record: only needed when a bin's piece of a
 * part has to be cut (it does not fit an empty buffer), because then the engine's sum for the piece cannot be used. */
static inline uint32_t kmc_record_plus_x(const uint8_t *rec, uint32_t kmer_len, uint32_t max_x, bool both_strands)
{
	if (!max_x)
		return 0;
	const uint32_t n = kmer_len + rec[0];
	if (!both_strands)
		return 1 + (n - kmer_len) / (max_x + 1);
	auto sym = [&](uint32_t i) -> uint32_t { return (rec[1 + (i >> 2)] >> (6 - 2 * (i & 3))) & 3u; };
	uint8_t fwd = (uint8_t)((sym(0) << 6) | (sym(1) << 4) | (sym(2) << 2) | sym(3));
	uint8_t rc = (uint8_t)(((3 - sym(kmer_len - 1)) << 6) | ((3 - sym(kmer_len - 2)) << 4) | ((3 - sym(kmer_len - 3)) << 2) | (3 - sym(kmer_len - 4)));
	auto cmp = [](uint8_t a, uint8_t b) { return a < b ? 0 : (b < a ? 1 : 2); };
	int state = cmp(fwd, rc);
	uint32_t run = 0, total = 0;
	for (uint32_t i = 0; i + kmer_len < n; ++i) {
		rc = (uint8_t)((rc >> 2) | ((3 - sym(kmer_len + i)) << 6));
		fwd = (uint8_t)((fwd << 2) | sym(4 + i));
		const int s = cmp(fwd, rc);
		if (s == state) {
			if (s == 2)
				++total; /* a tie: every k-mer is its own record */
			else
				++run;
		} else {
			state = s;
			total += 1 + run / (max_x + 1);
			run = 0;
		}
	}
	return total + 1 + run / (max_x + 1);
}

class CWSplitter {
	struct BinBuf {
		uchar *buf = nullptr;
		uint32 pos = 0;
		uint64 n_recs = 0, n_plus_x = 0, n_super = 0;
	};
	CPartQueue *pq;
	CBinPartQueue *bpq;
	CBinDesc *bd;
	CMemoryPool *pmm_fastq, *pmm_bins;
	CKMCParams *params;
	CKMCQueues *queues;
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
	std::unique_ptr<CWSplitter_ref> ref;    /* non-null: the reference worker runs the whole job */
	std::unique_ptr<CSplitter> ref_splitter; /* uncovered parts of this thread */
	bool fallback_on = false;               /* $KMC_HIP_S1_FALLBACK=1 */
#endif
	std::unique_ptr<KmcSplitEngine> engine;
	std::vector<BinBuf> bins;
	uint32 kmer_len, max_x, buffer_size;
	bool both_strands;
	uint64 n_reads = 0;
	/* KMC_HIP_VERBOSE=1: one line per worker on stderr when it finishes */
	uint64 st_parts = 0, st_long_parts = 0, st_uncovered_parts = 0, st_pieces = 0, st_cut_pieces = 0, st_pushes = 0, st_bytes = 0;
	long long st_engine_ns = 0;

	static const char *uncovered_job(const CKMCParams &P)
	{
		if (P.file_type != InputType::FASTA && P.file_type != InputType::FASTQ)
			return "an input format other than FASTA / FASTQ (multi-line FASTA, BAM, KMC)";
		if (P.homopolymer_compressed)
			return "homopolymer compression (-hc)";
		if (P.estimateHistogramCfg == KMC::EstimateHistogramCfg::ESTIMATE_AND_COUNT_KMERS)
			return "histogram estimation while counting (--opt-out-size)";
		return nullptr;
	}

	void push(uint32 bin_no) /* CKmerBinCollector::Flush, kb_collector.cpp:88-106 */
	{
		BinBuf &b = bins[bin_no];
		if (!b.buf)
			return;
		std::list<std::pair<uint64, uint64>> packs;
		if (b.pos)
			packs.push_back(std::make_pair((uint64)b.pos, b.n_plus_x));
		bpq->push(bin_no, b.buf, b.pos, buffer_size, packs);
		bd->update(bin_no, b.pos, b.n_recs, b.n_plus_x, b.n_super);
		++st_pushes;
		st_bytes += b.pos;
		b = BinBuf();
	}
	void append(uint32 bin_no, const uint8_t *recs, uint64 bytes, uint64 kmers, uint64 supers, uint64 plus_x)
	{
		BinBuf &b = bins[bin_no];
		if (b.buf && b.pos + bytes > buffer_size)
			push(bin_no);
		++st_pieces;
		if (bytes <= buffer_size) {
			if (!b.buf)
				pmm_bins->reserve(b.buf);
			memcpy(b.buf + b.pos, recs, bytes);
			b.pos += (uint32)bytes;
			b.n_recs += kmers;
			b.n_super += supers;
			b.n_plus_x += plus_x;
			return;
		}
		/* a piece larger than a buffer (one bin took most of a part): record by record, like the reference's collector */
		++st_cut_pieces;
		for (uint64 at = 0; at < bytes;) {
			const uint8_t *rec = recs + at;
			const uint32 len = 1 + (kmer_len + rec[0] + 3) / 4;
			if (b.buf && b.pos + len > buffer_size)
				push(bin_no);
			if (!b.buf)
				pmm_bins->reserve(b.buf);
			memcpy(b.buf + b.pos, rec, len);
			b.pos += len;
			b.n_recs += rec[0] + 1u;
			b.n_super += 1;
			b.n_plus_x += kmc_record_plus_x(rec, kmer_len, max_x, both_strands);
			at += len;
		}
	}
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
	/* the reference's own splitter for this part. Its collectors reserve n_bins pmm_bins buffers (one each, CKmerBinCollector's constructor), and the
	 * pool is sized for n_splitters x n_bins buffers beyond the storer's share (kmc.h:491-501): a worker must never hold two sets. So ours go to the
	 * storer first, and the reference splitter is completed (its collectors flushed and their buffers handed over) and dropped as soon as the part is
	 * through — held until the end of the run, a memory-tight run with mixed parts (nanopore FASTQ) could leave every worker waiting in
	 * pmm_bins->reserve with nothing left to free. */
	void to_reference(uchar *part, uint64 size, ReadType read_type)
	{
		push_all();
		ref_splitter = std::make_unique<CSplitter>(*params, *queues);
		ref_splitter->InitBins(*params, *queues);
		ref_splitter->ProcessReads(part, size, read_type);
		pmm_fastq->free(part);
		ref_splitter->Complete();
		uint64 n = 0;
		ref_splitter->GetTotal(n);
		n_reads += n;
		ref_splitter.reset();
	}
#endif
	void push_all()
	{
		for (uint32 i = 0; i < (uint32)bins.size(); ++i)
			push(i);
	}

public:
	CWSplitter(CKMCParams &Params, CKMCQueues &Queues)
	{
		params = &Params;
		queues = &Queues;
		pq = Queues.part_queue.get();
		bpq = Queues.bpq.get();
		bd = Queues.bd.get();
		pmm_fastq = Queues.pmm_fastq.get();
		pmm_bins = Queues.pmm_bins.get();
		kmer_len = Params.kmer_len;
		max_x = Params.max_x;
		both_strands = Params.both_strands;
		buffer_size = Params.bin_part_size;
		if (const char *what = uncovered_job(Params)) {
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
			const char *fb = getenv("KMC_HIP_S1_FALLBACK");
			if (fb && fb[0] == '1') {
				static std::atomic<int> said{0};
				if (!said++)
					fprintf(stderr, "[kmc_hip stage 1] %s: the REFERENCE splitter runs stage 1 of this job (KMC_HIP_S1_FALLBACK=1); nothing of stage 1 is on the device\n", what);
				ref = std::make_unique<CWSplitter_ref>(Params, Queues);
				return;
			}
#endif
			std::ostringstream ostr;
			ostr << "Error: stage 1 on the device does not cover " << what << ". Run this job with kmc_hip (the reference's stage 1, stage 2 on the device).";
			CCriticalErrorHandler::Inst().HandleCriticalError(ostr.str());
		}
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
		{
			const char *fb = getenv("KMC_HIP_S1_FALLBACK");
			fallback_on = fb && fb[0] == '1';
		}
#endif
		bins.resize(Params.n_bins);
		KmcSplitParams sp;
		sp.kmer_len = Params.kmer_len;
		sp.signature_len = Params.signature_len;
		sp.n_bins = Params.n_bins;
		sp.max_x = Params.max_x;
		sp.both_strands = Params.both_strands ? 1 : 0;
		sp.file_type = Params.file_type == InputType::FASTQ ? 1 : 0;
		sp.line_cap = (uint64_t)Params.mem_part_pmm_reads;
		sp.sig_to_bin = Queues.s_mapper->GetMap();
		static std::atomic<int> next_idx{0};
		engine.reset(kmc_make_split_engine(sp, next_idx++ % (int)Params.n_splitters, (int)Params.n_splitters));
		if (!engine)
			CCriticalErrorHandler::Inst().HandleCriticalError("Error: no stage-1 split engine available");
	}

	void operator()()
	{
		KmcTimeline::mark_first_last("splitter: first worker started", nullptr);
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
		if (ref) {
			(*ref)();
			KmcTimeline::mark_first_last(nullptr, "splitter: last worker done");
			return;
		}
#endif
		while (!pq->completed()) {
			uchar *part;
			uint64 size;
			ReadType read_type;
			if (!pq->pop(part, size, read_type))
				continue;
			if (read_type == ReadType::na) /* only the multi-line FASTA reader makes these (fastq_reader.cpp:253-341), and that format is refused above */
				CCriticalErrorHandler::Inst().HandleCriticalError("Error: stage 1 on the device got a part of ReadType::na");
			const bool long_read = read_type == ReadType::long_read;
			st_long_parts += long_read ? 1 : 0;
			KmcSplitResult r;
			const auto t0 = std::chrono::steady_clock::now();
			const int rc = engine->split_part(part, size, long_read, r);
			st_engine_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
			if (rc == KMC_SPLIT_UNCOVERED) {
				++st_uncovered_parts;
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
				if (fallback_on) {
					to_reference(part, size, read_type);
					continue;
				}
#endif
				CCriticalErrorHandler::Inst().HandleCriticalError(
				    "Error: stage 1 on the device: a part of the input is malformed FASTA / FASTQ text (a blank line, a quality string of another length than its "
				    "sequence, a control character or a lone carriage return). The device splitter does not guess what such text means; "
				    "run this input with kmc_hip (the reference's stage 1, stage 2 on the device).");
			}
			if (rc != 0) {
				std::ostringstream ostr;
				ostr << "Error: stage-1 split engine failed (code " << rc << "): " << engine->last_error();
				CCriticalErrorHandler::Inst().HandleCriticalError(ostr.str());
			}
			++st_parts;
			pmm_fastq->free(part);
			n_reads += r.n_reads;
			for (uint32 b = 0; b < (uint32)bins.size(); ++b)
				if (r.bin_bytes[b])
					append(b, r.recs + r.bin_off[b], r.bin_bytes[b], r.bin_kmers[b], r.bin_superkmers[b], r.bin_plus_x[b]);
		}
		push_all();
		bpq->mark_completed();
		engine.reset();
		KmcTimeline::mark_first_last(nullptr, "splitter: last worker done");
		if (getenv("KMC_HIP_VERBOSE"))
			fprintf(stderr, "[kmc_hip stage 1] worker: %llu parts through the engine (%.3f s inside; %llu of them long-read parts), %llu uncovered parts, "
			                "%llu bin pieces (%llu cut record by record), %llu buffers / %.1f MB pushed\n",
			        (unsigned long long)st_parts, st_engine_ns * 1e-9, (unsigned long long)st_long_parts, (unsigned long long)st_uncovered_parts,
			        (unsigned long long)st_pieces,
			        (unsigned long long)st_cut_pieces, (unsigned long long)st_pushes, st_bytes / 1e6);
	}

	void GetTotal(uint64 &_n_reads)
	{
#ifdef KMC_HIP_S1_REFERENCE_FALLBACK
		if (ref) {
			ref->GetTotal(_n_reads);
			return;
		}
#endif
		_n_reads = n_reads;
	}
	~CWSplitter() {}
};

#endif
