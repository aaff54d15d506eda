/*
 * kmc_amd/host/kb_splitter_plugin.h — the stage-1 splitter worker of kmc_core with the splitting done by an engine
 * (DESIGN.md §9): the drop-in boundary of a GPU stage 1.
 *
 * Drop-in for the reference's CWSplitter (splitter.h:100-113, splitter.cpp:814-867): the same class name, constructor, operator()(),
 * GetTotal() and destructor, so CKMC<SIZE>::ProcessStage1_impl (kmc.h:1274-1362) builds against it unchanged. Compiled in with
 * `-include kb_splitter_plugin.h` ahead of kmc_runner.cpp; the reference's own splitter.cpp is compiled with -DCWSplitter=CWSplitter_ref
 * (oracle/Makefile), which is what this header declares first: CSplitter stays the reference's (statistics, small k, histogram estimation
 * use it directly) and CWSplitter_ref runs whenever the fast path below does not cover the job.
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
 * Falls back to the reference, per job: input other than FASTA / FASTQ, homopolymer compression, histogram estimation while counting;
 * per part: ReadType::long_read (the reader could not delimit whole records, queues.h:40), and any part the engine reports as
 * KMC_SPLIT_UNCOVERED (text on which it does not reproduce CSplitter::GetSeq) — our buffers are pushed first, then a reference CSplitter of
 * this thread takes the part.
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
	std::unique_ptr<CWSplitter_ref> ref;    /* non-null: the reference worker runs the whole job */
	std::unique_ptr<CSplitter> ref_splitter; /* long-read parts of this thread */
	std::unique_ptr<KmcSplitEngine> engine;
	std::vector<BinBuf> bins;
	uint32 kmer_len, max_x, buffer_size;
	bool both_strands;
	uint64 n_reads = 0;
	/* KMC_HIP_VERBOSE=1: one line per worker on stderr when it finishes */
	uint64 st_parts = 0, st_long_parts = 0, st_uncovered_parts = 0, st_pieces = 0, st_cut_pieces = 0, st_pushes = 0, st_bytes = 0;
	long long st_engine_ns = 0;

	static bool covered(const CKMCParams &P)
	{
		return (P.file_type == InputType::FASTA || P.file_type == InputType::FASTQ) && !P.homopolymer_compressed &&
		       P.estimateHistogramCfg != KMC::EstimateHistogramCfg::ESTIMATE_AND_COUNT_KMERS;
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
		if (!covered(Params) || getenv("KMC_HIP_SPLITTER_REF")) {
			ref = std::make_unique<CWSplitter_ref>(Params, Queues);
			return;
		}
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
		if (ref) {
			(*ref)();
			KmcTimeline::mark_first_last(nullptr, "splitter: last worker done");
			return;
		}
		while (!pq->completed()) {
			uchar *part;
			uint64 size;
			ReadType read_type;
			if (!pq->pop(part, size, read_type))
				continue;
			if (read_type != ReadType::normal_read) {
				++st_long_parts;
				to_reference(part, size, read_type);
				continue;
			}
			KmcSplitResult r;
			const auto t0 = std::chrono::steady_clock::now();
			const int rc = engine->split_part(part, size, r);
			st_engine_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
			if (rc == KMC_SPLIT_UNCOVERED) {
				++st_uncovered_parts;
				to_reference(part, size, read_type);
				continue;
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
		if (ref_splitter) {
			ref_splitter->Complete();
			uint64 n = 0;
			ref_splitter->GetTotal(n);
			n_reads += n;
			ref_splitter.reset();
		}
		bpq->mark_completed();
		engine.reset();
		KmcTimeline::mark_first_last(nullptr, "splitter: last worker done");
		if (getenv("KMC_HIP_VERBOSE"))
			fprintf(stderr, "[kmc_hip stage 1] worker: %llu parts through the engine (%.3f s inside), %llu long-read parts and %llu uncovered parts to the reference splitter, "
			                "%llu bin pieces (%llu cut record by record), %llu buffers / %.1f MB pushed\n",
			        (unsigned long long)st_parts, st_engine_ns * 1e-9, (unsigned long long)st_long_parts, (unsigned long long)st_uncovered_parts,
			        (unsigned long long)st_pieces,
			        (unsigned long long)st_cut_pieces, (unsigned long long)st_pushes, st_bytes / 1e6);
	}

	void GetTotal(uint64 &_n_reads)
	{
		if (ref)
			ref->GetTotal(_n_reads);
		else
			_n_reads = n_reads;
	}
	~CWSplitter() {}
};

#endif
