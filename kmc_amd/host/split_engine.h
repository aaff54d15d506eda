/*
 * kmc_amd/host/split_engine.h — the per-part compute engine the stage-1 worker (kb_splitter_plugin.h) drives.
 *
 * One engine instance per splitter thread. split_part() does for one part of input text what CSplitter::ProcessReads does with its n_bins
 * CKmerBinCollectors (splitter.cpp:557-672, kb_collector.cpp:34-106) up to, but not including, the bin-part buffers: it returns the bin
 * records of the part grouped by bin, with the three sums the collector keeps per buffer. The worker copies them into pmm_bins buffers
 * and pushes those to the storer exactly as the collectors would.
 *   - oracle engine: oracle/stage1_oracle.c (TEST ONLY: oracle/oracle_engine_s1.h -> oracle/_ref/kmc_oracle_s1; pins the worker's protocol
 *     and the oracle's parser and k+x-mer bookkeeping to the reference: the database must be byte-identical)
 *   - emulated engine: the stage-1 kernels under the CPU emulation (TEST ONLY: tests/hipemu/emu_split_engine.cpp -> oracle/_ref/kmc_emu_s1)
 *   - HIP engine   : kmc_hip_split_part of include/kmc_hip.h through hip_split_loader.cpp -> oracle/_ref/kmc_hip_s1 (proven on the
 *     CPU over an emulated HIP runtime; its first run on a GPU is the round-end bench / xfail-marked test of round 2)
 */
#ifndef KMC_AMD_SPLIT_ENGINE_H
#define KMC_AMD_SPLIT_ENGINE_H

#include <stdint.h>
#include <string>

struct KmcSplitParams {
	uint32_t kmer_len, signature_len, n_bins, max_x;
	int both_strands;
	int file_type;             /* 0 = FASTA (one line per sequence), 1 = FASTQ */
	uint64_t line_cap;         /* mem_part_pmm_reads: longer lines are cut into pieces overlapping by kmer_len - 1 symbols (splitter.cpp:141-145) */
	const int32_t *sig_to_bin; /* CSignatureMapper's map, 4^signature_len + 1 entries (s_mapper.h:232) */
};

/* valid until the next split_part() on the same engine */
struct KmcSplitResult {
	const uint8_t *recs;       /* the records of all bins; bin b's are recs[bin_off[b] .. bin_off[b] + bin_bytes[b]) (a device engine aligns bins) */
	const uint64_t *bin_off, *bin_bytes; /* n_bins each */
	const uint64_t *bin_kmers, *bin_superkmers, *bin_plus_x; /* n_bins each: n_recs, n_super_kmers, n_plus_x_recs of the collector */
	uint64_t n_reads;          /* records whose title line the part holds (CSplitter::n_reads) */
};

/* split_part() returns 0, a negative error code, or KMC_SPLIT_UNCOVERED: the part is MALFORMED text the engine does not reproduce CSplitter::GetSeq
 * on (blank lines, quality of another length than its sequence, control characters ...): nothing was produced, and the worker stops the run (a build
 * with -DKMC_HIP_S1_REFERENCE_FALLBACK and $KMC_HIP_S1_FALLBACK=1 gives the part to the reference splitter instead).
 * long_read: the reader labelled the part ReadType::long_read (queues.h:40) — an optional title, then symbols only (GetSeqLongRead, splitter.cpp:70-86). */
enum { KMC_SPLIT_UNCOVERED = 1 };

struct KmcSplitEngine {
	virtual ~KmcSplitEngine() {}
	virtual int split_part(const uint8_t *text, uint64_t size, bool long_read, KmcSplitResult &out) = 0;
	virtual std::string last_error() = 0;
};

/* Provided by exactly one engine implementation linked into the binary. */
KmcSplitEngine *kmc_make_split_engine(const KmcSplitParams &params, int worker_idx, int n_workers);

#endif
