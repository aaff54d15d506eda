/*
 * oracle/oracle_engine_s1.h — TEST INFRASTRUCTURE, not product: the stage-1 CPU restatement (oracle/stage1_oracle.c) wrapped as a
 * KmcSplitEngine, so that the reference pipeline + the splitter worker of kmc_amd/host/kb_splitter_plugin.h can be run with the oracle as
 * the per-part engine (`oracle/_ref/kmc_oracle_s1`, oracle/Makefile). That build pins three things to the reference at once, because its
 * database must equal `kmc`'s byte for byte (tests/test_stage1_plugin.py): the worker's protocol towards the storer and the bin descriptors,
 * the oracle's text parser (oracle_s1_parse_part) and its k+x-mer bookkeeping (oracle_s1_kxmer_recs) — the reference's stage 2 sizes and
 * fills its arrays with those sums. Nothing in the product includes this file.
 */
#ifndef KMC_ORACLE_ENGINE_S1_H
#define KMC_ORACLE_ENGINE_S1_H

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "split_engine.h"

extern "C" {
typedef struct {
	uint32_t pos, len, signature;
} oracle_s1_superkmer;
int oracle_s1_norm(uint32_t len, uint32_t *norm);
uint64_t oracle_s1_split(const int8_t *seq, uint32_t seq_size, uint32_t kmer_len, uint32_t signature_len, const uint32_t *norm, oracle_s1_superkmer *out, uint64_t cap);
uint32_t oracle_s1_pack(const int8_t *seq, uint32_t n, uint32_t kmer_len, uint8_t *dst);
uint32_t oracle_s1_kxmer_recs(const int8_t *seq, uint32_t n, uint32_t kmer_len, uint32_t max_x, int both_strands);
int64_t oracle_s1_parse_part(const uint8_t *part, uint64_t part_size, int file_type, uint32_t kmer_len, uint64_t line_cap, int8_t *codes_out, uint64_t *seq_off,
                             uint64_t seq_cap, uint64_t *n_reads);
int64_t oracle_s1_parse_long_read_part(const uint8_t *part, uint64_t part_size, int file_type, uint32_t kmer_len, uint64_t line_cap, int8_t *codes_out,
                                       uint64_t *seq_off, uint64_t seq_cap, uint64_t *n_reads);
}

struct KmcOracleSplitEngine : KmcSplitEngine {
	KmcSplitParams P;
	std::string err;
	std::vector<uint32_t> norm;
	std::vector<int8_t> codes;
	std::vector<uint64_t> seq_off, bin_off, bytes, kmers, supers, plus_x, fill;
	std::vector<oracle_s1_superkmer> sk;
	std::vector<uint32_t> sk_seq;
	std::vector<uint8_t> recs;

	explicit KmcOracleSplitEngine(const KmcSplitParams &p) : P(p)
	{
		norm.resize((size_t)1 << (2 * P.signature_len));
		oracle_s1_norm(P.signature_len, norm.data());
	}
	std::string last_error() override { return err; }
	int split_part(const uint8_t *text, uint64_t size, bool long_read, KmcSplitResult &out) override
	{
		const uint32_t nb = P.n_bins;
		codes.resize(size + (size / (P.line_cap - P.kmer_len + 1) + 2) * P.kmer_len + 16); /* the pieces of an over-long line overlap by k - 1 */
		seq_off.resize(size / 2 + 16);
		uint64_t n_reads = 0;
		const int64_t n_seq = long_read ? oracle_s1_parse_long_read_part(text, size, P.file_type, P.kmer_len, P.line_cap, codes.data(), seq_off.data(), seq_off.size() - 1, &n_reads)
		                                : oracle_s1_parse_part(text, size, P.file_type, P.kmer_len, P.line_cap, codes.data(), seq_off.data(), seq_off.size() - 1, &n_reads);
		if (n_seq < 0) {
			err = "oracle_s1_parse_part: too many sequences";
			return -1;
		}
		/* pass 1: the super-k-mers of every sequence, and the byte count of each bin */
		sk.clear();
		sk_seq.clear();
		bin_off.assign(nb + 1, 0);
		kmers.assign(nb, 0);
		supers.assign(nb, 0);
		plus_x.assign(nb, 0);
		std::vector<oracle_s1_superkmer> tmp(1024);
		for (int64_t s = 0; s < n_seq; ++s) {
			const int8_t *q = codes.data() + seq_off[s];
			const uint32_t qn = (uint32_t)(seq_off[s + 1] - seq_off[s]);
			if (tmp.size() < qn + 8u)
				tmp.resize(qn + 8u);
			const uint64_t m = oracle_s1_split(q, qn, P.kmer_len, P.signature_len, norm.data(), tmp.data(), tmp.size());
			for (uint64_t i = 0; i < m; ++i) {
				const int32_t b = P.sig_to_bin[tmp[i].signature];
				if (b < 0 || (uint32_t)b >= nb) {
					err = "signature without a bin";
					return -2;
				}
				sk.push_back(tmp[i]);
				sk_seq.push_back((uint32_t)s);
				bin_off[b + 1] += 1 + (tmp[i].len + 3) / 4;
				kmers[b] += tmp[i].len - P.kmer_len + 1;
				supers[b] += 1;
				plus_x[b] += oracle_s1_kxmer_recs(q + tmp[i].pos, tmp[i].len, P.kmer_len, P.max_x, P.both_strands);
			}
		}
		for (uint32_t b = 0; b < nb; ++b)
			bin_off[b + 1] += bin_off[b];
		/* pass 2: the records, bin after bin, read order inside a bin */
		recs.resize(bin_off[nb] + 8);
		fill.assign(bin_off.begin(), bin_off.end() - 1);
		for (size_t i = 0; i < sk.size(); ++i) {
			const int32_t b = P.sig_to_bin[sk[i].signature];
			fill[b] += oracle_s1_pack(codes.data() + seq_off[sk_seq[i]] + sk[i].pos, sk[i].len, P.kmer_len, recs.data() + fill[b]);
		}
		out.recs = recs.data();
		out.bin_off = bin_off.data();
		bytes.resize(nb);
		for (uint32_t b = 0; b < nb; ++b)
			bytes[b] = bin_off[b + 1] - bin_off[b];
		out.bin_bytes = bytes.data();
		out.bin_kmers = kmers.data();
		out.bin_superkmers = supers.data();
		out.bin_plus_x = plus_x.data();
		out.n_reads = n_reads;
		return 0;
	}
};

KmcSplitEngine *kmc_make_split_engine(const KmcSplitParams &params, int, int) { return new KmcOracleSplitEngine(params); }

#endif
