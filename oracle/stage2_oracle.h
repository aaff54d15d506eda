/*
 * oracle/stage2_oracle.h — TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * Plain-C, single-threaded CPU restatement of KMC 3.2.4 stage 2 for ONE bin:
 *   super-k-mer bytes -> canonical k-mer records -> ascending sort -> (suffix,count) bytes
 *   + prefix LUT + 4 tallies.
 * It restates WHAT the reference computes (file:line cited at each function in the .c),
 * using the plain-k-mer formulation (the reference's own max_x==0 path,
 * kb_sorter.h:299-362 + :1128-1281), which SURVEY.md §8a "Equivalence note" shows is
 * output-identical to the k+x-mer path for every k.
 *
 * Parity pin: oracle/_ref/kmc_oracle (the reference pipeline with this oracle plugged in as
 * the stage-2 sorter) must write .kmc_pre/.kmc_suf byte-identical to oracle/_ref/kmc -sr1;
 * tests/test_oracle_vs_reference.py checks that here, and tests/golden/ holds per-bin
 * fixtures dumped from those runs for the GPU box.
 */
#ifndef KMC_STAGE2_ORACLE_H
#define KMC_STAGE2_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_WORDS 8 /* MAX_K 256 -> KMER_WORDS 8, defs.h:44,84 */

typedef struct {
	uint32_t kmer_len;       /* k, 1..256 */
	uint32_t both_strands;   /* 1: canonical k-mers (default), 0: -b */
	uint32_t cutoff_min;     /* -ci */
	uint32_t without_output; /* 1: tallies only */
	uint64_t cutoff_max;     /* -cx (the sorter compares against (uint32)cutoff_max, kb_sorter.h:186) */
	uint64_t counter_max;    /* -cs */
	uint32_t lut_prefix_len; /* p; 0 for KFF */
	uint32_t output_type;    /* 0 = KMC, 1 = KFF (count bytes big-endian, no LUT) */
} oracle_params;

/* words per record: SIZE = ceil(k/32)  (kmc_runner.cpp:57 dispatch) */
uint32_t oracle_words(uint32_t kmer_len);
/* counter bytes: defs.h:154-159 */
uint32_t oracle_counter_size(uint64_t cutoff_max, uint64_t counter_max);
/* stored record bytes = suffix bytes + counter bytes (kb_sorter.h:1132-1142) */
uint32_t oracle_out_rec_bytes(const oracle_params *p);

/* Count super-k-mers / k-mers in a bin image; returns 0, or -1 if the byte stream is ragged. */
int oracle_scan(uint32_t kmer_len, const uint8_t *data, uint64_t size, uint64_t *n_super, uint64_t *n_kmers);

/* Expand: writes n_kmers records of `words` uint64 (word 0 least significant). */
int oracle_expand(const oracle_params *p, const uint8_t *data, uint64_t size, uint64_t *recs, uint64_t cap_recs,
                  uint64_t *n_out);

/* Ascending sort of `n` records of `words` words (compare from the top word down, kmer.h:271-278). */
void oracle_sort(uint64_t *recs, uint64_t n, uint32_t words);

/* Compact a sorted record array. lut has 4^p entries (zero-filled here); stats = {n_unique, n_cutoff_min,
 * n_cutoff_max, n_total}. Returns 0, or -2 when out_capacity is too small. */
int oracle_compact(const oracle_params *p, const uint64_t *sorted, uint64_t n, uint8_t *out, uint64_t out_capacity,
                   uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4]);

/* Whole bin. `out`/`lut` may alias `data` (the reference arena does, queues.h:1352-1369): all input is
 * consumed before the first output byte is written. */
int oracle_process_bin(const oracle_params *p, const uint8_t *data, uint64_t size, uint64_t n_rec, uint8_t *out,
                       uint64_t out_capacity, uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4]);

#ifdef __cplusplus
}
#endif
#endif
