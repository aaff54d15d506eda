/*
 * oracle/stage2_oracle.c — TEST INFRASTRUCTURE ONLY. See stage2_oracle.h.
 *
 * CPU restatement of KMC 3.2.4 stage 2 (one bin). Every function cites the reference
 * file:line (relative to /root/reference) whose behaviour it restates. Nothing here is
 * derived from the product's HIP code, and the product never calls into this file.
 */
#include "stage2_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small helpers */

uint32_t oracle_words(uint32_t kmer_len) { return (kmer_len + 31) / 32; }

/* defs.h:121 BYTE_LOG, defs.h:154-159 calc_counter_size */
static uint32_t byte_log(uint64_t x)
{
	return x < (1u << 8) ? 1 : x < (1u << 16) ? 2 : x < (1u << 24) ? 3 : 4;
}
uint32_t oracle_counter_size(uint64_t cutoff_max, uint64_t counter_max)
{
	if (counter_max == 1)
		return 0;
	uint32_t a = byte_log(cutoff_max), b = byte_log(counter_max);
	return a < b ? a : b;
}

/* kb_sorter.h:1132-1135: suffix bytes = (k-p)/4, or ceil(k/4) when p == 0 */
static uint32_t suffix_bytes(const oracle_params *p)
{
	uint32_t sym = p->kmer_len - p->lut_prefix_len;
	return p->lut_prefix_len ? sym / 4 : (sym + 3) / 4;
}
uint32_t oracle_out_rec_bytes(const oracle_params *p)
{
	return suffix_bytes(p) + oracle_counter_size(p->cutoff_max, p->counter_max);
}

/* One super-k-mer on disk: [1 B: e = extra symbols][ceil((k+e)/4) B: k+e symbols, 2 bits each,
 * first symbol in the two top bits of the first byte] — written at kb_collector.cpp:57-71. */
static uint64_t superkmer_bytes(uint32_t k, uint32_t e) { return 1 + (uint64_t)(k + e + 3) / 4; }

static uint32_t symbol_at(const uint8_t *seq /* first packed byte */, uint32_t t)
{
	return (seq[t >> 2] >> (6 - 2 * (t & 3))) & 3; /* kb_sorter.h:239-249 GetNextSymb */
}

int oracle_scan(uint32_t k, const uint8_t *data, uint64_t size, uint64_t *n_super, uint64_t *n_kmers)
{
	uint64_t pos = 0, ns = 0, nk = 0;
	while (pos < size) {
		uint32_t e = data[pos];
		uint64_t len = superkmer_bytes(k, e);
		if (pos + len > size)
			return -1;
		pos += len;
		++ns;
		nk += (uint64_t)e + 1;
	}
	if (n_super)
		*n_super = ns;
	if (n_kmers)
		*n_kmers = nk;
	return 0;
}

/* ------------------------------------------------------------------ multiword k-mer ops
 * Record layout: kmer.h:22-67 — uint64 data[SIZE], data[0] least significant, k-mer right
 * aligned in the low 2k bits, first base most significant. */

static void kw_shl2_insert(uint64_t *x, uint32_t words, uint64_t sym) /* kmer.h SHL_insert_2bits */
{
	for (uint32_t i = words - 1; i > 0; --i)
		x[i] = (x[i] << 2) | (x[i - 1] >> 62);
	x[0] = (x[0] << 2) | sym;
}
static void kw_shr2_insert(uint64_t *x, uint32_t words, uint64_t sym, uint32_t bitpos) /* SHR_insert_2bits */
{
	for (uint32_t i = 0; i + 1 < words; ++i)
		x[i] = (x[i] >> 2) | (x[i + 1] << 62);
	x[words - 1] >>= 2;
	x[bitpos >> 6] |= sym << (bitpos & 63);
}
static void kw_mask(uint64_t *x, uint32_t words, uint32_t nbits) /* keep the low nbits (set_n_1 + mask) */
{
	for (uint32_t i = 0; i < words; ++i) {
		uint32_t lo = 64 * i;
		if (nbits <= lo)
			x[i] = 0;
		else if (nbits - lo < 64)
			x[i] &= (1ull << (nbits - lo)) - 1;
	}
}
static int kw_less(const uint64_t *a, const uint64_t *b, uint32_t words) /* kmer.h:271-278 */
{
	for (int32_t i = (int32_t)words - 1; i >= 0; --i) {
		if (a[i] < b[i])
			return 1;
		if (a[i] > b[i])
			return 0;
	}
	return 0;
}
static int kw_equal(const uint64_t *a, const uint64_t *b, uint32_t words)
{
	for (uint32_t i = 0; i < words; ++i)
		if (a[i] != b[i])
			return 0;
	return 1;
}
static uint8_t kw_get_byte(const uint64_t *x, uint32_t p) /* kmer.h:242-245 */
{
	return (uint8_t)(x[p >> 3] >> ((p & 7) << 3));
}
static uint64_t kw_remove_suffix(const uint64_t *x, uint32_t words, uint32_t n) /* kmer.h:294-303 */
{
	uint32_t p = n >> 6, r = n & 63;
	if (p == words - 1 || r == 0)
		return x[p] >> r;
	return (x[p + 1] << (64 - r)) | (x[p] >> r);
}

/* ------------------------------------------------------------------ expand
 * Semantics of ExpandKmersBoth / ExpandKmersAll (kb_sorter.h:299-362, :251-298): for every
 * window of k symbols of every super-k-mer emit the k-mer, or min(k-mer, reverse complement)
 * when both_strands (ties: equal values, kb_sorter.h:340). */
int oracle_expand(const oracle_params *p, const uint8_t *data, uint64_t size, uint64_t *recs, uint64_t cap_recs,
                  uint64_t *n_out)
{
	const uint32_t k = p->kmer_len, words = oracle_words(k);
	uint64_t pos = 0, out = 0;
	uint64_t fwd[ORACLE_MAX_WORDS], rev[ORACLE_MAX_WORDS];

	while (pos < size) {
		uint32_t e = data[pos];
		uint64_t len = superkmer_bytes(k, e);
		if (pos + len > size)
			return -1;
		const uint8_t *seq = data + pos + 1;
		memset(fwd, 0, sizeof fwd);
		memset(rev, 0, sizeof rev);
		for (uint32_t t = 0; t < k + e; ++t) {
			uint64_t s = symbol_at(seq, t);
			kw_shl2_insert(fwd, words, s);
			kw_mask(fwd, words, 2 * k);
			kw_shr2_insert(rev, words, 3 - s, 2 * k - 2);
			if (t + 1 >= k) {
				if (out >= cap_recs)
					return -2;
				const uint64_t *src = (p->both_strands && !kw_less(fwd, rev, words)) ? rev : fwd;
				memcpy(recs + out * words, src, words * sizeof(uint64_t));
				++out;
			}
		}
		pos += len;
	}
	if (n_out)
		*n_out = out;
	return 0;
}

/* ------------------------------------------------------------------ sort
 * Contract of SortFunction (raduls.h:19-20, kb_sorter.h:757-780): ascending by the record's
 * unsigned value. The reference's algorithm (MSD radix, raduls_impl.h:546-754) is not
 * observable in the result because equal keys are bit-identical records. */
static _Thread_local uint32_t g_cmp_words; /* per thread: several workers of the oracle-engine binaries sort at once */ /* qsort has no context argument */
static int cmp_recs(const void *a, const void *b)
{
	const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
	for (int32_t i = (int32_t)g_cmp_words - 1; i >= 0; --i) {
		if (x[i] < y[i])
			return -1;
		if (x[i] > y[i])
			return 1;
	}
	return 0;
}
/* stable byte-wise LSD counting sort, used above 1M records so tests stay quick; below that, qsort */
static void lsd_sort(uint64_t *recs, uint64_t n, uint32_t words)
{
	uint64_t *tmp = (uint64_t *)malloc(n * words * sizeof(uint64_t));
	uint64_t *src = recs, *dst = tmp;
	for (uint32_t byte = 0; byte < 8 * words; ++byte) {
		uint64_t cnt[256] = {0};
		uint32_t w = byte >> 3, sh = (byte & 7) << 3;
		for (uint64_t i = 0; i < n; ++i)
			++cnt[(src[i * words + w] >> sh) & 0xFF];
		if (cnt[(src[w] >> sh) & 0xFF] == n)
			continue; /* all records share this byte */
		uint64_t sum = 0;
		for (uint32_t d = 0; d < 256; ++d) {
			uint64_t c = cnt[d];
			cnt[d] = sum;
			sum += c;
		}
		for (uint64_t i = 0; i < n; ++i) {
			uint64_t o = cnt[(src[i * words + w] >> sh) & 0xFF]++;
			memcpy(dst + o * words, src + i * words, words * sizeof(uint64_t));
		}
		uint64_t *t = src;
		src = dst;
		dst = t;
	}
	if (src != recs)
		memcpy(recs, src, n * words * sizeof(uint64_t));
	free(tmp);
}
void oracle_sort(uint64_t *recs, uint64_t n, uint32_t words)
{
	if (n < 2)
		return;
	if (n > (1u << 20)) {
		lsd_sort(recs, n, words);
		return;
	}
	g_cmp_words = words;
	qsort(recs, n, words * sizeof(uint64_t), cmp_recs);
}

/* ------------------------------------------------------------------ compact
 * CompactKmers, kb_sorter.h:1128-1281 (and identically the emit blocks of CompactKxmers,
 * :1020-1103): per run of equal k-mers: ++n_unique; count < cutoff_min -> ++n_cutoff_min;
 * count > cutoff_max -> ++n_cutoff_max; else clamp to counter_max and emit
 *   suffix bytes get_byte(kmer_bytes-1 .. 0)  +  counter bytes (LE for KMC :1200, BE for KFF :1210)
 * and lut[kmer >> 2(k-p)]++ (KMC only, :1203). n_total = n_rec (:1166). count is uint32 (:1153). */
static void emit(const oracle_params *p, const uint64_t *kmer, uint32_t words, uint32_t count, uint32_t sbytes,
                 uint32_t cbytes, uint8_t *out, uint64_t *out_pos, uint64_t *lut)
{
	for (int32_t j = (int32_t)sbytes - 1; j >= 0; --j)
		out[(*out_pos)++] = kw_get_byte(kmer, (uint32_t)j);
	if (p->output_type == 0) {
		for (uint32_t j = 0; j < cbytes; ++j)
			out[(*out_pos)++] = (uint8_t)(count >> (8 * j));
		if (p->lut_prefix_len)
			lut[kw_remove_suffix(kmer, words, 2 * (p->kmer_len - p->lut_prefix_len))]++;
	} else {
		for (int32_t j = (int32_t)cbytes - 1; j >= 0; --j)
			out[(*out_pos)++] = (uint8_t)(count >> (8 * j));
	}
}

int oracle_compact(const oracle_params *p, const uint64_t *sorted, uint64_t n, uint8_t *out, uint64_t out_capacity,
                   uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4])
{
	const uint32_t words = oracle_words(p->kmer_len);
	const uint32_t sbytes = suffix_bytes(p);
	const uint32_t cbytes = oracle_counter_size(p->cutoff_max, p->counter_max);
	const uint32_t cutoff_max = (uint32_t)p->cutoff_max, counter_max = (uint32_t)p->counter_max; /* :186-187 */
	const uint64_t lut_recs = p->lut_prefix_len ? 1ull << (2 * p->lut_prefix_len) : 0;
	uint64_t out_pos = 0, n_unique = 0, n_min = 0, n_max = 0;

	if (lut)
		memset(lut, 0, lut_recs * sizeof(uint64_t));
	uint64_t i = 0;
	while (i < n) {
		uint64_t j = i + 1;
		while (j < n && kw_equal(sorted + i * words, sorted + j * words, words))
			++j;
		uint32_t count = (uint32_t)(j - i);
		++n_unique;
		if (count < p->cutoff_min)
			++n_min;
		else if (count > cutoff_max)
			++n_max;
		else {
			if (count > counter_max)
				count = counter_max;
			if (!p->without_output) {
				if (out_pos + sbytes + cbytes > out_capacity)
					return -2;
				emit(p, sorted + i * words, words, count, sbytes, cbytes, out, &out_pos, lut);
			}
		}
		i = j;
	}
	if (out_bytes)
		*out_bytes = out_pos;
	stats[0] = n_unique;
	stats[1] = n_min;
	stats[2] = n_max;
	stats[3] = n;
	return 0;
}

int oracle_process_bin(const oracle_params *p, const uint8_t *data, uint64_t size, uint64_t n_rec, uint8_t *out,
                       uint64_t out_capacity, uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4])
{
	const uint32_t words = oracle_words(p->kmer_len);
	uint64_t n_super = 0, n_kmers = 0;
	if (oracle_scan(p->kmer_len, data, size, &n_super, &n_kmers) != 0)
		return -1;
	if (n_kmers != n_rec)
		return -3; /* CBinDesc n_rec must agree with the byte stream (kb_collector.cpp:74) */
	uint64_t *recs = (uint64_t *)malloc((n_kmers ? n_kmers : 1) * words * sizeof(uint64_t));
	if (!recs)
		return -4;
	uint64_t n = 0;
	int rc = oracle_expand(p, data, size, recs, n_kmers, &n);
	if (rc == 0) {
		oracle_sort(recs, n, words);
		rc = oracle_compact(p, recs, n, out, out_capacity, out_bytes, lut, stats);
	}
	free(recs);
	return rc;
}
