/*
 * oracle/stage1_oracle.c — TEST INFRASTRUCTURE, not product: a plain C restatement of the part of KMC's STAGE 1 that a GPU would
 * take over: minimizer signatures, super-k-mer cutting and the bin-buffer record format. It exists so that
 * stage-1 kernels can be checked bit for bit the way the stage-2 kernels are; it is pinned to the real reference by
 * tests/test_stage1_oracle.py (every bin image of a reference run is reproduced from the reads). The signature -> bin map
 * (CSignatureMapper, s_mapper.h:143-233: built on the host from a sample of the input) is NOT restated: it stays with the reference.
 *
 *   oracle_s1_norm      kmc_api/mmer.h:39-95   (is_allowed, get_rev, init_norm)
 *   oracle_s1_split     kmc_core/splitter.cpp:557-672 (CSplitter::ProcessReads, one sequence), CMmer::insert/set/compare mmer.h:122-190
 *   oracle_s1_pack      kmc_core/kb_collector.cpp:57-71 (CKmerBinCollector::PutExtendedKmer: the record written to the bin)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
	uint32_t pos;       /* first symbol of the super-k-mer inside the sequence */
	uint32_t len;       /* symbols: kmer_len + extra, extra <= 255 */
	uint32_t signature; /* normalised minimizer value; 4^sig_len = "special" (no allowed m-mer in the window) */
} oracle_s1_superkmer;

static int s1_is_allowed(uint32_t mmer, uint32_t len) /* mmer.h:39-64 */
{
	if ((mmer & 0x3f) == 0x3f) /* TTT suffix */
		return 0;
	if ((mmer & 0x3f) == 0x3b) /* TGT suffix */
		return 0;
	if ((mmer & 0x3c) == 0x3c) /* TG* suffix */
		return 0;
	for (uint32_t j = 0; j < len - 3; ++j) {
		if ((mmer & 0xf) == 0) /* AA inside */
			return 0;
		mmer >>= 2;
	}
	if (mmer == 0) /* AAA prefix */
		return 0;
	if (mmer == 0x04) /* ACA prefix */
		return 0;
	if ((mmer & 0xf) == 0) /* *AA prefix */
		return 0;
	return 1;
}

static uint32_t s1_rev(uint32_t mmer, uint32_t len) /* mmer.h:69-80: reverse complement of an m-mer */
{
	uint32_t rev = 0, shift = len * 2 - 2;
	for (uint32_t i = 0; i < len; ++i) {
		rev += (3 - (mmer & 3)) << shift;
		mmer >>= 2;
		shift -= 2;
	}
	return rev;
}

/* norm[m] = min over both strands of (m if allowed else 4^len)   (mmer.h:82-92) */
int oracle_s1_norm(uint32_t len, uint32_t *norm)
{
	if (len < 5 || len > 11)
		return -1;
	const uint32_t special = 1u << (len * 2);
	for (uint32_t i = 0; i < special; ++i) {
		const uint32_t rev = s1_rev(i, len);
		const uint32_t a = s1_is_allowed(i, len) ? i : special, b = s1_is_allowed(rev, len) ? rev : special;
		norm[i] = a < b ? a : b;
	}
	return 0;
}
int oracle_s1_is_allowed(uint32_t mmer, uint32_t len) { return s1_is_allowed(mmer, len); }

typedef struct {
	uint32_t str, mask, cur, len;
	const uint32_t *norm;
} s1_mmer; /* CMmer, mmer.h:25-36 */
static void mm_insert(s1_mmer *m, int8_t symb) /* mmer.h:122-129 */
{
	m->str = ((m->str << 2) + (uint32_t)symb) & m->mask;
	m->cur = m->norm[m->str];
}
static void mm_insert_seq(s1_mmer *m, const int8_t *seq) /* mmer.h:181-198: the first len symbols at seq */
{
	m->str = 0;
	for (uint32_t i = 0; i < m->len; ++i)
		m->str = (m->str << 2) + (uint32_t)seq[i];
	m->str &= m->mask;
	m->cur = m->norm[m->str];
}

/* One sequence (codes 0..3, negative = N) -> its super-k-mers in emission order. Returns their number (may exceed cap: then only the
 * first cap were stored). Control flow follows splitter.cpp:573-667 line by line; `emit` = bins[bin_no]->PutExtendedKmer(seq + at, len). */
uint64_t oracle_s1_split(const int8_t *seq, uint32_t seq_size, uint32_t kmer_len, uint32_t signature_len, const uint32_t *norm,
                         oracle_s1_superkmer *out, uint64_t cap)
{
	uint64_t n = 0;
#define EMIT(at, l)                                                                                                    \
	do {                                                                                                               \
		if (n < cap) {                                                                                                 \
			out[n].pos = (uint32_t)(at);                                                                               \
			out[n].len = (l);                                                                                          \
			out[n].signature = cur.cur;                                                                                \
		}                                                                                                              \
		++n;                                                                                                           \
	} while (0)
	s1_mmer cur = {0, (1u << (signature_len * 2)) - 1, 0, signature_len, norm}, end = cur;
	uint32_t signature_start_pos = 0, i = 0, len = 0;
	while (i + kmer_len - 1 < seq_size) {
		int contains_N = 0;
		for (uint32_t j = 0; j < signature_len; ++j, ++i)
			if (seq[i] < 0) {
				contains_N = 1;
				break;
			}
		if (contains_N) {
			++i;
			continue;
		}
		len = signature_len;
		signature_start_pos = i - signature_len;
		mm_insert_seq(&cur, seq + signature_start_pos);
		end = cur;
		for (; i < seq_size; ++i) {
			if (seq[i] < 0) {
				if (len >= kmer_len)
					EMIT(i - len, len);
				len = 0;
				++i;
				break;
			}
			mm_insert(&end, seq[i]);
			if (end.cur < cur.cur) {
				if (len >= kmer_len) {
					EMIT(i - len, len);
					len = kmer_len - 1;
				}
				cur = end;
				signature_start_pos = i - signature_len + 1;
			} else if (end.cur == cur.cur) {
				cur = end;
				signature_start_pos = i - signature_len + 1;
			} else if (signature_start_pos + kmer_len - 1 < i) {
				EMIT(i - len, len);
				len = kmer_len - 1;
				++signature_start_pos;
				mm_insert_seq(&end, seq + signature_start_pos);
				cur = end;
				for (uint32_t j = signature_start_pos + signature_len; j <= i; ++j) {
					mm_insert(&end, seq[j]);
					if (end.cur <= cur.cur) {
						cur = end;
						signature_start_pos = j - signature_len + 1;
					}
				}
			}
			++len;
			if (len == kmer_len + 255) {
				EMIT(i + 1 - len, len);
				i -= kmer_len - 2;
				len = 0;
				break;
			}
		}
	}
	if (len >= kmer_len)
		EMIT(i - len, len);
#undef EMIT
	return n;
}

/* the bin-buffer record of one super-k-mer: [len - kmer_len][ceil(len/4) bytes, 4 symbols per byte, first symbol in bits 7:6]; returns its size */
uint32_t oracle_s1_pack(const int8_t *seq, uint32_t n, uint32_t kmer_len, uint8_t *dst)
{
	uint32_t p = 0;
	dst[p++] = (uint8_t)(n - kmer_len);
	for (uint32_t i = 0, j = 0; i < n / 4; ++i, j += 4)
		dst[p++] = (uint8_t)((seq[j] << 6) + (seq[j + 1] << 4) + (seq[j + 2] << 2) + seq[j + 3]);
	switch (n % 4) {
	case 1: dst[p++] = (uint8_t)(seq[n - 1] << 6); break;
	case 2: dst[p++] = (uint8_t)((seq[n - 2] << 6) + (seq[n - 1] << 4)); break;
	case 3: dst[p++] = (uint8_t)((seq[n - 3] << 6) + (seq[n - 2] << 4) + (seq[n - 1] << 2)); break;
	}
	return p;
}

/* Whole read set in one call (what a test wants): `seqs` = codes of all sequences back to back, seq_off[n_seq + 1]. Fills, per
 * super-k-mer in emission order, its signature, the sequence it came from, and appends its bin record to `recs` (rec_off[i] = start).
 * Returns the number of super-k-mers, or -1 if `cap_sk` / `cap_bytes` are too small. */
int64_t oracle_s1_split_all(const int8_t *seqs, const uint64_t *seq_off, uint64_t n_seq, uint32_t kmer_len, uint32_t signature_len, uint32_t *sig, uint64_t *rec_off,
                            uint8_t *recs, uint64_t cap_sk, uint64_t cap_bytes)
{
	uint32_t *norm = (uint32_t *)malloc(sizeof(uint32_t) << (2 * signature_len));
	if (!norm || oracle_s1_norm(signature_len, norm)) {
		free(norm);
		return -2;
	}
	uint64_t n = 0, bytes = 0;
	oracle_s1_superkmer *tmp = (oracle_s1_superkmer *)malloc(sizeof(oracle_s1_superkmer) * 65536);
	for (uint64_t s = 0; s < n_seq; ++s) {
		const int8_t *q = seqs + seq_off[s];
		const uint32_t qn = (uint32_t)(seq_off[s + 1] - seq_off[s]);
		const uint64_t m = oracle_s1_split(q, qn, kmer_len, signature_len, norm, tmp, 65536);
		if (m > 65536 || n + m > cap_sk) {
			free(norm);
			free(tmp);
			return -1;
		}
		for (uint64_t i = 0; i < m; ++i) {
			if (bytes + 1 + (tmp[i].len + 3) / 4 > cap_bytes) {
				free(norm);
				free(tmp);
				return -1;
			}
			sig[n] = tmp[i].signature;
			rec_off[n] = bytes;
			bytes += oracle_s1_pack(q + tmp[i].pos, tmp[i].len, kmer_len, recs + bytes);
			++n;
		}
	}
	rec_off[n] = bytes;
	free(norm);
	free(tmp);
	return (int64_t)n;
}

/* ------------------------------------------------------------------------------------------------ collector bookkeeping
 * n_plus_x_recs of one super-k-mer: how many (k+x)-mer records the reference's stage 2 will expand it into — the collector adds it up per
 * bin part (kb_collector.cpp:83-100, kb_collector.h:72-118) and stage 2 sizes its arrays with the sums. seq: the super-k-mer's n symbols. */
uint32_t oracle_s1_kxmer_recs(const int8_t *seq, uint32_t n, uint32_t kmer_len, uint32_t max_x, int both_strands)
{
	if (!max_x)
		return 0; /* kb_collector.cpp:85: plain k-mers are sorted, nothing is counted */
	if (!both_strands)
		return 1 + (n - kmer_len) / (max_x + 1); /* kb_collector.cpp:87-88 */
	/* kb_collector.h:72-118, DIVIDE_FACTOR = max_x + 1: a run of k-mers on which "k-mer < its reverse complement" (compared on their
	 * first four symbols only, one byte each) keeps one answer makes 1 + run / (max_x + 1) records; every k-mer of a tie is its own record */
	const uint32_t div = max_x + 1;
	uint8_t kmer = (uint8_t)((seq[0] << 6) + (seq[1] << 4) + (seq[2] << 2) + seq[3]);
	uint8_t rev = (uint8_t)(((3 - seq[kmer_len - 1]) << 6) + ((3 - seq[kmer_len - 2]) << 4) + ((3 - seq[kmer_len - 3]) << 2) + (3 - seq[kmer_len - 4]));
	uint32_t kmer_pos = 4, rev_pos = kmer_len, x = 0, total = 0;
	int state = kmer < rev ? 0 : (rev < kmer ? 1 : 2), ns;
	for (uint32_t i = 0; i < n - kmer_len; ++i) {
		rev = (uint8_t)(rev >> 2);
		rev = (uint8_t)(rev + ((3 - seq[rev_pos++]) << 6));
		kmer = (uint8_t)(kmer << 2);
		kmer = (uint8_t)(kmer + seq[kmer_pos++]);
		ns = kmer < rev ? 0 : (rev < kmer ? 1 : 2);
		if (ns == state) {
			if (state == 2)
				++total;
			else
				++x;
		} else {
			state = ns;
			total += 1 + x / div;
			x = 0;
		}
	}
	return total + 1 + x / div;
}

/* ------------------------------------------------------------------------------------------------ parts of input text
 * CSplitter::GetSeq for short reads (splitter.cpp:92-303): one call hands out the next sequence of a part of a FASTA (file_type 0) or FASTQ
 * (file_type 1) file as codes. `st` carries the reference's member state between calls. Returns 0 when the part is exhausted.
 * line_cap = mem_part_pmm_reads: a longer line comes in pieces that overlap by kmer_len - 1 symbols. */
typedef struct {
	const uint8_t *part;
	uint64_t part_size, part_pos, n_reads;
	uint32_t curr_read_len;
} oracle_s1_part;

static int8_t s1_code(uint8_t c) /* splitter.cpp:41-47; a function, not a lazily filled table: several splitter threads parse at once */
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return -1;
	}
}

static int s1_is_eol(uint8_t c) { return c == '\n' || c == '\r'; }

static int s1_get_seq(oracle_s1_part *st, int file_type, uint32_t kmer_len, uint64_t line_cap, int8_t *seq, uint32_t *seq_size)
{
	const uint8_t *part = st->part;
	if (st->part_pos >= st->part_size) /* :94 */
		return 0;
	uint8_t c = 0;
	uint32_t pos = 0;
	const uint8_t marker = file_type == 0 ? '>' : '@';
	if (st->curr_read_len == 0) {
		c = part[st->part_pos++]; /* title, :105-117 / :194-206 */
		if (c != marker)
			return 0;
		++st->n_reads;
		while (st->part_pos < st->part_size) {
			c = part[st->part_pos++];
			if (s1_is_eol(c))
				break;
		}
		if (st->part_pos >= st->part_size)
			return 0;
		c = part[st->part_pos++]; /* second end-of-line byte, unless the read is empty, :119-123 / :208-212 */
		if (c >= 32 || c == part[st->part_pos - 2])
			st->part_pos--;
		else if (st->part_pos >= st->part_size)
			return 0;
	}
	/* sequence, :126-132 / :215-221 (first piece) and :152-158 / :236-242 (a further piece of a long line) */
	while (st->part_pos < st->part_size && pos < line_cap) {
		c = part[st->part_pos++];
		if (s1_is_eol(c))
			break;
		seq[pos++] = s1_code(c);
	}
	if (file_type == 0) { /* FASTA: the part may end with the sequence, :134-137 / :160-163 */
		*seq_size = pos;
		if (st->part_pos >= st->part_size)
			return 1;
	} else { /* FASTQ: the quality line must follow, :222-223 / :243-244 */
		if (st->part_pos >= st->part_size)
			return 0;
		*seq_size = pos;
	}
	const int first = st->curr_read_len == 0;
	if (first)
		st->curr_read_len = pos; /* :139 / :226 */
	else
		st->curr_read_len += pos - kmer_len + 1; /* :165 / :246 */
	if (pos >= line_cap) { /* :141-145 etc.: the line goes on; the next piece starts kmer_len - 1 symbols back */
		st->part_pos -= kmer_len - 1;
		return 1;
	}
	if (file_type == 0) { /* :172-183 */
		st->curr_read_len = 0;
		if (st->part_pos >= st->part_size)
			return 1;
		const uint8_t tmp = part[st->part_pos++];
		if (!s1_is_eol(tmp))
			st->part_pos--;
		else if (st->part_pos >= st->part_size)
			return 1;
		return 1;
	}
	c = part[st->part_pos++]; /* FASTQ :254-258: second end-of-line byte of the sequence line */
	if (!s1_is_eol(c))
		st->part_pos--;
	else if (st->part_pos >= st->part_size)
		return 0;
	c = part[st->part_pos++]; /* plus line, :260-272 */
	if (st->part_pos >= st->part_size)
		return 0;
	if (c != '+')
		return 0;
	while (st->part_pos < st->part_size) {
		c = part[st->part_pos++];
		if (s1_is_eol(c))
			break;
	}
	if (st->part_pos >= st->part_size)
		return 0;
	c = part[st->part_pos++]; /* :274-278: second end-of-line byte, unless the quality is empty */
	if (c >= 32 || c == part[st->part_pos - 2])
		st->part_pos--;
	else if (st->part_pos >= st->part_size)
		return 0;
	st->part_pos += st->curr_read_len; /* quality skipped by length, :281 */
	st->curr_read_len = 0;
	if (st->part_pos >= st->part_size)
		return 0;
	c = part[st->part_pos++]; /* :287 */
	if (st->part_pos >= st->part_size) /* end of the last record, :290-291 */
		return 1;
	const uint8_t tmp = part[st->part_pos++]; /* :294-298 */
	if (!s1_is_eol(tmp))
		st->part_pos--;
	else if (st->part_pos >= st->part_size)
		return 1;
	return 1;
}

/* A part the reader labelled ReadType::long_read (queues.h:40): CSplitter::GetSeqLongRead, splitter.cpp:70-86 — a title only if the part starts
 * with the marker (one read counted; the scan stops AT its end of line, which is then handed out as a symbol: an invalid one), after that every
 * byte is a symbol, in pieces of line_cap that overlap by kmer_len - 1. Same outputs as oracle_s1_parse_part; codes_out must hold
 * part_size + (part_size / (line_cap - kmer_len + 1) + 2) * kmer_len bytes. */
int64_t oracle_s1_parse_long_read_part(const uint8_t *part, uint64_t part_size, int file_type, uint32_t kmer_len, uint64_t line_cap, int8_t *codes_out,
                                       uint64_t *seq_off, uint64_t seq_cap, uint64_t *n_reads)
{
	const uint8_t marker = file_type == 0 ? '>' : '@';
	uint64_t part_pos = 0, n = 0, at = 0;
	*n_reads = 0;
	seq_off[0] = 0;
	while (part_pos < part_size) { /* GetSeq :94 */
		uint64_t pos = 0;
		if (part_pos == 0 && part[0] == marker) { /* :74-79 */
			++*n_reads;
			while (part_pos < part_size && part[part_pos] != '\n' && part[part_pos] != '\r')
				++part_pos;
		}
		while (pos < line_cap && part_pos < part_size) /* :80-81 */
			codes_out[at + pos++] = s1_code(part[part_pos++]);
		if (n >= seq_cap)
			return -1;
		at += pos;
		seq_off[++n] = at;
		if (part_pos < part_size) { /* :83-84 */
			if (line_cap < kmer_len)
				return -1; /* would never advance */
			part_pos -= kmer_len - 1;
		}
	}
	return (int64_t)n;
}

/* All sequences of one part: codes back to back into `codes_out` (capacity >= part_size + (part_size / (line_cap - kmer_len + 1) + 2) * kmer_len: the
 * pieces of an over-long line overlap), seq_off[0 .. n_seq]; *n_reads = records whose
 * title was seen. Returns the number of sequences handed out, or -1 when seq_cap is too small. */
int64_t oracle_s1_parse_part(const uint8_t *part, uint64_t part_size, int file_type, uint32_t kmer_len, uint64_t line_cap, int8_t *codes_out, uint64_t *seq_off,
                             uint64_t seq_cap, uint64_t *n_reads)
{
	oracle_s1_part st = {part, part_size, 0, 0, 0};
	int8_t *line = (int8_t *)malloc(line_cap + 16);
	uint64_t n = 0, at = 0;
	uint32_t sz = 0;
	seq_off[0] = 0;
	while (s1_get_seq(&st, file_type, kmer_len, line_cap, line, &sz)) {
		if (n >= seq_cap) {
			free(line);
			return -1;
		}
		memcpy(codes_out + at, line, sz);
		at += sz;
		seq_off[++n] = at;
	}
	free(line);
	*n_reads = st.n_reads;
	return (int64_t)n;
}
