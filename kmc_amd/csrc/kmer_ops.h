/*
 * kmc_amd/csrc/kmer_ops.h — per-record arithmetic shared by every kernel (and compilable on the host
 * so tests/test_oracle.py::test_kmer_ops_host_matches_oracle can check it bit for bit against the oracle without a GPU, and tests/hipemu can run the kernels on the CPU).
 *
 * Record = CKmer<SIZE> of the reference (kmc_core/kmer.h:22-67): SIZE x uint64, word 0 least
 * significant, k-mer right-aligned in the low 2k bits, first base in the most significant pair.
 * Bin image = super-k-mers [1 B e][ceil((k+e)/4) B packed, first symbol in bits 7:6] (kb_collector.cpp:57-71).
 */
#ifndef KMC_AMD_KMER_OPS_H
#define KMC_AMD_KMER_OPS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KMC_HD __host__ __device__ __forceinline__
#else
#define KMC_HD static inline
#endif

typedef unsigned long long kmc_u64;
typedef unsigned int kmc_u32;

/* reverse the order of the 32 two-bit symbols of a 64-bit word */
KMC_HD kmc_u64 kmc_rev2(kmc_u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	/* v_bfrev_b32 on each half reverses single bits, then the two bits of every symbol are swapped back — per 32-bit half: pairs never straddle the halves,
	 * and written on 64 bits the two one-bit shifts become 64-bit shift instructions */
	kmc_u32 h = __brev((kmc_u32)x), l = __brev((kmc_u32)(x >> 32));
	h = ((h >> 1) & 0x55555555u) | ((h & 0x55555555u) << 1);
	l = ((l >> 1) & 0x55555555u) | ((l & 0x55555555u) << 1);
	return ((kmc_u64)h << 32) | l;
#else
	x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
	x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
	x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
	return (x >> 32) | (x << 32);
#endif
}

/* x >>= s over W words (0 <= s < 64) */
template <int W> KMC_HD void kmc_shr(kmc_u64 (&x)[W], kmc_u32 s)
{
	if (s == 0)
		return;
#pragma unroll
	for (int i = 0; i < W - 1; ++i)
		x[i] = (x[i] >> s) | (x[i + 1] << (64 - s));
	x[W - 1] >>= s;
}

/* keep the low nbits of a W-word value */
template <int W> KMC_HD void kmc_mask_low(kmc_u64 (&x)[W], kmc_u32 nbits)
{
#pragma unroll
	for (int i = 0; i < W; ++i) {
		const kmc_u32 lo = 64u * i;
		if (nbits <= lo)
			x[i] = 0;
		else if (nbits - lo < 64)
			x[i] &= (1ull << (nbits - lo)) - 1;
	}
}

/* Forward k-mer number `off` (0..e) of the super-k-mer whose packed symbols start at `seq`.
 * Semantics of the window walk in ExpandKmersAll (kb_sorter.h:251-298): symbols off..off+k-1, MSB first. */
template <int SIZE> KMC_HD void kmc_extract_kmer(const uint8_t *seq, kmc_u32 off, kmc_u32 k, kmc_u64 (&out)[SIZE])
{
	const kmc_u32 phase = off & 3;                      /* symbols to skip inside the first byte */
	const kmc_u32 nbytes = (2 * phase + 2 * k + 7) >> 3; /* bytes that hold the window */
	const uint8_t *p = seq + (off >> 2);
	kmc_u64 acc[SIZE + 1];
#pragma unroll
	for (int i = 0; i <= SIZE; ++i)
		acc[i] = 0;
	for (kmc_u32 b = 0; b < nbytes; ++b) {
#pragma unroll
		for (int i = SIZE; i > 0; --i)
			acc[i] = (acc[i] << 8) | (acc[i - 1] >> 56);
		acc[0] = (acc[0] << 8) | p[b];
	}
	kmc_shr<SIZE + 1>(acc, 8 * nbytes - 2 * phase - 2 * k); /* drop the symbols after the window (0..6 bits) */
#pragma unroll
	for (int i = 0; i < SIZE; ++i)
		out[i] = acc[i];
	kmc_mask_low<SIZE>(out, 2 * k);                        /* drop the symbols before the window */
}

/* reverse complement of a k-mer held in the low 2k bits (rev_kmer of ExpandKmersBoth, kb_sorter.h:324-338,355) */
template <int SIZE> KMC_HD void kmc_revcomp(const kmc_u64 (&x)[SIZE], kmc_u32 k, kmc_u64 (&out)[SIZE])
{
#pragma unroll
	for (int i = 0; i < SIZE; ++i)
		out[SIZE - 1 - i] = ~kmc_rev2(x[i]); /* symbol order reversed over the whole 32*SIZE-symbol register, complemented */
	const kmc_u32 sh = 64u * SIZE - 2 * k;  /* the k-mer now sits in the TOP 2k bits: bring it down */
	const kmc_u32 ws = sh >> 6, bs = sh & 63;
	if (ws) {
#pragma unroll
		for (int i = 0; i < SIZE; ++i)
			out[i] = (i + (int)ws < SIZE) ? out[i + ws] : 0;
	}
	kmc_shr<SIZE>(out, bs);
	kmc_mask_low<SIZE>(out, 2 * k);
}

template <int SIZE> KMC_HD bool kmc_less(const kmc_u64 (&a)[SIZE], const kmc_u64 (&b)[SIZE]) /* kmer.h:271-278 */
{
#pragma unroll
	for (int i = SIZE - 1; i >= 0; --i) {
		if (a[i] < b[i])
			return true;
		if (a[i] > b[i])
			return false;
	}
	return false;
}

template <int SIZE> KMC_HD bool kmc_equal(const kmc_u64 (&a)[SIZE], const kmc_u64 (&b)[SIZE])
{
	bool eq = true;
#pragma unroll
	for (int i = 0; i < SIZE; ++i)
		eq = eq && (a[i] == b[i]);
	return eq;
}

/* canonical k-mer as the sorter sees it: min(kmer, revcomp) (kb_sorter.h:340,356) or the k-mer itself (-b) */
template <int SIZE>
KMC_HD void kmc_canonical_at(const uint8_t *seq, kmc_u32 off, kmc_u32 k, bool both_strands, kmc_u64 (&out)[SIZE])
{
	kmc_extract_kmer<SIZE>(seq, off, k, out);
	if (both_strands) {
		kmc_u64 rc[SIZE];
		kmc_revcomp<SIZE>(out, k, rc);
		if (kmc_less<SIZE>(rc, out)) {
#pragma unroll
			for (int i = 0; i < SIZE; ++i)
				out[i] = rc[i];
		}
	}
}

/* byte p of a record (kmer.h:242-245); the radix digit of pass `p` */
template <int SIZE> KMC_HD kmc_u32 kmc_get_byte(const kmc_u64 (&x)[SIZE], kmc_u32 p)
{
	return (kmc_u32)(x[p >> 3] >> ((p & 7) << 3)) & 0xFFu;
}

/* kmer >> nbits, low 64 bits (kmer.h:294-303 remove_suffix); nbits = 2(k-p) */
template <int SIZE> KMC_HD kmc_u64 kmc_remove_suffix(const kmc_u64 (&x)[SIZE], kmc_u32 nbits)
{
	const kmc_u32 w = nbits >> 6, r = nbits & 63;
	kmc_u64 v = x[w] >> r;
	if (r && (int)w + 1 < SIZE)
		v |= x[w + 1] << (64 - r);
	return v;
}

/* bytes of one stored record: suffix bytes (kb_sorter.h:1132-1135) + counter bytes (defs.h:154-159) */
KMC_HD kmc_u32 kmc_suffix_bytes(kmc_u32 k, kmc_u32 lut_prefix_len)
{
	const kmc_u32 sym = k - lut_prefix_len;
	return lut_prefix_len ? sym / 4 : (sym + 3) / 4;
}
KMC_HD kmc_u32 kmc_byte_log(kmc_u64 x) { return x < (1u << 8) ? 1 : x < (1u << 16) ? 2 : x < (1u << 24) ? 3 : 4; }
KMC_HD kmc_u32 kmc_counter_bytes(kmc_u64 cutoff_max, kmc_u64 counter_max)
{
	if (counter_max == 1)
		return 0;
	const kmc_u32 a = kmc_byte_log(cutoff_max), b = kmc_byte_log(counter_max);
	return a < b ? a : b;
}

/* write one (suffix, count) record: suffix bytes high->low (kb_sorter.h:1198-1199), then the counter
 * little-endian for KMC (:1200-1201) or big-endian for KFF (:1210-1211) */
template <int SIZE>
KMC_HD void kmc_emit_record(uint8_t *dst, const kmc_u64 (&kmer)[SIZE], kmc_u32 count, kmc_u32 sbytes, kmc_u32 cbytes, bool kff)
{
	for (kmc_u32 j = 0; j < sbytes; ++j)
		dst[j] = (uint8_t)kmc_get_byte<SIZE>(kmer, sbytes - 1 - j);
	for (kmc_u32 j = 0; j < cbytes; ++j)
		dst[sbytes + j] = (uint8_t)(count >> (8 * (kff ? (cbytes - 1 - j) : j)));
}

#endif
