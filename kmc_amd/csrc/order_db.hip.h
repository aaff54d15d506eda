/*
 * kmc_amd/csrc/order_db.hip.h — a GLOBALLY ordered database on the device (what `kmc_tools transform <db> sort <out>` does
 * with KMC's database on the CPU, kmc_tools/kmc1_db_writer.h:368-395).
 *
 * Stage 2 leaves the counted k-mers ordered INSIDE every signature bin: per bin a run of (suffix, count) records and a LUT of how many records
 * each lut_prefix_len-symbol prefix has (kb_sorter.h:1196-1203). Bins partition the k-mers by signature, so every counted k-mer is in exactly
 * one bin: the globally ordered database is a sort of the union, no counts to merge.
 *   k_db_cumsum   one workgroup: a bin's LUT counts -> exclusive prefix sums (record index of the first record of every prefix)
 *   k_db_unpack   one thread per record of a bin: prefix by binary search in those sums, k-mer = prefix . suffix, record = (k-mer words, count)
 *   (sort)        the library's own 8-bit LSD passes over the k-mer bytes of those records (stable: the count rides along above the key)
 *   k_db_pack     one thread per record: (suffix, count) bytes for the NEW lut_prefix_len (kmc1_db_writer.h:388-391: kmer.store big-endian, counter
 *                 little-endian), and the KMC1 LUT (entry i = records with a prefix below i, :376-383) from the prefix boundaries — no atomics
 * HBM-bound byte shuffling; a utility next to the hot path, not part of it (bench.py does not time it).
 */
#ifndef KMC_AMD_ORDER_DB_HIP_H
#define KMC_AMD_ORDER_DB_HIP_H

#include "kernels.hip.h"

__global__ void __launch_bounds__(256) k_db_cumsum(const u64 *__restrict__ counts, u64 n_entries, u64 *__restrict__ sums /* [n_entries + 1] */)
{
	__shared__ u64 s_scan[5];
	u64 carry = 0;
	for (u64 c0 = 0; c0 < n_entries; c0 += 256) {
		const u64 i = c0 + threadIdx.x;
		const u64 v = i < n_entries ? counts[i] : 0;
		u64 total;
		const u64 ex = block_excl_sum<4, u64>(v, s_scan, total);
		if (i < n_entries)
			sums[i] = carry + ex;
		carry += total;
	}
	if (threadIdx.x == 0)
		sums[n_entries] = carry;
}

/* records of one bin: [sbytes suffix bytes, most significant first][cbytes count bytes, least significant first] -> (SIZE k-mer words, 1 count word) */
template <int SIZE>
__global__ void __launch_bounds__(256) k_db_unpack(const uint8_t *__restrict__ recs, u64 n, const u64 *__restrict__ sums, u32 n_entries, u32 k, u32 p, u32 sbytes,
                                                  u32 cbytes, u64 *__restrict__ out /* [n][SIZE + 1] */)
{
	const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
	if (j >= n)
		return;
	u32 lo = 0, hi = n_entries; /* largest i with sums[i] <= j */
	while (hi - lo > 1) {
		const u32 mid = (lo + hi) >> 1;
		if (sums[mid] <= j)
			lo = mid;
		else
			hi = mid;
	}
	const uint8_t *r = recs + j * (u64)(sbytes + cbytes);
	u64 x[SIZE];
#pragma unroll
	for (int w = 0; w < SIZE; ++w)
		x[w] = 0;
	for (u32 q = 0; q < sbytes; ++q) { /* byte sbytes-1-q of the k-mer */
		const u32 pb = sbytes - 1 - q;
#pragma unroll
		for (int w = 0; w < SIZE; ++w)
			if ((pb >> 3) == (u32)w)
				x[w] |= (u64)r[q] << ((pb & 7) * 8);
	}
	const u32 pbit = 2 * (k - p); /* the prefix sits above the suffix symbols */
#pragma unroll
	for (int w = 0; w < SIZE; ++w) {
		if ((pbit >> 6) == (u32)w)
			x[w] |= (u64)lo << (pbit & 63);
		if ((pbit & 63) && (pbit >> 6) + 1 == (u32)w)
			x[w] |= (u64)lo >> (64 - (pbit & 63));
	}
	u64 c = 0;
	for (u32 q = 0; q < cbytes; ++q)
		c |= (u64)r[sbytes + q] << (8 * q);
	u64 *o = out + j * (u64)(SIZE + 1);
#pragma unroll
	for (int w = 0; w < SIZE; ++w)
		o[w] = x[w];
	o[SIZE] = c;
}

template <int SIZE>
__global__ void __launch_bounds__(256) k_db_pack(const u64 *__restrict__ recs /* [n][SIZE + 1] ascending */, u64 n, u32 k, u32 p_out, u32 cbytes, uint8_t *__restrict__ out,
                                                u64 *__restrict__ lut /* [4^p_out], zeroed */)
{
	const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n)
		return;
	const u32 sbytes = (k - p_out) / 4, rb = sbytes + cbytes;
	u64 x[SIZE], nx[SIZE];
	const u64 *r = recs + i * (u64)(SIZE + 1);
#pragma unroll
	for (int w = 0; w < SIZE; ++w) {
		x[w] = r[w];
		nx[w] = i + 1 < n ? r[SIZE + 1 + w] : 0;
	}
	const u64 cnt = r[SIZE];
	uint8_t *o = out + i * (u64)rb;
	for (u32 q = 0; q < sbytes; ++q)
		o[q] = (uint8_t)kmc_get_byte<SIZE>(x, sbytes - 1 - q);
	for (u32 q = 0; q < cbytes; ++q)
		o[sbytes + q] = (uint8_t)(cnt >> (8 * q));
	const u32 pshift = 2 * (k - p_out);
	const u64 n_pref = 1ull << (2 * p_out);
	const u64 pa = kmc_remove_suffix<SIZE>(x, pshift) & (n_pref - 1);
	const u64 pb = i + 1 < n ? (kmc_remove_suffix<SIZE>(nx, pshift) & (n_pref - 1)) : n_pref - 1 + 1;
	for (u64 q = pa + 1; q <= pb && q < n_pref; ++q)
		lut[q] = i + 1; /* records with a prefix below q */
}

#endif
