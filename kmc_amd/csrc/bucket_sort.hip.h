/*
 * kmc_amd/csrc/bucket_sort.hip.h — the LDS half of the hybrid radix sort (gfx950, wave64).
 *
 * The 8-bit LSD passes of kernels.hip.h cost 16 bytes of HBM traffic per record and pass, and a k-mer has ceil(k/4) of them
 * (7 at k = 27, 32 at k = 127). But after the passes over the TOP H bytes of the key (LSD over those H bytes alone = sorted by
 * the top 8H bits) the array is a sequence of buckets — runs of records that share their top 8H bits — and with
 * n / 2^(8H) small, many whole buckets fit into LDS at once. So the remaining ceil(k/4) - H bytes never go through HBM:
 *
 *   k_bucket_bounds   cuts the array into tiles that start and end on bucket boundaries: tile j = the buckets that START in
 *                     window [j S, (j+1) S). One wave per window finds "first bucket boundary at or after j S".
 *   k_bucket_sort     one workgroup per tile: records -> registers -> LDS, grouped by an order-preserving sub-bucket number
 *                     (the tile's key range cut into NB equal slices: counting with returning LDS atomics, one scan, one
 *                     placement), then every sub-bucket (a few records, or the copies of one k-mer) is finished by one thread
 *                     with an insertion sort whose fast path is "not smaller than the last one" (duplicates cost one compare);
 *                     the sorted tile is written back in place, coalesced. 8 B read + 8 B written per record, once, whatever k.
 *
 * This replaces what the reference does below its first radix levels — RadulsSort's recursion into small buckets and the
 * insertion / shell sorts of CSmallSort (raduls_impl.h:133-141,497-510; small_sort.h:29-179) — with an LDS-resident equivalent.
 *
 * A tile longer than the LDS capacity (a bucket far larger than n / 2^(8H): one k-mer repeated thousands of times, or
 * adversarial input) and a tile whose sub-bucket sort exceeds its move budget set `*flag`: the host then sorts that group
 * again with LSD passes over all bytes (kmc_hip.hip: "redo"). Nothing is ever left partially sorted silently.
 *
 * Contract (SortFunction, kb_sorter.h:761-775): the bytes of a record above the key are zero, so comparing whole records
 * orders them by the key.
 */
#ifndef KMC_AMD_BUCKET_SORT_HIP_H
#define KMC_AMD_BUCKET_SORT_HIP_H

#include "kernels.hip.h"

#ifndef BS_BLOCK_THREADS
#define BS_BLOCK_THREADS 768 /* 12 waves; two workgroups per CU at 56 KB of LDS each */
#endif
#ifndef BS_WORDS_PER_THREAD
#define BS_WORDS_PER_THREAD 8 /* 8-byte words a thread holds while a tile is loaded */
#endif
#ifndef BS_LOG_NB
#define BS_LOG_NB 11 /* sub-bucket counters per tile (2048): ~2-3 records per sub-bucket on distinct keys */
#endif
#ifndef BS_MOVE_LIMIT
#define BS_MOVE_LIMIT 4096 /* record moves one thread may spend on its sub-buckets before the tile is handed back to the host */
#endif

template <int SIZE> struct BsCfg {
	static constexpr int THREADS = BS_BLOCK_THREADS;
	static constexpr int ITEMS = (BS_WORDS_PER_THREAD / SIZE) > 2 ? (BS_WORDS_PER_THREAD / SIZE) : 2;
	static constexpr int CAP = THREADS * ITEMS;   /* records a tile may hold */
	static constexpr int STRIDE = CAP / 3 * 2;    /* window length S: a tile is S records on average, CAP - S of slack for its last bucket */
	static constexpr int LOG_NB = BS_LOG_NB, NB = 1 << LOG_NB;
	static_assert(CAP < 65536, "tile-relative positions are kept in 16 bits");
	static_assert(NB % 4 == 0 && NB / 4 <= THREADS, "the counter scan gives 4 counters to a thread");
};
template <int SIZE> constexpr size_t bs_lds_bytes()
{
	return (size_t)BsCfg<SIZE>::CAP * SIZE * 8 + ((size_t)BsCfg<SIZE>::NB + 1) * 4 + (BsCfg<SIZE>::THREADS / 64 + 2) * 4 + 16;
}
/* average bucket size the host aims for when it picks H: two orders of magnitude below the slack, because k-mers that share a
 * minimizer are clustered (measured on the bench's bins: the largest of 2^22 buckets holds 100x the average) */
template <int SIZE> constexpr u64 bs_target_bucket() { return (BsCfg<SIZE>::CAP - BsCfg<SIZE>::STRIDE) / 64 > 4 ? (BsCfg<SIZE>::CAP - BsCfg<SIZE>::STRIDE) / 64 : 4; }

/* the top 64 bits of the key (key_bits = 8 * key bytes, bits [key_bits-1 : 0] of the record), left-aligned */
template <int SIZE> __device__ __forceinline__ u64 bs_p64(const u64 (&x)[SIZE], u32 key_bits)
{
	if constexpr (SIZE == 1)
		return x[0] << (64 - key_bits);
	else {
		const u32 tw = (key_bits - 1) >> 6, tb = key_bits - 64 * tw; /* top word, bits used in it (1..64) */
		u64 hi = x[0], lo = 0;
#pragma unroll
		for (int i = 1; i < SIZE; ++i) {
			if (tw == (u32)i) {
				hi = x[i];
				lo = x[i - 1];
			}
		}
		return tb == 64 ? hi : ((hi << (64 - tb)) | (tw ? (lo >> tb) : 0ull));
	}
}

/* One wave per window j (0..n_win): bounds[j] = the first index i >= j S at which a bucket starts (i == 0, or the top `hbits`
 * bits of record i differ from those of record i-1), n if there is none. hbits == 0: the whole array is one bucket. */
template <int SIZE>
__global__ void __launch_bounds__(256) k_bucket_bounds(const u64 *__restrict__ recs, u64 n, u64 n_win, u32 key_bits, u32 hbits, u64 *__restrict__ bounds)
{
	constexpr u64 S = BsCfg<SIZE>::STRIDE;
	const u32 lane = threadIdx.x & 63;
	const u64 j = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (j > n_win)
		return;
	const u64 p = j * S;
	u64 b;
	if (j == 0)
		b = 0;
	else if (p >= n || hbits == 0)
		b = n;
	else {
		const u32 sh = 64 - hbits;
		auto bucket_at = [&](u64 i) {
			u64 x[SIZE];
			load_rec<SIZE>(recs + i * SIZE, x);
			return bs_p64<SIZE>(x, key_bits) >> sh;
		};
		const u64 v = bucket_at(p - 1);
		/* the boundary is nearly always within a few records: look at [p, p + 64) first, then 64-ary search on the monotone predicate
		 * "bucket != v" over what is left */
		u64 lo = p, hi = n; /* answer in [lo, hi]; everything below lo is in bucket v, hi stands for "a different bucket" */
		{
			const u64 i = lo + lane;
			const bool t = i < hi && bucket_at(i) != v;
			const u64 m = __ballot(t);
			if (m)
				hi = lo = lo + (u64)(__ffsll((long long)m) - 1);
			else
				lo = (hi - lo) < 64 ? hi : lo + 64;
		}
		while (hi - lo > 64) {
			const u64 step = (hi - lo + 63) / 64;
			const u64 q = lo + (u64)lane * step;
			const bool in = q < hi;
			const bool t = in && bucket_at(q) != v;
			const u64 m = __ballot(t), inm = __ballot(in);
			if (m) {
				const u64 f = (u64)(__ffsll((long long)m) - 1);
				hi = lo + f * step;              /* the first probe in a different bucket: the answer is at or below it */
				lo = f ? lo + (f - 1) * step + 1 : lo; /* f == 0: cannot happen (lo itself is probed and is still in bucket v or is the answer) */
				if (f == 0)
					hi = lo;
			} else {
				const u64 last = 63 - (u64)__clzll((long long)inm);
				lo = lo + last * step + 1;
			}
		}
		if (lo < hi) {
			const u64 i = lo + lane;
			const bool t = i < hi && bucket_at(i) != v;
			const u64 m = __ballot(t);
			b = m ? lo + (u64)(__ffsll((long long)m) - 1) : hi;
		} else
			b = lo;
	}
	if (lane == 0)
		bounds[j] = b;
}

/* One workgroup per window j: sorts tile [bounds[j], bounds[j+1]) in place (empty when no bucket starts in the window). */
template <int SIZE>
__global__ void __launch_bounds__(BsCfg<SIZE>::THREADS) k_bucket_sort(u64 *__restrict__ recs, u32 key_bits, u32 hbits, const u64 *__restrict__ bounds, u32 *flag)
{
	constexpr int THREADS = BsCfg<SIZE>::THREADS, ITEMS = BsCfg<SIZE>::ITEMS, CAP = BsCfg<SIZE>::CAP, NB = BsCfg<SIZE>::NB, LOG_NB = BsCfg<SIZE>::LOG_NB;
	constexpr u64 S = BsCfg<SIZE>::STRIDE;
	KMC_DYN_LDS(unsigned char, s_raw);
	u64 *s_rec = reinterpret_cast<u64 *>(s_raw);                  /* [CAP * SIZE] record-major */
	u32 *s_cnt = reinterpret_cast<u32 *>(s_rec + (size_t)CAP * SIZE); /* [NB + 1] counts -> first slot of each sub-bucket, [NB] = tile length */
	u32 *s_tmp = s_cnt + NB + 1;                                  /* [THREADS/64 + 1] scan scratch, then [1] the "over budget" mark */
	u32 *s_fail = s_tmp + THREADS / 64 + 1;

	const u64 j = blockIdx.x;
	const u64 b0 = bounds[j], b1 = bounds[j + 1];
	if (b0 >= (j + 1) * S || b0 >= b1)
		return; /* no bucket starts in this window */
	if (b1 - b0 > (u64)CAP) {
		if (threadIdx.x == 0)
			atomicOr(flag, 1u); /* a bucket (or two) far beyond the expected size: the host sorts this group again with LSD passes */
		return;
	}
	const u32 len = (u32)(b1 - b0);
	const u32 tid = threadIdx.x;
	u64 *__restrict__ T = recs + b0 * SIZE;

	/* the tile's key range, from its first and last bucket: [lo64, hi64] in units of the left-aligned top 64 key bits */
	u64 lo64, hi64;
	{
		u64 f[SIZE], l[SIZE];
		load_rec<SIZE>(T, f);
		load_rec<SIZE>(T + (size_t)(len - 1) * SIZE, l);
		const u64 hmask = hbits == 0 ? ~0ull : (hbits >= 64 ? 0ull : ((1ull << (64 - hbits)) - 1));
		lo64 = bs_p64<SIZE>(f, key_bits) & ~hmask;
		hi64 = bs_p64<SIZE>(l, key_bits) | hmask;
	}
	const u64 span = hi64 - lo64;
	const u32 bits = span ? 64u - (u32)__clzll((long long)span) : 0u;
	const u32 sh = bits > (u32)LOG_NB ? bits - LOG_NB : 0u; /* (span >> sh) < NB */

	for (u32 i = tid; i <= (u32)NB; i += THREADS)
		s_cnt[i] = 0;
	if (tid == 0)
		*s_fail = 0;
	u64 key[ITEMS][SIZE];
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = r * THREADS + tid;
		if (idx < len)
			load_rec<SIZE>(T + (size_t)idx * SIZE, key[r]);
		else {
#pragma unroll
			for (int w = 0; w < SIZE; ++w)
				key[r][w] = 0;
		}
	}
	__syncthreads();
	/* sub-bucket number (order-preserving) and arrival rank inside the sub-bucket */
	u32 ir[ITEMS]; /* [15:0] sub-bucket, [31:16] rank */
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = r * THREADS + tid;
		ir[r] = 0;
		if (idx < len) {
			const u64 rel = (bs_p64<SIZE>(key[r], key_bits) - lo64) >> sh;
			const u32 id = rel < (u64)NB ? (u32)rel : (u32)NB - 1; /* in range by construction; the clamp is for records of a corrupt bin (bits above the
			                                                          * key, a count that disagrees with the stream: the error word is already set) */
			const u32 rk = __hip_atomic_fetch_add(&s_cnt[id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			ir[r] = id | (rk << 16);
		}
	}
	__syncthreads();
	/* counts -> first slots: thread t < NB/4 owns counters 4t .. 4t+3 */
	{
		u32 c[4] = {0, 0, 0, 0}, sum = 0;
		if (tid < (u32)NB / 4) {
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				c[q] = s_cnt[tid * 4 + q];
				sum += c[q];
			}
		}
		u32 total;
		u32 run = block_excl_sum<THREADS / 64, u32>(sum, s_tmp, total);
		if (tid < (u32)NB / 4) {
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				s_cnt[tid * 4 + q] = run;
				run += c[q];
			}
		}
		if (tid == 0)
			s_cnt[NB] = total; /* == len */
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = r * THREADS + tid;
		if (idx < len) {
			const u32 pos = s_cnt[ir[r] & 0xFFFFu] + (ir[r] >> 16);
			store_rec<SIZE>(s_rec + (size_t)pos * SIZE, key[r]);
		}
	}
	__syncthreads();
	/* every sub-bucket is finished by one thread. The common shapes — one or two distinct k-mers, each with all its copies — cost one
	 * compare per record ("not smaller than the largest so far"). */
	{
		const u32 n_ids = (u32)(span >> sh) + 1; /* <= NB */
		u32 moves = 0;
		for (u32 id = tid; id < n_ids; id += THREADS) {
			const u32 a = s_cnt[id], b = s_cnt[id + 1];
			if (b - a < 2)
				continue;
			u64 last[SIZE];
			load_rec<SIZE>(s_rec + (size_t)a * SIZE, last);
			for (u32 i = a + 1; i < b; ++i) {
				u64 x[SIZE];
				load_rec<SIZE>(s_rec + (size_t)i * SIZE, x);
				if (!kmc_less<SIZE>(x, last)) {
#pragma unroll
					for (int w = 0; w < SIZE; ++w)
						last[w] = x[w];
					continue;
				}
				u32 q = i;
				while (true) { /* x < record q-1: shift it up */
					u64 y[SIZE];
					load_rec<SIZE>(s_rec + (size_t)(q - 1) * SIZE, y);
					if (!kmc_less<SIZE>(x, y))
						break;
					store_rec<SIZE>(s_rec + (size_t)q * SIZE, y);
					++moves;
					if (--q == a)
						break;
				}
				store_rec<SIZE>(s_rec + (size_t)q * SIZE, x);
				if (moves > (u32)BS_MOVE_LIMIT)
					break; /* between two insertions: the sub-bucket still holds all its records */
			}
			if (moves > (u32)BS_MOVE_LIMIT) {
				*s_fail = 1;
				break;
			}
		}
	}
	__syncthreads();
	if (*s_fail && tid == 0)
		atomicOr(flag, 1u); /* many distinct keys that the tile's NB slices do not separate: LSD passes will sort them (the tile stays a permutation) */
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = r * THREADS + tid;
		if (idx < len) {
			u64 x[SIZE];
			load_rec<SIZE>(s_rec + (size_t)idx * SIZE, x);
			store_rec<SIZE>(T + (size_t)idx * SIZE, x);
		}
	}
}

#endif
