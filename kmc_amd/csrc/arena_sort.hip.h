/*
 * kmc_amd/csrc/arena_sort.hip.h — the repeat-rich buckets of a group, sorted together (round 6; one-word records, gfx950, wave64).
 *
 * After the HBM passes over the top key bytes a group is a sequence of buckets (bucket_sort.hip.h). k_bucket_rank finishes a tile inside LDS by pairwise ranking:
 * the work is the sum of the squares of the tile's buckets, fine for the buckets of 30x reads of a random genome (a k-mer's copies and their error neighbours) and
 * wrong for what a real genome adds — repeat families whose k-mers occur hundreds to millions of times. Until round 5 those buckets were walked pairwise by the whole
 * workgroup (a few hundred to a few thousand records), sorted by ONE workgroup each (beyond a tile's capacity: k_giant_tiles) or sent back to the host with their whole bin
 * (beyond 2^20 records: LSD passes over every byte, again). What the reference does with a bucket that stays large is give it a share of the threads and another radix
 * level (raduls_impl.h:680-737), and sort the small ones in O(n log n) (raduls_impl.h:497-510, small_sort.h:29-179): cost independent of the bucket's shape.
 *
 * Here: every bucket beyond BR_MID records — whatever its length — becomes an ENTRY of the group's arena (k_bucket_rank lists it: bucket_sort.hip.h ArenaEntry), and
 *
 *   k_arena_plan     one workgroup: entry e gets the arena records [off[e], off[e+1]), its work items, and its bucket number (the key bits above `rbits`)
 *   k_arena_gather   arena[off[e] + i] = e << rbits | (the low rbits of the bucket's record i): the entry's ordinal rides above what is left of the key. Digit histograms of
 *                    every pass in the same read; the look-back rows of the passes are cleared here (their number is only known on the device)
 *   k_onesweep_dyn   the library's own 8-bit LSD pass (kernels.hip.h), ceil((rbits + bits of the last ordinal) / 8) times over the arena: all 256 CUs, stable, blind to
 *                    what the digits look like. The arena ends ordered by (entry, key)
 *   k_arena_finish   per work item. A bucket that fits a tile (kind 1) goes back to its place in the group's array, in order — its tile is ranked and counted afterwards
 *                    by k_bucket_rank_heavy, which leaves such a bucket where it lies. A giant bucket (kind 0) is counted from the arena, cut into segments that report
 *                    into the output slots of the windows the bucket covers (those windows hold no tile: nothing else starts in them), each segment by one workgroup:
 *                    run tails, counts (a run may reach back over segments: one 64-ary search), cutoffs, records, LUT, tallies — what k_giant_tiles did for a whole tile
 *                    with one workgroup.
 *
 * Nothing listed: every launch here returns after one load. Failure modes produce LESS output, never wrong output: an overflowing list or an arena beyond one pass
 * portion raises the group's redo flag and cancels the listed work (k_arena_plan).
 */
#ifndef KMC_AMD_ARENA_SORT_HIP_H
#define KMC_AMD_ARENA_SORT_HIP_H

#include "bucket_sort.hip.h"

#ifndef AR_THREADS
#define AR_THREADS 1024
#endif
constexpr int AR_ITEMS = 8, AR_CHUNK = AR_THREADS * AR_ITEMS, AR_NW = AR_THREADS / 64;
#ifndef AF_THREADS
#define AF_THREADS 256 /* k_arena_finish: its items are a few hundred to ~11 000 records — small workgroups, many of them in flight (1024 threads: 0.70 ms per group of the spectrum leg) */
#endif
constexpr u32 AR_MAX_PASS = 8;
constexpr u64 AR_MAX_RECORDS = 1ull << 29;             /* one portion of k_onesweep (30-bit look-back counts) */
constexpr int AR_CHUNK_ENT = AR_CHUNK / BR_MID + 3;    /* entries a chunk of the arena can touch: every entry has more than BR_MID records */
static_assert(AR_THREADS >= 256 && AR_CHUNK_ENT <= AR_THREADS, "one digit / one chunk entry per thread");

struct ArenaWork {
	u32 *arena_off;    /* [cap + 1] first arena record of entry e; [entries] = records in the arena */
	u32 *item_off;     /* [cap + 1] first work item of entry e */
	u64 *bucket_hi;    /* [cap] the entry's bucket number: its records' key >> rbits */
	u64 *A, *B;        /* the arena and its twin (the passes alternate) */
	u64 *ghist;        /* [AR_MAX_PASS][256] digit histograms (zeroed by the host with the group's zero region) */
	u64 *dbase;        /* [AR_MAX_PASS + 1][256] digit bases (k_hist_scan), then a row nobody reads (digit_base_next of the passes) */
	u32 *status;       /* [AR_MAX_PASS][status_stride] look-back rows of the passes: cleared by k_arena_gather as far as this group's arena needs them */
	u32 status_stride; /* words */
	const u64 *S0;     /* the group's ordered array (gr.S[0]) */
};

/* the entry that holds arena record x (off[e] <= x < off[e + 1]; off[0] = 0 <= x < off[n]): a 64-ary search, executed by ONE full wave — three round trips for 2^18
 * entries where a binary search by one thread takes eighteen */
__device__ __forceinline__ u32 ar_find_entry(const u32 *__restrict__ off, u32 n, u32 x, u32 lane)
{
	u32 a = 0, b = n; /* the answer is in [a, b) */
	while (b - a > 1) {
		const u32 step = (b - a + 63) / 64;
		const u32 q = a + lane * step;
		const bool le = q < b && off[q] <= x; /* true for a prefix of the lanes (lane 0: q = a) */
		const u64 m = __ballot(le);
		const u32 last = 63u - (u32)__clzll((long long)m);
		a = a + last * step;
		b = a + step < b ? a + step : b;
	}
	return a;
}

/* first i in (lo, hi] with pred(i), for a predicate that is false up to some index and true from there on, given pred(hi) (hi may be a virtual end): 64-ary, executed by ONE
 * full wave. Indices are modulo 2^64: lo = -1 stands for "nothing in front of index 0". */
template <typename Pred> __device__ __forceinline__ u64 bd_first(u64 lo, u64 hi, u32 lane, Pred pred)
{
	while (hi - lo > 1) {
		const u64 span = hi - lo - 1; /* candidates lo + 1 .. hi - 1 */
		const u64 stp = (span + 63) / 64;
		const u64 t = lo + 1 + (u64)lane * stp;
		const bool in = t - lo - 1 < span; /* (modular: t < hi) */
		const bool pr = in && pred(t);
		const u64 m = __ballot(pr), inm = __ballot(in);
		if (m) {
			const u32 f = (u32)__ffsll((long long)m) - 1u;
			hi = lo + 1 + (u64)f * stp;
			if (f)
				lo = hi - stp;
			else
				break; /* lo + 1 itself */
		} else
			lo = lo + 1 + (u64)((u32)__popcll(inm) - 1u) * stp;
	}
	return hi;
}

struct GrpDetect {
	u32 g, blk_prefix[GRP_MAX + 1]; /* blocks of 64 BD_STRIDE records of bin b */
	u64 n[GRP_MAX];
};
/* The buckets beyond BR_MID records of a group, found without reading it: one wave per block of 64 x BD_STRIDE records looks at every BD_STRIDE-th record (a 64-byte
 * sector each: 1/88 of the array's sectors). Two neighbouring samples with the same bucket number are a bucket of BD_STRIDE + 1 records at least, and a bucket of
 * 2 BD_STRIDE records cannot lie between the samples: every bucket beyond BR_MID >= 2 BD_STRIDE - 1 records shows. The wave that holds the FIRST sample of such a run finds the
 * bucket's first record (behind the sample in front of it: one or two rounds) and its end (the probes' distance grows by 64 x per round, then narrows by 64 x per round: a satellite of
 * millions of records costs a dozen round trips) and lists it when it is longer than BR_MID: kind 1 up to a tile's capacity, kind 0 (giant) beyond. Replaces the listing by
 * k_bucket_rank's workgroups and their second visit (round 6, first version: the tiles with such a bucket were read and their bucket starts found twice). */
__global__ void __launch_bounds__(256) k_bucket_detect(const GrpRank gr, const GrpDetect gd, u32 rbits)
{
	constexpr u64 DS = BD_STRIDE, S = BrCfg<1>::STRIDE;
	const u32 lane = threadIdx.x & 63;
	const u32 item = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (item >= gd.blk_prefix[gd.g])
		return;
	const u32 bin = (u32)__builtin_amdgcn_readfirstlane((int)grp_find(gd.blk_prefix, gd.g, item));
	const u64 *__restrict__ recs = gr.S[bin];
	const u64 n = gd.n[bin];
	const u64 p0 = (u64)(item - gd.blk_prefix[bin]) * (64 * DS);
	auto bucket_at = [&](u64 i) -> u64 { return recs[i] >> rbits; };
	const u64 pos = p0 + (u64)lane * DS;
	const bool valid = pos < n;
	/* the block's 64 samples, the one in front of them (lane 0) and the one behind them (lane 63): three loads under way together */
	const bool edge_p = lane == 0 && valid && pos >= DS, edge_n = lane == 63 && pos + DS < n;
	const u64 x = recs[valid ? pos : 0], xe = recs[edge_p ? pos - DS : (edge_n ? pos + DS : 0)];
	const u64 bk = valid ? x >> rbits : 0ull;
	u64 pb = __shfl_up(bk, 1), nb = __shfl_down(bk, 1);
	bool pvalid = true, nvalid = (bool)__shfl_down((int)valid, 1);
	if (lane == 0) {
		pvalid = edge_p;
		pb = xe >> rbits;
	}
	if (lane == 63) {
		nvalid = edge_n;
		nb = xe >> rbits;
	}
	const bool same_prev = valid && pvalid && pb == bk, same_next = valid && nvalid && nb == bk;
	u64 starts = __ballot(same_next && !same_prev);
	while (starts) { /* wave-uniform */
		const u32 l = (u32)__ffsll((long long)starts) - 1u;
		starts &= starts - 1;
		const u64 b = __shfl(bk, (int)l);
		const u64 q = p0 + (u64)l * DS;
		/* the sample at q - DS (if there is one) lies in another bucket, the one at q + DS in this one */
		const u64 first = bd_first(q - DS, q, lane, [&](u64 i) { return (long long)i >= 0 && bucket_at(i) == b; });
		u64 lo = q + DS, hi, step = DS;
		while (true) { /* outwards */
			const u64 t = lo + (u64)(lane + 1) * step;
			const bool diff = t >= n || t < lo || bucket_at(t) != b;
			const u64 m = __ballot(diff);
			if (m) {
				const u32 f = (u32)__ffsll((long long)m) - 1u;
				hi = lo + (u64)(f + 1) * step;
				if (hi > n || hi < lo)
					hi = n;
				lo = lo + (u64)f * step;
				break;
			}
			lo += 64 * step;
			if (step < (1ull << 40))
				step *= 64;
		}
		const u64 end = bd_first(lo, hi, lane, [&](u64 i) { return i >= n || bucket_at(i) != b; });
		const u64 len = end - first;
		if (lane == 0 && len > (u64)BR_MID) {
			if (len >= (1ull << 32))
				atomicOr(&gr.arena_dyn[AR_OVERFLOW], 1u);
			else {
				const u32 e = atomicAdd(&gr.arena_dyn[AR_N_ENT], 1u);
				if (e < gr.arena_cap)
					gr.arena_ent[e] = ArenaEntry{((u64)(recs - gr.S[0]) + first) | ((u64)bin << 40) | (len > (u64)BrCfg<1>::CAP ? 0ull : 1ull << 44), (u32)len,
					                             gr.win_prefix[bin] + (u32)(first / S)};
				else
					atomicOr(&gr.arena_dyn[AR_OVERFLOW], 1u);
			}
		}
	}
}

__global__ void __launch_bounds__(AR_THREADS) k_arena_plan(const GrpRank gr, const ArenaWork aw, u32 rbits, u32 *flag)
{
	constexpr u64 S = BrCfg<1>::STRIDE;
	__shared__ u64 s_scan64[AR_NW + 1];
	__shared__ u32 s_scan32[AR_NW + 1];
	u32 *dyn = gr.arena_dyn;
	const u32 tid = threadIdx.x;
	const u32 listed = dyn[AR_N_ENT];
	const bool overflow = dyn[AR_OVERFLOW] != 0 || listed > gr.arena_cap;
	if (listed == 0 && !overflow)
		return; /* the usual case: every dyn word stays zero, every later launch returns */
	const u32 obits = listed > 1 ? 32u - (u32)__clz((int)(listed - 1)) : 0u;
	bool bail = overflow || rbits + obits > 64u || (rbits + obits + 7) / 8 > AR_MAX_PASS;
	u64 m_run = 0;
	u32 it_run = 0;
	if (!bail) {
		for (u32 base = 0; base < listed; base += AR_THREADS) {
			const u32 e = base + tid;
			u64 len = 0;
			u32 items = 0;
			if (e < listed) {
				const ArenaEntry en = gr.arena_ent[e];
				len = en.len;
				items = 1;
				if (!((en.w0 >> 44) & 1ull)) { /* a giant bucket: one segment for the rest of its first window, two for every window it covers entirely */
					const u32 bin = (u32)(en.w0 >> 40) & 15u;
					const u32 w = en.gtile - gr.win_prefix[bin];
					const u64 b1 = gr.bounds[bin][w + 1];
					const u64 j_last = (b1 - 1) / S;
					items = j_last > (u64)w + 1 ? (u32)(2 * (j_last - w) - 1) : 1u;
				}
				aw.bucket_hi[e] = rbits < 64 ? aw.S0[en.w0 & ((1ull << 40) - 1)] >> rbits : 0ull;
			}
			u64 tot64;
			const u64 ex64 = block_excl_sum<AR_NW, u64>(len, s_scan64, tot64);
			u32 tot32;
			const u32 ex32 = block_excl_sum<AR_NW, u32>(items, s_scan32, tot32);
			if (e < listed) {
				aw.arena_off[e] = (u32)(m_run + ex64); /* (meaningless beyond AR_MAX_RECORDS: the plan is dropped then) */
				aw.item_off[e] = it_run + ex32;
			}
			m_run += tot64;
			it_run += tot32;
		}
		if (m_run > AR_MAX_RECORDS)
			bail = true;
	}
	if (tid == 0) {
		if (bail) { /* nothing of the listed work may run and k_bucket_rank reports nothing either: the group comes back (LSD passes over every byte) */
			dyn[AR_M] = dyn[AR_N_PASS] = dyn[AR_N_ITEMS] = 0;
			dyn[AR_OVERFLOW] = 2; /* k_bucket_rank: no bucket of this group has been put in order — leave every tile alone */
			atomicOr(flag, 1u);
		} else {
			aw.arena_off[listed] = (u32)m_run;
			aw.item_off[listed] = it_run;
			dyn[AR_M] = (u32)m_run;
			dyn[AR_N_PASS] = (rbits + obits + 7) / 8;
			dyn[AR_N_ITEMS] = it_run;
		}
	}
}

/* grid: persistent workgroups over chunks of AR_CHUNK arena records. LDS: n_pass_max x 256 counters. */
__global__ void __launch_bounds__(AR_THREADS) k_arena_gather(const GrpRank gr, const ArenaWork aw, u32 rbits, u32 n_pass_max)
{
	KMC_DYN_LDS(u32, s_h); /* [n_pass_max][256] */
	__shared__ u32 s_off[AR_CHUNK_ENT + 1];
	__shared__ u64 s_gpos[AR_CHUNK_ENT + 1];
	__shared__ u32 s_e0;
	const u32 *dyn = gr.arena_dyn;
	const u32 M = dyn[AR_M];
	if (M == 0)
		return;
	const u32 n_pass = dyn[AR_N_PASS], n_ent = dyn[AR_N_ENT];
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	{ /* the look-back rows this arena's passes will use */
		const u32 words = ((M + (u32)RsCfg<1>::TILE - 1) / (u32)RsCfg<1>::TILE) * 256u;
		for (u32 p = 0; p < n_pass; ++p)
			for (u32 i = blockIdx.x * AR_THREADS + tid; i < words; i += gridDim.x * AR_THREADS)
				aw.status[(size_t)p * aw.status_stride + i] = 0;
	}
	for (u32 i = tid; i < n_pass_max * 256; i += AR_THREADS)
		s_h[i] = 0;
	const u64 rmask = rbits >= 64 ? ~0ull : ((1ull << rbits) - 1);
	for (u32 base = blockIdx.x * (u32)AR_CHUNK; base < M; base += gridDim.x * (u32)AR_CHUNK) {
		__syncthreads(); /* s_off / s_e0 of the chunk before; the counters' zeroes */
		if (wave == 0) {
			const u32 e = ar_find_entry(aw.arena_off, n_ent, base, lane);
			if (lane == 0)
				s_e0 = e;
		}
		__syncthreads();
		const u32 e0 = s_e0;
		if (tid <= (u32)AR_CHUNK_ENT) {
			s_off[tid] = e0 + tid <= n_ent ? aw.arena_off[e0 + tid] : 0xFFFFFFFFu; /* arena_off[n_ent] = M */
			s_gpos[tid] = e0 + tid < n_ent ? gr.arena_ent[e0 + tid].w0 & ((1ull << 40) - 1) : 0ull;
		}
		__syncthreads();
		u64 v[AR_ITEMS];
#pragma unroll
		for (int r = 0; r < AR_ITEMS; ++r) { /* every row's load is under way before the first one is waited for */
			const u32 i = base + wave * (AR_ITEMS * 64) + r * 64 + lane;
			v[r] = 0;
			if (i < M) {
				u32 j = 0; /* the chunk's entry that holds record i: a row of 64 lies inside one entry nearly always */
				while (s_off[j + 1] <= i)
					++j;
				const u64 x = aw.S0[s_gpos[j] + (i - s_off[j])];
				v[r] = rbits >= 64 ? x : (((u64)(e0 + j) << rbits) | (x & rmask));
			}
		}
#pragma unroll
		for (int r = 0; r < AR_ITEMS; ++r) {
			const u32 i = base + wave * (AR_ITEMS * 64) + r * 64 + lane;
			const bool valid = i < M;
			if (valid)
				aw.A[i] = v[r];
			const u64 act = __ballot(valid);
			for (u32 b = 0; b < n_pass; ++b) { /* as k_hist: a wave whose lanes hold one digit value — the copies of one k-mer — adds once */
				const u32 d = (u32)(v[r] >> (8 * b)) & 0xFFu;
				const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
				if (act && __ballot(valid && d == d0) == act) {
					if (lane == (u32)__ffsll((long long)act) - 1u)
						atomicAdd(&s_h[b * 256 + d0], (u32)__popcll(act));
				} else if (valid)
					atomicAdd(&s_h[b * 256 + d], 1u);
			}
		}
	}
	__syncthreads();
	for (u32 i = tid; i < n_pass * 256; i += AR_THREADS) {
		const u32 c = s_h[i];
		if (c)
			atomicAdd(&aw.ghist[i], (u64)c);
	}
}

/* grid: workgroup b of G takes the work items that START in its share [M b / G, M (b + 1) / G) of the arena (items are 385 .. ~11 000 records: shares of equal length
 * are shares of equal work, and nothing is drawn from a counter: a ticket + a binary search by one thread per item were 12 us of latency in front of ~3 us of work). */
__global__ void __launch_bounds__(AF_THREADS) k_arena_finish(const GrpRank gr, const ArenaWork aw, DevParams P, u32 rbits, u32 lut_shards, u64 lut_stride, u32 lut_mask, u32 *err)
{
	constexpr int THREADS = AF_THREADS, ITEMS = AR_ITEMS, NW = AF_THREADS / 64, CHUNK = AF_THREADS * AR_ITEMS;
	constexpr u32 NONE = 0xFFFFFFFFu;
	__shared__ u32 s_ent, s_prev;
	__shared__ u32 s_wlast[NW], s_wcnt[NW], s_tal[NW * 3];
	u32 *dyn = gr.arena_dyn;
	const u32 M = dyn[AR_M];
	if (M == 0)
		return;
	const u32 n_ent = dyn[AR_N_ENT];
	const u64 *__restrict__ src = (dyn[AR_N_PASS] & 1u) ? aw.B : aw.A;
	const u32 tid = threadIdx.x, lane = tid & 63;
	const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const u64 rmask = rbits >= 64 ? ~0ull : ((1ull << rbits) - 1);
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const bool use_lut = P.lut_prefix_len != 0 && !P.kff && !P.without_output;
	const u32 pshift = 2 * (P.k - P.lut_prefix_len);
	const u64 kmask = 2 * P.k < 64 ? ((1ull << (2 * P.k)) - 1) : ~0ull; /* drops a group tag above the k-mer */
	const u64 lane_lt = (1ull << lane) - 1;
	/* the LUT prefix of a k-mer lies in the key bits above rbits whenever 2 (k - p) >= rbits: every k-mer of a bucket has the same one, and a segment adds its counted k-mers
	 * to it ONCE (one add per counted k-mer on one address — same-address device atomics take ~11 ns each — was most of this kernel's time on repeat-rich input) */
	const bool lut_once = pshift >= rbits;
	const u32 lo = (u32)((u64)M * blockIdx.x / gridDim.x), hi = (u32)((u64)M * (blockIdx.x + 1) / gridDim.x);
	if (lo >= hi)
		return;
	if (wave == 0) { /* the entry that holds arena record `lo` */
		const u32 a = ar_find_entry(aw.arena_off, n_ent, lo, lane);
		if (lane == 0)
			s_ent = a;
	}
	__syncthreads();
	for (u32 e = s_ent; e < n_ent; ++e) {
		const u32 off = aw.arena_off[e];
		if (off >= hi)
			break;
		const u32 e_items = aw.item_off[e + 1] - aw.item_off[e], e_len = gr.arena_ent[e].len;
		/* the entry's segments: segment s starts at off + e_len s / e_items */
		u32 seg = 0;
		if (off < lo) { /* the first segment that starts at or behind lo */
			seg = (u32)((((u64)(lo - off)) * e_items + e_len - 1) / e_len);
			while (seg > 0 && off + (u32)((u64)e_len * (seg - 1) / e_items) >= lo)
				--seg;
			while (seg < e_items && off + (u32)((u64)e_len * seg / e_items) < lo)
				++seg;
		}
		for (; seg < e_items; ++seg) {
			if (off + (u32)((u64)e_len * seg / e_items) >= hi)
				break;
			__syncthreads(); /* the LDS words of the item before */
		const ArenaEntry en = gr.arena_ent[e];
		const u64 gpos = en.w0 & ((1ull << 40) - 1);
		const u32 len = en.len;
		const u64 *__restrict__ B = src + off; /* the bucket, in order */
		const u64 khi = rbits < 64 ? aw.bucket_hi[e] << rbits : 0ull; /* the key bits above rbits */
		if ((en.w0 >> 44) & 1ull) { /* back to where it came from, in order; k_bucket_rank_heavy takes the tile from there */
			u64 *dst = const_cast<u64 *>(aw.S0) + gpos;
			for (u32 i = tid; i < len; i += THREADS)
				dst[i] = khi | (B[i] & rmask);
			if (tid == 0) {
				atomicAdd(&dyn[AR_STAT_MID_N], 1u);
				atomicAdd(reinterpret_cast<u64 *>(dyn + AR_STAT_MID_REC), (u64)len);
			}
			continue;
		}
		/* ---- a segment of a giant bucket: run lengths, cutoffs, records (kb_sorter.h:1128-1281), chunk by chunk, into the segment's own output slot */
		const u32 bin = (u32)(en.w0 >> 40) & 15u;
		const u32 w = en.gtile - gr.win_prefix[bin];
		const u32 nseg = e_items;
		const u32 s0 = (u32)((u64)len * seg / nseg), s1 = (u32)((u64)len * (seg + 1) / nseg);
		const u32 slot_id = seg == 0 ? 2 * w + 1 : 2 * (w + 1) + (seg - 1);
		const u64 pos = gpos - (u64)(gr.S[bin] - gr.S[0]); /* bin-relative */
		uint8_t *const span = gr.scratch[bin] + (pos + s0) * 8ull;
		u64 *lut = use_lut ? gr.lut_base[bin] + (size_t)(slot_id % lut_shards) * lut_stride : nullptr;
		if (wave == 0) { /* the last run tail in front of the segment: the record before it, unless that one is a copy of the segment's first — then the run's start - 1 */
			u32 pt = NONE;
			if (s0 > 0) {
				const u64 first[1] = {B[s0]};
				pt = B[s0 - 1] != first[0] ? s0 - 1 : (u32)run_start_search<1>(B, (u64)s0, lane, first) - 1u;
			}
			if (lane == 0)
				s_prev = pt;
		}
		__syncthreads();
		u32 prev_tail = s_prev; /* bucket-relative, -1 = none */
		u32 counted_total = 0, nu = 0, nb = 0, na = 0;
		for (u32 c0 = s0; c0 < s1; c0 += CHUNK) {
			const u32 crel = wave * (ITEMS * 64);
			u64 key[ITEMS];
			u32 tail_bits = 0, wlast = NONE;
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 idx = c0 + crel + r * 64 + lane;
				bool is_tail = false;
				key[r] = 0;
				if (idx < s1) {
					key[r] = B[idx];
					is_tail = idx + 1 >= len || B[idx + 1] != key[r]; /* (the record behind a segment's last belongs to the next segment of the same bucket) */
				}
				const u64 m = __ballot(is_tail);
				if (is_tail)
					tail_bits |= 1u << r;
				if (m)
					wlast = crel + r * 64 + 63 - (u32)__clzll((long long)m);
			}
			if (lane == 0)
				s_wlast[wave] = wlast;
			__syncthreads();
			u32 carry = prev_tail - c0, chunk_last = NONE; /* chunk-relative, modulo 2^32 */
#pragma unroll
			for (int x = 0; x < NW; ++x) {
				const u32 t = s_wlast[x];
				if (t != NONE) {
					chunk_last = t;
					if (x < (int)wave)
						carry = t;
				}
			}
			carry = (u32)__builtin_amdgcn_readfirstlane((int)carry);
			u32 cnt[ITEMS], rk[ITEMS], nc = 0;
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 rowrel = crel + r * 64;
				const u64 m = __ballot((tail_bits >> r) & 1u);
				const u64 m_lt = m & lane_lt;
				const u32 prev = m_lt ? rowrel + 63 - (u32)__clzll((long long)m_lt) : carry;
				const u32 c = rowrel + lane - prev; /* uint32 like the reference counter */
				const u64 mb = __ballot(c < P.cutoff_min) & m;
				const u64 ma = __ballot(c > P.cutoff_max) & m & ~mb;
				const u64 mc = m & ~mb & ~ma;
				cnt[r] = c > P.counter_max ? P.counter_max : c;
				rk[r] = ((mc >> lane) & 1ull) ? __builtin_amdgcn_mbcnt_hi((u32)(mc >> 32), __builtin_amdgcn_mbcnt_lo((u32)mc, nc)) : NONE;
				nu += (u32)__popcll(m);
				nb += (u32)__popcll(mb);
				na += (u32)__popcll(ma);
				nc += (u32)__popcll(mc);
				if (m)
					carry = rowrel + 63 - (u32)__clzll((long long)m);
			}
			if (lane == 0)
				s_wcnt[wave] = nc;
			__syncthreads();
			u32 wave_off = 0, chunk_counted = 0;
#pragma unroll
			for (int x = 0; x < NW; ++x) {
				const u32 t = s_wcnt[x];
				if (x < (int)wave)
					wave_off += t;
				chunk_counted += t;
			}
			if (!P.without_output) {
#pragma unroll
				for (int r = 0; r < ITEMS; ++r) {
					if (rk[r] != NONE) {
						const u64 kx[1] = {(khi | (key[r] & rmask)) & kmask};
						kmc_emit_record<1>(span + (size_t)(counted_total + wave_off + rk[r]) * rec_bytes, kx, cnt[r], P.sbytes, P.cbytes, P.kff != 0);
						if (use_lut && !lut_once)
							atomicAdd(&lut[(u32)kmc_remove_suffix<1>(kx, pshift) & lut_mask], 1ull);
					}
				}
			}
			counted_total += chunk_counted;
			if (chunk_last != NONE)
				prev_tail = c0 + chunk_last;
			__syncthreads(); /* s_wlast / s_wcnt are rewritten by the next chunk */
		}
		if (lane == 0) { /* nu, nb, na are per wave */
			s_tal[wave * 3 + 0] = nu;
			s_tal[wave * 3 + 1] = nb;
			s_tal[wave * 3 + 2] = na;
		}
		__syncthreads();
		if (tid == 0) {
			u32 tu = 0, tb = 0, ta = 0;
#pragma unroll
			for (int x = 0; x < NW; ++x) {
				tu += s_tal[x * 3 + 0];
				tb += s_tal[x * 3 + 1];
				ta += s_tal[x * 3 + 2];
			}
			u64 *sh = gr.tally[bin] + (size_t)(slot_id % CP_SHARDS) * 4;
			if (tu)
				atomicAdd(&sh[0], (u64)tu);
			if (tb)
				atomicAdd(&sh[1], (u64)tb);
			if (ta)
				atomicAdd(&sh[2], (u64)ta);
			if (!P.without_output && counted_total) {
				gr.status[bin][slot_id] = counted_total;
				gr.chunk_src[bin][slot_id] = pos + s0;
				if (use_lut && lut_once) {
					const u64 kx[1] = {khi & kmask};
					atomicAdd(&lut[(u32)kmc_remove_suffix<1>(kx, pshift) & lut_mask], (u64)counted_total);
				}
			}
			if (seg == 0) { /* statistics: the buckets and records that took this road (the stream's error block, as k_giant_tiles) */
				atomicAdd(&err[12], 1u);
				atomicAdd(reinterpret_cast<u64 *>(err + 14), (u64)len);
			}
		}
		} /* segments */
	} /* entries */
}

#endif
