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
 *   k_bucket_rank     one workgroup per tile: every record finds its place inside its bucket by counting the records of the bucket below it (pairwise, in LDS),
 *                     and the ordered tile is counted where it lies — run lengths, cutoffs, (suffix, counter) records, LUT, tallies (below: "rank inside the
 *                     bucket, count in place"). 8 SIZE bytes read per record, once, whatever k; the sorted tile never goes back to HBM.
 *   k_giant_tiles     a tile whose largest bucket does not fit LDS (records of two words and more; one-word records: arena_sort.hip.h).
 *
 * (Rounds 3-5 also shipped k_bucket_sort — sub-buckets finished by one thread each, sort-only — and k_bucket_count — tiles counted through an LDS hash table without
 * sorting; neither was on a default path after round 4 and both left the library in round 6: profiles/r06/experiments/removed_finishers.patch.)
 *
 * This replaces what the reference does below its first radix levels — RadulsSort's recursion into small buckets and the
 * insertion / shell sorts of CSmallSort (raduls_impl.h:133-141,497-510; small_sort.h:29-179) — with an LDS-resident equivalent.
 *
 * What the kernels cannot take (a bucket beyond GT_MAX_RECORDS of records of two words and more, an arena that overflows) sets `*flag`: the host then sorts that
 * bin or group again with LSD passes over all bytes (host_plan_and_groups.hip.h: "redo"). Nothing is ever left partially sorted silently.
 *
 * Contract (SortFunction, kb_sorter.h:761-775): the bytes of a record above the key are zero, so comparing whole records
 * orders them by the key.
 */
#ifndef KMC_AMD_BUCKET_SORT_HIP_H
#define KMC_AMD_BUCKET_SORT_HIP_H

#include "kernels.hip.h"

#ifndef BR_NARROW_LOOP
#define BR_NARROW_LOOP 2 /* k_bucket_rank's walk over 32-bit pairs: 2 = steps of 8 / 4 / one masked step (shipped); 0 = the walk of rounds 3-5 (tools/build_variants.py brnl0) */
#endif
#ifndef BR_WIDE_LOOP
#define BR_WIDE_LOOP 1 /* k_bucket_rank<1>'s walk over 64-bit pairs: 1 = steps of BR_WIDE_STEP, 4, 2, 1 (shipped); 0 = steps of 2 + one (tools/build_variants.py brwl0) */
#endif
#ifndef BR_WIDE_STEP
#define BR_WIDE_STEP 8
#endif
#ifndef BR_SLACK_DIV
#define BR_SLACK_DIV 12 /* k_bucket_rank: windows are 11/12 of a tile's capacity, the rest is room for the bucket that is open at the end of the window (waves
                         * without records still walk through the kernel: windows of 2/3 32.4, of 5/6 33.4, of 11/12 33.5 Gk-mers/s on the quarter workload).
                         * A tile that outgrows the capacity is taken in two chunks of whole buckets */
#endif
#ifndef BR_INDIRECT_MIN_SIZE
#define BR_INDIRECT_MIN_SIZE 2 /* = INDIRECT_MIN_WORDS of kmc_hip.hip at most */
#endif
#ifndef BR_MIN_WAVES
#define BR_MIN_WAVES 6 /* waves per SIMD the register allocator must leave room for: two workgroups of 12 waves per CU (<= 80 VGPRs; it takes 62) */
#endif
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

/* The arrays k_bucket_bounds cuts into tiles: the slices of up to GRP_MAX bins (a sort-only call: one array). Bin b has items
 * item_prefix[b] .. item_prefix[b+1]-1 = its windows 0 .. n_win, one more than it has windows (bounds[n_win] = n). */
struct GrpBounds {
	u32 g, item_prefix[GRP_MAX + 1];
	const u64 *S[GRP_MAX];
	u64 n[GRP_MAX];
	u64 *bounds[GRP_MAX];
};

/* One wave per window j (0..n_win) of a bin: bounds[j] = the first index i >= j stride at which a bucket starts (i == 0, or the top `hbits`
 * bits of record i differ from those of record i-1), n if there is none. hbits == 0: the whole array is one bucket. */
template <int SIZE>
__global__ void __launch_bounds__(256) k_bucket_bounds(const GrpBounds gb, u32 stride, u32 key_bits, u32 hbits)
{
	const u32 lane = threadIdx.x & 63;
	const u32 item = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (item >= gb.item_prefix[gb.g])
		return;
	const u32 bin = grp_find(gb.item_prefix, gb.g, item);
	const u64 j = item - gb.item_prefix[bin];
	const u64 *__restrict__ recs = gb.S[bin];
	const u64 n = gb.n[bin];
	const u64 p = j * stride;
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
				const u64 f = (u64)(__ffsll((long long)m) - 1); /* the first probe in a different bucket: the answer is at or below it, above the probe before it */
				hi = lo + f * step;
				lo = f ? hi - step + 1 : hi;
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
		gb.bounds[bin][j] = b;
}

/* ------------------------------------------------------------------------------------------------ rank inside the bucket, count in place
 * At sequencing depth a bucket is the ~30 copies of one k-mer plus a few one-off neighbours (read errors below the bucket bits). Round 3's first finisher gave
 * every sub-bucket to ONE thread, which walked those records serially while 29 of 30 threads had nothing to do. Here every record finds its own place, all at once:
 *       place(i) = bucket start + #{ j in the bucket : (rem_j, j) < (rem_i, i) }        rem = the key bits below the bucket bits
 * A step of the count is an LDS read (the lanes of a bucket read the same address: a broadcast), a compare and an add-with-carry; the wave takes as
 * many steps as its largest bucket has records (the bench's bins with the top 30 key bits ordered: 25 on average). Stable (ties go by index), no
 * atomics, no data-dependent failure: the only thing it cannot take is a bucket larger than the tile (k_giant_tiles / the arena). Work grows with sum(bucket^2): the host asks for enough HBM passes to keep buckets at a few dozen records (plan_sort).
 * What is compared, by record width:
 *   one word     (rem, bucket-relative index) in ONE word: 32 bits whenever the tile's largest bucket leaves room for its indices next to rem (always, on
 *                sequencing data with 24 key bits left), else 64 with the index in the low 16 (host: key_bits - hbits <= 48)
 *   two words    rem <= 80 bits (host): A = rem >> 16 (64 bits), B = (rem & 0xFFFF) << 16 | index (32 bits); less = A' < A or (A' == A and B' < B)
 *   wider        the records themselves, arrival order in LDS, lexicographic from the top word (inside a bucket the bucket bits are equal), ties by index
 * FUSED (round 4): the tile, now in order inside LDS and made of whole buckets — no run of equal k-mers leaves it —, is counted where it lies: run tails,
 * counts, cutoffs, clamp, ranks of the counted k-mers exactly as k_compact does on rows of 64 records (kb_sorter.h:1128-1281), the (suffix, counter)
 * records assembled in LDS and written, coalesced, to the tile's span of the free record array (two-phase output: k_compact_fold turns the tiles' counts
 * into offsets, k_compact_gather moves the records), LUT and tallies sharded as in k_compact. The sorted tile never goes back to HBM: 8 SIZE bytes per
 * record are read once, ~0.3 bytes per record are written. Not FUSED: the sorted tile is written back in place (k_compact follows; parameter sets
 * whose records may outgrow a span). */
#ifndef BR_THREADS
#define BR_THREADS 768 /* 12 waves; two workgroups per CU (one-word records: 48 KB of pairs / records + 24 KB of bucket starts each) */
#endif
/* Buckets beyond BR_BIG records are not ranked by their own records' walks: a record's walk is as long as its bucket, a bucket sits in one or two waves, and the
 * workgroup waits for its slowest wave — on repeat-rich input a bucket of a few hundred records in 40 % of the tiles made the rank loops 2.7 x as long as on
 * uniform reads although the pair work per record was the same 20-25 (profiles/r05: bucket_hist_*.txt, experiments). Their records are dealt out again
 * over ALL waves (k_bucket_rank: "big buckets"). 30x reads of a random genome have no bucket beyond 128. */
#ifndef BR_BIG
#define BR_BIG 128
#endif
template <int SIZE> struct BrCfg {
	static constexpr int THREADS = BR_THREADS;
	static constexpr int ITEMS = SIZE == 1 ? 8 : (SIZE == 2 ? 4 : 2); /* rows of 64 records per wave */
	static constexpr int CAP = THREADS * ITEMS;                        /* records a tile (or a chunk of one) may hold: 6144 / 3072 / 1536 */
	static constexpr int STRIDE = CAP - CAP / BR_SLACK_DIV;            /* window length */
	static constexpr int NW = THREADS / 64;
	static constexpr size_t R0 = (size_t)CAP * SIZE * 8;    /* the pairs; then the records in order; then the staged output */
	static constexpr size_t R1 = ((size_t)CAP + 2) * 4;     /* the table of bucket starts; then LUT prefixes / counts of the staged records */
	static constexpr size_t TRAILER_WORDS = 10 * NW + 16;   /* s_wfirst, s_wmax, s_wlast, s_wcnt [NW each], s_wtal [5 NW + 8], s_nbig (the
	                                                         * trailer had outgrown the 7 NW + 8 words reserved for it and lived on the allocation granule) */
	static constexpr size_t LDS = R0 + R1 + TRAILER_WORDS * 4 + 16;
	static_assert(CAP < 65536, "tile-relative positions are kept in 16 bits");
};
template <int SIZE> constexpr size_t br_lds_bytes() { return BrCfg<SIZE>::LDS; }
/* key bits that may be left below the bucket bits (plan_sort) */
/* rank += ((x[NW-1] .. x[0], xi) < (y[NW-1] .. y[0], yi)), both read as ONE number of NW + 1 dwords with xi / yi the least significant: a borrow chain — one
 * subtract-with-borrow per dword and an add-with-carry at the end, the borrow in a scalar register pair. Written as `less || (equal && index before)` over 64-bit
 * words the compiler emits 64-bit compares under per-lane branches (k = 55: ~11 vector instructions + two branches per pair; a chain written with
 * __builtin_subc is folded back into the same compares), and the pair loops are where k_bucket_rank<2..> spends its time. */
template <int NW> __device__ __forceinline__ void br_rank_add_less(u32 &rank, const u32 (&x)[NW], u32 xi, const u32 (&y)[NW], u32 yi)
{
#if defined(__HIP_DEVICE_COMPILE__)
	u32 t;
	u64 c;
	asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(t), "=s"(c) : "v"(xi), "v"(yi));
#pragma unroll
	for (int i = 0; i < NW; ++i)
		asm("v_subb_co_u32_e64 %0, %1, %2, %3, %1" : "=v"(t), "+s"(c) : "v"(x[i]), "v"(y[i]));
	asm("v_addc_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(rank), "+s"(c));
#else /* tests/hipemu */
	bool less = xi < yi;
	for (int i = 0; i < NW; ++i)
		less = x[i] < y[i] || (x[i] == y[i] && less);
	rank += less ? 1u : 0u;
#endif
}
template <int SIZE> constexpr u32 br_rem_limit() { return SIZE == 1 ? 48u : (SIZE == 2 ? 80u : 64u * SIZE); }

/* The tiles of up to GRP_MAX bins (not FUSED: one array, g = 1). Tile t of bin b = [bounds[b][t], bounds[b][t+1]); one that outgrows the capacity is taken
 * in two chunks of whole buckets by its two workgroups (blockIdx.y). Chunk y of tile t reports into slot 2 t + y. */
struct GrpRank {
	u32 g, win_prefix[GRP_MAX + 1];
	u64 *S[GRP_MAX];            /* the bin's slice of the record array, ordered by the top `hbits` key bits */
	const u64 *bounds[GRP_MAX]; /* [windows + 1] (k_bucket_bounds) */
	uint8_t *scratch[GRP_MAX];  /* FUSED: the bin's slice of the free record array: a chunk's records go to scratch + (its first record) * 8 SIZE */
	u64 *status[GRP_MAX];       /* FUSED: [2 windows + 1] counted k-mers of every chunk (zeroed by the host) */
	u64 *chunk_src[GRP_MAX];    /* FUSED: [2 windows] first record of every chunk that counted something (the gather's source) */
	u64 *lut_base[GRP_MAX];
	u64 *tally[GRP_MAX];        /* [CP_SHARDS][4] */
	u32 *giant;                 /* FUSED: [0] number of tiles handed to k_giant_tiles, [1] how many of them have been taken, [2..] their numbers (zeroed by the host) */
	const u64 *rec_base;        /* indirect sort (records of three words and more): S is the ordered array of (top four key bytes << 32 | record number) PAIRS, one word
	                             * each, and the records stay where k_expand wrote them: record number i of the group is rec_base + i SIZE. NULL: S holds the records. */
	u64 *giant_T[GRP_MAX];      /* indirect sort: the bin's slice of a third record array — k_giant_tiles gathers a tile's records there before it sorts them in place */
	/* one-word records, FUSED (round 6, arena_sort.hip.h): the group's repeat-rich buckets are sorted together by HBM passes of their own. NULL: as in round 5. */
	u32 *arena_dyn;             /* AR_* words (zeroed by the host) */
	struct ArenaEntry *arena_ent; /* [arena_cap] the listed buckets */
	u32 arena_cap;
};
/* A bucket the arena sorts. kind 0: a GIANT bucket (beyond a tile's capacity, any length): sorted and counted from the arena, segment by segment, into the output slots of the
 * windows it covers. kind 1: a bucket of BR_MID < records <= capacity inside a tile: sorted in the arena, written back in place BEFORE its tile is ranked. */
struct ArenaEntry {
	u64 w0;    /* [39:0] first record (its index in the group's ordered array: gr.S[0] + ...), [43:40] the bin's number in the group, [44] kind */
	u32 len;   /* records */
	u32 gtile; /* kind 0: the tile (group-wide number) the bucket starts in */
};
constexpr u32 AR_N_ENT = 0, AR_M = 1, AR_N_PASS = 2, AR_N_ITEMS = 3, AR_OVERFLOW = 5, AR_PASS_TICKET = 8 /* .. 15 */,
              AR_STAT_MID_N = 20, AR_STAT_MID_REC = 22 /* u64: words 22-23 */, AR_DYN_WORDS = 32;
#ifndef BR_MID
#define BR_MID 351 /* one-word records: a bucket beyond this many records is not ranked pairwise at all (the work grows with its square: 0.07 ps x n^2 against ~40 ps x n of
                    * arena passes) — it goes through the arena and comes back in order */
#endif
#ifndef BD_STRIDE_N
#define BD_STRIDE_N 176
#endif
constexpr u32 BD_STRIDE = BD_STRIDE_N; /* k_bucket_detect looks at every 176th record: two neighbouring samples with one bucket number = a bucket of 177+ records; a bucket of 2 x 176
                               * records and more cannot hide between the samples. Every sample is a 64-byte sector of its own: stride 88 (BR_MID 255) cost 83 us per group of 190 M
                               * records with nothing to find, and the spectrum leg ran no faster for it (23.8 against 24.0 Gk-mers/s with BR_MID 383) */
static_assert(BR_MID >= BR_BIG, "buckets the arena takes are a subset of the big ones");
static_assert(BR_MID + 1 >= 2 * BD_STRIDE, "every bucket beyond BR_MID records must show in two neighbouring samples");
/* A tile whose LARGEST BUCKET does not fit the capacity — one k-mer repeated thousands of times: every genome has those — is not ranked pairwise (the work
 * grows with the square of a bucket) and, since round 4, no longer sends its whole group back to the host either: k_bucket_rank puts it on a list and
 * k_giant_tiles sorts it on its own (below). Only a tile beyond GT_MAX_RECORDS still raises the group's flag (-> the host's LSD passes). */
#ifndef GT_THREADS
#define GT_THREADS 1024 /* 16 waves; chunks of 8192 one-word records (4096 / 2048 of wider ones): fewer barrier-separated rounds per tile than 512 x 4 (skew leg 24.1 -> see DESIGN) */
#endif
#ifndef GT_MAX_RECORDS_LOG2
#define GT_MAX_RECORDS_LOG2 20 /* one workgroup sorts up to a million records by itself (~1-2 ms); beyond that the group goes back to the host */
#endif
constexpr u64 GT_MAX_RECORDS = 1ull << GT_MAX_RECORDS_LOG2;
constexpr u32 GT_CUT_SHIFT = 19; /* a list entry of k_giant_tiles: [18:0] the tile's number in the group, [31:19] the records in front of the giant part (< 8192) */

/* chunk `by` (0, 1) of tile `gtile`. With the arena (one-word records, fused): the buckets beyond BR_MID records have been found (k_bucket_detect), sorted by the arena's passes
 * and written back BEFORE this kernel runs — they are in order where they lie and keep their places; giant buckets are counted from the arena and only cut off here. */
template <int SIZE, bool FUSED>
__device__ __forceinline__ void br_tile(const GrpRank &gr, const DevParams &P, const u32 key_bits, const u32 hbits, const u32 lut_shards, const u64 lut_stride, const u32 lut_mask,
                                        u32 *flag, const u32 gtile, const u32 by, unsigned char *s_raw)
{
	constexpr int THREADS = BrCfg<SIZE>::THREADS, ITEMS = BrCfg<SIZE>::ITEMS, CAP = BrCfg<SIZE>::CAP, NW = BrCfg<SIZE>::NW;
	constexpr u64 S = BrCfg<SIZE>::STRIDE;
	constexpr u32 NONE = 0xFFFFFFFFu;
	const bool arena = SIZE == 1 && FUSED && gr.arena_dyn != nullptr;
	u64 *s_key = reinterpret_cast<u64 *>(s_raw);                              /* R0 */
	u32 *s_start = reinterpret_cast<u32 *>(s_raw + BrCfg<SIZE>::R0);          /* R1 [CAP + 2] */
	u32 *s_wfirst = s_start + CAP + 2;                                        /* [NW] bucket starts in wave w's rows */
	u32 *s_wmax = s_wfirst + NW;                                              /* [NW] largest bucket a wave has seen; [0]: the cut of a long tile */
	u32 *s_wlast = s_wmax + NW;                                               /* [NW] FUSED: the last run tail inside wave w's rows */
	u32 *s_wcnt = s_wlast + NW;                                               /* [NW] FUSED: counted k-mers of wave w */
	u32 *s_wtal = s_wcnt + NW;                                                /* [NW][3] FUSED: distinct / below min / above max of wave w */
	u32 *s_nbig = s_wtal + 5 * NW + 8;                                        /* [1] != 0: the tile has a bucket beyond BR_BIG records */
	static_assert(9 * NW + 8 + 1 <= (int)BrCfg<SIZE>::TRAILER_WORDS, "the trailer arrays lie inside the dynamic LDS the launch asks for");

	const u32 bin = (u32)__builtin_amdgcn_readfirstlane((int)grp_find(gr.win_prefix, gr.g, gtile));
	const u32 tile = gtile - gr.win_prefix[bin];
	const u64 *__restrict__ bounds = gr.bounds[bin];
	u64 *__restrict__ recs = gr.S[bin];
	const u64 b0 = bounds[tile], b1 = bounds[tile + 1];
	if (b0 >= ((u64)tile + 1) * S || b0 >= b1)
		return; /* no bucket starts in this window */
	if (arena && gr.arena_dyn[AR_OVERFLOW])
		return; /* the arena's plan was dropped (k_arena_plan: the group's flag is up): the buckets beyond BR_MID records are NOT in order — nothing may be reported */
	const u32 tid = threadIdx.x, lane = tid & 63;
	const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const u32 crel = wave * (ITEMS * 64); /* the wave owns ITEMS rows of 64 consecutive records */
	if (tid == 0)
		*s_nbig = 0; /* (the barrier behind the bucket starts orders it before the heads' atomics) */
	const u32 bsh = 64 - hbits;
	auto bucket_of = [&](const u64(&x)[SIZE]) -> u64 { return hbits ? bs_p64<SIZE>(x, key_bits) >> bsh : 0ull; };
	const u64 *__restrict__ rec_base = SIZE >= BR_INDIRECT_MIN_SIZE ? gr.rec_base : nullptr; /* indirect: `recs` are pairs (never for one-word records: folded away at compile time) */
	auto bucket_at = [&](u64 i) -> u64 {
		if (rec_base)
			return recs[i] >> 32; /* the top four key bytes ARE the bucket number: the bits above the key in the top byte are zero (kmc_hip.hip: hbits = 32 - those bits) */
		u64 x[SIZE];
		load_rec<SIZE>(recs + i * SIZE, x);
		return bucket_of(x);
	};
	/* A tile longer than the capacity (windows are nearly as long as the capacity: BR_SLACK_DIV) is taken in two chunks of whole buckets, by the two
	 * workgroups (by = 0, 1) every tile has: the first takes the records up to the last bucket start inside the capacity, the second the rest.
	 * Nearly every tile fits, and its second workgroup returns at once. */
	u64 c0 = b0;
	u32 len;
	if (b1 - b0 <= (u64)CAP) {
		if (by)
			return;
		len = (u32)(b1 - b0);
	} else {
		if (tid == 0)
			*s_wmax = 0;
		__syncthreads();
		for (u32 idx = tid + 1; idx <= (u32)CAP; idx += THREADS) /* b0 + CAP < b1 */
			if (bucket_at(b0 + idx) != bucket_at(b0 + idx - 1))
				atomicMax(s_wmax, idx);
		__syncthreads();
		const u32 cut = *s_wmax;
		__syncthreads();
		if (cut == 0 || (b1 - b0) - cut > (u64)CAP) {
			/* one bucket (or two) beyond the capacity. Round 5: only what lies BEHIND the cut — the giant bucket itself: nothing else starts in the window once a
			 * bucket longer than the capacity has — goes to k_giant_tiles (list entry: tile | cut << GT_CUT_SHIFT; it reports into the tile's second slot); the
			 * buckets in front of it are an ordinary first chunk and stay here. k_giant_tiles then sorts ONE bucket: only the bits below the bucket bits (three
			 * passes at k = 27 instead of six over the whole tile's key range). Not fused, or beyond GT_MAX_RECORDS: the host's LSD passes, as ever. */
			/* Round 6, one-word records with the arena: the giant bucket — of ANY length — is an entry of the group's arena already (k_bucket_detect listed it; sorted by HBM
			 * passes over all the listed buckets at once, counted segment by segment by the whole GPU: arena_sort.hip.h); no bin comes back for a satellite any more. */
			const u64 glen = (b1 - b0) - cut;
			const bool listed = FUSED && (arena || (gr.giant && glen <= GT_MAX_RECORDS));
			if (tid == 0 && by == 0 && !arena) {
				if (listed)
					gr.giant[2 + atomicAdd(&gr.giant[0], 1u)] = gtile | (cut << GT_CUT_SHIFT);
				else
					atomicOr(flag, FUSED ? 0x10000u << bin : 1u); /* fused: only this BIN comes back (bits 16 + its number in the group); in place: the group */
			}
			if (!listed || cut == 0 || by != 0)
				return;
			len = cut; /* by == 0: the buckets in front of the giant one */
		} else if (by == 0)
			len = cut;
		else {
			c0 = b0 + cut;
			len = (u32)(b1 - c0);
		}
	}
	u64 *__restrict__ T = recs + c0 * (rec_base ? 1 : SIZE);

	u64 key[ITEMS][SIZE];
	if (rec_base) { /* a gather of whole records by number (16+ bytes each: one or two HBM sectors). Every row's pair first, then every row's record: two round trips
	                 * to HBM — written row by row, each row's pair load waited for the row before (eight round trips per tile of two-word records) */
		u32 number[ITEMS];
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = crel + r * 64 + lane;
			number[r] = (u32)T[idx < len ? idx : 0u]; /* (no branch around the load: the compiler waits for a load at the end of its branch) */
		}
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = crel + r * 64 + lane;
			load_rec<SIZE>(rec_base + (size_t)number[r] * SIZE, key[r]); /* rows behind the tile's end read record 0 and are cleared below */
			if (idx >= len) {
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					key[r][w] = 0;
			}
		}
	} else {
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = crel + r * 64 + lane;
			if (idx < len)
				load_rec<SIZE>(T + (size_t)idx * SIZE, key[r]);
			else {
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					key[r][w] = 0;
			}
		}
	}
	/* Bucket starts ("heads") and, from them, every record's bucket: a record's bucket is known by its ORDINAL among the tile's
	 * buckets (heads at or before the record - 1: a popcount below the lane + the heads of the rows and waves before); the start positions go into a
	 * table indexed by that ordinal, and a record finds the start and the end of its bucket with two LDS reads. The predecessor's bucket comes through
	 * a DPP wave shift, not through the LDS crossbar. */
	u64 prev_b = 0;
	if (crel > 0 && crel - 1 < len)
		prev_b = bucket_at(c0 + crel - 1);
	u32 headbits = 0, below[ITEMS], wheads = 0;
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = crel + r * 64 + lane;
		const u64 bk = bucket_of(key[r]);
		const u32 plo = wave_shift_up1((u32)bk, (u32)prev_b, lane), phi = wave_shift_up1((u32)(bk >> 32), (u32)(prev_b >> 32), lane);
		const bool head = idx < len && (idx == 0 || (((u64)phi << 32) | plo) != bk);
		const u64 m = __ballot(head);
		headbits |= head ? 1u << r : 0u;
		below[r] = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, wheads));
		wheads += (u32)__popcll(m);
		prev_b = __shfl(bk, 63);
	}
	if (lane == 0)
		s_wfirst[wave] = wheads;
	__syncthreads();
	u32 wave_heads_before, total_heads;
	{
		const u32 v = lane < (u32)NW ? s_wfirst[lane] : 0u;
		const u32 inc = wave_incl_sum_u32(v, lane);
		total_heads = __shfl(inc, NW - 1);
		wave_heads_before = __shfl(inc - v, (int)wave);
	}
#pragma unroll
	for (int r = 0; r < ITEMS; ++r)
		if ((headbits >> r) & 1u)
			s_start[wave_heads_before + below[r]] = crel + r * 64 + lane;
	if (tid == 0)
		s_start[total_heads] = len;
	__syncthreads();
	u32 span[ITEMS], rel[ITEMS], widest = 0; /* span: [15:0] start of the record's bucket, [31:16] its end (tile-relative; CAP < 65536) */
	bool any_big = false;
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = crel + r * 64 + lane;
		span[r] = 0; /* nothing to count */
		rel[r] = 0;
		if (idx < len) {
			const u32 ord = wave_heads_before + below[r] + ((headbits >> r) & 1u) - 1u;
			const u32 bstart = s_start[ord], bend = s_start[ord + 1];
			if (arena && bend - bstart > (u32)BR_MID) {
				/* a bucket the arena has put in order (k_bucket_detect, arena_sort.hip.h): the record lies where it belongs. An empty span at its own position: nothing is
				 * walked, place = position; the bucket does not count for the width of the pairs either */
				span[r] = idx | (idx << 16);
			} else {
				span[r] = bstart | (bend << 16);
				rel[r] = idx - bstart;
				widest = widest > bend - bstart ? widest : bend - bstart;
				if (bend - bstart > (u32)BR_BIG)
					any_big = true;
			}
		}
	}
	if (any_big)
		atomicOr(s_nbig, 1u); /* (no value returned: one LDS OR; the barrier behind the pair stores publishes it) */
	/* Big buckets (beyond BR_BIG records): their records change hands. A record's walk is as long as its bucket and a bucket lies in one or two waves, so the
	 * owners' walks made the whole workgroup wait for those waves. Instead the big records are dealt out again by their POSITION — record idx to thread idx mod
	 * THREADS: a bucket's consecutive records to consecutive threads of several waves — each thread walks the bucket of the records it was dealt, and the owners
	 * read the ranks back. R1 carries both ways (the bucket table is dead once the spans are in registers): the owner leaves its span there with bit 31 set
	 * (table entries are positions: bit 31 clear), the dealer puts the rank in its place. No list, no atomics: the first version registered the buckets through a
	 * returning LDS atomic and lost entries on the device (never under the emulation; profiles/r05/experiments). `less(q, i)`: is the pair at tile position q in
	 * front of the pair at i — given as `count(bs, be, i)`: how many pairs of [bs, be) are in front of the pair at i (the one-word forms read eight pairs per step: a walk
	 * of one load, one compare and one add at a time is bound by the LDS latency, and only the few waves the bucket was dealt to are running). Called by all threads,
	 * after the barrier behind the pair stores. */
	u32 place[ITEMS];
	u32 *s_rank = s_start;
	auto is_big = [&](int r) -> bool { return (span[r] >> 16) - (span[r] & 0xFFFFu) > (u32)BR_BIG; };
	auto rank_big_buckets = [&](auto count) {
		if (!*s_nbig) /* uniform */
			return;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r)
			if (crel + r * 64 + lane < len) /* every slot below len is written: what the table or an earlier workgroup left there must not be taken for a span */
				s_rank[crel + r * 64 + lane] = is_big(r) ? span[r] | 0x80000000u : 0u;
		__syncthreads();
#pragma unroll 1
		for (int m = 0; m < ITEMS; ++m) {
			const u32 i = (u32)m * THREADS + tid;
			const u32 v = i < len ? s_rank[i] : 0u;
			if (v >> 31) {
				s_rank[i] = count(v & 0xFFFFu, (v >> 16) & 0x7FFFu, i); /* read and written by this thread only */
			}
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < ITEMS; ++r)
			if (is_big(r))
				place[r] = (span[r] & 0xFFFFu) + s_rank[crel + r * 64 + lane];
	};
#if defined(BR_STOP_AFTER) && BR_STOP_AFTER <= 1 /* tuning builds only: what does each phase cost? (the output is garbage) */
	if (widest != 0x7FFFFFFFu)
		return;
#endif
	const u32 rbits = key_bits - hbits; /* <= br_rem_limit (host) */
	if constexpr (SIZE == 1) {
		const u64 rmask = rbits >= 64 ? ~0ull : ((1ull << rbits) - 1);
		/* the largest bucket of the tile decides the width of the pairs: (rem, index) in 32 bits whenever they fit — half the LDS traffic and one-pass compares */
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) {
			const u32 other = (u32)__shfl((int)widest, (int)(lane ^ (u32)o));
			widest = widest > other ? widest : other;
		}
		if (lane == 0)
			s_wmax[wave] = widest;
		__syncthreads();
#pragma unroll
		for (int w = 0; w < NW; ++w)
			widest = widest > s_wmax[w] ? widest : s_wmax[w];
		const bool narrow = rbits < 32 && widest <= (1u << (32 - rbits));
		if (narrow) {
			u32 *s_k32 = reinterpret_cast<u32 *>(s_key);
			const u32 sh = 32 - rbits;
			u32 c32[ITEMS];
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 idx = crel + r * 64 + lane;
				c32[r] = ((u32)(key[r][0] & rmask) << sh) | rel[r]; /* rel < widest <= 2^sh */
				if (idx < len)
					s_k32[idx] = c32[r];
			}
			__syncthreads();
			/* How many pairs of [q, bend) are in front of the pair c. Steps of 8, one step of 4, and ONE masked step for the last 1..3 pairs (the slots behind the
			 * bucket's end are read and not counted) — spelled out and kept from the loop optimiser: what it made of `for (; q + 4 <= bend; q += 4)` + a pairwise
			 * tail was a 16-pair body with an 8-pair, a 4-pair and a pairwise epilogue, each entered by every wave with one lane in need of it, and a tail of up to
			 * three DEPENDENT LDS round trips per row. Round 5, sessions x / y: 1.53-1.57 -> 1.39-1.41 ms per group of 190 M records, 35.7-36.0 -> 36.3-36.9 Gk-mers/s
			 * on the quarter workload (profiles/r05/experiments/README.md 10; -DBR_NARROW_LOOP=0 builds the old walk). */
			auto walk32 = [&](u32 q, const u32 bend, const u32 c) -> u32 {
				u32 rank = 0;
#if BR_NARROW_LOOP == 0
				for (; q + 4 <= bend; q += 4) {
					const u32 a0 = s_k32[q], a1 = s_k32[q + 1], a2 = s_k32[q + 2], a3 = s_k32[q + 3];
					rank += (a0 < c ? 1u : 0u) + (a1 < c ? 1u : 0u) + (a2 < c ? 1u : 0u) + (a3 < c ? 1u : 0u);
				}
				for (; q < bend; ++q)
					rank += s_k32[q] < c ? 1u : 0u;
#else
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
				for (; q + 8 <= bend; q += 8) {
					u32 a[8];
#pragma unroll
					for (int u = 0; u < 8; ++u)
						a[u] = s_k32[q + u];
#pragma unroll
					for (int u = 0; u < 8; ++u)
						rank += a[u] < c ? 1u : 0u;
				}
				if (q + 4 <= bend) {
					const u32 a0 = s_k32[q], a1 = s_k32[q + 1], a2 = s_k32[q + 2], a3 = s_k32[q + 3];
					rank += (a0 < c ? 1u : 0u) + (a1 < c ? 1u : 0u) + (a2 < c ? 1u : 0u) + (a3 < c ? 1u : 0u);
					q += 4;
				}
				if (q < bend) { /* one to three pairs are left */
					const u32 a0 = s_k32[q], a1 = s_k32[q + 1], a2 = s_k32[q + 2];
					rank += (a0 < c ? 1u : 0u) + ((q + 1 < bend) & (a1 < c) ? 1u : 0u) + ((q + 2 < bend) & (a2 < c) ? 1u : 0u);
				}
#endif
				return rank;
			};
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 bstart = span[r] & 0xFFFFu, bend = span[r] >> 16;
				place[r] = bstart + walk32(is_big(r) ? bend /* ranked by the whole workgroup below */ : bstart, bend, c32[r]);
			}
			rank_big_buckets([&](u32 bs, u32 be, u32 i) { return walk32(bs, be, s_k32[i]); });
		} else {
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 idx = crel + r * 64 + lane;
				if (idx < len)
					s_key[idx] = ((key[r][0] & rmask) << 16) | (u64)rel[r];
			}
			__syncthreads();
			/* walk32's shape for 64-bit pairs (k = 28..32, and any tile of narrower records with a bucket beyond what a 32-bit pair numbers: the repeat-rich legs): steps of
			 * BR_WIDE_STEP, halved down to one — no dependent round trip per two pairs, no interleaving by the compiler. -DBR_WIDE_LOOP=0: steps of 2 + one (rounds 3-5). */
			auto walk64 = [&](u32 q, const u32 bend, const u64 c) -> u32 {
				u32 rank = 0;
#if BR_WIDE_LOOP == 0
				for (; q + 2 <= bend; q += 2) {
					const u64 a = s_key[q], b = s_key[q + 1];
					rank += (a < c ? 1u : 0u) + (b < c ? 1u : 0u);
				}
#else
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
				for (; q + BR_WIDE_STEP <= bend; q += BR_WIDE_STEP) {
					u64 a[BR_WIDE_STEP];
#pragma unroll
					for (int u = 0; u < BR_WIDE_STEP; ++u)
						a[u] = s_key[q + u];
#pragma unroll
					for (int u = 0; u < BR_WIDE_STEP; ++u)
						rank += a[u] < c ? 1u : 0u;
				}
#if BR_WIDE_STEP > 4
				if (q + 4 <= bend) {
					const u64 a0 = s_key[q], a1 = s_key[q + 1], a2 = s_key[q + 2], a3 = s_key[q + 3];
					rank += (a0 < c ? 1u : 0u) + (a1 < c ? 1u : 0u) + (a2 < c ? 1u : 0u) + (a3 < c ? 1u : 0u);
					q += 4;
				}
#endif
				if (q + 2 <= bend) {
					const u64 a = s_key[q], b = s_key[q + 1];
					rank += (a < c ? 1u : 0u) + (b < c ? 1u : 0u);
					q += 2;
				}
#endif
				if (q < bend)
					rank += s_key[q] < c ? 1u : 0u;
				return rank;
			};
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 bstart = span[r] & 0xFFFFu, bend = span[r] >> 16;
				place[r] = bstart + walk64(is_big(r) ? bend : bstart, bend, ((key[r][0] & rmask) << 16) | (u64)rel[r]);
			}
			rank_big_buckets([&](u32 bs, u32 be, u32 i) { return walk64(bs, be, s_key[i]); });
		}
	} else if constexpr (SIZE == 2) {
		u64 *s_A = s_key;                                     /* [CAP] rem >> 16 */
		u32 *s_B = reinterpret_cast<u32 *>(s_key + CAP);      /* [CAP] (rem & 0xFFFF) << 16 | index */
		const u64 m1 = rbits > 64 ? ((1ull << (rbits - 64)) - 1) : 0ull; /* rbits - 64 <= 16 */
		const u64 m0 = rbits >= 64 ? ~0ull : ((1ull << rbits) - 1);
		u64 cA[ITEMS];
		u32 cB[ITEMS];
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = crel + r * 64 + lane;
			const u64 x0 = key[r][0] & m0, x1 = key[r][1] & m1;
			cA[r] = (x1 << 48) | (x0 >> 16);
			cB[r] = ((u32)(x0 & 0xFFFFu) << 16) | rel[r];
			if (idx < len) {
				s_A[idx] = cA[r];
				s_B[idx] = cB[r];
			}
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 bstart = span[r] & 0xFFFFu, bend = span[r] >> 16;
			const u32 A[2] = {(u32)cA[r], (u32)(cA[r] >> 32)};
			const u32 B = cB[r];
			u32 rank = 0, q = bstart;
			if (is_big(r))
				q = bend;
			for (; q + 4 <= bend; q += 4) { /* (A, B) is one 96-bit number: B holds the low rem bits and the index */
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const u64 a = s_A[q + u];
					const u32 x[2] = {(u32)a, (u32)(a >> 32)};
					br_rank_add_less<2>(rank, x, s_B[q + u], A, B);
				}
			}
			for (; q < bend; ++q) {
				const u64 a = s_A[q];
				const u32 x[2] = {(u32)a, (u32)(a >> 32)};
				br_rank_add_less<2>(rank, x, s_B[q], A, B);
			}
			place[r] = bstart + rank;
		}
		rank_big_buckets([&](u32 bs, u32 be, u32 i) {
			const u64 a = s_A[i];
			const u32 A[2] = {(u32)a, (u32)(a >> 32)};
			const u32 B = s_B[i];
			u32 n = 0, q = bs;
			for (; q + 4 <= be; q += 4) {
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const u64 o = s_A[q + u];
					const u32 x[2] = {(u32)o, (u32)(o >> 32)};
					br_rank_add_less<2>(n, x, s_B[q + u], A, B);
				}
			}
			for (; q < be; ++q) {
				const u64 o = s_A[q];
				const u32 x[2] = {(u32)o, (u32)(o >> 32)};
				br_rank_add_less<2>(n, x, s_B[q], A, B);
			}
			return n;
		});
	} else {
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = crel + r * 64 + lane;
			if (idx < len)
				store_rec<SIZE>(s_key + (size_t)idx * SIZE, key[r]);
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 bstart = span[r] & 0xFFFFu, bend = span[r] >> 16;
			const u32 me = bstart + rel[r];
			u32 rank = 0;
			u32 y[2 * SIZE];
#pragma unroll
			for (int w = 0; w < SIZE; ++w) {
				y[2 * w] = (u32)key[r][w];
				y[2 * w + 1] = (u32)(key[r][w] >> 32);
			}
			auto count_one = [&](u32 q) { /* records before this one: smaller ones, and equal ones that stand in front of it */
				u64 o[SIZE];
				load_rec<SIZE>(s_key + (size_t)q * SIZE, o);
				u32 x[2 * SIZE];
#pragma unroll
				for (int w = 0; w < SIZE; ++w) {
					x[2 * w] = (u32)o[w];
					x[2 * w + 1] = (u32)(o[w] >> 32);
				}
				br_rank_add_less<2 * SIZE>(rank, x, q, y, me);
			};
			u32 q = bstart;
			if (is_big(r))
				q = bend;
			for (; q + 2 <= bend; q += 2) { /* two chains per iteration (a `#pragma unroll 2` is refused around the inline assembly) */
				count_one(q);
				count_one(q + 1);
			}
			if (q < bend)
				count_one(q);
			place[r] = bstart + rank;
		}
		rank_big_buckets([&](u32 bs, u32 be, u32 i) { /* whole records, the position as the last word: as in the owners' walks */
			u64 m[SIZE];
			load_rec<SIZE>(s_key + (size_t)i * SIZE, m);
			u32 y[2 * SIZE], n = 0;
#pragma unroll
			for (int w = 0; w < SIZE; ++w) {
				y[2 * w] = (u32)m[w];
				y[2 * w + 1] = (u32)(m[w] >> 32);
			}
			for (u32 q = bs; q < be; ++q) {
				u64 o[SIZE];
				load_rec<SIZE>(s_key + (size_t)q * SIZE, o);
				u32 x[2 * SIZE];
#pragma unroll
				for (int w = 0; w < SIZE; ++w) {
					x[2 * w] = (u32)o[w];
					x[2 * w + 1] = (u32)(o[w] >> 32);
				}
				br_rank_add_less<2 * SIZE>(n, x, q, y, i);
			}
			return n;
		});
	}
#if defined(BR_STOP_AFTER) && BR_STOP_AFTER <= 2
	{
		u32 acc = 0;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r)
			acc |= place[r];
		if (acc != 0x7FFFFFFFu)
			return;
	}
#endif
	__syncthreads(); /* every pair has been read */
#pragma unroll
	for (int r = 0; r < ITEMS; ++r) {
		const u32 idx = crel + r * 64 + lane;
		if (idx < len)
			store_rec<SIZE>(s_key + (size_t)place[r] * SIZE, key[r]);
	}
	__syncthreads();
#if defined(BR_STOP_AFTER) && BR_STOP_AFTER <= 3
	if (s_key[tid] != 0x7FFFFFFF12345ull)
		return;
#endif
	if constexpr (!FUSED) {
		for (u32 idx = tid; idx < len; idx += THREADS) {
			u64 x[SIZE];
			load_rec<SIZE>(s_key + (size_t)idx * SIZE, x);
			store_rec<SIZE>(T + (size_t)idx * SIZE, x);
		}
	} else {
		/* ---- the tile is in order in LDS and made of whole buckets: count it where it lies (k_compact's tile body; nothing below the tile matters) */
		const u32 rec_bytes = P.sbytes + P.cbytes;
		const bool use_lut = P.lut_prefix_len != 0 && !P.kff && !P.without_output;
		const u64 lane_lt = (1ull << lane) - 1;
		u32 tail_bits = 0, wlast = NONE;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = crel + r * 64 + lane;
			bool is_tail = false;
			if (idx < len) {
				load_rec<SIZE>(s_key + (size_t)idx * SIZE, key[r]);
				is_tail = true;
				if (idx + 1 < len) {
					u64 nx[SIZE];
					load_rec<SIZE>(s_key + (size_t)(idx + 1) * SIZE, nx);
					is_tail = !kmc_equal<SIZE>(nx, key[r]);
				}
			}
			const u64 m = __ballot(is_tail);
			if (is_tail)
				tail_bits |= 1u << r;
			if (m)
				wlast = crel + r * 64 + 63 - (u32)__clzll((long long)m);
		}
		if (lane == 0)
			s_wlast[wave] = wlast;
		__syncthreads(); /* the records in order have been read: R0 and R1 are free */
		u32 carry = NONE; /* -1: no tail before record 0 */
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const u32 x = s_wlast[w];
			if (w < (int)wave && x != NONE)
				carry = x;
		}
		carry = (u32)__builtin_amdgcn_readfirstlane((int)carry);
		u32 cnt[ITEMS];
		u32 rank2[(ITEMS + 1) / 2]; /* wave-relative rank among counted k-mers, 16 bits each; 0xFFFF = not counted */
		u32 nu = 0, nb = 0, na = 0, nc = 0;
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
			const u32 rk = __builtin_amdgcn_mbcnt_hi((u32)(mc >> 32), __builtin_amdgcn_mbcnt_lo((u32)mc, nc));
			const u32 rk16 = ((mc >> lane) & 1ull) ? rk : 0xFFFFu;
			if (r & 1)
				rank2[r >> 1] |= rk16 << 16;
			else
				rank2[r >> 1] = rk16;
			nu += (u32)__popcll(m);
			nb += (u32)__popcll(mb);
			na += (u32)__popcll(ma);
			nc += (u32)__popcll(mc);
			if (m)
				carry = rowrel + 63 - (u32)__clzll((long long)m);
			__builtin_amdgcn_sched_barrier(0);
		}
		if (lane == 0) {
			s_wcnt[wave] = nc;
			s_wtal[wave * 3 + 0] = nu;
			s_wtal[wave * 3 + 1] = nb;
			s_wtal[wave * 3 + 2] = na;
		}
		__syncthreads();
		u32 wave_off = 0, tile_counted = 0;
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const u32 x = s_wcnt[w];
			if (w < (int)wave)
				wave_off += x;
			tile_counted += x;
		}
#if defined(BR_STOP_AFTER) && BR_STOP_AFTER <= 4
		if (tile_counted != 0x7FFFFFFFu)
			return;
#endif
		const u32 slot = 2 * tile + by;
		if (tid == 0) {
			u32 tu = 0, tb = 0, ta = 0;
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				tu += s_wtal[w * 3 + 0];
				tb += s_wtal[w * 3 + 1];
				ta += s_wtal[w * 3 + 2];
			}
			u64 *sh = gr.tally[bin] + (size_t)(slot % CP_SHARDS) * 4;
			if (tu)
				atomicAdd(&sh[0], (u64)tu);
			if (tb)
				atomicAdd(&sh[1], (u64)tb);
			if (ta)
				atomicAdd(&sh[2], (u64)ta);
			if (!P.without_output && tile_counted) {
				gr.status[bin][slot] = tile_counted;
				gr.chunk_src[bin][slot] = c0;
			}
		}
		if (!P.without_output && tile_counted) { /* uniform over the workgroup */
			const u32 tile_bytes = tile_counted * rec_bytes;
			const u32 pshift = 2 * (P.k - P.lut_prefix_len);
			uint8_t *const dst = gr.scratch[bin] + c0 * (u64)(SIZE * 8); /* 8-byte aligned; room for 8 SIZE bytes per record of the chunk */
			u32 *dst32 = reinterpret_cast<u32 *>(dst);
			const u32 ndw = (tile_bytes + 3) >> 2; /* whole dwords: the bytes behind the last record are the span's own */
			const u32 inv = rec_bytes > 1 ? (u32)(((1ull << 32) + rec_bytes - 1) / rec_bytes) : 0u; /* x / rec_bytes = umulhi(x, inv), exact for x < 2^29 */
			u32 *s_aux = s_start; /* R1: LUT prefixes (one word) / counts (wider) of the staged records */
			if constexpr (SIZE == 1) {
				/* counted k-mers and their counts go to their ranks as they are (two LDS stores per row); the records — one 64-bit value in output byte order
				 * (rec_bytes <= 8: host) — and the LUT prefixes are then made by ONE pass of tile_counted threads (~5 % of the records at cutoff 2) instead of by
				 * every row of every wave */
#pragma unroll
				for (int r = 0; r < ITEMS; ++r) {
					const u32 rk16 = (rank2[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu;
					if (rk16 != 0xFFFFu) {
						s_key[wave_off + rk16] = key[r][0];
						s_aux[wave_off + rk16] = cnt[r];
					}
				}
				__syncthreads();
				for (u32 j = tid; j < tile_counted; j += THREADS) {
					/* without a LUT prefix (KFF) the suffix bytes reach up to the top of the k-mer: a group tag above bit 2k must not get into them */
					u64 kx[1] = {(2 * P.k < 64) ? (s_key[j] & ((1ull << (2 * P.k)) - 1)) : s_key[j]};
					const u32 c = s_aux[j];
					u64 rv = P.sbytes ? __builtin_bswap64(kx[0] << (8 * (8 - P.sbytes))) : 0ull;
					if (P.cbytes) {
						const u32 cv = P.kff ? (__builtin_bswap32(c) >> (8 * (4 - P.cbytes))) : c;
						rv |= (u64)cv << (8 * P.sbytes);
					}
					s_key[j] = rv;
					if (use_lut)
						s_aux[j] = (u32)kmc_remove_suffix<1>(kx, pshift) & lut_mask;
				}
				__syncthreads();
				for (u32 w = tid; w < ndw; w += THREADS) {
					const u32 i0 = w << 2;
					u32 ri = rec_bytes > 1 ? __umulhi(i0, inv) : i0, q = i0 - ri * rec_bytes;
					u64 cur = s_key[ri];
					u32 word = 0;
#pragma unroll
					for (int t = 0; t < 4; ++t) {
						word |= ((u32)(cur >> (8 * q)) & 0xFFu) << (8 * t);
						if (++q == rec_bytes) {
							q = 0;
							++ri;
							cur = s_key[ri < (u32)CAP ? ri : (u32)CAP - 1];
						}
					}
					dst32[w] = word;
				}
				if (use_lut) {
					u64 *lut = gr.lut_base[bin] + (size_t)(slot % lut_shards) * lut_stride;
					for (u32 j = tid; j < tile_counted; j += THREADS) {
						const u32 pf = s_aux[j];
						if (j + 1 == tile_counted || s_aux[j + 1] != pf)
							atomicAdd(&lut[pf], (u64)(j + 1));
						if (j > 0 && s_aux[j - 1] != pf)
							atomicAdd(&lut[pf], (u64)0 - (u64)j);
					}
				}
			} else {
				/* staged: the k-mer (tag bits cleared) at its rank where the records were, its count in R1 */
#pragma unroll
				for (int r = 0; r < ITEMS; ++r) {
					const u32 rk16 = (rank2[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu;
					if (rk16 != 0xFFFFu) {
						const u32 rank = wave_off + rk16;
						u64 kx[SIZE];
#pragma unroll
						for (int w = 0; w < SIZE; ++w)
							kx[w] = key[r][w];
						kmc_mask_low<SIZE>(kx, 2 * P.k);
						store_rec<SIZE>(s_key + (size_t)rank * SIZE, kx);
						s_aux[rank] = cnt[r];
					}
				}
				__syncthreads();
				/* byte i of the chunk's output is byte i % rec_bytes of record i / rec_bytes — suffix bytes high -> low (kb_sorter.h:1198-1199), then the
				 * counter, little-endian for KMC (:1200-1201), big-endian for KFF (:1210-1211) */
				auto out_byte = [&](u32 i) -> u32 {
					const u32 ri = rec_bytes > 1 ? __umulhi(i, inv) : i, q = i - ri * rec_bytes;
					if (q < P.sbytes) {
						const u32 pbyte = P.sbytes - 1 - q;
						return (u32)(s_key[(size_t)ri * SIZE + (pbyte >> 3)] >> ((pbyte & 7) * 8)) & 0xFFu;
					}
					const u32 cq = q - P.sbytes;
					return (s_aux[ri] >> (8 * (P.kff ? (P.cbytes - 1 - cq) : cq))) & 0xFFu;
				};
#pragma clang loop unroll(disable) vectorize(disable)
				for (u32 wd = tid; wd < ndw; wd += THREADS) {
					const u32 i0 = wd * 4;
					u32 word = out_byte(i0);
					word |= (i0 + 1 < tile_bytes ? out_byte(i0 + 1) : 0u) << 8;
					word |= (i0 + 2 < tile_bytes ? out_byte(i0 + 2) : 0u) << 16;
					word |= (i0 + 3 < tile_bytes ? out_byte(i0 + 3) : 0u) << 24;
					dst32[wd] = word;
				}
				if (use_lut) {
					u64 *lut = gr.lut_base[bin] + (size_t)(slot % lut_shards) * lut_stride;
					auto prefix_of = [&](u32 j) -> u32 {
						u64 x[SIZE];
						load_rec<SIZE>(s_key + (size_t)j * SIZE, x);
						return (u32)kmc_remove_suffix<SIZE>(x, pshift) & lut_mask;
					};
					for (u32 j = tid; j < tile_counted; j += THREADS) {
						const u32 pf = prefix_of(j);
						if (j + 1 == tile_counted || prefix_of(j + 1) != pf)
							atomicAdd(&lut[pf], (u64)(j + 1));
						if (j > 0 && prefix_of(j - 1) != pf)
							atomicAdd(&lut[pf], (u64)0 - (u64)j);
					}
				}
			}
		}
	}
}

template <int SIZE, bool FUSED>
__global__ void __launch_bounds__(BrCfg<SIZE>::THREADS, BR_MIN_WAVES) k_bucket_rank(const GrpRank gr, DevParams P, u32 key_bits, u32 hbits, u32 lut_shards, u64 lut_stride,
                                                                                  u32 lut_mask, u32 *flag)
{
	KMC_DYN_LDS(unsigned char, s_raw);
	br_tile<SIZE, FUSED>(gr, P, key_bits, hbits, lut_shards, lut_stride, lut_mask, flag, blockIdx.x, blockIdx.y, s_raw);
}

/* ------------------------------------------------------------------------------------------------ tiles with a bucket beyond the LDS capacity
 * What the reference does with a bucket that stays large is recurse (raduls_impl.h:680-737: a big bucket takes a share of the threads and another radix
 * level); round 3 sent the whole GROUP of bins back through LSD passes over every byte. Here ONE workgroup takes such a tile — CAP < records <=
 * GT_MAX_RECORDS — and sorts it by itself: stable 8-bit LSD passes between the tile's slice of the record array and its slice of the free array (the span
 * its output will go to), over the key bits that can differ inside the tile (the bits below the bucket bits + the bits in which its first and last bucket
 * number differ; an even number of passes, so the records end where they started), each pass = a histogram read + chunks of THREADS x 8 / 4 / 2 records (by record width) ranked as
 * k_onesweep ranks a tile (per-wave digit counts, match-any ballots), with the digit bases running in LDS instead of a look-back. Then it streams through
 * the sorted records once more — run tails, counts (a run may be as long as the tile), cutoffs, records straight to the span — and reports like a tile of
 * k_bucket_rank (status, chunk_src, LUT, tallies). Persistent workgroups take the listed tiles one by one; with nothing listed the kernel costs a launch.
 * It is the rare path (a few tiles per group on repeat-rich input): simple before fast. */
template <int SIZE>
__global__ void __launch_bounds__(GT_THREADS) k_giant_tiles(const GrpRank gr, DevParams P, u32 stride, u32 key_bits, u32 hbits, u32 lut_shards, u64 lut_stride, u32 lut_mask,
                                                              u32 *err /* the stream's error block: words 12 and 14-15 count the tiles and records taken here (statistics) */)
{
	constexpr int THREADS = GT_THREADS, ITEMS = SIZE == 1 ? 8 : (SIZE == 2 ? 4 : 2), NW = THREADS / 64, CHUNK = THREADS * ITEMS;
	constexpr u32 NONE = 0xFFFFFFFFu;
	static_assert(THREADS >= 256, "one digit per thread");
	__shared__ u32 s_whist[NW * 256];
	__shared__ u32 s_base[256];
	__shared__ u32 s_scan[NW + 1];
	__shared__ u32 s_pick;
	__shared__ u32 s_wlast[NW], s_wcnt[NW];
	const u32 tid = threadIdx.x, lane = tid & 63;
	const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const bool use_lut = P.lut_prefix_len != 0 && !P.kff && !P.without_output;
	const u32 pshift = 2 * (P.k - P.lut_prefix_len);
	const u32 bsh = 64 - hbits, rbits = key_bits - hbits;
	const u64 lane_lt = (1ull << lane) - 1;
	while (true) {
		__syncthreads();
		if (tid == 0)
			s_pick = atomicAdd(&gr.giant[1], 1u);
		__syncthreads();
		const u32 pick = s_pick;
		if (pick >= ld_agent(&gr.giant[0])) /* final: k_bucket_rank ran before this kernel on the stream */
			break;
		const u32 entry = gr.giant[2 + pick];
		const u32 gtile = entry & ((1u << GT_CUT_SHIFT) - 1u), cut = entry >> GT_CUT_SHIFT; /* cut != 0: k_bucket_rank kept the tile's first `cut` records (its first chunk) */
		const u32 bin = (u32)__builtin_amdgcn_readfirstlane((int)grp_find(gr.win_prefix, gr.g, gtile));
		const u32 tile = gtile - gr.win_prefix[bin];
		const u64 b0 = gr.bounds[bin][tile] + cut, b1 = gr.bounds[bin][tile + 1];
		const u32 L = (u32)(b1 - b0);
		u64 *T = gr.S[bin] + b0 * SIZE;
		if (gr.rec_base) { /* indirect sort: the tile is a run of (key top, record number) pairs; its records are brought together first */
			T = gr.giant_T[bin] + b0 * SIZE;
			const u64 *pairs = gr.S[bin] + b0;
			for (u32 i = tid; i < L; i += THREADS) {
				u64 x[SIZE];
				load_rec<SIZE>(gr.rec_base + (size_t)(u32)pairs[i] * SIZE, x);
				store_rec<SIZE>(T + (size_t)i * SIZE, x);
			}
			__syncthreads();
		}
		u64 *U = reinterpret_cast<u64 *>(gr.scratch[bin] + b0 * (u64)(SIZE * 8));
		(void)stride;
		/* the key bits to sort by */
		u32 passes;
		{
			u64 x0[SIZE], xl[SIZE];
			load_rec<SIZE>(T, x0);
			load_rec<SIZE>(T + (size_t)(L - 1) * SIZE, xl);
			const u64 k0 = hbits ? bs_p64<SIZE>(x0, key_bits) >> bsh : 0ull, kl = hbits ? bs_p64<SIZE>(xl, key_bits) >> bsh : 0ull;
			const u64 diff = k0 ^ kl;
			const u32 dbits = diff ? 64u - (u32)__clzll((long long)diff) : 0u;
			const u32 nbits = rbits + dbits < key_bits ? rbits + dbits : key_bits;
			passes = (nbits + 7) / 8; /* an odd number leaves the sorted records in U, the span the output goes to: the counting below then works in place (it
			                           * writes behind what it has read: a counted k-mer's record is no longer than the record it was counted from) */
		}
		u64 *src = T, *dst = U;
		for (u32 p = 0; p < passes; ++p) {
			if (tid < 256)
				s_base[tid] = 0;
			__syncthreads();
			for (u32 i = tid; i < L; i += THREADS) {
				u64 x[SIZE];
				load_rec<SIZE>(src + (size_t)i * SIZE, x);
				(void)__hip_atomic_fetch_add(&s_base[kmc_get_byte<SIZE>(x, p)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
			__syncthreads();
			{
				const u32 v = tid < 256 ? s_base[tid] : 0u;
				u32 total;
				const u32 ex = block_excl_sum<NW, u32>(v, s_scan, total);
				if (tid < 256)
					s_base[tid] = ex;
			}
			__syncthreads();
			for (u32 c0 = 0; c0 < L; c0 += CHUNK) {
				for (u32 i = tid; i < (u32)NW * 256; i += THREADS)
					s_whist[i] = 0;
				__syncthreads();
				u64 key[ITEMS][SIZE];
				u32 dg[ITEMS];
#pragma unroll
				for (int r = 0; r < ITEMS; ++r) {
					const u32 idx = c0 + wave * (ITEMS * 64) + r * 64 + lane;
					dg[r] = 0;
					if (idx < L) {
						load_rec<SIZE>(src + (size_t)idx * SIZE, key[r]);
						dg[r] = kmc_get_byte<SIZE>(key[r], p);
						(void)__hip_atomic_fetch_add(&s_whist[wave * 256 + dg[r]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					} else {
#pragma unroll
						for (int w = 0; w < SIZE; ++w)
							key[r][w] = 0;
					}
				}
				__syncthreads();
				if (tid < 256) { /* counts -> first position of (wave, digit) in dst; the digit's base moves on to the next chunk */
					u32 run = s_base[tid];
#pragma unroll
					for (int w = 0; w < NW; ++w) {
						const u32 t = s_whist[w * 256 + tid];
						s_whist[w * 256 + tid] = run;
						run += t;
					}
					s_base[tid] = run;
				}
				__syncthreads();
#pragma unroll
				for (int r = 0; r < ITEMS; ++r) {
					const u32 idx = c0 + wave * (ITEMS * 64) + r * 64 + lane;
					const bool valid = idx < L;
					const u64 vm = __ballot(valid);
					u32 lo = (u32)vm, hi = (u32)(vm >> 32);
#pragma unroll
					for (int b = 0; b < 8; ++b) { /* the lanes that hold the same digit (k_onesweep's match-any) */
						const u32 sb = (u32)__builtin_amdgcn_sbfe((int)dg[r], b, 1);
						const u64 m = __builtin_amdgcn_uicmp(sb, 0u, 33 /* ICMP_NE */);
						lo = __builtin_amdgcn_bitop3_b32(sb, lo, (u32)m, 0x84);
						hi = __builtin_amdgcn_bitop3_b32(sb, hi, (u32)(m >> 32), 0x84);
					}
					const u32 below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
					u32 *ctr = &s_whist[wave * 256 + dg[r]];
					const u32 slot = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					KMC_WAVE_LOCKSTEP(); /* every lane has read the counter before the lowest peer moves it on */
					if (valid && below == 0)
						__hip_atomic_store(ctr, slot + (u32)(__popc(lo) + __popc(hi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					KMC_WAVE_LOCKSTEP();
					if (valid)
						store_rec<SIZE>(dst + (size_t)(slot + below) * SIZE, key[r]);
				}
				__syncthreads();
			}
			u64 *t = src;
			src = dst;
			dst = t;
			__syncthreads(); /* a workgroup's global stores are visible to its own loads after the barrier (one CU, one L1) */
		}
		/* ---- the tile is in order in `src` (T, or U after an odd number of passes): run lengths, cutoffs, records (kb_sorter.h:1128-1281), chunk by chunk */
		const u64 *const Sd = src;
		uint8_t *const span = reinterpret_cast<uint8_t *>(U);
		const u32 slot_id = 2 * tile + (cut ? 1u : 0u);
		u64 *lut = use_lut ? gr.lut_base[bin] + (size_t)(slot_id % lut_shards) * lut_stride : nullptr;
		u32 prev_tail = NONE; /* absolute position of the last tail so far, -1 = none */
		u32 counted_total = 0, nu = 0, nb = 0, na = 0;
		for (u32 c0 = 0; c0 < L; c0 += CHUNK) {
			const u32 crel = wave * (ITEMS * 64);
			u64 key[ITEMS][SIZE];
			u32 tail_bits = 0, wlast = NONE;
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 idx = c0 + crel + r * 64 + lane;
				bool is_tail = false;
				if (idx < L) {
					load_rec<SIZE>(Sd + (size_t)idx * SIZE, key[r]);
					is_tail = true;
					if (idx + 1 < L) {
						u64 nx[SIZE];
						load_rec<SIZE>(Sd + (size_t)(idx + 1) * SIZE, nx);
						is_tail = !kmc_equal<SIZE>(nx, key[r]);
					}
				} else {
#pragma unroll
					for (int w = 0; w < SIZE; ++w)
						key[r][w] = 0;
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
			for (int w = 0; w < NW; ++w) {
				const u32 x = s_wlast[w];
				if (x != NONE) {
					chunk_last = x;
					if (w < (int)wave)
						carry = x;
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
			for (int w = 0; w < NW; ++w) {
				const u32 x = s_wcnt[w];
				if (w < (int)wave)
					wave_off += x;
				chunk_counted += x;
			}
			if (!P.without_output) {
#pragma unroll
				for (int r = 0; r < ITEMS; ++r) {
					if (rk[r] != NONE) {
						u64 kx[SIZE];
#pragma unroll
						for (int w = 0; w < SIZE; ++w)
							kx[w] = key[r][w];
						kmc_mask_low<SIZE>(kx, 2 * P.k); /* drops a group tag above the k-mer */
						kmc_emit_record<SIZE>(span + (size_t)(counted_total + wave_off + rk[r]) * rec_bytes, kx, cnt[r], P.sbytes, P.cbytes, P.kff != 0);
						if (use_lut)
							atomicAdd(&lut[(u32)kmc_remove_suffix<SIZE>(kx, pshift) & lut_mask], 1ull);
					}
				}
			}
			counted_total += chunk_counted;
			if (chunk_last != NONE)
				prev_tail = c0 + chunk_last;
			__syncthreads(); /* s_wlast / s_wcnt are rewritten by the next chunk */
		}
		if (lane == 0) { /* nu, nb, na are per wave: summed through LDS */
			s_whist[wave * 3 + 0] = nu;
			s_whist[wave * 3 + 1] = nb;
			s_whist[wave * 3 + 2] = na;
		}
		__syncthreads();
		if (tid == 0) {
			u32 tu = 0, tb = 0, ta = 0;
#pragma unroll
			for (int w = 0; w < NW; ++w) {
				tu += s_whist[w * 3 + 0];
				tb += s_whist[w * 3 + 1];
				ta += s_whist[w * 3 + 2];
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
				gr.chunk_src[bin][slot_id] = b0;
			}
			atomicAdd(&err[12], 1u);
			atomicAdd(reinterpret_cast<u64 *>(err + 14), (u64)L);
		}
	}
}

#endif
