/*
 * kmc_amd/csrc/kernels.hip.h — gfx950 (MI355X, wave64) kernels of the KMC stage-2 hot path.
 *
 *   index    k_pack_scan / k_pack_offsets / k_pack_index : locate super-k-mers inside expander packs
 *   expand   k_expand<SIZE>      : super-k-mer bytes -> canonical k-mer records   (ref kb_sorter.h:299-362)
 *   sort     k_hist<SIZE>        : all per-pass byte histograms in ONE read of the records
 *            k_hist_scan         : exclusive scan -> global digit bases
 *            k_onesweep<SIZE,..> : one 8-bit LSD pass, single read + single write per record, decoupled
 *                                  look-back across tiles, wave64 ballot ranking, LDS-staged scatter
 *                                  (replaces raduls_impl.h:546-754 / radix.h:469-842)
 *   compact  k_compact<SIZE>     : run-length count + cutoffs + suffix/counter bytes + prefix LUT + tallies
 *                                  in ONE read of the sorted records (ref kb_sorter.h:1128-1281)
 *
 * All work is integer/byte permutation: HBM-bound, no MFMA. Design rules applied (cdna_hip_programming.md):
 * 64-wide ballots/popcounts, coalesced 512 B..1 KiB per wave-instruction, per-wave private LDS histograms
 * (no LDS atomics on the ranking path), >=4 workgroups per CU resident, inter-workgroup hand-off only through
 * single-word relaxed agent-scope atomics where the word IS the flag (Guideline 16 "R2"), every spin bounded.
 */
#ifndef KMC_AMD_KERNELS_HIP_H
#define KMC_AMD_KERNELS_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmer_ops.h"

typedef kmc_u64 u64;
typedef kmc_u32 u32;

/* device-side error bits (d_err) */
enum : u32 { KERR_CORRUPT = 1u, KERR_NREC = 2u, KERR_CAPACITY = 4u, KERR_WATCHDOG = 8u };

/* tile geometry */
constexpr int EXP_BLOCK = 256, EXP_ITEMS = 4, EXP_TILE = EXP_BLOCK * EXP_ITEMS; /* k-mers per expand workgroup   */
#ifndef RS_BLOCK_THREADS
#define RS_BLOCK_THREADS 512 /* 8192-record tiles: 16 % faster than 256 x 16 (fewer tiles to look back over, 256-B runs) */
#endif
constexpr int RS_BLOCK = RS_BLOCK_THREADS, RS_WAVES = RS_BLOCK / 64;                           /* radix scatter workgroup       */
#ifndef CP_BLOCK_THREADS
#define CP_BLOCK_THREADS 256
#endif
constexpr int CP_BLOCK = CP_BLOCK_THREADS;                                                     /* compaction workgroup          */
constexpr u32 SPIN_LIMIT = 1u << 24;                                              /* look-back watchdog (polls)    */

#ifndef RS_WORDS_PER_THREAD
#define RS_WORDS_PER_THREAD 16 /* 8-byte words held per thread in a scatter tile */
#endif
template <int SIZE> struct RsCfg { /* records per thread in a scatter tile: RS_WORDS_PER_THREAD x 8 B per thread for every SIZE */
	static constexpr int ITEMS = (RS_WORDS_PER_THREAD / SIZE) > 2 ? (RS_WORDS_PER_THREAD / SIZE) : 2;
	static constexpr int TILE = RS_BLOCK * ITEMS;
};
#ifndef CP_WORDS_PER_THREAD
#define CP_WORDS_PER_THREAD 16
#endif
template <int SIZE> struct CpCfg {
	static constexpr int ITEMS = (CP_WORDS_PER_THREAD / SIZE) > 2 ? (CP_WORDS_PER_THREAD / SIZE) : 2;
	static constexpr int TILE = CP_BLOCK * ITEMS;
};

/* per-run constants handed to the kernels by value */
struct DevParams {
	u32 k, both_strands, cutoff_min, cutoff_max, counter_max, lut_prefix_len, sbytes, cbytes, kff, without_output;
};

/* ------------------------------------------------------------------------------------------------ tracing (tuning builds only)
 * -DKMC_TRACE: thread 0 of sampled tiles stamps s_memtime at phase boundaries into g_trace (8 u64 per tile). */
#ifdef KMC_TRACE
constexpr int TRACE_SLOTS = 1 << 17;
__device__ unsigned long long g_trace[TRACE_SLOTS * 8];
#define TRACE_STAMP(kind, tile, j)                                                                                              \
	do {                                                                                                                        \
		if (threadIdx.x == 0 && (tile) < (u32)TRACE_SLOTS / 4)                                                                  \
			g_trace[((kind) * (TRACE_SLOTS / 4) + (tile)) * 8 + (j)] = (j) == 0 ? wall_clock64() : __builtin_readcyclecounter(); \
	} while (0)
#define TRACE_VALUE(kind, tile, j, val)                                                                                         \
	do {                                                                                                                        \
		if (threadIdx.x == 0 && (tile) < (u32)TRACE_SLOTS / 4)                                                                  \
			g_trace[((kind) * (TRACE_SLOTS / 4) + (tile)) * 8 + (j)] = (val);                                                   \
	} while (0)
#else
#define TRACE_STAMP(kind, tile, j) do { } while (0)
#define TRACE_VALUE(kind, tile, j, val) do { } while (0)
#endif

/* ------------------------------------------------------------------------------------------------ helpers */

__device__ __forceinline__ u32 ld_agent(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_agent(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename T> __device__ __forceinline__ T wave_incl_sum(T v, u32 lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		T t = __shfl_up(v, o);
		if ((int)lane >= o)
			v += t;
	}
	return v;
}
template <typename T> __device__ __forceinline__ T wave_incl_max(T v, u32 lane)
{
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		T t = __shfl_up(v, o);
		if ((int)lane >= o)
			v = t > v ? t : v;
	}
	return v;
}
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_down(v, o);
	return v; /* lane 0 holds the sum */
}

/* Block-wide exclusive scans over one value per thread (NW waves). `tmp` has NW+1 entries of T in LDS.
 * All threads must call; returns the exclusive prefix, `total` = sum/max over the block. */
template <int NW, typename T> __device__ __forceinline__ T block_excl_sum(T v, T *tmp, T &total)
{
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	T inc = wave_incl_sum<T>(v, lane);
	if (lane == 63)
		tmp[wave] = inc;
	__syncthreads();
	if (wave == 0) {
		T w = lane < NW ? tmp[lane] : (T)0;
		T winc = wave_incl_sum<T>(w, lane);
		if (lane < NW)
			tmp[lane] = winc - w;
		if (lane == NW - 1)
			tmp[NW] = winc;
	}
	__syncthreads();
	T res = tmp[wave] + inc - v;
	total = tmp[NW];
	__syncthreads();
	return res;
}
/* exclusive MAX scan (identity 0) */
template <int NW, typename T> __device__ __forceinline__ T block_excl_max(T v, T *tmp)
{
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	T inc = wave_incl_max<T>(v, lane);
	T prev = __shfl_up(inc, 1);
	if (lane == 0)
		prev = 0;
	if (lane == 63)
		tmp[wave] = inc;
	__syncthreads();
	if (wave == 0) {
		T w = lane < NW ? tmp[lane] : (T)0;
		T winc = wave_incl_max<T>(w, lane);
		T wprev = __shfl_up(winc, 1);
		if (lane == 0)
			wprev = 0;
		if (lane < NW)
			tmp[lane] = wprev;
	}
	__syncthreads();
	T res = tmp[wave] > prev ? tmp[wave] : prev;
	__syncthreads();
	return res;
}

template <int SIZE> __device__ __forceinline__ void load_rec(const u64 *p, u64 (&x)[SIZE])
{
	if constexpr (SIZE % 2 == 0) { /* 16-byte vector loads; records are 16-B aligned when SIZE is even */
		const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
#pragma unroll
		for (int i = 0; i < SIZE / 2; ++i) {
			ulonglong2 v = q[i];
			x[2 * i] = v.x;
			x[2 * i + 1] = v.y;
		}
	} else {
#pragma unroll
		for (int i = 0; i < SIZE; ++i)
			x[i] = p[i];
	}
}
template <int SIZE> __device__ __forceinline__ void store_rec(u64 *p, const u64 (&x)[SIZE])
{
	if constexpr (SIZE % 2 == 0) {
		ulonglong2 *q = reinterpret_cast<ulonglong2 *>(p);
#pragma unroll
		for (int i = 0; i < SIZE / 2; ++i)
			q[i] = make_ulonglong2(x[2 * i], x[2 * i + 1]);
	} else {
#pragma unroll
		for (int i = 0; i < SIZE; ++i)
			p[i] = x[i];
	}
}

/* ------------------------------------------------------------------------------------------------ index
 * The bin image is a chain of variable-length records; the only random-access entry points the caller has are
 * the expander-pack boundaries (CExpanderPackDesc, queues.h:376-396; <= 4096 super-k-mers each,
 * kb_collector.h:46). One LANE walks one pack (64 independent chains per wave, every pack of the bin in flight
 * at once), twice: first to count, then — after an exclusive scan over packs — to write the per-super-k-mer index. */

__global__ void __launch_bounds__(64) k_pack_scan(const uint8_t *__restrict__ data, const u64 *__restrict__ pack_start, u32 n_packs,
                                                   u32 k, u32 *__restrict__ pack_nsk, u64 *__restrict__ pack_nk, u32 *err)
{
	const u32 p = blockIdx.x * 64 + threadIdx.x;
	if (p >= n_packs)
		return;
	u64 pos = pack_start[p];
	const u64 end = pack_start[p + 1];
	u32 nsk = 0;
	u64 nk = 0;
	while (pos < end) {
		const u32 e = data[pos];
		pos += 1 + ((k + e + 3) >> 2);
		++nsk;
		nk += e + 1;
	}
	if (pos != end)
		atomicOr(err, KERR_CORRUPT);
	pack_nsk[p] = nsk;
	pack_nk[p] = nk;
}

/* single workgroup: exclusive scans of pack_nsk / pack_nk; totals[0] = #super-k-mers, totals[1] = #k-mers */
__global__ void __launch_bounds__(1024) k_pack_offsets(const u32 *__restrict__ pack_nsk, const u64 *__restrict__ pack_nk, u32 n_packs,
                                                        u64 *__restrict__ pack_sk_off, u64 *__restrict__ pack_k_off, u64 *totals,
                                                        u64 n_rec_expected, u32 *err)
{
	__shared__ u64 tmp[17];
	u64 carry_s = 0, carry_k = 0;
	for (u32 base = 0; base < n_packs; base += 1024) {
		const u32 i = base + threadIdx.x;
		const u64 a = i < n_packs ? (u64)pack_nsk[i] : 0;
		const u64 b = i < n_packs ? pack_nk[i] : 0;
		u64 ta, tb;
		const u64 ea = block_excl_sum<16, u64>(a, tmp, ta);
		const u64 eb = block_excl_sum<16, u64>(b, tmp, tb);
		if (i < n_packs) {
			pack_sk_off[i] = carry_s + ea;
			pack_k_off[i] = carry_k + eb;
		}
		carry_s += ta;
		carry_k += tb;
	}
	if (threadIdx.x == 0) {
		totals[0] = carry_s;
		totals[1] = carry_k;
		if (carry_k != n_rec_expected)
			atomicOr(err, KERR_NREC);
	}
}

/* second walk: sk_pos[s] = byte offset of super-k-mer s (its `e` byte), sk_koff[s] = index of its first k-mer;
 * tile_first[m] = the super-k-mer that contains k-mer m*EXP_TILE (entry point of expand tile m). */
__global__ void __launch_bounds__(64) k_pack_index(const uint8_t *__restrict__ data, const u64 *__restrict__ pack_start, u32 n_packs,
                                                    u32 k, const u64 *__restrict__ pack_sk_off, const u64 *__restrict__ pack_k_off,
                                                    u64 *__restrict__ sk_pos, u64 *__restrict__ sk_koff, u64 *__restrict__ tile_first,
                                                    u64 n_tiles, u64 sk_capacity)
{
	const u32 p = blockIdx.x * 64 + threadIdx.x;
	if (p >= n_packs)
		return;
	u64 pos = pack_start[p];
	const u64 end = pack_start[p + 1];
	u64 s = pack_sk_off[p], koff = pack_k_off[p];
	while (pos < end && s < sk_capacity) {
		const u32 e = data[pos];
		sk_pos[s] = pos;
		sk_koff[s] = koff;
		const u64 m = (koff + EXP_TILE - 1) / EXP_TILE; /* first tile boundary at or after koff */
		if (m * EXP_TILE < koff + e + 1 && m < n_tiles)
			tile_first[m] = s;
		pos += 1 + ((k + e + 3) >> 2);
		koff += e + 1;
		++s;
	}
}

/* ------------------------------------------------------------------------------------------------ expand
 * One THREAD per k-mer. A workgroup owns k-mers [m*T, (m+1)*T): it stages the (position, first-k-mer) pairs of
 * the super-k-mers overlapping that range in LDS, each thread binary-searches its super-k-mer there, pulls the
 * <= ceil((2k+13)/8) bytes of its window (neighbouring threads read overlapping bytes: L1 hits), builds the
 * forward k-mer and its reverse complement with bit tricks (kmer_ops.h) and writes the smaller one:
 * consecutive threads write consecutive records (coalesced 8*SIZE B per lane). */
template <int SIZE>
__global__ void __launch_bounds__(EXP_BLOCK) k_expand(const uint8_t *__restrict__ data, const u64 *__restrict__ sk_pos,
                                                       const u64 *__restrict__ sk_koff, const u64 *__restrict__ tile_first,
                                                       const u64 *__restrict__ totals, u64 n_rec, u64 n_tiles, u32 k,
                                                       u32 both_strands, u64 *__restrict__ out)
{
	__shared__ u64 s_pos[EXP_TILE + 1];
	__shared__ int s_rel[EXP_TILE + 1];
	const u64 n_sk = totals[0];
	if (n_sk == 0)
		return; /* only with a corrupt image (error already flagged by the index kernels) */
	const u64 m = blockIdx.x;
	const u64 j0 = m * EXP_TILE;
	u64 s_lo = tile_first[m];
	u64 s_hi = (m + 1 < n_tiles) ? tile_first[m + 1] : (n_sk - 1);
	if (s_lo > n_sk - 1) /* clamps matter only for corrupt images: never index outside the tables */
		s_lo = n_sk - 1;
	if (s_hi > n_sk - 1)
		s_hi = n_sk - 1;
	if (s_hi < s_lo)
		s_hi = s_lo;
	u32 cnt = (u32)(s_hi - s_lo + 1);
	if (cnt > EXP_TILE + 1)
		cnt = EXP_TILE + 1; /* cannot happen for a well-formed index (every super-k-mer holds >= 1 k-mer) */
	for (u32 i = threadIdx.x; i < cnt; i += EXP_BLOCK) {
		s_pos[i] = sk_pos[s_lo + i];
		s_rel[i] = (int)((long long)sk_koff[s_lo + i] - (long long)j0);
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < EXP_ITEMS; ++r) {
		const int idx = r * EXP_BLOCK + threadIdx.x;
		const u64 j = j0 + idx;
		if (j >= n_rec)
			continue;
		/* largest i with s_rel[i] <= idx (s_rel is strictly increasing, s_rel[0] <= 0) */
		u32 lo = 0, hi = cnt;
		while (hi - lo > 1) {
			const u32 mid = (lo + hi) >> 1;
			if (s_rel[mid] <= idx)
				lo = mid;
			else
				hi = mid;
		}
		const u32 off = (u32)(idx - s_rel[lo]);
		u64 v[SIZE];
		kmc_canonical_at<SIZE>(data + s_pos[lo] + 1, off, k, both_strands != 0, v);
		store_rec<SIZE>(out + j * SIZE, v);
	}
}

/* ------------------------------------------------------------------------------------------------ histogram
 * ONE pass over the records builds the 256-bin histogram of EVERY digit position (n_pass <= 8*SIZE), in LDS
 * (u32 per workgroup), flushed with 64-bit global atomics. A wave whose lanes all hold the same digit value
 * (zero high bytes, poly-A bins) adds once instead of issuing a 64-way conflicting LDS atomic. */
template <int SIZE>
__global__ void __launch_bounds__(256) k_hist(const u64 *__restrict__ recs, u64 n, u32 n_pass, u64 *__restrict__ ghist)
{
	extern __shared__ __attribute__((aligned(16))) u32 s_h[]; /* n_pass * 256 */
	for (u32 i = threadIdx.x; i < n_pass * 256; i += 256)
		s_h[i] = 0;
	__syncthreads();
	const u64 stride = (u64)gridDim.x * 256;
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		u64 x[SIZE];
		load_rec<SIZE>(recs + i * SIZE, x);
		const u64 act = __ballot(1);
		for (u32 b = 0; b < n_pass; ++b) {
			const u32 d = kmc_get_byte<SIZE>(x, b);
			const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
			if (__ballot(d == d0) == act) {
				if ((threadIdx.x & 63) == (u32)(__ffsll(act) - 1))
					atomicAdd(&s_h[b * 256 + d0], (u32)__popcll(act));
			} else {
				atomicAdd(&s_h[b * 256 + d], 1u);
			}
		}
	}
	__syncthreads();
	for (u32 i = threadIdx.x; i < n_pass * 256; i += 256) {
		const u32 v = s_h[i];
		if (v)
			atomicAdd(&ghist[i], (u64)v);
	}
}

/* grid = n_pass workgroups of 256: digit_base[pass][d] = number of records whose digit < d */
__global__ void __launch_bounds__(256) k_hist_scan(const u64 *__restrict__ ghist, u64 *__restrict__ digit_base)
{
	__shared__ u64 tmp[5];
	const u32 i = blockIdx.x * 256 + threadIdx.x;
	u64 total;
	digit_base[i] = block_excl_sum<4, u64>(ghist[i], tmp, total);
}

/* ------------------------------------------------------------------------------------------------ radix scatter
 * One 8-bit LSD pass over a portion of <= 2^29 records ("onesweep": single read, single write per record).
 *
 *  1. tile id from an atomic ticket (so every lower-numbered tile is already running: look-back cannot deadlock)
 *  2. wave w loads ITEMS x 64 consecutive records (512*SIZE B per wave-instruction), in index order
 *  3. ranking: for each of the ITEMS rounds the wave finds, with 8 ballots, the lanes holding the same digit
 *     ("match-any"); rank = wave-private running count of that digit (plain LDS read-modify-write by the lowest
 *     peer lane — no atomics) + number of lower peer lanes. Index order is preserved => the pass is STABLE.
 *  4. digit d's tile count = sum over waves; published as AGGREGATE in status[tile][d]; threads then look back
 *     over earlier tiles (each thread owns one digit) until a PREFIX is met, publish their own PREFIX.
 *  5. records are placed in LDS in digit order, then streamed out: consecutive threads write consecutive
 *     addresses inside each digit run (TILE/256 = 16 records = 128 B per run on uniform digits).
 * status word: [31:30] flag (0 empty, 1 aggregate, 2 inclusive prefix), [29:0] count — one relaxed agent-scope
 * 32-bit word that is both data and flag (no fences needed; cdna_hip_programming.md Guideline 16, form R2). */
constexpr u32 ST_AGG = 1u << 30, ST_PREFIX = 2u << 30, ST_MASK = (1u << 30) - 1;

#ifndef RS_MIN_WAVES
#define RS_MIN_WAVES 4 /* waves per SIMD the register allocator must leave room for (4 workgroups of 256 per CU) */
#endif
#ifndef RS_RANK_LDS
#define RS_RANK_LDS 0 /* measured: ballots 58.8 ms vs LDS words 62.0 ms per 7 passes of 1.65 G records */
#endif
#ifndef RS_LOOKBACK_K
#define RS_LOOKBACK_K 4
#endif
#ifndef RS_TPB
#define RS_TPB 1 /* tiles per ticket. Keep 1: a workgroup that owns consecutive tiles publishes the later ones late and every
                   * successor's look-back stalls on them (measured 150x slower at 2); larger tiles are the way to fewer tickets */
#endif

template <int SIZE>
__global__ void __launch_bounds__(RS_BLOCK, RS_MIN_WAVES) k_onesweep(const u64 *__restrict__ in, u64 *__restrict__ out, u32 n, u32 byte_idx,
                                                        const u64 *__restrict__ digit_base_in, u64 *__restrict__ digit_base_next,
                                                        u32 *status, u32 *tile_counter, u32 num_tiles, u32 *err)
{
	constexpr int ITEMS = RsCfg<SIZE>::ITEMS;
	constexpr int TILE = RsCfg<SIZE>::TILE;
	extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
	u64 *s_goff = reinterpret_cast<u64 *>(s_raw);              /* [256]  global index of LDS slot 0 as seen by digit d */
	u64 *s_keys = s_goff + 256;                                /* [SIZE*TILE] word-major: s_keys[w*TILE + slot]         */
	u32 *s_whist = reinterpret_cast<u32 *>(s_keys + SIZE * TILE); /* [RS_WAVES*256] per-wave digit counters -> offsets    */
	u32 *s_doff = s_whist + RS_WAVES * 256;                    /* [256]  first LDS slot of digit d                      */
	u32 *s_wsum = s_doff + 256;                                /* [4]                                                    */
	u32 *s_tile = s_wsum + 4;                                  /* [1]                                                    */

	if (threadIdx.x == 0)
		*s_tile = atomicAdd(tile_counter, 1u);
	__syncthreads();
	const u32 ticket = *s_tile;

#pragma unroll 1
	for (int it = 0; it < RS_TPB; ++it) {
		const u32 tile = ticket * RS_TPB + it;
		if (tile >= num_tiles)
			break;
		/* lane-constant addresses must be recomputed per tile: hoisted out of this loop they cost ~60 VGPRs and spill */
		u32 tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		const u32 lane = tid & 63, wave = tid >> 6;
#pragma unroll
		for (int i = 0; i < 4; ++i)
			s_whist[wave * 256 + i * 64 + lane] = 0;
		const u64 tile_base = (u64)tile * TILE;
		const u32 tile_n = (n - tile_base) < (u64)TILE ? (u32)(n - tile_base) : (u32)TILE;
		TRACE_STAMP(0, tile, 0);
		TRACE_STAMP(0, tile, 1);

		u64 key[ITEMS][SIZE];
		u32 rank[ITEMS];
		const u32 wbase = wave * (ITEMS * 64) + lane;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = wbase + r * 64;
			if (idx < tile_n)
				load_rec<SIZE>(in + (tile_base + idx) * SIZE, key[r]);
			else {
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					key[r][w] = 0;
			}
		}
		/* ranking, phase 1: find the lanes of this wave-round that hold the same digit ("match-any"), lowest peer
		 * lane adds the peer count to the wave's private counter with ONE returning LDS atomic. The counter atomics
		 * of all rounds are in flight together (LDS executes a wave's operations in order, so round r+1 sees round
		 * r's add). Two ways to get the peer mask:
		 *   RS_RANK_LDS=1  every lane ORs its lane bit into a per-(wave,digit) 64-bit LDS word, then reads it back
		 *                  (3 LDS ops per round instead of ~50 VALU/SALU; the words live in the not-yet-used key
		 *                  staging area and are cleared by the lowest peer)
		 *   RS_RANK_LDS=0  8 ballots, one per digit bit */
#if RS_RANK_LDS
		u64 *s_mask = s_keys + wave * 256; /* [256] per wave; s_keys is not touched before the LDS scatter phase */
#pragma unroll
		for (int i = 0; i < 4; ++i)
			s_mask[i * 64 + lane] = 0;
		const u64 lane_bit = 1ull << lane;
#endif
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const bool valid = (wbase + r * 64) < tile_n;
			const u32 d = kmc_get_byte<SIZE>(key[r], byte_idx);
#if RS_RANK_LDS
			if (valid)
				atomicOr(&s_mask[d], lane_bit);
			const u64 pm = s_mask[d];
			const u32 lo = valid ? (u32)pm : 0u, hi = valid ? (u32)(pm >> 32) : 0u;
#else
			const u64 vm = __ballot(valid);
			u32 lo = (u32)vm, hi = (u32)(vm >> 32);
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				const int sb = -(int)((d >> b) & 1); /* 0 or ~0 */
				const u64 m = __ballot(sb != 0);
				lo &= ~((u32)m ^ (u32)sb);
				hi &= ~((u32)(m >> 32) ^ (u32)sb);
			}
#endif
			const u32 below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
			const u32 leader = lo ? (u32)(__ffs((int)lo) - 1) : (hi ? (u32)(31 + __ffs((int)hi)) : 0u);
			u32 old = 0;
			if (valid && below == 0) {
#if RS_RANK_LDS
				s_mask[d] = 0;
#endif
				old = atomicAdd(&s_whist[wave * 256 + d], (u32)(__popc(lo) + __popc(hi)));
			}
			rank[r] = (old << 16) | (below << 8) | leader; /* old: meaningful in the leader lane only, until phase 2 */
			__builtin_amdgcn_sched_barrier(0); /* keep rounds in order: interleaving them only inflates SGPR/VGPR live ranges */
		}
		/* phase 2: fetch the leader's counter value */
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 meta = rank[r];
			const u32 old = (u32)__shfl((int)(meta >> 16), (int)(meta & 0xFF));
			rank[r] = old + ((meta >> 8) & 0xFF);
		}
		TRACE_STAMP(0, tile, 2);
		__syncthreads();
		TRACE_STAMP(0, tile, 3);

		/* thread `tid` (< 256) owns digit `tid` from here to the end of the look-back */
		u32 cnt = 0, inc = 0;
		if (tid < 256) {
#pragma unroll
			for (int w = 0; w < RS_WAVES; ++w) {
				const u32 t = s_whist[w * 256 + tid];
				s_whist[w * 256 + tid] = cnt;
				cnt += t;
			}
			st_agent(&status[(u64)tile * 256 + tid], (tile == 0 ? ST_PREFIX : ST_AGG) | cnt);
			inc = wave_incl_sum<u32>(cnt, lane);
			if (lane == 63)
				s_wsum[wave] = inc;
		}
		TRACE_STAMP(2, tile, 6);
		__syncthreads();
		TRACE_STAMP(2, tile, 7);
		if (tid < 256) {
			u32 doff = inc - cnt;
#pragma unroll
			for (int w = 0; w < 4; ++w)
				if (w < (int)wave)
					doff += s_wsum[w];
			s_doff[tid] = doff;

			u32 excl = 0;
			if (tile > 0) {
				/* walk back over earlier tiles, RS_LOOKBACK_K status words per round trip: tiles start ~25-40 per
				 * microsecond while one dependent global load costs ~0.5-1 us, so a one-word-per-hop walk spends most
				 * of the tile's life here (measured: 39 % of it) */
				int t = (int)tile - 1;
				u32 spins = 0;
				bool done = false;
				u32 rounds = 0;
				TRACE_STAMP(2, tile, 1);
				while (!done) {
					++rounds;
					u32 v[RS_LOOKBACK_K];
#pragma unroll
					for (int j = 0; j < RS_LOOKBACK_K; ++j)
						v[j] = (t - j >= 0) ? ld_agent(&status[(u64)(t - j) * 256 + tid]) : ST_PREFIX;
					int used = 0;
#pragma unroll
					for (int j = 0; j < RS_LOOKBACK_K; ++j) {
						if (!done && used == j) {
							const u32 flag = v[j] & ~ST_MASK;
							if (flag != 0) {
								excl += v[j] & ST_MASK;
								++used;
								if (flag == ST_PREFIX)
									done = true;
							}
						}
					}
					t -= used;
					if (!done && used < RS_LOOKBACK_K) { /* ran into a tile that has not published yet */
						if (++spins > SPIN_LIMIT || (spins % 1024 == 0 && ld_agent(err) & KERR_WATCHDOG)) {
							atomicOr(err, KERR_WATCHDOG);
							break;
						}
						__builtin_amdgcn_s_sleep(1);
					}
				}
				TRACE_STAMP(2, tile, 2);
				TRACE_VALUE(2, tile, 3, rounds);
				TRACE_VALUE(2, tile, 4, spins);
				TRACE_VALUE(2, tile, 5, (u64)((int)tile - 1 - t));
				st_agent(&status[(u64)tile * 256 + tid], ST_PREFIX | (excl + cnt));
			}
			const u64 gbase = digit_base_in[tid] + excl;
			s_goff[tid] = gbase - doff;
			if (tile == num_tiles - 1)
				digit_base_next[tid] = gbase + cnt; /* where the next portion continues this digit */
		}
		TRACE_STAMP(0, tile, 4);
		__syncthreads();
		TRACE_STAMP(0, tile, 5);

#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			if ((wbase + r * 64) < tile_n) {
				const u32 d = kmc_get_byte<SIZE>(key[r], byte_idx);
				const u32 slot = s_doff[d] + s_whist[wave * 256 + d] + rank[r];
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					s_keys[w * TILE + slot] = key[r][w];
			}
		}
		__syncthreads();
		TRACE_STAMP(0, tile, 6);
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const u32 slot = i * RS_BLOCK + tid;
			if (slot < tile_n) {
				u64 x[SIZE];
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					x[w] = s_keys[w * TILE + slot];
				const u32 d = kmc_get_byte<SIZE>(x, byte_idx);
				store_rec<SIZE>(out + (s_goff[d] + slot) * SIZE, x);
			}
		}
		TRACE_STAMP(0, tile, 7);
		__syncthreads(); /* LDS is reused by the next tile of this ticket */
	}
}

template <int SIZE> constexpr size_t rs_lds_bytes()
{
	return 256 * 8 + (size_t)SIZE * RsCfg<SIZE>::TILE * 8 + RS_WAVES * 256 * 4 + 256 * 4 + 4 * 4 + 16;
}

/* ------------------------------------------------------------------------------------------------ compaction
 * ONE read of the sorted records. Thread t of a tile owns ITEMS consecutive records (run detection is then a
 * register-only loop); a run is attributed to the tile that holds its LAST record:
 *   count = index(last) - index(first) + 1, where `first` is found in-thread, else from a workgroup max-scan of
 *   "last run head so far", else (the run started before the tile — at most one such run per tile) by a
 *   wave-cooperative 64-ary lower_bound in the sorted array.
 * Cutoff/clamp semantics: kb_sorter.h:1174-1192 (compare BEFORE clamping; count is uint32). Output records and
 * LUT: kb_sorter.h:1196-1203. Output order must be ascending k-mer => tile order: the tile's offset among counted
 * k-mers comes from a 64-bit decoupled look-back (one word per tile). LUT updates are aggregated per tile: the
 * prefixes of the tile's counted k-mers are a sorted list in LDS, and each run of equal prefixes costs two global
 * atomics (+end, -begin) instead of one per k-mer. */
constexpr u64 ST64_AGG = 1ull << 62, ST64_PREFIX = 2ull << 62, ST64_MASK = (1ull << 62) - 1;

template <int SIZE>
__device__ __forceinline__ u64 run_start_search(const u64 *__restrict__ S, u64 base, u32 lane)
{
	/* smallest i <= base with S[i] == S[base]; executed by one full wave */
	u64 v[SIZE];
	load_rec<SIZE>(S + base * SIZE, v);
	u64 lo = 0, hi = base;
	{ /* round 1: the 64 records just below the tile */
		bool eq = false;
		if (base >= (u64)lane + 1) {
			u64 x[SIZE];
			load_rec<SIZE>(S + (base - 1 - lane) * SIZE, x);
			eq = kmc_equal<SIZE>(x, v);
		}
		const u64 mask = __ballot(eq);
		if (~mask)
			return base - (u64)(__ffsll(~mask) - 1);
		hi = base - 64;
	}
	while (lo < hi) {
		const u64 span = hi - lo;
		const u64 step = (span + 63) / 64;
		const u64 p = lo + (u64)lane * step;
		const bool in = p < hi;
		bool eq = false;
		if (in) {
			u64 x[SIZE];
			load_rec<SIZE>(S + p * SIZE, x);
			eq = kmc_equal<SIZE>(x, v);
		}
		const u64 mask = __ballot(eq), inmask = __ballot(in);
		if (mask) {
			const u64 f = (u64)(__ffsll(mask) - 1);
			hi = lo + f * step;
			lo = f ? (lo + (f - 1) * step + 1) : hi;
		} else {
			const u64 last = 63 - (u64)__clzll(inmask);
			lo = lo + last * step + 1;
		}
	}
	return lo;
}

#ifndef CP_TPB
#define CP_TPB 1 /* tiles per ticket (see RS_TPB) */
#endif
constexpr int CP_STAGE = 16384; /* bytes of output assembled in LDS per window */
constexpr int CP_SHARDS = 32; /* tally shards: same-address device atomics serialise at ~11 ns each */

template <int SIZE>
__global__ void __launch_bounds__(CP_BLOCK) k_compact(const u64 *__restrict__ S, u64 n, DevParams P, uint8_t *__restrict__ out,
                                                       u64 out_capacity, u64 *__restrict__ lut, u64 *stat_shards /* [CP_SHARDS][4] */,
                                                       u64 *out_bytes, u64 *status, u32 *tile_counter, u32 num_tiles, u32 *err)
{
	constexpr int ITEMS = CpCfg<SIZE>::ITEMS;
	constexpr int TILE = CpCfg<SIZE>::TILE;
	__shared__ u64 s_tmp[CP_BLOCK / 64 + 1];
	__shared__ u32 s_tmp32[CP_BLOCK / 64 + 1];
	__shared__ u32 s_pref[TILE];
	__shared__ __attribute__((aligned(16))) uint8_t s_stage[CP_STAGE];
	__shared__ u32 s_tal[3];
	__shared__ u32 s_tile, s_need;
	__shared__ u64 s_run_start, s_tile_off;

	if (threadIdx.x == 0)
		s_tile = atomicAdd(tile_counter, 1u);
	__syncthreads();
	const u32 ticket = s_tile;
	u64 acc_u = 0, acc_b = 0, acc_a = 0; /* thread 0: tallies of this workgroup's tiles */
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const bool use_lut = P.lut_prefix_len != 0 && !P.kff && !P.without_output;

#pragma unroll 1
	for (int it = 0; it < CP_TPB; ++it) {
		const u32 tile = ticket * CP_TPB + it;
		if (tile >= num_tiles)
			break;
		u32 tid = threadIdx.x;
		asm volatile("" : "+v"(tid)); /* keep per-tile addresses out of the loop preheader (register pressure) */
		const u32 lane = tid & 63, wave = tid >> 6;
		if (tid == 0) {
			s_need = 0;
			s_tal[0] = s_tal[1] = s_tal[2] = 0;
		}
		__syncthreads();
		TRACE_STAMP(1, tile, 0);
		TRACE_STAMP(1, tile, 1);
		const u64 base = (u64)tile * TILE;
		const u64 first = base + (u64)tid * ITEMS;
		const int cnt_t = first >= n ? 0 : ((n - first) < (u64)ITEMS ? (int)(n - first) : ITEMS);

		u64 key[ITEMS][SIZE], prev[SIZE], next[SIZE];
#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (i < cnt_t)
				load_rec<SIZE>(S + (first + i) * SIZE, key[i]);
		const bool have_prev = cnt_t > 0 && first > 0;
		const bool have_next = cnt_t > 0 && first + cnt_t < n;
		if (have_prev)
			load_rec<SIZE>(S + (first - 1) * SIZE, prev);
		if (have_next)
			load_rec<SIZE>(S + (first + cnt_t) * SIZE, next);

		/* pass A: head/tail flags, last head position in this thread (as index+1, 0 = none) */
		u32 head_bits = 0, tail_bits = 0;
		u64 last_head1 = 0;
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			if (i < cnt_t) {
				bool head, tail;
				if (i == 0)
					head = !have_prev || !kmc_equal<SIZE>(key[0], prev);
				else
					head = !kmc_equal<SIZE>(key[i], key[i - 1]);
				if (i == cnt_t - 1)
					tail = !have_next || !kmc_equal<SIZE>(key[i], next);
				else
					tail = !kmc_equal<SIZE>(key[i], key[i + 1 < ITEMS ? i + 1 : i]);
				if (head) {
					head_bits |= 1u << i;
					last_head1 = first + i + 1;
				}
				if (tail)
					tail_bits |= 1u << i;
			}
		}
		TRACE_STAMP(1, tile, 2);
		const u64 carry1 = block_excl_max<CP_BLOCK / 64, u64>(last_head1, s_tmp); /* last head before this thread, in-tile */
		TRACE_STAMP(1, tile, 3);
		/* does this thread hold a tail whose run started before the tile? (tail before any head, no head carried in) */
		bool pending = false;
		if (carry1 == 0 && tail_bits) {
			const u32 first_tail = (u32)__ffs((int)tail_bits) - 1;
			const u32 heads_before = head_bits & ((2u << first_tail) - 1);
			pending = heads_before == 0;
		}
		if (pending)
			s_need = 1;
		__syncthreads();
		if (s_need && wave == 0) {
			const u64 st = run_start_search<SIZE>(S, base, lane);
			if (lane == 0)
				s_run_start = st;
		}
		__syncthreads();

		TRACE_STAMP(1, tile, 4);
		/* pass B: counts and classes */
		u32 count[ITEMS];
		u32 counted_bits = 0, nu = 0, nb = 0, na = 0, nc = 0;
		{
			u64 cur_head1 = carry1 ? carry1 : (s_need ? s_run_start + 1 : 0);
#pragma unroll
			for (int i = 0; i < ITEMS; ++i) {
				count[i] = 0;
				if (i < cnt_t) {
					if (head_bits & (1u << i))
						cur_head1 = first + i + 1;
					if (tail_bits & (1u << i)) {
						const u32 c = (u32)(first + i + 1 - (cur_head1 - 1)); /* uint32 like the reference counter */
						++nu;
						if (c < P.cutoff_min)
							++nb;
						else if (c > P.cutoff_max)
							++na;
						else {
							++nc;
							counted_bits |= 1u << i;
							count[i] = c > P.counter_max ? P.counter_max : c;
						}
					}
				}
			}
		}
		/* tallies: wave reduce -> LDS */
		{
			const u32 a = wave_sum<u32>(nu), b2 = wave_sum<u32>(nb), c = wave_sum<u32>(na);
			if (lane == 0) {
				atomicAdd(&s_tal[0], a);
				atomicAdd(&s_tal[1], b2);
				atomicAdd(&s_tal[2], c);
			}
		}
		u32 tile_counted;
		const u32 thread_off = block_excl_sum<CP_BLOCK / 64, u32>(nc, s_tmp32, tile_counted);
		TRACE_STAMP(1, tile, 5);
		if (wave == 0) {
			/* tile offset among counted k-mers: 64-bit decoupled look-back, one word per tile, inspected 64 tiles
			 * at a time by the lanes of wave 0 (a one-word-per-hop walk costs ~1 us per hop) */
			u64 excl = 0;
			if (tile == 0) {
				if (lane == 0)
					st_agent(&status[0], ST64_PREFIX | (u64)tile_counted);
			} else {
				if (lane == 0)
					st_agent(&status[tile], ST64_AGG | (u64)tile_counted);
				long long tbase = (long long)tile - 1;
				u32 spins = 0;
				while (true) {
					const long long t = tbase - (long long)lane;
					const u64 v = t >= 0 ? ld_agent(&status[t]) : ST64_PREFIX; /* virtual empty prefix before tile 0 */
					const u64 flag = v & ~ST64_MASK;
					const u64 m_pref = __ballot(flag == ST64_PREFIX);
					const u64 m_zero = __ballot(flag == 0);
					const int pl = m_pref ? (__ffsll(m_pref) - 1) : 64; /* nearest tile that already has its prefix */
					const u64 need = pl < 63 ? ((2ull << pl) - 1) : ~0ull;
					if (m_zero & need) { /* a tile we depend on has not published yet */
						if (++spins > SPIN_LIMIT || (spins % 1024 == 0 && ld_agent(err) & KERR_WATCHDOG)) {
							if (lane == 0)
								atomicOr(err, KERR_WATCHDOG);
							break;
						}
						__builtin_amdgcn_s_sleep(1);
						continue;
					}
					const u64 part = wave_sum<u64>((int)lane <= pl ? (v & ST64_MASK) : 0ull);
					excl += part; /* valid in lane 0 */
					if (pl < 64)
						break;
					tbase -= 64;
				}
				if (lane == 0)
					st_agent(&status[tile], ST64_PREFIX | (excl + tile_counted));
			}
			if (lane == 0) {
				acc_u += s_tal[0];
				acc_b += s_tal[1];
				acc_a += s_tal[2];
				s_tile_off = excl;
				if (tile == num_tiles - 1)
					*out_bytes = P.without_output ? 0 : (excl + tile_counted) * (u64)rec_bytes;
			}
		}
		TRACE_STAMP(1, tile, 6);
		__syncthreads();

		/* pass C: emit. Records are assembled in an LDS window and streamed out as aligned dwords: per-lane byte
		 * stores straight to HBM were 59 % of this kernel's time. */
		if (!P.without_output) {
			const u64 gbyte0 = s_tile_off * rec_bytes;            /* global byte offset of this tile's first record */
			const u32 tile_bytes = tile_counted * rec_bytes;
			const bool fits = gbyte0 + tile_bytes <= out_capacity; /* uniform over the workgroup */
			if (!fits && tid == 0)
				atomicOr(err, KERR_CAPACITY);
			if (use_lut) {
				u32 j = thread_off;
#pragma unroll
				for (int i = 0; i < ITEMS; ++i)
					if (counted_bits & (1u << i))
						s_pref[j++] = (u32)kmc_remove_suffix<SIZE>(key[i], 2 * (P.k - P.lut_prefix_len));
			}
			for (u32 c0 = 0; fits && c0 < tile_bytes; c0 += CP_STAGE) {
				const u32 c1 = (tile_bytes - c0) < (u32)CP_STAGE ? tile_bytes : c0 + CP_STAGE;
				u32 j = thread_off;
#pragma unroll
				for (int i = 0; i < ITEMS; ++i) {
					if (counted_bits & (1u << i)) {
						const u32 b0 = j * rec_bytes;
						if (b0 < c1 && b0 + rec_bytes > c0) {
							for (u32 q = 0; q < rec_bytes; ++q) {
								const u32 bpos = b0 + q;
								if (bpos >= c0 && bpos < c1) {
									u32 val;
									if (q < P.sbytes)
										val = kmc_get_byte<SIZE>(key[i], P.sbytes - 1 - q);
									else {
										const u32 cq = q - P.sbytes;
										val = count[i] >> (8 * (P.kff ? (P.cbytes - 1 - cq) : cq));
									}
									s_stage[bpos - c0] = (uint8_t)val;
								}
							}
						}
						++j;
					}
				}
				__syncthreads();
				/* window [c0,c1) -> out[gbyte0+c0 ...): leading bytes up to 4-byte alignment, dwords, trailing bytes */
				const u64 g0 = gbyte0 + c0;
				const u32 len = c1 - c0;
				u32 head = (u32)((4 - ((uintptr_t)(out + g0) & 3)) & 3);
				if (head > len)
					head = len;
				const u32 ndw = (len - head) >> 2;
				const u32 tail0 = head + (ndw << 2);
				if (tid < head)
					out[g0 + tid] = s_stage[tid];
				if (tid >= 32 && tid - 32 < len - tail0)
					out[g0 + tail0 + (tid - 32)] = s_stage[tail0 + (tid - 32)];
				u32 *gd = reinterpret_cast<u32 *>(out + g0 + head);
				for (u32 w = tid; w < ndw; w += CP_BLOCK) {
					const uint8_t *sp = s_stage + head + (w << 2);
					gd[w] = (u32)sp[0] | ((u32)sp[1] << 8) | ((u32)sp[2] << 16) | ((u32)sp[3] << 24);
				}
				__syncthreads();
			}
		}
		__syncthreads();
		if (use_lut) {
			for (u32 j = tid; j < tile_counted; j += CP_BLOCK) {
				const u32 pf = s_pref[j];
				if (j + 1 == tile_counted || s_pref[j + 1] != pf)
					atomicAdd(&lut[pf], (u64)(j + 1));
				if (j > 0 && s_pref[j - 1] != pf)
					atomicAdd(&lut[pf], (u64)0 - (u64)j);
			}
		}
		TRACE_STAMP(1, tile, 7);
		__syncthreads(); /* LDS scratch is reused by the next tile */
	}
	if (threadIdx.x == 0) {
		u64 *sh = stat_shards + (size_t)(ticket % CP_SHARDS) * 4;
		if (acc_u)
			atomicAdd(&sh[0], acc_u);
		if (acc_b)
			atomicAdd(&sh[1], acc_b);
		if (acc_a)
			atomicAdd(&sh[2], acc_a);
	}
}

/* stats[0..2] = sum of the shards; stats[3] = n_total = n_rec (kb_sorter.h:1166) */
__global__ void __launch_bounds__(64) k_stats_reduce(const u64 *__restrict__ shards, u64 *__restrict__ stats, u64 n)
{
	const u32 lane = threadIdx.x;
	for (int j = 0; j < 3; ++j) {
		u64 v = lane < CP_SHARDS ? shards[lane * 4 + j] : 0;
		v = wave_sum<u64>(v);
		if (lane == 0)
			stats[j] = v;
	}
	if (lane == 0)
		stats[3] = n;
}

#endif
