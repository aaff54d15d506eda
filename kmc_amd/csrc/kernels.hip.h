/*
 * kmc_amd/csrc/kernels.hip.h — gfx950 (MI355X, wave64) kernels of the KMC stage-2 hot path.
 *
 *   parse    k_parse_packs       : mark the super-k-mer starts of every expander pack in a bitmap (speculative chain
 *                                  resolution in LDS)
 *   expand   k_expand<SIZE>      : super-k-mer bytes -> canonical k-mer records (ref kb_sorter.h:299-362), the sort's byte
 *                                  histograms fused in, digit bases by the last workgroup
 *   sort     k_hist<SIZE>, k_hist_scan : histograms / digit bases where they are not fused (k > 64, sort-only calls)
 *            k_onesweep<SIZE>    : one 8-bit LSD pass, single read + single write per record, decoupled
 *                                  look-back across tiles, wave64 ballot ranking, LDS-staged scatter
 *                                  (replaces raduls_impl.h:546-754 / radix.h:469-842)
 *   compact  k_compact<SIZE>     : run-length count + cutoffs + suffix/counter bytes + prefix LUT + tallies
 *                                  in ONE coalesced read of the sorted records (ref kb_sorter.h:1128-1281)
 *            k_compact_fold      : tally / LUT shards -> the caller's stats and LUT
 *   parse, expand, compact and fold work on GROUPS of bins (Grp* descriptors): the bins that share one sort.
 *
 * All work is integer/byte permutation: HBM/LDS-bound, no MFMA. Design rules applied (cdna_hip_programming.md):
 * 64-wide ballots/popcounts, coalesced 512 B..1 KiB per wave-instruction, per-wave private LDS histograms
 * (no LDS atomics on the ranking path), inter-workgroup hand-off only through single-word relaxed agent-scope
 * atomics where the word IS the flag (Guideline 16 "R2"), every spin bounded. The same source runs on the CPU under
 * tests/hipemu (test infrastructure) — hence the four KMC_* macros below.
 */
#ifndef KMC_AMD_KERNELS_HIP_H
#define KMC_AMD_KERNELS_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include "kmer_ops.h"

typedef kmc_u64 u64;
typedef kmc_u32 u32;

/* Four constructs have no spelling outside the device compiler. They are macros so that tests/hipemu (a host-side emulation of
 * the device language, test infrastructure for the `-m "not gpu"` suite) can run this very source on the CPU; the product build
 * always sees the definitions below. */
#ifndef KMC_DYN_LDS
#define KMC_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[] /* the launch's dynamic LDS */
#define KMC_LAUNDER(x) asm volatile("" : "+v"(x)) /* hide a lane-constant value from LICM: hoisted per-tile addresses cost VGPRs */
#define KMC_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory") /* this wave's outstanding memory operations are performed */
#define KMC_WAVE_LOCKSTEP() ((void)0) /* a point where the code relies on the wave's lanes executing an instruction together */
#endif

/* device-side error bits (word 0 of a stream's 256-byte error block, d_err) */
enum : u32 { KERR_CORRUPT = 1u, KERR_NREC = 2u, KERR_CAPACITY = 4u,
             KERR_WATCHDOG = 8u, /* a look-back TIMED OUT: polled more than SPIN_LIMIT times over more than WATCHDOG_TICKS of the 100 MHz clock */
             KERR_PEER = 0x10u,  /* a look-back gave up early because KERR_WATCHDOG was already set in its stream's word (never set on its own) */
             /* which look-back gave up (diagnostics; reported with KERR_WATCHDOG) */
             KERR_AT_SCATTER = 0x100u, KERR_AT_EXPAND = 0x200u, KERR_AT_COMPACT = 0x400u, KERR_AT_STAGE1 = 0x800u };
/* Words 1.. of the error block: what the FIRST look-back that timed out saw (the host prints them with the error, kmc_hip.hip err_to_code):
 * [1] claim (0 -> 1 by the first reporter), [2] KERR_AT_* | lane or digit << 16, [3] the tile that waited, [4] the tile it waited for, [5..6] the status
 * word it last read there, [7] polls, [8..9] ticks of the 100 MHz clock between its first and its last clock sample, [10] tiles of the launch (0: unknown). */
constexpr int KERR_DIAG_WORDS = 11;

/* tile geometry */
#ifndef RS_BLOCK_THREADS
#define RS_BLOCK_THREADS 1024 /* 8192-record tiles. Measured scatter time of the 1.65 G bin: 256 thr x 16 rec 70.9 ms, 512 x 16 58.5,
                               * 1024 x 8 with the LDS staging in two halves 54.8 (64 VGPRs -> 32 waves/CU instead of 16) */
#endif
constexpr int RS_BLOCK = RS_BLOCK_THREADS, RS_WAVES = RS_BLOCK / 64;                           /* radix scatter workgroup       */
#ifndef CP_BLOCK_THREADS
#define CP_BLOCK_THREADS 512 /* compaction: 8 waves x ROWS rows of 64 records. One 48 M k-mer bin: 256 threads (4096-record tiles) 0.326 ms,
                              * 512 threads 0.237 ms, 256 threads x 8 rows 0.566 ms: per-tile costs (look-back, barriers, end-of-tile atomics) dominate */
#endif
constexpr int CP_BLOCK = CP_BLOCK_THREADS;                                                     /* compaction workgroup          */

#ifndef RS_WORDS_PER_THREAD
#define RS_WORDS_PER_THREAD 16 /* 8-byte words held per thread in a scatter tile, records of 4+ words. Round 4 (k = 127, 32-byte records, quarter workload): 8 words
                                * (2048-record tiles, 256-byte runs) 550 us per launch = 0.51 of the HBM peak, 12: 0.53, 16 in 4 slices (512-byte runs): 506 us = 0.556 */
#endif
#ifndef RS_WORDS_PER_THREAD_23
#define RS_WORDS_PER_THREAD_23 12 /* records of 2-3 words. k = 55 (16-byte records): 8 words (4096-record tiles) 1068 us = 0.536, 12 (6144 records, 384-byte runs) 983 us =
                                   * 0.582, 16 in 4 slices 1015 us = 0.564 */
#endif
#ifndef RS_WORDS_PER_THREAD_1
#define RS_WORDS_PER_THREAD_1 10 /* one-word records (k <= 32): 10 240-record tiles, 320-byte runs. One launch of 412 M records (round 2):
                                  * 8 words 1.82-1.83 ms, 9: 1.83, 10: 1.69-1.74, 12 (9 VGPRs spilled): 1.71, 16 (19 spilled): 2.16 */
#endif
#ifndef RS_STAGES
#define RS_STAGES 2 /* LDS staging slices per tile, records of 1-3 words. 10 words: 2 slices 1.69-1.74 ms, 5 slices 1.88 */
#endif
#ifndef RS_STAGES_WIDE
#define RS_STAGES_WIDE 4 /* records of 4+ words */
#endif
template <int SIZE> struct RsCfg { /* records per thread in a scatter tile */
	static constexpr int WORDS = SIZE == 1 ? RS_WORDS_PER_THREAD_1 : (SIZE <= 3 ? RS_WORDS_PER_THREAD_23 : RS_WORDS_PER_THREAD);
	static constexpr int ITEMS = (WORDS / SIZE) > 2 ? (WORDS / SIZE) : 2;
	static constexpr int TILE = RS_BLOCK * ITEMS;
	static_assert(TILE <= 32768, "tile-relative slots are kept as 16-bit values (0xFFFF marks an absent record)");
	static constexpr int STAGES_REQ = SIZE <= 3 ? RS_STAGES : RS_STAGES_WIDE;
	static constexpr int STAGES = (ITEMS % STAGES_REQ == 0) ? STAGES_REQ : ITEMS; /* LDS staging slices per tile (else: one record per thread and slice) */
};
#ifndef CP_WORDS_PER_THREAD
#define CP_WORDS_PER_THREAD 16 /* 8-byte words per lane of a compaction tile (records of 2+ words): rows per wave = words / SIZE */
#endif
#ifndef CP_WORDS_PER_THREAD_1
#define CP_WORDS_PER_THREAD_1 8 /* one-word records: 8 rows per wave (4096-record tiles), 64 VGPRs -> 4 workgroups per CU. One 48 M k-mer bin:
                                 * 16 rows / 128 VGPRs / 2 per CU 0.225 ms, 8 rows 0.215 ms */
#endif
template <int SIZE> struct CpCfg {
	static constexpr int WORDS = SIZE == 1 ? CP_WORDS_PER_THREAD_1 : CP_WORDS_PER_THREAD;
	static constexpr int ITEMS = (WORDS / SIZE) > 2 ? (WORDS / SIZE) : 2; /* rows of 64 records per wave */
	static constexpr int TILE = CP_BLOCK * ITEMS;
#ifndef CP_MIN_WAVES_1
#define CP_MIN_WAVES_1 8
#endif
	static constexpr int MIN_WAVES = SIZE == 1 ? CP_MIN_WAVES_1 : 4; /* waves per SIMD the register allocator must leave room for */
};

/* per-run constants handed to the kernels by value */
struct DevParams {
	u32 k, both_strands, cutoff_min, cutoff_max, counter_max, lut_prefix_len, sbytes, cbytes, kff, without_output;
};

/* Group descriptors: parse, expand, compaction and fold work on up to GRP_MAX bins per launch (the bins of one grouped sort, kmc_hip.hip
 * run_group_device_t; a bin on its own is a group of one). Passed BY VALUE as kernel arguments: a workgroup finds its bin by walking the
 * prefix array of work items (packs / slices / tiles) — scalar loads from the kernel-argument segment, nothing to upload. */
constexpr int GRP_MAX = 16;
struct GrpParse {
	u32 g, pack_prefix[GRP_MAX + 1]; /* packs of bin b: [pack_prefix[b], pack_prefix[b+1]) */
	const uint8_t *data[GRP_MAX];
	const u64 *pack_start[GRP_MAX];
	u32 *bitmap[GRP_MAX];
};
struct GrpExpand {
	u32 g, chunk_prefix[GRP_MAX + 1]; /* EXP_CHUNK-byte slices */
	const uint8_t *data[GRP_MAX];
	u64 size[GRP_MAX], n_rec[GRP_MAX];
	const u32 *bitmap[GRP_MAX];
	u64 *out[GRP_MAX];    /* the bin's slice of the shared record array */
	u64 *status[GRP_MAX]; /* look-back words, one per slice of the bin */
	u64 tag[GRP_MAX];     /* the bin's number inside the group, shifted to bit 2k of the record word that holds it */
	u64 *pair_out[GRP_MAX]; /* indirect sort (kmc_hip.hip run_group_device_t): the bin's slice of the group's (top four key bytes << 32 | record number) array, or all NULL */
	u64 *pair_base;         /* ... and the array's first word: a record's number is its distance from there */
};
struct GrpCompact {
	u32 g, tile_prefix[GRP_MAX + 1];
	const u64 *S[GRP_MAX]; /* the bin's slice of the sorted array */
	u64 n[GRP_MAX];
	uint8_t *out[GRP_MAX];
	u64 out_capacity[GRP_MAX];
	u64 *lut_base[GRP_MAX]; /* the bin's LUT (or its shards) */
	u64 *tally[GRP_MAX];    /* [CP_SHARDS][4] */
	u64 *out_bytes[GRP_MAX];
	u64 *status[GRP_MAX];    /* one word per tile: look-back word, or (two-phase output) the tile's number of counted k-mers */
	uint8_t *scratch[GRP_MAX]; /* two-phase output: the bin's slice of the record array the sort left free */
};
struct GrpFold {
	const u64 *tally[GRP_MAX];
	u64 *stats[GRP_MAX];
	u64 n[GRP_MAX];
	const u64 *lut_base[GRP_MAX];
	u64 *lut_out[GRP_MAX];
	/* two-phase output: tile counts -> exclusive prefixes (in place, total at [n_tiles]), out_bytes, capacity check */
	u64 *status[GRP_MAX];
	u32 n_tiles[GRP_MAX];
	u64 *out_bytes[GRP_MAX];
	u64 out_capacity[GRP_MAX];
};
struct GrpGather { /* two-phase output: tile t's records move from scratch + t * tile_pitch to out + prefix[t] * rec_bytes */
	u32 g, tile_prefix[GRP_MAX + 1];
	const uint8_t *scratch[GRP_MAX];
	const u64 *prefix[GRP_MAX]; /* [n_tiles + 1] */
	uint8_t *out[GRP_MAX];
	u64 out_capacity[GRP_MAX];
	const u64 *src_rec[GRP_MAX]; /* NULL, or (k_bucket_rank's tiles) the first record of every tile: tile t's records lie at scratch + src_rec[t] * tile_pitch */
};
/* bin of work item `item` (wave-uniform) */
__device__ __forceinline__ u32 grp_find(const u32 (&prefix)[GRP_MAX + 1], u32 g, u32 item)
{
	u32 b = 0;
	while (b + 1 < g && item >= prefix[b + 1])
		++b;
	return b;
}

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

/* inclusive wave scan of 32-bit values on the VALU's data-parallel primitives: four row shifts inside the rows of 16 lanes, then the last lane of row 0
 * (and of row 2) broadcast into row 1 (3), then lane 31 into rows 2-3 — six v_add with DPP operands instead of six ds_bpermute round trips through the LDS
 * pipeline (what __shfl_up compiles to). gfx9-family controls (row_bcast exists up to gfx950). The CPU emulation of tests/hipemu keeps the shuffles. */
__device__ __forceinline__ u32 wave_incl_sum_u32(u32 v, u32 lane)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KMC_HIPEMU)
	(void)lane;
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
	v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
	return v;
#else
	return wave_incl_sum<u32>(v, lane);
#endif
}
/* lane i receives lane i-1's value, lane 0 receives `first` (one v_mov with a DPP wave shift instead of a ds_bpermute + select) */
__device__ __forceinline__ u32 wave_shift_up1(u32 v, u32 first, u32 lane)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KMC_HIPEMU)
	(void)lane;
	return (u32)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
	const u32 t = __shfl_up(v, 1);
	return lane == 0 ? first : t;
#endif
}
/* Block-wide exclusive sum of one 32-bit value per thread with ONE barrier: every wave scans the NW wave totals itself (block_excl_sum below lets wave 0
 * do it and needs two more barriers). `tmp` has NW entries and must not be written again before the caller's next barrier. */
template <int NW> __device__ __forceinline__ u32 block_excl_sum_1b(u32 v, u32 *tmp, u32 &total)
{
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const u32 inc = wave_incl_sum_u32(v, lane);
	if (lane == 63)
		tmp[wave] = inc;
	__syncthreads();
	const u32 w = lane < (u32)NW ? tmp[lane] : 0u;
	const u32 winc = wave_incl_sum_u32(w, lane);
	total = __shfl(winc, NW - 1);
	return __shfl(winc - w, (int)wave) + inc - v;
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

/* 64-bit look-back words (one per tile/slice): [63:62] flag (0 empty, 1 aggregate, 2 inclusive prefix), [61:0] count */
constexpr u64 ST64_AGG = 1ull << 62, ST64_PREFIX = 2ull << 62, ST64_MASK = (1ull << 62) - 1;
/* Look-back watchdog. A wait is over when the tile in front publishes — microseconds — or never (a ticket counter that did not start at zero, a
 * status area somebody else wrote into): the watchdog turns "never" into KMC_HIP_EINTERNAL instead of a hang. It counts POLLS, nothing else: a poll is an
 * s_sleep + a device-coherent load (~1 us), so SPIN_LIMIT = 2^24 polls is on the order of 10+ seconds of one wave waiting — three to four orders of magnitude
 * beyond the longest wait the storm test produces (16 streams sharing the CUs). A clock-based second condition (wall_clock64) was tried in round 4 and taken out
 * again: inside lb_blocked it cost k_onesweep<1> 13-35 spilled registers (the comment had kept describing it). */
constexpr u32 SPIN_LIMIT = 1u << 24;
struct LbWatch {
	u32 spins = 0;
};
/* one blocked poll of a look-back; true = give up (lb_gave_up then sets the error word). Kept to round 3's two compares: anything more in here (a clock, the
 * diagnostics) cost k_onesweep<1> 13-35 more spilled registers at its 64-register budget and k_expand a wave per SIMD. */
__device__ __forceinline__ bool lb_blocked(LbWatch &w, const u32 *err)
{
	++w.spins;
	return w.spins > SPIN_LIMIT || ((w.spins & 1023u) == 0 && (ld_agent(err) & KERR_WATCHDOG) != 0);
}
/* after the loop of a look-back that gave up: a genuine time-out (the polls ran out) leaves its diagnostics and KERR_WATCHDOG | at_bits, a look-back that
 * only saw somebody else's KERR_WATCHDOG leaves KERR_PEER */
__device__ __forceinline__ void lb_gave_up(const LbWatch &w, u32 *err, u32 at_bits, u32 who, u32 tile, long long waiting_for, u32 num_tiles)
{
	if (w.spins > SPIN_LIMIT) {
		if (atomicCAS(&err[1], 0u, 1u) == 0u) {
			err[2] = (at_bits & ~KERR_WATCHDOG) | (who << 16);
			err[3] = tile;
			err[4] = (u32)waiting_for;
			err[7] = w.spins;
			err[10] = num_tiles;
		}
		atomicOr(err, KERR_WATCHDOG | at_bits);
	} else
		atomicOr(err, KERR_PEER);
}

#ifndef LB64_WINDOWS
#define LB64_WINDOWS 1 /* 64-tile windows fetched per round trip of the 64-bit look-back. One 48 M k-mer bin, compaction: 1 window 0.222 ms,
                        * 4: 0.253, 8: 0.273, 16: 0.311 */
#endif
/* Decoupled look-back over 64-bit status words, executed by ONE full wave of tile `tile`: publishes the tile's aggregate, walks back over
 * earlier tiles until it meets an inclusive prefix, publishes the tile's own inclusive prefix and returns the exclusive one (valid in lane 0).
 * Tiles start ~20 ns apart and publish their aggregate microseconds later, so the nearest tile that already HAS its prefix is hundreds of
 * tiles back (compaction: ~6 windows of 64, 6.4 of a tile's 16 us, profiles/r02/trace_report_compact_4096.txt). Fetching several windows per
 * round trip does NOT help (measured above: the walk got slower, 8.4 us at 8 windows): as in the scatter kernel the walk is bound by the
 * device-coherent status bytes that ~1000 tiles in flight pull from the same few cache lines, not by the number of round trips. A window with
 * an unpublished tile in front of the nearest prefix stops the round: what lies before it is kept, the rest is fetched again after a short
 * sleep. The spin is bounded (watchdog). */
__device__ __forceinline__ u64 lookback64(u64 *status, u32 tile, u64 aggregate, u32 lane, u32 *err, u32 err_watchdog_bit)
{
	if (tile == 0) {
		if (lane == 0)
			st_agent(&status[0], ST64_PREFIX | aggregate);
		return 0;
	}
	if (lane == 0)
		st_agent(&status[tile], ST64_AGG | aggregate);
	long long tbase = (long long)tile - 1;
	u64 acc = 0; /* this lane's share of the exclusive prefix */
	LbWatch watch;
	bool done = false;
	while (!done) {
		u64 v[LB64_WINDOWS];
#pragma unroll
		for (int j = 0; j < LB64_WINDOWS; ++j) {
			const long long t = tbase - 64 * j - (long long)lane;
			v[j] = t >= 0 ? ld_agent(&status[t]) : ST64_PREFIX; /* virtual empty prefix before tile 0 */
		}
		int used = 0;
		bool blocked = false;
#pragma unroll
		for (int j = 0; j < LB64_WINDOWS; ++j) {
			if (!done && !blocked) { /* wave-uniform */
				const u64 flag = v[j] & ~ST64_MASK;
				const u64 m_pref = __ballot(flag == ST64_PREFIX);
				const u64 m_zero = __ballot(flag == 0);
				const int pl = m_pref ? (__ffsll(m_pref) - 1) : 64; /* nearest tile of this window that already has its prefix */
				const u64 need = pl < 63 ? ((2ull << pl) - 1) : ~0ull;
				if (m_zero & need)
					blocked = true; /* a tile we depend on has not published yet */
				else {
					if ((int)lane <= pl)
						acc += v[j] & ST64_MASK;
					++used;
					if (pl < 64)
						done = true;
				}
			}
		}
		tbase -= 64 * used;
		if (!done && blocked) {
			if (lb_blocked(watch, err))
				break;
			__builtin_amdgcn_s_sleep(1);
		}
	}
	if (!done && lane == 0)
		lb_gave_up(watch, err, err_watchdog_bit, lane, tile, tbase, 0u);
	const u64 excl = wave_sum<u64>(acc);
	if (lane == 0)
		st_agent(&status[tile], ST64_PREFIX | (excl + aggregate));
	return excl;
}

/* ------------------------------------------------------------------------------------------------ parse
 * The bin image is a chain of variable-length records (one byte of length information per record); the only
 * random-access entry points the caller has are the expander-pack boundaries (CExpanderPackDesc, queues.h:376-396;
 * <= 4096 super-k-mers each, kb_collector.h:46). One workgroup per pack marks every record start in a global
 * bitmap (1 bit per input byte). Inside a pack the chain is resolved in LDS, PARSE_CHUNK bytes at a time, speculatively:
 * every byte position is treated as a possible record start (see the three levels in the kernel). History: v1 walked
 * the chain with one lane per pack (4096 dependent global loads per pack, ~1 us each: 8 ms per bin however small);
 * v2 used pointer doubling over all positions (log2 rounds of two LDS sweeps: 53 k cycles per chunk). */
constexpr int PARSE_CHUNK = 4096, PARSE_SUB = 128, PARSE_NSUB = PARSE_CHUNK / PARSE_SUB;
#ifndef PARSE_CAND_POS
#define PARSE_CAND_POS 16 /* entry positions per sub-block resolved speculatively; deeper entries take the exact slow path */
#endif
#ifndef PARSE_THREADS
#define PARSE_THREADS 64 /* ONE wave per pack (round 4; 256 threads before: a pack is a serial chain of ~12 chunks x ~65 dependent LDS hops whatever the width of the
                          * workgroup, so what counts is how many packs a CU holds at once — 8 workgroups of 4 waves (13 KB of LDS each) meant three rounds of packs
                          * per group of bins; 28 single waves (5.7 KB each) hold every pack of a group at once: 0.27 -> see DESIGN ms per 190 M-record group) */
#endif
constexpr int PARSE_CAND = PARSE_CAND_POS, PARSE_BLOCK = PARSE_THREADS;
static_assert(PARSE_CAND >= 8 && PARSE_CAND <= PARSE_SUB && (PARSE_CAND & (PARSE_CAND - 1)) == 0, "PARSE_CAND");
static_assert(PARSE_BLOCK >= PARSE_NSUB && (PARSE_NSUB * PARSE_CAND) % PARSE_BLOCK == 0 && (PARSE_CHUNK / 16) % PARSE_BLOCK == 0, "PARSE_THREADS");

__global__ void __launch_bounds__(PARSE_BLOCK) k_parse_packs(const GrpParse gp, u32 k, u32 *err)
{
	/* Three levels per PARSE_CHUNK bytes staged in LDS (the chain has <= chunk/Lmin hops; walking it serially costs one
	 * dependent load per hop):
	 *   L1  the first PARSE_CAND positions of every 128-byte sub-block as possible entries, in parallel: X(p) = where the chain
	 *       started at p leaves the sub-block (<= 128/Lmin hops; a thread interleaves its chains so the LDS latency pipelines)
	 *   L2  one lane hops sub-block to sub-block from the chunk's entry offset: 32 dependent LDS reads instead of ~500
	 *   L3  one lane per sub-block walks it from its now-known entry and builds the sub-block's 128 start bits */
	constexpr int NV = PARSE_CHUNK / 16 / PARSE_BLOCK; /* 16-byte pieces of a chunk per thread */
	__shared__ __attribute__((aligned(16))) uint8_t s_b[PARSE_CHUNK];
	__shared__ unsigned short s_X[PARSE_NSUB * PARSE_CAND]; /* [sub-block][candidate] */
	__shared__ unsigned short s_ent[PARSE_NSUB]; /* entry position + 1 of each sub-block (0 = chain does not start here) */
	__shared__ u32 s_vis[PARSE_CHUNK / 32];
	__shared__ u32 s_exit;
	const u32 tid = threadIdx.x;
	if (blockIdx.x >= gp.pack_prefix[gp.g])
		return;
	const u32 bin = grp_find(gp.pack_prefix, gp.g, blockIdx.x);
	const u32 p = blockIdx.x - gp.pack_prefix[bin];
	const uint8_t *__restrict__ data = gp.data[bin];
	u32 *__restrict__ bitmap = gp.bitmap[bin];
	const u64 pos0 = gp.pack_start[bin][p], end = gp.pack_start[bin][p + 1];
	/* Chunks are cut at multiples of PARSE_CHUNK of the IMAGE (not of the pack), and the first one starts at the 16-byte boundary
	 * below the pack start: every chunk is staged with aligned 16-byte loads (byte loads cost 16 instructions per
	 * thread and chunk). The few bytes in front of the pack are never visited: the chain starts at `entry`. */
	u32 entry = (u32)(pos0 & 15); /* offset inside the current chunk of the first record start */
	u64 c_next = pos0 & ~15ull;
	static_assert(PARSE_CHUNK == 4096, "chunk boundaries are computed with shifts");
	/* the chunk after the one being resolved is already on its way (NV 16-byte registers per thread): a pack is ~12 chunks that depend
	 * on each other through `entry`, and each used to start with an exposed trip to HBM */
	uint4 staged[NV];
#pragma unroll
	for (int v = 0; v < NV; ++v)
		staged[v] = make_uint4(0, 0, 0, 0);
	if (c_next < end) {
		const u64 b0 = ((c_next >> 12) + 1) << 12;
		const u64 lim = (b0 < end ? b0 : end) - c_next;
#pragma unroll
		for (int v = 0; v < NV; ++v)
			if ((u64)(tid + v * PARSE_BLOCK) * 16 < lim)
				staged[v] = reinterpret_cast<const uint4 *>(data + c_next)[tid + v * PARSE_BLOCK]; /* 16-byte aligned; the image has >= 256 readable bytes of slack */
	}
	while (c_next < end) {
		const u64 c0 = c_next;
		const u64 bound = ((c0 >> 12) + 1) << 12;
		const u64 c1 = bound < end ? bound : end;
		const u32 clen = (u32)(c1 - c0);
		c_next = c1;
#pragma unroll
		for (int v = 0; v < NV; ++v)
			if ((tid + v * PARSE_BLOCK) * 16 < clen)
				reinterpret_cast<uint4 *>(s_b)[tid + v * PARSE_BLOCK] = staged[v]; /* every reader of the previous chunk is past the barrier that ends the loop body */
		if (c1 < end) {
			const u64 b1 = c1 + PARSE_CHUNK; /* c1 is a multiple of PARSE_CHUNK here */
			const u64 lim = (b1 < end ? b1 : end) - c1;
#pragma unroll
			for (int v = 0; v < NV; ++v)
				if ((u64)(tid + v * PARSE_BLOCK) * 16 < lim)
					staged[v] = reinterpret_cast<const uint4 *>(data + c1)[tid + v * PARSE_BLOCK];
		}
		if (entry >= clen) { /* only for a ragged image; the final check below reports it */
			entry -= clen;
			__syncthreads();
			continue;
		}
		if (tid < PARSE_NSUB)
			s_ent[tid] = 0;
		__syncthreads();
		/* L1. Thread t runs the candidates c = t + PARSE_BLOCK i (sub-block c / PARSE_CAND, position c % PARSE_CAND in it); the
		 * hop loop stops as soon as the wave has no chain left inside its sub-blocks. */
		{
			constexpr int NC = PARSE_NSUB * PARSE_CAND / PARSE_BLOCK;
			/* a record is at most maxlen = 1 + ceil((k+255)/4) >= 65 bytes (e <= 255, splitter.cpp:656), so a chain can enter a
			 * sub-block anywhere in its first maxlen positions — but a record of real data is far shorter than the format allows
			 * (e rarely exceeds a few dozen), so only the first PARSE_CAND positions are speculated on (2 rounds instead of the 9
			 * that maxlen asks for at k=27); the rare chain that enters deeper is walked by L2 itself.
			 * Exactness does not depend on the bound. */
			u32 q[NC];
#pragma unroll
			for (int i = 0; i < NC; ++i) {
				const u32 c = tid + (u32)PARSE_BLOCK * i;
				q[i] = (c / PARSE_CAND) * PARSE_SUB + (c % PARSE_CAND);
			}
			const u32 max_hops = PARSE_SUB / (1 + ((k + 3) >> 2)) + 1;
			for (u32 h = 0; h < max_hops; ++h) {
				bool moved = false;
#pragma unroll
				for (int i = 0; i < NC; ++i) {
					const u32 sb_end = (((tid + (u32)PARSE_BLOCK * i) / PARSE_CAND) + 1) * PARSE_SUB;
					if (q[i] < sb_end && q[i] < clen) {
						q[i] += 1 + ((k + s_b[q[i]] + 3) >> 2);
						moved = true;
					}
				}
				if (!__any(moved))
					break;
			}
#pragma unroll
			for (int i = 0; i < NC; ++i)
				s_X[tid + (u32)PARSE_BLOCK * i] = (unsigned short)q[i]; /* < clen + 130; entries at or behind clen are never looked up */
		}
		__syncthreads();
		/* L2 */
		if (tid == 0) {
			u32 q = entry;
			while (q < clen) {
				s_ent[q / PARSE_SUB] = (unsigned short)(q + 1);
				if ((q & (PARSE_SUB - 1)) < (u32)PARSE_CAND)
					q = s_X[(q / PARSE_SUB) * PARSE_CAND + (q & (PARSE_SUB - 1))];
				else { /* entered deeper than L1 speculated: walk this sub-block here */
					const u32 sb_end = (q | (PARSE_SUB - 1)) + 1;
					while (q < sb_end && q < clen)
						q += 1 + ((k + s_b[q] + 3) >> 2);
				}
			}
			s_exit = q;
		}
		__syncthreads();
		/* L3 */
		if (tid < PARSE_NSUB) {
			u32 w[PARSE_SUB / 32] = {0, 0, 0, 0};
			const u32 e1 = s_ent[tid];
			if (e1) {
				u32 q = e1 - 1;
				const u32 sb0 = tid * PARSE_SUB, sb_end = sb0 + PARSE_SUB;
				while (q < sb_end && q < clen) {
					const u32 r = q - sb0;
#pragma unroll
					for (int j = 0; j < PARSE_SUB / 32; ++j)
						if ((r >> 5) == (u32)j)
							w[j] |= 1u << (r & 31);
					q += 1 + ((k + s_b[q] + 3) >> 2);
				}
			}
#pragma unroll
			for (int j = 0; j < PARSE_SUB / 32; ++j)
				s_vis[tid * (PARSE_SUB / 32) + j] = w[j];
		}
		__syncthreads();
		const u32 next_entry = s_exit - clen;
		/* publish the chunk's bits [c0, c0+clen) into the global bitmap (bit i of the bitmap = byte i of the image) */
		{
			const u32 sh = (u32)(c0 & 31);
			const u64 gw0 = c0 >> 5;
			const u32 n_gw = (u32)(((c0 + clen + 31) >> 5) - gw0);
			for (u32 g = tid; g < n_gw; g += PARSE_BLOCK) {
				/* global word g covers local bits [32g - sh, 32g - sh + 32) */
				const int lw = (int)g - (sh ? 1 : 0);
				const u32 lo = (lw >= 0 && lw < PARSE_CHUNK / 32) ? s_vis[lw] : 0;
				const u32 hi = (sh && lw + 1 >= 0 && lw + 1 < PARSE_CHUNK / 32) ? s_vis[lw + 1] : 0;
				const u32 word = sh ? ((lo >> (32 - sh)) | (hi << sh)) : lo;
				if (word) {
					if (g == 0 || g == n_gw - 1)
						atomicOr(&bitmap[gw0 + g], word); /* boundary words are shared with the neighbouring chunk/pack */
					else
						bitmap[gw0 + g] = word;
				}
			}
		}
		entry = next_entry;
		__syncthreads();
	}
	if (tid == 0 && entry != 0)
		atomicOr(err, KERR_CORRUPT); /* the last record does not end on the pack boundary */
}

/* ------------------------------------------------------------------------------------------------ expand
 * Fully parallel over EXP_CHUNK-byte slices of the image (packs no longer matter): a workgroup reads its slice and
 * the slice's start bits, lists the super-k-mers that START in it (position, first k-mer index) in LDS, gets the
 * slice's k-mer offset by a 64-bit decoupled look-back over slices, and then runs one THREAD per k-mer: its super-k-mer
 * from one start bit per k-mer + per-row counts (round 4; a binary search of the list in round 1, a per-k-mer index
 * built by windowed max-scans in rounds 2-3), window extraction from the LDS copy of the bytes by funnel shifts, reverse
 * complement by bit tricks (kmer_ops.h), canonical = min; consecutive threads write consecutive records. The byte
 * histograms of the sort's HBM passes are accumulated in LDS on the way (one global flush per persistent workgroup),
 * which removes the separate 8 B/record histogram read of the first version; for records of two words and more the
 * (key top, record number) pair of the indirect sort is written next to the record (kmc_hip.hip run_group_device_t). */
#ifndef EXP_BLOCK_THREADS
#define EXP_BLOCK_THREADS 512 /* measured on the 1.65 G k-mer bin: 256 thr 13.7 ms, 512 thr 9.2 ms, 1024 thr 12.1 ms */
#endif
#ifndef EXP_CHUNK_BYTES
#define EXP_CHUNK_BYTES 16384 /* the 1.65 G k-mer bin: 4 KB slices 9.15 ms, 8 KB 7.85, 16 KB 7.15 (fewer scans and look-backs per byte) */
#endif
constexpr u32 EXP_FUSE_MAX_PASS = 16; /* the sort's histograms are fused into the expansion up to this many passes (k <= 64) */
constexpr int EXP_CHUNK = EXP_CHUNK_BYTES, EXP_TAIL = 160, EXP_BLOCK = EXP_BLOCK_THREADS;
constexpr u32 EXP_MAX_K = 4 * EXP_CHUNK; /* k-mers of the records that start in one slice: fewer than 4 per byte of them (a record of L bytes holds < 4 L - k) */
/* x >> s for a shift count that is the same in every lane (0..63). The plain expression is a 64-bit shift instruction per lane (the compiler cannot know the
 * count is uniform, nor that a funnel shift of two dwords by 0..31 bits is v_alignbit_b32); here: one funnel shift, one 32-bit shift, two selects on a scalar
 * condition, no branch */
__device__ __forceinline__ u64 shr64_uniform(u64 x, u32 s)
{
	const u32 hi = (u32)(x >> 32), lo = (u32)x;
	const u32 t = hi >> (s & 31), al = __builtin_amdgcn_alignbit(hi, lo, s & 31);
	const u32 m = 0u - (u32)(s < 32); /* as a mask, not as a condition: the compiler turns a select on a uniform condition into a branch */
	return ((u64)(t & m) << 32) | ((al & m) | (t & ~m));
}
/* most super-k-mers that can START inside one slice: a record is 1 + ceil((k+e)/4) >= 1 + ceil(k/4) bytes long. The two LDS lists are
 * sized by this (k=27: 1025 entries instead of a worst case of 4096 -> 38 KB instead of 57 KB per workgroup: 4 workgroups per CU, not 2) */
__host__ __device__ constexpr u32 exp_max_sk(u32 k) { return (u32)EXP_CHUNK / (1 + ((k + 3) >> 2)) + 2; }

static_assert(EXP_CHUNK / 32 <= EXP_BLOCK && EXP_CHUNK <= 65536, "one bitmap word per thread; 16-bit positions inside a slice");
static_assert(EXP_BLOCK >= 256, "the last workgroup scans 256 digits per pass, one per thread");
#ifndef EXP_FULL_OCCUPANCY_SIZE
#define EXP_FULL_OCCUPANCY_SIZE 4 /* record widths whose k_expand is compiled for 8 waves per SIMD (<= 64 VGPRs, <= 96 SGPRs): four 512-thread workgroups per CU fit their LDS */
#endif
template <int SIZE, bool FUSE_HIST>
__global__ void __launch_bounds__(EXP_BLOCK, (SIZE <= EXP_FULL_OCCUPANCY_SIZE ? 8 : (SIZE <= 4 ? 6 : 4))) k_expand(const GrpExpand ge, u32 k, u32 both_strands, u32 n_pass, u64 *__restrict__ ghist, u32 *ticket_ctr,
                                                 u32 *err, u64 *__restrict__ digit_base, u32 *done_ctr, u32 pass_lo)
{
	/* n_pass histograms are fused: those of key bytes pass_lo .. pass_lo + n_pass - 1 (the hybrid sort of bucket_sort.hip.h only sends the top
	 * bytes of the key through HBM passes; pass_lo = 0 and n_pass = ceil(k/4) is the plain LSD sort) */
	/* The slices of all bins of the group form one ticket space. ge.tag[bin] is OR-ed into the record word that holds bit 2k — the spare
	 * bits of the top radix digit (8 ceil(k/4) - 2k of them): bins expanded into one record array with tags 0, 1, 2 ... are sorted by ONE set
	 * of passes into bin-major order (kmc_hip.hip run_group_device_t); 0 for a bin on its own. */
	const u32 n_chunks = ge.chunk_prefix[ge.g];
	const u32 MAX_SK = exp_max_sk(k);
	const u64 kmask = 2 * k >= 64 ? ~0ull : (1ull << (2 * k)) - 1; /* one-word records: the k-mer's bits */
	KMC_DYN_LDS(unsigned char, s_raw);
	u64 *s_base = reinterpret_cast<u64 *>(s_raw);                             /* [2] (16 bytes keeps s_b 16-B aligned) */
	uint8_t *s_b = s_raw + 16;                                                /* [EXP_CHUNK + EXP_TAIL] */
	u32 *s_sk = reinterpret_cast<u32 *>(s_b + EXP_CHUNK + EXP_TAIL);          /* [MAX_SK + 1] per super-k-mer: byte position << 16 | first k-mer (both < 2^16: a slice
	                                                                           * is <= 65536 bytes, and a record of L bytes holds fewer than 4 L k-mers) */
	u32 *s_tmp = s_sk + MAX_SK + 1;                                           /* [24] scan scratch */
	u32 *s_ticket = s_tmp + 24;                                                /* [3] */
	u32 *s_h = s_ticket + 3;                                                  /* [n_pass * 256] when FUSE_HIST */
	u32 *s_start = s_h + (FUSE_HIST ? n_pass * 256 : 0);                      /* [EXP_MAX_K / 32] one bit per k-mer of the slice; bit p: a super-k-mer starts at k-mer p + 1 */
	unsigned short *s_rowsk = reinterpret_cast<unsigned short *>(s_start + EXP_MAX_K / 32); /* [EXP_MAX_K / 64] bits in front of each row of 64 */

	if (FUSE_HIST) {
		for (u32 i = threadIdx.x; i < n_pass * 256; i += EXP_BLOCK)
			s_h[i] = 0;
	}
	/* (Round 4 tried to draw the NEXT slice's ticket while the current slice is worked on, to hide the atomic's round trip: 0.63 -> 0.93 ms. A ticket drawn a slice
	 * early is a tile of the look-back that publishes a slice late, and every slice behind it polls for that long.) */
	while (true) {
		__syncthreads();
		if (threadIdx.x == 0)
			s_ticket[0] = atomicAdd(ticket_ctr, 1u);
		__syncthreads();
		const u32 cg = s_ticket[0];
		if (cg >= n_chunks)
			break;
		const u32 bin = (u32)__builtin_amdgcn_readfirstlane((int)grp_find(ge.chunk_prefix, ge.g, cg));
		const u32 c = cg - ge.chunk_prefix[bin];                 /* slice of the bin */
		const u32 bin_chunks = ge.chunk_prefix[bin + 1] - ge.chunk_prefix[bin];
		const uint8_t *__restrict__ data = ge.data[bin];
		const u32 *__restrict__ bitmap = ge.bitmap[bin];
		const u64 size = ge.size[bin], n_rec = ge.n_rec[bin], tag = ge.tag[bin];
		u64 *__restrict__ out = ge.out[bin];
		u64 *__restrict__ pair_out = ge.pair_out[bin];
		u64 *status = ge.status[bin];
		u32 tid = threadIdx.x;
		KMC_LAUNDER(tid);
		const u32 lane = tid & 63, wave = tid >> 6;
		const u64 c0 = (u64)c * EXP_CHUNK;
		const u32 clen = (size - c0) < (u64)EXP_CHUNK ? (u32)(size - c0) : (u32)EXP_CHUNK;
		const u32 avail = (size - c0) < (u64)(EXP_CHUNK + EXP_TAIL) ? (u32)(size - c0) : (u32)(EXP_CHUNK + EXP_TAIL);
		/* stage the slice (+ tail for records that start in it and end in the next one): 16-byte loads */
		{
			const uint4 *g = reinterpret_cast<const uint4 *>(data + c0); /* c0 is a multiple of EXP_CHUNK; `data` is 16-B aligned */
			uint4 *l = reinterpret_cast<uint4 *>(s_b);
			for (u32 i = tid; i < (avail + 15) / 16; i += EXP_BLOCK)
				l[i] = g[i]; /* the image has >= 256 readable bytes of slack after `size` */
		}
		for (u32 i = tid; i < EXP_MAX_K / 32; i += EXP_BLOCK)
			s_start[i] = 0; /* (nobody reads the previous slice's bits any more: the barrier at the top of the loop) */
		/* this thread's 32 positions = one bitmap word */
		const u32 bits = (tid < (u32)EXP_CHUNK / 32 && tid * 32 < clen) ? bitmap[(c0 >> 5) + tid] : 0; /* one bitmap word (32 positions) per thread */
		__syncthreads();
		u32 my_sk = (u32)__popc(bits), my_k = 0;
		{
			u32 bb = bits;
			while (bb) {
				const u32 bpos = (u32)__ffs((int)bb) - 1;
				bb &= bb - 1;
				my_k += (u32)s_b[tid * 32 + bpos] + 1;
			}
		}
		/* one scan for both counts: super-k-mers in the low 14 bits (a slice holds <= 8194 record starts, at k = 1), k-mers above (< 2^16 in a record chain, see s_sk) */
		u32 tot_packed;
		const u32 off_packed = block_excl_sum<EXP_BLOCK / 64, u32>(my_sk | (my_k << 14), s_tmp, tot_packed);
		const u32 tot_sk = tot_packed & 0x3FFFu, tot_k = tot_packed >> 14, off_sk = off_packed & 0x3FFFu, off_k = off_packed >> 14;
		{
			u32 bb = bits, i = off_sk, ko = off_k;
			while (bb) {
				const u32 bpos = (u32)__ffs((int)bb) - 1;
				bb &= bb - 1;
				if (i < MAX_SK && ko < EXP_MAX_K) {
					s_sk[i] = ((tid * 32 + bpos) << 16) | ko;
					if (ko)
						atomicOr(&s_start[(ko - 1) >> 5], 1u << ((ko - 1) & 31)); /* bit p: a super-k-mer starts at k-mer p + 1 (an atomic: up to 32 of one k-mer each share a word) */
				}
				ko += (u32)s_b[tid * 32 + bpos] + 1;
				++i;
			}
		}
		/* slice offset among k-mers: decoupled look-back, one 64-bit word per slice, 64 slices per round trip */
		/* (Round 4 tried to publish the aggregate here and walk back only after the first window's map was built: 0.63 -> 0.82 ms. An inclusive prefix that comes out
		 * late makes every slice behind it walk back through hundreds of aggregates — the look-back is short only because prefixes appear at once.) */
		if (wave == 0) {
			const u64 excl = lookback64(status, c, (u64)tot_k, lane, err, KERR_WATCHDOG | KERR_AT_EXPAND);
			if (lane == 0) {
				*s_base = excl;
				if (c == bin_chunks - 1 && excl + tot_k != n_rec)
					atomicOr(err, KERR_NREC); /* CBinDesc n_rec must agree with the byte stream */
			}
		}
		if (tid == 0) {
			s_sk[tot_sk < MAX_SK ? tot_sk : MAX_SK] = tot_k & 0xFFFFu; /* sentinel */
			if (tot_k > 0xFFFFu)
				atomicOr(err, KERR_CORRUPT); /* more k-mers than the bytes of a slice can hold: length bytes that are not a record chain's */
		}
		__syncthreads();
		const u32 n_sk = tot_k > 0xFFFFu ? 0u : (tot_sk < MAX_SK ? tot_sk : MAX_SK);
		const u64 base = *s_base;
		/* what the k-mer loop needs of the 64-bit record numbers, as scalars: the first record's address and how many k-mers of this slice are inside the bin */
		u64 *const out_base = out + base * SIZE;
		u64 *const pair_row = pair_out ? pair_out + base : nullptr;
		const u32 pair_first = pair_out ? (u32)(pair_row - ge.pair_base) : 0u; /* the group holds fewer than 2^32 records (the host checks) */
		const u32 j_limit = base >= n_rec ? 0u : (n_rec - base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(n_rec - base));
		/* k-mer -> super-k-mer: the super-k-mers are numbered in k-mer order and number 0 starts at k-mer 0, so k-mer j belongs to number (super-k-mers that start
		 * at k-mers 1..j) = (bits in front of position j). A wave's 64 k-mers are one aligned row of the bit array: s_rowsk[row] + the bits of the row's two words
		 * below the lane (v_mbcnt). Round 4; before, a 16-bit index per k-mer was built per window of 4-8 K k-mers by clear + mark + two sweeps of a max-scan: five
		 * barriers per window, 0.17 of the kernel's 0.73 ms. */
		{
			constexpr int RPT = (int)(EXP_MAX_K / 64 / EXP_BLOCK); /* rows per thread */
			static_assert(RPT >= 1 && RPT * 64 * EXP_BLOCK == (int)EXP_MAX_K, "rows of the start bits per thread");
			const u32 n_rows = n_sk ? (tot_k + 63) >> 6 : 0;
			u32 cnt[RPT], mine = 0;
#pragma unroll
			for (int q = 0; q < RPT; ++q) {
				const u32 row = tid * RPT + q;
				cnt[q] = row < n_rows ? (u32)__popc(s_start[2 * row]) + (u32)__popc(s_start[2 * row + 1]) : 0u;
				mine += cnt[q];
			}
			u32 total;
			u32 run = block_excl_sum<EXP_BLOCK / 64, u32>(mine, s_tmp, total);
#pragma unroll
			for (int q = 0; q < RPT; ++q) {
				const u32 row = tid * RPT + q;
				if (row < n_rows)
					s_rowsk[row] = (unsigned short)run;
				run += cnt[q];
			}
		}
		__syncthreads();
		{
			const u32 wn = n_sk ? tot_k : 0u;
			for (u32 r = tid; r < wn; r += EXP_BLOCK) {
#if defined(EXP_CUT) && EXP_CUT == 1 /* tuning builds only: everything but the k-mer loop (output garbage) */
				break;
#endif
				const u32 j = r;
				const u32 m_lo = s_start[2 * (r >> 6)], m_hi = s_start[2 * (r >> 6) + 1]; /* the wave's row (r - lane is a multiple of 64) */
				const u32 si = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, (u32)s_rowsk[r >> 6]));
				if (j < j_limit) {
					u64 v[SIZE];
					const u32 sk = s_sk[si];
					const u32 off = j - (sk & 0xFFFFu);
					if constexpr (SIZE == 1) {
						/* k <= 32: the window of 2k bits ends in some aligned LDS dword; that dword and the two before it, byte-swapped, are a big-endian bit
						 * stream and each half of the k-mer is ONE funnel shift (v_alignbit) of two neighbours — no 64-bit shifts, no branch on "does the window
						 * start on a dword" (round 4; before: three dwords from the window's FIRST dword, shifted left as a 64-bit pair under a per-lane branch) */
						/* the window's LAST bit, counted from s_raw (s_b = s_raw + 16; a record's symbols start one byte behind its position; 4 symbols a byte) */
						const u32 last = 8u * (17u + (sk >> 16)) + 2u * off + 2 * k - 1;
						const u32 *we = reinterpret_cast<const u32 *>(s_raw + ((last >> 3) & ~3u)); /* the dword that holds it; we[-2] >= s_raw */
						const u32 rs = ~last & 31u;                                                 /* bits behind the window inside that dword */
						const u32 e0 = __builtin_bswap32(we[-2]), e1 = __builtin_bswap32(we[-1]), e2 = __builtin_bswap32(we[0]);
						const u32 lo = __builtin_amdgcn_alignbit(e1, e2, rs), hi = __builtin_amdgcn_alignbit(e0, e1, rs);
						u64 f = (((u64)hi << 32) | lo) & kmask;
						if (both_strands) {
							const u64 rc = shr64_uniform(~kmc_rev2(f), 64 - 2 * k); /* complement + reverse, realigned to the low 2k bits */
							f = rc < f ? rc : f;
						}
						v[0] = f;
					} else {
						/* wider records, the same way (round 4; kmc_extract_kmer walks the window byte by byte: ~150 VALU instructions per k-mer at k = 55,
						 * ~500 at k = 127): the window of 2k bits ends in some aligned LDS dword; that dword and the 2 SIZE before it, byte-swapped, are a big-endian
						 * bit stream, and limb i of the k-mer is ONE funnel shift of two neighbouring dwords. Every register index is static: the per-lane part is
						 * the address of the last dword and the shift. (A per-lane choice between two register indices — the first version — is turned into a
						 * dynamic index by the compiler and expanded into nine-way compare/select chains.) */
						constexpr int M = 2 * SIZE + 1;
						const u32 last = 8u * (17u + (sk >> 16)) + 2u * off + 2 * k - 1;        /* the window's last bit, counted from s_raw (see SIZE == 1) */
						const u32 *we = reinterpret_cast<const u32 *>(s_raw + ((last >> 3) & ~3u)); /* the dword that holds it */
						const u32 rs = ~last & 31u;                                                 /* bits behind the window inside that dword: 0..31 */
						u32 E[M]; /* E[M-1] = the window's last dword; dwords in front of the window (up to three, still inside s_raw, which starts 16 bytes before
						           * the slice) land in bits that kmc_mask_low clears */
#pragma unroll
						for (int t = 0; t < M; ++t)
							E[M - 1 - t] = __builtin_bswap32(we[-t]);
#pragma unroll
						for (int i = 0; i < 2 * SIZE; ++i) {
							const u32 limb = (u32)((((u64)E[M - 2 - i] << 32) | E[M - 1 - i]) >> rs);
							if (i & 1)
								v[i >> 1] |= (u64)limb << 32;
							else
								v[i >> 1] = limb;
						}
						kmc_mask_low<SIZE>(v, 2 * k);
						if (both_strands) {
							u64 rc[SIZE];
							kmc_revcomp<SIZE>(v, k, rc);
							if (kmc_less<SIZE>(rc, v)) {
#pragma unroll
								for (int w = 0; w < SIZE; ++w)
									v[w] = rc[w];
							}
						}
					}
#pragma unroll
					for (int w = 0; w < SIZE; ++w)
						if (((2 * k) >> 6) == (u32)w)
							v[w] |= tag; /* already shifted to its place inside the word that holds bit 2k (k = 32 SIZE: no spare bits, tag 0) */
					store_rec<SIZE>(out_base + (size_t)j * SIZE, v);
#ifndef EXP_NO_HIST /* tuning builds only (-DEXP_NO_HIST): what do the fused histograms cost? (the sort is garbage then) */
					if (FUSE_HIST) {
						if constexpr (SIZE == 1) { /* static byte positions below one uniform shift: a bit-field extract + an add per digit */
							const u64 t = shr64_uniform(v[0], 8 * pass_lo);
							const u32 tl = (u32)t, th = (u32)(t >> 32);
							if (n_pass == 4) { /* the hybrid sort's four top bytes (k = 13..32): no branch per digit */
#pragma unroll
								for (int b = 0; b < 4; ++b)
									atomicAdd(&s_h[b * 256 + ((tl >> (8 * b)) & 0xFFu)], 1u);
							} else {
#pragma unroll
								for (int b = 0; b < 8; ++b)
									if ((u32)b < n_pass)
										atomicAdd(&s_h[b * 256 + (((b < 4 ? tl : th) >> (8 * (b & 3))) & 0xFFu)], 1u);
							}
						} else {
							u32 top = 0; /* the digits of the HBM passes, lowest first: with four passes, the key's top four bytes */
							for (u32 b = 0; b < n_pass; ++b) {
								const u32 dg = kmc_get_byte<SIZE>(v, pass_lo + b);
								atomicAdd(&s_h[b * 256 + dg], 1u);
								top |= dg << ((8 * b) & 31);
							}
							if (pair_row)
								pair_row[j] = ((u64)top << 32) | (pair_first + j);
						}
					}
#endif
				}
			}
		}
	}
	if (FUSE_HIST) {
		__syncthreads();
		for (u32 i = threadIdx.x; i < n_pass * 256; i += EXP_BLOCK) {
			const u32 v = s_h[i];
			if (v)
				atomicAdd(&ghist[i], (u64)v);
		}
		/* the workgroup that finishes LAST turns the complete histograms into the digit bases of every pass (one launch and its
		 * gap less per bin than a separate scan kernel): its own atomics and everybody else's are performed before the
		 * respective done_ctr increment, so the agent-scope loads below see the final counts. Everything involved is a device-scope
		 * atomic, so waiting for this wave's outstanding memory operations is all the ordering needed; a __threadfence() here would
		 * also write back and invalidate the L2 (61 ms instead of 11 in the compaction, where every tile passes this point). */
		KMC_WAIT_VMEM();
		__syncthreads();
		if (threadIdx.x == 0)
			s_ticket[1] = atomicAdd(done_ctr, 1u);
		__syncthreads();
		if (s_ticket[1] == gridDim.x - 1) {
			u64 *s_scan = reinterpret_cast<u64 *>(s_raw + 16); /* the slice buffer is free now */
			for (u32 b = 0; b < n_pass; ++b) {
				const u64 v = threadIdx.x < 256 ? ld_agent(&ghist[b * 256 + threadIdx.x]) : 0ull;
				u64 total;
				const u64 ex = block_excl_sum<EXP_BLOCK / 64, u64>(v, s_scan, total);
				if (threadIdx.x < 256)
					digit_base[b * 256 + threadIdx.x] = ex;
			}
		}
	}
}
template <bool FUSE_HIST> constexpr size_t exp_lds_bytes(u32 n_pass, u32 k)
{
	return 16 + (size_t)EXP_CHUNK + EXP_TAIL + ((size_t)exp_max_sk(k) + 1 + 24 + 3) * 4 + (FUSE_HIST ? (size_t)n_pass * 1024 : 0) + EXP_MAX_K / 8 + EXP_MAX_K / 32 + 16;
}

/* ------------------------------------------------------------------------------------------------ histogram
 * ONE pass over the records builds the 256-bin histogram of EVERY digit position (n_pass <= 8*SIZE), in LDS
 * (u32 per workgroup), flushed with 64-bit global atomics. A wave whose lanes all hold the same digit value
 * (zero high bytes, poly-A bins) adds once instead of issuing a 64-way conflicting LDS atomic. */
template <int SIZE>
__global__ void __launch_bounds__(256) k_hist(const u64 *__restrict__ recs, u64 n, u32 n_pass, u64 *__restrict__ ghist, u32 pass_lo)
{
	KMC_DYN_LDS(u32, s_h); /* n_pass * 256 */
	for (u32 i = threadIdx.x; i < n_pass * 256; i += 256)
		s_h[i] = 0;
	__syncthreads();
	const u64 stride = (u64)gridDim.x * 256;
	const u32 lane = threadIdx.x & 63;
	/* the loop bound is wave-uniform (all 64 lanes take the same trips, the tail is a predicate): the cross-lane operations below then
	 * always see the whole wave */
	for (u64 i0 = (u64)blockIdx.x * 256 + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
		const u64 i = i0 + lane;
		const bool valid = i < n;
		u64 x[SIZE];
#pragma unroll
		for (int w = 0; w < SIZE; ++w)
			x[w] = 0;
		if (valid)
			load_rec<SIZE>(recs + i * SIZE, x);
		const u64 act = __ballot(valid); /* lane 0 is valid whenever any lane is */
		for (u32 b = 0; b < n_pass; ++b) {
			const u32 d = kmc_get_byte<SIZE>(x, pass_lo + b);
			const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
			if (__ballot(valid && d == d0) == act) {
				if (lane == 0)
					atomicAdd(&s_h[b * 256 + d0], (u32)__popcll(act));
			} else if (valid) {
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
 *  1. tile id from an atomic ticket (so every lower-numbered tile is already running: waiting on it cannot deadlock)
 *  2. wave w loads ITEMS x 64 consecutive records (512*SIZE B per wave-instruction), in index order
 *  3. counting: every record adds 1 to its wave's private LDS counter of its digit; digit d's tile count = sum over
 *     waves is published at once as AGGREGATE in status[tile][d] — before the ranking, so that by the time the
 *     successors look back (step 5) every aggregate they need has long been visible (a store needs ~1.5 us to be seen
 *     by another CU, a dependent status load ~1 us): published after the ranking, the walk kept running into tiles
 *     that had not published yet and took 10 of a tile's 19 us; now 5.6. The same pass over the counters turns them
 *     into "first LDS slot of (wave, digit)".
 *  4. ranking: for each of the ITEMS rounds the wave finds, with 8 ballots, the lanes holding the same digit
 *     ("match-any"); slot = the wave's running slot counter of that digit + number of lower peer lanes; the lowest
 *     peer advances the counter. Index order is preserved => the pass is STABLE. The slot is final (tile-relative).
 *  5. decoupled look-back: the digit owners (thread d < 256) walk back over earlier tiles' status words until they
 *     meet an inclusive PREFIX (~21 tiles back), then publish this tile's. The walk is bound by the BYTES of status
 *     rows it reads (device-coherent loads that miss every cache, 1 KB per tile and hop), not by round trips:
 *     reading more rows per round trip (K = 8..16), a workgroup-wide first round of 32-64 rows, or dedicated
 *     "propagator" workgroups that turn aggregates into prefixes for everybody were all measured slower or equal
 *     (the code is in the history at commit "count first and publish the tile aggregate ...").
 *  6. records are placed in LDS in digit order, then streamed out: consecutive threads write consecutive
 *     addresses inside each digit run (TILE/256 = 32 records = 256 B per run on uniform digits).
 * status word: [31:30] flag (0 empty, 1 aggregate, 2 inclusive prefix), [29:0] count — one relaxed agent-scope
 * 32-bit word that is both data and flag (no fences needed; cdna_hip_programming.md Guideline 16, form R2). */
constexpr u32 ST_AGG = 1u << 30, ST_PREFIX = 2u << 30, ST_MASK = (1u << 30) - 1;

#ifndef RS_MIN_WAVES
#define RS_MIN_WAVES 8 /* waves per SIMD the register allocator must leave room for (2 workgroups of 1024 per CU) */
#endif
#ifndef RS_LOOKBACK_K
#define RS_LOOKBACK_K 4 /* status words per look-back round trip */
#endif
#ifndef RS_TILE_FROM_BLOCKIDX
#define RS_TILE_FROM_BLOCKIDX 0 /* 1: tile = blockIdx.x instead of an atomic ticket (saves the ticket's round trip in front of every tile's loads).
                                 * The look-back then relies on workgroups being STARTED in blockIdx order (lower ids are running or done when a
                                 * higher one spins) — what the dispatcher does, but not an architectural promise; the watchdog turns a violation
                                 * into KMC_HIP_EINTERNAL instead of a hang. Tuning option. */
#endif
#ifndef RS_TPB
#define RS_TPB 1 /* tiles per ticket. Keep 1: a workgroup that owns consecutive tiles publishes the later ones late and every
                   * successor stalls on them (measured 150x slower at 2); larger tiles are the way to fewer tickets */
#endif

/* radix digit of pass `byte_idx` (= kmc_get_byte) without a dynamically indexed register array (which the compiler
 * parks in scratch once two copies of the tile body share the function's promote-alloca budget): byte_idx is
 * wave-uniform, so the 32-bit half that holds the digit is picked by a chain of v_cndmask on scalar conditions. */
template <int SIZE> __device__ __forceinline__ u32 rs_digit(const u64 (&x)[SIZE], u32 byte_idx)
{
	const u32 hw = byte_idx >> 2;
	u32 half = (u32)x[0];
#pragma unroll
	for (int i = 1; i < 2 * SIZE; ++i)
		half = (hw == (u32)i) ? (u32)(x[i >> 1] >> ((i & 1) * 32)) : half;
	return __builtin_amdgcn_ubfe(half, (byte_idx & 3) * 8, 8);
}

/* one tile of a pass (steps 2-6 above); the caller has the tile's number and runs a barrier behind it before the LDS is used again */
template <int SIZE>
__device__ __forceinline__ void onesweep_tile(const u64 *__restrict__ in, u64 *__restrict__ out, const u32 n, const u32 byte_idx, const u64 *__restrict__ digit_base_in,
                                              u64 *__restrict__ digit_base_next, u32 *status, const u32 tile, const u32 num_tiles, u32 *err, u64 *s_goff, u64 *s_keys, u32 *s_whist,
                                              u32 *s_wsum)
{
	constexpr int ITEMS = RsCfg<SIZE>::ITEMS;
	constexpr int TILE = RsCfg<SIZE>::TILE;
	{
		/* lane-constant addresses must be recomputed per tile: hoisted out of this loop they cost ~60 VGPRs and spill */
		u32 tid = threadIdx.x;
		KMC_LAUNDER(tid);
		const u32 lane = tid & 63, wave = tid >> 6;
#pragma unroll
		for (int i = 0; i < 4; ++i)
			s_whist[wave * 256 + i * 64 + lane] = 0;
		KMC_WAVE_LOCKSTEP(); /* the wave's own counters are zero before any of its lanes counts into them */
		const u64 tile_base = (u64)tile * TILE;
		/* (32-bit arithmetic: tile < num_tiles, so tile * TILE < n. The 64-bit form `(n - tile_base) < TILE ? ... : TILE` compiled to a v_cmp_lt_u64 into vcc and an
		 * s_cselect_b32 on whatever SCC held — the "is this the last tile" compare of a few lines further down in k_onesweep, by luck the same condition there; in
		 * k_onesweep_dyn the borrow of the subtraction: a partial tile taken for a full one, records dropped. Round 6, found on the device, not under the emulation.) */
		const u32 tile_rest = n - tile * (u32)TILE;
		const u32 tile_n = tile_rest < (u32)TILE ? tile_rest : (u32)TILE;
		TRACE_STAMP(0, tile, 0);

		/* the body is instantiated twice: all tiles but the last of a portion are full, and for them every bounds
		 * check (a v_cmp + exec juggling per record in four places) folds away */
		auto tile_body = [&](auto full_tag) __attribute__((always_inline)) {
		constexpr bool FULL = decltype(full_tag)::value;
		u64 key[ITEMS][SIZE];
		u32 rank2[(ITEMS + 1) / 2]; /* tile-relative slots (< TILE <= 16384), two per register */
		const u32 wbase = wave * (ITEMS * 64) + lane;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 idx = wbase + r * 64;
			if (FULL || idx < tile_n)
				load_rec<SIZE>(in + (tile_base + idx) * SIZE, key[r]);
			else {
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					key[r][w] = 0;
			}
		}
#ifdef KMC_TRACE
		KMC_WAIT_VMEM(); /* tuning build only: separate the load latency from the rest */
#endif
		TRACE_STAMP(0, tile, 1);

		/* ---- step 3: count. The digits are extracted once and kept, 4 per register. */
		u32 dpack[(ITEMS + 3) / 4] = {};
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const u32 d = rs_digit<SIZE>(key[r], byte_idx);
			dpack[r >> 2] |= d << ((r & 3) * 8);
			if (FULL || (wbase + r * 64) < tile_n)
				(void)__hip_atomic_fetch_add(&s_whist[wave * 256 + d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		__syncthreads();
		TRACE_STAMP(0, tile, 2);

		/* thread `tid` (< 256) owns digit `tid` from here to the end of step 5 */
		u32 cnt = 0, inc = 0;
		if (tid < 256) {
#pragma unroll
			for (int w = 0; w < RS_WAVES; ++w)
				cnt += s_whist[w * 256 + tid];
			st_agent(&status[(u64)tile * 256 + tid], (tile == 0 ? ST_PREFIX : ST_AGG) | cnt);
			inc = wave_incl_sum<u32>(cnt, lane);
			if (lane == 63)
				s_wsum[wave] = inc;
		}
		TRACE_VALUE(2, tile, 7, wall_clock64()); /* when the aggregate went out, on the clock all CUs share */
		__syncthreads();
		u32 doff = 0;
		if (tid < 256) {
			doff = inc - cnt; /* first LDS slot of digit `tid` */
#pragma unroll
			for (int w = 0; w < 4; ++w)
				if (w < (int)wave)
					doff += s_wsum[w];
			u32 run = doff;
#pragma unroll
			for (int w = 0; w < RS_WAVES; ++w) { /* count -> first slot of (wave, digit) */
				const u32 t = s_whist[w * 256 + tid];
				s_whist[w * 256 + tid] = run;
				run += t;
			}
		}
		__syncthreads();
		TRACE_STAMP(0, tile, 3);

		/* ---- step 4: ranking. The slot counter read in round r is consumed one round later, the lowest peer advances
		 * it with a NON-returning LDS add: LDS executes one wave's operations in order, so the read of round r+1 sees
		 * the add of round r and nothing waits on LDS inside a round. */
		u32 below_prev = 0, base_prev = 0;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r)
			if ((r & 1) == 0)
				rank2[r >> 1] = 0;
#pragma unroll
		for (int r = 0; r < ITEMS; ++r) {
			const bool valid = FULL || (wbase + r * 64) < tile_n;
			const u32 d = (dpack[r >> 2] >> ((r & 3) * 8)) & 0xFF;
			const u64 vm = __ballot(valid);
			u32 lo = (u32)vm, hi = (u32)(vm >> 32);
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				/* 4 VALU per bit: v_bfe_i32, v_cmp (= the ballot), 2 x v_bitop3 [src1 & ~(src0 ^ src2)]. (__ballot(pred)
				 * costs two more: it materialises the predicate as 0/1 and compares again.) */
				const u32 sb = (u32)__builtin_amdgcn_sbfe((int)d, b, 1); /* 0 or ~0 */
				const u64 m = __builtin_amdgcn_uicmp(sb, 0u, 33 /* ICMP_NE */);
				lo = __builtin_amdgcn_bitop3_b32(sb, lo, (u32)m, 0x84);
				hi = __builtin_amdgcn_bitop3_b32(sb, hi, (u32)(m >> 32), 0x84);
			}
			const u32 below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
			if (r > 0) {
				const u32 full = (!FULL && (wbase + (r - 1) * 64) >= tile_n) ? 0xFFFFu : (base_prev + below_prev);
				rank2[(r - 1) >> 1] |= full << (((r - 1) & 1) * 16);
			}
			u32 *ctr = &s_whist[wave * 256 + d];
			base_prev = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* other lanes add to it */
			KMC_WAVE_LOCKSTEP(); /* every lane has read the counter before any lane adds to it: free on the hardware (a wave executes one LDS
			                      * instruction for all its lanes at once), a real rendezvous under tests/hipemu */
			if (valid && below == 0)
				(void)__hip_atomic_fetch_add(ctr, (u32)(__popc(lo) + __popc(hi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			below_prev = below;
			__builtin_amdgcn_sched_barrier(0); /* keep rounds in order: interleaving them only inflates SGPR/VGPR live ranges */
		}
		{
			const u32 full = (!FULL && (wbase + (ITEMS - 1) * 64) >= tile_n) ? 0xFFFFu : (base_prev + below_prev);
			rank2[(ITEMS - 1) >> 1] |= full << (((ITEMS - 1) & 1) * 16);
		}
		TRACE_STAMP(0, tile, 4);

		/* ---- step 5: the exclusive prefix of the tile, digit by digit */
		if (tid < 256) {
			u32 excl = 0;
			if (tile > 0) {
				int t = (int)tile - 1;
				LbWatch watch;
				[[maybe_unused]] u32 rounds = 0; /* read by the trace build only */
				TRACE_STAMP(2, tile, 1);
				/* decoupled look-back: walk back over earlier tiles, RS_LOOKBACK_K status words per round trip, until an
				 * inclusive prefix is met; then publish this tile's */
				bool done = false;
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
					if (!done && used < RS_LOOKBACK_K) { /* ran into a tile that has not published yet: tile t, now */
						if (lb_blocked(watch, err))
							break;
						__builtin_amdgcn_s_sleep(1);
					}
				}
				if (!done)
					lb_gave_up(watch, err, KERR_WATCHDOG | KERR_AT_SCATTER, tid, tile, t, num_tiles);
				st_agent(&status[(u64)tile * 256 + tid], ST_PREFIX | (excl + cnt));
				TRACE_STAMP(2, tile, 2);
				TRACE_VALUE(2, tile, 3, rounds);
				TRACE_VALUE(2, tile, 4, watch.spins);
				TRACE_VALUE(2, tile, 5, (u64)((int)tile - 1 - t));
			}
			const u64 gbase = digit_base_in[tid] + excl;
			s_goff[tid] = gbase - doff;
			if (tile == num_tiles - 1)
				digit_base_next[tid] = gbase + cnt; /* where the next portion continues this digit */
		}
		TRACE_STAMP(0, tile, 5);

		/* ---- step 6. The tile goes through LDS in RS_STAGES slices of STAGE_N slots (digit order): with 2 slices the
		 * staging area is 32 KB instead of 64 KB (the kernel is bound by how many tiles are in flight). The first
		 * barrier below is also the one that makes s_goff visible. */
		constexpr int STAGES = RsCfg<SIZE>::STAGES;
		constexpr int STAGE_N = TILE / STAGES;
#pragma unroll 1
		for (int h = 0; h < STAGES; ++h) {
			const u32 lo = h * STAGE_N;
			if (lo >= tile_n)
				break;
#pragma unroll
			for (int r = 0; r < ITEMS; ++r) {
				const u32 rel = ((rank2[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu) - lo; /* invalid records carry 0xFFFF: never inside a slice */
				if (rel < (u32)STAGE_N) {
#pragma unroll
					for (int w = 0; w < SIZE; ++w)
						s_keys[w * STAGE_N + rel] = key[r][w];
				}
			}
			__syncthreads();
			if (h == 0)
				TRACE_STAMP(0, tile, 6);
#pragma unroll
			for (int i = 0; i < ITEMS / STAGES; ++i) {
				const u32 rel = i * RS_BLOCK + tid;
				const u32 slot = lo + rel;
				if (FULL || slot < tile_n) {
					u64 x[SIZE];
#pragma unroll
					for (int w = 0; w < SIZE; ++w)
						x[w] = s_keys[w * STAGE_N + rel];
					const u32 d = rs_digit<SIZE>(x, byte_idx);
					store_rec<SIZE>(out + (s_goff[d] + slot) * SIZE, x);
				}
			}
			if (h + 1 < STAGES)
				__syncthreads();
		}
		TRACE_STAMP(0, tile, 7);
		};
		if (tile_n == (u32)TILE)
			tile_body(std::true_type{});
		else
			tile_body(std::false_type{});
	}
}

template <int SIZE>
__global__ void __launch_bounds__(RS_BLOCK, (SIZE <= 4 ? RS_MIN_WAVES : RS_MIN_WAVES / 2)) k_onesweep(const u64 *__restrict__ in, u64 *__restrict__ out, u32 n, u32 byte_idx,
                                                        const u64 *__restrict__ digit_base_in, u64 *__restrict__ digit_base_next,
                                                        u32 *status, u32 *tile_counter, u32 num_tiles, u32 *err)
{
	constexpr int TILE = RsCfg<SIZE>::TILE;
	KMC_DYN_LDS(unsigned char, s_raw);
	u64 *s_goff = reinterpret_cast<u64 *>(s_raw);              /* [256]  global index of LDS slot 0 as seen by digit d */
	u64 *s_keys = s_goff + 256;                                /* [SIZE*TILE/STAGES] word-major staging area            */
	u32 *s_whist = reinterpret_cast<u32 *>(s_keys + SIZE * TILE / RsCfg<SIZE>::STAGES); /* [RS_WAVES*256] per-wave digit counters -> slots */
	u32 *s_wsum = s_whist + RS_WAVES * 256;                    /* [4]                                                    */
	u32 *s_tile = s_wsum + 4;                                  /* [1]                                                    */

#if RS_TILE_FROM_BLOCKIDX
	const u32 ticket = blockIdx.x;
	(void)tile_counter;
	(void)s_tile;
#else
	if (threadIdx.x == 0)
		*s_tile = atomicAdd(tile_counter, 1u);
	__syncthreads();
	const u32 ticket = (u32)__builtin_amdgcn_readfirstlane((int)*s_tile); /* scalar: tile-level addresses live in SGPRs */
#endif

#pragma unroll 1
	for (int it = 0; it < RS_TPB; ++it) {
		const u32 tile = ticket * RS_TPB + it;
		if (tile >= num_tiles)
			break;
		onesweep_tile<SIZE>(in, out, n, byte_idx, digit_base_in, digit_base_next, status, tile, num_tiles, err, s_goff, s_keys, s_whist, s_wsum);
		__syncthreads(); /* LDS is reused by the next tile of this ticket */
	}
}

/* The same pass over an array whose LENGTH is only known on the device (arena_sort.hip.h: the records of a group's repeat-rich buckets): dyn[0] = records, dyn[1] =
 * passes to run (launch `pass` >= dyn[1]: nothing to do). A fixed grid of persistent workgroups takes tiles by ticket until none is left — a workgroup that
 * draws ticket t is running, so is (or was) every workgroup that drew a lower one: the look-back cannot wait for a tile nobody has. */
template <int SIZE>
__global__ void __launch_bounds__(RS_BLOCK, (SIZE <= 4 ? RS_MIN_WAVES : RS_MIN_WAVES / 2)) k_onesweep_dyn(const u64 *__restrict__ in, u64 *__restrict__ out, const u32 *__restrict__ dyn, u32 pass,
                                                        const u64 *__restrict__ digit_base_in, u64 *__restrict__ digit_base_next, u32 *status, u32 *tile_counter, u32 *err)
{
	constexpr int TILE = RsCfg<SIZE>::TILE;
	KMC_DYN_LDS(unsigned char, s_raw);
	u64 *s_goff = reinterpret_cast<u64 *>(s_raw);
	u64 *s_keys = s_goff + 256;
	u32 *s_whist = reinterpret_cast<u32 *>(s_keys + SIZE * TILE / RsCfg<SIZE>::STAGES);
	u32 *s_wsum = s_whist + RS_WAVES * 256;
	u32 *s_tile = s_wsum + 4;
	const u32 n = dyn[0];
	if (n == 0 || pass >= dyn[1])
		return;
	const u32 num_tiles = (n + (u32)TILE - 1) / (u32)TILE;
#pragma unroll 1
	while (true) {
		if (threadIdx.x == 0)
			*s_tile = atomicAdd(tile_counter, 1u);
		__syncthreads();
		const u32 tile = (u32)__builtin_amdgcn_readfirstlane((int)*s_tile);
		if (tile >= num_tiles)
			break;
		onesweep_tile<SIZE>(in, out, n, pass, digit_base_in, digit_base_next, status, tile, num_tiles, err, s_goff, s_keys, s_whist, s_wsum);
		__syncthreads(); /* LDS (and the ticket word) are reused by the next tile */
	}
}

template <int SIZE> constexpr size_t rs_lds_bytes()
{
	return 256 * 8 + (size_t)SIZE * RsCfg<SIZE>::TILE * 8 / RsCfg<SIZE>::STAGES + RS_WAVES * 256 * 4 + 4 * 4 + 16;
}

/* ------------------------------------------------------------------------------------------------ compaction
 * ONE coalesced read of the sorted records. Wave w of a tile owns ROWS consecutive rows of 64 records; lane l of row r holds
 * record cbase + 64 r + l, so every load instruction of a wave reads 512*SIZE contiguous bytes.
 *   tails   record i ends a run iff S[i] != S[i+1] (neighbour lane: one shuffle; the row's last lane takes the next row's first
 *           record). One ballot per row gives the row's 64 tail bits as a scalar mask.
 *   counts  count of the run ending at i = i - (position of the previous tail). Inside a row that is bit arithmetic on the mask
 *           (clz of the bits below the lane); across rows a scalar carry; across waves one LDS word per wave ("my last tail");
 *           across tiles the 64 records below the tile are inspected the same way, and only a run longer than that needs the
 *           64-ary search below. No workgroup-wide scan, one barrier.
 *   classes cutoffs and clamp per tail lane (kb_sorter.h:1174-1192: compare BEFORE clamping; the count is uint32), three ballots per
 *           row give counted / below / above as masks: the tallies are popcounts, a counted record's rank in the tile is
 *           wave offset + row offset + mbcnt.
 *   output  ascending k-mer = tile order: the tile's offset among counted k-mers comes from a 64-bit decoupled look-back (one word
 *           per tile) run by wave 0 while the other waves already place their records in the LDS window (tile-relative ranks);
 *           the window is streamed out as aligned dwords. Records and LUT: kb_sorter.h:1196-1203. LUT updates are aggregated per
 *           tile: the prefixes of the tile's counted k-mers are a sorted list in LDS and each run of equal prefixes costs two
 *           global atomics (+end, -begin).
 * (Round 1-2a kept 16 consecutive records per THREAD: 16 load instructions of 64 scattered 8-byte pieces each, a workgroup-wide
 * max-scan for the run heads and one for the ranks: 26.7 us per 8192-record tile, 10.1 of them in the loads — profiles/r02/trace_report_496M_bin.txt.) */

/* smallest i in [0, hi] with S[i] == v, given S[hi] == v and S sorted; executed by one full wave (64-ary search) */
template <int SIZE> __device__ __forceinline__ u64 run_start_search(const u64 *__restrict__ S, u64 hi, u32 lane, const u64 (&v)[SIZE])
{
	u64 lo = 0;
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

__device__ __forceinline__ u64 wave_first(u64 v) /* lane 0's value in every lane (two v_readfirstlane) */
{
	const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
	return ((u64)hi << 32) | lo;
}

constexpr int CP_STAGE = 16384; /* bytes of output assembled in LDS per window */
constexpr int CP_SHARDS = 32; /* tally shards: same-address device atomics serialise at ~11 ns each */
#ifndef CP_FOLD_CHUNK
#define CP_FOLD_CHUNK 4096 /* tile counts scanned per round of k_compact_fold (two-phase output) */
#endif
#ifndef CP_TILE_FROM_BLOCKIDX
#define CP_TILE_FROM_BLOCKIDX 0
#endif
#ifndef CP_TPB
#define CP_TPB 1 /* tiles per ticket; must stay 1 (see RS_TPB) */
#endif
static_assert(CP_TPB == 1, "one compaction tile per workgroup");

template <int SIZE>
__global__ void __launch_bounds__(CP_BLOCK, CpCfg<SIZE>::MIN_WAVES) k_compact(const GrpCompact gc, DevParams P, u32 lut_shards, u64 lut_stride, u32 *tile_counter,
                                                                               u32 *err, u32 lut_mask /* 4^p - 1: drops a group tag above the k-mer */,
                                                                               u32 two_phase)
{
	/* two_phase: a tile does not wait for its offset in the output (the look-back is 6.4 of a tile's 16 us) — it leaves its records at the
	 * start of its own span of the record array the sort no longer needs and its count in status[tile]; k_compact_fold turns the counts into
	 * offsets, k_compact_gather moves the records (5 % of the k-mers at cutoff_min 2) to their place. The host picks the mode: the span
	 * (TILE records of 8 SIZE bytes) must hold the tile's counted records whatever the data (kmc_hip.hip compact_group). */
	constexpr int ROWS = CpCfg<SIZE>::ITEMS;
	constexpr int TILE = CpCfg<SIZE>::TILE;
	constexpr int NW = CP_BLOCK / 64;
	__shared__ u32 s_wlast[NW]; /* tile-relative position of the last tail inside wave w's rows, 0xFFFFFFFF if there is none */
	__shared__ u32 s_carry_in;  /* position of the last tail below the tile, relative to the tile (mod 2^32; -1 = none before record 0) */
	__shared__ u32 s_wcnt[NW];        /* counted k-mers of wave w */
	__shared__ u32 s_wtal[NW][3];     /* unique / below min / above max of wave w */
	__shared__ u32 s_pref[TILE];
	__shared__ __attribute__((aligned(16))) uint8_t s_stage[CP_STAGE];
	__shared__ u32 s_tile;
	__shared__ u64 s_tile_off;

#if CP_TILE_FROM_BLOCKIDX
	const u32 gtile = blockIdx.x; /* see RS_TILE_FROM_BLOCKIDX */
	(void)tile_counter;
#else
	if (threadIdx.x == 0)
		s_tile = atomicAdd(tile_counter, 1u);
	__syncthreads();
	const u32 gtile = (u32)__builtin_amdgcn_readfirstlane((int)s_tile);
#endif
	/* the tiles of all bins of the group form one ticket space; everything below is about ONE bin */
	const u32 total_tiles = gc.tile_prefix[gc.g];
	const u32 bin = gtile < total_tiles ? (u32)__builtin_amdgcn_readfirstlane((int)grp_find(gc.tile_prefix, gc.g, gtile)) : 0;
	const u32 tile = gtile - gc.tile_prefix[bin];
	const u32 num_tiles = gtile < total_tiles ? gc.tile_prefix[bin + 1] - gc.tile_prefix[bin] : 0; /* 0: no tile for this workgroup */
	const u64 *__restrict__ S = gc.S[bin];
	const u64 n = gc.n[bin];
	uint8_t *__restrict__ out = gc.out[bin];
	const u64 out_capacity = gc.out_capacity[bin];
	u64 *__restrict__ lut_base = gc.lut_base[bin];
	u64 *stat_shards = gc.tally[bin];
	u64 *out_bytes = gc.out_bytes[bin];
	u64 *status = gc.status[bin];
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const bool use_lut = P.lut_prefix_len != 0 && !P.kff && !P.without_output;

	if (tile < num_tiles) {
		const u32 tid = threadIdx.x;
		const u32 lane = tid & 63;
		const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6)); /* scalar: everything derived from it stays on the scalar unit */
		const u64 lane_lt = (1ull << lane) - 1;
		TRACE_STAMP(1, tile, 0);
		TRACE_STAMP(1, tile, 1);
		/* Positions are kept TILE-RELATIVE in 32 bits (the kernel is bound by its VALU instructions: 2027 per wave and tile in the first
		 * version of this design, profiles/r02/pmc_sq_counters_quarter.txt): a count is a difference of two positions modulo 2^32, which is
		 * exactly the reference's uint32 counter, also for a previous tail far below the tile. */
		const u64 base = (u64)tile * TILE;
		const u32 n_rel = (n - base) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(n - base); /* records from the tile's first one to the end of the bin */
		const u32 crel = wave * (ROWS * 64);                                             /* the wave's first record */
		const u64 *Sw = S + (base + crel) * SIZE;

		/* ---- loads: the wave's rows, the record after them, and (last wave) the 64 records below the tile */
		u64 key[ROWS][SIZE];
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			if (crel + r * 64 + lane < n_rel)
				load_rec<SIZE>(Sw + (size_t)(r * 64 + lane) * SIZE, key[r]);
			else {
#pragma unroll
				for (int w = 0; w < SIZE; ++w)
					key[r][w] = 0;
			}
		}
		u64 after[SIZE];
		if (crel + ROWS * 64 < n_rel) /* the same address in every lane */
			load_rec<SIZE>(Sw + (size_t)(ROWS * 64) * SIZE, after);
		else {
#pragma unroll
			for (int w = 0; w < SIZE; ++w)
				after[w] = 0;
		}
		u64 b0[SIZE], b1[SIZE];
		bool have_b = false;
		if (wave == NW - 1 && base + lane >= 64) { /* j = base - 64 + lane >= 0; j + 1 <= base < n */
			load_rec<SIZE>(S + (base - 64 + lane) * SIZE, b0);
			load_rec<SIZE>(S + (base - 63 + lane) * SIZE, b1);
			have_b = true;
		}

		/* ---- tails: record i ends a run iff it differs from record i+1, or is the last one */
		u32 tail_bits = 0;       /* bit r: this lane's record of row r ends a run */
		u32 wlast = 0xFFFFFFFFu; /* tile-relative position of the wave's last tail */
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			bool differs = false;
#pragma unroll
			for (int w = 0; w < SIZE; ++w) {
				u64 nx = __shfl_down(key[r][w], 1);
				const u64 first_next = (r + 1 < ROWS) ? wave_first(key[r + 1 < ROWS ? r + 1 : r][w]) : after[w];
				if (lane == 63)
					nx = first_next;
				differs = differs || (nx != key[r][w]);
			}
			const u32 pos = crel + r * 64 + lane;
			const bool is_tail = pos < n_rel && (differs || pos + 1 == n_rel);
			const u64 m = __ballot(is_tail);
			if (is_tail)
				tail_bits |= 1u << r;
			if (m)
				wlast = crel + r * 64 + 63 - (u32)__clzll((long long)m);
			__builtin_amdgcn_sched_barrier(0); /* rows in order: interleaved, their scalar masks and first-lane values overflow the SGPRs */
		}
		if (lane == 0)
			s_wlast[wave] = wlast;
		if (wave == NW - 1) {
			u32 carry = 0xFFFFFFFFu; /* -1: no tail before record 0 */
			if (base > 0) {
				const u64 bm = __ballot(have_b && !kmc_equal<SIZE>(b0, b1)); /* tail at j iff S[j] != S[j+1] */
				if (bm)
					carry = (u32)(63 - __clzll((long long)bm)) - 64u;
				else if (base > 64) {
					/* S[base-64 .. base] are all equal: the run that crosses into the tile started further down */
					u64 v[SIZE];
					load_rec<SIZE>(S + (base - 64) * SIZE, v);
					carry = (u32)(run_start_search<SIZE>(S, base - 64, lane, v) - 1 - base);
				}
			}
			if (lane == 0)
				s_carry_in = carry;
		}
		TRACE_STAMP(1, tile, 2);
		__syncthreads();
		TRACE_STAMP(1, tile, 3);

		/* ---- counts, classes and ranks */
		u32 carry = s_carry_in;
#pragma unroll
		for (int w = 0; w < NW; ++w) {
			const u32 x = s_wlast[w];
			if (w < (int)wave && x != 0xFFFFFFFFu)
				carry = x;
		}
		carry = (u32)__builtin_amdgcn_readfirstlane((int)carry);
		TRACE_STAMP(1, tile, 4);
		u32 cnt[ROWS];
		u32 rank2[(ROWS + 1) / 2]; /* wave-relative rank among counted k-mers, 16 bits each; 0xFFFF = not counted */
		u32 nu = 0, nb = 0, na = 0, nc = 0; /* wave totals (scalar) */
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
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
			s_wtal[wave][0] = nu;
			s_wtal[wave][1] = nb;
			s_wtal[wave][2] = na;
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
		TRACE_STAMP(1, tile, 5);
		if (wave == 0) {
			/* tile offset among counted k-mers: 64-bit decoupled look-back, one word per tile, inspected 64 tiles
			 * at a time by the lanes of wave 0 (a one-word-per-hop walk costs ~1 us per hop) */
			u64 excl = 0;
			if (two_phase) {
				if (lane == 0)
					status[tile] = tile_counted;
			} else
				excl = lookback64(status, tile, (u64)tile_counted, lane, err, KERR_WATCHDOG | KERR_AT_COMPACT);
			if (lane == 0) {
				s_tile_off = excl;
				if (!two_phase && tile == num_tiles - 1)
					*out_bytes = P.without_output ? 0 : (excl + tile_counted) * (u64)rec_bytes;
				/* the tile's tallies, sharded */
				u32 tu = 0, tb = 0, ta = 0;
#pragma unroll
				for (int w = 0; w < NW; ++w) {
					tu += s_wtal[w][0];
					tb += s_wtal[w][1];
					ta += s_wtal[w][2];
				}
				u64 *sh = stat_shards + (size_t)(tile % CP_SHARDS) * 4;
				if (tu)
					atomicAdd(&sh[0], (u64)tu);
				if (tb)
					atomicAdd(&sh[1], (u64)tb);
				if (ta)
					atomicAdd(&sh[2], (u64)ta);
			}
		}
		TRACE_STAMP(1, tile, 6);

		/* ---- emit. Records are assembled in an LDS window at their tile-relative rank (known without the look-back) and
		 * streamed out as aligned dwords once the tile's offset is known: per-lane byte stores straight to HBM were 59 % of
		 * the first version's time. */
		if (!P.without_output && tile_counted) {
			const u32 tile_bytes = tile_counted * rec_bytes;
			const u32 pshift = 2 * (P.k - P.lut_prefix_len);
			uint8_t *const dst = two_phase ? gc.scratch[bin] + (u64)tile * ((u64)TILE * SIZE * 8) : out; /* where this tile's bytes go */
			if (rec_bytes <= 8) {
				/* fast path (k <= ~36): a record is one 64-bit value in output byte order; records go to an LDS window,
				 * then every thread composes aligned output dwords from it */
				u64 *s_rec = reinterpret_cast<u64 *>(s_stage);
				constexpr u32 WREC = CP_STAGE / 8;
				/* byte offset -> record: x / rec_bytes as a multiply and a shift (a 32-bit division is ~40 VALU instructions and the
				 * copy-out needs five); exact for x < 2^14 = CP_STAGE and rec_bytes <= 8: the error x e / (d 2^17) < 1/8 <= 1/d */
				static_assert(CP_STAGE <= 16384, "div_rb is exact below 2^14");
				const u32 inv17 = (131072u + rec_bytes - 1) / rec_bytes;
				auto div_rb = [&](u32 x) { return (x * inv17) >> 17; };
				for (u32 r0 = 0; r0 < tile_counted; r0 += WREC) {
					const u32 r1 = (tile_counted - r0) < WREC ? tile_counted : r0 + WREC;
#pragma unroll
					for (int r = 0; r < ROWS; ++r) {
						const u32 rk16 = (rank2[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu;
						if (rk16 != 0xFFFFu) {
							const u32 rank = wave_off + rk16;
							if (use_lut && r0 == 0)
								s_pref[rank] = (u32)kmc_remove_suffix<SIZE>(key[r], pshift) & lut_mask;
							if (rank >= r0 && rank < r1) {
								/* without a LUT prefix (KFF) the suffix bytes reach up to the top of the k-mer: a group tag above bit 2k must not get into them */
								const u64 k0 = (2 * P.k < 64) ? (key[r][0] & ((1ull << (2 * P.k)) - 1)) : key[r][0];
								u64 rv = P.sbytes ? __builtin_bswap64(k0 << (8 * (8 - P.sbytes))) : 0ull;
								if (P.cbytes) {
									const u32 cv = P.kff ? (__builtin_bswap32(cnt[r]) >> (8 * (4 - P.cbytes))) : cnt[r];
									rv |= (u64)cv << (8 * P.sbytes);
								}
								s_rec[rank - r0] = rv;
							}
						}
					}
					__syncthreads(); /* the window is complete; in the first round this also publishes s_tile_off */
					const u64 gbyte0 = two_phase ? 0 : s_tile_off * rec_bytes; /* byte offset of this tile's first record in dst */
					const bool fits = two_phase || gbyte0 + tile_bytes <= out_capacity; /* uniform over the workgroup */
					if (!fits && tid == 0)
						atomicOr(err, KERR_CAPACITY);
					if (fits) {
						const u64 g0 = gbyte0 + (u64)r0 * rec_bytes;
						const u32 len = (r1 - r0) * rec_bytes;
						u32 head = (u32)((4 - ((uintptr_t)(dst + g0) & 3)) & 3);
						if (head > len)
							head = len;
						const u32 ndw = (len - head) >> 2;
						const u32 tail0 = head + (ndw << 2);
						if (tid < head) /* head <= 3 < rec_bytes unless records are shorter than that: tid / rec_bytes via div_rb */
							dst[g0 + tid] = (uint8_t)(s_rec[div_rb(tid)] >> (8 * (tid - div_rb(tid) * rec_bytes)));
						if (tid >= 32 && tid - 32 < len - tail0) {
							const u32 bi = tail0 + (tid - 32), br = div_rb(bi);
							dst[g0 + bi] = (uint8_t)(s_rec[br] >> (8 * (bi - br * rec_bytes)));
						}
						u32 *gd = reinterpret_cast<u32 *>(dst + g0 + head);
						for (u32 w = tid; w < ndw; w += CP_BLOCK) {
							const u32 i0 = head + (w << 2);
							u32 ri = div_rb(i0), q = i0 - ri * rec_bytes;
							u64 cur = s_rec[ri];
							u32 word = 0;
#pragma unroll
							for (int t = 0; t < 4; ++t) {
								word |= ((u32)(cur >> (8 * q)) & 0xFFu) << (8 * t);
								if (++q == rec_bytes) {
									q = 0;
									++ri;
									cur = s_rec[ri < WREC ? ri : WREC - 1];
								}
							}
							gd[w] = word;
						}
					}
					__syncthreads();
				}
			} else {
				for (u32 c0 = 0; c0 < tile_bytes; c0 += CP_STAGE) {
					const u32 c1 = (tile_bytes - c0) < (u32)CP_STAGE ? tile_bytes : c0 + CP_STAGE;
#pragma unroll
					for (int r = 0; r < ROWS; ++r) {
						const u32 rk16 = (rank2[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu;
						if (rk16 != 0xFFFFu) {
							const u32 rank = wave_off + rk16;
							if (use_lut && c0 == 0)
								s_pref[rank] = (u32)kmc_remove_suffix<SIZE>(key[r], pshift) & lut_mask;
							const u32 bb = rank * rec_bytes;
							if (bb < c1 && bb + rec_bytes > c0) {
								for (u32 q = 0; q < rec_bytes; ++q) {
									const u32 bpos = bb + q;
									if (bpos >= c0 && bpos < c1) {
										u32 val;
										if (q < P.sbytes)
											{
												/* (a group tag above bit 2k must not get into the top suffix byte: see the fast path) */
												const u32 pb = P.sbytes - 1 - q, top = 2 * P.k;
												val = kmc_get_byte<SIZE>(key[r], pb);
												if (8 * pb + 8 > top)
													val = top > 8 * pb ? (val & ((1u << (top - 8 * pb)) - 1)) : 0u;
											}
										else {
											const u32 cq = q - P.sbytes;
											val = cnt[r] >> (8 * (P.kff ? (P.cbytes - 1 - cq) : cq));
										}
										s_stage[bpos - c0] = (uint8_t)val;
									}
								}
							}
						}
					}
					__syncthreads();
					const u64 gbyte0 = two_phase ? 0 : s_tile_off * rec_bytes;
					const bool fits = two_phase || gbyte0 + tile_bytes <= out_capacity;
					if (!fits && tid == 0)
						atomicOr(err, KERR_CAPACITY);
					if (fits) {
						/* window [c0,c1) -> dst at gbyte0 + c0: leading bytes up to 4-byte alignment, dwords, trailing bytes */
						const u64 g0 = gbyte0 + c0;
						const u32 len = c1 - c0;
						u32 head = (u32)((4 - ((uintptr_t)(dst + g0) & 3)) & 3);
						if (head > len)
							head = len;
						const u32 ndw = (len - head) >> 2;
						const u32 tail0 = head + (ndw << 2);
						if (tid < head)
							dst[g0 + tid] = s_stage[tid];
						if (tid >= 32 && tid - 32 < len - tail0)
							dst[g0 + tail0 + (tid - 32)] = s_stage[tail0 + (tid - 32)];
						u32 *gd = reinterpret_cast<u32 *>(dst + g0 + head);
						for (u32 w = tid; w < ndw; w += CP_BLOCK) {
							const uint8_t *sp = s_stage + head + (w << 2);
							gd[w] = (u32)sp[0] | ((u32)sp[1] << 8) | ((u32)sp[2] << 16) | ((u32)sp[3] << 24);
						}
					}
					__syncthreads();
				}
			}
			if (use_lut) {
				/* small LUTs are sharded: in sorted order every tile in flight updates the same one or two entries, and
				 * same-address device atomics serialise at ~11 ns each (that alone was 11 of this kernel's 12 ms in round 1) */
				u64 *lut = lut_base + (size_t)(tile % lut_shards) * lut_stride;
				for (u32 j = tid; j < tile_counted; j += CP_BLOCK) {
					const u32 pf = s_pref[j];
					if (j + 1 == tile_counted || s_pref[j + 1] != pf)
						atomicAdd(&lut[pf], (u64)(j + 1));
					if (j > 0 && s_pref[j - 1] != pf)
						atomicAdd(&lut[pf], (u64)0 - (u64)j);
				}
			}
		}
		TRACE_STAMP(1, tile, 7);
	}
}

/* End of the bin: fold the tally shards into stats[0..2], stats[3] = n_total = n_rec (kb_sorter.h:1166), and sum the LUT shards into the
 * caller's LUT. One small workgroup after the compaction. (Rounds 2a-2b had the compaction's last workgroup do this: every tile then ended
 * with "wait for my atomics, count myself in, barrier" — ~4 us during which the workgroup's registers and LDS sat idle; with ~12 rounds of
 * tiles per CU and bin that cost more than this launch does.) */
__global__ void __launch_bounds__(256) k_compact_fold(const GrpFold gf, u32 lut_shards, u64 lut_stride, u32 two_phase, u32 rec_bytes, u32 *err)
{
	const u32 bin = blockIdx.x; /* one workgroup per bin of the group */
	const u64 *__restrict__ stat_shards = gf.tally[bin];
	u64 *__restrict__ stats = gf.stats[bin];
	const u64 *__restrict__ lut_base = gf.lut_base[bin];
	u64 *__restrict__ lut_out = gf.lut_out[bin];
	if (threadIdx.x < 64) {
		const u32 lane = threadIdx.x;
		for (int j = 0; j < 3; ++j) {
			u64 v = lane < CP_SHARDS ? stat_shards[lane * 4 + j] : 0;
			v = wave_sum<u64>(v);
			if (lane == 0)
				stats[j] = v;
		}
		if (lane == 0)
			stats[3] = gf.n[bin];
	}
	if (two_phase) { /* tile counts -> exclusive prefixes, in place; the total goes to [n_tiles] and, in bytes, to out_bytes */
		__shared__ u64 s_scan[5];
		u64 *st = gf.status[bin];
		const u32 nt = gf.n_tiles[bin];
		/* 4096 tiles per round: coalesced into LDS, each thread scans its 16 consecutive words there, one workgroup scan, coalesced back
		 * (a 48 M k-mer bin is ~12 000 tiles; a strided scan straight from memory took 39 us per group) */
		constexpr u32 CH = CP_FOLD_CHUNK, PER = CH / 256;
		static_assert(CH % 256 == 0 && CH >= 256, "CP_FOLD_CHUNK");
		__shared__ u64 s_buf[CH];
		u64 carry = 0;
		for (u32 c0 = 0; c0 < nt; c0 += CH) {
			const u32 cn = (nt - c0) < CH ? (nt - c0) : CH;
			for (u32 i = threadIdx.x; i < CH; i += 256)
				s_buf[i] = i < cn ? st[c0 + i] : 0;
			__syncthreads();
			u64 sum = 0;
#pragma unroll
			for (u32 j = 0; j < PER; ++j)
				sum += s_buf[threadIdx.x * PER + j];
			u64 total;
			u64 run = carry + block_excl_sum<4, u64>(sum, s_scan, total);
#pragma unroll
			for (u32 j = 0; j < PER; ++j) {
				const u64 v = s_buf[threadIdx.x * PER + j];
				s_buf[threadIdx.x * PER + j] = run;
				run += v;
			}
			__syncthreads();
			for (u32 i = threadIdx.x; i < cn; i += 256)
				st[c0 + i] = s_buf[i];
			carry += total;
			__syncthreads();
		}
		if (threadIdx.x == 0) {
			st[nt] = carry;
			*gf.out_bytes[bin] = carry * rec_bytes;
			if (carry * rec_bytes > gf.out_capacity[bin])
				atomicOr(err, KERR_CAPACITY);
		}
	}
	if (lut_shards > 1) { /* <= 8 K loads (kmc_hip.hip lut_shards_for), 8 in flight per thread */
		for (u64 i = threadIdx.x; i < lut_stride; i += 256) {
			u64 v = 0;
			for (u32 s0 = 0; s0 < lut_shards; s0 += 8) {
				u64 part[8];
#pragma unroll
				for (int q = 0; q < 8; ++q)
					part[q] = (s0 + q < lut_shards) ? lut_base[(size_t)(s0 + q) * lut_stride + i] : 0ull;
#pragma unroll
				for (int q = 0; q < 8; ++q)
					v += part[q];
			}
			lut_out[i] = v;
		}
	}
}

/* Two-phase output, second phase: one WAVE per tile moves the tile's records from its span of the scratch array to their place in the bin's
 * output (prefix from k_compact_fold). The source is dword-aligned, the destination is not: leading bytes, then aligned destination dwords
 * funnel-shifted out of two source dwords, then trailing bytes. No barriers. */
__global__ void __launch_bounds__(256) k_compact_gather(const GrpGather gg, u32 rec_bytes, u64 tile_pitch)
{
	const u32 lane = threadIdx.x & 63;
	const u32 gtile = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (gtile >= gg.tile_prefix[gg.g])
		return;
	const u32 bin = grp_find(gg.tile_prefix, gg.g, gtile);
	const u32 tile = gtile - gg.tile_prefix[bin];
	const u32 nt = gg.tile_prefix[bin + 1] - gg.tile_prefix[bin];
	const u64 *__restrict__ prefix = gg.prefix[bin];
	if (prefix[nt] * rec_bytes > gg.out_capacity[bin])
		return; /* KERR_CAPACITY was raised by the fold */
	const u64 first = prefix[tile];
	const u32 len = (u32)(prefix[tile + 1] - first) * rec_bytes;
	if (!len)
		return;
	const uint8_t *__restrict__ src = gg.scratch[bin] + (gg.src_rec[bin] ? gg.src_rec[bin][tile] : (u64)tile) * tile_pitch;
	uint8_t *__restrict__ dst = gg.out[bin] + first * rec_bytes;
	u32 head = (u32)((4 - ((uintptr_t)dst & 3)) & 3);
	if (head > len)
		head = len;
	const u32 ndw = (len - head) >> 2, tail0 = head + (ndw << 2);
	if (lane < head)
		dst[lane] = src[lane];
	if (lane >= 32 && lane - 32 < len - tail0)
		dst[tail0 + lane - 32] = src[tail0 + lane - 32];
	const u32 *__restrict__ s32 = reinterpret_cast<const u32 *>(src);
	u32 *__restrict__ d32 = reinterpret_cast<u32 *>(dst + head);
	const u32 sh = 8 * head; /* destination dword w = source bytes [head + 4w, head + 4w + 4) */
	for (u32 w = lane; w < ndw; w += 64) {
		const u32 lo = s32[w];
		const u32 hi = head ? s32[w + 1] : 0; /* inside the tile's span: head + 4w + 4 <= len */
		d32[w] = head ? ((lo >> sh) | (hi << (32 - sh))) : lo;
	}
}

#endif
