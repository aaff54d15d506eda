/*
 * kmc_amd/csrc/stage1_kernels.hip.h — the kernels of KMC's STAGE 1 on gfx950 (docs/history/DESIGN_rounds_1_to_5.md §9): text of a FASTA/FASTQ part ->
 * codes -> minimizer signatures -> super-k-mers -> bin records + the collector's sums. Reachable through kmc_hip_split_part (one part, host
 * text -> host records: the engine of the stage-1 worker plug-in), kmc_hip_split_reads_plan/_emit (codes in HBM -> bins in HBM in the layout
 * kmc_hip_process_bins_device takes) and the test hook kmc_hip_debug_split_reads. The signature -> bin map is an input (stage 0 stays the
 * reference's). GPU-validated: signatures, cut, bin totals / layout / emit; under emulation only so far: text -> codes, record check, k+x sums.
 *
 * What the reference does (kmc_core/splitter.cpp:557-672, CSplitter::ProcessReads) is a sequential scan per read with a two-variable state
 * (current signature, its position). Its RESULT has a data-parallel description, which oracle/stage1_oracle.c's line-by-line restatement
 * confirms on every test (tests/test_stage1_emulated.py):
 *   - symbols are codes 0..3, anything else (N, read boundaries: the host joins reads with a separator) is "invalid";
 *   - the signature of the k-mer at q is the MINIMUM of norm[m-mer] over the k - m + 1 m-mers inside it (norm: kmc_api/mmer.h:39-95,
 *     the smaller strand among the allowed m-mers, 4^m if none) — the reference's tie and fall-out rules only decide WHICH occurrence it
 *     remembers, never the value;
 *   - a super-k-mer is a maximal run of consecutive valid k-mers with one signature value, cut into pieces of 256 k-mers counted from the
 *     run's start (one byte holds the number of extra symbols, splitter.cpp:651-658); its bin record is kb_collector.cpp:57-71.
 *
 *   k_s1_signatures : codes -> signature per k-mer position (0xFFFFFFFF where no valid k-mer starts); only the test hook stores them
 *   k_s1_cut<FUSED> : signatures (from memory, or computed by the tile itself) -> the super-k-mers in position order: first symbol,
 *                     length in symbols, signature
 *   k_s1_bin_totals / k_s1_bin_layout / k_s1_emit : super-k-mers -> bin records, scattered into per-bin byte streams (see below)
 * Both are tile-parallel; k_s1_cut carries "where did the current run start" and "how many super-k-mers so far" across workgroups with two
 * decoupled look-backs (latest non-zero, and sum) over 64-bit status words.
 */
#ifndef KMC_AMD_STAGE1_KERNELS_HIP_H
#define KMC_AMD_STAGE1_KERNELS_HIP_H

#include "kernels.hip.h"

constexpr int S1_BLOCK = 256, S1_PER = 4, S1_TILE = S1_BLOCK * S1_PER; /* positions per workgroup */
constexpr int S1_MAX_K = 256;
constexpr u32 S1_NOSIG = 0xFFFFFFFFu;
/* A line of mem_part_pmm_reads symbols or more reaches CSplitter::ProcessReads in pieces that overlap by k - 1 symbols (splitter.cpp:141-145,
 * :226-231; a long-read part likewise, GetSeqLongRead :80-84), and every piece starts its super-k-mers afresh: the k-mer that starts at a
 * multiple of stride = mem_part_pmm_reads - k + 1 (counted from the line's first symbol) never shares a super-k-mer with the k-mer before it.
 * k_s1_check_records / k_s1_mark_raw put this bit into the code of those positions (codes are 0..3 or negative: every consumer masks with 3
 * or tests the sign), k_s1_cut starts a run there. The k-mers — and so the database — do not depend on it; "Total no. of super-k-mers" does. */
constexpr int8_t S1_PIECE_MARK = 0x40;

/* norm of an m-mer (kmc_api/mmer.h:39-95: the smaller of the m-mer and its reverse complement among the ALLOWED ones, 4^m if neither is)
 * computed, not looked up: the reference's table is 1 MB at m = 9, and one gather per position from it made the L2 -> L1 path the limit of
 * the first version of these kernels (4.4 ms per 300 M positions, profiles/r02/s1_bench_v2_*). */
__host__ __device__ __forceinline__ bool s1_allowed(u32 x, u32 m) /* mmer.h:39-64 */
{
	if ((x & 0x3f) == 0x3f || (x & 0x3f) == 0x3b || (x & 0x3c) == 0x3c) /* ends with TTT or TGT, or has TT in front of its last symbol */
		return false;
	const u32 is_a = ~(x | (x >> 1)) & 0x55555555u & ((1u << (2 * m)) - 1u); /* bit 2j: symbol j (from the end) is A */
	const u32 aa = is_a & (is_a >> 2);                                      /* bit 2j: symbols j and j + 1 are both A */
	if (aa & ((1u << (2 * (m - 2))) - 1u))                                  /* AA anywhere but in the two leading symbols */
		return false;
	return (x >> (2 * (m - 3))) != 0x04u; /* does not start with ACA */
}
__host__ __device__ __forceinline__ u32 s1_revcomp(u32 x, u32 m) /* mmer.h:69-80 */
{
	u32 y = ~x;
	y = ((y & 0x33333333u) << 2) | ((y >> 2) & 0x33333333u); /* reverse the sixteen 2-bit groups of the word */
	y = ((y & 0x0F0F0F0Fu) << 4) | ((y >> 4) & 0x0F0F0F0Fu);
	y = ((y & 0x00FF00FFu) << 8) | ((y >> 8) & 0x00FF00FFu);
	y = (y << 16) | (y >> 16);
	return y >> (32 - 2 * m);
}
__host__ __device__ __forceinline__ u32 s1_norm(u32 x, u32 m)
{
	const u32 special = 1u << (2 * m), r = s1_revcomp(x, m);
	const u32 a = s1_allowed(x, m) ? x : special, b = s1_allowed(r, m) ? r : special;
	return a < b ? a : b;
}

/* Signatures of the k-mers that start at positions base .. base + cnt - 1 (base may be -1; a position outside [0, n) has none) into
 * s_sig[0 .. cnt), cnt <= S1_TILE + 2. All threads of the block call; the result is visible to all of them on return.
 * A thread owns S1_SIG_PER CONSECUTIVE positions in both steps (an odd number: its LDS accesses are bank-conflict free): the m-mer is rolled
 * from one position to the next (2 byte reads instead of m), and the minimum over the k - m + 1 m-mers of a k-mer is put together from what
 * the thread's windows share, a suffix of the first S1_SIG_PER - 1 values and a prefix of the last ones (w + 4 reads for 5 windows instead
 * of 5 w). A k-mer is valid iff all its m-mers are: the maximum over the same window tells (S1_NOSIG marks an m-mer with an invalid symbol). */
constexpr int S1_SIG_PER = 5;
static_assert(S1_BLOCK * S1_SIG_PER >= S1_TILE + 2 + S1_MAX_K - 5, "one round covers every m-mer of the span");
struct S1SigLds {
	int8_t c[(S1_TILE + 2 + S1_MAX_K + 3) / 4 * 4]; /* symbols of the span: cnt + k - 1 */
	u32 mm[S1_BLOCK * S1_SIG_PER + S1_MAX_K + 8];   /* norm of the m-mer at each position of the span (reads run past the last one, unused) */
};
struct S1MinMax {
	u32 mn, mx;
	__device__ __forceinline__ void add(u32 v)
	{
		mn = v < mn ? v : mn;
		mx = v > mx ? v : mx;
	}
	__device__ __forceinline__ void add(const S1MinMax &o)
	{
		mn = o.mn < mn ? o.mn : mn;
		mx = o.mx > mx ? o.mx : mx;
	}
};
__device__ __forceinline__ void s1_signatures_to_lds(const int8_t *__restrict__ codes, u64 n, long long base, u32 cnt, u32 k, u32 m, S1SigLds &L, u32 *s_sig)
{
	const u32 tid = threadIdx.x;
	const u32 span = cnt + k - 1; /* symbols looked at */
	for (u32 i = tid; i < span; i += S1_BLOCK) {
		const long long p = base + (long long)i;
		L.c[i] = (p >= 0 && (u64)p < n) ? codes[p] : (int8_t)-1;
	}
	__syncthreads();
	/* norm of every m-mer that starts at one of the cnt positions or in the k - m positions after them */
	const u32 n_mm = cnt + k - m, i0 = tid * S1_SIG_PER;
	if (i0 < n_mm) {
		const u32 mask = (1u << (2 * m)) - 1u;
		u32 x = 0, bad = 0;
		for (u32 j = 0; j < m; ++j) {
			const int8_t c = L.c[i0 + j];
			bad += c < 0 ? 1u : 0u;
			x = (x << 2) | (u32)(c & 3);
		}
#pragma unroll
		for (u32 jj = 0; jj < (u32)S1_SIG_PER; ++jj) {
			const u32 i = i0 + jj;
			if (i >= n_mm)
				break;
			L.mm[i] = bad ? S1_NOSIG : s1_norm(x & mask, m);
			if (i + 1 < n_mm) {
				const int8_t cin = L.c[i + m], cout = L.c[i];
				bad += (cin < 0 ? 1u : 0u) - (cout < 0 ? 1u : 0u);
				x = (x << 2) | (u32)(cin & 3);
			}
		}
	}
	__syncthreads();
	const u32 w = k - m + 1;
	if (i0 < cnt) {
		if (w >= (u32)S1_SIG_PER) {
			S1MinMax mid{S1_NOSIG, 0u};
			for (u32 j = S1_SIG_PER - 1; j < w; ++j) /* shared by the thread's windows */
				mid.add(L.mm[i0 + j]);
			u32 a[S1_SIG_PER - 1], b[S1_SIG_PER - 1];
#pragma unroll
			for (int j = 0; j < S1_SIG_PER - 1; ++j) {
				a[j] = L.mm[i0 + j];
				b[j] = L.mm[i0 + w + j]; /* past the span for the last positions: only windows that do not exist use those */
			}
#pragma unroll
			for (int i = 0; i < S1_SIG_PER; ++i) {
				S1MinMax r = mid;
#pragma unroll
				for (int j = i; j < S1_SIG_PER - 1; ++j)
					r.add(a[j]);
#pragma unroll
				for (int j = 0; j < i; ++j)
					r.add(b[j]);
				if (i0 + i < cnt)
					s_sig[i0 + i] = r.mx == S1_NOSIG ? S1_NOSIG : r.mn;
			}
		} else { /* k within 3 of the signature length */
			for (u32 i = i0; i < i0 + S1_SIG_PER && i < cnt; ++i) {
				S1MinMax r{S1_NOSIG, 0u};
				for (u32 j = 0; j < w; ++j)
					r.add(L.mm[i + j]);
				s_sig[i] = r.mx == S1_NOSIG ? S1_NOSIG : r.mn;
			}
		}
	}
	__syncthreads();
}

/* test hook path: the signature of every position to global memory */
__global__ void __launch_bounds__(S1_BLOCK) k_s1_signatures(const int8_t *__restrict__ codes, u64 n, u32 k, u32 m, u32 *__restrict__ sig)
{
	__shared__ S1SigLds L;
	__shared__ u32 s_sig[S1_TILE + 2];
	const u64 t0 = (u64)blockIdx.x * S1_TILE;
	s1_signatures_to_lds(codes, n, (long long)t0, S1_TILE, k, m, L, s_sig);
	for (u32 i = threadIdx.x; i < (u32)S1_TILE; i += S1_BLOCK)
		if (t0 + i < n)
			sig[t0 + i] = s_sig[i];
}

/* decoupled look-back like lookback64, but the combination is "the latest non-zero value" (value = position + 1 of the last run start):
 * returns the last run start + 1 before this tile (0 = none), valid in lane 0 */
__device__ __forceinline__ u64 lookback64_last(u64 *status, u32 tile, u64 own_last1, u32 lane, u32 *err)
{
	if (tile == 0) {
		if (lane == 0)
			st_agent(&status[0], ST64_PREFIX | own_last1);
		return 0;
	}
	if (lane == 0)
		st_agent(&status[tile], ST64_AGG | own_last1);
	long long tbase = (long long)tile - 1;
	u64 found = 0;
	LbWatch watch;
	while (true) {
		const long long t = tbase - (long long)lane;
		const u64 v = t >= 0 ? ld_agent(&status[t]) : ST64_PREFIX;
		const u64 flag = v & ~ST64_MASK;
		const u64 m_pref = __ballot(flag == ST64_PREFIX);
		const u64 m_zero = __ballot(flag == 0);
		const int pl = m_pref ? (__ffsll(m_pref) - 1) : 64;
		const u64 need = pl < 63 ? ((2ull << pl) - 1) : ~0ull;
		if (m_zero & need) {
			if (lb_blocked(watch, err)) {
				if (lane == 0)
					lb_gave_up(watch, err, KERR_WATCHDOG | KERR_AT_STAGE1, lane, tile, tbase, 0u);
				break;
			}
			__builtin_amdgcn_s_sleep(1);
			continue;
		}
		/* the nearest tile (lowest lane <= pl) with a non-zero value wins */
		const u64 m_has = __ballot((int)lane <= pl && (v & ST64_MASK) != 0);
		if (m_has) {
			const int src = __ffsll(m_has) - 1;
			found = __shfl(v & ST64_MASK, src);
			break;
		}
		if (pl < 64)
			break;
		tbase -= 64;
	}
	if (lane == 0)
		st_agent(&status[tile], ST64_PREFIX | (own_last1 ? own_last1 : found));
	return found;
}

/* One workgroup cuts S1_SUB consecutive tiles (S1_WG_TILE positions): one ticket and one pair of look-backs per workgroup — with one tile per
 * workgroup the kernel ran at the rate of its same-address ticket atomic (50 tiles/us: 5.9 ms per 300 M positions, profiles/r02/s1_bench_v1*).
 * status_last / status_cnt: one zeroed u64 per WORKGROUP tile each (s1_cut_tiles(n) of them). sk_* receive the super-k-mers in position order;
 * *n_sk their number (written by the last tile). sk_cap bounds the writes (KERR_CAPACITY beyond it).
 * FUSED = false: signatures come from `sig` (k_s1_signatures ran before; codes, m unused). FUSED = true: the workgroup computes the
 * signatures it needs in LDS itself (sig unused): 1 byte per symbol read instead of 4 written + 4 read. */
#ifndef S1_SUB_N
#define S1_SUB_N 4
#endif
constexpr int S1_SUB = S1_SUB_N, S1_WG_TILE = S1_TILE * S1_SUB;
static_assert(S1_PER == 4, "the pieces of a thread's positions are packed into one 32-bit word");
__host__ __device__ inline u64 s1_cut_tiles(u64 n) { return (n + S1_WG_TILE - 1) / S1_WG_TILE; }

template <bool FUSED>
__global__ void __launch_bounds__(S1_BLOCK) k_s1_cut(const u32 *__restrict__ sig, const int8_t *__restrict__ codes, u32 m, u64 n, u32 k,
                                                      u64 *status_last, u64 *status_cnt, u32 *ticket_ctr, u64 *__restrict__ sk_pos, u32 *__restrict__ sk_len,
                                                      u32 *__restrict__ sk_sig, u64 sk_cap, u64 *n_sk, const u64 *has_marks, u32 *err)
{
	__shared__ u32 s_sig[S1_WG_TILE + 2]; /* signatures of positions w0 - 1 .. w0 + S1_WG_TILE */
	__shared__ u32 s_mark;
	__shared__ u64 s_tmp64[S1_BLOCK / 64 + 1];
	__shared__ u32 s_tmp32[S1_BLOCK / 64 + 1];
	__shared__ u64 s_carry_last1, s_carry_cnt;
	__shared__ u32 s_ticket;
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0)
		s_ticket = atomicAdd(ticket_ctr, 1u);
	__syncthreads();
	const u32 tile = s_ticket;
	const u32 num_tiles = (u32)s1_cut_tiles(n);
	if (tile >= num_tiles)
		return;
	const u64 w0 = (u64)tile * S1_WG_TILE;
	if constexpr (FUSED) {
		__shared__ S1SigLds L;
#pragma unroll 1
		for (int sub = 0; sub < S1_SUB; ++sub)
			s1_signatures_to_lds(codes, n, (long long)w0 - 1 + (long long)sub * S1_TILE, sub == S1_SUB - 1 ? S1_TILE + 2 : S1_TILE, k, m, L, s_sig + sub * S1_TILE);
	} else {
		for (u32 i = tid; i < (u32)S1_WG_TILE + 2; i += S1_BLOCK) {
			const long long q = (long long)w0 - 1 + (long long)i;
			s_sig[i] = (q >= 0 && (u64)q < n) ? sig[q] : S1_NOSIG;
		}
		__syncthreads();
	}
	if (FUSED && has_marks && *has_marks) {
		/* S1_PIECE_MARK somewhere in the part (uniform, rare: a line of mem_part_pmm_reads symbols or more): a marked position starts a run whatever
		 * the signatures say. Marks are >= stride positions apart and the host refuses a stride within a workgroup's window, so the window holds at
		 * most one: the signatures from it on get bit 31 — they then differ from the one before the mark and from nothing else they are compared with. */
		if (tid == 0)
			s_mark = 0xFFFFFFFFu;
		__syncthreads();
		for (u32 i = tid; i < (u32)S1_WG_TILE + 2; i += S1_BLOCK) {
			const long long q = (long long)w0 - 1 + (long long)i;
			if (q >= 0 && (u64)q < n && (codes[q] & 0xC0) == S1_PIECE_MARK)
				atomicMin(&s_mark, i);
		}
		__syncthreads();
		const u32 mk = s_mark;
		if (mk != 0xFFFFFFFFu)
			for (u32 i = tid; i < (u32)S1_WG_TILE + 2; i += S1_BLOCK)
				if (i >= mk && s_sig[i] != S1_NOSIG)
					s_sig[i] |= 0x80000000u;
		__syncthreads();
	}
	/* Pass A, per sub-tile: run starts inside this thread's S1_PER positions (valid, and the k-mer before is invalid or has another
	 * signature); the last one before the thread inside the sub-tile; the sub-tile's own last one. */
	u64 before1[S1_SUB], sub_last1[S1_SUB]; /* position + 1, 0 = none */
#pragma unroll
	for (int sub = 0; sub < S1_SUB; ++sub) {
		const u32 l0 = (u32)sub * S1_TILE + tid * S1_PER; /* index of position t0 - 1 in s_sig */
		const u64 t0 = w0 + l0;
		u64 last1 = 0;
#pragma unroll
		for (int j = 0; j < S1_PER; ++j)
			if (s_sig[l0 + j + 1] != S1_NOSIG && s_sig[l0 + j] != s_sig[l0 + j + 1])
				last1 = t0 + j + 1;
		before1[sub] = block_excl_max<S1_BLOCK / 64, u64>(last1, s_tmp64);
		u64 mx = wave_incl_max<u64>(last1, lane);
		if (lane == 63)
			s_tmp64[wave] = mx;
		__syncthreads();
		u64 w = 0;
#pragma unroll
		for (int i = 0; i < S1_BLOCK / 64; ++i)
			w = s_tmp64[i] > w ? s_tmp64[i] : w;
		sub_last1[sub] = w;
		__syncthreads();
	}
	if (wave == 0) {
		u64 w = 0;
#pragma unroll
		for (int sub = 0; sub < S1_SUB; ++sub)
			w = sub_last1[sub] ? sub_last1[sub] : w; /* positions grow with sub: the last non-zero one */
		const u64 carry = lookback64_last(status_last, tile, w, lane, err);
		if (lane == 0)
			s_carry_last1 = carry;
	}
	__syncthreads();
	/* Pass B: ends. A super-k-mer ends at q if q is valid and the next k-mer is invalid / has another signature / q is the 256th k-mer of its
	 * piece (pieces are counted from the run's start, wherever that was). */
	u64 carry_last1 = s_carry_last1; /* the last run start before the sub-tile at hand */
	u32 end_bits[S1_SUB], pieces[S1_SUB], off[S1_SUB], wg_ends = 0;
#pragma unroll
	for (int sub = 0; sub < S1_SUB; ++sub) {
		const u32 l0 = (u32)sub * S1_TILE + tid * S1_PER;
		const u64 t0 = w0 + l0;
		u64 cur1 = before1[sub] ? before1[sub] : carry_last1;
		u32 n_end = 0;
		end_bits[sub] = 0, pieces[sub] = 0;
#pragma unroll
		for (int j = 0; j < S1_PER; ++j) {
			const u64 q = t0 + j;
			const u32 prev = s_sig[l0 + j], me = s_sig[l0 + j + 1], next = s_sig[l0 + j + 2];
			if (me != S1_NOSIG) {
				if (prev != me)
					cur1 = q + 1; /* a run starts here */
				const u32 in_piece = (u32)((q - (cur1 - 1)) & 255u); /* k-mers of this piece before q */
				if (next != me || in_piece == 255u) {
					end_bits[sub] |= 1u << j;
					pieces[sub] |= in_piece << (8 * j);
					++n_end;
				}
			}
		}
		u32 sub_ends;
		off[sub] = wg_ends + block_excl_sum<S1_BLOCK / 64, u32>(n_end, s_tmp32, sub_ends);
		wg_ends += sub_ends;
		carry_last1 = sub_last1[sub] ? sub_last1[sub] : carry_last1;
	}
	if (wave == 0) {
		const u64 excl = lookback64(status_cnt, tile, (u64)wg_ends, lane, err, KERR_WATCHDOG | KERR_AT_STAGE1);
		if (lane == 0) {
			s_carry_cnt = excl;
			if (tile == num_tiles - 1)
				*n_sk = excl + wg_ends;
		}
	}
	__syncthreads();
	/* Pass C: the super-k-mers, in position order */
#pragma unroll
	for (int sub = 0; sub < S1_SUB; ++sub) {
		const u32 l0 = (u32)sub * S1_TILE + tid * S1_PER;
		const u64 t0 = w0 + l0;
		u64 idx = s_carry_cnt + off[sub];
#pragma unroll
		for (int j = 0; j < S1_PER; ++j)
			if (end_bits[sub] & (1u << j)) {
				const u32 piece = (pieces[sub] >> (8 * j)) & 255u;
				if (idx < sk_cap) {
					sk_pos[idx] = t0 + j - piece;
					sk_len[idx] = k + piece;
					sk_sig[idx] = s_sig[l0 + j + 1] & 0x7FFFFFFFu;
				} else
					atomicOr(err, KERR_CAPACITY);
				++idx;
			}
	}
}

/* ------------------------------------------------------------------------------------------------ bin scatter
 * super-k-mers -> bin records ([len - k][ceil(len/4) bytes of 2-bit symbols, first symbol in bits 7:6], kb_collector.cpp:57-71) in per-bin
 * byte streams laid out one after the other in ONE buffer, each with the list of expander-pack boundaries stage 2 wants: what
 * kmc_hip_process_bins_device takes as kmc_hip_bin_desc {d_superkmers, size, n_rec, d_pack_start, n_packs}.
 *   k_s1_bin_totals : bytes, super-k-mers and k-mers per bin (LDS counters per workgroup, one global atomic per bin a workgroup touched)
 *   k_s1_bin_layout : one workgroup: bin_base[b] (256-byte aligned; stage 2 stages the image with aligned 16-byte loads), the bin's slice of
 *                     the pack-start array (ceil(bytes / S1_PACK_BYTES) packs + the closing entry), cursors at the bases
 *   k_s1_emit       : a workgroup reserves, per bin it touches, ONE contiguous segment for its tile's records (global cursor atomic) and
 *                     writes the records of the tile into their segments in arbitrary order. A segment starts on a record boundary, so the
 *                     segment that covers the j-th multiple of S1_PACK_BYTES of its bin supplies pack boundary j (queues.h:376-396: a pack
 *                     is any run of whole records): one boundary per multiple, none missing, packs of at most 2 S1_PACK_BYTES.
 * The order of super-k-mers inside a bin is therefore not the read order. The reference's is not either with more than one splitter thread
 * (each thread flushes its own buffers, kb_collector.cpp:88-106), and stage 2 sees only the multiset of k-mers. */
#ifndef S1_SK_TILE_N
#define S1_SK_TILE_N 1024
#endif
#ifndef S1_PACK_BYTES_N
#define S1_PACK_BYTES_N (1u << 18)
#endif
constexpr int S1_SK_TILE = S1_SK_TILE_N;      /* super-k-mers per workgroup of k_s1_bin_totals / k_s1_emit */
constexpr int S1_MAX_BINS = 2048;             /* KMC allows -n up to 2000 bins */
constexpr u32 S1_PACK_BYTES = S1_PACK_BYTES_N; /* > the bytes one tile can put into one bin (1024 records of <= 1 + (256 + 255 + 3) / 4 bytes): a
                                                * segment covers at most one multiple */
constexpr u32 S1_BIN_ALIGN = 256;
static_assert((u32)S1_SK_TILE * (1u + ((u32)S1_MAX_K + 255u + 3u) / 4u) < S1_PACK_BYTES, "a tile's segment must cover at most one pack boundary");

__global__ void __launch_bounds__(256) k_s1_bin_totals(const u32 *__restrict__ sk_len, const u32 *__restrict__ sk_sig, u64 n_sk, u32 k, const int *__restrict__ sig_to_bin,
                                                        u32 n_bins, u64 *__restrict__ bin_bytes, u64 *__restrict__ bin_sk, u64 *__restrict__ bin_kmers, u32 *err)
{
	__shared__ u32 s_bytes[S1_MAX_BINS], s_cnt[S1_MAX_BINS], s_km[S1_MAX_BINS];
	for (u32 b = threadIdx.x; b < n_bins; b += 256)
		s_bytes[b] = s_cnt[b] = s_km[b] = 0;
	__syncthreads();
	const u64 i0 = (u64)blockIdx.x * S1_SK_TILE;
	for (u32 j = threadIdx.x; j < (u32)S1_SK_TILE; j += 256) {
		const u64 i = i0 + j;
		if (i < n_sk) {
			const int b = sig_to_bin[sk_sig[i]];
			if (b < 0 || (u32)b >= n_bins)
				atomicOr(err, KERR_CORRUPT); /* a signature the map does not know */
			else {
				atomicAdd(&s_bytes[b], 1u + (sk_len[i] + 3u) / 4u);
				atomicAdd(&s_cnt[b], 1u);
				atomicAdd(&s_km[b], sk_len[i] - k + 1u);
			}
		}
	}
	__syncthreads();
	for (u32 b = threadIdx.x; b < n_bins; b += 256)
		if (s_cnt[b]) {
			atomicAdd(&bin_bytes[b], (u64)s_bytes[b]);
			atomicAdd(&bin_sk[b], (u64)s_cnt[b]);
			atomicAdd(&bin_kmers[b], (u64)s_km[b]);
		}
}

/* one workgroup of 256. bin_base[b]: first byte of bin b in the buffer, bin_base[n_bins]: bytes the buffer needs (incl. 256 B of readable slack
 * behind the last bin); pack_base[b]: first entry of bin b in the pack-start array (bin b owns packs + 1 entries, the last one = its size),
 * pack_base[n_bins]: entries in all. pack_start may be NULL (sizing call). */
__global__ void __launch_bounds__(256) k_s1_bin_layout(const u64 *__restrict__ bin_bytes, u32 n_bins, u64 *__restrict__ bin_base, u64 *__restrict__ pack_base,
                                                        u64 *__restrict__ cursor, u64 *__restrict__ pack_start)
{
	__shared__ u64 s_tmp[5];
	const u32 per = (n_bins + 255) / 256, lo = threadIdx.x * per;
	u64 sum = 0, packs = 0;
	for (u32 j = 0; j < per; ++j)
		if (lo + j < n_bins) {
			const u64 by = bin_bytes[lo + j];
			sum += (by + S1_BIN_ALIGN - 1) / S1_BIN_ALIGN * S1_BIN_ALIGN;
			packs += (by + S1_PACK_BYTES - 1) / S1_PACK_BYTES + 1;
		}
	u64 total, total_packs;
	u64 run = block_excl_sum<4, u64>(sum, s_tmp, total);
	u64 prun = block_excl_sum<4, u64>(packs, s_tmp, total_packs);
	for (u32 j = 0; j < per; ++j)
		if (lo + j < n_bins) {
			const u64 by = bin_bytes[lo + j], np = (by + S1_PACK_BYTES - 1) / S1_PACK_BYTES;
			bin_base[lo + j] = run;
			cursor[lo + j] = run;
			pack_base[lo + j] = prun;
			if (pack_start)
				pack_start[prun + np] = by;
			run += (by + S1_BIN_ALIGN - 1) / S1_BIN_ALIGN * S1_BIN_ALIGN;
			prun += np + 1;
		}
	if (threadIdx.x == 0) {
		bin_base[n_bins] = total + S1_BIN_ALIGN;
		pack_base[n_bins] = total_packs;
	}
}

__global__ void __launch_bounds__(256) k_s1_emit(const int8_t *__restrict__ codes, const u64 *__restrict__ sk_pos, const u32 *__restrict__ sk_len,
                                                  const u32 *__restrict__ sk_sig, u64 n_sk, u32 k, const int *__restrict__ sig_to_bin, u32 n_bins,
                                                  const u64 *__restrict__ bin_base, const u64 *__restrict__ pack_base, u64 *cursor, uint8_t *__restrict__ out,
                                                  u64 *__restrict__ pack_start)
{
	__shared__ u32 s_bytes[S1_MAX_BINS]; /* bytes of this tile per bin, then the running offset inside the tile's segment */
	__shared__ u64 s_base[S1_MAX_BINS];  /* where the tile's segment of each bin starts in `out` */
	for (u32 b = threadIdx.x; b < n_bins; b += 256)
		s_bytes[b] = 0;
	__syncthreads();
	const u64 i0 = (u64)blockIdx.x * S1_SK_TILE;
	for (u32 j = threadIdx.x; j < (u32)S1_SK_TILE; j += 256) {
		const u64 i = i0 + j;
		if (i < n_sk) {
			const int b = sig_to_bin[sk_sig[i]];
			if (b >= 0 && (u32)b < n_bins)
				atomicAdd(&s_bytes[b], 1u + (sk_len[i] + 3u) / 4u);
		}
	}
	__syncthreads();
	for (u32 b = threadIdx.x; b < n_bins; b += 256) {
		const u32 bytes = s_bytes[b];
		if (bytes) {
			const u64 at = atomicAdd(&cursor[b], (u64)bytes);
			s_base[b] = at;
			const u64 rel = at - bin_base[b], j = (rel + S1_PACK_BYTES - 1) / S1_PACK_BYTES;
			if (j * S1_PACK_BYTES < rel + bytes) /* this segment covers the j-th multiple of the pack size: its start is pack boundary j */
				pack_start[pack_base[b] + j] = rel;
		}
		s_bytes[b] = 0;
	}
	__syncthreads();
	for (u32 j = threadIdx.x; j < (u32)S1_SK_TILE; j += 256) {
		const u64 i = i0 + j;
		if (i >= n_sk)
			continue;
		const int b = sig_to_bin[sk_sig[i]];
		if (b < 0 || (u32)b >= n_bins)
			continue;
		const u32 len = sk_len[i], bytes = 1u + (len + 3u) / 4u;
		uint8_t *dst = out + s_base[b] + atomicAdd(&s_bytes[b], bytes);
		const int8_t *src = codes + sk_pos[i];
		dst[0] = (uint8_t)(len - k);
		/* four symbols per byte, first one in bits 7:6. Eight symbols per (unaligned) 8-byte load while they last: one load per symbol made
		 * this kernel wait for its 36 scattered byte loads per record */
		u32 q = 0;
		for (; 8u * (q / 2u) + 8u <= len; q += 2) {
			u64 w8;
			__builtin_memcpy(&w8, src + 4u * q, 8);
			const u32 lo = (u32)w8, hi = (u32)(w8 >> 32);
			dst[1 + q] = (uint8_t)(((lo & 3u) << 6) | (((lo >> 8) & 3u) << 4) | (((lo >> 16) & 3u) << 2) | ((lo >> 24) & 3u));
			dst[2 + q] = (uint8_t)(((hi & 3u) << 6) | (((hi >> 8) & 3u) << 4) | (((hi >> 16) & 3u) << 2) | ((hi >> 24) & 3u));
		}
		for (; q < (len + 3u) / 4u; ++q) { /* the last 1 - 7 symbols; missing ones are zero */
			u32 v = 0;
#pragma unroll
			for (u32 t = 0; t < 4; ++t) {
				const u32 p = 4 * q + t;
				v = (v << 2) | (p < len ? (u32)(src[p] & 3) : 0u);
			}
			dst[1 + q] = (uint8_t)v;
		}
	}
}

/* ------------------------------------------------------------------------------------------------ text -> codes, k+x-mer sums
 * The two kernels a HIP KmcSplitEngine (kmc_amd/host/split_engine.h) still needs. EMULATION-TESTED ONLY (tests/test_stage1_emulated.py): written
 * after the round's GPU budget was spent, launched by no host code yet.
 *
 * k_s1_text_to_codes: one part of FASTA (lines_per_record 2) or FASTQ (4) text as the reference's readers cut it — it starts at a record's
 * title (fastq_reader.cpp) — to the code stream the kernels above take: the symbols of every sequence line (splitter.cpp:41-47: ACGT acgt ->
 * 0..3, anything else negative) followed by ONE negative byte where the line ends. A line ends at '\n'; a '\r' right in front of it is
 * dropped. Line number = number of '\n' before the byte (sum look-back over tiles), sequence lines are those with number = 1 mod
 * lines_per_record, the output position of a byte = number of bytes kept before it (second look-back). nl_pos[i] = position of the i-th '\n'.
 * This is CSplitter::GetSeq (splitter.cpp:92-303) for the inputs it is meant for; what GetSeq does with anything else (blank lines, a lone
 * '\r', a quality line of another length than its sequence: it skips quality by LENGTH, :281) is not reproduced — k_s1_check_records
 * recognises every such part, and the engine must hand those to the reference splitter. */
constexpr int S1_TXT_PER = 16, S1_TXT_TILE = S1_BLOCK * S1_TXT_PER;
constexpr u32 S1_TEXT_BAD = 0x1000u; /* error bit: the part is outside what k_s1_text_to_codes reproduces (its own bit: 0x10 is KERR_PEER) */

__device__ __forceinline__ int8_t s1_symbol_code(uint8_t c)
{
	const uint8_t l = c | 0x20u;
	return l == 'a' ? 0 : l == 'c' ? 1 : l == 'g' ? 2 : l == 't' ? 3 : -1;
}

__global__ void __launch_bounds__(S1_BLOCK) k_s1_text_to_codes(const uint8_t *__restrict__ text, u64 n, u32 lines_per_record, u64 *status_lines, u64 *status_out,
                                                                u32 *ticket_ctr, int8_t *__restrict__ codes, u64 *__restrict__ nl_pos, u64 nl_cap, u64 *__restrict__ seq_start,
                                                                u64 *totals, u32 *err)
{
	__shared__ u32 s_tmp[S1_BLOCK / 64 + 1];
	__shared__ u64 s_carry;
	__shared__ u32 s_ticket;
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0)
		s_ticket = atomicAdd(ticket_ctr, 1u);
	__syncthreads();
	const u32 tile = s_ticket;
	const u32 num_tiles = (u32)((n + S1_TXT_TILE - 1) / S1_TXT_TILE);
	if (tile >= num_tiles)
		return;
	const u64 p0 = (u64)tile * S1_TXT_TILE + (u64)tid * S1_TXT_PER;
	uint8_t c[S1_TXT_PER + 1]; /* the thread's bytes and the one after them ('\r' looks ahead) */
	static_assert(S1_TXT_PER == 16, "one 16-byte load per thread");
	if (p0 + S1_TXT_PER <= n) {
		u32 w[4];
		__builtin_memcpy(w, text + p0, 16); /* one 16-byte load instead of sixteen byte loads */
#pragma unroll
		for (int j = 0; j < S1_TXT_PER; ++j)
			c[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
	} else {
#pragma unroll
		for (int j = 0; j < S1_TXT_PER; ++j)
			c[j] = p0 + j < n ? text[p0 + j] : (uint8_t)0;
	}
	c[S1_TXT_PER] = p0 + S1_TXT_PER < n ? text[p0 + S1_TXT_PER] : (uint8_t)0;
	if (lines_per_record == 0) {
		/* a LONG-READ part behind its title (queues.h:40; CSplitter::GetSeqLongRead, splitter.cpp:70-86): no lines, no records — every byte is a
		 * symbol, the ends of line among them (their code is negative like any other byte that is not ACGT) */
		if (tile == num_tiles - 1 && tid == 0)
			totals[1] = n;
#pragma unroll
		for (int j = 0; j < S1_TXT_PER; ++j)
			if (p0 + j < n)
				codes[p0 + j] = s1_symbol_code(c[j]);
		return;
	}
	u32 my_nl = 0;
#pragma unroll
	for (int j = 0; j < S1_TXT_PER; ++j)
		my_nl += (p0 + j < n && c[j] == '\n') ? 1u : 0u;
	u32 tile_nl;
	const u32 nl_before = block_excl_sum<S1_BLOCK / 64, u32>(my_nl, s_tmp, tile_nl);
	if (wave == 0) {
		const u64 excl = lookback64(status_lines, tile, (u64)tile_nl, lane, err, KERR_WATCHDOG | KERR_AT_STAGE1);
		if (lane == 0) {
			s_carry = excl;
			if (tile == num_tiles - 1)
				totals[0] = excl + tile_nl; /* '\n' in the part */
		}
	}
	__syncthreads();
	const u64 line0 = s_carry + nl_before;
	u64 line = line0;
	__syncthreads();
	/* which bytes are kept: every byte of a sequence line except a '\r' (which must be followed by '\n'), the '\n' included (it becomes the separator) */
	u32 keep = 0, n_keep = 0;
#pragma unroll
	for (int j = 0; j < S1_TXT_PER; ++j) {
		if (p0 + j >= n)
			break;
		const bool is_nl = c[j] == '\n';
		if (c[j] == '\r' && !(p0 + j + 1 < n && c[j + 1] == '\n'))
			atomicOr(err, S1_TEXT_BAD);
		if (c[j] < 32 && !is_nl && c[j] != '\r')
			atomicOr(err, S1_TEXT_BAD); /* GetSeq swallows a control character behind a title's end of line (splitter.cpp:119-123) */
		if ((line & (u64)(lines_per_record - 1)) == 1 && c[j] != '\r') { /* lines_per_record is 2 or 4 */
			keep |= 1u << j;
			++n_keep;
		}
		if (is_nl) {
			if (line < nl_cap)
				nl_pos[line] = p0 + j;
			else
				atomicOr(err, KERR_CAPACITY);
			++line;
		}
	}
	u32 tile_keep;
	const u32 keep_before = block_excl_sum<S1_BLOCK / 64, u32>(n_keep, s_tmp, tile_keep);
	if (wave == 0) {
		const u64 excl = lookback64(status_out, tile, (u64)tile_keep, lane, err, KERR_WATCHDOG | KERR_AT_STAGE1);
		if (lane == 0) {
			s_carry = excl;
			if (tile == num_tiles - 1)
				totals[1] = excl + tile_keep; /* bytes of the code stream */
		}
	}
	__syncthreads();
	u64 at = s_carry + keep_before;
	line = line0;
#pragma unroll
	for (int j = 0; j < S1_TXT_PER; ++j) {
		if (keep & (1u << j))
			codes[at++] = c[j] == '\n' ? (int8_t)-1 : s1_symbol_code(c[j]);
		if (p0 + j < n && c[j] == '\n') {
			/* the end of a title line: the record's sequence starts at the next code (k_s1_check_records marks the pieces of an over-long line from there) */
			if ((line & (u64)(lines_per_record - 1)) == 0 && line < nl_cap)
				seq_start[line / lines_per_record] = at;
			++line;
		}
	}
}

/* The piece starts of a long-read part (see S1_PIECE_MARK): every stride-th position of the raw code stream. One workgroup. */
__global__ void __launch_bounds__(256) k_s1_mark_raw(int8_t *__restrict__ codes, u64 n, u64 stride, u64 *has_marks)
{
	for (u64 p = ((u64)threadIdx.x + 1) * stride; p < n; p += 256 * stride) {
		if (codes[p] >= 0)
			codes[p] |= S1_PIECE_MARK;
		*has_marks = 1;
	}
}

/* One thread per record (n_lines / lines_per_record of them; a FASTA part may end inside its last sequence line): the title starts with the
 * marker, the third line of a FASTQ record with '+', sequence and quality have one length, the sequence is shorter than line_cap
 * (mem_part_pmm_reads: GetSeq cuts longer lines into overlapping pieces). Raises S1_TEXT_BAD. */
__global__ void __launch_bounds__(256) k_s1_check_records(const uint8_t *__restrict__ text, u64 n, const u64 *__restrict__ nl_pos, u64 n_lines, u32 lines_per_record,
                                                            u64 line_cap, u64 stride, const u64 *__restrict__ seq_start, int8_t *__restrict__ codes, u64 *has_marks, u32 *err)
{
	const u64 r = (u64)blockIdx.x * 256 + threadIdx.x;
	const u64 first = r * lines_per_record; /* number of the record's title line */
	if (first > n_lines || (first == n_lines && (n_lines == 0 ? n == 0 : nl_pos[n_lines - 1] + 1 >= n)))
		return; /* no such record: the text ends with the previous one */
	const u64 start = first ? nl_pos[first - 1] + 1 : 0;
	const uint8_t marker = lines_per_record == 4 ? '@' : '>';
	bool bad = text[start] != marker;
	auto line_len = [&](u64 ln) -> u64 { /* without its '\r' */
		const u64 b = ln ? nl_pos[ln - 1] + 1 : 0, e = nl_pos[ln];
		return e - b - ((e > b && text[e - 1] == '\r') ? 1 : 0);
	};
	u64 seq_len = 0;
	if (lines_per_record == 4) {
		if (first + 4 > n_lines)
			bad = true; /* a FASTQ record must be whole, every line terminated (GetSeq drops it otherwise, splitter.cpp:222-223, :283-284) */
		else {
			bad = bad || text[nl_pos[first + 1] + 1] != '+';
			seq_len = line_len(first + 1);
			bad = bad || seq_len != line_len(first + 3);
		}
	} else if (first + 1 > n_lines)
		bad = true; /* a FASTA title without its end of line */
	else {
		const u64 b = nl_pos[first] + 1, e = first + 1 < n_lines ? nl_pos[first + 1] : n;
		seq_len = e - b - ((e > b && text[e - 1] == '\r') ? 1 : 0);
	}
	if (bad) {
		atomicOr(err, S1_TEXT_BAD);
		return;
	}
	if (seq_len >= line_cap) { /* GetSeq hands such a line out in overlapping pieces (splitter.cpp:141-145, :226-231): S1_PIECE_MARK */
		const u64 s0 = seq_start[r];
		for (u64 p = stride; p < seq_len; p += stride)
			if (codes[s0 + p] >= 0)
				codes[s0 + p] |= S1_PIECE_MARK;
		*has_marks = 1;
	}
}

/* n_plus_x_recs per bin: how many (k+x)-mer records the reference's stage 2 expands each super-k-mer into (kb_collector.cpp:83-100,
 * kb_collector.h:72-118) — the third sum a CKmerBinCollector keeps, which stage 2 sizes its arrays with. One thread per super-k-mer walks
 * its k-mers comparing the first four symbols of the k-mer with those of its reverse complement. */
__global__ void __launch_bounds__(256) k_s1_bin_plus_x(const int8_t *__restrict__ codes, const u64 *__restrict__ sk_pos, const u32 *__restrict__ sk_len,
                                                        const u32 *__restrict__ sk_sig, u64 n_sk, u32 k, u32 max_x, u32 both_strands, const int *__restrict__ sig_to_bin,
                                                        u32 n_bins, u64 *__restrict__ bin_plus_x)
{
	__shared__ u32 s_px[S1_MAX_BINS];
	for (u32 b = threadIdx.x; b < n_bins; b += 256)
		s_px[b] = 0;
	__syncthreads();
	const u64 i0 = (u64)blockIdx.x * S1_SK_TILE;
	for (u32 j = threadIdx.x; j < (u32)S1_SK_TILE && max_x; j += 256) {
		const u64 i = i0 + j;
		if (i >= n_sk)
			break;
		const int b = sig_to_bin[sk_sig[i]];
		if (b < 0 || (u32)b >= n_bins)
			continue;
		const u32 n = sk_len[i];
		u32 total;
		if (!both_strands)
			total = 1 + (n - k) / (max_x + 1);
		else {
			const int8_t *q = codes + sk_pos[i];
			/* & 3: a code may carry S1_PIECE_MARK */
			u32 fwd = ((u32)(q[0] & 3) << 6) | ((u32)(q[1] & 3) << 4) | ((u32)(q[2] & 3) << 2) | (u32)(q[3] & 3);
			u32 rc = ((3u - (q[k - 1] & 3)) << 6) | ((3u - (q[k - 2] & 3)) << 4) | ((3u - (q[k - 3] & 3)) << 2) | (3u - (q[k - 4] & 3));
			u32 state = fwd < rc ? 0u : (rc < fwd ? 1u : 2u), run = 0;
			total = 0;
			for (u32 t = 0; t + k < n; ++t) {
				rc = (rc >> 2) | ((3u - (q[k + t] & 3)) << 6);
				fwd = ((fwd << 2) & 0xFFu) | (u32)(q[4 + t] & 3);
				const u32 st = fwd < rc ? 0u : (rc < fwd ? 1u : 2u);
				if (st == state) {
					if (st == 2)
						++total;
					else
						++run;
				} else {
					state = st;
					total += 1 + run / (max_x + 1);
					run = 0;
				}
			}
			total += 1 + run / (max_x + 1);
		}
		atomicAdd(&s_px[b], total);
	}
	__syncthreads();
	for (u32 b = threadIdx.x; b < n_bins; b += 256)
		if (s_px[b])
			atomicAdd(&bin_plus_x[b], (u64)s_px[b]);
}

/* ------------------------------------------------------------------------------------------------ emit through a sort (alternative to k_s1_emit)
 * k_s1_emit claims output space with atomics: 512 partial-line write streams, ~2 records per bin and tile, 2.9 of the 6.1 ms of a 300 M symbol
 * part (profiles/r02/s1_bench_v3_*). The alternative orders the super-k-mers by bin first — k_s1_sort_keys makes 8-byte records
 * (index << 16) | bin, the EXISTING k_onesweep sorts them by their two low bytes (stable: a bin keeps read order, like the reference with one
 * splitter thread) — and k_s1_emit_sorted gives every record its final position from one exclusive scan of the record sizes in that order:
 *   position = bin_base[bin] + (bytes of all records before it in sorted order) - (bytes of all bins before its bin),
 * so consecutive threads write consecutive bytes. EMULATION-TESTED ONLY, not the default (S1PartParams::sorted_emit): to be measured first. */
__global__ void __launch_bounds__(256) k_s1_sort_keys(const u32 *__restrict__ sk_sig, u64 n_sk, const int *__restrict__ sig_to_bin, u32 n_bins, u64 *__restrict__ keys, u32 *err)
{
	const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_sk)
		return;
	const int b = sig_to_bin[sk_sig[i]];
	if (b < 0 || (u32)b >= n_bins) {
		atomicOr(err, KERR_CORRUPT);
		keys[i] = (i << 16) | 0xFFFFu;
	} else
		keys[i] = (i << 16) | (u64)b;
}

/* sorted: the keys in bin-major order; cum_bytes[b] = bytes of the records of all bins < b (no alignment). status: one zeroed u64 per tile. */
__global__ void __launch_bounds__(S1_BLOCK) k_s1_emit_sorted(const u64 *__restrict__ sorted, u64 n_sk, const int8_t *__restrict__ codes, const u64 *__restrict__ sk_pos,
                                                              const u32 *__restrict__ sk_len, u32 k, u32 n_bins, const u64 *__restrict__ bin_base,
                                                              const u64 *__restrict__ pack_base, const u64 *__restrict__ cum_bytes, u64 *status, u32 *ticket_ctr,
                                                              uint8_t *__restrict__ out, u64 *__restrict__ pack_start, u32 *err)
{
	__shared__ u32 s_tmp[S1_BLOCK / 64 + 1];
	__shared__ u64 s_carry;
	__shared__ u32 s_ticket;
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0)
		s_ticket = atomicAdd(ticket_ctr, 1u);
	__syncthreads();
	const u32 tile = s_ticket;
	const u32 num_tiles = (u32)((n_sk + S1_TILE - 1) / S1_TILE);
	if (tile >= num_tiles)
		return;
	const u64 j0 = (u64)tile * S1_TILE + (u64)tid * S1_PER; /* this thread's S1_PER consecutive records of the sorted order */
	u64 idx[S1_PER];
	u32 bin[S1_PER], len[S1_PER], bytes[S1_PER], mine = 0;
#pragma unroll
	for (int t = 0; t < S1_PER; ++t) {
		bytes[t] = 0;
		if (j0 + t < n_sk) {
			const u64 key = sorted[j0 + t];
			idx[t] = key >> 16;
			bin[t] = (u32)(key & 0xFFFFu);
			if (bin[t] >= n_bins) { /* k_s1_sort_keys marked a signature without a bin (and raised the error) */
				atomicOr(err, KERR_CORRUPT);
				continue;
			}
			len[t] = sk_len[idx[t]];
			bytes[t] = 1u + (len[t] + 3u) / 4u;
			mine += bytes[t];
		}
	}
	u32 tile_bytes;
	const u32 before = block_excl_sum<S1_BLOCK / 64, u32>(mine, s_tmp, tile_bytes);
	if (wave == 0) {
		const u64 excl = lookback64(status, tile, (u64)tile_bytes, lane, err, KERR_WATCHDOG | KERR_AT_STAGE1);
		if (lane == 0)
			s_carry = excl;
	}
	__syncthreads();
	u64 g = s_carry + before; /* bytes of all records before this one in sorted order */
#pragma unroll
	for (int t = 0; t < S1_PER; ++t) {
		if (!bytes[t])
			continue;
		const u32 b = bin[t];
		const u64 rel = g - cum_bytes[b]; /* offset inside the bin's stream */
		g += bytes[t];
		const u64 m = (rel + S1_PACK_BYTES - 1) / S1_PACK_BYTES;
		if (m * S1_PACK_BYTES < rel + bytes[t]) /* the record that covers the m-th multiple of the pack size starts pack m */
			pack_start[pack_base[b] + m] = rel;
		uint8_t *dst = out + bin_base[b] + rel;
		const int8_t *src = codes + sk_pos[idx[t]];
		dst[0] = (uint8_t)(len[t] - k);
		u32 q = 0;
		for (; 4u * q + 8u <= len[t]; q += 2) {
			u64 w8;
			__builtin_memcpy(&w8, src + 4u * q, 8);
			const u32 lo = (u32)w8, hi = (u32)(w8 >> 32);
			dst[1 + q] = (uint8_t)(((lo & 3u) << 6) | (((lo >> 8) & 3u) << 4) | (((lo >> 16) & 3u) << 2) | ((lo >> 24) & 3u));
			dst[2 + q] = (uint8_t)(((hi & 3u) << 6) | (((hi >> 8) & 3u) << 4) | (((hi >> 16) & 3u) << 2) | ((hi >> 24) & 3u));
		}
		for (; q < (len[t] + 3u) / 4u; ++q) {
			u32 v = 0;
#pragma unroll
			for (u32 u = 0; u < 4; ++u) {
				const u32 p = 4 * q + u;
				v = (v << 2) | (p < len[t] ? (u32)(src[p] & 3) : 0u);
			}
			dst[1 + q] = (uint8_t)v;
		}
	}
}

#endif
