/*
 * kmc_amd/csrc/stage1_kernels.hip.h — FIRST kernels of KMC's STAGE 1 on gfx950 (groundwork, not yet a drop-in:
 * no bin scatter, no FASTQ parsing, reachable only through the test hook kmc_hip_debug_split_reads).
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
 *   k_s1_signatures : codes -> signature per k-mer position (0xFFFFFFFF where no valid k-mer starts)
 *   k_s1_cut        : signatures -> the super-k-mers in position order: first symbol, length in symbols, signature
 * Both are tile-parallel; k_s1_cut carries "where did the current run start" and "how many super-k-mers so far" across tiles with two
 * decoupled look-backs (max and sum) over 64-bit status words.
 */
#ifndef KMC_AMD_STAGE1_KERNELS_HIP_H
#define KMC_AMD_STAGE1_KERNELS_HIP_H

#include "kernels.hip.h"

constexpr int S1_BLOCK = 256, S1_PER = 4, S1_TILE = S1_BLOCK * S1_PER; /* positions per workgroup */
constexpr int S1_MAX_K = 256;
constexpr u32 S1_NOSIG = 0xFFFFFFFFu;

/* norm[] (4^m uint32) lives in global memory: 1 MB at m = 9, L2-resident, one gather per position */
__global__ void __launch_bounds__(S1_BLOCK) k_s1_signatures(const int8_t *__restrict__ codes, u64 n, u32 k, u32 m, const u32 *__restrict__ norm,
                                                             u32 *__restrict__ sig)
{
	__shared__ int8_t s_c[S1_TILE + S1_MAX_K];     /* symbols of the tile + the k - 1 after it */
	__shared__ u32 s_mm[S1_TILE + S1_MAX_K];       /* norm of the m-mer at each position */
	__shared__ u32 s_bad[S1_TILE + S1_MAX_K + 1];  /* exclusive prefix count of invalid symbols */
	__shared__ u32 s_tmp[S1_BLOCK / 64 + 1];
	const u32 tid = threadIdx.x;
	const u64 t0 = (u64)blockIdx.x * S1_TILE;
	const u32 span = S1_TILE + k - 1; /* symbols this tile looks at */
	for (u32 i = tid; i < span; i += S1_BLOCK) {
		const u64 p = t0 + i;
		s_c[i] = p < n ? codes[p] : (int8_t)-1;
	}
	__syncthreads();
	/* prefix count of invalid symbols over the span: thread t owns ceil(span / 256) consecutive symbols */
	{
		const u32 per = (span + S1_BLOCK - 1) / S1_BLOCK, lo = tid * per;
		u32 c = 0;
		for (u32 j = 0; j < per; ++j)
			if (lo + j < span && s_c[lo + j] < 0)
				++c;
		u32 total;
		u32 run = block_excl_sum<S1_BLOCK / 64, u32>(c, s_tmp, total);
		for (u32 j = 0; j < per; ++j)
			if (lo + j < span) {
				s_bad[lo + j] = run;
				run += s_c[lo + j] < 0 ? 1u : 0u;
			}
		if (tid == S1_BLOCK - 1)
			s_bad[span] = total;
	}
	/* norm of every m-mer that starts in the tile or in the k - m positions after it */
	const u32 n_mm = S1_TILE + k - m;
	for (u32 i = tid; i < n_mm; i += S1_BLOCK) {
		u32 x = 0;
		bool ok = true;
		for (u32 j = 0; j < m; ++j) {
			const int8_t c = s_c[i + j];
			ok = ok && c >= 0;
			x = (x << 2) | (u32)(c & 3);
		}
		s_mm[i] = ok ? norm[x] : S1_NOSIG;
	}
	__syncthreads();
	for (u32 i = tid; i < (u32)S1_TILE; i += S1_BLOCK) {
		const u64 q = t0 + i;
		if (q >= n)
			break;
		u32 s = S1_NOSIG;
		if (q + k <= n && s_bad[i + k] == s_bad[i]) { /* a valid k-mer starts here */
			const u32 w = k - m + 1;
			for (u32 j = 0; j < w; ++j) {
				const u32 v = s_mm[i + j];
				s = v < s ? v : s;
			}
		}
		sig[q] = s;
	}
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
	u32 spins = 0;
	while (true) {
		const long long t = tbase - (long long)lane;
		const u64 v = t >= 0 ? ld_agent(&status[t]) : ST64_PREFIX;
		const u64 flag = v & ~ST64_MASK;
		const u64 m_pref = __ballot(flag == ST64_PREFIX);
		const u64 m_zero = __ballot(flag == 0);
		const int pl = m_pref ? (__ffsll(m_pref) - 1) : 64;
		const u64 need = pl < 63 ? ((2ull << pl) - 1) : ~0ull;
		if (m_zero & need) {
			if (++spins > SPIN_LIMIT || (spins % 1024 == 0 && (ld_agent(err) & KERR_WATCHDOG))) {
				if (lane == 0)
					atomicOr(err, KERR_WATCHDOG);
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

/* status_last / status_cnt: one zeroed u64 per tile each. sk_* receive the super-k-mers in position order; *n_sk their number (written by
 * the last tile). sk_cap bounds the writes (KERR_CAPACITY beyond it). */
__global__ void __launch_bounds__(S1_BLOCK) k_s1_cut(const u32 *__restrict__ sig, u64 n, u32 k, u64 *status_last, u64 *status_cnt, u32 *ticket_ctr,
                                                      u64 *__restrict__ sk_pos, u32 *__restrict__ sk_len, u32 *__restrict__ sk_sig, u64 sk_cap, u64 *n_sk,
                                                      u32 *err)
{
	__shared__ u64 s_tmp64[S1_BLOCK / 64 + 1];
	__shared__ u32 s_tmp32[S1_BLOCK / 64 + 1];
	__shared__ u64 s_carry_last1, s_carry_cnt;
	__shared__ u32 s_ticket;
	const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (tid == 0)
		s_ticket = atomicAdd(ticket_ctr, 1u);
	__syncthreads();
	const u32 tile = s_ticket;
	const u32 num_tiles = (u32)((n + S1_TILE - 1) / S1_TILE);
	if (tile >= num_tiles)
		return;
	const u64 t0 = (u64)tile * S1_TILE + (u64)tid * S1_PER; /* this thread's S1_PER consecutive positions */
	u32 s[S1_PER + 2];                                        /* sig[t0 - 1 .. t0 + S1_PER] */
#pragma unroll
	for (int j = 0; j < S1_PER + 2; ++j) {
		const long long q = (long long)t0 - 1 + j;
		s[j] = (q >= 0 && (u64)q < n) ? sig[q] : S1_NOSIG;
	}
	/* run starts inside this thread's positions: valid, and the k-mer before is invalid or has another signature */
	u64 last1 = 0; /* position + 1 of the thread's last run start */
#pragma unroll
	for (int j = 0; j < S1_PER; ++j)
		if (s[j + 1] != S1_NOSIG && s[j] != s[j + 1])
			last1 = t0 + j + 1;
	const u64 before1 = block_excl_max<S1_BLOCK / 64, u64>(last1, s_tmp64); /* last run start + 1 before this thread, inside the tile */
	/* the tile's own last run start: the maximum over the threads */
	u64 tile_last1 = wave_incl_max<u64>(last1, lane);
	if (lane == 63)
		s_tmp64[wave] = tile_last1;
	__syncthreads();
	if (wave == 0) {
		u64 w = 0;
#pragma unroll
		for (int i = 0; i < S1_BLOCK / 64; ++i)
			w = s_tmp64[i] > w ? s_tmp64[i] : w;
		const u64 carry = lookback64_last(status_last, tile, w, lane, err);
		if (lane == 0)
			s_carry_last1 = carry;
	}
	__syncthreads();
	const u64 carry_last1 = s_carry_last1;
	/* ends: a super-k-mer ends at q if q is valid and the next k-mer is invalid / has another signature / q is the 256th k-mer of its piece */
	u64 cur1 = before1 ? before1 : carry_last1;
	u32 n_end = 0, end_bits = 0;
	u32 piece[S1_PER];
#pragma unroll
	for (int j = 0; j < S1_PER; ++j) {
		const u64 q = t0 + j;
		piece[j] = 0;
		if (s[j + 1] != S1_NOSIG) {
			if (s[j] != s[j + 1])
				cur1 = q + 1; /* a run starts here */
			const u32 in_piece = (u32)((q - (cur1 - 1)) & 255u); /* k-mers of this piece before q */
			if (s[j + 2] != s[j + 1] || in_piece == 255u) {
				end_bits |= 1u << j;
				piece[j] = in_piece;
				++n_end;
			}
		}
	}
	u32 tile_ends;
	const u32 off = block_excl_sum<S1_BLOCK / 64, u32>(n_end, s_tmp32, tile_ends);
	if (wave == 0) {
		const u64 excl = lookback64(status_cnt, tile, (u64)tile_ends, lane, err, KERR_WATCHDOG);
		if (lane == 0) {
			s_carry_cnt = excl;
			if (tile == num_tiles - 1)
				*n_sk = excl + tile_ends;
		}
	}
	__syncthreads();
	u64 idx = s_carry_cnt + off;
#pragma unroll
	for (int j = 0; j < S1_PER; ++j)
		if (end_bits & (1u << j)) {
			if (idx < sk_cap) {
				sk_pos[idx] = t0 + j - piece[j];
				sk_len[idx] = k + piece[j];
				sk_sig[idx] = s[j + 1];
			} else
				atomicOr(err, KERR_CAPACITY);
			++idx;
		}
}

#endif
