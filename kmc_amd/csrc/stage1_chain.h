/*
 * kmc_amd/csrc/stage1_chain.h — one part of input text through the stage-1 kernels, written ONCE for two backends:
 *   - HipBackend (kmc_hip.hip, kmc_hip_split_part): hipMalloc / hipLaunchKernelGGL / hipMemcpy on a stream
 *   - EmuBackend (tests/hipemu/emu_split_engine.cpp): host memory and the CPU emulation of the kernels
 * so that the order of launches, the grid and buffer sizes, the synchronisation points, the retry of the cutting kernel and the assembly of
 * the result are exercised inside the real KMC pipeline on the CPU (oracle/_ref/kmc_emu_s1, tests/test_stage1_plugin.py) before a GPU sees
 * them. Does for one part what CSplitter::ProcessReads + the n_bins CKmerBinCollectors do (splitter.cpp:557-672, kb_collector.cpp:34-106),
 * up to the bin-part buffers.
 *
 * A backend B provides:
 *   void *alloc(size_t bytes)                  zeroed device memory, owned by the backend until release()
 *   void *alloc_uninit(size_t bytes)           the same without the zeroing: for arrays that are written before they are read (super-k-mer lists, sort keys)
 *   void  zero(void *p, size_t bytes)          (stream-ordered)
 *   bool  d2h(void *dst, const void *src, size_t bytes)   copies and waits for everything launched before; false = device failure
 *   void  h2d(void *dst, const void *src, size_t bytes)   (the source may be reused when it returns)
 *   S1_LAUNCH(B, be, kernel, grid, block, args...)         launch `kernel` (a macro per backend: the emulator calls kernels as functions)
 *   u64  *sort_by_low16(u64 *keys, u64 *tmp, u64 n)         stable sort of 8-byte records by their two low bytes; returns where the result is
 *                                                           (the HIP backend runs the library's own radix passes, test backends sort on the host)
 */
#ifndef KMC_AMD_STAGE1_CHAIN_H
#define KMC_AMD_STAGE1_CHAIN_H

#include <vector>

#include "stage1_kernels.hip.h"

struct S1PartParams {
	u32 k, m, n_bins, max_x, both_strands, lines_per_record; /* lines_per_record: 2 = FASTA, 4 = FASTQ, 0 = the symbols of a long-read part (the caller
	                                                           * has taken the title off: d_text starts at its end of line, GetSeqLongRead splitter.cpp:70-86) */
	u64 line_cap;                                             /* mem_part_pmm_reads */
	const int *d_sig_to_bin;                                  /* device: 4^m + 1 entries */
	u64 sk_guess_div = 8;                                     /* first guess of the number of super-k-mers: symbols / this + 4096 */
	bool sorted_emit = false;                                 /* records through a sort by bin (k_s1_emit_sorted) instead of k_s1_emit: bins in read order */
};
struct S1PartResult {
	const uint8_t *d_recs = nullptr; /* device: bin b's records at d_recs + bin_off[b], bin_bytes[b] of them */
	u64 recs_bytes = 0;
	std::vector<u64> bin_off, bin_bytes, bin_sk, bin_kmers, bin_plus_x;
	u64 n_reads = 0, n_symbols = 0, n_superkmers = 0;
	u32 device_error = 0;
};
enum { S1_CHAIN_OK = 0, S1_CHAIN_UNCOVERED = 1, S1_CHAIN_DEVICE_ERROR = -1, S1_CHAIN_BACKEND_FAILURE = -2 };

/* A long-read part (ReadType::long_read, queues.h:40; CSplitter::GetSeqLongRead, splitter.cpp:70-86): if it starts with the format's marker it carries the
 * read's title — one read counted, and the symbols start AT the title's end of line (that byte is a symbol like every other: an invalid one). Returns the
 * number of bytes to take off the front; the rest goes through s1_split_part with lines_per_record = 0. */
static inline u64 s1_long_read_title(const uint8_t *text, u64 size, u32 file_type /* 0 FASTA, 1 FASTQ */, u64 &n_reads)
{
	n_reads = 0;
	if (!size || text[0] != (file_type == 1 ? '@' : '>'))
		return 0;
	n_reads = 1;
	u64 p = 0;
	while (p < size && text[p] != '\n' && text[p] != '\r')
		++p;
	return p;
}

template <class B> int s1_split_part(B &be, const uint8_t *d_text, u64 size, bool text_ends_with_newline, const S1PartParams &P, S1PartResult &R)
{
	const u32 nb = P.n_bins, lpr = P.lines_per_record;
	R = S1PartResult();
	R.bin_off.assign(nb, 0);
	R.bin_bytes.assign(nb, 0);
	R.bin_sk.assign(nb, 0);
	R.bin_kmers.assign(nb, 0);
	R.bin_plus_x.assign(nb, 0);
	if (!size)
		return S1_CHAIN_OK;
	/* small block: [0] '\\n' count | [1] code bytes | [2] super-k-mers | [3] lo: ticket, hi: error word | [4] != 0: some code carries S1_PIECE_MARK */
	u64 *d_small = (u64 *)be.alloc(64);
	u32 *d_ticket = (u32 *)(d_small + 3), *d_err = d_ticket + 1;
	u64 *d_has_marks = d_small + 4;
	u64 small[4];
	/* pieces of an over-long line start every `stride` symbols (S1_PIECE_MARK); k_s1_cut takes at most one mark per workgroup window */
	if (P.line_cap < (u64)P.k + S1_WG_TILE + 2)
		return S1_CHAIN_UNCOVERED;
	const u64 stride = P.line_cap - P.k + 1;
	/* ---- text -> codes. A line of real reads is tens of bytes; text with more line ends than size / 4 is not taken. */
	const u64 nl_cap = lpr ? size / 4 + 1024 : 1;
	const u32 tiles = (u32)((size + S1_TXT_TILE - 1) / S1_TXT_TILE);
	int8_t *d_codes = (int8_t *)be.alloc(size + 16);
	u64 *d_nl = (u64 *)be.alloc(nl_cap * 8);
	u64 *d_seq_start = (u64 *)be.alloc_uninit((nl_cap / (lpr ? lpr : 1) + 2) * 8);
	u64 *d_status = (u64 *)be.alloc((size_t)tiles * 16);
	S1_LAUNCH(B, be, k_s1_text_to_codes, dim3(tiles), dim3(S1_BLOCK), d_text, size, lpr, d_status, d_status + tiles, d_ticket, d_codes, d_nl, nl_cap, d_seq_start, d_small,
	          d_err);
	if (!be.d2h(small, d_small, sizeof small))
		return S1_CHAIN_BACKEND_FAILURE;
	u32 err = (u32)(small[3] >> 32);
	if (err & (S1_TEXT_BAD | KERR_CAPACITY))
		return S1_CHAIN_UNCOVERED;
	if (err) {
		R.device_error = err;
		return S1_CHAIN_DEVICE_ERROR;
	}
	const u64 n_lines = small[0], n = small[1];
	R.n_symbols = n;
	if (lpr) {
		const u64 lines = n_lines + (text_ends_with_newline ? 0 : 1); /* titles in the part: every lpr-th line, an unterminated last line included */
		R.n_reads = (lines + lpr - 1) / lpr;
		S1_LAUNCH(B, be, k_s1_check_records, dim3((u32)((n_lines / lpr + 1 + 255) / 256)), dim3(256), d_text, size, (const u64 *)d_nl, n_lines, lpr, P.line_cap, stride,
		          (const u64 *)d_seq_start, d_codes, d_has_marks, d_err);
	} else if (n > stride) /* n_reads of a long-read part: the caller knows whether it took a title off */
		S1_LAUNCH(B, be, k_s1_mark_raw, dim3(1), dim3(256), d_codes, n, stride, d_has_marks);
	/* ---- codes -> super-k-mers. Their number is only known afterwards: a guess, and a second cut with the exact number when it was short. */
	u64 n_sk = 0, cap = n / P.sk_guess_div + 4096;
	u64 *d_pos = nullptr;
	u32 *d_len = nullptr, *d_sig = nullptr;
	if (n) {
		const u32 ct = (u32)s1_cut_tiles(n);
		u64 *d_cstat = (u64 *)be.alloc((size_t)ct * 16);
		for (int attempt = 0; attempt < 2; ++attempt) {
			d_pos = (u64 *)be.alloc_uninit(cap * 8);
			d_len = (u32 *)be.alloc_uninit(cap * 4);
			d_sig = (u32 *)be.alloc_uninit(cap * 4);
			if (attempt) {
				be.zero(d_cstat, (size_t)ct * 16);
				be.zero(d_small + 2, 8);
			}
			be.zero(d_ticket, 4);
			S1_LAUNCH(B, be, (k_s1_cut<true>), dim3(ct), dim3(S1_BLOCK), (const u32 *)nullptr, (const int8_t *)d_codes, P.m, n, P.k, d_cstat, d_cstat + ct, d_ticket, d_pos,
			          d_len, d_sig, cap, d_small + 2, (const u64 *)d_has_marks, d_err);
			if (!be.d2h(small, d_small, sizeof small))
				return S1_CHAIN_BACKEND_FAILURE;
			err = (u32)(small[3] >> 32);
			n_sk = small[2];
			if (n_sk <= cap || (err & S1_TEXT_BAD))
				break;
			cap = n_sk; /* KERR_CAPACITY was raised by the short attempt: cleared with the word below */
			be.zero(d_err, 4);
			err &= ~KERR_CAPACITY;
			if (err)
				break;
		}
	} else {
		if (!be.d2h(small, d_small, sizeof small))
			return S1_CHAIN_BACKEND_FAILURE;
		err = (u32)(small[3] >> 32);
	}
	if (err & S1_TEXT_BAD)
		return S1_CHAIN_UNCOVERED; /* the record check ran beside the cut */
	if (err) {
		R.device_error = err;
		return S1_CHAIN_DEVICE_ERROR;
	}
	R.n_superkmers = n_sk;
	/* ---- per-bin sums, layout, records */
	u64 *d_tot = (u64 *)be.alloc((size_t)4 * nb * 8); /* bytes | super-k-mers | k-mers | n_plus_x_recs */
	u64 *d_lay = (u64 *)be.alloc((size_t)(3 * nb + 2) * 8);
	const u32 sk_tiles = (u32)((n_sk + S1_SK_TILE - 1) / S1_SK_TILE);
	if (sk_tiles) {
		S1_LAUNCH(B, be, k_s1_bin_totals, dim3(sk_tiles), dim3(256), (const u32 *)d_len, (const u32 *)d_sig, n_sk, P.k, P.d_sig_to_bin, nb, d_tot, d_tot + nb, d_tot + 2 * nb,
		          d_err);
		S1_LAUNCH(B, be, k_s1_bin_plus_x, dim3(sk_tiles), dim3(256), (const int8_t *)d_codes, (const u64 *)d_pos, (const u32 *)d_len, (const u32 *)d_sig, n_sk, P.k, P.max_x,
		          P.both_strands, P.d_sig_to_bin, nb, d_tot + 3 * nb);
	}
	S1_LAUNCH(B, be, k_s1_bin_layout, dim3(1), dim3(256), (const u64 *)d_tot, nb, d_lay, d_lay + nb + 1, d_lay + 2 * nb + 2, (u64 *)nullptr);
	std::vector<u64> lay(2 * (size_t)nb + 2);
	if (!be.d2h(lay.data(), d_lay, lay.size() * 8) || !be.d2h(small, d_small, sizeof small))
		return S1_CHAIN_BACKEND_FAILURE;
	err = (u32)(small[3] >> 32);
	if (err) { /* a signature the map does not know (KERR_CORRUPT from k_s1_bin_totals): nothing is emitted */
		R.device_error = err;
		return S1_CHAIN_DEVICE_ERROR;
	}
	const u64 recs_bytes = lay[nb], n_packs = lay[2 * (size_t)nb + 1];
	uint8_t *d_recs = (uint8_t *)be.alloc(recs_bytes + 16);
	u64 *d_packs = (u64 *)be.alloc((n_packs + 1) * 8);
	S1_LAUNCH(B, be, k_s1_bin_layout, dim3(1), dim3(256), (const u64 *)d_tot, nb, d_lay, d_lay + nb + 1, d_lay + 2 * nb + 2, d_packs);
	if (sk_tiles && !P.sorted_emit)
		S1_LAUNCH(B, be, k_s1_emit, dim3(sk_tiles), dim3(256), (const int8_t *)d_codes, (const u64 *)d_pos, (const u32 *)d_len, (const u32 *)d_sig, n_sk, P.k, P.d_sig_to_bin, nb,
		          (const u64 *)d_lay, (const u64 *)(d_lay + nb + 1), d_lay + 2 * nb + 2, d_recs, d_packs);
	if (sk_tiles && P.sorted_emit) {
		/* bytes of all bins before each bin, from the sums already on the device */
		std::vector<u64> cum(nb + 1, 0), bytes_now(nb);
		if (!be.d2h(bytes_now.data(), d_tot, (size_t)nb * 8))
			return S1_CHAIN_BACKEND_FAILURE;
		for (u32 b = 0; b < nb; ++b)
			cum[b + 1] = cum[b] + bytes_now[b];
		u64 *d_cum = (u64 *)be.alloc((size_t)(nb + 1) * 8);
		be.h2d(d_cum, cum.data(), (size_t)(nb + 1) * 8);
		u64 *d_keys = (u64 *)be.alloc_uninit(n_sk * 8), *d_ktmp = (u64 *)be.alloc_uninit(n_sk * 8);
		S1_LAUNCH(B, be, k_s1_sort_keys, dim3((u32)((n_sk + 255) / 256)), dim3(256), (const u32 *)d_sig, n_sk, P.d_sig_to_bin, nb, d_keys, d_err);
		const u64 *d_sorted = be.sort_by_low16(d_keys, d_ktmp, n_sk);
		const u32 et = (u32)((n_sk + S1_TILE - 1) / S1_TILE);
		u64 *d_estat = (u64 *)be.alloc((size_t)et * 8);
		be.zero(d_ticket, 4);
		S1_LAUNCH(B, be, k_s1_emit_sorted, dim3(et), dim3(S1_BLOCK), d_sorted, n_sk, (const int8_t *)d_codes, (const u64 *)d_pos, (const u32 *)d_len, P.k, nb, (const u64 *)d_lay,
		          (const u64 *)(d_lay + nb + 1), (const u64 *)d_cum, d_estat, d_ticket, d_recs, d_packs, d_err);
	}
	std::vector<u64> tot(4 * (size_t)nb);
	if (!be.d2h(tot.data(), d_tot, tot.size() * 8) || !be.d2h(small, d_small, sizeof small))
		return S1_CHAIN_BACKEND_FAILURE;
	err = (u32)(small[3] >> 32);
	if (err) {
		R.device_error = err;
		return S1_CHAIN_DEVICE_ERROR;
	}
	for (u32 b = 0; b < nb; ++b) {
		R.bin_off[b] = lay[b];
		R.bin_bytes[b] = tot[b];
		R.bin_sk[b] = tot[nb + b];
		R.bin_kmers[b] = tot[2 * (size_t)nb + b];
		R.bin_plus_x[b] = tot[3 * (size_t)nb + b];
	}
	R.d_recs = d_recs;
	R.recs_bytes = recs_bytes;
	return S1_CHAIN_OK;
}

#endif
