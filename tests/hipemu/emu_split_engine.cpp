/*
 * tests/hipemu/emu_split_engine.cpp — TEST INFRASTRUCTURE: a KmcSplitEngine (kmc_amd/host/split_engine.h) whose split_part() runs
 * kmc_amd/csrc/stage1_chain.h — the launch sequence kmc_hip_split_part uses on the GPU — with host memory and the CPU emulation of the
 * kernels (tests/hipemu) as its backend. Linked into oracle/_ref/kmc_emu_s1 (reference pipeline + kb_splitter_plugin.h): the kernel chain
 * inside the real KMC, checked by the database it leads to (tests/test_stage1_plugin.py). Emulated "LDS" is static storage, so calls are
 * serialised. $KMC_EMU_SK_GUESS_DIV sets the first guess of the super-k-mer count (a huge value forces the second cut).
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../kmc_amd/csrc/kernels.hip.h"
#include "../../kmc_amd/csrc/stage1_kernels.hip.h"

struct EmuBackend {
	std::vector<std::unique_ptr<uint8_t[]>> blocks;
	void *alloc_uninit(size_t bytes) { return alloc(bytes); }
	void *alloc(size_t bytes)
	{
		blocks.emplace_back(new uint8_t[bytes + 64]());
		return blocks.back().get();
	}
	void zero(void *p, size_t bytes) { memset(p, 0, bytes); }
	bool d2h(void *dst, const void *src, size_t bytes)
	{
		memcpy(dst, src, bytes);
		return true;
	}
	void h2d(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
	u64 *sort_by_low16(u64 *keys, u64 *, u64 n) /* the radix passes themselves are tested elsewhere (tests/test_kernels_emulated.py) */
	{
		std::stable_sort(keys, keys + n, [](u64 a, u64 b) { return (a & 0xFFFFu) < (b & 0xFFFFu); });
		return keys;
	}
	void release() { blocks.clear(); }
};
#define S1_LAUNCH(B, be, kernel, grid, block, ...) hipemu::launch(grid, block, 0, [&] { kernel(__VA_ARGS__); })
#include "../../kmc_amd/csrc/stage1_chain.h"
#include "../../kmc_amd/host/split_engine.h"

namespace {
std::mutex g_launch_mtx;

struct EmuSplitEngine : KmcSplitEngine {
	KmcSplitParams P;
	std::string err_msg;
	EmuBackend be;
	S1PartResult R;

	explicit EmuSplitEngine(const KmcSplitParams &p) : P(p) {}
	std::string last_error() override { return err_msg; }

	int split_part(const uint8_t *text, uint64_t size, bool long_read, KmcSplitResult &out) override
	{
		std::lock_guard<std::mutex> lck(g_launch_mtx);
		be.release();
		S1PartParams sp;
		sp.k = P.kmer_len;
		sp.m = P.signature_len;
		sp.n_bins = P.n_bins;
		sp.max_x = P.max_x;
		sp.both_strands = P.both_strands ? 1u : 0u;
		sp.lines_per_record = P.file_type == 1 ? 4u : 2u;
		sp.line_cap = P.line_cap;
		sp.d_sig_to_bin = P.sig_to_bin;
		sp.sorted_emit = getenv("KMC_HIP_S1_SORTED_EMIT") != nullptr;
		if (const char *g = getenv("KMC_EMU_SK_GUESS_DIV"))
			sp.sk_guess_div = strtoull(g, nullptr, 10);
		u64 long_reads = 0;
		std::vector<uint8_t> aligned; /* the kernels load 16 aligned bytes at a time */
		if (long_read) {
			const u64 skip = s1_long_read_title(text, size, (u32)P.file_type, long_reads);
			aligned.assign(text + skip, text + size);
			aligned.resize(aligned.size() + 16);
			text = aligned.data();
			size -= skip;
			sp.lines_per_record = 0;
		}
		const int rc = s1_split_part(be, text, size, size && text[size - 1] == '\n', sp, R);
		if (long_read)
			R.n_reads = long_reads;
		if (rc == S1_CHAIN_UNCOVERED)
			return KMC_SPLIT_UNCOVERED;
		if (rc != S1_CHAIN_OK) {
			err_msg = "stage-1 chain failed: code " + std::to_string(rc) + ", device error word " + std::to_string(R.device_error);
			return rc;
		}
		out.recs = R.d_recs;
		out.bin_off = (const uint64_t *)R.bin_off.data();
		out.bin_bytes = (const uint64_t *)R.bin_bytes.data();
		out.bin_kmers = (const uint64_t *)R.bin_kmers.data();
		out.bin_superkmers = (const uint64_t *)R.bin_sk.data();
		out.bin_plus_x = (const uint64_t *)R.bin_plus_x.data();
		out.n_reads = R.n_reads;
		return 0;
	}
};
} // namespace

KmcSplitEngine *kmc_make_split_engine(const KmcSplitParams &params, int, int) { return new EmuSplitEngine(params); }
