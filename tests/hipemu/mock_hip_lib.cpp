/*
 * tests/hipemu/mock_hip_lib.cpp — TEST INFRASTRUCTURE: a stand-in for libkmc_hip.so on a box without a GPU, exporting the entry points
 * the product's loaders bind (kmc_amd/host/hip_loader.cpp, hip_split_loader.cpp). Behind them: the stage-2 ORACLE for a bin and the
 * EMULATED stage-1 kernel chain (kmc_amd/csrc/stage1_chain.h over tests/hipemu) for a part. Purpose: run the PRODUCT binaries
 * oracle/_ref/kmc_hip and kmc_hip_s1 — plug-ins, dlopen loaders, argument marshalling, slot handling — end to end on the CPU and compare
 * their database with the reference's (tests/test_stage1_plugin.py). It is only ever loaded through an explicit KMC_HIP_LIB=<this file>;
 * nothing in the product looks for it, and the GPU library is what every -m gpu test and bench.py load.
 */
#include <atomic>
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

struct MockBackend {
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
};
#define S1_LAUNCH(B, be, kernel, grid, block, ...) hipemu::launch(grid, block, 0, [&] { kernel(__VA_ARGS__); })
#include "../../kmc_amd/csrc/stage1_chain.h"
#include "../../include/kmc_hip.h"
extern "C" {
#include "../../oracle/stage2_oracle.h"
}

struct kmc_hip_ctx {
	int n_dev = 1;
	std::vector<int32_t> sig_map;
	uint32_t sig_len = 0;
	struct Pending {
		uint64_t out_bytes = 0, stats[4] = {0, 0, 0, 0};
		int rc = 0;
		std::vector<uint64_t> grp_out_bytes, grp_stats; /* kmc_hip_process_bins_submit: per bin */
	} pending[64][16];
	std::mutex mtx;
};

namespace {
thread_local std::string g_err;
std::mutex g_emu_mtx; /* emulated "LDS" is static storage */
int fail(int code, const char *msg)
{
	g_err = msg;
	return code;
}
} // namespace

#define MOCK_API extern "C" __attribute__((visibility("default")))

MOCK_API int kmc_hip_init(const int *, int n_dev, kmc_hip_ctx **out)
{
	if (!out || n_dev < 1 || n_dev > 64)
		return fail(KMC_HIP_EINVAL, "mock kmc_hip_init: bad arguments");
	*out = new kmc_hip_ctx;
	(*out)->n_dev = n_dev;
	return 0;
}
MOCK_API void kmc_hip_destroy(kmc_hip_ctx *ctx) { delete ctx; }
MOCK_API const char *kmc_hip_last_error(kmc_hip_ctx *) { return g_err.c_str(); }
MOCK_API int kmc_hip_abi_version(void) { return KMC_HIP_ABI_VERSION; }
/* "pinned" buffers for the reader plug-in's pool (host_pool.h): plain memory here; counted, so that a test can see the pool at work */
static std::atomic<long long> g_host_allocs{0};
MOCK_API int kmc_hip_host_alloc(kmc_hip_ctx *, uint64_t bytes, void **p)
{
	*p = malloc(bytes ? bytes : 1);
	++g_host_allocs;
	if (getenv("KMC_HIP_VERBOSE") && g_host_allocs == 1)
		fprintf(stderr, "[mock] kmc_hip_host_alloc in use (reader plug-in's pinned pool)\n");
	return *p ? 0 : -3;
}
MOCK_API int kmc_hip_host_free(kmc_hip_ctx *, void *p)
{
	free(p);
	return 0;
}
MOCK_API int kmc_hip_backend_kind(void) { return 2; }
MOCK_API int kmc_hip_num_slots(void) { return 4; }
MOCK_API int kmc_hip_sort_records_into(kmc_hip_ctx *, int, const void *recs, void *dst, uint64_t n, uint32_t words, uint32_t)
{
	memcpy(dst, recs, (size_t)n * words * 8);
	oracle_sort((uint64_t *)dst, n, words);
	return 0;
}
MOCK_API int kmc_hip_process_bin_submit(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_bin_params *p, const uint8_t *sk, uint64_t size, uint64_t n_rec,
                                        const uint64_t *, uint64_t, uint8_t *out, uint64_t cap, uint64_t *lut)
{
	if (!ctx || dev < 0 || dev >= ctx->n_dev || slot < 0 || slot >= 4 || !p)
		return fail(KMC_HIP_EINVAL, "mock submit: bad arguments");
	oracle_params op;
	op.kmer_len = p->kmer_len;
	op.both_strands = p->both_strands;
	op.cutoff_min = p->cutoff_min;
	op.without_output = p->without_output;
	op.cutoff_max = p->cutoff_max;
	op.counter_max = p->counter_max;
	op.lut_prefix_len = p->lut_prefix_len;
	op.output_type = p->output_type;
	kmc_hip_ctx::Pending &pd = ctx->pending[dev][slot];
	pd.rc = oracle_process_bin(&op, sk, size, n_rec, out, cap, &pd.out_bytes, lut, pd.stats);
	return 0;
}
MOCK_API int kmc_hip_process_bin_wait(kmc_hip_ctx *ctx, int dev, int slot, uint64_t *out_bytes, uint64_t stats[4])
{
	kmc_hip_ctx::Pending &pd = ctx->pending[dev][slot];
	*out_bytes = pd.out_bytes;
	memcpy(stats, pd.stats, sizeof pd.stats);
	return pd.rc ? fail(KMC_HIP_ECORRUPT, "mock: oracle_process_bin failed") : 0;
}
MOCK_API int kmc_hip_process_bins_submit(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_bin_params *p, const kmc_hip_host_bin *bins, uint32_t n_bins)
{
	if (!ctx || dev < 0 || dev >= ctx->n_dev || slot < 0 || slot >= 4 || !p || !bins || n_bins < 1 || n_bins > 16)
		return fail(KMC_HIP_EINVAL, "mock submit (bins): bad arguments");
	kmc_hip_ctx::Pending &pd = ctx->pending[dev][slot];
	pd.grp_out_bytes.assign(n_bins, 0);
	pd.grp_stats.assign(4 * (size_t)n_bins, 0);
	int first_rc = 0;
	for (uint32_t i = 0; i < n_bins; ++i) { /* every bin on its own through the single-bin mock: the oracle per bin */
		if (int rc = kmc_hip_process_bin_submit(ctx, dev, slot, p, bins[i].superkmers, bins[i].size, bins[i].n_rec, bins[i].pack_bytes, bins[i].n_packs, bins[i].out_suffix,
		                                        bins[i].out_capacity, bins[i].lut))
			return rc;
		if (pd.rc && !first_rc)
			first_rc = pd.rc;
		pd.grp_out_bytes[i] = pd.out_bytes;
		memcpy(&pd.grp_stats[4 * i], pd.stats, sizeof pd.stats);
	}
	pd.rc = first_rc;
	return 0;
}
MOCK_API int kmc_hip_process_bins_wait(kmc_hip_ctx *ctx, int dev, int slot, uint64_t *out_bytes, uint64_t *stats)
{
	kmc_hip_ctx::Pending &pd = ctx->pending[dev][slot];
	memcpy(out_bytes, pd.grp_out_bytes.data(), pd.grp_out_bytes.size() * 8);
	memcpy(stats, pd.grp_stats.data(), pd.grp_stats.size() * 8);
	return pd.rc ? fail(KMC_HIP_ECORRUPT, "mock: oracle_process_bin failed") : 0;
}
MOCK_API int kmc_hip_split_set_map(kmc_hip_ctx *ctx, int dev, const int32_t *sig_to_bin, uint32_t signature_len)
{
	if (!ctx || dev < 0 || dev >= ctx->n_dev || !sig_to_bin || signature_len < 5 || signature_len > 11)
		return fail(KMC_HIP_EINVAL, "mock set_map: bad arguments");
	std::lock_guard<std::mutex> lck(ctx->mtx);
	ctx->sig_map.assign(sig_to_bin, sig_to_bin + ((size_t)1 << (2 * signature_len)) + 1);
	ctx->sig_len = signature_len;
	return 0;
}
MOCK_API int kmc_hip_split_part(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_split_params *p, const uint8_t *text, uint64_t size, uint8_t *recs,
                                uint64_t recs_capacity, uint64_t *recs_bytes, uint64_t *bin_off, uint64_t *bin_bytes, uint64_t *bin_kmers, uint64_t *bin_superkmers,
                                uint64_t *bin_plus_x, uint64_t *n_reads)
{
	if (!ctx || dev < 0 || dev >= ctx->n_dev || slot < 0 || slot >= 4 || !p || !recs || !recs_bytes)
		return fail(KMC_HIP_EINVAL, "mock split_part: bad arguments");
	if (ctx->sig_len != p->signature_len)
		return fail(KMC_HIP_EINVAL, "mock split_part: kmc_hip_split_set_map was not called for this signature length");
	std::lock_guard<std::mutex> lck(g_emu_mtx);
	MockBackend be;
	S1PartParams sp;
	sp.k = p->kmer_len;
	sp.m = p->signature_len;
	sp.n_bins = p->n_bins;
	sp.max_x = p->max_x;
	sp.both_strands = p->both_strands ? 1u : 0u;
	sp.lines_per_record = p->file_type == 1 ? 4u : 2u;
	sp.line_cap = p->line_cap;
	sp.d_sig_to_bin = ctx->sig_map.data();
	sp.sorted_emit = getenv("KMC_HIP_S1_SORTED_EMIT") != nullptr;
	S1PartResult R;
	u64 long_reads = 0;
	std::vector<uint8_t> aligned; /* the kernels load 16 aligned bytes at a time */
	if (p->part_kind == 1) {
		const u64 skip = s1_long_read_title(text, size, p->file_type, long_reads);
		aligned.assign(text + skip, text + size);
		aligned.resize(aligned.size() + 16);
		text = aligned.data();
		size -= skip;
		sp.lines_per_record = 0;
	}
	const int rc = s1_split_part(be, text, size, size && text[size - 1] == '\n', sp, R);
	if (p->part_kind == 1)
		R.n_reads = long_reads;
	if (rc == S1_CHAIN_UNCOVERED)
		return KMC_HIP_UNCOVERED;
	if (rc != S1_CHAIN_OK)
		return fail(KMC_HIP_EDEVICE, "mock split_part: chain failed");
	*recs_bytes = R.recs_bytes;
	if (R.recs_bytes > recs_capacity)
		return fail(KMC_HIP_ECAPACITY, "mock split_part: recs_capacity too small");
	memcpy(recs, R.d_recs, R.recs_bytes);
	for (uint32_t b = 0; b < p->n_bins; ++b) {
		bin_off[b] = R.bin_off[b];
		bin_bytes[b] = R.bin_bytes[b];
		bin_kmers[b] = R.bin_kmers[b];
		bin_superkmers[b] = R.bin_sk[b];
		bin_plus_x[b] = R.bin_plus_x[b];
	}
	*n_reads = R.n_reads;
	return 0;
}
