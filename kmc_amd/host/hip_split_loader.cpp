/*
 * kmc_amd/host/hip_split_loader.cpp — the HIP split engine of the stage-1 worker (kb_splitter_plugin.h): kmc_hip_split_part of
 * include/kmc_hip.h through the library hip_loader.cpp already loaded. Had not met a real GPU when round 2 ended; proven on the
 * CPU twice: bound to a mock of the library and to the library's own source over an emulated HIP runtime (tests/test_stage1_plugin.py,
 * tests/test_hostlib_emulated.py).
 * No CPU fallback: without the library or a GPU the engine reports the error and the worker raises it.
 */
#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

#include "kmc_hip.h"
#include "split_engine.h"

bool kmc_hip_loader_handles(void *&so, kmc_hip_ctx *&ctx, int &n_dev, int &n_slots, std::string &err); /* hip_loader.cpp */

namespace {
struct SplitApi {
	int (*set_map)(kmc_hip_ctx *, int, const int32_t *, uint32_t) = nullptr;
	int (*split_part)(kmc_hip_ctx *, int, int, const kmc_hip_split_params *, const uint8_t *, uint64_t, uint8_t *, uint64_t, uint64_t *, uint64_t *, uint64_t *,
	                  uint64_t *, uint64_t *, uint64_t *, uint64_t *) = nullptr;
	const char *(*last_error)(kmc_hip_ctx *) = nullptr;
	kmc_hip_ctx *ctx = nullptr;
	int n_dev = 1, n_slots = 1;
	std::string err;
	std::mutex map_mtx;
	std::vector<uint64_t> map_hash; /* per device: hash of the signature -> bin map that was uploaded last (0 = none) */
} g_split;
std::once_flag g_split_once;

void bind()
{
	void *so = nullptr;
	if (!kmc_hip_loader_handles(so, g_split.ctx, g_split.n_dev, g_split.n_slots, g_split.err))
		return;
	g_split.set_map = reinterpret_cast<decltype(g_split.set_map)>(dlsym(so, "kmc_hip_split_set_map"));
	g_split.split_part = reinterpret_cast<decltype(g_split.split_part)>(dlsym(so, "kmc_hip_split_part"));
	g_split.last_error = reinterpret_cast<decltype(g_split.last_error)>(dlsym(so, "kmc_hip_last_error"));
	if (!g_split.set_map || !g_split.split_part || !g_split.last_error) {
		g_split.err = "libkmc_hip.so lacks kmc_hip_split_set_map / kmc_hip_split_part";
		g_split.ctx = nullptr;
	}
	g_split.map_hash.assign((size_t)g_split.n_dev, 0);
}

struct HipSplitEngine : KmcSplitEngine {
	KmcSplitParams P;
	int dev, slot;
	std::string err;
	std::vector<uint8_t> recs;
	std::vector<uint64_t> arrays; /* bin_off | bin_bytes | bin_kmers | bin_superkmers | bin_plus_x */

	uint64_t map_hash = 0; /* of this run's map: a second KMC run in the same process (library API) with another map must not split with the first one's */
	HipSplitEngine(const KmcSplitParams &p, int dev, int slot) : P(p), dev(dev), slot(slot)
	{
		arrays.assign((size_t)5 * p.n_bins, 0);
		uint64_t h = 1469598103934665603ull ^ p.signature_len; /* FNV-1a over the map's entries, once per engine */
		const size_t n = ((size_t)1 << (2 * p.signature_len)) + 1;
		for (size_t i = 0; i < n; ++i)
			h = (h ^ (uint32_t)p.sig_to_bin[i]) * 1099511628211ull;
		map_hash = h ? h : 1;
	}
	std::string last_error() override { return err; }
	int split_part(const uint8_t *text, uint64_t size, bool long_read, KmcSplitResult &out) override
	{
		if (!g_split.ctx) {
			err = g_split.err.empty() ? "HIP split engine not initialised" : g_split.err;
			return KMC_HIP_EDEVICE;
		}
		{
			std::lock_guard<std::mutex> lck(g_split.map_mtx);
			if (g_split.map_hash[dev] != map_hash) {
				if (int rc = g_split.set_map(g_split.ctx, dev, P.sig_to_bin, P.signature_len)) {
					err = g_split.last_error(g_split.ctx);
					return rc;
				}
				g_split.map_hash[dev] = map_hash;
			}
		}
		kmc_hip_split_params hp;
		hp.kmer_len = P.kmer_len;
		hp.signature_len = P.signature_len;
		hp.n_bins = P.n_bins;
		hp.max_x = P.max_x;
		hp.both_strands = P.both_strands ? 1u : 0u;
		hp.file_type = (uint32_t)P.file_type;
		hp.line_cap = P.line_cap;
		hp.part_kind = long_read ? 1u : 0u;
		hp.reserved = 0;
		/* records of real reads take ~0.3 bytes per symbol; text whose k-mers are nearly all their own super-k-mer needs more: second call */
		if (recs.size() < size + 256ull * (P.n_bins + 1))
			recs.resize(size + 256ull * (P.n_bins + 1));
		const size_t nb = P.n_bins;
		uint64_t n_reads = 0, need = 0;
		int rc = g_split.split_part(g_split.ctx, dev, slot, &hp, text, size, recs.data(), recs.size(), &need, &arrays[0], &arrays[nb], &arrays[2 * nb], &arrays[3 * nb],
		                            &arrays[4 * nb], &n_reads);
		if (rc == KMC_HIP_ECAPACITY && need > recs.size()) {
			recs.resize(need);
			rc = g_split.split_part(g_split.ctx, dev, slot, &hp, text, size, recs.data(), recs.size(), &need, &arrays[0], &arrays[nb], &arrays[2 * nb], &arrays[3 * nb],
			                        &arrays[4 * nb], &n_reads);
		}
		if (rc == KMC_HIP_UNCOVERED)
			return KMC_SPLIT_UNCOVERED;
		if (rc) {
			err = g_split.last_error(g_split.ctx);
			return rc;
		}
		out.recs = recs.data();
		out.bin_off = &arrays[0];
		out.bin_bytes = &arrays[nb];
		out.bin_kmers = &arrays[2 * nb];
		out.bin_superkmers = &arrays[3 * nb];
		out.bin_plus_x = &arrays[4 * nb];
		out.n_reads = n_reads;
		return 0;
	}
};
} // namespace

KmcSplitEngine *kmc_make_split_engine(const KmcSplitParams &params, int worker_idx, int /*n_workers*/)
{
	std::call_once(g_split_once, bind);
	const int n = g_split.n_dev > 0 ? g_split.n_dev : 1;
	return new HipSplitEngine(params, worker_idx % n, (worker_idx / n) % (g_split.n_slots > 0 ? g_split.n_slots : 1));
}
