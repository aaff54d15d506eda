/*
 * kmc_amd/host/bin_engine.h — the per-bin compute engine the stage-2 worker drives.
 *
 * One engine instance per worker thread. The signature of process_bin() is exactly
 * kmc_hip_process_bin() from include/kmc_hip.h minus (ctx, dev), so the worker
 * (kb_sorter_plugin.h) is identical whichever engine is linked:
 *   - HIP engine   : libkmc_hip.so through its C-ABI (hip_loader.cpp dlopens it)   -> oracle/_ref/kmc_hip
 *   - oracle engine: oracle/stage2_oracle.c (TEST ONLY; pins the oracle to the reference) -> oracle/_ref/kmc_oracle
 */
#ifndef KMC_AMD_BIN_ENGINE_H
#define KMC_AMD_BIN_ENGINE_H

#include <stdint.h>
#include <string>

#include "kmc_hip.h"

struct KmcBinEngine {
	virtual ~KmcBinEngine() {}
	virtual int process_bin(const kmc_hip_bin_params &params, const uint8_t *superkmers, uint64_t size, uint64_t n_rec,
	                        const uint64_t *pack_bytes, uint64_t n_packs, uint8_t *out_suffix, uint64_t out_capacity,
	                        uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4]) = 0;
	/* Several bins at once (a worker that found more than one bin waiting): results per bin as process_bin's. The default takes them one after the
	 * other; the HIP engine hands them to kmc_hip_process_bins_submit/_wait, where they share one sort (include/kmc_hip.h). Returns the first error. */
	virtual int process_bins(const kmc_hip_bin_params &params, const kmc_hip_host_bin *bins, uint32_t n_bins, uint64_t *out_bytes, uint64_t *stats)
	{
		for (uint32_t i = 0; i < n_bins; ++i)
			if (int rc = process_bin(params, bins[i].superkmers, bins[i].size, bins[i].n_rec, bins[i].pack_bytes, bins[i].n_packs, bins[i].out_suffix,
			                         bins[i].out_capacity, out_bytes + i, bins[i].lut, stats + 4 * i))
				return rc;
		return 0;
	}
	virtual std::string last_error() = 0;
};

/* Provided by exactly one of: hip_loader.cpp (HIP) or the oracle adapter inside kb_sorter_plugin.h. */
KmcBinEngine *kmc_make_bin_engine(int worker_idx, int n_workers);

#endif
