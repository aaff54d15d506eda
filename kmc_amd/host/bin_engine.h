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
	virtual std::string last_error() = 0;
};

/* Provided by exactly one of: hip_loader.cpp (HIP) or the oracle adapter inside kb_sorter_plugin.h. */
KmcBinEngine *kmc_make_bin_engine(int worker_idx, int n_workers);

#endif
