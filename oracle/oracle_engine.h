/*
 * oracle/oracle_engine.h — TEST INFRASTRUCTURE, not product: the CPU restatement (oracle/stage2_oracle.c) wrapped as a
 * KmcBinEngine, so that the reference pipeline + the stage-2 worker of kmc_amd/host/kb_sorter_plugin.h can be run with the
 * oracle as the per-bin engine (`oracle/_ref/kmc_oracle`, oracle/Makefile). That build pins the oracle to the reference: its
 * database must equal `kmc -sr1` byte for byte (tests/test_oracle.py). Nothing in the product (libkmc_hip.so, kmc_hip)
 * includes this file. Optionally tees every bin (input image + outputs) to $KMC_BIN_DUMP so tests/golden fixtures can be
 * cut from real reference stage-1 bins (tests/golden/make_golden.py).
 */
#ifndef KMC_ORACLE_ENGINE_H
#define KMC_ORACLE_ENGINE_H

#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "bin_engine.h"
#include "stage2_oracle.h"
#include <cstdlib>
#include <cstring>
struct KmcOracleEngine : KmcBinEngine {
	std::string err;
	int process_bin(const kmc_hip_bin_params &p, const uint8_t *sk, uint64_t size, uint64_t n_rec,
	                const uint64_t *pack_bytes, uint64_t n_packs, uint8_t *out, uint64_t cap, uint64_t *out_bytes,
	                uint64_t *lut, uint64_t stats[4]) override
	{
		oracle_params op;
		op.kmer_len = p.kmer_len;
		op.both_strands = p.both_strands;
		op.cutoff_min = p.cutoff_min;
		op.without_output = p.without_output;
		op.cutoff_max = p.cutoff_max;
		op.counter_max = p.counter_max;
		op.lut_prefix_len = p.lut_prefix_len;
		op.output_type = p.output_type;
		const char *dump = getenv("KMC_BIN_DUMP");
		std::vector<uint8_t> copy;
		if (dump)
			copy.assign(sk, sk + size); /* out may alias sk */
		int rc = oracle_process_bin(&op, sk, size, n_rec, out, cap, out_bytes, lut, stats);
		if (rc) {
			err = "oracle_process_bin failed, code " + std::to_string(rc);
			return rc;
		}
		if (dump) {
			static std::mutex mtx;
			std::lock_guard<std::mutex> lck(mtx);
			FILE *f = fopen(dump, "ab");
			if (f) {
				uint64_t lut_n = p.lut_prefix_len ? 1ull << (2 * p.lut_prefix_len) : 0;
				uint64_t hdr[8] = {0x4B4D4342494E3031ull /* "KMCBIN01" */, size, n_rec, n_packs, *out_bytes, lut_n, 0, 0};
				fwrite(hdr, 8, 8, f);
				fwrite(&p, sizeof p, 1, f);
				fwrite(stats, 8, 4, f);
				fwrite(pack_bytes, 8, n_packs, f);
				fwrite(copy.data(), 1, size, f);
				fwrite(out, 1, *out_bytes, f);
				fwrite(lut, 8, lut_n, f);
				fclose(f);
			}
		}
		return 0;
	}
	std::string last_error() override { return err; }
};
inline KmcBinEngine *kmc_make_bin_engine(int, int) { return new KmcOracleEngine(); }

#endif
