/*
 * tests/hipemu/emu_split_engine.cpp — TEST INFRASTRUCTURE: a KmcSplitEngine (kmc_amd/host/split_engine.h) whose split_part() runs the stage-1
 * KERNELS of kmc_amd/csrc/stage1_kernels.hip.h under the CPU emulation of tests/hipemu, in the order a HIP engine will launch them:
 *   k_s1_text_to_codes -> k_s1_check_records -> k_s1_cut<true> -> k_s1_bin_totals + k_s1_bin_plus_x -> k_s1_bin_layout -> k_s1_emit.
 * Linked into oracle/_ref/kmc_emu_s1 (reference pipeline + kb_splitter_plugin.h): the kernel chain inside the real KMC, checked by the
 * database it leads to (tests/test_stage1_plugin.py). Emulated "LDS" is static storage, so calls are serialised.
 */
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../kmc_amd/csrc/kernels.hip.h"
#include "../../kmc_amd/csrc/stage1_kernels.hip.h"
#include "../../kmc_amd/host/split_engine.h"

namespace {
std::mutex g_launch_mtx;

struct EmuSplitEngine : KmcSplitEngine {
	KmcSplitParams P;
	std::string err_msg;
	std::vector<int8_t> codes;
	std::vector<u64> nl_pos, sk_pos, tot, lay, pack_start, plus_x, st_a, st_b, bin_off, bin_bytes, bin_kmers, bin_sk;
	std::vector<u32> sk_len, sk_sig;
	std::vector<uint8_t> recs;

	explicit EmuSplitEngine(const KmcSplitParams &p) : P(p) {}
	std::string last_error() override { return err_msg; }

	int split_part(const uint8_t *text, uint64_t size, KmcSplitResult &out) override
	{
		std::lock_guard<std::mutex> lck(g_launch_mtx);
		const u32 nb = P.n_bins, lpr = P.file_type == 1 ? 4u : 2u;
		u32 err = 0, ticket = 0;
		u64 totals[2] = {0, 0};
		/* text -> codes */
		codes.assign(size + 16, 0);
		nl_pos.assign(size + 16, 0);
		if (size) {
			const u32 tiles = (u32)((size + S1_TXT_TILE - 1) / S1_TXT_TILE);
			st_a.assign(tiles, 0);
			st_b.assign(tiles, 0);
			hipemu::launch(dim3(tiles), dim3(S1_BLOCK), 0,
			               [&] { k_s1_text_to_codes(text, size, lpr, st_a.data(), st_b.data(), &ticket, codes.data(), nl_pos.data(), nl_pos.size(), totals, &err); });
			const u64 n_rec = totals[0] / lpr + 1;
			hipemu::launch(dim3((u32)((n_rec + 255) / 256)), dim3(256), 0, [&] { k_s1_check_records(text, size, nl_pos.data(), totals[0], lpr, P.line_cap, &err); });
		}
		if (err & S1_TEXT_BAD)
			return KMC_SPLIT_UNCOVERED;
		if (err)
			return fail(err, "text_to_codes");
		const u64 n = totals[1];
		/* codes -> super-k-mers */
		u64 n_sk = 0;
		sk_pos.assign(n + 8, 0);
		sk_len.assign(n + 8, 0);
		sk_sig.assign(n + 8, 0);
		if (n) {
			const u32 ct = (u32)s1_cut_tiles(n);
			st_a.assign(ct, 0);
			st_b.assign(ct, 0);
			ticket = 0;
			hipemu::launch(dim3(ct), dim3(S1_BLOCK), 0, [&] {
				k_s1_cut<true>((const u32 *)nullptr, codes.data(), P.signature_len, n, P.kmer_len, st_a.data(), st_b.data(), &ticket, sk_pos.data(), sk_len.data(),
				               sk_sig.data(), sk_pos.size(), &n_sk, &err);
			});
		}
		if (err)
			return fail(err, "cut");
		/* per-bin sums, layout, records */
		tot.assign(3 * (size_t)nb, 0);
		plus_x.assign(nb, 0);
		lay.assign(3 * (size_t)nb + 2, 0);
		const u32 sk_tiles = (u32)((n_sk + S1_SK_TILE - 1) / S1_SK_TILE);
		if (sk_tiles) {
			hipemu::launch(dim3(sk_tiles), dim3(256), 0, [&] {
				k_s1_bin_totals(sk_len.data(), sk_sig.data(), n_sk, P.kmer_len, P.sig_to_bin, nb, tot.data(), tot.data() + nb, tot.data() + 2 * nb, &err);
			});
			hipemu::launch(dim3(sk_tiles), dim3(256), 0, [&] {
				k_s1_bin_plus_x(codes.data(), sk_pos.data(), sk_len.data(), sk_sig.data(), n_sk, P.kmer_len, P.max_x, (u32)P.both_strands, P.sig_to_bin, nb, plus_x.data());
			});
		}
		if (err)
			return fail(err, "bin_totals");
		hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_bin_layout(tot.data(), nb, lay.data(), lay.data() + nb + 1, lay.data() + 2 * nb + 2, (u64 *)nullptr); });
		recs.assign(lay[nb] + 16, 0);
		pack_start.assign(lay[2 * (size_t)nb + 1] + 1, 0);
		hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_bin_layout(tot.data(), nb, lay.data(), lay.data() + nb + 1, lay.data() + 2 * nb + 2, pack_start.data()); });
		if (sk_tiles)
			hipemu::launch(dim3(sk_tiles), dim3(256), 0, [&] {
				k_s1_emit(codes.data(), sk_pos.data(), sk_len.data(), sk_sig.data(), n_sk, P.kmer_len, P.sig_to_bin, nb, lay.data(), lay.data() + nb + 1,
				          lay.data() + 2 * nb + 2, recs.data(), pack_start.data());
			});
		bin_off.assign(lay.begin(), lay.begin() + nb);
		bin_bytes.assign(tot.begin(), tot.begin() + nb);
		bin_sk.assign(tot.begin() + nb, tot.begin() + 2 * nb);
		bin_kmers.assign(tot.begin() + 2 * nb, tot.begin() + 3 * nb);
		out.recs = recs.data();
		out.bin_off = (const uint64_t *)bin_off.data();
		out.bin_bytes = (const uint64_t *)bin_bytes.data();
		out.bin_kmers = (const uint64_t *)bin_kmers.data();
		out.bin_superkmers = (const uint64_t *)bin_sk.data();
		out.bin_plus_x = (const uint64_t *)plus_x.data();
		/* titles in the part: every lines_per_record-th line, an unterminated last line included */
		const u64 lines = totals[0] + ((size && text[size - 1] != '\n') ? 1 : 0);
		out.n_reads = (lines + lpr - 1) / lpr;
		return 0;
	}
	int fail(u32 e, const char *where)
	{
		err_msg = std::string("emulated kernel error in ") + where + ", device error word " + std::to_string(e);
		return -(int)e;
	}
};
} // namespace

KmcSplitEngine *kmc_make_split_engine(const KmcSplitParams &params, int, int) { return new EmuSplitEngine(params); }
