/*
 * tests/hipemu/emu_kernels.cpp — TEST INFRASTRUCTURE: runs the kernels of kmc_amd/csrc/kernels.hip.h on the CPU under the
 * emulation in tests/hipemu/include/hip/hip_runtime.h, launched the way kmc_amd/csrc/kmc_hip.hip launches them on the GPU
 * (same grids, workgroup sizes, LDS sizes, zeroed work areas). Used only by tests/test_kernels_emulated.py (`-m "not gpu"`):
 * it lets kernel logic be checked against the oracle in a container without a GPU. Never linked into libkmc_hip.so.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "../../kmc_amd/csrc/kernels.hip.h"
#include "../../kmc_amd/csrc/stage1_kernels.hip.h"

namespace {

struct Params {
	u32 k, both_strands, cutoff_min, without_output;
	u64 cutoff_max, counter_max;
	u32 lut_prefix_len, output_type;
};

DevParams dev_params(const Params &p)
{
	DevParams P;
	P.k = p.k;
	P.both_strands = p.both_strands ? 1 : 0;
	P.cutoff_min = p.cutoff_min;
	P.cutoff_max = (u32)p.cutoff_max;
	P.counter_max = (u32)p.counter_max;
	P.lut_prefix_len = p.lut_prefix_len;
	P.sbytes = kmc_suffix_bytes(p.k, p.lut_prefix_len);
	P.cbytes = kmc_counter_bytes(p.cutoff_max, p.counter_max);
	P.kff = p.output_type == 1;
	P.without_output = p.without_output ? 1 : 0;
	return P;
}

u32 lut_shards_for(u64 lut_entries) { return lut_entries <= 256 ? 32u : (lut_entries <= 1024 ? 8u : 1u); }

template <int SIZE> int expand_t(const DevParams &P, const uint8_t *img, u64 size, u64 n_rec, const u64 *pack_start, u64 n_packs, u64 *recs, u64 *dbase, u32 *err)
{
	const u32 n_pass = (2 * P.k + 7) / 8;
	std::vector<u32> bitmap((size + 31) / 32 + 2, 0);
	const u64 n_chunks = (size + EXP_CHUNK - 1) / EXP_CHUNK;
	std::vector<u64> status(n_chunks + 1, 0), ghist((size_t)n_pass * 256, 0);
	u32 counters[2] = {0, 0};
	/* the image needs 256 readable bytes of slack, like the device buffer */
	std::vector<uint8_t> in(size + 512, 0);
	memcpy(in.data(), img, size);
	/* a group of one, as kmc_hip.hip front_end_group builds it */
	GrpParse gp = {};
	GrpExpand ge = {};
	gp.g = ge.g = 1;
	gp.pack_prefix[1] = (u32)n_packs;
	ge.chunk_prefix[1] = (u32)n_chunks;
	gp.data[0] = ge.data[0] = in.data();
	gp.pack_start[0] = pack_start;
	gp.bitmap[0] = bitmap.data();
	ge.bitmap[0] = bitmap.data();
	ge.size[0] = size;
	ge.n_rec[0] = n_rec;
	ge.out[0] = recs;
	ge.status[0] = status.data();
	hipemu::launch(dim3((u32)n_packs), dim3(PARSE_BLOCK), 0, [&] { k_parse_packs(gp, P.k, err); });
	if (getenv("KMC_EMU_VERBOSE"))
		fprintf(stderr, "[emu] parse done, err %u\n", *err);
	const bool fuse = n_pass <= EXP_FUSE_MAX_PASS && n_rec >= 2;
	const u32 blocks = (u32)std::min<u64>(n_chunks, 4); /* persistent workgroups pulling slice tickets */
	if (fuse)
		hipemu::launch(dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<true>(n_pass, P.k),
		               [&] { k_expand<SIZE, true>(ge, P.k, P.both_strands, n_pass, ghist.data(), &counters[0], err, dbase, &counters[1], 0u); });
	else
		hipemu::launch(dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<false>(n_pass, P.k),
		               [&] { k_expand<SIZE, false>(ge, P.k, P.both_strands, n_pass, ghist.data(), &counters[0], err, nullptr, &counters[1], 0u); });
	return fuse ? 1 : 0;
}

template <int SIZE> u64 *sort_t(u64 *a, u64 *b, u64 n, u32 n_pass, const u64 *dbase_in, bool hist_done, u32 *err)
{
	if (n < 2 || n_pass == 0)
		return a;
	std::vector<u64> ghist((size_t)n_pass * 256, 0), dbase((size_t)n_pass * 256, 0);
	if (hist_done)
		memcpy(dbase.data(), dbase_in, dbase.size() * 8);
	else {
		hipemu::launch(dim3((u32)std::min<u64>((n + 255) / 256, 4)), dim3(256), (size_t)n_pass * 1024, [&] { k_hist<SIZE>(a, n, n_pass, ghist.data(), 0u); });
		hipemu::launch(dim3(n_pass), dim3(256), 0, [&] { k_hist_scan(ghist.data(), dbase.data()); });
	}
	u64 *src = a, *dst = b;
	const u32 tiles = (u32)((n + RsCfg<SIZE>::TILE - 1) / RsCfg<SIZE>::TILE);
	for (u32 pass = 0; pass < n_pass; ++pass) {
		std::vector<u32> status((size_t)tiles * 256, 0);
		u32 counter = 0;
		u64 next[256];
		hipemu::launch(dim3(tiles), dim3(RS_BLOCK), rs_lds_bytes<SIZE>(), [&] {
			k_onesweep<SIZE>(src, dst, (u32)n, pass, dbase.data() + (size_t)pass * 256, next, status.data(), &counter, tiles, err);
		});
		std::swap(src, dst);
	}
	return src;
}

template <int SIZE>
void compact_t(const DevParams &P, const u64 *sorted, u64 n, uint8_t *out, u64 out_capacity, u64 *out_bytes, u64 *lut, u64 lut_entries, u64 *stats, u32 *err)
{
	const u64 c_tiles = (n + CpCfg<SIZE>::TILE - 1) / CpCfg<SIZE>::TILE;
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = use_lut ? lut_shards_for(lut_entries) : 1u;
	std::vector<u64> lutsh(n_sh > 1 ? (size_t)n_sh * lut_entries : 1, 0), status(c_tiles + 1, 0), shards(CP_SHARDS * 4, 0);
	u64 *lut_base = lut;
	if (use_lut && n_sh > 1)
		lut_base = lutsh.data();
	else if (use_lut)
		memset(lut, 0, lut_entries * 8);
	u32 counter = 0;
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const u64 tile_pitch = (u64)CpCfg<SIZE>::TILE * SIZE * 8;
	const bool two_phase = !getenv("KMC_EMU_NO_TWO_PHASE") && !P.without_output &&
	                       ((u64)CpCfg<SIZE>::TILE / std::max<u32>(P.cutoff_min, 1) + 1) * rec_bytes <= tile_pitch; /* the host's rule (compact_group) */
	std::vector<uint8_t> scratch(two_phase ? c_tiles * tile_pitch + 256 : 1, 0xEE);
	GrpCompact gc = {};
	GrpFold gf = {};
	GrpGather gg = {};
	gc.g = gg.g = 1;
	gc.tile_prefix[1] = gg.tile_prefix[1] = (u32)c_tiles;
	gc.scratch[0] = two_phase ? scratch.data() : nullptr;
	gf.status[0] = status.data();
	gf.n_tiles[0] = (u32)c_tiles;
	gf.out_bytes[0] = out_bytes;
	gf.out_capacity[0] = out_capacity;
	gg.scratch[0] = scratch.data();
	gg.prefix[0] = status.data();
	gg.out[0] = out;
	gg.out_capacity[0] = out_capacity;
	gc.S[0] = sorted;
	gc.n[0] = gf.n[0] = n;
	gc.out[0] = out;
	gc.out_capacity[0] = out_capacity;
	gc.lut_base[0] = lut_base;
	gc.tally[0] = shards.data();
	gc.out_bytes[0] = out_bytes;
	gc.status[0] = status.data();
	gf.tally[0] = shards.data();
	gf.stats[0] = stats;
	gf.lut_base[0] = lut_base;
	gf.lut_out[0] = lut;
	hipemu::launch(dim3((u32)c_tiles), dim3(CP_BLOCK), 0, [&] {
		k_compact<SIZE>(gc, P, n_sh, lut_entries, &counter, err, P.lut_prefix_len ? (u32)((1ull << (2 * P.lut_prefix_len)) - 1) : 0u, two_phase ? 1u : 0u);
	});
	hipemu::launch(dim3(1), dim3(256), 0, [&] { k_compact_fold(gf, use_lut ? n_sh : 1u, lut_entries, two_phase ? 1u : 0u, rec_bytes, err); });
	if (two_phase)
		hipemu::launch(dim3((u32)((c_tiles + 3) / 4)), dim3(256), 0, [&] { k_compact_gather(gg, rec_bytes, tile_pitch); });
}

template <int SIZE>
int run_t(const Params &p, int stage_mask, const uint8_t *img, u64 size, u64 n_rec, const u64 *pack_start, u64 n_packs, u64 *recs /* [2][n_rec*SIZE] */,
          u64 **sorted_out, uint8_t *out, u64 out_capacity, u64 *out_bytes, u64 *lut, u64 *stats, u32 *err)
{
	const DevParams P = dev_params(p);
	const u32 n_pass = (2 * P.k + 7) / 8;
	u64 *a = recs, *b = recs + n_rec * SIZE;
	std::vector<u64> dbase((size_t)n_pass * 256, 0);
	int hist_done = 0;
	if (stage_mask & 1)
		hist_done = expand_t<SIZE>(P, img, size, n_rec, pack_start, n_packs, a, dbase.data(), err);
	u64 *sorted = a;
	if (stage_mask & 2)
		sorted = sort_t<SIZE>(a, b, n_rec, n_pass, dbase.data(), hist_done != 0, err);
	if (sorted_out)
		*sorted_out = sorted;
	if (stage_mask & 4) {
		const u64 lut_entries = (P.kff || !p.lut_prefix_len) ? 0 : 1ull << (2 * p.lut_prefix_len);
		compact_t<SIZE>(P, sorted, n_rec, out, out_capacity, out_bytes, lut, lut_entries, stats, err);
	}
	return 0;
}

/* ---- a group of bins, the way kmc_hip.hip front_end_group / compact_group build the descriptors ---- */
template <int SIZE>
int group_front_t(const DevParams &P, int g, const uint8_t *const *imgs, const u64 *sizes, const u64 *n_recs, const u64 *const *pack_starts, const u64 *n_packs,
                  u64 *recs, u32 n_pass, u32 *err)
{
	GrpParse gp = {};
	GrpExpand ge = {};
	gp.g = ge.g = (u32)g;
	std::vector<std::vector<uint8_t>> in(g);
	std::vector<std::vector<u32>> bitmap(g);
	std::vector<std::vector<u64>> status(g);
	u64 packs = 0, chunks = 0, rec_off = 0;
	const u32 tag_shift = (2 * P.k) & 63;
	for (int i = 0; i < g; ++i) {
		in[i].assign(sizes[i] + 512, 0);
		memcpy(in[i].data(), imgs[i], sizes[i]);
		bitmap[i].assign((sizes[i] + 31) / 32 + 2, 0);
		const u64 nc = (sizes[i] + EXP_CHUNK - 1) / EXP_CHUNK;
		status[i].assign(nc + 1, 0);
		gp.pack_prefix[i] = (u32)packs;
		ge.chunk_prefix[i] = (u32)chunks;
		packs += n_packs[i];
		chunks += nc;
		gp.data[i] = ge.data[i] = in[i].data();
		gp.pack_start[i] = pack_starts[i];
		gp.bitmap[i] = bitmap[i].data();
		ge.bitmap[i] = bitmap[i].data();
		ge.size[i] = sizes[i];
		ge.n_rec[i] = n_recs[i];
		ge.out[i] = recs + rec_off * SIZE;
		ge.status[i] = status[i].data();
		ge.tag[i] = (u64)i << tag_shift;
		rec_off += n_recs[i];
	}
	gp.pack_prefix[g] = (u32)packs;
	ge.chunk_prefix[g] = (u32)chunks;
	std::vector<u64> ghist((size_t)n_pass * 256, 0), dbase((size_t)n_pass * 256, 0);
	u32 counters[2] = {0, 0};
	hipemu::launch(dim3((u32)packs), dim3(PARSE_BLOCK), 0, [&] { k_parse_packs(gp, P.k, err); });
	const u32 blocks = (u32)std::min<u64>(chunks, 3);
	hipemu::launch(dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<true>(n_pass, P.k),
	               [&] { k_expand<SIZE, true>(ge, P.k, P.both_strands, n_pass, ghist.data(), &counters[0], err, dbase.data(), &counters[1], 0u); });
	/* the fused histograms must cover every record of the group, tag included: digit totals == records */
	for (u32 b = 0; b < n_pass; ++b) {
		u64 tot = 0;
		for (int d = 0; d < 256; ++d)
			tot += ghist[(size_t)b * 256 + d];
		if (tot != rec_off)
			return -2;
	}
	return 0;
}

template <int SIZE>
int group_compact_t(const DevParams &P, int g, const u64 *sorted, const u64 *n_recs, uint8_t *const *outs, u64 out_capacity, u64 *out_bytes, u64 *luts, u64 lut_entries,
                    u64 *stats, u32 *err)
{
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = use_lut ? lut_shards_for(lut_entries) : 1u;
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const u64 tile_pitch = (u64)CpCfg<SIZE>::TILE * SIZE * 8;
	const bool two_phase = !getenv("KMC_EMU_NO_TWO_PHASE") && !P.without_output &&
	                       ((u64)CpCfg<SIZE>::TILE / std::max<u32>(P.cutoff_min, 1) + 1) * rec_bytes <= tile_pitch;
	GrpCompact gc = {};
	GrpFold gf = {};
	GrpGather gg = {};
	gc.g = gg.g = (u32)g;
	std::vector<std::vector<u64>> lutsh(g), status(g), shards(g);
	std::vector<std::vector<uint8_t>> scratch(g);
	u64 tiles = 0, rec_off = 0;
	for (int i = 0; i < g; ++i) {
		const u64 nt = (n_recs[i] + CpCfg<SIZE>::TILE - 1) / CpCfg<SIZE>::TILE;
		lutsh[i].assign(n_sh > 1 ? (size_t)n_sh * lut_entries : 1, 0);
		status[i].assign(nt + 1, 0);
		shards[i].assign(CP_SHARDS * 4, 0);
		u64 *lut_i = luts + (size_t)i * (lut_entries ? lut_entries : 1);
		u64 *lut_base = lut_i;
		if (use_lut && n_sh > 1)
			lut_base = lutsh[i].data();
		else if (use_lut)
			memset(lut_i, 0, lut_entries * 8);
		gc.tile_prefix[i] = gg.tile_prefix[i] = (u32)tiles;
		tiles += nt;
		scratch[i].assign(two_phase ? nt * tile_pitch + 256 : 1, 0xEE);
		gc.scratch[i] = two_phase ? scratch[i].data() : nullptr;
		gf.status[i] = status[i].data();
		gf.n_tiles[i] = (u32)nt;
		gf.out_bytes[i] = out_bytes + i;
		gf.out_capacity[i] = out_capacity;
		gg.scratch[i] = scratch[i].data();
		gg.prefix[i] = status[i].data();
		gg.out[i] = outs[i];
		gg.out_capacity[i] = out_capacity;
		gc.S[i] = sorted + rec_off * SIZE;
		gc.n[i] = gf.n[i] = n_recs[i];
		gc.out[i] = outs[i];
		gc.out_capacity[i] = out_capacity;
		gc.lut_base[i] = lut_base;
		gc.tally[i] = shards[i].data();
		gc.out_bytes[i] = out_bytes + i;
		gc.status[i] = status[i].data();
		gf.tally[i] = shards[i].data();
		gf.stats[i] = stats + 4 * i;
		gf.lut_base[i] = lut_base;
		gf.lut_out[i] = lut_i;
		rec_off += n_recs[i];
	}
	gc.tile_prefix[g] = gg.tile_prefix[g] = (u32)tiles;
	u32 counter = 0;
	hipemu::launch(dim3((u32)tiles), dim3(CP_BLOCK), 0, [&] {
		k_compact<SIZE>(gc, P, n_sh, lut_entries, &counter, err, P.lut_prefix_len ? (u32)((1ull << (2 * P.lut_prefix_len)) - 1) : 0u, two_phase ? 1u : 0u);
	});
	hipemu::launch(dim3((u32)g), dim3(256), 0, [&] { k_compact_fold(gf, use_lut ? n_sh : 1u, lut_entries, two_phase ? 1u : 0u, rec_bytes, err); });
	if (two_phase)
		hipemu::launch(dim3((u32)((tiles + 3) / 4)), dim3(256), 0, [&] { k_compact_gather(gg, rec_bytes, tile_pitch); });
	return 0;
}

Params params_of(const unsigned *params10)
{
	Params p;
	p.k = params10[0];
	p.both_strands = params10[1];
	p.cutoff_min = params10[2];
	p.without_output = params10[3];
	p.cutoff_max = params10[4];
	p.counter_max = params10[5];
	p.lut_prefix_len = params10[6];
	p.output_type = params10[7];
	return p;
}

} // namespace

extern "C" {
#define EMU_API __attribute__((visibility("default")))

/* lookback64 on a prepared status array (no concurrency): returns the exclusive prefix wave 0 / lane 0 computes for `tile`; status[tile] is
 * overwritten with the inclusive prefix word */
EMU_API unsigned long long emu_lookback(u64 *status, unsigned tile, unsigned long long aggregate, unsigned *err)
{
	u64 res = 0;
	hipemu::launch(dim3(1), dim3(64), 0, [&] {
		const u64 e = lookback64(status, tile, aggregate, threadIdx.x & 63, err, KERR_WATCHDOG);
		if (threadIdx.x == 0)
			res = e;
	});
	return res;
}

/* stage_mask: 1 = parse + expand (image -> recs[0]), 2 = sort (recs[0] -> *sorted_index = 0 or 1: which half of `recs` holds the result),
 * 4 = compaction of the sorted half. `recs` = 2 x n_rec x words uint64. Returns the device error word. */
EMU_API int emu_run(const unsigned *params10, int stage_mask, const uint8_t *img, u64 size, u64 n_rec, const u64 *pack_start, u64 n_packs, u64 *recs,
            int *sorted_index, uint8_t *out, u64 out_capacity, u64 *out_bytes, u64 *lut, u64 *stats)
{
	Params p;
	p.k = params10[0];
	p.both_strands = params10[1];
	p.cutoff_min = params10[2];
	p.without_output = params10[3];
	p.cutoff_max = params10[4];
	p.counter_max = params10[5];
	p.lut_prefix_len = params10[6];
	p.output_type = params10[7];
	u32 err = 0;
	u64 *sorted = nullptr;
	const u32 words = (p.k + 31) / 32;
#define RUN(N) run_t<N>(p, stage_mask, img, size, n_rec, pack_start, n_packs, recs, &sorted, out, out_capacity, out_bytes, lut, stats, &err)
	switch (words) {
	case 1: RUN(1); break;
	case 2: RUN(2); break;
	case 3: RUN(3); break;
	case 4: RUN(4); break;
	case 5: RUN(5); break;
	case 6: RUN(6); break;
	case 7: RUN(7); break;
	case 8: RUN(8); break;
	default: return -1;
	}
#undef RUN
	if (sorted_index)
		*sorted_index = sorted == recs ? 0 : 1;
	return (int)err;
}

/* parse + expand of a group of g bins into ONE record array (records of bin i carry tag i above the k-mer); n_pass as the host picks it
 * for the group. Returns the device error word, -2 if the fused histograms do not cover every record. */
EMU_API int emu_group_front(const unsigned *params10, int g, const uint8_t *const *imgs, const u64 *sizes, const u64 *n_recs, const u64 *const *pack_starts,
                            const u64 *n_packs, u64 *recs, unsigned n_pass)
{
	const DevParams P = dev_params(params_of(params10));
	u32 err = 0;
	int rc = 0;
	switch ((P.k + 31) / 32) {
	case 1: rc = group_front_t<1>(P, g, imgs, sizes, n_recs, pack_starts, n_packs, recs, n_pass, &err); break;
	case 2: rc = group_front_t<2>(P, g, imgs, sizes, n_recs, pack_starts, n_packs, recs, n_pass, &err); break;
	default: return -1;
	}
	return rc ? rc : (int)err;
}

/* compaction + fold of a group on the bin-major sorted array; outs[g] buffers of out_capacity bytes, luts g x 4^p, stats g x 4 */
EMU_API int emu_group_compact(const unsigned *params10, int g, const u64 *sorted, const u64 *n_recs, uint8_t *const *outs, u64 out_capacity, u64 *out_bytes,
                              u64 *luts, u64 *stats)
{
	const Params p = params_of(params10);
	const DevParams P = dev_params(p);
	const u64 lut_entries = (P.kff || !p.lut_prefix_len) ? 0 : 1ull << (2 * p.lut_prefix_len);
	u32 err = 0;
	switch ((P.k + 31) / 32) {
	case 1: group_compact_t<1>(P, g, sorted, n_recs, outs, out_capacity, out_bytes, luts, lut_entries, stats, &err); break;
	case 2: group_compact_t<2>(P, g, sorted, n_recs, outs, out_capacity, out_bytes, luts, lut_entries, stats, &err); break;
	default: return -1;
	}
	return (int)err;
}

/* stage-1 kernels: codes (0..3, negative = invalid / separator) -> signature per k-mer position, and the super-k-mers in position order.
 * Returns the device error word; *n_sk = number of super-k-mers. */
/* fused != 0: the cutting kernel computes the signatures itself (the product path of kmc_hip_split_reads_plan); `sig` is still filled, by k_s1_signatures */
EMU_API int emu_s1_split(const int8_t *codes, u64 n, unsigned k, unsigned m, u32 *sig, u64 *sk_pos, u32 *sk_len, u32 *sk_sig, u64 sk_cap, u64 *n_sk,
                         int fused)
{
	u32 err = 0;
	*n_sk = 0;
	if (!n)
		return 0;
	const u32 tiles = (u32)((n + S1_TILE - 1) / S1_TILE), ctiles = (u32)s1_cut_tiles(n);
	std::vector<u64> st_last(ctiles, 0), st_cnt(ctiles, 0);
	u32 ticket = 0;
	hipemu::launch(dim3(tiles), dim3(S1_BLOCK), 0, [&] { k_s1_signatures(codes, n, k, m, sig); });
	if (fused)
		hipemu::launch(dim3(ctiles), dim3(S1_BLOCK), 0, [&] {
			k_s1_cut<true>((const u32 *)nullptr, codes, m, n, k, st_last.data(), st_cnt.data(), &ticket, sk_pos, sk_len, sk_sig, sk_cap, n_sk, (const u64 *)nullptr, &err);
		});
	else
		hipemu::launch(dim3(ctiles), dim3(S1_BLOCK), 0, [&] {
			k_s1_cut<false>(sig, (const int8_t *)nullptr, 0u, n, k, st_last.data(), st_cnt.data(), &ticket, sk_pos, sk_len, sk_sig, sk_cap, n_sk, (const u64 *)nullptr, &err);
		});
	return (int)err;
}

/* the kernels' computed m-mer normalisation for every m-mer of length m: out[4^m] */
EMU_API void emu_s1_norm_all(unsigned m, u32 *out)
{
	for (u32 x = 0; x < (1u << (2 * m)); ++x)
		out[x] = s1_norm(x, m);
}

/* stage-1 bin scatter: super-k-mers -> bin records in per-bin streams + pack boundaries (k_s1_bin_totals, k_s1_bin_layout, k_s1_emit).
 * bin_base, pack_base: n_bins + 1; totals: [3][n_bins] bytes / super-k-mers / k-mers; out: out_cap bytes; pack_start: pack_cap entries */
EMU_API void emu_s1_geometry(u32 *g)
{
	g[0] = S1_SK_TILE;
	g[1] = S1_PACK_BYTES;
	g[2] = S1_BIN_ALIGN;
}
EMU_API int emu_s1_scatter(const int8_t *codes, const u64 *sk_pos, const u32 *sk_len, const u32 *sk_sig, u64 n_sk, unsigned k, const int *sig_to_bin, unsigned n_bins,
                           u64 *bin_base, u64 *pack_base, u64 *totals, uint8_t *out, u64 out_cap, u64 *pack_start, u64 pack_cap)
{
	u32 err = 0;
	std::vector<u64> cursor(n_bins, 0);
	for (u64 i = 0; i < 3ull * n_bins; ++i)
		totals[i] = 0;
	const u32 tiles = (u32)((n_sk + S1_SK_TILE - 1) / S1_SK_TILE);
	if (tiles)
		hipemu::launch(dim3(tiles), dim3(256), 0,
		               [&] { k_s1_bin_totals(sk_len, sk_sig, n_sk, k, sig_to_bin, n_bins, totals, totals + n_bins, totals + 2 * n_bins, &err); });
	if (err)
		return (int)err;
	/* sizing call first (what a caller does before it allocates), then the real one */
	hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_bin_layout(totals, n_bins, bin_base, pack_base, cursor.data(), (u64 *)nullptr); });
	if (bin_base[n_bins] > out_cap || pack_base[n_bins] > pack_cap)
		return -1;
	hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_bin_layout(totals, n_bins, bin_base, pack_base, cursor.data(), pack_start); });
	if (tiles)
		hipemu::launch(dim3(tiles), dim3(256), 0,
		               [&] { k_s1_emit(codes, sk_pos, sk_len, sk_sig, n_sk, k, sig_to_bin, n_bins, bin_base, pack_base, cursor.data(), out, pack_start); });
	for (unsigned b = 0; b < n_bins; ++b)
		if (cursor[b] != bin_base[b] + totals[b])
			return -2; /* every reserved byte accounted for */
	return 0;
}

/* text of one part -> code stream (+ positions of the line ends) and the record check; totals[0] = '\n' count, totals[1] = bytes of codes.
 * lines_per_record 0: the symbols of a long-read part. line_cap: pieces of longer lines are marked in the codes (S1_PIECE_MARK; *has_marks tells). */
EMU_API int emu_s1_text_to_codes_cap(const uint8_t *text, u64 n, unsigned lines_per_record, int8_t *codes, u64 *nl_pos, u64 nl_cap, u64 *totals, u64 line_cap, unsigned k,
                                     u64 *has_marks)
{
	u32 err = 0, ticket = 0;
	totals[0] = totals[1] = 0;
	*has_marks = 0;
	if (!n)
		return 0;
	const u32 tiles = (u32)((n + S1_TXT_TILE - 1) / S1_TXT_TILE);
	std::vector<u64> st_a(tiles, 0), st_b(tiles, 0), seq_start(nl_cap + 2, 0);
	hipemu::launch(dim3(tiles), dim3(S1_BLOCK), 0,
	               [&] { k_s1_text_to_codes(text, n, lines_per_record, st_a.data(), st_b.data(), &ticket, codes, nl_pos, nl_cap, seq_start.data(), totals, &err); });
	if (err & KERR_CAPACITY)
		return (int)err;
	const u64 stride = line_cap - k + 1;
	if (!lines_per_record) {
		if (totals[1] > stride)
			hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_mark_raw(codes, totals[1], stride, has_marks); });
		return (int)err;
	}
	const u64 recs = totals[0] / lines_per_record + 1;
	hipemu::launch(dim3((u32)((recs + 255) / 256)), dim3(256), 0,
	               [&] { k_s1_check_records(text, n, nl_pos, totals[0], lines_per_record, line_cap, stride, seq_start.data(), codes, has_marks, &err); });
	return (int)err;
}
EMU_API int emu_s1_text_to_codes(const uint8_t *text, u64 n, unsigned lines_per_record, int8_t *codes, u64 *nl_pos, u64 nl_cap, u64 *totals)
{
	u64 has_marks = 0;
	return emu_s1_text_to_codes_cap(text, n, lines_per_record, codes, nl_pos, nl_cap, totals, (u64)131080, 27u, &has_marks);
}

EMU_API void emu_s1_plus_x(const int8_t *codes, const u64 *sk_pos, const u32 *sk_len, const u32 *sk_sig, u64 n_sk, unsigned k, unsigned max_x, unsigned both_strands,
                           const int *sig_to_bin, unsigned n_bins, u64 *bin_plus_x)
{
	for (unsigned b = 0; b < n_bins; ++b)
		bin_plus_x[b] = 0;
	const u32 tiles = (u32)((n_sk + S1_SK_TILE - 1) / S1_SK_TILE);
	if (tiles)
		hipemu::launch(dim3(tiles), dim3(256), 0, [&] { k_s1_bin_plus_x(codes, sk_pos, sk_len, sk_sig, n_sk, k, max_x, both_strands, sig_to_bin, n_bins, bin_plus_x); });
}

/* the alternative emit: keys -> (stable sort by bin, on the host here: the radix passes have their own tests) -> k_s1_emit_sorted */
EMU_API int emu_s1_scatter_sorted(const int8_t *codes, const u64 *sk_pos, const u32 *sk_len, const u32 *sk_sig, u64 n_sk, unsigned k, const int *sig_to_bin, unsigned n_bins,
                                  u64 *bin_base, u64 *pack_base, u64 *totals, uint8_t *out, u64 out_cap, u64 *pack_start, u64 pack_cap)
{
	u32 err = 0, ticket = 0;
	std::vector<u64> cursor(n_bins, 0), cum(n_bins + 1, 0), keys(n_sk + 1, 0);
	for (u64 i = 0; i < 3ull * n_bins; ++i)
		totals[i] = 0;
	const u32 tiles = (u32)((n_sk + S1_SK_TILE - 1) / S1_SK_TILE);
	if (tiles)
		hipemu::launch(dim3(tiles), dim3(256), 0,
		               [&] { k_s1_bin_totals(sk_len, sk_sig, n_sk, k, sig_to_bin, n_bins, totals, totals + n_bins, totals + 2 * n_bins, &err); });
	if (err)
		return (int)err;
	hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_bin_layout(totals, n_bins, bin_base, pack_base, cursor.data(), (u64 *)nullptr); });
	if (bin_base[n_bins] > out_cap || pack_base[n_bins] > pack_cap)
		return -1;
	hipemu::launch(dim3(1), dim3(256), 0, [&] { k_s1_bin_layout(totals, n_bins, bin_base, pack_base, cursor.data(), pack_start); });
	if (!n_sk)
		return 0;
	for (unsigned b = 0; b < n_bins; ++b)
		cum[b + 1] = cum[b] + totals[b];
	hipemu::launch(dim3((u32)((n_sk + 255) / 256)), dim3(256), 0, [&] { k_s1_sort_keys(sk_sig, n_sk, sig_to_bin, n_bins, keys.data(), &err); });
	std::stable_sort(keys.begin(), keys.begin() + n_sk, [](u64 a, u64 b) { return (a & 0xFFFFu) < (b & 0xFFFFu); });
	const u32 et = (u32)((n_sk + S1_TILE - 1) / S1_TILE);
	std::vector<u64> status(et, 0);
	hipemu::launch(dim3(et), dim3(S1_BLOCK), 0, [&] {
		k_s1_emit_sorted(keys.data(), n_sk, codes, sk_pos, sk_len, k, n_bins, bin_base, pack_base, cum.data(), status.data(), &ticket, out, pack_start, &err);
	});
	return (int)err;
}
}
