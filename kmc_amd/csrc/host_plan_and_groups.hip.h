/* kmc_amd/csrc/host_plan_and_groups.hip.h — part of kmc_hip.hip (included there, not compiled on its own): the sort planner, the HBM passes, the groups of bins (front end, LSD / bucket-count / rank finishers), the redo of flagged groups. */
/* ---- zero-region planning --------------------------------------------------------------------------------------- */
/* Small LUTs are sharded: in sorted order every tile in flight updates the same one or two entries, and same-address device atomics
 * serialise (r01: 11 of the compaction's 12 ms at 64 entries). From 4^6 entries on the tiles in flight spread over enough of them.
 * The shards are summed by k_compact_fold (one small workgroup per bin), so shards x entries stays small (<= 8 K loads). */
u32 lut_shards_for(u64 lut_entries) { return lut_entries <= 256 ? 32u : (lut_entries <= 1024 ? 8u : 1u); }

/* one bin's share of a group (a bin on its own is a group of one): its buffers and where it sits in the zero region and the shared arrays */
struct BinPlan {
	const uint8_t *d_in = nullptr;
	u64 size = 0, n_rec = 0, n_packs = 0;
	const u64 *d_pack_start = nullptr;
	uint8_t *d_out = nullptr;
	u64 out_capacity = 0;
	u64 *d_out_bytes = nullptr, *d_lut = nullptr, *d_stats = nullptr;
	size_t off_bitmap = 0, off_exp_status = 0, off_cp_status = 0, off_lutsh = 0, off_tally = 0;
	u64 rec_off = 0; /* first record of the bin in the group's record arrays */
};

/* ---- the sort's shape ----------------------------------------------------------------------------------------------
 * key_bytes = ceil(key bits / 8) byte positions; the TOP `top` of them are sorted by 8-bit LSD passes through HBM (k_onesweep), the rest
 * inside LDS by k_bucket_rank on bucket-aligned tiles (bucket_sort.hip.h). top == key_bytes: the plain LSD sort of rounds 1-2 (sort-only calls, redo runs, tiny groups). */
struct SortPlan {
	u32 key_bytes = 0, top = 0;
	u32 key_bits = 0; /* significant bits of the key (2k + tag bits; 8 key_bytes when the caller cannot tell) */
	bool rank = false; /* the LDS half is k_bucket_rank: every tile put in order by pairwise ranking and (fused) counted in place; one-word records whose output
	                    * may outgrow a span: sorted in place, k_compact follows */
	u32 pass_lo() const { return key_bytes - top; }
	bool local() const { return top < key_bytes; }
	u32 hbits() const { return top ? 8 * top - (8 * key_bytes - key_bits) : 0; } /* top bits of the significant key that the HBM passes order (plan_sort: 8 top > spare bits) */
};
std::atomic<u64> g_path[4] = {}; /* groups by the path they took: 0 rank + count in LDS, 1 rank in place + k_compact, 2 unused since round 6 (was k_bucket_count), 3 LSD passes over every byte */
std::atomic<u64> g_indirect_groups{0}; /* ... of path 0: sorted through (key top, record number) pairs */
std::atomic<u64> g_hybrid_groups{0}, g_redo_groups{0}; /* process-wide: input whose buckets keep overflowing the tiles stops being tried */
std::atomic<u32> g_extra_top{0}; /* HBM passes added to the plan after a group came back (finer buckets for the groups after it) */
void note_redo() { g_redo_groups.fetch_add(1, std::memory_order_relaxed); }
void raise_top() /* once per drained stream / synchronous redo: the groups of one asynchronous call all come back at its end and must count once */
{
	u32 e = g_extra_top.load(std::memory_order_relaxed);
	if (e < 2)
		g_extra_top.compare_exchange_strong(e, e + 1, std::memory_order_relaxed);
}
std::atomic<int> g_hybrid_override{INT32_MIN}; /* kmc_hip_set_hybrid */
int hybrid_mode()
{
	const int o = g_hybrid_override.load(std::memory_order_relaxed);
	if (o != INT32_MIN)
		return o;
	static const int v = [] {
		/* 0 = LSD passes over every byte + k_compact (rounds 1-2); 1 = default: the top key bytes through HBM passes, every bucket-aligned tile ranked and
		 * counted inside LDS by k_bucket_rank (round 4: every record width; KMC_HIP_RANK_FUSE=0: one-word records ranked in place + k_compact, wider ones LSD);
		 * -h = force `h` top bytes (tuning). (2 — round 3's k_bucket_count / k_bucket_sort — left the library in round 6 and is taken as 1.) */
		const char *e = getenv("KMC_HIP_HYBRID");
		const int m = e ? atoi(e) : 1;
		return m >= 2 ? 1 : m;
	}();
	return v;
}
bool arena_enabled()
{
	static const bool v = [] {
		const char *e = getenv("KMC_HIP_ARENA"); /* 0 (A/B runs): round 5's finisher — big buckets walked by the whole workgroup, k_giant_tiles, bins with a satellite redone */
		return !e || atoi(e) != 0;
	}();
	return v;
}
bool rank_enabled()
{
	static const bool v = [] {
		const char *e = getenv("KMC_HIP_RANK"); /* 0: groups of one-word records keep the LSD passes over every byte */
		return !e || atoi(e) != 0;
	}();
	return v;
}
template <int SIZE>
SortPlan plan_sort(u64 n, u32 key_bytes, u32 key_bits, bool classic, bool rank = false /* the records of a group: k_bucket_rank finishes the low bytes inside LDS */)
{
	SortPlan sp;
	sp.key_bytes = sp.top = key_bytes;
	sp.key_bits = key_bits;
	const int mode = hybrid_mode();
	if (classic || mode == 0 || key_bytes < 3 || n < 2 || !rank)
		return sp; /* LSD passes over every byte: sort-only calls, redo runs, tiny inputs */
	const u32 rem_limit = br_rem_limit<SIZE>(); /* key bits that may stay below the bucket bits */
	const u32 spare = 8 * key_bytes - key_bits;
	if (mode < 0) {
		const u32 h = (u32)(-mode);
		if (h + 1 <= key_bytes && 8 * h > spare)
			sp.top = h;
		if (sp.local() && key_bits - sp.hbits() <= rem_limit && sp.hbits() <= 48)
			sp.rank = true;
		else
			sp.top = key_bytes;
		return sp;
	}
	if (n <= (u64)BrCfg<SIZE>::THREADS * (8 / SIZE > 2 ? 8 / SIZE : 2))
		return sp; /* a tiny group (a few thousand records) is bound by launches, not by bytes */
	const u64 redo = g_redo_groups.load(std::memory_order_relaxed);
	if (g_extra_top.load(std::memory_order_relaxed) >= 2 && redo > 128 && redo * 8 > g_hybrid_groups.load(std::memory_order_relaxed))
		return sp; /* this input defeats the bucket tiles again and again */
	/* the fewest top bytes that leave the finisher buckets it can rank, and only if at least two passes are saved */
	for (u32 h = 1; h + 2 <= key_bytes && h <= 6u; ++h) { /* (groups always take one pass at least: the bins' tags must be ordered) */
		const u32 eff = 8 * h > spare ? 8 * h - spare : 0;
		/* the k-mers of a signature bin that BEGIN with one of the bin's minimizers share ~18 key bits, whatever the size of the bin — 1/19 of the
		 * records in a few hundred prefixes — and the work grows with the square of a bucket: the passes must reach well below those bits (measured:
		 * 512 bins of 3.2 M k-mers with 22 key bits ordered 17.7 Gk-mers/s, the 7 LSD passes 24.1; 190 M-record groups with 22 bits 7.2, with 30 bits 31.5) */
		if (eff >= 28 && (eff >= 63 || (n >> eff) <= 2) && key_bits - eff <= rem_limit) {
			sp.top = h;
			break;
		}
	}
	if (sp.top + 2 > key_bytes)
		sp.top = key_bytes;
	/* (no finer buckets after a redo since round 4: a tile with a bucket beyond LDS goes to k_giant_tiles / the arena, and what still comes back is ONE k-mer repeated a
	 * million times — no number of passes splits that; a fifth pass would only cost every later group its time, and records of two words and more their indirect sort) */
	static const int forced = [] {
		const char *e = getenv("KMC_HIP_RANK_TOP"); /* tuning: this many top bytes through HBM */
		return e ? atoi(e) : 0;
	}();
	if (forced >= 1 && (u32)forced + 1 <= key_bytes && 8 * (u32)forced > spare)
		sp.top = (u32)forced;
	if (sp.local() && sp.top >= 1 && key_bits - sp.hbits() <= rem_limit && sp.hbits() <= 48)
		sp.rank = true;
	else
		sp.top = key_bytes; /* the pair (key bits below the bucket, index) must fit its words */
	return sp;
}

/* lays out the zero region of a group: small block | per bin: bitmap, expand look-back words, compaction look-back words, LUT shards, tally
 * shards | digit histograms | one scatter status area per onesweep launch over the group's `n_total` records */
template <int SIZE>
ZeroPlan plan_group(const Slot &s, std::vector<BinPlan> &bins, u64 n_total, u32 n_pass /* passes through HBM */, bool front, bool sort, bool compact, u64 lut_shard_entries,
                    u64 cp_tile = CpCfg<SIZE>::TILE /* records per compaction tile: k_compact's, or the window of k_bucket_rank */,
                    u32 cp_words = 1 /* status words per tile (k_bucket_rank: one per chunk, two chunks) */, u64 giant_entries = 0, bool arena = false)
{
	ZeroPlan z;
	size_t off = up256(SM_BYTES);
	if (giant_entries) {
		z.giant = off;
		off += up256((size_t)(giant_entries + 2) * 4);
	}
	if (arena) {
		z.arena = off;
		off += up256((size_t)AR_DYN_WORDS * 4) + up256((size_t)AR_MAX_PASS * 256 * 8);
	}
	for (BinPlan &b : bins) {
		if (front) {
			b.off_bitmap = off;
			off += up256(((b.size + 31) / 32 + 2) * 4);
			b.off_exp_status = off;
			off += up256(((b.size + EXP_CHUNK - 1) / EXP_CHUNK) * 8 + 8);
		}
		if (compact) {
			b.off_cp_status = off;
			off += up256(((b.n_rec + cp_tile - 1) / cp_tile) * 8 * cp_words + 8);
			b.off_lutsh = off;
			off += up256(lut_shard_entries * 8); /* n_shards x entries when the LUT is sharded */
			b.off_tally = off;
			off += up256(CP_SHARDS * 4 * 8);
		}
	}
	if (front || sort) {
		z.ghist = off;
		off += up256((size_t)n_pass * 256 * 8);
	}
	if (sort && n_total >= 2) {
		const u64 max_tiles = (std::min(n_total, s.portion) + RsCfg<SIZE>::TILE - 1) / RsCfg<SIZE>::TILE;
		const u64 n_launch = (u64)n_pass * ((n_total + s.portion - 1) / s.portion);
		z.sc_status = off;
		z.sc_stride = up256((size_t)max_tiles * 256 * 4);
		off += z.sc_stride * n_launch;
	}
	z.total = off;
	return z;
}

int apply_plan(Slot &s, const ZeroPlan &z)
{
	s.zero_grew = z.total > s.zero.cap;
	++s.groups_run;
	if (int rc = ensure(s.zero, z.total))
		return rc;
	HIPCHK(hipMemsetAsync(s.zero.p, 0, z.total, s.stream));
	return 0;
}

/* ---- the sort: histograms of the digits that go through HBM + one onesweep launch per such digit (and portion), then — hybrid — the
 * bucket-aligned LDS sort of the remaining bytes, in place (one-word records, k_bucket_rank<1, false>). `d_flag`: where a tile that could not be sorted is reported. ---- */
template <int SIZE>
int sort_device_t(Slot &s, const ZeroPlan &z, u64 *d_recs, u64 *d_tmp, u64 n, const SortPlan &sp, u64 **d_result, u32 &counter_idx, bool hist_done, u32 *d_flag,
                  bool local_by_caller = false /* the caller finishes the low bytes itself (rank_group) */)
{
	u64 *src = d_recs, *dst = d_tmp;
	if (n < 2 || sp.key_bytes == 0) {
		*d_result = src;
		if (s.timed)
			HIPCHK(hipEventRecord(s.ev[3], s.stream));
		return 0;
	}
	const u32 n_pass = sp.top, pass_lo = sp.pass_lo();
	u32 *err = err_ptr(s);
	if (n_pass) {
		if (int rc = ensure(s.dbase, (size_t)n_pass * 256 * 8))
			return rc;
		u64 *ghist = zero_ptr<u64>(s, z.ghist), *dbase = (u64 *)s.dbase.p;
		u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
		u64 *work = small_ptr<u64>(s, SM_DBASE_WORK);

		if (!hist_done) { /* digit bases: the expansion's last workgroup made them when the histograms were fused into it */
			u64 blocks = (n + 255) / 256;
			if (blocks > 256 * 8)
				blocks = 256 * 8; /* 8 workgroups per CU, grid-stride */
			k_hist<SIZE><<<dim3((u32)blocks), dim3(256), (size_t)n_pass * 1024, s.stream>>>(src, n, n_pass, ghist, pass_lo);
			k_hist_scan<<<dim3(n_pass), dim3(256), 0, s.stream>>>(ghist, dbase);
		}
		if (s.timed)
			HIPCHK(hipEventRecord(s.ev[3], s.stream));
		u32 launch = 0;
		for (u32 pass = 0; pass < n_pass; ++pass) {
			const u64 *base_in = dbase + (size_t)pass * 256;
			int flip = 0;
			for (u64 start = 0; start < n; start += s.portion) {
				const u32 cnt = (u32)std::min(s.portion, n - start);
				const u32 tiles = (cnt + RsCfg<SIZE>::TILE - 1) / RsCfg<SIZE>::TILE;
				if (counter_idx >= N_COUNTERS)
					return fail(KMC_HIP_EINVAL, "too many scatter launches for one bin");
				u32 *status = zero_ptr<u32>(s, z.sc_status + (size_t)launch * z.sc_stride);
				u64 *base_out = work + (size_t)flip * 256;
				hipEvent_t e0 = nullptr, e1 = nullptr;
				if (s.timed) {
					if (int rc = sc_event_pair(s, e0, e1, cnt))
						return rc;
					HIPCHK(hipEventRecord(e0, s.stream));
				}
				/* experiment (round 6): dynamic LDS beyond what the kernel uses = ONE scatter workgroup per CU instead of two, so that with several groups in
				 * flight a finisher workgroup (12 waves, 72 KB) of another group fits beside it (16 waves, 59 + 24 KB) */
				static const size_t scatter_pad = [] {
					const char *e = getenv("KMC_HIP_SCATTER_LDS_PAD");
					const size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0;
					return v > 32 * 1024 ? (size_t)32 * 1024 : v;
				}();
				k_onesweep<SIZE><<<dim3((tiles + RS_TPB - 1) / RS_TPB), dim3(RS_BLOCK), rs_lds_bytes<SIZE>() + scatter_pad, s.stream>>>(
				    src + start * SIZE, dst, cnt, pass_lo + pass, base_in, base_out, status, counters + counter_idx, tiles, err);
				if (s.timed)
					HIPCHK(hipEventRecord(e1, s.stream));
				++counter_idx;
				++launch;
				base_in = base_out;
				flip ^= 1;
			}
			std::swap(src, dst);
		}
	} else if (s.timed)
		HIPCHK(hipEventRecord(s.ev[3], s.stream));
	if (sp.local() && !local_by_caller) { /* one-word records whose output may outgrow a tile's span: the tiles sorted in place, the caller's k_compact follows */
		if constexpr (SIZE == 1) {
			if (!sp.rank)
				return fail(KMC_HIP_EINTERNAL, "a local sort plan without the rank finisher");
			const u64 S = (u64)BrCfg<1>::STRIDE;
			const u64 n_win = (n + S - 1) / S;
			if (n_win > 0x7FFFFFF0ull)
				return fail(KMC_HIP_EINVAL, "bin too large");
			if (int rc = ensure(s.bounds, (size_t)(n_win + 2) * 8))
				return rc;
			u64 *bounds = (u64 *)s.bounds.p;
			hipEvent_t e0 = nullptr, e1 = nullptr;
			if (s.timed) {
				if (int rc = ls_event_pair(s, e0, e1, n))
					return rc;
				HIPCHK(hipEventRecord(e0, s.stream));
			}
			GrpBounds gbn = {};
			gbn.g = 1;
			gbn.item_prefix[1] = (u32)(n_win + 1);
			gbn.S[0] = src;
			gbn.n[0] = n;
			gbn.bounds[0] = bounds;
			k_bucket_bounds<1><<<dim3((u32)((n_win + 1 + 3) / 4)), dim3(256), 0, s.stream>>>(gbn, (u32)S, sp.key_bits, sp.hbits());
			GrpRank gr = {}; /* the whole array as one "bin" */
			gr.g = 1;
			gr.win_prefix[1] = (u32)n_win;
			gr.S[0] = src;
			gr.bounds[0] = bounds;
			k_bucket_rank<1, false><<<dim3((u32)n_win, 2), dim3(BrCfg<1>::THREADS), br_lds_bytes<1>(), s.stream>>>(gr, DevParams{}, sp.key_bits, sp.hbits(), 1u, 0ull, 0u, d_flag);
			if (s.timed)
				HIPCHK(hipEventRecord(e1, s.stream));
		} else
			return fail(KMC_HIP_EINTERNAL, "records of two words and more have no in-place local sort");
	}
	HIPCHK(hipGetLastError());
	*d_result = src;
	return 0;
}

/* sort-only calls (narrow boundary, stage 1's key sort). `stable_lsd`: the caller's records carry payload above the key and rely on the LSD
 * passes' stability (kmc_hip_split_part sorts (index << 16) | bin by its low 2 bytes) — the hybrid sort compares whole records. */
template <int SIZE> int sort_only_t(Slot &s, u64 *d_recs, u64 *d_tmp, u64 n, u32 key_bytes, u64 **d_result, bool stable_lsd)
{
	std::vector<BinPlan> none;
	const SortPlan sp = plan_sort<SIZE>(n, key_bytes, 8 * key_bytes, stable_lsd);
	const ZeroPlan z = plan_group<SIZE>(s, none, n, sp.top, false, true, false, 0);
	if (int rc = apply_plan(s, z))
		return rc;
	u32 counter_idx = 0;
	return sort_device_t<SIZE>(s, z, d_recs, d_tmp, n, sp, d_result, counter_idx, false, small_ptr<u32>(s, SM_REDO));
}

int sort_device(Slot &s, u64 *d_recs, u64 *d_tmp, u64 n, u32 words, u32 n_pass, u64 **d_result, bool stable_lsd)
{
	switch (words) {
	case 1: return sort_only_t<1>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 2: return sort_only_t<2>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 3: return sort_only_t<3>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 4: return sort_only_t<4>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 5: return sort_only_t<5>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 6: return sort_only_t<6>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 7: return sort_only_t<7>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	case 8: return sort_only_t<8>(s, d_recs, d_tmp, n, n_pass, d_result, stable_lsd);
	}
	return fail(KMC_HIP_EINVAL, "words must be 1..8");
}

int check_params(const kmc_hip_bin_params *p, DevParams &P)
{
	if (!p)
		return fail(KMC_HIP_EINVAL, "params == NULL");
	if (p->kmer_len < 1 || p->kmer_len > 256)
		return fail(KMC_HIP_EINVAL, "kmer_len must be 1..256");
	if (p->output_type > 1)
		return fail(KMC_HIP_EINVAL, "output_type must be 0 (KMC) or 1 (KFF)");
	if (p->lut_prefix_len >= p->kmer_len && p->lut_prefix_len)
		return fail(KMC_HIP_EINVAL, "lut_prefix_len must be < kmer_len");
	if (p->lut_prefix_len > 15)
		return fail(KMC_HIP_EINVAL, "lut_prefix_len must be <= 15");
	if (p->lut_prefix_len && (p->kmer_len - p->lut_prefix_len) % 4)
		return fail(KMC_HIP_EINVAL, "(kmer_len - lut_prefix_len) must be a multiple of 4 (kmc.h:1454-1456)");
	P.k = p->kmer_len;
	P.both_strands = p->both_strands ? 1 : 0;
	P.cutoff_min = p->cutoff_min;
	P.cutoff_max = (u32)p->cutoff_max; /* kb_sorter.h:186 */
	P.counter_max = (u32)p->counter_max;
	P.lut_prefix_len = p->lut_prefix_len;
	P.sbytes = kmc_suffix_bytes(p->kmer_len, p->lut_prefix_len);
	P.cbytes = counter_bytes(p->cutoff_max, p->counter_max);
	P.kff = p->output_type == 1;
	P.without_output = p->without_output ? 1 : 0;
	return 0;
}

/* ---- front end of a group: mark super-k-mer starts (one workgroup per pack of any bin), then expand slice-parallel (one ticket space over
 * the slices of all bins) with the sort's histograms fused in; the last workgroup turns the histograms into digit bases ---- */
template <int SIZE>
int front_end_group(Slot &s, const std::vector<BinPlan> &bins, size_t off_ghist, const DevParams &P, u32 n_pass /* digits through HBM */, u32 pass_lo, u32 &counter_idx,
                    bool &hist_done, u64 *d_recs, bool fuse, u64 *d_pairs = nullptr /* indirect sort: (top key bytes, record number) per record */)
{
	if (bins.empty())
		return 0;
	if (bins.size() > (size_t)GRP_MAX)
		return fail(KMC_HIP_EINVAL, "group too large");
	u32 *err = err_ptr(s);
	u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
	if (counter_idx + 2 > N_COUNTERS)
		return fail(KMC_HIP_EINVAL, "too many launches for one bin");
	GrpParse gp = {};
	GrpExpand ge = {};
	gp.g = ge.g = (u32)bins.size();
	u64 packs = 0, chunks = 0;
	const u32 tag_shift = (2 * P.k) & 63;
	for (size_t i = 0; i < bins.size(); ++i) {
		const BinPlan &b = bins[i];
		gp.pack_prefix[i] = (u32)packs;
		ge.chunk_prefix[i] = (u32)chunks;
		packs += b.n_packs;
		chunks += (b.size + EXP_CHUNK - 1) / EXP_CHUNK;
		if (packs > 0x7FFFFFF0ull || chunks > 0x7FFFFFF0ull)
			return fail(KMC_HIP_EINVAL, "bin too large");
		gp.data[i] = ge.data[i] = b.d_in;
		gp.pack_start[i] = b.d_pack_start;
		gp.bitmap[i] = zero_ptr<u32>(s, b.off_bitmap);
		ge.bitmap[i] = gp.bitmap[i];
		ge.size[i] = b.size;
		ge.n_rec[i] = b.n_rec;
		ge.out[i] = d_recs + b.rec_off * SIZE;
		ge.pair_out[i] = d_pairs ? d_pairs + b.rec_off : nullptr;
		ge.status[i] = zero_ptr<u64>(s, b.off_exp_status);
		ge.tag[i] = (u64)i << tag_shift; /* 0 for a group of one */
	}
	gp.pack_prefix[bins.size()] = (u32)packs;
	ge.chunk_prefix[bins.size()] = (u32)chunks;
	ge.pair_base = d_pairs;
	u64 *ghist = zero_ptr<u64>(s, off_ghist);
	k_parse_packs<<<dim3((u32)packs), dim3(PARSE_BLOCK), 0, s.stream>>>(gp, P.k, err);
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[1], s.stream));
	const u32 blocks = (u32)std::min<u64>(chunks, 256 * 4 * (512 / EXP_BLOCK)); /* persistent workgroups, up to 4 per CU */
	if (fuse) { /* LDS: 1 KB of counters per pass next to the slice state */
		if (int rc = ensure(s.dbase, (size_t)n_pass * 256 * 8))
			return rc;
		k_expand<SIZE, true><<<dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<true>(n_pass, P.k), s.stream>>>(
		    ge, P.k, P.both_strands, n_pass, ghist, counters + counter_idx, err, (u64 *)s.dbase.p, counters + counter_idx + 1, pass_lo);
	} else
		k_expand<SIZE, false><<<dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<false>(n_pass, P.k), s.stream>>>(
		    ge, P.k, P.both_strands, n_pass, ghist, counters + counter_idx, err, nullptr, counters + counter_idx + 1, pass_lo);
	counter_idx += 2;
	hist_done = fuse;
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[2], s.stream));
	HIPCHK(hipGetLastError());
	return 0;
}

/* ---- compaction of a group (one ticket space over the tiles of all bins, each bin on its slice of the sorted array) + the fold of tally /
 * LUT shards (one small workgroup per bin) ---- */
/* `scratch`: the record array the sort left free (same layout as `sorted`), or NULL. With it, and when a tile's span of it is certain to hold
 * the tile's counted records — at most TILE / cutoff_min + 1 of them — the output is written in two phases (kernels.hip.h k_compact two_phase):
 * no tile waits for its offset. */
template <int SIZE>
int compact_group(Slot &s, const std::vector<BinPlan> &bins, const u64 *sorted, u64 *scratch, const DevParams &P, u64 lut_entries, u32 &counter_idx)
{
	if (bins.empty())
		return 0;
	u32 *err = err_ptr(s);
	u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
	if (counter_idx >= N_COUNTERS)
		return fail(KMC_HIP_EINVAL, "too many launches for one bin");
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = !use_lut ? 1u : lut_shards_for(lut_entries);
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const u64 tile_pitch = (u64)CpCfg<SIZE>::TILE * SIZE * 8;
	static const bool allow_two_phase = [] {
		const char *e = getenv("KMC_HIP_TWO_PHASE"); /* 0 = always the look-back */
		return !e || atoi(e) != 0;
	}();
	const bool two_phase = allow_two_phase && scratch && !P.without_output &&
	                       ((u64)CpCfg<SIZE>::TILE / std::max<u32>(P.cutoff_min, 1) + 1) * rec_bytes <= tile_pitch;
	GrpCompact gc = {};
	GrpFold gf = {};
	GrpGather gg = {};
	gc.g = gg.g = (u32)bins.size();
	u64 tiles = 0;
	for (size_t i = 0; i < bins.size(); ++i) {
		const BinPlan &b = bins[i];
		gc.tile_prefix[i] = gg.tile_prefix[i] = (u32)tiles;
		const u64 bin_tiles = (b.n_rec + CpCfg<SIZE>::TILE - 1) / CpCfg<SIZE>::TILE;
		tiles += bin_tiles;
		if (tiles > 0x7FFFFFFFull)
			return fail(KMC_HIP_EINVAL, "bin too large");
		u64 *lut_base = b.d_lut;
		if (use_lut && n_sh > 1)
			lut_base = zero_ptr<u64>(s, b.off_lutsh); /* zeroed with the rest of the zero region */
		else if (lut_entries && !P.kff && b.d_lut) /* also without output: the caller's LUT is zero-filled by the callee (include/kmc_hip.h) */
			HIPCHK(hipMemsetAsync(b.d_lut, 0, lut_entries * 8, s.stream));
		gc.S[i] = sorted + b.rec_off * SIZE;
		gc.n[i] = gf.n[i] = b.n_rec;
		gc.out[i] = b.d_out;
		gc.out_capacity[i] = b.out_capacity;
		gc.lut_base[i] = lut_base;
		gc.tally[i] = zero_ptr<u64>(s, b.off_tally);
		gc.out_bytes[i] = b.d_out_bytes;
		gc.status[i] = zero_ptr<u64>(s, b.off_cp_status);
		gf.tally[i] = gc.tally[i];
		gf.stats[i] = b.d_stats;
		gf.lut_base[i] = lut_base;
		gf.lut_out[i] = b.d_lut;
		gc.scratch[i] = two_phase ? (uint8_t *)(scratch + b.rec_off * SIZE) : nullptr;
		gf.status[i] = gc.status[i];
		gf.n_tiles[i] = (u32)bin_tiles;
		gf.out_bytes[i] = b.d_out_bytes;
		gf.out_capacity[i] = b.out_capacity;
		gg.scratch[i] = gc.scratch[i];
		gg.prefix[i] = gc.status[i];
		gg.out[i] = b.d_out;
		gg.out_capacity[i] = b.out_capacity;
	}
	gc.tile_prefix[bins.size()] = gg.tile_prefix[bins.size()] = (u32)tiles;
	k_compact<SIZE><<<dim3((u32)tiles), dim3(CP_BLOCK), 0, s.stream>>>(gc, P, n_sh, lut_entries, counters + counter_idx, err,
	                                                                   P.lut_prefix_len ? (u32)((1ull << (2 * P.lut_prefix_len)) - 1) : 0u, two_phase ? 1u : 0u);
	counter_idx += 1;
	k_compact_fold<<<dim3((u32)bins.size()), dim3(256), 0, s.stream>>>(gf, use_lut ? n_sh : 1u, lut_entries, two_phase ? 1u : 0u, rec_bytes, err);
	if (two_phase)
		k_compact_gather<<<dim3((u32)((tiles + 3) / 4)), dim3(256), 0, s.stream>>>(gg, rec_bytes, tile_pitch);
	HIPCHK(hipGetLastError());
	return 0;
}

/* ---- may a tile be counted where it lies (k_bucket_rank fused)? A tile's (suffix, counter) records go to the tile's span of the free record array; then the fold and
 * the gather of the two-phase output as after k_compact. ---- */
template <int SIZE> bool count_applicable(const DevParams &P)
{
	static const bool allow_two_phase = [] {
		const char *e = getenv("KMC_HIP_TWO_PHASE");
		return !e || atoi(e) != 0;
	}();
	/* a tile of L records counts at most L / cutoff_min k-mers, and its span of the free array has 8 SIZE bytes per record (+ 3 bytes of dword padding,
	 * inside the span as long as a stored record is not longer than that) */
	const u32 rec_bytes = P.sbytes + P.cbytes;
	return allow_two_phase && (P.without_output || rec_bytes <= (u32)(SIZE * 8));
}
/* ---- rank groups (default since round 4): the array is ordered by its top bytes only; k_bucket_rank puts every bucket-aligned tile of every bin in order
 * inside LDS and counts it there, straight into the tile's span of the free record array; then the fold and the gather of the two-phase output. A tile has
 * two output slots (one per chunk: a tile that outgrows the capacity is taken by two workgroups). ---- */
template <int SIZE>
int rank_group(Slot &s, const std::vector<BinPlan> &bins, u64 *sorted, u64 *scratch, const DevParams &P, u64 lut_entries, const SortPlan &sp, u64 n_total, u32 *d_flag,
               u32 *d_giant, const u64 *d_recs_indirect = nullptr /* indirect sort: `sorted` is the ordered PAIR array (one word per record), the records are here */,
               u32 *d_arena = nullptr /* one-word records: the arena's words in the zero region (plan_group), the buckets beyond BR_MID records go through arena_sort.hip.h */)
{
	if (bins.empty())
		return 0;
	u32 *err = err_ptr(s);
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = !use_lut ? 1u : lut_shards_for(lut_entries);
	const u32 rec_bytes = P.sbytes + P.cbytes;
	constexpr u64 S = BrCfg<SIZE>::STRIDE;
	GrpBounds gbn = {};
	GrpRank gr = {};
	GrpFold gf = {};
	GrpGather gg = {};
	gbn.g = gr.g = gg.g = (u32)bins.size();
	u64 wins = 0, items = 0;
	for (const BinPlan &b : bins) {
		items += (b.n_rec + S - 1) / S + 1;
		wins += (b.n_rec + S - 1) / S;
	}
	if (items > 0x3FFFFFF0ull)
		return fail(KMC_HIP_EINVAL, "bin too large");
	if (int rc = ensure(s.bounds, (size_t)(items + 2 + 2 * wins) * 8)) /* tile boundaries of every bin, then the chunks' source offsets */
		return rc;
	u64 *bounds = (u64 *)s.bounds.p, *chunk_src = bounds + items + 2;
	items = wins = 0;
	for (size_t i = 0; i < bins.size(); ++i) {
		const BinPlan &b = bins[i];
		const u64 bin_wins = (b.n_rec + S - 1) / S;
		gbn.item_prefix[i] = (u32)items;
		gr.win_prefix[i] = (u32)wins;
		gg.tile_prefix[i] = (u32)(2 * wins);
		u64 *lut_base = b.d_lut;
		if (use_lut && n_sh > 1)
			lut_base = zero_ptr<u64>(s, b.off_lutsh);
		else if (lut_entries && !P.kff && b.d_lut) /* also without output: the caller's LUT is zero-filled by the callee (include/kmc_hip.h) */
			HIPCHK(hipMemsetAsync(b.d_lut, 0, lut_entries * 8, s.stream));
		gbn.S[i] = sorted + b.rec_off * (d_recs_indirect ? 1 : SIZE);
		gr.S[i] = sorted + b.rec_off * (d_recs_indirect ? 1 : SIZE);
		gbn.n[i] = gf.n[i] = b.n_rec;
		gbn.bounds[i] = bounds + items;
		gr.bounds[i] = bounds + items;
		gr.scratch[i] = (uint8_t *)(scratch + b.rec_off * SIZE);
		gr.giant_T[i] = d_recs_indirect ? (u64 *)s.recC.p + b.rec_off * SIZE : nullptr; /* k_giant_tiles sorts records in place: it gathers a listed tile's records here first */
		gr.status[i] = zero_ptr<u64>(s, b.off_cp_status);
		gr.chunk_src[i] = chunk_src + 2 * wins;
		gr.lut_base[i] = lut_base;
		gr.tally[i] = zero_ptr<u64>(s, b.off_tally);
		gf.tally[i] = gr.tally[i];
		gf.stats[i] = b.d_stats;
		gf.lut_base[i] = lut_base;
		gf.lut_out[i] = b.d_lut;
		gf.status[i] = gr.status[i];
		gf.n_tiles[i] = (u32)(2 * bin_wins);
		gf.out_bytes[i] = b.d_out_bytes;
		gf.out_capacity[i] = b.out_capacity;
		gg.scratch[i] = gr.scratch[i];
		gg.prefix[i] = gr.status[i];
		gg.out[i] = b.d_out;
		gg.out_capacity[i] = b.out_capacity;
		gg.src_rec[i] = gr.chunk_src[i];
		items += bin_wins + 1;
		wins += bin_wins;
	}
	gbn.item_prefix[bins.size()] = (u32)items;
	gr.win_prefix[bins.size()] = (u32)wins;
	/* a giant-list entry keeps the tile's number in GT_CUT_SHIFT bits (bucket_sort.hip.h): a group with more tiles (a single bin beyond ~2.9 G records, as a group of
	 * one) gets no list — a tile with a bucket beyond LDS then raises its bin's redo flag, as tiles beyond GT_MAX_RECORDS do */
	const bool giant_list = wins < (1ull << GT_CUT_SHIFT);
	gr.giant = giant_list ? d_giant : nullptr;
	gr.rec_base = d_recs_indirect;
	gg.tile_prefix[bins.size()] = (u32)(2 * wins);
	/* The arena (one-word records): every bucket beyond BR_MID records is disjoint from every other, a giant one owns a tile — n_total / BR_MID + wins entries can never be
	 * exceeded. Work area: entries | arena_off | item_off | bucket numbers | digit bases | look-back rows of up to AR_MAX_PASS passes over up to n_total records. */
	ArenaWork aw = {};
	u32 ar_pass_max = 0;
	const u32 rbits = sp.key_bits - sp.hbits();
	if constexpr (SIZE == 1) {
		if (d_arena && wins < (1ull << 30) && n_total < (1ull << 40)) {
			static const u64 cap_limit = [] { /* tests: a list too short for the group's buckets — the plan is dropped, the group comes back through LSD passes */
				const char *e = getenv("KMC_HIP_ARENA_CAP");
				return e ? (u64)strtoull(e, nullptr, 10) : ~0ull;
			}();
			const u64 cap = std::min<u64>(n_total / BR_MID + wins + 16, cap_limit);
			const u64 max_tiles = (std::min<u64>(n_total, AR_MAX_RECORDS) + RsCfg<1>::TILE - 1) / RsCfg<1>::TILE;
			u32 obits = 0;
			while (obits < 32 && (1ull << obits) < cap)
				++obits;
			ar_pass_max = std::min<u32>(AR_MAX_PASS, (rbits + obits + 7) / 8);
			size_t off = 0;
			const size_t o_ent = off;
			off += up256(cap * sizeof(ArenaEntry));
			const size_t o_aoff = off;
			off += up256((cap + 1) * 4);
			const size_t o_ioff = off;
			off += up256((cap + 1) * 4);
			const size_t o_hi = off;
			off += up256(cap * 8);
			const size_t o_dbase = off;
			off += up256((size_t)(AR_MAX_PASS + 1) * 256 * 8);
			const size_t o_status = off;
			const size_t stride = up256(max_tiles * 256 * 4);
			off += stride * ar_pass_max;
			if (int rc = ensure(s.arena_work, off))
				return rc;
			if (int rc = ensure(s.pairA, n_total * 8 + 256))
				return rc;
			if (int rc = ensure(s.pairB, n_total * 8 + 256))
				return rc;
			char *base = (char *)s.arena_work.p;
			gr.arena_dyn = d_arena;
			gr.arena_ent = (ArenaEntry *)(base + o_ent);
			gr.arena_cap = (u32)std::min<u64>(cap, 0xFFFFFFF0ull);
			aw.arena_off = (u32 *)(base + o_aoff);
			aw.item_off = (u32 *)(base + o_ioff);
			aw.bucket_hi = (u64 *)(base + o_hi);
			aw.A = (u64 *)s.pairA.p;
			aw.B = (u64 *)s.pairB.p;
			aw.ghist = (u64 *)((char *)d_arena + up256((size_t)AR_DYN_WORDS * 4));
			aw.dbase = (u64 *)(base + o_dbase);
			aw.status = (u32 *)(base + o_status);
			aw.status_stride = (u32)(stride / 4);
			aw.S0 = gr.S[0];
		}
	}
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (s.timed) {
		if (int rc = ls_event_pair(s, e0, e1, n_total))
			return rc;
		HIPCHK(hipEventRecord(e0, s.stream));
	}
	if (d_recs_indirect) /* a pair's top half is the bucket number */
		k_bucket_bounds<1><<<dim3((u32)((items + 3) / 4)), dim3(256), 0, s.stream>>>(gbn, (u32)S, 64u, 32u);
	else
		k_bucket_bounds<SIZE><<<dim3((u32)((items + 3) / 4)), dim3(256), 0, s.stream>>>(gbn, (u32)S, sp.key_bits, sp.hbits());
	const u32 lut_mask = P.lut_prefix_len ? (u32)((1ull << (2 * P.lut_prefix_len)) - 1) : 0u;
	static const size_t lds_pad = [] { /* experiments: more dynamic LDS than the kernel uses = fewer workgroups per CU (room for another stream's kernels beside it) */
		const char *e = getenv("KMC_HIP_RANK_LDS_PAD");
		const size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0;
		return v > 32 * 1024 ? (size_t)32 * 1024 : v;
	}();
	bool arena_ran = false;
	if constexpr (SIZE == 1) {
		if (gr.arena_dyn) { /* the buckets beyond BR_MID records, sorted together (arena_sort.hip.h); nothing listed: launches that return */
			arena_ran = true;
			u32 *dyn = gr.arena_dyn;
			auto grid_of = [](const char *name) -> u32 { /* tuning / debugging: persistent workgroups per kernel */
				const char *e = getenv(name);
				const int v = e ? atoi(e) : 0;
				return v > 0 ? (u32)v : 256u * 2u;
			};
			static const u32 g_gather = grid_of("KMC_HIP_ARENA_GRID_GATHER"), g_sweep = grid_of("KMC_HIP_ARENA_GRID_SWEEP"), g_finish = grid_of("KMC_HIP_ARENA_GRID_FINISH");
			GrpDetect gd = {};
			gd.g = gr.g;
			u64 blocks = 0;
			for (size_t i = 0; i < bins.size(); ++i) {
				gd.blk_prefix[i] = (u32)blocks;
				gd.n[i] = bins[i].n_rec;
				blocks += (bins[i].n_rec + 64 * BD_STRIDE - 1) / (64 * BD_STRIDE);
			}
			gd.blk_prefix[bins.size()] = (u32)blocks;
			k_bucket_detect<<<dim3((u32)((blocks + 3) / 4)), dim3(256), 0, s.stream>>>(gr, gd, rbits);
			k_arena_plan<<<dim3(1), dim3(AR_THREADS), 0, s.stream>>>(gr, aw, rbits, d_flag);
			k_arena_gather<<<dim3(g_gather), dim3(AR_THREADS), (size_t)ar_pass_max * 1024, s.stream>>>(gr, aw, rbits, ar_pass_max);
			k_hist_scan<<<dim3(ar_pass_max), dim3(256), 0, s.stream>>>(aw.ghist, aw.dbase);
			static const bool debug_steps = getenv("KMC_HIP_ARENA_DEBUG") && atoi(getenv("KMC_HIP_ARENA_DEBUG")) >= 2;
			std::vector<u64> dbg_prev;
			auto debug_step = [&](int pass) -> int { /* diagnostics: the arena after the gather (pass -1) / after pass `pass`, against what the host makes of the step's input */
				HIPCHK(hipStreamSynchronize(s.stream));
				u32 h_dyn[AR_DYN_WORDS];
				HIPCHK(hipMemcpy(h_dyn, dyn, sizeof h_dyn, hipMemcpyDeviceToHost));
				const u32 M = h_dyn[AR_M], ne = h_dyn[AR_N_ENT], np = h_dyn[AR_N_PASS];
				if (!M || pass >= (int)np)
					return 0;
				std::vector<u64> cur(M);
				HIPCHK(hipMemcpy(cur.data(), (pass & 1) ? aw.B : aw.A, (size_t)M * 8, hipMemcpyDeviceToHost)); /* pass -1 -> A (odd as a bit pattern: & 1 == 1)... handled below */
				if (pass < 0) {
					HIPCHK(hipMemcpy(cur.data(), aw.A, (size_t)M * 8, hipMemcpyDeviceToHost));
					std::vector<u32> off(ne + 1);
					std::vector<ArenaEntry> ent(ne);
					HIPCHK(hipMemcpy(off.data(), aw.arena_off, (size_t)(ne + 1) * 4, hipMemcpyDeviceToHost));
					HIPCHK(hipMemcpy(ent.data(), gr.arena_ent, (size_t)ne * sizeof(ArenaEntry), hipMemcpyDeviceToHost));
					const u64 rmask = rbits >= 64 ? ~0ull : ((1ull << rbits) - 1);
					u64 bad = 0, first = ~0ull, badoff = 0;
					std::vector<u64> slice;
					u64 run = 0;
					for (u32 e = 0; e < ne; ++e) {
						if (off[e] != run)
							++badoff;
						run += ent[e].len;
						slice.resize(ent[e].len);
						HIPCHK(hipMemcpy(slice.data(), aw.S0 + (ent[e].w0 & ((1ull << 40) - 1)), (size_t)ent[e].len * 8, hipMemcpyDeviceToHost));
						for (u32 i = 0; i < ent[e].len; ++i)
							if (off[e] + i < M && cur[off[e] + i] != (((u64)e << rbits) | (slice[i] & rmask))) {
								if (!bad)
									first = off[e] + i;
								++bad;
							}
					}
					std::vector<u64> gh((size_t)AR_MAX_PASS * 256), want((size_t)AR_MAX_PASS * 256, 0);
					HIPCHK(hipMemcpy(gh.data(), aw.ghist, gh.size() * 8, hipMemcpyDeviceToHost));
					for (u32 i = 0; i < M; ++i)
						for (u32 b = 0; b < np; ++b)
							++want[b * 256 + ((cur[i] >> (8 * b)) & 255)];
					u64 badh = 0;
					for (size_t i = 0; i < (size_t)np * 256; ++i)
						badh += gh[i] != want[i];
					fprintf(stderr, "[arena debug] gather: %llu of %u records differ from the host's (first %llu), %llu offsets off (sum %llu), %llu histogram cells differ\n", (unsigned long long)bad, M,
					        (unsigned long long)first, (unsigned long long)badoff, (unsigned long long)run, (unsigned long long)badh);
				} else {
					HIPCHK(hipMemcpy(cur.data(), (pass & 1) ? aw.A : aw.B, (size_t)M * 8, hipMemcpyDeviceToHost)); /* pass p writes B when p is even */
					std::vector<u64> want(dbg_prev);
					std::stable_sort(want.begin(), want.end(), [&](u64 a, u64 b) { return ((a >> (8 * pass)) & 255) < ((b >> (8 * pass)) & 255); });
					u64 bad = 0, first = ~0ull;
					for (u32 i = 0; i < M; ++i)
						if (cur[i] != want[i]) {
							if (!bad)
								first = i;
							++bad;
						}
					fprintf(stderr, "[arena debug] pass %d: %llu of %u records differ from a stable sort of the pass's input by its digit (first %llu)\n", pass, (unsigned long long)bad, M,
					        (unsigned long long)first);
					u32 shown = 0;
					for (u32 i = 0; i < M && shown < 12; ++i)
						if (cur[i] != want[i]) {
							u64 where = ~0ull; /* where the host expected the record the device put here */
							for (u32 j = 0; j < M; ++j)
								if (want[j] == cur[i]) {
									where = j;
									break;
								}
							fprintf(stderr, "[arena debug]   at %u: device %016llx host %016llx (the device's record belongs at %lld)\n", i, (unsigned long long)cur[i], (unsigned long long)want[i], (long long)where);
							++shown;
						}
				}
				dbg_prev.swap(cur);
				return 0;
			};
			if (debug_steps)
				if (int rc = debug_step(-1))
					return rc;
			for (u32 pass = 0; pass < ar_pass_max; ++pass) {
				if (debug_steps && pass)
					if (int rc = debug_step((int)pass - 1))
						return rc;
				static const bool debug_static = getenv("KMC_HIP_ARENA_DEBUG") && atoi(getenv("KMC_HIP_ARENA_DEBUG")) >= 3;
				if (debug_static) { /* diagnostics: the pass by the ordinary kernel, its length read back by the host */
					HIPCHK(hipStreamSynchronize(s.stream));
					u32 h_dyn[AR_DYN_WORDS];
					HIPCHK(hipMemcpy(h_dyn, dyn, sizeof h_dyn, hipMemcpyDeviceToHost));
					if (h_dyn[AR_M] && pass < h_dyn[AR_N_PASS]) {
						const u32 tiles = (h_dyn[AR_M] + RsCfg<1>::TILE - 1) / RsCfg<1>::TILE;
						k_onesweep<1><<<dim3(tiles), dim3(RS_BLOCK), rs_lds_bytes<1>(), s.stream>>>((pass & 1u) ? aw.B : aw.A, (pass & 1u) ? aw.A : aw.B, h_dyn[AR_M], pass,
						                                                                         aw.dbase + (size_t)pass * 256, aw.dbase + (size_t)AR_MAX_PASS * 256,
						                                                                         aw.status + (size_t)pass * aw.status_stride, dyn + AR_PASS_TICKET + pass, tiles, err);
					}
					continue;
				}
				k_onesweep_dyn<1><<<dim3(g_sweep), dim3(RS_BLOCK), rs_lds_bytes<1>(), s.stream>>>((pass & 1u) ? aw.B : aw.A, (pass & 1u) ? aw.A : aw.B, dyn + AR_M, pass,
				                                                                                  aw.dbase + (size_t)pass * 256, aw.dbase + (size_t)AR_MAX_PASS * 256,
				                                                                                  aw.status + (size_t)pass * aw.status_stride, dyn + AR_PASS_TICKET + pass, err);
			}
			if (debug_steps)
				if (int rc = debug_step((int)ar_pass_max - 1))
					return rc;
			static const bool debug = getenv("KMC_HIP_ARENA_DEBUG") != nullptr;
			if (debug) { /* is the arena in order, and is every entry's slice the multiset of its bucket? (the stream is drained: diagnostics only) */
				HIPCHK(hipStreamSynchronize(s.stream));
				u32 h_dyn[AR_DYN_WORDS];
				HIPCHK(hipMemcpy(h_dyn, dyn, sizeof h_dyn, hipMemcpyDeviceToHost));
				const u32 M = h_dyn[AR_M], ne = h_dyn[AR_N_ENT], np = h_dyn[AR_N_PASS];
				fprintf(stderr, "[arena debug] entries %u records %u passes %u (max %u) items %u heavy %u overflow %u rbits %u\n", ne, M, np, ar_pass_max, h_dyn[AR_N_ITEMS], 0u,
				        h_dyn[AR_OVERFLOW], rbits);
				if (M) {
					std::vector<u64> ar(M);
					std::vector<u32> off(ne + 1);
					std::vector<ArenaEntry> ent(ne);
					HIPCHK(hipMemcpy(ar.data(), (np & 1u) ? aw.B : aw.A, (size_t)M * 8, hipMemcpyDeviceToHost));
					HIPCHK(hipMemcpy(off.data(), aw.arena_off, (size_t)(ne + 1) * 4, hipMemcpyDeviceToHost));
					HIPCHK(hipMemcpy(ent.data(), gr.arena_ent, (size_t)ne * sizeof(ArenaEntry), hipMemcpyDeviceToHost));
					u64 inversions = 0, first_inv = ~0ull;
					for (u32 i = 1; i < M; ++i)
						if (ar[i - 1] > ar[i]) {
							if (!inversions)
								first_inv = i;
							++inversions;
						}
					u64 bad_entries = 0, first_bad = ~0ull, bad_ord = 0;
					const u64 rmask = rbits >= 64 ? ~0ull : ((1ull << rbits) - 1);
					std::vector<u64> slice;
					for (u32 e = 0; e < ne; ++e) {
						const u64 gpos = ent[e].w0 & ((1ull << 40) - 1);
						slice.resize(ent[e].len);
						HIPCHK(hipMemcpy(slice.data(), aw.S0 + gpos, (size_t)ent[e].len * 8, hipMemcpyDeviceToHost));
						for (auto &x : slice)
							x = ((u64)e << rbits) | (x & rmask);
						std::sort(slice.begin(), slice.end());
						bool ok = off[e + 1] - off[e] == ent[e].len;
						for (u32 i = 0; ok && i < ent[e].len; ++i)
							ok = ar[off[e] + i] == slice[i];
						for (u32 i = off[e]; i < off[e + 1] && i < M; ++i)
							if ((ar[i] >> rbits) != e)
								++bad_ord;
						if (!ok) {
							if (!bad_entries)
								first_bad = e;
							++bad_entries;
						}
					}
					fprintf(stderr, "[arena debug] inversions %llu (first at %llu), entries whose slice is not their bucket in order: %llu (first %llu), records under a foreign ordinal %llu\n",
					        (unsigned long long)inversions, (unsigned long long)first_inv, (unsigned long long)bad_entries, (unsigned long long)first_bad, (unsigned long long)bad_ord);
				}
			}
			k_arena_finish<<<dim3(g_finish * (AR_THREADS / AF_THREADS)), dim3(AF_THREADS), 0, s.stream>>>(gr, aw, P, rbits, n_sh, lut_entries, lut_mask, err);
		}
	}
	/* every tile ranked and counted inside LDS (with the arena: its buckets beyond BR_MID records are in order by now and keep their places) */
	k_bucket_rank<SIZE, true><<<dim3((u32)wins, 2), dim3(BrCfg<SIZE>::THREADS), br_lds_bytes<SIZE>() + lds_pad, s.stream>>>(gr, P, sp.key_bits, sp.hbits(), n_sh, lut_entries, lut_mask, d_flag);
	/* the tiles with a bucket beyond the LDS capacity (k-mers repeated thousands of times), one workgroup each; nothing listed: a launch that returns */
	if (!arena_ran && giant_list)
		k_giant_tiles<SIZE><<<dim3((u32)std::min<u64>(wins, 256 * (1024 / GT_THREADS))), dim3(GT_THREADS), 0, s.stream>>>(gr, P, (u32)S, sp.key_bits, sp.hbits(), n_sh, lut_entries, lut_mask, err);
	if (s.timed)
		HIPCHK(hipEventRecord(e1, s.stream));
	k_compact_fold<<<dim3((u32)bins.size()), dim3(256), 0, s.stream>>>(gf, use_lut ? n_sh : 1u, lut_entries, 1u, rec_bytes, err);
	if (!P.without_output)
		k_compact_gather<<<dim3((u32)((2 * wins + 3) / 4)), dim3(256), 0, s.stream>>>(gg, rec_bytes, (u64)SIZE * 8);
	HIPCHK(hipGetLastError());
	return 0;
}

/* ---- a group of bins, everything device resident -------------------------------------------------------------------
 * The top radix digit of a k-mer has 8 ceil(k/4) - 2k spare bits (2 at k = 27, 55, 127). Bins expanded into one record array with the bin's
 * number inside the group in those bits are put into bin-major order by the SAME number of passes one bin needs — as launches 2^spare times
 * as large (a 48 M-record launch runs at 0.46-0.47 of the HBM peak, a 190 M-record one at 0.51: fewer ramps and drains per record) and
 * 2^spare times fewer of them. Parse, expand, compaction and fold are one launch each per group as well (kernels.hip.h Grp*). A bin on its
 * own is a group of one. */
u32 group_capacity(u32 k, bool small_bins)
{
	static const int limit = [] {
		const char *e = getenv("KMC_HIP_GROUP"); /* 1 = every bin on its own */
		const int v = e ? atoi(e) : GRP_MAX;
		return v < 1 ? 1 : (v > GRP_MAX ? GRP_MAX : v);
	}();
	const u32 words = (k + 31) / 32;
	const u32 spare = 8 * ((2 * k + 7) / 8) - 2 * k; /* bits of the top digit above the k-mer: tags that cost no pass */
	const u32 room = 64 * words - 2 * k;             /* bits of the record above the k-mer */
	/* Small bins are bound by launches, not by bytes: they are grouped GRP_MAX at a time even when the tag then needs a digit of its own
	 * (one more pass over little data, and 3-4x fewer launches per bin). */
	const u32 bits = small_bins ? (room > 4 ? 4 : room) : (spare > 4 ? 4 : spare);
	const u32 cap = 1u << bits;
	return cap < (u32)limit ? cap : (u32)limit;
}
constexpr u64 GROUP_SMALL_BIN_RECORDS = 2ull << 20; /* average records per bin below which bins count as small. Measured: 512 bins of 0.48 M k-mers
                                                      * 15.7 (groups of 4) vs 18.3 Gk-mers/s (groups of 16 + one pass); 512 bins of 3.2 M k-mers 22.1 vs 21.0 */
constexpr u64 GROUP_MAX_RECORD_BYTES = 6ull << 30; /* per record array of a group */
#ifndef INDIRECT_MIN_WORDS
#define INDIRECT_MIN_WORDS 2 /* record widths (64-bit words) from which a group is sorted through (key top, record number) pairs: run_group_device_t. Measured (quarter
                              * workloads): k = 127 10.6 -> 18.1 Gk-mers/s (gathering 32-byte records costs the finisher 0.14 ms, the passes shrink from 2.03 to 0.61);
                              * k = 55 21.3 -> 22.7 (16-byte gathers waste half of every HBM sector: finisher 1.81 -> 2.91 ms, passes 3.92 -> 2.20) */
#endif

/* d_stats / d_out_bytes == NULL in a descriptor (groups of one only): the slot's own small block (host-boundary path) */
template <int SIZE>
int run_group_device_t(Slot &s, const DevParams &P, const kmc_hip_bin_desc *const *descs, u32 g, u64 lut_entries, bool classic, u32 *d_flag, bool *used_hybrid)
{
	const u32 k = P.k;
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = use_lut ? lut_shards_for(lut_entries) : 1u;
	std::vector<BinPlan> bins; /* the non-empty bins */
	u64 N = 0;
	for (u32 i = 0; i < g; ++i) {
		const kmc_hip_bin_desc &d = *descs[i];
		if ((d.n_rec == 0) != (d.size == 0))
			return fail(KMC_HIP_ECORRUPT, "exactly one of size / n_rec is zero");
		if (d.n_rec == 0)
			continue;
		if (d.n_packs == 0 || d.n_packs > 0xFFFFFFF0ull)
			return fail(KMC_HIP_EINVAL, "n_packs out of range");
		BinPlan b;
		b.d_in = d.d_superkmers;
		b.size = d.size;
		b.n_rec = d.n_rec;
		b.n_packs = d.n_packs;
		b.d_pack_start = (const u64 *)d.d_pack_start;
		b.d_out = d.d_out;
		b.out_capacity = d.out_capacity;
		b.d_out_bytes = (u64 *)d.d_out_bytes;
		b.d_lut = (u64 *)d.d_lut;
		b.d_stats = (u64 *)d.d_stats;
		b.rec_off = N;
		N += d.n_rec;
		bins.push_back(b);
	}
	/* passes: ceil(k/4) = rec_len of the plain k-mer path (kb_sorter.h:769) — plus one when the group's tags do not fit the spare bits of
	 * the top digit (groups of small bins, group_capacity) */
	u32 tag_bits = 0;
	while ((1u << tag_bits) < bins.size())
		++tag_bits;
	if (2 * k + tag_bits > 64u * SIZE)
		return fail(KMC_HIP_EINVAL, "group too large for the record width");
	const u32 key_bytes = (2 * k + tag_bits + 7) / 8;
	/* default run (round 4): only the top bytes of the key go through HBM passes, every tile is put in order inside LDS by k_bucket_rank and counted there (fused:
	 * whenever a tile's records fit its span of the free array; else, one-word records only, the tile is sorted in place and k_compact follows). Where the rank
	 * plan does not apply (too many key bits left below the buckets, tiny groups, redo runs): LSD passes over every byte + k_compact */
	static const bool fuse_enabled = [] {
		const char *e = getenv("KMC_HIP_RANK_FUSE"); /* 0 (A/B runs): one-word records ranked in place + k_compact, wider ones LSD passes over every byte */
		return !e || atoi(e) != 0;
	}();
	const bool can_fuse = count_applicable<SIZE>(P);
	const bool by_rank = !classic && hybrid_mode() == 1 && rank_enabled() && (SIZE == 1 || (can_fuse && fuse_enabled));
	const SortPlan sp = plan_sort<SIZE>(N, key_bytes, 2 * k + tag_bits, !by_rank, by_rank);
	const bool rank_fused = sp.rank && can_fuse && fuse_enabled;
	const u32 n_pass = sp.top;
	if (used_hybrid)
		*used_hybrid = sp.local() && N >= 2;
	/* the histograms of the HBM passes are fused into the expansion up to 16 of them (plain LSD: k <= 64); a bin on its own with a single record has nothing to sort */
	const bool fuse = n_pass >= 1 && n_pass <= EXP_FUSE_MAX_PASS && N >= 2;
	/* Indirect sort (end of round 4), records of INDIRECT_MIN_WORDS (two) words and more: what goes through the four HBM passes is one word per record — the key's top four
	 * bytes (exactly the digits of those passes) above the record's number in the group —, sorted by k_onesweep<1>; the records stay where k_expand wrote them and
	 * k_bucket_rank gathers each tile's records by number. Per record 4 x 16 bytes of passes + 8 written + 8 SIZE gathered instead of 4 x 16 SIZE (k = 127: ~130
	 * instead of ~290 bytes per k-mer). k_giant_tiles (which sorts a tile's records in place) first gathers a listed tile's records into its slice of a third array. */
	static const bool indirect_enabled = [] {
		const char *e = getenv("KMC_HIP_INDIRECT"); /* 0 (A/B runs): records of every width go through the passes themselves */
		return !e || atoi(e) != 0;
	}();
	bool indirect = SIZE >= INDIRECT_MIN_WORDS && indirect_enabled && rank_fused && sp.local() && n_pass == 4 && fuse && N < (1ull << 32);
	int rc = 0;
	if (N && ((rc = ensure(s.recA, N * SIZE * 8 + 256)) || (rc = ensure(s.recB, N * SIZE * 8 + 256))))
		return rc;
	/* the indirect sort wants two pair arrays and a third record array on top: when the device cannot give them, the group goes the direct way (the records
	 * through the passes, round 3's footprint) instead of failing */
	if (indirect && (ensure(s.pairA, N * 8 + 256) || ensure(s.pairB, N * 8 + 256) || ensure(s.recC, N * SIZE * 8 + 256))) {
		(void)hipGetLastError();
		indirect = false;
	}
	u64 rank_tiles = 0;
	if (rank_fused)
		for (const BinPlan &b : bins)
			rank_tiles += (b.n_rec + BrCfg<SIZE>::STRIDE - 1) / BrCfg<SIZE>::STRIDE;
	const ZeroPlan z = plan_group<SIZE>(s, bins, N, n_pass, true, true, true, n_sh > 1 ? (u64)n_sh * lut_entries : 0,
	                                    rank_fused ? (u64)BrCfg<SIZE>::STRIDE : (u64)CpCfg<SIZE>::TILE, rank_fused ? 2u : 1u,
	                                    rank_tiles, rank_fused && SIZE == 1 && arena_enabled());
	if ((rc = apply_plan(s, z))) /* ONE memset per group: small block, bitmaps, look-back words, histograms, LUT and tally shards, scatter status */
		return rc;
	for (BinPlan &b : bins) { /* resolved only AFTER apply_plan: growing the zero region moves the small block */
		if (!b.d_stats)
			b.d_stats = small_ptr<u64>(s, SM_STATS);
		if (!b.d_out_bytes)
			b.d_out_bytes = small_ptr<u64>(s, SM_OUTBYTES);
	}
	for (u32 i = 0; i < g; ++i) { /* empty bins: zero results, nothing else */
		const kmc_hip_bin_desc &d = *descs[i];
		if (d.n_rec)
			continue;
		u64 *st = d.d_stats ? (u64 *)d.d_stats : small_ptr<u64>(s, SM_STATS), *ob = d.d_out_bytes ? (u64 *)d.d_out_bytes : small_ptr<u64>(s, SM_OUTBYTES);
		HIPCHK(hipMemsetAsync(st, 0, 4 * 8, s.stream));
		HIPCHK(hipMemsetAsync(ob, 0, 8, s.stream));
		if (lut_entries && !P.without_output)
			HIPCHK(hipMemsetAsync(d.d_lut, 0, lut_entries * 8, s.stream));
	}
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[0], s.stream));
	u32 counter_idx = 0;
	bool hist_done = false;
	if ((rc = front_end_group<SIZE>(s, bins, z.ghist, P, n_pass, sp.pass_lo(), counter_idx, hist_done, (u64 *)s.recA.p, fuse, indirect ? (u64 *)s.pairA.p : nullptr)))
		return rc;
	if (n_pass == 0)
		hist_done = true; /* no HBM pass, no histogram */
	if (s.timed && bins.empty()) {
		HIPCHK(hipEventRecord(s.ev[1], s.stream));
		HIPCHK(hipEventRecord(s.ev[2], s.stream));
	}
	u64 *sorted = (u64 *)s.recA.p;
	u32 *const flag = d_flag ? d_flag : small_ptr<u32>(s, SM_REDO);
	if (indirect) { /* the pairs' bytes 4..7 are the key's bytes pass_lo .. pass_lo + 3: the same four digit histograms, the same digit bases */
		SortPlan pp;
		pp.key_bytes = 8;
		pp.top = 4;
		pp.key_bits = 64;
		if ((rc = sort_device_t<1>(s, z, (u64 *)s.pairA.p, (u64 *)s.pairB.p, N, pp, &sorted, counter_idx, hist_done, flag, true)))
			return rc;
	} else if (N && (rc = sort_device_t<SIZE>(s, z, (u64 *)s.recA.p, (u64 *)s.recB.p, N, sp, &sorted, counter_idx, hist_done, flag, !sp.rank || rank_fused)))
		return rc;
	if (s.timed) {
		if (!N)
			HIPCHK(hipEventRecord(s.ev[3], s.stream));
		HIPCHK(hipEventRecord(s.ev[4], s.stream));
	}
	u64 *const free_array = indirect ? (u64 *)s.recB.p : (N ? (sorted == (u64 *)s.recA.p ? (u64 *)s.recB.p : (u64 *)s.recA.p) : nullptr);
	if (N >= 2)
		g_path[rank_fused && sp.local() ? 0 : (sp.rank && sp.local() ? 1 : (sp.local() ? 2 : 3))].fetch_add(1, std::memory_order_relaxed);
	if (indirect && N >= 2)
		g_indirect_groups.fetch_add(1, std::memory_order_relaxed);
	if (rank_fused && sp.local() && N >= 2)
		rc = rank_group<SIZE>(s, bins, sorted, free_array, P, lut_entries, sp, N, flag, zero_ptr<u32>(s, z.giant), indirect ? (const u64 *)s.recA.p : nullptr,
		                      z.arena ? zero_ptr<u32>(s, z.arena) : nullptr);
	else
		rc = compact_group<SIZE>(s, bins, sorted, free_array, P, lut_entries, counter_idx);
	if (rc)
		return rc;
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[5], s.stream));
	HIPCHK(hipGetLastError());
	return 0;
}

/* caller holds s.mtx and has set s.timed. `classic`: LSD passes over every byte (the redo of a group whose hybrid sort reported a tile it could
 * not handle). `d_flag`: the device word that report goes to — NULL = the slot's small block (SM_REDO), which the group's memset clears. */
int run_group_device(Slot &s, const DevParams &P, const kmc_hip_bin_desc *const *descs, u32 g, u64 lut_entries, bool classic = false, u32 *d_flag = nullptr,
                     bool *used_hybrid = nullptr)
{
	bool hyb = false;
	int rc = KMC_HIP_EINVAL;
	switch ((P.k + 31) / 32) {
	case 1: rc = run_group_device_t<1>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 2: rc = run_group_device_t<2>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 3: rc = run_group_device_t<3>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 4: rc = run_group_device_t<4>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 5: rc = run_group_device_t<5>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 6: rc = run_group_device_t<6>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 7: rc = run_group_device_t<7>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	case 8: rc = run_group_device_t<8>(s, P, descs, g, lut_entries, classic, d_flag, &hyb); break;
	default: return fail(KMC_HIP_EINVAL, "kmer_len out of range");
	}
	if (!rc && hyb)
		g_hybrid_groups.fetch_add(1, std::memory_order_relaxed);
	if (used_hybrid)
		*used_hybrid = hyb;
	return rc;
}

/* ---- redo: asynchronous device-resident groups ---------------------------------------------------------------------
 * A group enqueued without a synchronisation point of its own gets a word of the slot's redo log; once the stream is idle, drain_redo reads the
 * log and sorts the flagged groups again with LSD passes over all bytes (their inputs — the bin images — are untouched, their outputs are
 * simply written again). */
constexpr u32 REDO_LOG_WORDS = 8192;
int drain_redo(Slot &s); /* below */
int run_group_async(Slot &s, const DevParams &P, const kmc_hip_bin_desc *const *descs, u32 g, u64 lut_entries)
{
	if (!s.redo_log.p) {
		if (int rc = ensure(s.redo_log, REDO_LOG_WORDS * 4))
			return rc;
		HIPCHK(hipMemsetAsync(s.redo_log.p, 0, REDO_LOG_WORDS * 4, s.stream));
	}
	if (s.pending_groups.size() >= REDO_LOG_WORDS)
		if (int rc = drain_redo(s))
			return rc;
	u32 *d_flag = (u32 *)s.redo_log.p + s.pending_groups.size();
	bool hyb = false;
	if (int rc = run_group_device(s, P, descs, g, lut_entries, false, d_flag, &hyb))
		return rc;
	Slot::PendingGroup pg;
	pg.P = P;
	pg.lut_entries = lut_entries;
	if (hyb)
		for (u32 i = 0; i < g; ++i)
			pg.descs.push_back(*descs[i]);
	s.pending_groups.push_back(std::move(pg)); /* a group sorted by LSD passes alone keeps its (never set) word, without descriptors */
	return 0;
}
/* caller holds s.mtx */
int drain_redo(Slot &s)
{
	if (s.pending_groups.empty())
		return 0;
	HIPCHK(hipStreamSynchronize(s.stream));
	std::vector<u32> log(s.pending_groups.size());
	HIPCHK(hipMemcpy(log.data(), s.redo_log.p, log.size() * 4, hipMemcpyDeviceToHost));
	std::vector<Slot::PendingGroup> groups;
	groups.swap(s.pending_groups);
	bool any = false;
	static const bool no_redo = getenv("KMC_HIP_NO_REDO") != nullptr; /* timing experiments only: flagged groups keep their (wrong) output */
	for (size_t i = 0; i < groups.size(); ++i) {
		if (!log[i] || groups[i].descs.empty())
			continue;
		if (getenv("KMC_HIP_VERBOSE")) {
			fprintf(stderr, "[kmc_hip] group %zu of %zu on this stream asked for a redo (flag %u): bins", i, groups.size(), log[i]);
			for (const auto &d : groups[i].descs)
				fprintf(stderr, " %llu", (unsigned long long)d.n_rec);
			fprintf(stderr, "\n");
		}
		if (no_redo) {
			note_redo();
			any = true;
			continue;
		}
		any = true;
		note_redo();
		/* the fused rank kernels name the BINS whose tiles they could not take (bits 16 + the bin's number among the group's non-empty bins: one k-mer beyond
		 * GT_MAX_RECORDS copies — a satellite); the other bins of the group are complete and stay as they are: each flagged bin comes back alone, a group of one
		 * through LSD passes over every byte (round 5; before, the whole group of up to 16 bins did). Any low bit: a producer that speaks for the group. */
		const u32 whole = log[i] & 0xFFFFu, mask = log[i] >> 16;
		std::vector<std::vector<const kmc_hip_bin_desc *>> runs;
		if (whole || !mask) {
			runs.emplace_back();
			for (const auto &d : groups[i].descs)
				runs.back().push_back(&d);
		} else {
			u32 j = 0;
			for (const auto &d : groups[i].descs) {
				if (!d.n_rec)
					continue;
				if ((mask >> j) & 1u)
					runs.push_back({&d});
				++j;
			}
		}
		const bool timed = s.timed;
		s.timed = false;
		int rc = 0;
		for (auto &ptrs : runs)
			if ((rc = run_group_device(s, groups[i].P, ptrs.data(), (u32)ptrs.size(), groups[i].lut_entries, true)))
				break;
		s.timed = timed;
		if (rc)
			return rc;
	}
	if (any) {
		raise_top();
		HIPCHK(hipMemsetAsync(s.redo_log.p, 0, log.size() * 4, s.stream));
		HIPCHK(hipStreamSynchronize(s.stream));
	}
	return 0;
}

/* one bin = a group of one */
int run_bin_device(Slot &s, const DevParams &P, const uint8_t *d_in, u64 size, u64 n_rec, const u64 *d_pack_start, u64 n_packs, uint8_t *d_out,
                   u64 out_capacity, u64 *d_out_bytes, u64 *d_lut, u64 lut_entries, u64 *d_stats, bool classic = false, bool async = false)
{
	kmc_hip_bin_desc d;
	d.d_superkmers = d_in;
	d.size = size;
	d.n_rec = n_rec;
	d.d_pack_start = (const uint64_t *)d_pack_start;
	d.n_packs = n_packs;
	d.d_out = d_out;
	d.out_capacity = out_capacity;
	d.d_out_bytes = (uint64_t *)d_out_bytes;
	d.d_lut = (uint64_t *)d_lut;
	d.d_stats = (uint64_t *)d_stats;
	const kmc_hip_bin_desc *p = &d;
	if (async)
		return run_group_async(s, P, &p, 1, lut_entries);
	return run_group_device(s, P, &p, 1, lut_entries, classic);
}

/* the slot's stream is idle: did the hybrid sort of the group that just ran ask for a redo (small block, SM_REDO)? */
int read_redo(Slot &s, bool &redo)
{
	u32 v = 0;
	HIPCHK(hipMemcpy(&v, small_ptr<u32>(s, SM_REDO), 4, hipMemcpyDeviceToHost));
	redo = v != 0;
	if (redo) {
		note_redo();
		raise_top();
	}
	return 0;
}

int err_to_code(u32 err)
{
	if (err & (KERR_WATCHDOG | KERR_PEER)) {
		char buf[512];
		int n = snprintf(buf, sizeof buf, "device look-back watchdog tripped (error word 0x%x:%s%s%s%s%s)", err, err & KERR_AT_SCATTER ? " scatter pass" : "",
		                 err & KERR_AT_EXPAND ? " expansion" : "", err & KERR_AT_COMPACT ? " compaction" : "", err & KERR_AT_STAGE1 ? " stage 1" : "",
		                 err & KERR_WATCHDOG ? "" : " — only the give-up of a peer, no time-out of its own: a stale bit");
		if (g_diag[0] == err && g_diag[1]) /* what the first look-back that timed out saw (kernels.hip.h lb_blocked) */
			snprintf(buf + n, sizeof buf - (size_t)n, "; first time-out: kernel bits 0x%x, lane/digit %u, tile %u of %u waited for tile %d, %u polls",
			         g_diag[2] & 0xFFFFu, g_diag[2] >> 16, g_diag[3], g_diag[10], (int)g_diag[4], g_diag[7]);
		if (g_diag[0] == err)
			strncat(buf, g_diag_slot, sizeof buf - strlen(buf) - 1);
		return fail(KMC_HIP_EINTERNAL, buf);
	}
	if (err & KERR_CORRUPT)
		return fail(KMC_HIP_ECORRUPT, "super-k-mer stream does not end on a pack boundary");
	if (err & KERR_NREC)
		return fail(KMC_HIP_ECORRUPT, "n_rec disagrees with the super-k-mer stream");
	if (err & KERR_CAPACITY)
		return fail(KMC_HIP_ECAPACITY, "out_capacity too small for the counted k-mers");
	return 0;
}

/* a bin whose record arrays exceed this fills the GPU on its own: it always takes slot 0, so that not every slot it
 * would visit keeps two arrays of that size (slot buffers only grow) */
constexpr u64 ASYNC_BIG_BYTES = 1ull << 31;
bool is_big(const DevParams &P, u64 n_rec) { return n_rec * (u64)((P.k + 31) / 32) * 8 * 2 > ASYNC_BIG_BYTES; }

} // namespace
