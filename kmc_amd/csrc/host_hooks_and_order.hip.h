/* kmc_amd/csrc/host_hooks_and_order.hip.h — part of kmc_hip.hip (included there, not compiled on its own): stage-isolating test hooks (device side) and the globally ordered database. */
/* ---- stage-isolating test hooks (tests/ use them to localise a parity failure to one kernel group) ---- */
namespace {
template <int SIZE>
int debug_expand_t(Slot &s, const DevParams &P, u64 size, u64 n_rec, u64 np)
{
	int rc = 0;
	if ((rc = ensure(s.recA, n_rec * SIZE * 8 + 256)))
		return rc;
	const u32 n_pass = (2 * P.k + 7) / 8;
	std::vector<BinPlan> bins(1);
	bins[0].d_in = (const uint8_t *)s.in.p;
	bins[0].size = size;
	bins[0].n_rec = n_rec;
	bins[0].n_packs = np;
	bins[0].d_pack_start = (const u64 *)s.pack_start.p;
	const ZeroPlan z = plan_group<SIZE>(s, bins, n_rec, n_pass, true, false, false, 0);
	if ((rc = apply_plan(s, z)))
		return rc;
	u32 counter_idx = 0;
	bool hist_done = false;
	return front_end_group<SIZE>(s, bins, z.ghist, P, n_pass, 0, counter_idx, hist_done, (u64 *)s.recA.p, n_pass <= EXP_FUSE_MAX_PASS && n_rec >= 2);
}
template <int SIZE>
int debug_compact_t(Slot &s, const DevParams &P, u64 n, u64 out_capacity, u64 lut_entries)
{
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = use_lut ? lut_shards_for(lut_entries) : 1u;
	std::vector<BinPlan> bins(1);
	bins[0].n_rec = n;
	bins[0].d_out = (uint8_t *)s.out.p;
	bins[0].out_capacity = out_capacity;
	bins[0].d_lut = (u64 *)s.lut.p;
	const ZeroPlan z = plan_group<SIZE>(s, bins, n, 0, false, false, true, n_sh > 1 ? (u64)n_sh * lut_entries : 0);
	if (int rc = apply_plan(s, z))
		return rc;
	bins[0].d_stats = small_ptr<u64>(s, SM_STATS);
	bins[0].d_out_bytes = small_ptr<u64>(s, SM_OUTBYTES);
	u32 counter_idx = 0;
	return compact_group<SIZE>(s, bins, (const u64 *)s.recA.p, nullptr, P, lut_entries, counter_idx);
}
} // namespace

/* ---- a globally ordered database on the device ---- */
namespace {
template <int SIZE>
int order_database_t(Slot &s, const DevParams &P, const kmc_hip_bin_desc *bins, const std::vector<u64> &n_of, u64 n_total, u32 p_out, uint8_t *d_out, u64 *d_lut_out)
{
	constexpr int W = SIZE + 1;
	const u32 rb_in = P.sbytes + P.cbytes;
	const u64 n_entries = 1ull << (2 * P.lut_prefix_len);
	int rc = 0;
	if ((rc = ensure(s.recA, n_total * W * 8 + 256)) || (rc = ensure(s.recB, n_total * W * 8 + 256)) || (rc = ensure(s.bounds, (n_entries + 2) * 8)))
		return rc;
	u64 *recs = (u64 *)s.recA.p, *sums = (u64 *)s.bounds.p;
	u64 off = 0;
	for (size_t b = 0; b < n_of.size(); ++b) {
		if (!n_of[b])
			continue;
		k_db_cumsum<<<dim3(1), dim3(256), 0, s.stream>>>((const u64 *)bins[b].d_lut, n_entries, sums);
		k_db_unpack<SIZE><<<dim3((u32)((n_of[b] + 255) / 256)), dim3(256), 0, s.stream>>>(bins[b].d_out, n_of[b], sums, (u32)n_entries, P.k, P.lut_prefix_len, P.sbytes, P.cbytes,
		                                                                             recs + off * W);
		off += n_of[b];
	}
	HIPCHK(hipGetLastError());
	(void)rb_in;
	u64 *sorted = recs;
	const u32 key_bytes = (2 * P.k + 7) / 8;
	if (n_total >= 2)
		if ((rc = sort_device(s, recs, (u64 *)s.recB.p, n_total, W, key_bytes, &sorted, true /* the count rides above the key: stable LSD passes */)))
			return rc;
	HIPCHK(hipMemsetAsync(d_lut_out, 0, (1ull << (2 * p_out)) * 8, s.stream));
	if (n_total)
		k_db_pack<SIZE><<<dim3((u32)((n_total + 255) / 256)), dim3(256), 0, s.stream>>>(sorted, n_total, P.k, p_out, P.cbytes, d_out, d_lut_out);
	HIPCHK(hipGetLastError());
	return 0;
}
} // namespace
