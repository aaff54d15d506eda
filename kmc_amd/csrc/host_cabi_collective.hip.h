/* kmc_amd/csrc/host_cabi_collective.hip.h — part of kmc_hip.hip (included there, not compiled on its own): the tally all-reduce over the devices of a context (RCCL) and the instrumentation entries. */
/* ---- tallies over devices: one RCCL all-reduce of 4 x uint64 ---- */
int kmc_hip_allreduce_stats(kmc_hip_ctx *ctx, uint64_t *per_dev_stats)
{
	if (!ctx || !per_dev_stats)
		return fail(KMC_HIP_EINVAL, "kmc_hip_allreduce_stats: bad arguments");
	std::lock_guard<std::mutex> lck(ctx->mtx);
	const int n = (int)ctx->devs.size();
	if (!ctx->comms_ready) {
		std::vector<int> ords(n);
		for (int i = 0; i < n; ++i)
			ords[i] = ctx->devs[i]->ordinal;
		ctx->comms.resize(n);
		ncclResult_t r = ncclCommInitAll(ctx->comms.data(), n, ords.data());
		if (r != ncclSuccess)
			return fail(KMC_HIP_EDEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
		ctx->comms_ready = true;
	}
	for (int i = 0; i < n; ++i) {
		if (int rc = set_dev(ctx, i))
			return rc;
		if (int rc = ensure(ctx->devs[i]->rccl_buf, 64))
			return rc;
		HIPCHK(hipMemcpy(ctx->devs[i]->rccl_buf.p, per_dev_stats + 4 * i, 32, hipMemcpyHostToDevice));
	}
	ncclResult_t r = ncclGroupStart();
	for (int i = 0; i < n && r == ncclSuccess; ++i) {
		(void)hipSetDevice(ctx->devs[i]->ordinal);
		r = ncclAllReduce(ctx->devs[i]->rccl_buf.p, ctx->devs[i]->rccl_buf.p, 4, ncclUint64, ncclSum, ctx->comms[i], ctx->devs[i]->slot[0].stream);
	}
	ncclResult_t r2 = ncclGroupEnd();
	if (r != ncclSuccess || r2 != ncclSuccess)
		return fail(KMC_HIP_EDEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
	for (int i = 0; i < n; ++i) {
		if (int rc = set_dev(ctx, i))
			return rc;
		HIPCHK(hipStreamSynchronize(ctx->devs[i]->slot[0].stream));
		HIPCHK(hipMemcpy(per_dev_stats + 4 * i, ctx->devs[i]->rccl_buf.p, 32, hipMemcpyDeviceToHost));
	}
	return 0;
}

#ifdef KMC_TRACE
/* tuning builds only: copy the device trace buffer (see kernels.hip.h TRACE_STAMP) */
int kmc_hip_debug_read_trace(kmc_hip_ctx *ctx, int dev, unsigned long long *dst, uint64_t n_words, int clear)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	HIPCHK(hipDeviceSynchronize());
	if (n_words > (uint64_t)TRACE_SLOTS * 8)
		n_words = (uint64_t)TRACE_SLOTS * 8;
	HIPCHK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), n_words * 8, 0, hipMemcpyDeviceToHost));
	if (clear) {
		void *p = nullptr;
		HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace)));
		HIPCHK(hipMemset(p, 0, (size_t)TRACE_SLOTS * 64));
	}
	return 0;
}
#endif

/* ---- instrumentation ---- */
int kmc_hip_last_timings(kmc_hip_ctx *ctx, int dev, float ms[6])
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	HIPCHK(hipStreamSynchronize(s.stream));
	for (int i = 0; i < 5; ++i)
		HIPCHK(hipEventElapsedTime(&ms[i], s.ev[i], s.ev[i + 1]));
	HIPCHK(hipEventElapsedTime(&ms[5], s.ev[0], s.ev[5]));
	return 0;
}

int kmc_hip_scatter_totals(kmc_hip_ctx *ctx, int dev, int reset, uint64_t *n_launches, double *total_ms, uint64_t *total_records)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	u64 nl = 0, keys = 0;
	double ms = 0;
	for (auto &s : ctx->devs[dev]->slot) {
		std::lock_guard<std::mutex> lck(s.mtx);
		HIPCHK(hipStreamSynchronize(s.stream));
		if (int rc = harvest(s))
			return rc;
		nl += s.sc_launch_total;
		keys += s.sc_keys_total;
		ms += s.sc_ms_total;
		if (reset) {
			s.async_seq = 0;
			s.sc_launch_total = 0;
			s.sc_keys_total = 0;
			s.sc_ms_total = 0;
		}
	}
	if (n_launches)
		*n_launches = nl;
	if (total_ms)
		*total_ms = ms;
	if (total_records)
		*total_records = keys;
	return 0;
}

int kmc_hip_order_database_device(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const kmc_hip_bin_desc *bins, uint64_t n_bins, uint32_t out_lut_prefix_len,
                                  uint8_t *d_out, uint64_t out_capacity, uint64_t *d_lut_out, uint64_t *n_kmers)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if ((n_bins && !bins) || !d_out || !d_lut_out || !n_kmers)
		return fail(KMC_HIP_EINVAL, "kmc_hip_order_database_device: NULL argument");
	if (P.kff || !P.lut_prefix_len || P.without_output)
		return fail(KMC_HIP_EINVAL, "kmc_hip_order_database_device: needs KMC-format bins (lut_prefix_len > 0, with output)");
	if (out_lut_prefix_len < 1 || out_lut_prefix_len > 15 || out_lut_prefix_len >= P.k || (P.k - out_lut_prefix_len) % 4)
		return fail(KMC_HIP_EINVAL, "kmc_hip_order_database_device: (kmer_len - out_lut_prefix_len) must be a positive multiple of 4, out_lut_prefix_len 1..15");
	const u32 words = (P.k + 31) / 32;
	if (words + 1 > 8)
		return fail(KMC_HIP_EINVAL, "kmc_hip_order_database_device: kmer_len <= 224");
	/* the bins may come from asynchronous kmc_hip_process_bins_device calls on any stream slot: wait for all of them, run the groups whose hybrid sort asked
	 * for LSD passes again, raise their deferred errors (the body of kmc_hip_synchronize) — before a single out_bytes is read */
	if (int rc = kmc_hip_synchronize(ctx, dev))
		return rc;
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	const u32 rb_in = P.sbytes + P.cbytes, rb_out = (P.k - out_lut_prefix_len) / 4 + P.cbytes;
	std::vector<u64> n_of((size_t)n_bins, 0);
	u64 n_total = 0;
	for (uint64_t b = 0; b < n_bins; ++b) {
		u64 ob = 0;
		HIPCHK(hipMemcpy(&ob, bins[b].d_out_bytes, 8, hipMemcpyDeviceToHost));
		if (ob % rb_in)
			return fail(KMC_HIP_ECORRUPT, "kmc_hip_order_database_device: a bin's out_bytes is not a whole number of records");
		n_of[b] = ob / rb_in;
		n_total += n_of[b];
	}
	*n_kmers = n_total;
	if (n_total * rb_out > out_capacity)
		return fail(KMC_HIP_ECAPACITY, "kmc_hip_order_database_device: out_capacity too small");
	s.timed = false;
	int rc = KMC_HIP_EINVAL;
	switch (words) {
	case 1: rc = order_database_t<1>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	case 2: rc = order_database_t<2>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	case 3: rc = order_database_t<3>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	case 4: rc = order_database_t<4>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	case 5: rc = order_database_t<5>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	case 6: rc = order_database_t<6>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	case 7: rc = order_database_t<7>(s, P, bins, n_of, n_total, out_lut_prefix_len, d_out, (u64 *)d_lut_out); break;
	}
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(s.stream));
	if (int rc2 = harvest(s))
		return rc2;
	u32 err = 0;
	if (int rc2 = read_and_clear_sticky(s, err))
		return rc2;
	return err_to_code(err);
}

int kmc_hip_set_hybrid(int mode)
{
	const int before = hybrid_mode();
	g_hybrid_override.store(mode >= 2 ? 1 : mode, std::memory_order_relaxed); /* 2 (round 3's finishers) left the library in round 6 */
	g_hybrid_groups.store(0);
	g_redo_groups.store(0);
	g_extra_top.store(0);
	for (auto &c : g_path)
		c.store(0);
	g_indirect_groups.store(0);
	return before;
}

int kmc_hip_path_counters(kmc_hip_ctx *ctx, int dev, uint64_t counters[8])
{
	if (!counters)
		return fail(KMC_HIP_EINVAL, "counters == NULL");
	for (int i = 0; i < 8; ++i)
		counters[i] = i < 4 ? g_path[i].load() : (i == 6 ? g_indirect_groups.load() : 0);
	if (!ctx)
		return 0;
	if (int rc = set_dev(ctx, dev))
		return rc;
	for (auto &s : ctx->devs[dev]->slot) { /* the tiles k_giant_tiles took: counted on the device, in every stream's error block */
		std::lock_guard<std::mutex> lck(s.mtx);
		HIPCHK(hipStreamSynchronize(s.stream));
		u32 w[4] = {};
		HIPCHK(hipMemcpy(w, (u32 *)s.sticky.p + 12, sizeof w, hipMemcpyDeviceToHost));
		counters[4] += w[0];
		counters[5] += ((u64)w[3] << 32) | w[2];
	}
	return 0;
}

int kmc_hip_local_sort_totals(kmc_hip_ctx *ctx, int dev, int reset, uint64_t *n_launches, double *total_ms, uint64_t *total_records, uint64_t *n_hybrid_groups,
                              uint64_t *n_redo_groups)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	u64 nl = 0, keys = 0;
	double ms = 0;
	for (auto &s : ctx->devs[dev]->slot) {
		std::lock_guard<std::mutex> lck(s.mtx);
		HIPCHK(hipStreamSynchronize(s.stream));
		if (int rc = harvest(s))
			return rc;
		nl += s.ls_launch_total;
		keys += s.ls_keys_total;
		ms += s.ls_ms_total;
		if (reset) {
			s.ls_launch_total = 0;
			s.ls_keys_total = 0;
			s.ls_ms_total = 0;
		}
	}
	if (n_launches)
		*n_launches = nl;
	if (total_ms)
		*total_ms = ms;
	if (total_records)
		*total_records = keys;
	if (n_hybrid_groups)
		*n_hybrid_groups = g_hybrid_groups.load();
	if (n_redo_groups)
		*n_redo_groups = g_redo_groups.load();
	return 0;
}

} /* extern "C" */
