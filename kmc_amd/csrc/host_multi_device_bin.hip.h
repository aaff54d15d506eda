/* kmc_amd/csrc/host_multi_device_bin.hip.h — part of kmc_hip.hip (included there, not compiled on its own): one bin over all devices of the context. */
/* ---- one bin over ALL devices of the context (the oversized-bin path) ----------------------------------------------
 * The reference's strict-memory mode cuts a bin that does not fit into sub-bins by its k-mers' leading symbols, sorts them one after the other and
 * merges (kmc.h:1607-1692, bkb_sorter.h:187, bkb_*.h). With several GPUs the cut goes ACROSS devices instead:
 *   1  device d takes a contiguous share of the bin's expander packs (by bytes), expands it, and counts the TOP radix byte of its records
 *      (k_expand's fused histogram: one digit)
 *   2  the host adds the n_dev histograms and cuts the 256 values of the top byte into n_dev contiguous ranges of about n_rec / n_dev records: device g
 *      will own the k-mers whose top byte is in range g (runs of equal k-mers cannot straddle a cut)
 *   3  every device orders its records by the top byte (ONE k_onesweep pass): what it owes to device g is then one contiguous slice
 *   4  all-to-all: slice (d -> g) lands in device g's receive buffer behind the slices of the devices before d. RCCL (ncclSend / ncclRecv between one
 *      ncclGroupStart / End: the all-to-all over xGMI) when the context's devices are distinct GPUs, peer copies when they are not (a context over
 *      (0, 0): how this path is tested on a one-GPU box)
 *   5  every device sorts what it received (LSD passes over every key byte) and compacts it: suffix records, LUT counts, tallies for ITS key range
 *   6  ordered emission: the devices' records one after the other in range order are the bin's records; LUT counts and tallies add up.
 * Synchronous, host buffers in and out like kmc_hip_process_bin; with one device it is that call by another road. */
namespace {
template <int SIZE>
int compact_array_t(Slot &s, const DevParams &P, const u64 *sorted, u64 n, u64 out_capacity, u64 lut_entries)
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
	return compact_group<SIZE>(s, bins, sorted, nullptr, P, lut_entries, counter_idx);
}

template <int SIZE>
int process_bin_multi_t(kmc_hip_ctx *ctx, const DevParams &P, u64 lut_entries, const uint8_t *img, u64 size, u64 n_rec, const std::vector<u64> &ps, uint8_t *out,
                        u64 out_capacity, u64 *out_bytes, u64 *lut, u64 stats[4])
{
	const int n_dev = (int)ctx->devs.size();
	const u32 key_bytes = (2 * P.k + 7) / 8, top = key_bytes - 1;
	const u32 rec_bytes = P.sbytes + P.cbytes;
	const u64 n_packs = ps.size() - 1;
	/* 1: shares of packs, and the k-mers of every share (the front end checks the count against the byte stream) */
	std::vector<u64> first((size_t)n_dev + 1, n_packs), nk((size_t)n_dev, 0);
	first[0] = 0;
	for (int d = 1; d < n_dev; ++d) {
		const u64 want = size / (u64)n_dev * (u64)d;
		first[d] = (u64)(std::lower_bound(ps.begin(), ps.end(), want) - ps.begin());
		first[d] = std::min(std::max(first[d], first[d - 1]), n_packs);
	}
	u64 seen = 0;
	for (int d = 0; d < n_dev; ++d) {
		for (u64 pos = ps[first[d]]; pos < ps[first[d + 1]];) {
			const u32 e = img[pos];
			nk[d] += e + 1u;
			pos += 1 + (P.k + e + 3) / 4;
		}
		seen += nk[d];
	}
	if (seen != n_rec)
		return fail(KMC_HIP_ECORRUPT, "n_rec disagrees with the super-k-mer stream");
	std::vector<std::unique_lock<std::mutex>> locks;
	for (int d = 0; d < n_dev; ++d)
		locks.emplace_back(ctx->devs[d]->slot[0].mtx);
	auto S = [&](int d) -> Slot & { return ctx->devs[d]->slot[0]; };
	int rc = 0;
	std::vector<u64 *> parted((size_t)n_dev, nullptr);
	std::vector<ZeroPlan> zs((size_t)n_dev);
	SortPlan sp1;
	sp1.key_bytes = key_bytes;
	sp1.top = 1;
	sp1.key_bits = 8 * key_bytes;
	/* expand + histogram of the top byte, then the pass over it, device by device on its own stream */
	for (int d = 0; d < n_dev; ++d) {
		if ((rc = set_dev(ctx, d)))
			return rc;
		Slot &s = S(d);
		s.timed = false;
		const u64 b0 = ps[first[d]], b1 = ps[first[d + 1]], sz = b1 - b0, np = first[d + 1] - first[d];
		if (!nk[d])
			continue;
		std::vector<u64> lps(np + 1);
		for (u64 i = 0; i <= np; ++i)
			lps[i] = ps[first[d] + i] - b0;
		if ((rc = ensure(s.in, sz + 256)) || (rc = ensure(s.pack_start, (np + 1) * 8)) || (rc = ensure(s.recA, nk[d] * SIZE * 8 + 256)) ||
		    (rc = ensure(s.recB, nk[d] * SIZE * 8 + 256)))
			return rc;
		HIPCHK(hipMemcpyAsync(s.in.p, img + b0, sz, hipMemcpyHostToDevice, s.stream));
		HIPCHK(hipMemsetAsync((char *)s.in.p + sz, 0, 256, s.stream));
		HIPCHK(hipMemcpy(s.pack_start.p, lps.data(), (np + 1) * 8, hipMemcpyHostToDevice)); /* lps is a local: synchronous */
		std::vector<BinPlan> bins(1);
		bins[0].d_in = (const uint8_t *)s.in.p;
		bins[0].size = sz;
		bins[0].n_rec = nk[d];
		bins[0].n_packs = np;
		bins[0].d_pack_start = (const u64 *)s.pack_start.p;
		zs[d] = plan_group<SIZE>(s, bins, nk[d], 1, true, true, false, 0);
		if ((rc = apply_plan(s, zs[d])))
			return rc;
		u32 counter_idx = 0;
		bool hist_done = false;
		if ((rc = front_end_group<SIZE>(s, bins, zs[d].ghist, P, 1, top, counter_idx, hist_done, (u64 *)s.recA.p, nk[d] >= 2)))
			return rc;
		if (nk[d] >= 2) {
			if ((rc = sort_device_t<SIZE>(s, zs[d], (u64 *)s.recA.p, (u64 *)s.recB.p, nk[d], sp1, &parted[d], counter_idx, hist_done, nullptr, true)))
				return rc;
		} else
			parted[d] = (u64 *)s.recA.p;
	}
	/* 2: histograms -> ranges. (A share of a single k-mer has no histogram: its one record is read back.) */
	std::vector<std::vector<u64>> hist((size_t)n_dev, std::vector<u64>(256, 0));
	for (int d = 0; d < n_dev; ++d) {
		if (!nk[d])
			continue;
		if ((rc = set_dev(ctx, d)))
			return rc;
		Slot &s = S(d);
		HIPCHK(hipStreamSynchronize(s.stream));
		if (nk[d] >= 2)
			HIPCHK(hipMemcpy(hist[d].data(), zero_ptr<u64>(s, zs[d].ghist), 256 * 8, hipMemcpyDeviceToHost));
		else {
			u64 rec[SIZE];
			HIPCHK(hipMemcpy(rec, s.recA.p, SIZE * 8, hipMemcpyDeviceToHost));
			hist[d][(rec[top >> 3] >> ((top & 7) * 8)) & 0xFF] = 1;
		}
		u32 err = 0;
		if ((rc = read_and_clear_sticky(s, err)) || (rc = err_to_code(err)))
			return rc;
	}
	std::vector<u32> cut((size_t)n_dev + 1, 256); /* device g owns top bytes [cut[g], cut[g+1]) */
	cut[0] = 0;
	{
		u64 acc = 0;
		int g = 1;
		for (u32 v = 0; v < 256 && g < n_dev; ++v) {
			for (int d = 0; d < n_dev; ++d)
				acc += hist[d][v];
			while (g < n_dev && acc >= n_rec / (u64)n_dev * (u64)g)
				cut[g++] = v + 1;
		}
	}
	std::vector<std::vector<u64>> cnt((size_t)n_dev, std::vector<u64>((size_t)n_dev, 0)), soff = cnt, roff = cnt;
	std::vector<u64> n_own((size_t)n_dev, 0);
	for (int d = 0; d < n_dev; ++d) {
		u64 run = 0;
		for (int g = 0; g < n_dev; ++g) {
			soff[d][g] = run;
			for (u32 v = cut[g]; v < cut[g + 1]; ++v)
				cnt[d][g] += hist[d][v];
			run += cnt[d][g];
		}
	}
	for (int g = 0; g < n_dev; ++g)
		for (int d = 0; d < n_dev; ++d) {
			roff[d][g] = n_own[g];
			n_own[g] += cnt[d][g];
		}
	/* 4: the exchange */
	bool distinct = n_dev > 1;
	for (int a = 0; a < n_dev; ++a)
		for (int b = a + 1; b < n_dev; ++b)
			distinct = distinct && ctx->devs[a]->ordinal != ctx->devs[b]->ordinal;
	for (int g = 0; g < n_dev; ++g) {
		if ((rc = set_dev(ctx, g)))
			return rc;
		if ((rc = ensure(ctx->devs[g]->xchg, n_own[g] * SIZE * 8 + 256)))
			return rc;
	}
	if (distinct) {
		if (!ctx->comms_ready) {
			std::vector<int> ords(n_dev);
			for (int i = 0; i < n_dev; ++i)
				ords[i] = ctx->devs[i]->ordinal;
			ctx->comms.resize(n_dev);
			ncclResult_t r = ncclCommInitAll(ctx->comms.data(), n_dev, ords.data());
			if (r != ncclSuccess)
				return fail(KMC_HIP_EDEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
			ctx->comms_ready = true;
		}
		ncclResult_t r = ncclGroupStart();
		for (int d = 0; d < n_dev && r == ncclSuccess; ++d) {
			(void)hipSetDevice(ctx->devs[d]->ordinal);
			for (int g = 0; g < n_dev && r == ncclSuccess; ++g) { /* what d sends to g, and what d receives from g */
				if (cnt[d][g])
					r = ncclSend(parted[d] + soff[d][g] * SIZE, cnt[d][g] * SIZE * 8, ncclUint8, g, ctx->comms[d], S(d).stream);
				if (cnt[g][d] && r == ncclSuccess)
					r = ncclRecv((u64 *)ctx->devs[d]->xchg.p + roff[g][d] * SIZE, cnt[g][d] * SIZE * 8, ncclUint8, g, ctx->comms[d], S(d).stream);
			}
		}
		ncclResult_t r2 = ncclGroupEnd();
		if (r != ncclSuccess || r2 != ncclSuccess)
			return fail(KMC_HIP_EDEVICE, std::string("ncclSend/ncclRecv: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
	} else {
		for (int d = 0; d < n_dev; ++d) {
			if ((rc = set_dev(ctx, d)))
				return rc;
			for (int g = 0; g < n_dev; ++g)
				if (cnt[d][g]) {
					/* a context that names some GPUs twice and others once, e.g. (0, 0, 1), takes this branch too: between two different GPUs the copy is a peer copy with
					 * both ordinals spelled out (no reliance on the runtime guessing the devices of a plain device-to-device copy) */
					const int od = ctx->devs[d]->ordinal, og = ctx->devs[g]->ordinal;
					void *dst = (u64 *)ctx->devs[g]->xchg.p + roff[d][g] * SIZE;
					const void *src = parted[d] + soff[d][g] * SIZE;
					if (od == og)
						HIPCHK(hipMemcpyAsync(dst, src, cnt[d][g] * SIZE * 8, hipMemcpyDeviceToDevice, S(d).stream));
					else
						HIPCHK(hipMemcpyPeerAsync(dst, og, src, od, cnt[d][g] * SIZE * 8, S(d).stream));
				}
		}
	}
	for (int d = 0; d < n_dev; ++d) {
		if ((rc = set_dev(ctx, d)))
			return rc;
		HIPCHK(hipStreamSynchronize(S(d).stream));
	}
	/* 5: every device sorts and compacts its range */
	std::vector<u64> cap((size_t)n_dev, 0);
	for (int g = 0; g < n_dev; ++g) {
		if (!n_own[g])
			continue;
		if ((rc = set_dev(ctx, g)))
			return rc;
		Slot &s = S(g);
		cap[g] = P.without_output ? 0 : ((n_own[g] + 1) / std::max<u32>(P.cutoff_min, 1)) * (u64)rec_bytes;
		if ((rc = ensure(s.recA, n_own[g] * SIZE * 8 + 256)) || (rc = ensure(s.out, cap[g] + 256)) || (rc = ensure(s.lut, lut_entries * 8 + 256)))
			return rc;
		u64 *sorted = (u64 *)ctx->devs[g]->xchg.p;
		if (n_own[g] >= 2)
			if ((rc = sort_device(s, (u64 *)ctx->devs[g]->xchg.p, (u64 *)s.recA.p, n_own[g], SIZE, key_bytes, &sorted, true)))
				return rc;
		if ((rc = compact_array_t<SIZE>(s, P, sorted, n_own[g], cap[g], lut_entries)))
			return rc;
	}
	/* 6: ordered emission */
	u64 total_bytes = 0, st[4] = {0, 0, 0, 0};
	if (lut_entries && !P.without_output)
		memset(lut, 0, lut_entries * 8);
	std::vector<u64> part(lut_entries ? lut_entries : 1);
	for (int g = 0; g < n_dev; ++g) {
		if (!n_own[g])
			continue;
		if ((rc = set_dev(ctx, g)))
			return rc;
		Slot &s = S(g);
		HIPCHK(hipStreamSynchronize(s.stream));
		u32 err = 0;
		if ((rc = read_and_clear_sticky(s, err)) || (rc = err_to_code(err)))
			return rc;
		HostRes r;
		HIPCHK(hipMemcpy(&r, s.zero.p, sizeof r, hipMemcpyDeviceToHost));
		if (!P.without_output) {
			if (total_bytes + r.out_bytes > out_capacity)
				return fail(KMC_HIP_ECAPACITY, "out_capacity too small for the counted k-mers");
			if (r.out_bytes)
				HIPCHK(hipMemcpy(out + total_bytes, s.out.p, r.out_bytes, hipMemcpyDeviceToHost));
			if (lut_entries) {
				HIPCHK(hipMemcpy(part.data(), s.lut.p, lut_entries * 8, hipMemcpyDeviceToHost));
				for (u64 i = 0; i < lut_entries; ++i)
					lut[i] += part[i];
			}
			total_bytes += r.out_bytes;
		}
		for (int i = 0; i < 3; ++i)
			st[i] += r.stats[i];
	}
	st[3] = n_rec; /* kb_sorter.h:1166 */
	*out_bytes = total_bytes;
	for (int i = 0; i < 4; ++i)
		stats[i] = st[i];
	return 0;
}
} // namespace
