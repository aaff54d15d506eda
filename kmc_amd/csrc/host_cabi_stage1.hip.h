/* kmc_amd/csrc/host_cabi_stage1.hip.h — part of kmc_hip.hip (included there, not compiled on its own): the C-ABI test hooks and stage 1 on the device (reads -> bins in HBM; one part of input text -> bin records). */
/* ---- stage-isolating test hooks ---- */
int kmc_hip_debug_expand(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint8_t *superkmers, uint64_t size,
                         uint64_t n_rec, const uint64_t *pack_bytes, uint64_t n_packs, uint64_t *out_recs)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if (!size || !n_rec || !n_packs)
		return fail(KMC_HIP_EINVAL, "kmc_hip_debug_expand needs a non-empty bin with packs");
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	std::vector<u64> ps(1, 0);
	for (u64 i = 0; i < n_packs; ++i)
		ps.push_back(ps.back() + pack_bytes[i]);
	if (ps.back() != size)
		return fail(KMC_HIP_ECORRUPT, "sum of pack_bytes != size");
	int rc = 0;
	if ((rc = ensure(s.in, size + 256)) || (rc = ensure(s.pack_start, ps.size() * 8)))
		return rc;
	HIPCHK(hipMemcpy(s.in.p, superkmers, size, hipMemcpyHostToDevice));
	HIPCHK(hipMemset((char *)s.in.p + size, 0, 256));
	HIPCHK(hipMemcpy(s.pack_start.p, ps.data(), ps.size() * 8, hipMemcpyHostToDevice));
	const u32 words = (P.k + 31) / 32;
	switch (words) {
	case 1: rc = debug_expand_t<1>(s, P, size, n_rec, n_packs); break;
	case 2: rc = debug_expand_t<2>(s, P, size, n_rec, n_packs); break;
	case 3: rc = debug_expand_t<3>(s, P, size, n_rec, n_packs); break;
	case 4: rc = debug_expand_t<4>(s, P, size, n_rec, n_packs); break;
	case 5: rc = debug_expand_t<5>(s, P, size, n_rec, n_packs); break;
	case 6: rc = debug_expand_t<6>(s, P, size, n_rec, n_packs); break;
	case 7: rc = debug_expand_t<7>(s, P, size, n_rec, n_packs); break;
	default: rc = debug_expand_t<8>(s, P, size, n_rec, n_packs); break;
	}
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(s.stream));
	u32 err = 0;
	if ((rc = read_and_clear_sticky(s, err)))
		return rc;
	if ((rc = err_to_code(err)))
		return rc;
	HIPCHK(hipMemcpy(out_recs, s.recA.p, n_rec * words * 8, hipMemcpyDeviceToHost));
	return 0;
}

int kmc_hip_debug_compact(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint64_t *sorted_recs, uint64_t n,
                          uint8_t *out_suffix, uint64_t out_capacity, uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4])
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if (!n)
		return fail(KMC_HIP_EINVAL, "kmc_hip_debug_compact needs n > 0");
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	const u32 words = (P.k + 31) / 32;
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	int rc = 0;
	if ((rc = ensure(s.recA, n * words * 8 + 256)) || (rc = ensure(s.out, out_capacity + 256)) || (rc = ensure(s.lut, lut_entries * 8 + 256)))
		return rc;
	HIPCHK(hipMemcpy(s.recA.p, sorted_recs, n * words * 8, hipMemcpyHostToDevice));
	switch (words) {
	case 1: rc = debug_compact_t<1>(s, P, n, out_capacity, lut_entries); break;
	case 2: rc = debug_compact_t<2>(s, P, n, out_capacity, lut_entries); break;
	case 3: rc = debug_compact_t<3>(s, P, n, out_capacity, lut_entries); break;
	case 4: rc = debug_compact_t<4>(s, P, n, out_capacity, lut_entries); break;
	case 5: rc = debug_compact_t<5>(s, P, n, out_capacity, lut_entries); break;
	case 6: rc = debug_compact_t<6>(s, P, n, out_capacity, lut_entries); break;
	case 7: rc = debug_compact_t<7>(s, P, n, out_capacity, lut_entries); break;
	default: rc = debug_compact_t<8>(s, P, n, out_capacity, lut_entries); break;
	}
	if (rc)
		return rc;
	HIPCHK(hipStreamSynchronize(s.stream));
	HostRes r;
	HIPCHK(hipMemcpy(&r, s.zero.p, sizeof r, hipMemcpyDeviceToHost));
	u32 err = 0;
	if ((rc = read_and_clear_sticky(s, err)))
		return rc;
	if ((rc = err_to_code(err)))
		return rc;
	if (r.out_bytes > out_capacity)
		return fail(KMC_HIP_ECAPACITY, "out_capacity too small");
	if (!P.without_output) {
		if (r.out_bytes)
			HIPCHK(hipMemcpy(out_suffix, s.out.p, r.out_bytes, hipMemcpyDeviceToHost));
		if (lut_entries)
			HIPCHK(hipMemcpy(lut, s.lut.p, lut_entries * 8, hipMemcpyDeviceToHost));
	}
	*out_bytes = r.out_bytes;
	for (int i = 0; i < 4; ++i)
		stats[i] = r.stats[i];
	return 0;
}

/* ---- stage 1, first kernels: test hook (synchronous, own temporary buffers) ---- */
int kmc_hip_debug_split_reads(kmc_hip_ctx *ctx, int dev, const int8_t *codes, uint64_t n, uint32_t kmer_len, uint32_t signature_len, uint32_t *sig,
                              uint64_t *sk_pos, uint32_t *sk_len, uint32_t *sk_sig, uint64_t sk_cap, uint64_t *n_sk)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (!codes || !sig || !n_sk || (sk_cap && (!sk_pos || !sk_len || !sk_sig)))
		return fail(KMC_HIP_EINVAL, "kmc_hip_debug_split_reads: NULL argument");
	if (kmer_len < 1 || kmer_len > (uint32_t)S1_MAX_K || signature_len < 5 || signature_len > 11 || signature_len > kmer_len)
		return fail(KMC_HIP_EINVAL, "kmc_hip_debug_split_reads: kmer_len 1..256, signature_len 5..11 and <= kmer_len");
	*n_sk = 0;
	if (!n)
		return 0;
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	const u64 tiles = (n + S1_TILE - 1) / S1_TILE;
	if (tiles > 0x7FFFFFFFull)
		return fail(KMC_HIP_EINVAL, "too many symbols for one call");
	void *d_codes = nullptr, *d_sig = nullptr, *d_status = nullptr, *d_pos = nullptr, *d_len = nullptr, *d_ssig = nullptr, *d_small = nullptr;
	auto release = [&] {
		for (void *p : {d_codes, d_sig, d_status, d_pos, d_len, d_ssig, d_small})
			if (p)
				(void)hipFree(p);
	};
#define S1CHK(call)                                                                                                    \
	do {                                                                                                               \
		hipError_t e__ = (call);                                                                                       \
		if (e__ != hipSuccess) {                                                                                       \
			release();                                                                                                 \
			return fail_hip(#call, e__);                                                                               \
		}                                                                                                              \
	} while (0)
	const u64 cap = sk_cap ? sk_cap : 1;
	S1CHK(hipMalloc(&d_codes, n));
	S1CHK(hipMalloc(&d_sig, n * 4));
	const u64 ctiles = s1_cut_tiles(n); /* the cutting kernel works on S1_SUB tiles per workgroup */
	S1CHK(hipMalloc(&d_status, ctiles * 16));
	S1CHK(hipMalloc(&d_pos, cap * 8));
	S1CHK(hipMalloc(&d_len, cap * 4));
	S1CHK(hipMalloc(&d_ssig, cap * 4));
	S1CHK(hipMalloc(&d_small, 64));
	S1CHK(hipMemcpyAsync(d_codes, codes, n, hipMemcpyHostToDevice, s.stream));
	S1CHK(hipMemsetAsync(d_status, 0, ctiles * 16, s.stream));
	S1CHK(hipMemsetAsync(d_small, 0, 64, s.stream));
	k_s1_signatures<<<dim3((u32)tiles), dim3(S1_BLOCK), 0, s.stream>>>((const int8_t *)d_codes, n, kmer_len, signature_len, (u32 *)d_sig);
	k_s1_cut<false><<<dim3((u32)ctiles), dim3(S1_BLOCK), 0, s.stream>>>((const u32 *)d_sig, (const int8_t *)nullptr, 0u, n, kmer_len,
	                                                                      (u64 *)d_status, (u64 *)d_status + ctiles, (u32 *)d_small + 2, (u64 *)d_pos, (u32 *)d_len,
	                                                                     (u32 *)d_ssig, sk_cap, (u64 *)d_small, (const u64 *)nullptr, err_ptr(s));
	S1CHK(hipGetLastError());
	u64 cnt = 0;
	S1CHK(hipMemcpyAsync(&cnt, d_small, 8, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipMemcpyAsync(sig, d_sig, n * 4, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipStreamSynchronize(s.stream));
	const u64 take = cnt < sk_cap ? cnt : sk_cap;
	if (take) {
		S1CHK(hipMemcpy(sk_pos, d_pos, take * 8, hipMemcpyDeviceToHost));
		S1CHK(hipMemcpy(sk_len, d_len, take * 4, hipMemcpyDeviceToHost));
		S1CHK(hipMemcpy(sk_sig, d_ssig, take * 4, hipMemcpyDeviceToHost));
	}
#undef S1CHK
	release();
	*n_sk = cnt;
	u32 err = 0;
	if (int rc = read_and_clear_sticky(s, err))
		return rc;
	return err_to_code(err);
}

/* ---- stage 1 on the device: reads -> bins in HBM, ready for kmc_hip_process_bins_device ---- */
struct kmc_hip_s1_plan {
	int dev = 0;
	uint32_t k = 0, n_bins = 0;
	u64 n = 0, n_sk = 0;
	const int8_t *d_codes = nullptr;
	const int *d_map = nullptr;
	void *d_pos = nullptr, *d_len = nullptr, *d_ssig = nullptr, *d_tot = nullptr, *d_lay = nullptr; /* d_lay: bin_base | pack_base | cursor */
	bool emitted = false;
};

static void s1_plan_release(kmc_hip_s1_plan *p)
{
	for (void *q : {p->d_pos, p->d_len, p->d_ssig, p->d_tot, p->d_lay})
		if (q)
			(void)hipFree(q);
	delete p;
}

int kmc_hip_split_reads_plan(kmc_hip_ctx *ctx, int dev, const int8_t *d_codes, uint64_t n, uint32_t kmer_len, uint32_t signature_len, const int32_t *d_sig_to_bin,
                             uint32_t n_bins, kmc_hip_s1_plan **plan, uint64_t *bin_base, uint64_t *bin_bytes, uint64_t *bin_superkmers, uint64_t *bin_kmers,
                             uint64_t *pack_base)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (!d_codes || !d_sig_to_bin || !plan || !bin_base || !bin_bytes || !bin_superkmers || !bin_kmers || !pack_base)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_reads_plan: NULL argument");
	if (kmer_len < 1 || kmer_len > (uint32_t)S1_MAX_K || signature_len < 5 || signature_len > 11 || signature_len > kmer_len)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_reads_plan: kmer_len 1..256, signature_len 5..11 and <= kmer_len");
	if (n_bins < 1 || n_bins > (uint32_t)S1_MAX_BINS)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_reads_plan: n_bins 1..2048");
	const u64 tiles = s1_cut_tiles(n);
	if (!n || tiles > 0x7FFFFFFFull)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_reads_plan: 1 .. 2^41 symbols per call");
	*plan = nullptr;
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	kmc_hip_s1_plan *p = new kmc_hip_s1_plan;
	p->dev = dev, p->k = kmer_len, p->n_bins = n_bins, p->n = n, p->d_codes = d_codes, p->d_map = d_sig_to_bin;
	void *d_status = nullptr, *d_small = nullptr;
	auto release_tmp = [&] {
		for (void *q : {d_status, d_small})
			if (q)
				(void)hipFree(q);
	};
#define S1CHK(call)                                                                                                    \
	do {                                                                                                               \
		hipError_t e__ = (call);                                                                                       \
		if (e__ != hipSuccess) {                                                                                       \
			release_tmp();                                                                                             \
			s1_plan_release(p);                                                                                        \
			return fail_hip(#call, e__);                                                                               \
		}                                                                                                              \
	} while (0)
	S1CHK(hipMalloc(&d_status, tiles * 16));
	S1CHK(hipMalloc(&d_small, 64));
	S1CHK(hipMalloc(&p->d_tot, (size_t)3 * n_bins * 8));
	S1CHK(hipMalloc(&p->d_lay, (size_t)(3 * n_bins + 2) * 8));
	/* signatures are computed inside the cutting kernel (never stored). The number of super-k-mers is only known after the cut: a first guess (one per 8 symbols; real reads give one per 10-40 at k = 27), and a
	 * second cut with the exact number when the guess was short */
	u64 cap = n / 8 + 4096, cnt = 0;
	for (int attempt = 0; attempt < 2; ++attempt) {
		S1CHK(hipMalloc(&p->d_pos, cap * 8));
		S1CHK(hipMalloc(&p->d_len, cap * 4));
		S1CHK(hipMalloc(&p->d_ssig, cap * 4));
		S1CHK(hipMemsetAsync(d_status, 0, tiles * 16, s.stream));
		S1CHK(hipMemsetAsync(d_small, 0, 64, s.stream));
		k_s1_cut<true><<<dim3((u32)tiles), dim3(S1_BLOCK), 0, s.stream>>>((const u32 *)nullptr, d_codes, signature_len, n, kmer_len, (u64 *)d_status,
		                                                                    (u64 *)d_status + tiles, (u32 *)d_small + 2, (u64 *)p->d_pos, (u32 *)p->d_len,
		                                                                    (u32 *)p->d_ssig, cap, (u64 *)d_small, (const u64 *)nullptr, (u32 *)d_small + 4);
		S1CHK(hipGetLastError());
		u64 small[3] = {0, 0, 0}; /* count | ticket | the cut's own error word: a short guess must not poison the stream's sticky word */
		S1CHK(hipMemcpyAsync(small, d_small, sizeof small, hipMemcpyDeviceToHost, s.stream));
		S1CHK(hipStreamSynchronize(s.stream));
		cnt = small[0];
		if ((u32)small[2] & ~KERR_CAPACITY) {
			release_tmp();
			s1_plan_release(p);
			return err_to_code((u32)small[2] & ~KERR_CAPACITY);
		}
		if (cnt <= cap)
			break;
		for (void **q : {&p->d_pos, &p->d_len, &p->d_ssig}) {
			(void)hipFree(*q);
			*q = nullptr;
		}
		cap = cnt;
	}
	p->n_sk = cnt;
	S1CHK(hipMemsetAsync(p->d_tot, 0, (size_t)3 * n_bins * 8, s.stream));
	u64 *tot = (u64 *)p->d_tot, *lay = (u64 *)p->d_lay;
	const u32 sk_tiles = (u32)((cnt + S1_SK_TILE - 1) / S1_SK_TILE);
	if (sk_tiles)
		k_s1_bin_totals<<<dim3(sk_tiles), dim3(256), 0, s.stream>>>((const u32 *)p->d_len, (const u32 *)p->d_ssig, cnt, kmer_len, d_sig_to_bin, n_bins, tot, tot + n_bins,
		                                                             tot + 2 * n_bins, err_ptr(s));
	k_s1_bin_layout<<<dim3(1), dim3(256), 0, s.stream>>>(tot, n_bins, lay, lay + n_bins + 1, lay + 2 * n_bins + 2, (u64 *)nullptr);
	S1CHK(hipGetLastError());
	S1CHK(hipMemcpyAsync(bin_bytes, tot, (size_t)n_bins * 8, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipMemcpyAsync(bin_superkmers, tot + n_bins, (size_t)n_bins * 8, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipMemcpyAsync(bin_kmers, tot + 2 * n_bins, (size_t)n_bins * 8, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipMemcpyAsync(bin_base, lay, (size_t)(n_bins + 1) * 8, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipMemcpyAsync(pack_base, lay + n_bins + 1, (size_t)(n_bins + 1) * 8, hipMemcpyDeviceToHost, s.stream));
	S1CHK(hipStreamSynchronize(s.stream));
#undef S1CHK
	release_tmp();
	u32 err = 0;
	if (int rc = read_and_clear_sticky(s, err)) {
		s1_plan_release(p);
		return rc;
	}
	if (err) {
		s1_plan_release(p);
		return err_to_code(err);
	}
	*plan = p;
	return 0;
}

int kmc_hip_split_reads_emit(kmc_hip_ctx *ctx, kmc_hip_s1_plan *p, uint8_t *d_bins, uint64_t *d_pack_start)
{
	if (!p || !d_bins || !d_pack_start)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_reads_emit: NULL argument");
	if (p->emitted)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_reads_emit: the plan was emitted already");
	if (int rc = set_dev(ctx, p->dev))
		return rc;
	Slot &s = ctx->devs[p->dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	u64 *tot = (u64 *)p->d_tot, *lay = (u64 *)p->d_lay;
	const u32 nb = p->n_bins;
	k_s1_bin_layout<<<dim3(1), dim3(256), 0, s.stream>>>(tot, nb, lay, lay + nb + 1, lay + 2 * nb + 2, (u64 *)d_pack_start);
	const u32 sk_tiles = (u32)((p->n_sk + S1_SK_TILE - 1) / S1_SK_TILE);
	if (sk_tiles)
		k_s1_emit<<<dim3(sk_tiles), dim3(256), 0, s.stream>>>(p->d_codes, (const u64 *)p->d_pos, (const u32 *)p->d_len, (const u32 *)p->d_ssig, p->n_sk, p->k, p->d_map, nb,
		                                                       lay, lay + nb + 1, lay + 2 * nb + 2, d_bins, (u64 *)d_pack_start);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess)
		e = hipStreamSynchronize(s.stream);
	if (e != hipSuccess)
		return fail_hip("k_s1_emit", e);
	p->emitted = true;
	u32 err = 0;
	if (int rc = read_and_clear_sticky(s, err))
		return rc;
	return err_to_code(err);
}

void kmc_hip_split_reads_free(kmc_hip_ctx *ctx, kmc_hip_s1_plan *p)
{
	if (!p)
		return;
	if (ctx)
		(void)set_dev(ctx, p->dev);
	s1_plan_release(p);
}

/* ---- stage 1, one part of input text: host text -> host records + collector sums (the engine of kb_splitter_plugin.h) ----
 * Had not met a real GPU when round 2 ended (written after the GPU budget was spent); runs on the CPU over the emulated HIP runtime of
 * tests/hipemu (tests/test_hostlib_emulated.py). The launch sequence itself is kmc_amd/csrc/stage1_chain.h, which
 * runs inside the real KMC pipeline under the CPU emulation (oracle/_ref/kmc_emu_s1); what is new here is the backend below. */

int kmc_hip_split_set_map(kmc_hip_ctx *ctx, int dev, const int32_t *sig_to_bin, uint32_t signature_len)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (!sig_to_bin || signature_len < 5 || signature_len > 11)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_set_map: map NULL or signature_len outside 5..11");
	Dev &d = *ctx->devs[dev];
	std::lock_guard<std::mutex> lck(d.map_mtx);
	const u32 entries = (1u << (2 * signature_len)) + 1;
	if (d.d_sig_map && d.sig_map_entries != entries) {
		(void)hipFree(d.d_sig_map);
		d.d_sig_map = nullptr;
	}
	if (!d.d_sig_map)
		HIPCHK(hipMalloc((void **)&d.d_sig_map, (size_t)entries * 4));
	HIPCHK(hipMemcpy(d.d_sig_map, sig_to_bin, (size_t)entries * 4, hipMemcpyHostToDevice));
	d.sig_map_entries = entries;
	return 0;
}

int kmc_hip_split_part(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_split_params *p, const uint8_t *text, uint64_t size, uint8_t *recs,
                       uint64_t recs_capacity, uint64_t *recs_bytes, uint64_t *bin_off, uint64_t *bin_bytes, uint64_t *bin_kmers, uint64_t *bin_superkmers, uint64_t *bin_plus_x,
                       uint64_t *n_reads)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (!p || (size && !text) || !recs || !recs_bytes || !bin_off || !bin_bytes || !bin_kmers || !bin_superkmers || !bin_plus_x || !n_reads || slot < 0 || slot >= N_SLOTS)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_part: bad argument");
	if (p->kmer_len < 1 || p->kmer_len > (uint32_t)S1_MAX_K || p->signature_len < 5 || p->signature_len > 11 || p->signature_len > p->kmer_len || p->n_bins < 1 ||
	    p->n_bins > (uint32_t)S1_MAX_BINS || p->max_x > 3 || p->file_type > 1 || p->part_kind > 1 || (p->max_x && p->kmer_len < 4))
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_part: unsupported parameters");
	/* a configuration limit, not a property of the text: the pieces of an over-long line are cut at workgroup windows (stage1_chain.h) */
	if (p->line_cap < (uint64_t)p->kmer_len + S1_WG_TILE + 2)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_part: line_cap (mem_part_pmm_reads) is too small for the device splitter: it needs kmer_len + " + std::to_string(S1_WG_TILE + 2) +
		                                " symbols at least");
	Dev &d = *ctx->devs[dev];
	if (!d.d_sig_map || d.sig_map_entries != (1u << (2 * p->signature_len)) + 1)
		return fail(KMC_HIP_EINVAL, "kmc_hip_split_part: kmc_hip_split_set_map was not called for this signature length");
	*recs_bytes = 0;
	Slot &s = d.slot[slot];
	std::lock_guard<std::mutex> lck(s.mtx);
	/* text + codes + line ends (2 B per byte of text at most, see stage1_chain.h) + super-k-mers (2 B per symbol at the first guess) +
	 * records (~0.3 B per symbol) + per-bin arrays */
	if (int rc = ensure(d.s1_arena[slot], (size_t)size * 8 + ((size_t)32 << 20)))
		return rc;
	S1HipBackend be;
	be.stream = s.stream;
	be.slot = &s;
	be.arena = &d.s1_arena[slot];
	S1PartParams sp;
	sp.k = p->kmer_len;
	sp.m = p->signature_len;
	sp.n_bins = p->n_bins;
	sp.max_x = p->max_x;
	sp.both_strands = p->both_strands ? 1u : 0u;
	sp.lines_per_record = p->file_type == 1 ? 4u : 2u;
	sp.line_cap = p->line_cap;
	sp.d_sig_to_bin = d.d_sig_map;
	sp.sorted_emit = getenv("KMC_HIP_S1_SORTED_EMIT") != nullptr; /* the alternative emit (stage1_kernels.hip.h): to be measured before it becomes the default */
	S1PartResult R;
	u64 long_reads = 0;
	if (p->part_kind == 1) { /* a long-read part: the title (if the part has it) is taken off here, the symbols go up from an aligned buffer */
		const u64 skip = s1_long_read_title(text, size, p->file_type, long_reads);
		text += skip;
		size -= skip;
		sp.lines_per_record = 0;
	}
	try {
		uint8_t *d_text = (uint8_t *)be.alloc(size + 16);
		if (size) {
			hipError_t e = hipMemcpyAsync(d_text, text, size, hipMemcpyHostToDevice, s.stream);
			if (e != hipSuccess)
				return fail_hip("hipMemcpyAsync(text)", e);
		}
		const int rc = s1_split_part(be, d_text, size, size && text[size - 1] == '\n', sp, R);
		if (p->part_kind == 1)
			R.n_reads = long_reads;
		if (rc == S1_CHAIN_UNCOVERED)
			return KMC_HIP_UNCOVERED;
		if (rc != S1_CHAIN_OK)
			return R.device_error ? err_to_code(R.device_error) : fail(KMC_HIP_EDEVICE, "kmc_hip_split_part: stage-1 chain failed");
		*recs_bytes = R.recs_bytes;
		if (R.recs_bytes > recs_capacity)
			return fail(KMC_HIP_ECAPACITY, "kmc_hip_split_part: recs_capacity too small, *recs_bytes holds what this part needs");
		if (R.recs_bytes)
			be.d2h(recs, R.d_recs, R.recs_bytes);
	} catch (const S1BackendFailure &f) {
		return fail_hip(f.what, f.e);
	}
	for (uint32_t b = 0; b < p->n_bins; ++b) {
		bin_off[b] = R.bin_off[b];
		bin_bytes[b] = R.bin_bytes[b];
		bin_kmers[b] = R.bin_kmers[b];
		bin_superkmers[b] = R.bin_sk[b];
		bin_plus_x[b] = R.bin_plus_x[b];
	}
	*n_reads = R.n_reads;
	return 0;
}
