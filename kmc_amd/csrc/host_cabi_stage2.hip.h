/* kmc_amd/csrc/host_cabi_stage2.hip.h — part of kmc_hip.hip (included there, not compiled on its own): the C-ABI of stage 2: lifetime, narrow boundary, full boundary (one bin / several bins per call, device-resident batches). */
/* ================================================================================================ C-ABI */

/* stage 1, one part of text: the backend of kmc_amd/csrc/stage1_chain.h on a HIP stream (used by kmc_hip_split_part below) */
extern "C" {
static int sort_records_device_locked(Slot &s, void *d_recs, void *d_tmp, uint64_t n, uint32_t words, uint32_t key_bytes, void **d_result, int stable_lsd); /* defined below */
}
namespace {
struct S1BackendFailure {
	hipError_t e;
	const char *what;
};
/* Work memory comes from a grow-only arena of the slot (hipMalloc / hipFree per part would cost more than the kernels: hipFree synchronises
 * the device); what does not fit — the arena was sized from the part's size before anything about its content was known — is a separate
 * allocation, freed when the call ends. */
struct S1HipBackend {
	hipStream_t stream;
	Slot *slot = nullptr; /* held by the caller: its stage-2 work areas are free for the sort of sort_by_low16 */
	DBuf *arena = nullptr;
	size_t used = 0;
	std::vector<void *> extra;
	void *alloc_uninit(size_t bytes)
	{
		const size_t want = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
		void *p = nullptr;
		if (arena && used + want <= arena->cap) {
			p = static_cast<char *>(arena->p) + used;
			used += want;
		} else {
			hipError_t e = hipMalloc(&p, want);
			if (e != hipSuccess)
				throw S1BackendFailure{e, "hipMalloc"};
			extra.push_back(p);
		}
		return p;
	}
	void *alloc(size_t bytes) /* zeroed: status words, tickets, totals, the text and code streams (read with slack behind their ends) */
	{
		void *p = alloc_uninit(bytes);
		hipError_t e = hipMemsetAsync(p, 0, ((bytes ? bytes : 1) + 255) & ~(size_t)255, stream);
		if (e != hipSuccess)
			throw S1BackendFailure{e, "hipMemsetAsync"};
		return p;
	}
	void zero(void *p, size_t bytes)
	{
		hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
		if (e != hipSuccess)
			throw S1BackendFailure{e, "hipMemsetAsync"};
	}
	bool d2h(void *dst, const void *src, size_t bytes)
	{
		hipError_t e = hipGetLastError(); /* a failed launch before this point */
		if (e == hipSuccess)
			e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(stream);
		if (e != hipSuccess)
			throw S1BackendFailure{e, "device to host copy"};
		return true;
	}
	void h2d(void *dst, const void *src, size_t bytes)
	{
		hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(stream); /* the source may be reused */
		if (e != hipSuccess)
			throw S1BackendFailure{e, "host to device copy"};
	}
	u64 *sort_by_low16(u64 *keys, u64 *tmp, u64 n)
	{
		void *res = keys;
		if (n > 1 && sort_records_device_locked(*slot, keys, tmp, n, 1, 2, &res, 1 /* payload above the key: stable LSD passes */) != 0)
			throw S1BackendFailure{hipGetLastError(), "sort of the super-k-mer keys"};
		return (u64 *)res;
	}
	void release()
	{
		if (!extra.empty())
			(void)hipStreamSynchronize(stream);
		for (void *p : extra)
			(void)hipFree(p);
		extra.clear();
		used = 0;
	}
	~S1HipBackend() { release(); }
};
} // namespace
#define S1_LAUNCH(B, be, kernel, grid, block, ...) hipLaunchKernelGGL(kernel, grid, block, 0, (be).stream, __VA_ARGS__)
#include "stage1_chain.h"

extern "C" {

int kmc_hip_abi_version(void) { return KMC_HIP_ABI_VERSION; }
int kmc_hip_backend_kind(void)
{
#ifdef KMC_HIPEMU /* tests/hipemu: this source compiled for the CPU emulation (tests/emu.py build_hostlib) */
	return 1;
#else
	return 0;
#endif
}
const char *kmc_hip_last_error(kmc_hip_ctx *) { return g_err.c_str(); }
uint32_t kmc_hip_words(uint32_t kmer_len) { return (kmer_len + 31) / 32; }
uint32_t kmc_hip_counter_size(uint64_t cutoff_max, uint64_t counter_max) { return counter_bytes(cutoff_max, counter_max); }
uint32_t kmc_hip_out_rec_bytes(const kmc_hip_bin_params *p)
{
	return kmc_suffix_bytes(p->kmer_len, p->lut_prefix_len) + counter_bytes(p->cutoff_max, p->counter_max);
}
uint64_t kmc_hip_lut_entries(const kmc_hip_bin_params *p) { return p->lut_prefix_len ? 1ull << (2 * p->lut_prefix_len) : 0; }

int kmc_hip_device_count(void)
{
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess)
		return 0;
	return count;
}

int kmc_hip_init(const int *device_ids, int n_dev, kmc_hip_ctx **out)
{
	if (!out || n_dev < 1)
		return fail(KMC_HIP_EINVAL, "kmc_hip_init: bad arguments");
	int count = 0;
	HIPCHK(hipGetDeviceCount(&count));
	if (count < 1)
		return fail(KMC_HIP_EDEVICE, "no HIP device visible");
	kmc_hip_ctx *ctx = new kmc_hip_ctx();
	if (const char *e = getenv("KMC_HIP_DEBUG_PORTION_LOG2")) {
		const int lg = atoi(e);
		if (lg >= 10 && lg <= 29)
			ctx->portion = 1ull << lg;
	}
	for (int i = 0; i < n_dev; ++i) {
		const int ord = device_ids ? device_ids[i] : i;
		if (ord < 0 || ord >= count) {
			kmc_hip_destroy(ctx);
			return fail(KMC_HIP_EINVAL, "device ordinal out of range");
		}
		ctx->devs.emplace_back(new Dev());
		ctx->devs[i]->ordinal = ord;
		hipError_t e = hipSetDevice(ord);
		if (e != hipSuccess) {
			kmc_hip_destroy(ctx);
			return fail_hip("hipSetDevice", e);
		}
		if (int rc = set_all_func_attrs()) {
			kmc_hip_destroy(ctx);
			return rc;
		}
		for (auto &s : ctx->devs[i]->slot)
			if (int rc = slot_init(s, ctx->portion)) {
				kmc_hip_destroy(ctx);
				return rc;
			}
	}
	*out = ctx;
	return 0;
}

static void hb_report();
void kmc_hip_destroy(kmc_hip_ctx *ctx)
{
	if (!ctx)
		return;
	hb_report();
	for (auto &d : ctx->devs) {
		(void)hipSetDevice(d->ordinal);
		(void)hipDeviceSynchronize();
		for (auto &s : d->slot)
			slot_destroy(s);
		if (d->rccl_buf.p)
			(void)hipFree(d->rccl_buf.p);
		if (d->xchg.p)
			(void)hipFree(d->xchg.p);
		if (d->d_sig_map)
			(void)hipFree(d->d_sig_map);
		for (auto &a : d->s1_arena)
			if (a.p)
				(void)hipFree(a.p);
	}
	if (ctx->comms_ready)
		for (auto &c : ctx->comms)
			(void)ncclCommDestroy(c);
	delete ctx;
}

int kmc_hip_num_devices(kmc_hip_ctx *ctx) { return ctx ? (int)ctx->devs.size() : 0; }
int kmc_hip_num_slots(void) { return N_SLOTS; }

int kmc_hip_malloc(kmc_hip_ctx *ctx, int dev, uint64_t bytes, void **d_ptr)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	HIPCHK(hipMalloc(d_ptr, bytes ? bytes : 1));
	return 0;
}
int kmc_hip_free(kmc_hip_ctx *ctx, int dev, void *d_ptr)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	HIPCHK(hipFree(d_ptr));
	return 0;
}
int kmc_hip_memcpy_h2d(kmc_hip_ctx *ctx, int dev, void *d_dst, const void *src, uint64_t bytes)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (bytes)
		HIPCHK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
	return 0;
}
int kmc_hip_memcpy_d2h(kmc_hip_ctx *ctx, int dev, void *dst, const void *d_src, uint64_t bytes)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (bytes)
		HIPCHK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
	return 0;
}
int kmc_hip_host_register(kmc_hip_ctx *ctx, void *ptr, uint64_t bytes)
{
	if (!ctx)
		return fail(KMC_HIP_EINVAL, "ctx == NULL");
	HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
	return 0;
}
int kmc_hip_host_unregister(kmc_hip_ctx *ctx, void *ptr)
{
	if (!ctx)
		return fail(KMC_HIP_EINVAL, "ctx == NULL");
	HIPCHK(hipHostUnregister(ptr));
	return 0;
}
int kmc_hip_host_alloc(kmc_hip_ctx *ctx, uint64_t bytes, void **ptr)
{
	if (!ctx || !ptr)
		return fail(KMC_HIP_EINVAL, "kmc_hip_host_alloc: bad arguments");
	HIPCHK(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocPortable));
	return 0;
}
int kmc_hip_host_free(kmc_hip_ctx *ctx, void *ptr)
{
	if (!ctx)
		return fail(KMC_HIP_EINVAL, "ctx == NULL");
	HIPCHK(hipHostFree(ptr));
	return 0;
}
int kmc_hip_synchronize(kmc_hip_ctx *ctx, int dev)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	u32 err = 0;
	for (auto &s : ctx->devs[dev]->slot) {
		std::lock_guard<std::mutex> lck(s.mtx);
		HIPCHK(hipStreamSynchronize(s.stream));
		if (int rc = drain_redo(s))
			return rc;
		if (int rc = harvest(s))
			return rc;
		u32 e1 = 0;
		if (int rc = read_and_clear_sticky(s, e1))
			return rc;
		err |= e1;
	}
	return err_to_code(err);
}

/* ---- narrow boundary ---- */
static int sort_records_device_locked(Slot &s, void *d_recs, void *d_tmp, uint64_t n, uint32_t words, uint32_t key_bytes, void **d_result, int stable_lsd)
{
	s.timed = true;
	u64 *res = nullptr;
	if (int rc = sort_device(s, (u64 *)d_recs, (u64 *)d_tmp, n, words, key_bytes, &res, stable_lsd != 0))
		return rc;
	HIPCHK(hipStreamSynchronize(s.stream));
	bool redo = false;
	if (!stable_lsd && n >= 2)
		if (int rc = read_redo(s, redo))
			return rc;
	if (redo) { /* a tile of the hybrid sort did not fit: the array is still a permutation of the input, LSD passes over all bytes sort it */
		u64 *other = res == (u64 *)d_recs ? (u64 *)d_tmp : (u64 *)d_recs;
		if (int rc = sort_device(s, res, other, n, words, key_bytes, &res, true))
			return rc;
		HIPCHK(hipStreamSynchronize(s.stream));
	}
	if (int rc = harvest(s))
		return rc;
	u32 err = 0;
	if (int rc = read_and_clear_sticky(s, err))
		return rc;
	*d_result = res;
	return err_to_code(err);
}

int kmc_hip_sort_records_device(kmc_hip_ctx *ctx, int dev, void *d_recs, void *d_tmp, uint64_t n, uint32_t words, uint32_t key_bytes,
                                void **d_result)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (words < 1 || words > 8 || key_bytes > 8 * words || !d_result)
		return fail(KMC_HIP_EINVAL, "kmc_hip_sort_records_device: bad arguments");
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx);
	return sort_records_device_locked(s, d_recs, d_tmp, n, words, key_bytes, d_result, 0);
}

int kmc_hip_sort_records_into(kmc_hip_ctx *ctx, int dev, const void *recs, void *dst, uint64_t n, uint32_t words, uint32_t key_bytes)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (words < 1 || words > 8 || key_bytes > 8 * words)
		return fail(KMC_HIP_EINVAL, "kmc_hip_sort_records: bad arguments");
	if (n && (!recs || !dst))
		return fail(KMC_HIP_EINVAL, "recs == NULL");
	const size_t bytes = (size_t)n * words * 8;
	if (n < 2) {
		if (n && dst != recs)
			memcpy(dst, recs, bytes);
		return 0;
	}
	Slot &s = ctx->devs[dev]->slot[0];
	std::lock_guard<std::mutex> lck(s.mtx); /* slot 0's record arrays are the staging area: one host sort at a time per device */
	int rc = 0;
	if ((rc = ensure(s.recA, bytes + 256)) || (rc = ensure(s.recB, bytes + 256)))
		return rc;
	HIPCHK(hipMemcpyAsync(s.recA.p, recs, bytes, hipMemcpyHostToDevice, s.stream));
	void *res = nullptr;
	if ((rc = sort_records_device_locked(s, s.recA.p, s.recB.p, n, words, key_bytes, &res, 0)))
		return rc;
	HIPCHK(hipMemcpyAsync(dst, res, bytes, hipMemcpyDeviceToHost, s.stream));
	HIPCHK(hipStreamSynchronize(s.stream));
	return 0;
}

int kmc_hip_sort_records(kmc_hip_ctx *ctx, int dev, void *recs, uint64_t n, uint32_t words, uint32_t key_bytes)
{
	return kmc_hip_sort_records_into(ctx, dev, recs, recs, n, words, key_bytes);
}

/* ---- full boundary ---- */
static int process_bin_device_on(kmc_hip_ctx *ctx, int dev, Slot &s, const DevParams &P, u64 lut_entries, const uint8_t *d_superkmers,
                                 uint64_t size, uint64_t n_rec, const uint64_t *d_pack_start, uint64_t n_packs, uint8_t *d_out,
                                 uint64_t out_capacity, uint64_t *d_out_bytes, uint64_t *d_lut, uint64_t *d_stats, int sync)
{
	(void)ctx;
	(void)dev;
	std::lock_guard<std::mutex> lck(s.mtx);
	s.timed = sync || (s.async_seq++ % TIMING_SAMPLE) == 0; /* async_seq restarts with kmc_hip_scatter_totals(reset) */
	if (int rc = run_bin_device(s, P, d_superkmers, size, n_rec, (const u64 *)d_pack_start, n_packs, d_out, out_capacity, (u64 *)d_out_bytes,
	                            (u64 *)d_lut, lut_entries, (u64 *)d_stats, false, !sync))
		return rc;
	if (!sync)
		return 0;
	HIPCHK(hipStreamSynchronize(s.stream));
	bool redo = false;
	if (n_rec >= 2)
		if (int rc = read_redo(s, redo))
			return rc;
	if (redo) {
		if (int rc = run_bin_device(s, P, d_superkmers, size, n_rec, (const u64 *)d_pack_start, n_packs, d_out, out_capacity, (u64 *)d_out_bytes,
		                            (u64 *)d_lut, lut_entries, (u64 *)d_stats, true))
			return rc;
		HIPCHK(hipStreamSynchronize(s.stream));
	}
	if (int rc = drain_redo(s)) /* asynchronous groups enqueued on this slot before */
		return rc;
	if (int rc = harvest(s))
		return rc;
	u32 err = 0;
	if (int rc = read_and_clear_sticky(s, err))
		return rc;
	return err_to_code(err);
}

int kmc_hip_process_bin_device(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint8_t *d_superkmers, uint64_t size,
                               uint64_t n_rec, const uint64_t *d_pack_start, uint64_t n_packs, uint8_t *d_out, uint64_t out_capacity,
                               uint64_t *d_out_bytes, uint64_t *d_lut, uint64_t *d_stats, int sync)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if (!d_out_bytes || !d_stats || (size && (!d_superkmers || !d_pack_start)))
		return fail(KMC_HIP_EINVAL, "kmc_hip_process_bin_device: NULL device pointer");
	/* asynchronous calls go round-robin over the device's stream slots, so the launch gaps and serial tails of one
	 * (small) bin are filled by the kernels of the next ones; a synchronous call always uses slot 0, and so does a big bin */
	Dev &d = *ctx->devs[dev];
	int si = 0;
	if (!sync && !is_big(P, n_rec)) {
		std::lock_guard<std::mutex> lck(d.rr_mtx);
		si = (int)(d.rr++ % N_BATCH_STREAMS);
	}
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	return process_bin_device_on(ctx, dev, d.slot[si], P, lut_entries, d_superkmers, size, n_rec, d_pack_start, n_packs, d_out, out_capacity,
	                             d_out_bytes, d_lut, d_stats, sync);
}

int kmc_hip_process_bins_device(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const kmc_hip_bin_desc *bins, uint64_t n_bins,
                                int n_streams)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if (n_bins && !bins)
		return fail(KMC_HIP_EINVAL, "bins == NULL");
	if (n_streams <= 0) {
		/* auto: bins whose record arrays are large fill the GPU on their own, one after the other on ONE stream (2 streams: +2 %
		 * at 48 M k-mers per bin, and the per-launch timings stop meaning anything); small bins need each other's company */
		u64 recs = 0;
		for (uint64_t i = 0; i < n_bins; ++i)
			recs += bins[i].n_rec;
		const u64 avg_bytes = n_bins ? recs / n_bins * (u64)((P.k + 31) / 32) * 8 : 0;
		n_streams = avg_bytes >= (64ull << 20) ? 1 : N_BATCH_STREAMS;
	}
	if (n_streams > N_SLOTS)
		n_streams = N_SLOTS;
	for (uint64_t i = 0; i < n_bins; ++i)
		if (!bins[i].d_out_bytes || !bins[i].d_stats || (bins[i].size && (!bins[i].d_superkmers || !bins[i].d_pack_start)))
			return fail(KMC_HIP_EINVAL, "kmc_hip_process_bins_device: NULL device pointer in a bin descriptor");
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	Dev &d = *ctx->devs[dev];
	/* Bin i goes to stream slot (i mod n_streams), in index order per slot; one host thread per slot enqueues (a group of bins is 13-14 launches:
	 * with hundreds of small bins a single submitting thread is the bottleneck, not the GPU). Big bins all take slot 0. */
	std::vector<int> rcs((size_t)n_streams, 0);
	std::vector<std::string> msgs((size_t)n_streams);
	u64 all_recs = 0;
	for (uint64_t i = 0; i < n_bins; ++i)
		all_recs += bins[i].n_rec;
	const u32 G = group_capacity(P.k, n_bins && all_recs / n_bins < GROUP_SMALL_BIN_RECORDS);
	const u64 rec_bytes_of = (u64)((P.k + 31) / 32) * 8;
	auto work = [&](int t) {
		if (hipSetDevice(d.ordinal) != hipSuccess) {
			rcs[t] = KMC_HIP_EDEVICE;
			msgs[t] = "hipSetDevice failed in a submitting thread";
			return;
		}
		/* this stream's bins, in index order; consecutive ones share one sort (run_group_device_t) while the group's record array stays small */
		std::vector<const kmc_hip_bin_desc *> grp;
		u64 grp_recs = 0;
		auto flush = [&]() -> int {
			int rc = 0;
			if (grp.size() == 1) {
				const kmc_hip_bin_desc &b = *grp[0];
				rc = process_bin_device_on(ctx, dev, d.slot[t], P, lut_entries, b.d_superkmers, b.size, b.n_rec, b.d_pack_start, b.n_packs, b.d_out, b.out_capacity,
				                           b.d_out_bytes, b.d_lut, b.d_stats, 0);
			} else if (grp.size() > 1) {
				Slot &sl = d.slot[t];
				std::lock_guard<std::mutex> lck(sl.mtx);
				sl.timed = (sl.async_seq++ % TIMING_SAMPLE) == 0;
				rc = run_group_async(sl, P, grp.data(), (u32)grp.size(), lut_entries);
			}
			grp.clear();
			grp_recs = 0;
			return rc;
		};
		for (uint64_t i = 0; i < n_bins; ++i) {
			const kmc_hip_bin_desc &b = bins[i];
			const int si = is_big(P, b.n_rec) ? 0 : (int)(i % (uint64_t)n_streams);
			if (si != t)
				continue;
			int rc = 0;
			if (!grp.empty() && (grp.size() >= G || (grp_recs + b.n_rec) * rec_bytes_of > GROUP_MAX_RECORD_BYTES))
				rc = flush();
			if (!rc) {
				grp.push_back(&b);
				grp_recs += b.n_rec;
				if (G < 2)
					rc = flush();
			}
			if (rc) {
				rcs[t] = rc;
				msgs[t] = g_err;
				return;
			}
		}
		if (int rc = flush()) {
			rcs[t] = rc;
			msgs[t] = g_err;
		}
	};
	if (n_streams == 1 || n_bins < 2) {
		for (int t = 0; t < n_streams; ++t)
			work(t);
	} else {
		std::vector<std::thread> th;
		for (int t = 1; t < n_streams; ++t)
			th.emplace_back(work, t);
		work(0);
		for (auto &x : th)
			x.join();
	}
	for (int t = 0; t < n_streams; ++t)
		if (rcs[t])
			return fail(rcs[t], msgs[t]);
	return 0;
}

/* byte offsets of a bin's expander packs, appended to `ps` (first entry 0, last entry `size`): from the caller's pack sizes, or — none given — by one
 * walk over the image, a boundary every 4096 super-k-mers */
/* Where the host-boundary calls spend their wall time, summed over all slots (KMC_HIP_VERBOSE prints them when the context is destroyed; the drop-in's workers sit in these
 * calls for most of stage 2 — profiles/r05/e2e_large_30gbp.json: 40 s summed over 16 workers for 0.73 s of kernels): [0] pack starts + buffers, [1] staging copy in (pageable
 * callers), [2] enqueue (copies + launches), [3] wait for the kernels (includes the H2D in front of them and every other slot's work queued before), [4] D2H of the exact-size
 * results, [5] staging copy out (pageable callers), [6] calls, [7] redo rounds */
static std::atomic<long long> g_hb_ns[8];
static inline long long hb_now()
{
	return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct HbLap {
	long long t = hb_now();
	void lap(int i)
	{
		const long long n = hb_now();
		g_hb_ns[i] += n - t;
		t = n;
	}
};
static void hb_report()
{
	if (!getenv("KMC_HIP_VERBOSE") || g_hb_ns[6].load() == 0)
		return;
	fprintf(stderr, "[kmc_hip host boundary] %lld calls (%lld redo rounds), seconds summed over slots: pack starts + buffers %.3f, staging copy in %.3f, enqueue %.3f, wait for "
	                "the kernels %.3f, D2H of the results %.3f, staging copy out %.3f\n",
	        g_hb_ns[6].load(), g_hb_ns[7].load(), g_hb_ns[0].load() * 1e-9, g_hb_ns[1].load() * 1e-9, g_hb_ns[2].load() * 1e-9, g_hb_ns[3].load() * 1e-9, g_hb_ns[4].load() * 1e-9,
	        g_hb_ns[5].load() * 1e-9);
}

int kmc_hip_reserve_slot(kmc_hip_ctx *ctx, int dev, int slot, uint64_t bytes)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (slot < 0 || slot >= N_SLOTS)
		return fail(KMC_HIP_EINVAL, "slot out of range (see kmc_hip_num_slots)");
	Slot &s = ctx->devs[dev]->slot[slot];
	std::lock_guard<std::mutex> lck(s.mtx);
	if (s.slab.p)
		return fail(KMC_HIP_EINVAL, "kmc_hip_reserve_slot: the slot has a slab already (buffers may point into it)");
	if (!bytes)
		return 0;
	const size_t want = ((size_t)bytes + 255) & ~(size_t)255;
	size_t free_b = 0, total_b = 0;
	HIPCHK(hipMemGetInfo(&free_b, &total_b));
	if (free_b < want || free_b - want < total_b / 2) /* reservations never take the device below half of its memory (several logical devices or processes on one GPU) */
		return fail(KMC_HIP_ECAPACITY, "kmc_hip_reserve_slot: refused — the device would be left with less than half of its memory free");
	HIPCHK(hipMalloc(&s.slab.p, want));
	s.slab.cap = want;
	s.slab.used = 0;
	/* the slot's pinned staging buffer for results (callers whose output buffers are ordinary memory: the drop-in's arena) comes up with the slab: a bin's counted
	 * records are ~2 % of what its record arrays take, and an allocation of pinned memory at the moment it is first needed waits for the same lock as a device
	 * allocation (30 Gbp: "D2H of the results" 9.6 s summed over 16 workers, most of it this) */
	if (int rc = ensure_pinned(s.h_stage_out, s.h_stage_out_cap, want / 64))
		return rc;
	return 0;
}

int kmc_hip_host_boundary_times(double seconds[8])
{
	if (!seconds)
		return fail(KMC_HIP_EINVAL, "seconds == NULL");
	for (int i = 0; i < 6; ++i)
		seconds[i] = g_hb_ns[i].load() * 1e-9;
	seconds[6] = (double)g_hb_ns[6].load();
	seconds[7] = (double)g_hb_ns[7].load();
	return 0;
}

static int append_pack_starts(const DevParams &P, const uint8_t *superkmers, u64 size, const uint64_t *pack_bytes, u64 n_packs, std::vector<u64> &ps)
{
	if (!size)
		return 0;
	ps.push_back(0);
	if (n_packs) {
		u64 acc = 0;
		for (u64 i = 0; i < n_packs; ++i) {
			if (pack_bytes[i] == 0)
				continue;
			acc += pack_bytes[i];
			ps.push_back(acc);
		}
		if (acc != size)
			return fail(KMC_HIP_ECORRUPT, "sum of pack_bytes != size");
		return 0;
	}
	u64 pos = 0;
	u32 in_pack = 0;
	while (pos < size) {
		const u32 e = superkmers[pos];
		pos += 1 + (P.k + e + 3) / 4;
		if (++in_pack == 4096 && pos < size) {
			ps.push_back(pos);
			in_pack = 0;
		}
	}
	if (pos != size)
		return fail(KMC_HIP_ECORRUPT, "super-k-mer stream is ragged");
	ps.push_back(size);
	return 0;
}

/* the kernels of the host-boundary bin whose image is in s.in, and the copy of its results block to pinned memory (caller holds s.mtx) */
static int enqueue_host_bin(Slot &s, bool classic)
{
	const DevParams &P = s.sub_P;
	if (int rc = run_bin_device(s, P, (const uint8_t *)s.in.p, s.sub_size, s.sub_n_rec, (const u64 *)s.pack_start.p, s.sub_np, (uint8_t *)s.out.p,
	                            P.without_output ? 0 : s.out_capacity, nullptr /* out_bytes and stats: the slot's small block */, (u64 *)s.lut.p,
	                            s.lut_entries, nullptr, classic))
		return rc;
	if (s.sub_n_rec == 0) /* the empty-bin path does not touch the small block */
		HIPCHK(hipMemsetAsync(s.zero.p, 0, 64, s.stream));
	HIPCHK(hipMemcpyAsync(small_ptr<u32>(s, SM_ERR), s.sticky.p, 4, hipMemcpyDeviceToDevice, s.stream));
	HIPCHK(hipMemcpyAsync(s.h_res, s.zero.p, sizeof(HostRes), hipMemcpyDeviceToHost, s.stream));
	HIPCHK(hipEventRecord(s.done_ev, s.stream));
	return 0;
}

int kmc_hip_process_bin_submit(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_bin_params *params, const uint8_t *superkmers,
                               uint64_t size, uint64_t n_rec, const uint64_t *pack_bytes, uint64_t n_packs, uint8_t *out_suffix,
                               uint64_t out_capacity, uint64_t *lut)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (slot < 0 || slot >= N_SLOTS)
		return fail(KMC_HIP_EINVAL, "slot out of range (see kmc_hip_num_slots)");
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	Slot &s = ctx->devs[dev]->slot[slot];
	std::lock_guard<std::mutex> lck(s.mtx);
	SlabScope slab_scope(s.slab);
	if (s.pending || s.hb_pending)
		return fail(KMC_HIP_EINVAL, "slot already has a bin in flight");
	if (size && !superkmers)
		return fail(KMC_HIP_EINVAL, "superkmers == NULL");
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	if (!P.without_output && ((out_capacity && !out_suffix) || (lut_entries && !lut)))
		return fail(KMC_HIP_EINVAL, "output buffers missing");

	/* pack starts (byte offsets). Without packs from the caller, walk the image once on the host. */
	HbLap lap;
	++g_hb_ns[6];
	std::vector<u64> &ps = s.h_pack_start;
	ps.clear();
	if (size) {
		ps.push_back(0);
		if (n_packs) {
			u64 acc = 0;
			for (u64 i = 0; i < n_packs; ++i) {
				if (pack_bytes[i] == 0)
					continue;
				acc += pack_bytes[i];
				ps.push_back(acc);
			}
			if (acc != size)
				return fail(KMC_HIP_ECORRUPT, "sum of pack_bytes != size");
		} else {
			u64 pos = 0;
			u32 in_pack = 0;
			while (pos < size) {
				const u32 e = superkmers[pos];
				pos += 1 + (P.k + e + 3) / 4;
				if (++in_pack == 4096 && pos < size) {
					ps.push_back(pos);
					in_pack = 0;
				}
			}
			if (pos != size)
				return fail(KMC_HIP_ECORRUPT, "super-k-mer stream is ragged");
			ps.push_back(size);
		}
	}
	const u64 np = ps.empty() ? 0 : ps.size() - 1;
	int rc = 0;
	if ((rc = ensure(s.in, size + 256)) || (rc = ensure(s.pack_start, (np + 1) * 8)) ||
	    (rc = ensure(s.out, (P.without_output ? 0 : out_capacity) + 256)) || (rc = ensure(s.lut, lut_entries * 8 + 256)))
		return rc;
	lap.lap(0);
	if (size) {
		const void *src = superkmers;
		if (!host_ptr_is_pinned(superkmers)) { /* pageable caller (the drop-in's arena): through the slot's pinned staging buffer */
			if ((rc = ensure_pinned(s.h_stage_in, s.h_stage_in_cap, size)))
				return rc;
			memcpy(s.h_stage_in, superkmers, size);
			src = s.h_stage_in;
			lap.lap(1);
		}
		HIPCHK(hipMemcpyAsync(s.in.p, src, size, hipMemcpyHostToDevice, s.stream));
		HIPCHK(hipMemsetAsync((char *)s.in.p + size, 0, 256, s.stream));
		HIPCHK(hipMemcpyAsync(s.pack_start.p, ps.data(), (np + 1) * 8, hipMemcpyHostToDevice, s.stream));
	}
	/* staged when EITHER destination is ordinary memory (the records and the LUT may come from different allocations), as the several-bins path decides it */
	s.out_staged = !P.without_output && ((out_capacity && !host_ptr_is_pinned((const void *)out_suffix)) || (lut_entries && !host_ptr_is_pinned((const void *)lut)));
	s.timed = true;
	s.sub_P = P;
	s.sub_size = size;
	s.sub_n_rec = n_rec;
	s.sub_np = np;
	s.out_capacity = out_capacity;
	s.lut_entries = lut_entries;
	if ((rc = enqueue_host_bin(s, false)))
		return rc;
	lap.lap(2);
	s.pending = true;
	s.h_out = out_suffix;
	s.h_lut = (u64 *)lut;
	s.out_capacity = out_capacity;
	s.lut_entries = lut_entries;
	s.without_output = P.without_output != 0;
	return 0;
}

int kmc_hip_process_bin_wait(kmc_hip_ctx *ctx, int dev, int slot, uint64_t *out_bytes, uint64_t stats[4])
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (slot < 0 || slot >= N_SLOTS)
		return fail(KMC_HIP_EINVAL, "slot out of range (see kmc_hip_num_slots)");
	Slot &s = ctx->devs[dev]->slot[slot];
	std::lock_guard<std::mutex> lck(s.mtx);
	SlabScope slab_scope(s.slab);
	if (!s.pending)
		return fail(KMC_HIP_EINVAL, "no bin in flight on this slot");
	s.pending = false;
	HbLap lap;
	HIPCHK(hipEventSynchronize(s.done_ev)); /* blocks in the kernel driver instead of spinning */
	if (int rc = harvest(s))
		return rc;
	lap.lap(3);
	HostRes r = *s.h_res;
	if (r.redo && !(r.err & ~KERR_CAPACITY)) { /* the hybrid sort met a tile it could not handle: the bin again (its image is still in s.in), LSD passes over every byte.
		                                      * A capacity error of the first attempt does not count: a tile that was handed back may have been compacted unsorted */
		note_redo();
		raise_top();
		if (r.err)
			if (int rc = clear_sticky(s, r.err))
				return rc;
		if (int rc = enqueue_host_bin(s, true))
			return rc;
		HIPCHK(hipEventSynchronize(s.done_ev));
		r = *s.h_res;
		++g_hb_ns[7];
		lap.lap(3);
	}
	if (r.err) {
		if (int rc = clear_sticky(s, r.err))
			return rc;
		return err_to_code(r.err);
	}
	if (r.out_bytes > s.out_capacity)
		return fail(KMC_HIP_ECAPACITY, "out_capacity too small for the counted k-mers");
	if (!s.without_output) {
		/* exact-size copies: out_bytes is only known now (the capacity is ~10x the counted bytes at the default cutoffs) */
		uint8_t *dst_out = s.h_out;
		u64 *dst_lut = s.h_lut;
		const size_t lut_bytes = (size_t)s.lut_entries * 8;
		if (s.out_staged) { /* pageable caller: records and LUT land in the slot's pinned staging buffer and are copied on from there */
			if (int rc = ensure_pinned(s.h_stage_out, s.h_stage_out_cap, r.out_bytes + lut_bytes + 16))
				return rc;
			dst_lut = (u64 *)s.h_stage_out;
			dst_out = (uint8_t *)s.h_stage_out + lut_bytes;
		}
		if (r.out_bytes)
			HIPCHK(hipMemcpyAsync(dst_out, s.out.p, r.out_bytes, hipMemcpyDeviceToHost, s.stream));
		if (s.lut_entries)
			HIPCHK(hipMemcpyAsync(dst_lut, s.lut.p, lut_bytes, hipMemcpyDeviceToHost, s.stream));
		HIPCHK(hipEventRecord(s.done_ev, s.stream));
		HIPCHK(hipEventSynchronize(s.done_ev));
		lap.lap(4);
		if (s.out_staged) {
			if (r.out_bytes)
				memcpy(s.h_out, dst_out, r.out_bytes);
			if (lut_bytes)
				memcpy(s.h_lut, dst_lut, lut_bytes);
			lap.lap(5);
		}
	}
	if (out_bytes)
		*out_bytes = r.out_bytes;
	if (stats)
		for (int i = 0; i < 4; ++i)
			stats[i] = r.stats[i];
	return 0;
}

int kmc_hip_process_bin(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint8_t *superkmers, uint64_t size,
                        uint64_t n_rec, const uint64_t *pack_bytes, uint64_t n_packs, uint8_t *out_suffix, uint64_t out_capacity,
                        uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4])
{
	if (int rc = kmc_hip_process_bin_submit(ctx, dev, 0, params, superkmers, size, n_rec, pack_bytes, n_packs, out_suffix, out_capacity, lut))
		return rc;
	return kmc_hip_process_bin_wait(ctx, dev, 0, out_bytes, stats);
}

/* ---- host-boundary GROUPS: up to HB_MAX bins per call, sorted together like the bins of kmc_hip_process_bins_device ---- */
static int hb_enqueue_results(Slot &s)
{
	HbRes *res = (HbRes *)s.hb_res.p;
	HIPCHK(hipMemcpyAsync(&res->err, s.sticky.p, 4, hipMemcpyDeviceToDevice, s.stream));
	HIPCHK(hipMemcpyAsync(s.h_hb_res, s.hb_res.p, sizeof(HbRes), hipMemcpyDeviceToHost, s.stream));
	HIPCHK(hipEventRecord(s.done_ev, s.stream));
	return 0;
}

int kmc_hip_process_bins_submit(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_bin_params *params, const kmc_hip_host_bin *bins, uint32_t n_bins)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (slot < 0 || slot >= N_SLOTS)
		return fail(KMC_HIP_EINVAL, "slot out of range (see kmc_hip_num_slots)");
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if (!bins || n_bins < 1 || n_bins > HB_MAX)
		return fail(KMC_HIP_EINVAL, "kmc_hip_process_bins_submit: 1..16 bins per call");
	Slot &s = ctx->devs[dev]->slot[slot];
	std::lock_guard<std::mutex> lck(s.mtx);
	SlabScope slab_scope(s.slab);
	if (s.pending || s.hb_pending)
		return fail(KMC_HIP_EINVAL, "slot already has a bin in flight");
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	const u64 lut_pitch = up256(lut_entries * 8);
	HbLap lap;
	++g_hb_ns[6];
	std::vector<u64> &ps = s.h_pack_start;
	ps.clear();
	std::vector<u64> in_off(n_bins), out_off(n_bins), ps_off(n_bins), np(n_bins);
	u64 in_total = 0, out_total = 0, recs = 0;
	for (u32 i = 0; i < n_bins; ++i) {
		const kmc_hip_host_bin &b = bins[i];
		if (b.size && !b.superkmers)
			return fail(KMC_HIP_EINVAL, "superkmers == NULL");
		if (!P.without_output && ((b.out_capacity && !b.out_suffix) || (lut_entries && !b.lut)))
			return fail(KMC_HIP_EINVAL, "output buffers missing");
		if ((b.n_rec == 0) != (b.size == 0))
			return fail(KMC_HIP_ECORRUPT, "exactly one of size / n_rec is zero");
		ps_off[i] = ps.size();
		if (int rc = append_pack_starts(P, b.superkmers, b.size, b.pack_bytes, b.n_packs, ps))
			return rc;
		np[i] = ps.size() > ps_off[i] ? ps.size() - ps_off[i] - 1 : 0;
		in_off[i] = in_total;
		in_total += up256(b.size + 256);
		out_off[i] = out_total;
		out_total += up256((P.without_output ? 0 : b.out_capacity) + 256);
		recs += b.n_rec;
	}
	int rc = 0;
	if ((rc = ensure(s.in, in_total + 256)) || (rc = ensure(s.pack_start, (ps.size() + 1) * 8)) || (rc = ensure(s.out, out_total + 256)) ||
	    (rc = ensure(s.lut, (u64)n_bins * lut_pitch + 256)) || (rc = ensure(s.hb_res, sizeof(HbRes))))
		return rc;
	if (!s.h_hb_res)
		HIPCHK(hipHostMalloc((void **)&s.h_hb_res, sizeof(HbRes), hipHostMallocDefault));
	HbRes *res = (HbRes *)s.hb_res.p;
	HIPCHK(hipMemsetAsync(res, 0, sizeof(HbRes), s.stream));
	if (!ps.empty())
		HIPCHK(hipMemcpyAsync(s.pack_start.p, ps.data(), ps.size() * 8, hipMemcpyHostToDevice, s.stream));
	s.hb.resize(n_bins);
	bool stage_in = false;
	s.out_staged = false;
	for (u32 i = 0; i < n_bins; ++i) { /* one pageable buffer among the call's: everything of the call goes through the pinned staging buffers */
		stage_in = stage_in || (bins[i].size && !host_ptr_is_pinned(bins[i].superkmers));
		if (!P.without_output)
			s.out_staged = s.out_staged || (bins[i].out_capacity && !host_ptr_is_pinned(bins[i].out_suffix)) || (lut_entries && !host_ptr_is_pinned(bins[i].lut));
	}
	if (stage_in && (rc = ensure_pinned(s.h_stage_in, s.h_stage_in_cap, in_total + 256)))
		return rc;
	lap.lap(0);
	for (u32 i = 0; i < n_bins; ++i) {
		const kmc_hip_host_bin &b = bins[i];
		uint8_t *d_img = (uint8_t *)s.in.p + in_off[i];
		if (b.size) {
			const void *src = b.superkmers;
			if (stage_in) {
				memcpy((char *)s.h_stage_in + in_off[i], b.superkmers, b.size);
				src = (char *)s.h_stage_in + in_off[i];
				lap.lap(1);
			}
			HIPCHK(hipMemcpyAsync(d_img, src, b.size, hipMemcpyHostToDevice, s.stream));
			HIPCHK(hipMemsetAsync(d_img + b.size, 0, 256, s.stream));
		}
		Slot::HostBin &h = s.hb[i];
		h.d.d_superkmers = d_img;
		h.d.size = b.size;
		h.d.n_rec = b.n_rec;
		h.d.d_pack_start = (const uint64_t *)s.pack_start.p + ps_off[i];
		h.d.n_packs = np[i];
		h.d.d_out = (uint8_t *)s.out.p + out_off[i];
		h.d.out_capacity = P.without_output ? 0 : b.out_capacity;
		h.d.d_out_bytes = (uint64_t *)&res->w[i][0];
		h.d.d_stats = (uint64_t *)&res->w[i][1];
		h.d.d_lut = (uint64_t *)((char *)s.lut.p + (u64)i * lut_pitch);
		h.h_out = b.out_suffix;
		h.h_lut = (u64 *)b.lut;
	}
	/* sort groups: as many consecutive bins as the spare bits of the top digit can tag (group_capacity), while the record array stays moderate */
	const u32 G = group_capacity(P.k, recs / n_bins < GROUP_SMALL_BIN_RECORDS);
	const u64 rec_bytes_of = (u64)((P.k + 31) / 32) * 8;
	s.hb_chunks.clear();
	s.hb_hybrid.clear();
	s.timed = true;
	s.hb_P = P;
	s.lut_entries = lut_entries;
	s.without_output = P.without_output != 0;
	for (u32 first = 0; first < n_bins;) {
		u32 cnt = 0;
		u64 grp_recs = 0;
		while (first + cnt < n_bins && cnt < G && (cnt == 0 || (grp_recs + bins[first + cnt].n_rec) * rec_bytes_of <= GROUP_MAX_RECORD_BYTES))
			grp_recs += bins[first + cnt++].n_rec;
		const kmc_hip_bin_desc *ptrs[HB_MAX];
		for (u32 j = 0; j < cnt; ++j)
			ptrs[j] = &s.hb[first + j].d;
		bool hyb = false;
		if ((rc = run_group_device(s, P, ptrs, cnt, lut_entries, false, &res->flag[s.hb_chunks.size()], &hyb)))
			return rc;
		s.hb_chunks.emplace_back(first, cnt);
		s.hb_hybrid.push_back(hyb ? 1 : 0);
		first += cnt;
	}
	if ((rc = hb_enqueue_results(s)))
		return rc;
	lap.lap(2);
	s.hb_pending = true;
	return 0;
}

int kmc_hip_process_bins_wait(kmc_hip_ctx *ctx, int dev, int slot, uint64_t *out_bytes, uint64_t *stats)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (slot < 0 || slot >= N_SLOTS)
		return fail(KMC_HIP_EINVAL, "slot out of range (see kmc_hip_num_slots)");
	Slot &s = ctx->devs[dev]->slot[slot];
	std::lock_guard<std::mutex> lck(s.mtx);
	SlabScope slab_scope(s.slab);
	if (!s.hb_pending)
		return fail(KMC_HIP_EINVAL, "no group of bins in flight on this slot");
	s.hb_pending = false;
	HbLap lap;
	HIPCHK(hipEventSynchronize(s.done_ev));
	if (int rc = harvest(s))
		return rc;
	lap.lap(3);
	const HbRes &r = *s.h_hb_res;
	bool any_flag = false;
	for (size_t c = 0; c < s.hb_chunks.size(); ++c)
		any_flag = any_flag || (s.hb_hybrid[c] && r.flag[c]);
	if (!(r.err & ~(any_flag ? KERR_CAPACITY : 0u))) { /* sort groups whose hybrid sort met a tile it could not handle: again (the images are still in s.in), LSD passes over
		                                                 * every byte; a capacity error next to a flag is the first attempt's (a tile handed back may have been compacted unsorted) */
		bool any = false;
		if (r.err)
			if (int rc = clear_sticky(s, r.err))
				return rc;
		for (size_t c = 0; c < s.hb_chunks.size(); ++c) {
			if (!s.hb_hybrid[c] || !r.flag[c])
				continue;
			any = true;
			note_redo();
			const kmc_hip_bin_desc *ptrs[HB_MAX];
			for (u32 j = 0; j < s.hb_chunks[c].second; ++j)
				ptrs[j] = &s.hb[s.hb_chunks[c].first + j].d;
			s.timed = false;
			if (int rc = run_group_device(s, s.hb_P, ptrs, s.hb_chunks[c].second, s.lut_entries, true))
				return rc;
		}
		if (any) {
			raise_top();
			if (int rc = hb_enqueue_results(s))
				return rc;
			HIPCHK(hipEventSynchronize(s.done_ev));
			++g_hb_ns[7];
			lap.lap(3);
		}
	}
	if (r.err) {
		if (int rc = clear_sticky(s, r.err))
			return rc;
		return err_to_code(r.err);
	}
	const size_t n = s.hb.size();
	for (size_t i = 0; i < n; ++i)
		if (r.w[i][0] > s.hb[i].d.out_capacity && !s.without_output)
			return fail(KMC_HIP_ECAPACITY, "out_capacity too small for the counted k-mers");
	if (!s.without_output) { /* exact-size copies */
		const size_t lut_bytes = (size_t)s.lut_entries * 8;
		std::vector<size_t> off(n + 1, 0);
		if (s.out_staged) {
			for (size_t i = 0; i < n; ++i)
				off[i + 1] = off[i] + (((size_t)r.w[i][0] + lut_bytes + 63) & ~(size_t)63);
			if (int rc = ensure_pinned(s.h_stage_out, s.h_stage_out_cap, off[n] + 64))
				return rc;
		}
		for (size_t i = 0; i < n; ++i) {
			uint8_t *dst_out = s.out_staged ? (uint8_t *)s.h_stage_out + off[i] + lut_bytes : s.hb[i].h_out;
			u64 *dst_lut = s.out_staged ? (u64 *)((uint8_t *)s.h_stage_out + off[i]) : s.hb[i].h_lut;
			if (r.w[i][0])
				HIPCHK(hipMemcpyAsync(dst_out, s.hb[i].d.d_out, r.w[i][0], hipMemcpyDeviceToHost, s.stream));
			if (s.lut_entries)
				HIPCHK(hipMemcpyAsync(dst_lut, s.hb[i].d.d_lut, lut_bytes, hipMemcpyDeviceToHost, s.stream));
		}
		HIPCHK(hipEventRecord(s.done_ev, s.stream));
		HIPCHK(hipEventSynchronize(s.done_ev));
		lap.lap(4);
		if (s.out_staged) {
			for (size_t i = 0; i < n; ++i) {
				if (r.w[i][0])
					memcpy(s.hb[i].h_out, (uint8_t *)s.h_stage_out + off[i] + lut_bytes, r.w[i][0]);
				if (lut_bytes)
					memcpy(s.hb[i].h_lut, (uint8_t *)s.h_stage_out + off[i], lut_bytes);
			}
			lap.lap(5);
		}
	}
	for (size_t i = 0; i < n; ++i) {
		if (out_bytes)
			out_bytes[i] = r.w[i][0];
		if (stats)
			for (int q = 0; q < 4; ++q)
				stats[4 * i + q] = r.w[i][1 + q];
	}
	return 0;
}

int kmc_hip_process_bin_multi(kmc_hip_ctx *ctx, const kmc_hip_bin_params *params, const uint8_t *superkmers, uint64_t size, uint64_t n_rec,
                              const uint64_t *pack_bytes, uint64_t n_packs, uint8_t *out_suffix, uint64_t out_capacity, uint64_t *out_bytes, uint64_t *lut,
                              uint64_t stats[4])
{
	if (!ctx || ctx->devs.empty())
		return fail(KMC_HIP_EINVAL, "bad ctx");
	DevParams P;
	if (int rc = check_params(params, P))
		return rc;
	if (!out_bytes || !stats || (size && !superkmers))
		return fail(KMC_HIP_EINVAL, "kmc_hip_process_bin_multi: NULL argument");
	if ((n_rec == 0) != (size == 0))
		return fail(KMC_HIP_ECORRUPT, "exactly one of size / n_rec is zero");
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	if (!P.without_output && ((out_capacity && !out_suffix) || (lut_entries && !lut)))
		return fail(KMC_HIP_EINVAL, "output buffers missing");
	std::lock_guard<std::mutex> lck(ctx->mtx);
	if (!size) {
		*out_bytes = 0;
		for (int i = 0; i < 4; ++i)
			stats[i] = 0;
		if (lut_entries && !P.without_output)
			memset(lut, 0, lut_entries * 8);
		return 0;
	}
	std::vector<u64> ps(1, 0);
	if (n_packs) {
		for (u64 i = 0; i < n_packs; ++i)
			if (pack_bytes[i])
				ps.push_back(ps.back() + pack_bytes[i]);
		if (ps.back() != size)
			return fail(KMC_HIP_ECORRUPT, "sum of pack_bytes != size");
	} else {
		u64 pos = 0;
		u32 in_pack = 0;
		while (pos < size) {
			pos += 1 + (P.k + superkmers[pos] + 3) / 4;
			if (++in_pack == 4096 && pos < size) {
				ps.push_back(pos);
				in_pack = 0;
			}
		}
		if (pos != size)
			return fail(KMC_HIP_ECORRUPT, "super-k-mer stream is ragged");
		ps.push_back(size);
	}
	switch ((P.k + 31) / 32) {
	case 1: return process_bin_multi_t<1>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 2: return process_bin_multi_t<2>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 3: return process_bin_multi_t<3>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 4: return process_bin_multi_t<4>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 5: return process_bin_multi_t<5>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 6: return process_bin_multi_t<6>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 7: return process_bin_multi_t<7>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	case 8: return process_bin_multi_t<8>(ctx, P, lut_entries, superkmers, size, n_rec, ps, out_suffix, out_capacity, (u64 *)out_bytes, (u64 *)lut, (u64 *)stats);
	}
	return fail(KMC_HIP_EINVAL, "kmer_len out of range");
}
