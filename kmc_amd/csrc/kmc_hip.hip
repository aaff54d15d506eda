/*
 * kmc_amd/csrc/kmc_hip.hip — host side of libkmc_hip.so: the C-ABI of include/kmc_hip.h over the gfx950
 * kernels in kernels.hip.h. One context owns, per device, two "slots" (stream + grow-only HBM buffers), so the
 * C++ worker can keep two bins in flight (H2D of bin i+1 under the kernels of bin i).
 *
 * Reference mapping: this file plays the role of CKmerBinSorter<SIZE>::ProcessBins' body
 * (kmc_core/kb_sorter.h:210-237): Expand -> Sort -> Compact for one bin, but as a queue of kernels on a
 * HIP stream. No CPU fallback exists: if HIP is unusable every entry point fails with KMC_HIP_EDEVICE.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kmc_hip.h"
#include "kernels.hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
	g_err = msg;
	return code;
}
int fail_hip(const char *what, hipError_t e)
{
	g_err = std::string(what) + ": " + hipGetErrorString(e);
	return e == hipErrorOutOfMemory ? KMC_HIP_ENOMEM : KMC_HIP_EDEVICE;
}
#define HIPCHK(call)                                                                                                   \
	do {                                                                                                               \
		hipError_t e__ = (call);                                                                                       \
		if (e__ != hipSuccess)                                                                                         \
			return fail_hip(#call, e__);                                                                               \
	} while (0)

struct DBuf {
	void *p = nullptr;
	size_t cap = 0;
};

int ensure(DBuf &b, size_t bytes)
{
	if (bytes <= b.cap)
		return 0;
	if (b.p) {
		HIPCHK(hipFree(b.p));
		b.p = nullptr;
		b.cap = 0;
	}
	size_t want = (bytes + 255) & ~(size_t)255;
	HIPCHK(hipMalloc(&b.p, want));
	b.cap = want;
	return 0;
}

/* layout of the per-slot "small" device block (bytes) */
constexpr size_t SM_TOTALS = 0;      /* u64[2]  #super-k-mers, #k-mers   */
constexpr size_t SM_STATS = 16;      /* u64[4]                           */
constexpr size_t SM_OUTBYTES = 48;   /* u64                              */
constexpr size_t SM_ERR = 56;        /* u32                              */
constexpr size_t SM_DBASE_WORK = 256; /* u64[2][256] per-portion digit bases (ping-pong) */
constexpr size_t SM_SHARDS = 256 + 2 * 256 * 8;   /* u64[CP_SHARDS][4] tally shards of the compaction */
constexpr size_t SM_COUNTERS = SM_SHARDS + CP_SHARDS * 4 * 8; /* u32[N_COUNTERS] ticket counters, one per launch */
constexpr size_t N_COUNTERS = 4096;
constexpr size_t SM_BYTES = SM_COUNTERS + N_COUNTERS * 4;

#ifndef KMC_N_SLOTS
#define KMC_N_SLOTS 8 /* 512 bins of 3.2 M k-mers: 1 slot 237 ms, 2 slots 133, 4 slots 104, 8 slots 95 (then the host launch rate binds) */
#endif
constexpr int N_SLOTS = KMC_N_SLOTS;
constexpr u64 PORTION_MAX = 1ull << 29; /* records per scatter launch (30-bit look-back counts) */
/* Tests shrink the portion ($KMC_HIP_DEBUG_PORTION_LOG2, 10..29, read at kmc_hip_init) so that a small, oracle-checkable sort
 * crosses many portion boundaries (digit bases carried from launch to launch) — the path a bin of > 2^29 k-mers takes. */
static u64 PORTION = PORTION_MAX;

struct HostRes {
	u64 totals[2];
	u64 stats[4];
	u64 out_bytes;
	u32 err;
	u32 pad;
};

struct Slot {
	hipStream_t stream = nullptr;
	DBuf in, pack_start, bitmap;
	DBuf recA, recB, ghist, dbase, status, out, lut, lutsh, small;
	HostRes *h_res = nullptr; /* pinned */
	hipEvent_t ev[6] = {};
	std::vector<hipEvent_t> sc_ev;
	u32 sc_used = 0;
	u64 sc_keys = 0;
	bool timed = false;
	/* pending async bin */
	bool pending = false;
	uint8_t *h_out = nullptr;
	u64 *h_lut = nullptr;
	u64 out_capacity = 0, lut_entries = 0;
	bool without_output = false;
	std::vector<u64> h_pack_start;
};

struct Dev {
	int ordinal = 0;
	u32 rr = 0; /* round-robin slot choice of asynchronous device-resident calls */
	Slot slot[N_SLOTS]; /* [0],[1]: the submit/wait slots of the host-buffer API; all of them: round-robin for async device-resident calls */
	DBuf rccl_buf;
};

u32 counter_bytes(u64 cutoff_max, u64 counter_max) { return kmc_counter_bytes(cutoff_max, counter_max); }

} // namespace

struct kmc_hip_ctx {
	std::vector<Dev> devs;
	std::vector<ncclComm_t> comms;
	bool comms_ready = false;
	std::mutex mtx;
};

namespace {

int set_dev(kmc_hip_ctx *ctx, int dev)
{
	if (!ctx || dev < 0 || dev >= (int)ctx->devs.size())
		return fail(KMC_HIP_EINVAL, "bad ctx/dev");
	HIPCHK(hipSetDevice(ctx->devs[dev].ordinal));
	return 0;
}

int slot_init(Slot &s)
{
	HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
	HIPCHK(hipHostMalloc((void **)&s.h_res, sizeof(HostRes), hipHostMallocDefault));
	memset(s.h_res, 0, sizeof(HostRes));
	for (auto &e : s.ev)
		HIPCHK(hipEventCreate(&e));
	if (int rc = ensure(s.small, SM_BYTES))
		return rc;
	return 0;
}

void slot_destroy(Slot &s)
{
	for (DBuf *b : {&s.in, &s.pack_start, &s.bitmap, &s.recA, &s.recB, &s.ghist, &s.dbase, &s.status, &s.out, &s.lut, &s.lutsh, &s.small})
		if (b->p)
			(void)hipFree(b->p);
	if (s.h_res)
		(void)hipHostFree(s.h_res);
	for (auto &e : s.ev)
		if (e)
			(void)hipEventDestroy(e);
	for (auto &e : s.sc_ev)
		(void)hipEventDestroy(e);
	if (s.stream)
		(void)hipStreamDestroy(s.stream);
}

template <typename T> T *small_ptr(Slot &s, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(s.small.p) + off); }

int sc_event(Slot &s, hipEvent_t &e)
{
	if (s.sc_used == s.sc_ev.size()) {
		hipEvent_t ne;
		HIPCHK(hipEventCreate(&ne));
		s.sc_ev.push_back(ne);
	}
	e = s.sc_ev[s.sc_used++];
	return 0;
}

/* ---- the sort: histogram of every digit + n_pass onesweep launches (per portion) --------------------------- */
template <int SIZE>
int sort_device_t(Slot &s, u64 *d_recs, u64 *d_tmp, u64 n, u32 n_pass, u64 **d_result, u32 &counter_idx, bool hist_done = false)
{
	u64 *src = d_recs, *dst = d_tmp;
	if (n < 2 || n_pass == 0) {
		*d_result = src;
		return 0;
	}
	if (int rc = ensure(s.ghist, (size_t)n_pass * 256 * 8))
		return rc;
	if (int rc = ensure(s.dbase, (size_t)n_pass * 256 * 8))
		return rc;
	const u64 max_tiles = (std::min(n, PORTION) + RsCfg<SIZE>::TILE - 1) / RsCfg<SIZE>::TILE;
	if (int rc = ensure(s.status, (size_t)max_tiles * 256 * 4))
		return rc;
	u64 *ghist = (u64 *)s.ghist.p, *dbase = (u64 *)s.dbase.p;
	u32 *status = (u32 *)s.status.p;
	u32 *err = small_ptr<u32>(s, SM_ERR);
	u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
	u64 *work = small_ptr<u64>(s, SM_DBASE_WORK);

	if (rs_lds_bytes<SIZE>() > 65536) {
		static bool attr_done = false; /* per instantiation */
		if (!attr_done) {
			HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_onesweep<SIZE>), hipFuncAttributeMaxDynamicSharedMemorySize,
			                           (int)rs_lds_bytes<SIZE>()));
			attr_done = true;
		}
	}
	if (!hist_done) {
		HIPCHK(hipMemsetAsync(ghist, 0, (size_t)n_pass * 256 * 8, s.stream));
		u64 blocks = (n + 255) / 256;
		if (blocks > 256 * 8)
			blocks = 256 * 8; /* 8 workgroups per CU, grid-stride */
		k_hist<SIZE><<<dim3((u32)blocks), dim3(256), (size_t)n_pass * 1024, s.stream>>>(src, n, n_pass, ghist);
	}
	k_hist_scan<<<dim3(n_pass), dim3(256), 0, s.stream>>>(ghist, dbase);
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[3], s.stream));
	for (u32 pass = 0; pass < n_pass; ++pass) {
		const u64 *base_in = dbase + (size_t)pass * 256;
		int flip = 0;
		for (u64 start = 0; start < n; start += PORTION) {
			const u32 cnt = (u32)std::min(PORTION, n - start);
			const u32 tiles = (cnt + RsCfg<SIZE>::TILE - 1) / RsCfg<SIZE>::TILE;
			if (counter_idx >= N_COUNTERS)
				return fail(KMC_HIP_EINVAL, "too many scatter launches for one bin");
			HIPCHK(hipMemsetAsync(status, 0, (size_t)tiles * 256 * 4, s.stream));
			u64 *base_out = work + (size_t)flip * 256;
			hipEvent_t e0 = nullptr, e1 = nullptr;
			if (s.timed) {
				if (int rc = sc_event(s, e0))
					return rc;
				if (int rc = sc_event(s, e1))
					return rc;
				HIPCHK(hipEventRecord(e0, s.stream));
			}
			k_onesweep<SIZE><<<dim3((tiles + RS_TPB - 1) / RS_TPB), dim3(RS_BLOCK), rs_lds_bytes<SIZE>(), s.stream>>>(
			    src + start * SIZE, dst, cnt, pass, base_in, base_out, status, counters + counter_idx, tiles, err);
			if (s.timed)
				HIPCHK(hipEventRecord(e1, s.stream));
			++counter_idx;
			s.sc_keys += cnt;
			base_in = base_out;
			flip ^= 1;
		}
		std::swap(src, dst);
	}
	HIPCHK(hipGetLastError());
	*d_result = src;
	return 0;
}

int sort_device(Slot &s, u64 *d_recs, u64 *d_tmp, u64 n, u32 words, u32 n_pass, u64 **d_result, u32 &counter_idx)
{
	switch (words) {
	case 1: return sort_device_t<1>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 2: return sort_device_t<2>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 3: return sort_device_t<3>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 4: return sort_device_t<4>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 5: return sort_device_t<5>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 6: return sort_device_t<6>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 7: return sort_device_t<7>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
	case 8: return sort_device_t<8>(s, d_recs, d_tmp, n, n_pass, d_result, counter_idx);
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

/* ---- front end: mark super-k-mer starts (per pack), then expand slice-parallel with the sort's histograms fused in ---- */
template <int SIZE>
int front_end(Slot &s, const DevParams &P, const uint8_t *d_in, u64 size, u64 n_rec, const u64 *d_pack_start, u64 n_packs, u32 n_pass,
              u32 &counter_idx, bool &hist_done)
{
	u32 *err = small_ptr<u32>(s, SM_ERR);
	u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
	const u64 bm_words = (size + 31) / 32 + 2;
	const u64 n_chunks = (size + EXP_CHUNK - 1) / EXP_CHUNK;
	if (n_chunks > 0x7FFFFFF0ull || n_packs > 0x7FFFFFF0ull)
		return fail(KMC_HIP_EINVAL, "bin too large");
	int rc = 0;
	if ((rc = ensure(s.bitmap, bm_words * 4)) || (rc = ensure(s.status, n_chunks * 8)) || (rc = ensure(s.ghist, (size_t)n_pass * 256 * 8)))
		return rc;
	HIPCHK(hipMemsetAsync(s.bitmap.p, 0, bm_words * 4, s.stream));
	HIPCHK(hipMemsetAsync(s.status.p, 0, n_chunks * 8, s.stream));
	k_parse_packs<<<dim3((u32)n_packs), dim3(256), 0, s.stream>>>(d_in, d_pack_start, (u32)n_packs, P.k, (u32 *)s.bitmap.p, err);
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[1], s.stream));
	const bool fuse = n_pass <= 16 && n_rec >= 2; /* LDS: 1 KB of counters per pass next to the 33 KB of slice state */
	if (fuse)
		HIPCHK(hipMemsetAsync(s.ghist.p, 0, (size_t)n_pass * 256 * 8, s.stream));
	const u32 blocks = (u32)std::min<u64>(n_chunks, 256 * 2 * (1024 / EXP_BLOCK));
	if (fuse)
		k_expand<SIZE, true><<<dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<true>(n_pass), s.stream>>>(
		    d_in, size, (const u32 *)s.bitmap.p, P.k, P.both_strands, n_pass, n_rec, (u64 *)s.recA.p, (u64 *)s.ghist.p, (u64 *)s.status.p,
		    counters + counter_idx, (u32)n_chunks, err);
	else
		k_expand<SIZE, false><<<dim3(blocks), dim3(EXP_BLOCK), exp_lds_bytes<false>(n_pass), s.stream>>>(
		    d_in, size, (const u32 *)s.bitmap.p, P.k, P.both_strands, n_pass, n_rec, (u64 *)s.recA.p, (u64 *)s.ghist.p, (u64 *)s.status.p,
		    counters + counter_idx, (u32)n_chunks, err);
	++counter_idx;
	hist_done = fuse;
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[2], s.stream));
	HIPCHK(hipGetLastError());
	return 0;
}

/* ---- compaction launch (+ tally / LUT shard reductions) ---- */
template <int SIZE>
int launch_compact(Slot &s, const u64 *sorted, u64 n, const DevParams &P, uint8_t *d_out, u64 out_capacity, u64 *d_lut, u64 lut_entries,
                   u64 *d_stats, u64 *d_out_bytes, u32 &counter_idx)
{
	u32 *err = small_ptr<u32>(s, SM_ERR);
	u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
	const u64 c_tiles = (n + CpCfg<SIZE>::TILE - 1) / CpCfg<SIZE>::TILE;
	if (c_tiles > 0x7FFFFFFFull)
		return fail(KMC_HIP_EINVAL, "bin too large");
	if (counter_idx >= N_COUNTERS)
		return fail(KMC_HIP_EINVAL, "too many launches for one bin");
	int rc = 0;
	if ((rc = ensure(s.status, c_tiles * 8)))
		return rc;
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = !use_lut ? 1u : (lut_entries <= 1024 ? 32u : (lut_entries <= 16384 ? 4u : 1u));
	u64 *lut_base = d_lut;
	if (use_lut && n_sh > 1) {
		if ((rc = ensure(s.lutsh, (size_t)n_sh * lut_entries * 8)))
			return rc;
		HIPCHK(hipMemsetAsync(s.lutsh.p, 0, (size_t)n_sh * lut_entries * 8, s.stream));
		lut_base = (u64 *)s.lutsh.p;
	}
	HIPCHK(hipMemsetAsync(s.status.p, 0, c_tiles * 8, s.stream));
	k_compact<SIZE><<<dim3((u32)((c_tiles + CP_TPB - 1) / CP_TPB)), dim3(CP_BLOCK), 0, s.stream>>>(
	    sorted, n, P, d_out, out_capacity, lut_base, n_sh, lut_entries, small_ptr<u64>(s, SM_SHARDS), d_out_bytes, (u64 *)s.status.p,
	    counters + counter_idx, (u32)c_tiles, err);
	++counter_idx;
	k_stats_reduce<<<dim3(1), dim3(64), 0, s.stream>>>(small_ptr<u64>(s, SM_SHARDS), d_stats, n);
	if (use_lut && n_sh > 1)
		k_lut_reduce<<<dim3((u32)((lut_entries + 255) / 256)), dim3(256), 0, s.stream>>>((const u64 *)s.lutsh.p, n_sh, lut_entries, d_lut);
	HIPCHK(hipGetLastError());
	return 0;
}

/* ---- one bin, everything device resident --------------------------------------------------------------------- */
template <int SIZE>
int run_bin_device_t(Slot &s, const DevParams &P, const uint8_t *d_in, u64 size, u64 n_rec, const u64 *d_pack_start,
                     u64 n_packs, uint8_t *d_out, u64 out_capacity, u64 *d_out_bytes, u64 *d_lut, u64 lut_entries, u64 *d_stats)
{
	const u32 k = P.k;
	const u32 n_pass = (2 * k + 7) / 8; /* = ceil(k/4) = rec_len of the plain k-mer path (kb_sorter.h:769) */
	u32 *err = small_ptr<u32>(s, SM_ERR);
	u64 *totals = small_ptr<u64>(s, SM_TOTALS);
	u32 *counters = small_ptr<u32>(s, SM_COUNTERS);
	u32 counter_idx = 0;
	s.sc_used = 0;
	s.sc_keys = 0;

	HIPCHK(hipMemsetAsync(s.small.p, 0, SM_BYTES, s.stream));
	HIPCHK(hipMemsetAsync(d_stats, 0, 4 * 8, s.stream));
	HIPCHK(hipMemsetAsync(d_out_bytes, 0, 8, s.stream));
	if (lut_entries && !P.without_output)
		HIPCHK(hipMemsetAsync(d_lut, 0, lut_entries * 8, s.stream));
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[0], s.stream));
	if (n_rec == 0 || size == 0) {
		if (s.timed)
			for (int i = 1; i < 6; ++i)
				HIPCHK(hipEventRecord(s.ev[i], s.stream));
		return 0;
	}
	if (n_packs == 0 || n_packs > 0xFFFFFFF0ull)
		return fail(KMC_HIP_EINVAL, "n_packs out of range");

	int rc = 0;
	if ((rc = ensure(s.recA, n_rec * SIZE * 8 + 256)) || (rc = ensure(s.recB, n_rec * SIZE * 8 + 256)))
		return rc;
	bool hist_done = false;
	if ((rc = front_end<SIZE>(s, P, d_in, size, n_rec, d_pack_start, n_packs, n_pass, counter_idx, hist_done)))
		return rc;
	/* sort */
	u64 *sorted = nullptr;
	if ((rc = sort_device_t<SIZE>(s, (u64 *)s.recA.p, (u64 *)s.recB.p, n_rec, n_pass, &sorted, counter_idx, hist_done)))
		return rc;
	if (s.timed) {
		if (n_rec < 2)
			HIPCHK(hipEventRecord(s.ev[3], s.stream));
		HIPCHK(hipEventRecord(s.ev[4], s.stream));
	}
	/* compact */
	if ((rc = launch_compact<SIZE>(s, sorted, n_rec, P, d_out, out_capacity, d_lut, lut_entries, d_stats, d_out_bytes, counter_idx)))
		return rc;
	if (s.timed)
		HIPCHK(hipEventRecord(s.ev[5], s.stream));
	HIPCHK(hipGetLastError());
	return 0;
}

int run_bin_device(Slot &s, const DevParams &P, const uint8_t *d_in, u64 size, u64 n_rec, const u64 *d_pack_start, u64 n_packs,
                   uint8_t *d_out, u64 out_capacity, u64 *d_out_bytes, u64 *d_lut, u64 lut_entries, u64 *d_stats)
{
#define RUN(N) return run_bin_device_t<N>(s, P, d_in, size, n_rec, d_pack_start, n_packs, d_out, out_capacity, d_out_bytes, d_lut, lut_entries, d_stats)
	switch ((P.k + 31) / 32) {
	case 1: RUN(1);
	case 2: RUN(2);
	case 3: RUN(3);
	case 4: RUN(4);
	case 5: RUN(5);
	case 6: RUN(6);
	case 7: RUN(7);
	case 8: RUN(8);
	}
#undef RUN
	return fail(KMC_HIP_EINVAL, "kmer_len out of range");
}

int err_to_code(u32 err)
{
	if (err & KERR_WATCHDOG)
		return fail(KMC_HIP_EINTERNAL, "device look-back watchdog tripped");
	if (err & KERR_CORRUPT)
		return fail(KMC_HIP_ECORRUPT, "super-k-mer stream does not end on a pack boundary");
	if (err & KERR_NREC)
		return fail(KMC_HIP_ECORRUPT, "n_rec disagrees with the super-k-mer stream");
	if (err & KERR_CAPACITY)
		return fail(KMC_HIP_ECAPACITY, "out_capacity too small for the counted k-mers");
	return 0;
}

} // namespace

/* ---- stage-isolating test hooks (tests/ use them to localise a parity failure to one kernel group) ---- */
namespace {
template <int SIZE>
int debug_expand_t(Slot &s, const DevParams &P, u64 size, u64 n_rec, u64 np)
{
	int rc = 0;
	if ((rc = ensure(s.recA, n_rec * SIZE * 8 + 256)))
		return rc;
	HIPCHK(hipMemsetAsync(s.small.p, 0, SM_BYTES, s.stream));
	u32 counter_idx = 0;
	bool hist_done = false;
	return front_end<SIZE>(s, P, (const uint8_t *)s.in.p, size, n_rec, (const u64 *)s.pack_start.p, np, (2 * P.k + 7) / 8, counter_idx, hist_done);
}
template <int SIZE>
int debug_compact_t(Slot &s, const DevParams &P, u64 n, u64 out_capacity, u64 lut_entries)
{
	HIPCHK(hipMemsetAsync(s.small.p, 0, SM_BYTES, s.stream));
	if (lut_entries)
		HIPCHK(hipMemsetAsync(s.lut.p, 0, lut_entries * 8, s.stream));
	u32 counter_idx = 0;
	return launch_compact<SIZE>(s, (const u64 *)s.recA.p, n, P, (uint8_t *)s.out.p, out_capacity, (u64 *)s.lut.p, lut_entries,
	                            small_ptr<u64>(s, SM_STATS), small_ptr<u64>(s, SM_OUTBYTES), counter_idx);
}
} // namespace


/* ================================================================================================ C-ABI */

extern "C" {

int kmc_hip_abi_version(void) { return KMC_HIP_ABI_VERSION; }
const char *kmc_hip_last_error(kmc_hip_ctx *) { return g_err.c_str(); }
uint32_t kmc_hip_words(uint32_t kmer_len) { return (kmer_len + 31) / 32; }
uint32_t kmc_hip_counter_size(uint64_t cutoff_max, uint64_t counter_max) { return counter_bytes(cutoff_max, counter_max); }
uint32_t kmc_hip_out_rec_bytes(const kmc_hip_bin_params *p)
{
	return kmc_suffix_bytes(p->kmer_len, p->lut_prefix_len) + counter_bytes(p->cutoff_max, p->counter_max);
}
uint64_t kmc_hip_lut_entries(const kmc_hip_bin_params *p) { return p->lut_prefix_len ? 1ull << (2 * p->lut_prefix_len) : 0; }

int kmc_hip_init(const int *device_ids, int n_dev, kmc_hip_ctx **out)
{
	if (!out || n_dev < 1)
		return fail(KMC_HIP_EINVAL, "kmc_hip_init: bad arguments");
	int count = 0;
	HIPCHK(hipGetDeviceCount(&count));
	if (count < 1)
		return fail(KMC_HIP_EDEVICE, "no HIP device visible");
	PORTION = PORTION_MAX;
	if (const char *e = getenv("KMC_HIP_DEBUG_PORTION_LOG2")) {
		const int lg = atoi(e);
		if (lg >= 10 && lg <= 29)
			PORTION = 1ull << lg;
	}
	kmc_hip_ctx *ctx = new kmc_hip_ctx();
	ctx->devs.resize(n_dev);
	for (int i = 0; i < n_dev; ++i) {
		const int ord = device_ids ? device_ids[i] : i;
		if (ord < 0 || ord >= count) {
			delete ctx;
			return fail(KMC_HIP_EINVAL, "device ordinal out of range");
		}
		ctx->devs[i].ordinal = ord;
		hipError_t e = hipSetDevice(ord);
		if (e != hipSuccess) {
			delete ctx;
			return fail_hip("hipSetDevice", e);
		}
		for (auto &s : ctx->devs[i].slot)
			if (int rc = slot_init(s)) {
				kmc_hip_destroy(ctx);
				return rc;
			}
	}
	*out = ctx;
	return 0;
}

void kmc_hip_destroy(kmc_hip_ctx *ctx)
{
	if (!ctx)
		return;
	for (auto &d : ctx->devs) {
		(void)hipSetDevice(d.ordinal);
		(void)hipDeviceSynchronize();
		for (auto &s : d.slot)
			slot_destroy(s);
		if (d.rccl_buf.p)
			(void)hipFree(d.rccl_buf.p);
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
int kmc_hip_synchronize(kmc_hip_ctx *ctx, int dev)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	for (auto &s : ctx->devs[dev].slot)
		HIPCHK(hipStreamSynchronize(s.stream));
	u32 err = 0, e1 = 0;
	for (auto &s : ctx->devs[dev].slot) {
		HIPCHK(hipMemcpy(&e1, small_ptr<u32>(s, SM_ERR), 4, hipMemcpyDeviceToHost));
		err |= e1;
	}
	return err_to_code(err);
}

/* ---- narrow boundary ---- */
int kmc_hip_sort_records_device(kmc_hip_ctx *ctx, int dev, void *d_recs, void *d_tmp, uint64_t n, uint32_t words, uint32_t key_bytes,
                                void **d_result)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (words < 1 || words > 8 || key_bytes > 8 * words || !d_result)
		return fail(KMC_HIP_EINVAL, "kmc_hip_sort_records_device: bad arguments");
	Slot &s = ctx->devs[dev].slot[0];
	s.timed = true;
	s.sc_used = 0;
	s.sc_keys = 0;
	HIPCHK(hipMemsetAsync(s.small.p, 0, SM_BYTES, s.stream));
	u32 counter_idx = 0;
	u64 *res = nullptr;
	if (int rc = sort_device(s, (u64 *)d_recs, (u64 *)d_tmp, n, words, key_bytes, &res, counter_idx))
		return rc;
	HIPCHK(hipStreamSynchronize(s.stream));
	u32 err = 0;
	HIPCHK(hipMemcpy(&err, small_ptr<u32>(s, SM_ERR), 4, hipMemcpyDeviceToHost));
	*d_result = res;
	return err_to_code(err);
}

int kmc_hip_sort_records(kmc_hip_ctx *ctx, int dev, void *recs, uint64_t n, uint32_t words, uint32_t key_bytes)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	if (words < 1 || words > 8 || key_bytes > 8 * words)
		return fail(KMC_HIP_EINVAL, "kmc_hip_sort_records: bad arguments");
	if (n < 2)
		return 0;
	if (!recs)
		return fail(KMC_HIP_EINVAL, "recs == NULL");
	Slot &s = ctx->devs[dev].slot[0];
	const size_t bytes = (size_t)n * words * 8;
	int rc = 0;
	if ((rc = ensure(s.recA, bytes + 256)) || (rc = ensure(s.recB, bytes + 256)))
		return rc;
	HIPCHK(hipMemcpy(s.recA.p, recs, bytes, hipMemcpyHostToDevice));
	void *res = nullptr;
	if ((rc = kmc_hip_sort_records_device(ctx, dev, s.recA.p, s.recB.p, n, words, key_bytes, &res)))
		return rc;
	HIPCHK(hipMemcpy(recs, res, bytes, hipMemcpyDeviceToHost));
	return 0;
}

/* ---- full boundary ---- */
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
	 * (small) bin are filled by the kernels of the next ones; a synchronous call always uses slot 0, and so does a bin
	 * whose record arrays exceed ASYNC_BIG_BYTES: it fills the GPU on its own, and every slot it visited would keep
	 * two arrays of that size (slot buffers only grow) */
	constexpr u64 ASYNC_BIG_BYTES = 1ull << 31;
	const bool big = n_rec * (u64)kmc_hip_words(P.k) * 8 * 2 > ASYNC_BIG_BYTES;
	Slot &s = ctx->devs[dev].slot[(sync || big) ? 0 : (ctx->devs[dev].rr++ % N_SLOTS)];
	s.timed = true;
	const u64 lut_entries = kmc_hip_lut_entries(params);
	if (int rc = run_bin_device(s, P, d_superkmers, size, n_rec, (const u64 *)d_pack_start, n_packs, d_out, out_capacity,
	                            (u64 *)d_out_bytes, (u64 *)d_lut, P.kff ? 0 : lut_entries, (u64 *)d_stats))
		return rc;
	if (!sync)
		return 0;
	HIPCHK(hipStreamSynchronize(s.stream));
	u32 err = 0;
	HIPCHK(hipMemcpy(&err, small_ptr<u32>(s, SM_ERR), 4, hipMemcpyDeviceToHost));
	return err_to_code(err);
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
	Slot &s = ctx->devs[dev].slot[slot];
	if (s.pending)
		return fail(KMC_HIP_EINVAL, "slot already has a bin in flight");
	if (size && !superkmers)
		return fail(KMC_HIP_EINVAL, "superkmers == NULL");
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	if (!P.without_output && ((out_capacity && !out_suffix) || (lut_entries && !lut)))
		return fail(KMC_HIP_EINVAL, "output buffers missing");

	/* pack starts (byte offsets). Without packs from the caller, walk the image once on the host. */
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
	if (size) {
		HIPCHK(hipMemcpyAsync(s.in.p, superkmers, size, hipMemcpyHostToDevice, s.stream));
		HIPCHK(hipMemsetAsync((char *)s.in.p + size, 0, 256, s.stream));
		HIPCHK(hipMemcpyAsync(s.pack_start.p, ps.data(), (np + 1) * 8, hipMemcpyHostToDevice, s.stream));
	}
	s.timed = true;
	if ((rc = run_bin_device(s, P, (const uint8_t *)s.in.p, size, n_rec, (const u64 *)s.pack_start.p, np, (uint8_t *)s.out.p,
	                         P.without_output ? 0 : out_capacity, small_ptr<u64>(s, SM_OUTBYTES), (u64 *)s.lut.p, lut_entries,
	                         small_ptr<u64>(s, SM_STATS))))
		return rc;
	HIPCHK(hipMemcpyAsync(s.h_res, s.small.p, sizeof(HostRes), hipMemcpyDeviceToHost, s.stream));
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
	Slot &s = ctx->devs[dev].slot[slot];
	if (!s.pending)
		return fail(KMC_HIP_EINVAL, "no bin in flight on this slot");
	s.pending = false;
	HIPCHK(hipStreamSynchronize(s.stream));
	const HostRes r = *s.h_res;
	if (int rc = err_to_code(r.err))
		return rc;
	if (r.out_bytes > s.out_capacity)
		return fail(KMC_HIP_ECAPACITY, "out_capacity too small for the counted k-mers");
	if (!s.without_output) {
		if (r.out_bytes)
			HIPCHK(hipMemcpy(s.h_out, s.out.p, r.out_bytes, hipMemcpyDeviceToHost));
		if (s.lut_entries)
			HIPCHK(hipMemcpy(s.h_lut, s.lut.p, s.lut_entries * 8, hipMemcpyDeviceToHost));
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
	Slot &s = ctx->devs[dev].slot[0];
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
	HIPCHK(hipMemcpy(&err, small_ptr<u32>(s, SM_ERR), 4, hipMemcpyDeviceToHost));
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
	Slot &s = ctx->devs[dev].slot[0];
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
	HIPCHK(hipMemcpy(&r, s.small.p, sizeof r, hipMemcpyDeviceToHost));
	if ((rc = err_to_code(r.err)))
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
			ords[i] = ctx->devs[i].ordinal;
		ctx->comms.resize(n);
		ncclResult_t r = ncclCommInitAll(ctx->comms.data(), n, ords.data());
		if (r != ncclSuccess)
			return fail(KMC_HIP_EDEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
		ctx->comms_ready = true;
	}
	for (int i = 0; i < n; ++i) {
		if (int rc = set_dev(ctx, i))
			return rc;
		if (int rc = ensure(ctx->devs[i].rccl_buf, 64))
			return rc;
		HIPCHK(hipMemcpy(ctx->devs[i].rccl_buf.p, per_dev_stats + 4 * i, 32, hipMemcpyHostToDevice));
	}
	ncclResult_t r = ncclGroupStart();
	for (int i = 0; i < n && r == ncclSuccess; ++i) {
		(void)hipSetDevice(ctx->devs[i].ordinal);
		r = ncclAllReduce(ctx->devs[i].rccl_buf.p, ctx->devs[i].rccl_buf.p, 4, ncclUint64, ncclSum, ctx->comms[i], ctx->devs[i].slot[0].stream);
	}
	ncclResult_t r2 = ncclGroupEnd();
	if (r != ncclSuccess || r2 != ncclSuccess)
		return fail(KMC_HIP_EDEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
	for (int i = 0; i < n; ++i) {
		if (int rc = set_dev(ctx, i))
			return rc;
		HIPCHK(hipStreamSynchronize(ctx->devs[i].slot[0].stream));
		HIPCHK(hipMemcpy(per_dev_stats + 4 * i, ctx->devs[i].rccl_buf.p, 32, hipMemcpyDeviceToHost));
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
	Slot &s = ctx->devs[dev].slot[0];
	HIPCHK(hipStreamSynchronize(s.stream));
	for (int i = 0; i < 5; ++i)
		HIPCHK(hipEventElapsedTime(&ms[i], s.ev[i], s.ev[i + 1]));
	HIPCHK(hipEventElapsedTime(&ms[5], s.ev[0], s.ev[5]));
	return 0;
}

int kmc_hip_last_scatter_stats(kmc_hip_ctx *ctx, int dev, uint32_t *n_launches, float *total_ms, uint64_t *keys_per_launch)
{
	if (int rc = set_dev(ctx, dev))
		return rc;
	Slot &s = ctx->devs[dev].slot[0];
	HIPCHK(hipStreamSynchronize(s.stream));
	float tot = 0;
	for (u32 i = 0; i + 1 < s.sc_used; i += 2) {
		float t = 0;
		HIPCHK(hipEventElapsedTime(&t, s.sc_ev[i], s.sc_ev[i + 1]));
		tot += t;
	}
	if (n_launches)
		*n_launches = s.sc_used / 2;
	if (total_ms)
		*total_ms = tot;
	if (keys_per_launch)
		*keys_per_launch = s.sc_used >= 2 ? s.sc_keys / (s.sc_used / 2) : 0; /* average records per launch */
	return 0;
}

} /* extern "C" */
