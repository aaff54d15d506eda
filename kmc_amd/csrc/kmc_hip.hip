/*
 * kmc_amd/csrc/kmc_hip.hip — host side of libkmc_hip.so: the C-ABI of include/kmc_hip.h over the gfx950
 * kernels in kernels.hip.h. One context owns, per device, N_SLOTS "slots" (stream + grow-only HBM buffers), so
 * callers can keep several bins in flight (H2D of one bin under the kernels of another; small bins fill each other's
 * launch gaps).
 *
 * Reference mapping: this file plays the role of CKmerBinSorter<SIZE>::ProcessBins' body
 * (kmc_core/kb_sorter.h:210-237): Expand -> Sort -> Compact for one bin, but as a queue of kernels on a
 * HIP stream. No CPU fallback exists: if HIP is unusable every entry point fails with KMC_HIP_EDEVICE.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kmc_hip.h"
#include "kernels.hip.h"
#include "bucket_sort.hip.h"
#include "order_db.hip.h"
#include "stage1_kernels.hip.h"

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

/* is `p` host memory the device can reach by DMA as it is (hipHostMalloc / hipHostRegister)? Ordinary memory is staged through a pinned buffer of the slot. */
bool host_ptr_is_pinned(const void *p)
{
	static const bool always = [] {
		const char *e = getenv("KMC_HIP_STAGE_PAGEABLE"); /* 0: copy straight from / to pageable memory (rounds 1-3) */
		return e && atoi(e) == 0;
	}();
	if (always || !p)
		return true;
	hipPointerAttribute_t at;
	if (hipPointerGetAttributes(&at, p) != hipSuccess) {
		(void)hipGetLastError(); /* "invalid value" for memory the runtime has never seen: ordinary memory */
		return false;
	}
	return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}
int ensure_pinned(void *&p, size_t &cap, size_t bytes)
{
	if (bytes <= cap)
		return 0;
	if (p) {
		HIPCHK(hipHostFree(p));
		p = nullptr;
		cap = 0;
	}
	const size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095; /* grow-only, with room: bins come in descending size order at first, then vary */
	HIPCHK(hipHostMalloc(&p, want, hipHostMallocDefault));
	cap = want;
	return 0;
}

/* layout of the per-slot "small" device block (bytes): the first SM_BYTES of the slot's zero region, cleared with it at the start
 * of every bin */
/* bytes 0..15: unused */
constexpr size_t SM_STATS = 16;      /* u64[4]                           */
constexpr size_t SM_OUTBYTES = 48;   /* u64                              */
constexpr size_t SM_ERR = 56;        /* u32: copy of the slot's sticky error word, taken when a host-boundary bin ends */
constexpr size_t SM_REDO = 60;       /* u32: set by k_bucket_sort when a tile of the hybrid sort did not fit: the host sorts the group again, LSD over all bytes */
constexpr size_t SM_DBASE_WORK = 256; /* u64[2][256] per-portion digit bases (ping-pong) */
constexpr size_t SM_COUNTERS = 256 + 2 * 256 * 8; /* u32[N_COUNTERS] ticket counters, one per launch (the tally shards of the compaction are per bin, BinPlan) */
constexpr size_t N_COUNTERS = 4096;
constexpr size_t SM_BYTES = SM_COUNTERS + N_COUNTERS * 4;

#ifndef KMC_N_SLOTS
#define KMC_N_SLOTS 16 /* 512 bins of 3.2 M k-mers, device resident: 1 stream 237 ms, 2: 133, 4: 104, 8: 95 (then the host launch rate
                        * binds); the stage-2 worker spreads up to 16 sorter threads over them */
#endif
constexpr int N_SLOTS = KMC_N_SLOTS;
constexpr int N_BATCH_STREAMS = 8; /* default fan-out of kmc_hip_process_bins_device for small bins */
constexpr u32 TIMING_SAMPLE = 8;   /* an event pair costs a few microseconds of stream time: asynchronous bins are timed 1 in 8 */
constexpr u64 PORTION_MAX = 1ull << 29; /* records per scatter launch (30-bit look-back counts) */
/* Tests shrink the portion ($KMC_HIP_DEBUG_PORTION_LOG2, 10..29, read at kmc_hip_init, kept per context) so that a small,
 * oracle-checkable sort crosses many portion boundaries (digit bases carried from launch to launch) — the path a bin of
 * more than 2^29 k-mers takes. */

struct HostRes {
	u64 totals[2];
	u64 stats[4];
	u64 out_bytes;
	u32 err;
	u32 redo;
};

/* Everything a bin needs zeroed on the device lies in ONE region (one memset per bin instead of ~14: with 512 small bins
 * per run the host launch rate is what binds): start bitmap | expand look-back words | digit histograms | compaction
 * look-back words | one scatter status area per onesweep launch. Offsets are 256-byte aligned. */
struct ZeroPlan {
	size_t ghist = 0, sc_status = 0, sc_stride = 0, total = 0; /* the per-bin parts are in BinPlan */
	size_t giant = 0; /* rank groups: the list of tiles handed to k_giant_tiles (count, taken, tile numbers) */
};
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

/* results of a host-boundary GROUP (kmc_hip_process_bins_submit): per bin out_bytes + the four tallies, per sort group the hybrid sort's "again"
 * word, the stream's sticky error word — one block on the device, copied to pinned memory when the group's kernels are done */
constexpr u32 HB_MAX = 16;
struct HbRes {
	u64 w[HB_MAX][8]; /* [0] out_bytes, [1..4] stats */
	u32 flag[HB_MAX];
	u32 err, pad[15];
};

struct Slot {
	hipStream_t stream = nullptr;
	std::mutex mtx; /* serialises enqueueing on this slot (asynchronous device-resident calls may come from several threads) */
	u64 portion = PORTION_MAX;
	DBuf in, pack_start;
	DBuf recA, recB, recC, pairA, pairB, zero, dbase, out, lut, sticky;
	DBuf bounds;   /* hybrid sort: tile boundaries of k_bucket_bounds, u64[windows + 1] */
	DBuf redo_log; /* hybrid sort: one "sort me again" word per asynchronous group since the last drain (drain_redo) */
	HostRes *h_res = nullptr; /* pinned */
	/* pinned staging for callers whose buffers are ordinary (pageable) memory — the drop-in worker's arena: a copy straight from / to such memory makes the
	 * runtime pin the caller's pages on the fly, and the unmapping of an arena that has been pinned piecewise costs twice as much at the end of KMC's stage 2
	 * (tools/ubench_munmap_hip.py: 0.23 s instead of 0.12 s for 2.3 GB; nothing left after a parallel MADV_DONTNEED only when nothing was ever pinned) */
	void *h_stage_in = nullptr, *h_stage_out = nullptr;
	size_t h_stage_in_cap = 0, h_stage_out_cap = 0;
	bool out_staged = false;
	u64 groups_run = 0;     /* groups enqueued on this slot since the context was made */
	bool zero_grew = false; /* the last group's zero region had to be re-allocated (diagnostics: reported with a watchdog error) */
	hipEvent_t ev[6] = {};
	hipEvent_t done_ev = nullptr; /* blocking-sync event: _wait must not spin (stage-2 workers outnumber the cores a container may use) */
	/* one event pair per scatter launch since the last harvest (roofline input) */
	std::vector<hipEvent_t> sc_ev;
	std::vector<u64> sc_cnt; /* records of the launch; bit 63: the pair brackets the LDS sort (k_bucket_bounds + k_bucket_sort), not a scatter pass */
	u32 sc_used = 0;
	double sc_ms_total = 0, ls_ms_total = 0;
	u64 sc_keys_total = 0, sc_launch_total = 0, ls_keys_total = 0, ls_launch_total = 0;
	bool timed = false;
	u32 async_seq = 0; /* asynchronous device-resident bins on this slot: every TIMING_SAMPLE-th one carries events */
	/* pending async bin */
	bool pending = false;
	uint8_t *h_out = nullptr;
	u64 *h_lut = nullptr;
	u64 out_capacity = 0, lut_entries = 0;
	bool without_output = false;
	std::vector<u64> h_pack_start;
	/* host-boundary bin in flight: what a redo needs */
	DevParams sub_P = {};
	u64 sub_size = 0, sub_n_rec = 0, sub_np = 0;
	/* asynchronous device-resident groups since the last drain, in redo_log order */
	struct PendingGroup {
		DevParams P;
		u64 lut_entries;
		std::vector<kmc_hip_bin_desc> descs;
	};
	std::vector<PendingGroup> pending_groups;
	/* host-boundary group in flight */
	struct HostBin {
		kmc_hip_bin_desc d; /* device side */
		uint8_t *h_out;
		u64 *h_lut;
	};
	std::vector<HostBin> hb;
	std::vector<std::pair<u32, u32>> hb_chunks; /* (first bin, bins) of every sort group */
	std::vector<char> hb_hybrid;
	DBuf hb_res;
	HbRes *h_hb_res = nullptr; /* pinned */
	DevParams hb_P = {};
	bool hb_pending = false;
};

struct Dev {
	int ordinal = 0;
	u32 rr = 0; /* round-robin slot choice of asynchronous device-resident calls (under rr_mtx) */
	std::mutex rr_mtx;
	Slot slot[N_SLOTS];
	DBuf rccl_buf;
	DBuf xchg; /* kmc_hip_process_bin_multi: the records this device receives from the others */
	int *d_sig_map = nullptr; /* stage 1: the signature -> bin map of kmc_hip_split_set_map */
	u32 sig_map_entries = 0;
	std::mutex map_mtx;
	DBuf s1_arena[N_SLOTS]; /* stage 1: grow-only work area of kmc_hip_split_part, one per stream slot */
};

u32 counter_bytes(u64 cutoff_max, u64 counter_max) { return kmc_counter_bytes(cutoff_max, counter_max); }

} // namespace

struct kmc_hip_ctx {
	std::vector<std::unique_ptr<Dev>> devs;
	std::vector<ncclComm_t> comms;
	bool comms_ready = false;
	u64 portion = PORTION_MAX;
	std::mutex mtx;
};

namespace {

int set_dev(kmc_hip_ctx *ctx, int dev)
{
	if (!ctx || dev < 0 || dev >= (int)ctx->devs.size())
		return fail(KMC_HIP_EINVAL, "bad ctx/dev");
	HIPCHK(hipSetDevice(ctx->devs[dev]->ordinal));
	return 0;
}

/* Kernels that want more than 64 KiB of dynamic LDS must opt in, per device (the attribute belongs to the device's copy of
 * the function): k_onesweep<SIZE> for SIZE >= 6 (k > 160), the histogram-fusing k_expand at 16 passes (k = 61..64). */
template <int SIZE> int set_func_attrs()
{
	if (rs_lds_bytes<SIZE>() > 65536)
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_onesweep<SIZE>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                           (int)rs_lds_bytes<SIZE>()));
	if (bc_lds_bytes<SIZE>() > 65536)
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bucket_count<SIZE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bc_lds_bytes<SIZE>()));
	if (bs_lds_bytes<SIZE>() > 65536)
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bucket_sort<SIZE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bs_lds_bytes<SIZE>()));
	if (br_lds_bytes<SIZE>() > 65536) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bucket_rank<SIZE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes<SIZE>()));
		if (SIZE == 1)
			HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bucket_rank<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes<1>()));
	}
	if (exp_lds_bytes<true>(EXP_FUSE_MAX_PASS, 1) > 65536) /* worst case: the shortest records (smallest k) and 16 fused passes */
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_expand<SIZE, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                           (int)exp_lds_bytes<true>(EXP_FUSE_MAX_PASS, 1)));
	return 0;
}
int set_all_func_attrs()
{
	int rc = 0;
	(void)((rc = set_func_attrs<1>()) || (rc = set_func_attrs<2>()) || (rc = set_func_attrs<3>()) || (rc = set_func_attrs<4>()) ||
	       (rc = set_func_attrs<5>()) || (rc = set_func_attrs<6>()) || (rc = set_func_attrs<7>()) || (rc = set_func_attrs<8>()));
	return rc;
}

int slot_init(Slot &s, u64 portion)
{
	s.portion = portion;
	HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
	HIPCHK(hipHostMalloc((void **)&s.h_res, sizeof(HostRes), hipHostMallocDefault));
	memset(s.h_res, 0, sizeof(HostRes));
	for (auto &e : s.ev)
		HIPCHK(hipEventCreate(&e));
	HIPCHK(hipEventCreateWithFlags(&s.done_ev, hipEventBlockingSync | hipEventDisableTiming));
	if (int rc = ensure(s.zero, SM_BYTES))
		return rc;
	HIPCHK(hipMemset(s.zero.p, 0, SM_BYTES));
	if (int rc = ensure(s.sticky, 256))
		return rc;
	HIPCHK(hipMemset(s.sticky.p, 0, 256));
	return 0;
}

void slot_destroy(Slot &s)
{
	for (DBuf *b : {&s.in, &s.pack_start, &s.recA, &s.recB, &s.recC, &s.pairA, &s.pairB, &s.zero, &s.dbase, &s.out, &s.lut, &s.sticky, &s.bounds, &s.redo_log, &s.hb_res})
		if (b->p)
			(void)hipFree(b->p);
	if (s.h_res)
		(void)hipHostFree(s.h_res);
	if (s.h_hb_res)
		(void)hipHostFree(s.h_hb_res);
	if (s.h_stage_in)
		(void)hipHostFree(s.h_stage_in);
	if (s.h_stage_out)
		(void)hipHostFree(s.h_stage_out);
	for (auto &e : s.ev)
		if (e)
			(void)hipEventDestroy(e);
	for (auto &e : s.sc_ev)
		(void)hipEventDestroy(e);
	if (s.done_ev)
		(void)hipEventDestroy(s.done_ev);
	if (s.stream)
		(void)hipStreamDestroy(s.stream);
}

template <typename T> T *small_ptr(Slot &s, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(s.zero.p) + off); }
template <typename T> T *zero_ptr(Slot &s, size_t off) { return reinterpret_cast<T *>(static_cast<char *>(s.zero.p) + off); }
u32 *err_ptr(Slot &s) { return static_cast<u32 *>(s.sticky.p); }

int sc_event_pair(Slot &s, hipEvent_t &e0, hipEvent_t &e1, u64 cnt)
{
	while (s.sc_ev.size() < (size_t)s.sc_used + 2) {
		hipEvent_t ne;
		HIPCHK(hipEventCreate(&ne));
		s.sc_ev.push_back(ne);
	}
	e0 = s.sc_ev[s.sc_used];
	e1 = s.sc_ev[s.sc_used + 1];
	s.sc_used += 2;
	s.sc_cnt.push_back(cnt);
	return 0;
}

int ls_event_pair(Slot &s, hipEvent_t &e0, hipEvent_t &e1, u64 cnt) { return sc_event_pair(s, e0, e1, cnt | (1ull << 63)); }

/* after the slot's stream is idle: fold the recorded scatter launches into the slot's totals */
int harvest(Slot &s)
{
	for (u32 i = 0; i + 1 < s.sc_used; i += 2) {
		float t = 0;
		HIPCHK(hipEventElapsedTime(&t, s.sc_ev[i], s.sc_ev[i + 1]));
		const u64 c = s.sc_cnt[i / 2];
		if (c >> 63) {
			s.ls_ms_total += t;
			s.ls_keys_total += c & ~(1ull << 63);
			++s.ls_launch_total;
		} else {
			s.sc_ms_total += t;
			s.sc_keys_total += c;
			++s.sc_launch_total;
		}
	}
	s.sc_used = 0;
	s.sc_cnt.clear();
	return 0;
}

/* the slot's sticky device error word: kernels OR into it, nothing on the per-bin path clears it, so an error raised by an
 * earlier asynchronous bin on this slot survives until somebody looks (kmc_hip_synchronize, a synchronous call, _wait). Behind it, in the same
 * 256-byte block, what the first look-back that timed out saw (kernels.hip.h KERR_DIAG_WORDS): kept per host thread for err_to_code's message. */
thread_local u32 g_diag[KERR_DIAG_WORDS] = {};
thread_local char g_diag_slot[96] = "";
int clear_sticky(Slot &s, u32 err)
{
	if (err & (KERR_WATCHDOG | KERR_PEER)) {
		HIPCHK(hipMemcpy(g_diag, s.sticky.p, sizeof g_diag, hipMemcpyDeviceToHost));
		g_diag[0] = err;
		snprintf(g_diag_slot, sizeof g_diag_slot, "; group %llu of its stream, zero region %s for it", (unsigned long long)s.groups_run, s.zero_grew ? "re-allocated" : "reused");
	} else
		g_diag[0] = 0;
	HIPCHK(hipMemset(s.sticky.p, 0, 4 * KERR_DIAG_WORDS));
	return 0;
}
int read_and_clear_sticky(Slot &s, u32 &err)
{
	HIPCHK(hipMemcpy(&err, s.sticky.p, 4, hipMemcpyDeviceToHost));
	if (err)
		return clear_sticky(s, err);
	return 0;
}

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
 * inside LDS by k_bucket_sort on bucket-aligned tiles (bucket_sort.hip.h). top == key_bytes: the plain LSD sort of rounds 1-2. */
struct SortPlan {
	u32 key_bytes = 0, top = 0;
	u32 key_bits = 0; /* significant bits of the key (2k + tag bits; 8 key_bytes when the caller cannot tell) */
	bool rank = false; /* the LDS half is k_bucket_rank: every tile put in order by pairwise ranking and (fused) counted in place; one-word records whose output
	                    * may outgrow a span: sorted in place, k_compact follows */
	u32 pass_lo() const { return key_bytes - top; }
	bool local() const { return top < key_bytes; }
	u32 hbits() const { return top ? 8 * top - (8 * key_bytes - key_bits) : 0; } /* top bits of the significant key that the HBM passes order (plan_sort: 8 top > spare bits) */
};
std::atomic<u64> g_path[4] = {}; /* groups by the path they took: 0 rank + count in LDS, 1 rank in place + k_compact, 2 k_bucket_count, 3 LSD passes over every byte */
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
		 * counted inside LDS by k_bucket_rank (round 4: every record width; KMC_HIP_RANK=0 / KMC_HIP_RANK_FUSE=0 give round 3's k_bucket_count for k >= 33 and
		 * rank-in-place + k_compact for k <= 32); 2 = k_bucket_count for every record width and the LDS sort for sort-only calls; -h = force `h` top bytes (tuning) */
		const char *e = getenv("KMC_HIP_HYBRID");
		return e ? atoi(e) : 1;
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
SortPlan plan_sort(u64 n, u32 key_bytes, u32 key_bits, bool classic, bool fused = false /* k_bucket_count's tiles, not k_bucket_sort's */,
                   bool rank = false /* the records of a group: k_bucket_rank */)
{
	SortPlan sp;
	sp.key_bytes = sp.top = key_bytes;
	sp.key_bits = key_bits;
	const int mode = hybrid_mode();
	if (classic || mode == 0 || key_bytes < 3 || n < 2)
		return sp;
	if (mode == 1 && (SIZE == 1 || !fused) && !rank)
		return sp;
	const u32 rem_limit = br_rem_limit<SIZE>(); /* rank: key bits that may stay below the bucket bits */
	const u32 spare = 8 * key_bytes - key_bits;
	if (mode < 0) {
		const u32 h = (u32)(-mode);
		if (h + 1 <= key_bytes && 8 * h > spare)
			sp.top = h;
		if (!rank && sp.local() && sp.hbits() > 32)
			sp.top = key_bytes; /* k_bucket_count keeps bucket numbers in 32 bits */
		if (rank && sp.local() && key_bits - sp.hbits() <= rem_limit && sp.hbits() <= 48)
			sp.rank = true;
		else if (rank)
			sp.top = key_bytes;
		return sp;
	}
	const u64 redo = g_redo_groups.load(std::memory_order_relaxed);
	if (g_extra_top.load(std::memory_order_relaxed) >= 2 && redo > 128 && redo * 8 > g_hybrid_groups.load(std::memory_order_relaxed))
		return sp; /* this input defeats the bucket tiles even with two passes more */
	/* the fewest top bytes that leave buckets of bs_target_bucket() records on average, and only if at least two passes are saved */
	for (u32 h = 0; h + 2 <= key_bytes && h <= (rank ? 6u : 4u); ++h) { /* (k_bucket_count keeps bucket numbers in 32 bits, k_bucket_rank in 64) */
		bool ok;
		if (h == 0)
			ok = !fused && n <= (u64)BsCfg<SIZE>::CAP; /* groups always take one pass at least: the bins' tags must be ordered */
		else {
			const u32 eff = 8 * h > spare ? 8 * h - spare : 0;
			/* rank: the k-mers of a signature bin that BEGIN with one of the bin's minimizers share ~18 key bits, whatever the size of the bin — 1/19 of the
			 * records in a few hundred prefixes — and the work grows with the square of a bucket: the passes must reach well below those bits (measured:
			 * 512 bins of 3.2 M k-mers with 22 key bits ordered 17.7 Gk-mers/s, the 7 LSD passes 24.1; 190 M-record groups with 22 bits 7.2, with 30 bits 31.5) */
			if (rank)
				ok = eff >= 28 && (eff >= 63 || (n >> eff) <= 2) && key_bits - eff <= rem_limit;
			else
				ok = eff > 0 && (eff >= 63 || (n >> eff) <= (fused ? bc_target_bucket<SIZE>() : bs_target_bucket<SIZE>()));
		}
		if (ok) {
			sp.top = h;
			break;
		}
	}
	if (sp.top < key_bytes && sp.top >= 1) { /* finer buckets after a redo, while that still saves passes and the bucket number fits 32 bits */
		/* (not for the rank path since round 4: a tile with a bucket beyond LDS goes to k_giant_tiles, and what still comes back is ONE k-mer repeated a million
		 * times — no number of passes splits that; a fifth pass would only cost every later group its time, and records of two words and more their indirect sort) */
		const u32 extra = rank ? 0u : g_extra_top.load(std::memory_order_relaxed);
		sp.top = std::min(std::max(sp.top, std::min(sp.top + extra, rank ? 6u : 4u)), key_bytes);
		if (sp.top + 2 > key_bytes)
			sp.top = key_bytes;
	}
	if (rank) {
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
	}
	return sp;
}

/* lays out the zero region of a group: small block | per bin: bitmap, expand look-back words, compaction look-back words, LUT shards, tally
 * shards | digit histograms | one scatter status area per onesweep launch over the group's `n_total` records */
template <int SIZE>
ZeroPlan plan_group(const Slot &s, std::vector<BinPlan> &bins, u64 n_total, u32 n_pass /* passes through HBM */, bool front, bool sort, bool compact, u64 lut_shard_entries,
                    u64 cp_tile = CpCfg<SIZE>::TILE /* records per compaction tile: k_compact's, or the window of k_bucket_count / k_bucket_rank */,
                    u32 cp_words = 1 /* status words per tile (k_bucket_rank: one per chunk, two chunks) */, u64 giant_entries = 0)
{
	ZeroPlan z;
	size_t off = up256(SM_BYTES);
	if (giant_entries) {
		z.giant = off;
		off += up256((size_t)(giant_entries + 2) * 4);
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
 * bucket-aligned LDS sort of the remaining bytes, in place. `d_flag`: where k_bucket_sort reports a tile it could not sort. ---- */
template <int SIZE>
int sort_device_t(Slot &s, const ZeroPlan &z, u64 *d_recs, u64 *d_tmp, u64 n, const SortPlan &sp, u64 **d_result, u32 &counter_idx, bool hist_done, u32 *d_flag,
                  bool local_by_caller = false /* the caller finishes the low bytes itself (count_group: k_bucket_count) */)
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
				k_onesweep<SIZE><<<dim3((tiles + RS_TPB - 1) / RS_TPB), dim3(RS_BLOCK), rs_lds_bytes<SIZE>(), s.stream>>>(
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
	if (sp.local() && !local_by_caller) {
		const u64 S = SIZE == 1 && sp.rank ? (u64)BrCfg<1>::STRIDE : (u64)BsCfg<SIZE>::STRIDE;
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
		k_bucket_bounds<SIZE><<<dim3((u32)((n_win + 1 + 3) / 4)), dim3(256), 0, s.stream>>>(gbn, (u32)S, sp.key_bits, sp.hbits());
		if constexpr (SIZE == 1) {
			if (sp.rank) { /* the whole array as one "bin": tiles sorted in place (the caller's k_compact follows) */
				GrpRank gr = {};
				gr.g = 1;
				gr.win_prefix[1] = (u32)n_win;
				gr.S[0] = src;
				gr.bounds[0] = bounds;
				k_bucket_rank<1, false><<<dim3((u32)n_win, 2), dim3(BrCfg<1>::THREADS), br_lds_bytes<1>(), s.stream>>>(gr, DevParams{}, sp.key_bits, sp.hbits(), 1u, 0ull, 0u, d_flag);
			} else
				k_bucket_sort<SIZE><<<dim3((u32)n_win), dim3(BsCfg<SIZE>::THREADS), bs_lds_bytes<SIZE>(), s.stream>>>(src, sp.key_bits, sp.hbits(), bounds, d_flag);
		} else
			k_bucket_sort<SIZE><<<dim3((u32)n_win), dim3(BsCfg<SIZE>::THREADS), bs_lds_bytes<SIZE>(), s.stream>>>(src, sp.key_bits, sp.hbits(), bounds, d_flag);
		if (s.timed)
			HIPCHK(hipEventRecord(e1, s.stream));
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

/* ---- hybrid groups: the array is ordered by its top bytes only; k_bucket_count turns bucket-aligned tiles straight into (k-mer, count) records in the
 * tiles' spans of the free record array (kernels: bucket_sort.hip.h), then the fold and the gather of the two-phase output as after k_compact. ---- */
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
template <int SIZE>
int count_group(Slot &s, const std::vector<BinPlan> &bins, const u64 *sorted, u64 *scratch, const DevParams &P, u64 lut_entries, const SortPlan &sp, u64 n_total, u32 *d_flag)
{
	if (bins.empty())
		return 0;
	u32 *err = err_ptr(s);
	const bool use_lut = lut_entries && !P.without_output && !P.kff;
	const u32 n_sh = !use_lut ? 1u : lut_shards_for(lut_entries);
	const u32 rec_bytes = P.sbytes + P.cbytes;
	constexpr u64 S = BcCfg<SIZE>::STRIDE;
	GrpBounds gbn = {};
	GrpBucket gb = {};
	GrpFold gf = {};
	GrpGather gg = {};
	gbn.g = gb.g = gg.g = (u32)bins.size();
	u64 wins = 0, items = 0;
	for (const BinPlan &b : bins)
		items += (b.n_rec + S - 1) / S + 1;
	if (items > 0x7FFFFFF0ull)
		return fail(KMC_HIP_EINVAL, "bin too large");
	if (int rc = ensure(s.bounds, (size_t)(items + 2) * 8))
		return rc;
	u64 *bounds = (u64 *)s.bounds.p;
	items = 0;
	for (size_t i = 0; i < bins.size(); ++i) {
		const BinPlan &b = bins[i];
		const u64 bin_wins = (b.n_rec + S - 1) / S;
		gbn.item_prefix[i] = (u32)items;
		gb.win_prefix[i] = gg.tile_prefix[i] = (u32)wins;
		u64 *lut_base = b.d_lut;
		if (use_lut && n_sh > 1)
			lut_base = zero_ptr<u64>(s, b.off_lutsh);
		else if (lut_entries && !P.kff && b.d_lut) /* also without output: the caller's LUT is zero-filled by the callee (include/kmc_hip.h) */
			HIPCHK(hipMemsetAsync(b.d_lut, 0, lut_entries * 8, s.stream));
		gbn.S[i] = gb.S[i] = sorted + b.rec_off * SIZE;
		gbn.n[i] = gf.n[i] = b.n_rec;
		gbn.bounds[i] = bounds + items;
		gb.bounds[i] = bounds + items;
		gb.scratch[i] = (uint8_t *)(scratch + b.rec_off * SIZE);
		gb.status[i] = zero_ptr<u64>(s, b.off_cp_status);
		gb.lut_base[i] = lut_base;
		gb.tally[i] = zero_ptr<u64>(s, b.off_tally);
		gf.tally[i] = gb.tally[i];
		gf.stats[i] = b.d_stats;
		gf.lut_base[i] = lut_base;
		gf.lut_out[i] = b.d_lut;
		gf.status[i] = gb.status[i];
		gf.n_tiles[i] = (u32)bin_wins;
		gf.out_bytes[i] = b.d_out_bytes;
		gf.out_capacity[i] = b.out_capacity;
		gg.scratch[i] = gb.scratch[i];
		gg.prefix[i] = gb.status[i];
		gg.out[i] = b.d_out;
		gg.out_capacity[i] = b.out_capacity;
		gg.src_rec[i] = bounds + items;
		items += bin_wins + 1;
		wins += bin_wins;
	}
	gbn.item_prefix[bins.size()] = (u32)items;
	gb.win_prefix[bins.size()] = gg.tile_prefix[bins.size()] = (u32)wins;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (s.timed) {
		if (int rc = ls_event_pair(s, e0, e1, n_total))
			return rc;
		HIPCHK(hipEventRecord(e0, s.stream));
	}
	k_bucket_bounds<SIZE><<<dim3((u32)((items + 3) / 4)), dim3(256), 0, s.stream>>>(gbn, (u32)S, sp.key_bits, sp.hbits());
	k_bucket_count<SIZE><<<dim3((u32)wins), dim3(BcCfg<SIZE>::THREADS), bc_lds_bytes<SIZE>(), s.stream>>>(
	    gb, P, sp.key_bits, sp.hbits(), n_sh, lut_entries, P.lut_prefix_len ? (u32)((1ull << (2 * P.lut_prefix_len)) - 1) : 0u, d_flag);
	if (s.timed)
		HIPCHK(hipEventRecord(e1, s.stream));
	k_compact_fold<<<dim3((u32)bins.size()), dim3(256), 0, s.stream>>>(gf, use_lut ? n_sh : 1u, lut_entries, 1u, rec_bytes, err);
	if (!P.without_output)
		k_compact_gather<<<dim3((u32)((wins + 3) / 4)), dim3(256), 0, s.stream>>>(gg, rec_bytes, (u64)SIZE * 8);
	HIPCHK(hipGetLastError());
	return 0;
}

/* ---- rank groups (default since round 4): the array is ordered by its top bytes only; k_bucket_rank puts every bucket-aligned tile of every bin in order
 * inside LDS and counts it there, straight into the tile's span of the free record array; then the fold and the gather of the two-phase output. A tile has
 * two output slots (one per chunk: a tile that outgrows the capacity is taken by two workgroups). ---- */
template <int SIZE>
int rank_group(Slot &s, const std::vector<BinPlan> &bins, u64 *sorted, u64 *scratch, const DevParams &P, u64 lut_entries, const SortPlan &sp, u64 n_total, u32 *d_flag,
               u32 *d_giant, const u64 *d_recs_indirect = nullptr /* indirect sort: `sorted` is the ordered PAIR array (one word per record), the records are here */)
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
	gr.giant = d_giant;
	gr.rec_base = d_recs_indirect;
	gg.tile_prefix[bins.size()] = (u32)(2 * wins);
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
	k_bucket_rank<SIZE, true><<<dim3((u32)wins, 2), dim3(BrCfg<SIZE>::THREADS), br_lds_bytes<SIZE>(), s.stream>>>(gr, P, sp.key_bits, sp.hbits(), n_sh, lut_entries, lut_mask, d_flag);
	/* the tiles with a bucket beyond the LDS capacity (k-mers repeated thousands of times), one workgroup each; nothing listed: a launch that returns */
	k_giant_tiles<SIZE><<<dim3((u32)std::min<u64>(wins, 256)), dim3(GT_THREADS), 0, s.stream>>>(gr, P, (u32)S, sp.key_bits, sp.hbits(), n_sh, lut_entries, lut_mask, err);
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
	/* hybrid: only the top bytes of the key go through HBM passes, the rest is sorted inside LDS (bucket_sort.hip.h) */
	/* ... and then the tiles are counted where they lie (k_bucket_count): possible whenever a tile's records fit its span of the free array */
	/* one-word records (k <= 32) of a default run: the top bytes through HBM, every tile put in order inside LDS by k_bucket_rank, then k_compact as ever */
	/* default run (round 4): the top bytes through HBM, every tile put in order inside LDS by k_bucket_rank and counted there (fused: whenever a tile's
	 * records fit its span of the free array; else, one-word records only, the tile is sorted in place and k_compact follows) */
	static const bool fuse_enabled = [] {
		const char *e = getenv("KMC_HIP_RANK_FUSE"); /* 0 (A/B runs): round 3's default — one-word records ranked in place + k_compact, wider ones k_bucket_count */
		return !e || atoi(e) != 0;
	}();
	const bool can_fuse = count_applicable<SIZE>(P);
	const bool by_rank = !classic && hybrid_mode() == 1 && rank_enabled() && (SIZE == 1 || (can_fuse && fuse_enabled));
	SortPlan sp = by_rank ? plan_sort<SIZE>(N, key_bytes, 2 * k + tag_bits, false, false, true)
	                      : plan_sort<SIZE>(N, key_bytes, 2 * k + tag_bits, classic || !can_fuse, true);
	if (by_rank && !sp.rank && SIZE > 1 && can_fuse) /* the rank plan did not apply (too many key bits left below the buckets): k_bucket_count as in round 3 */
		sp = plan_sort<SIZE>(N, key_bytes, 2 * k + tag_bits, false, true);
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
	const bool indirect = SIZE >= INDIRECT_MIN_WORDS && indirect_enabled && rank_fused && sp.local() && n_pass == 4 && fuse && N < (1ull << 32);
	int rc = 0;
	if (N && ((rc = ensure(s.recA, N * SIZE * 8 + 256)) || (rc = ensure(s.recB, N * SIZE * 8 + 256))))
		return rc;
	if (indirect && ((rc = ensure(s.pairA, N * 8 + 256)) || (rc = ensure(s.pairB, N * 8 + 256)) || (rc = ensure(s.recC, N * SIZE * 8 + 256))))
		return rc;
	u64 rank_tiles = 0;
	if (rank_fused)
		for (const BinPlan &b : bins)
			rank_tiles += (b.n_rec + BrCfg<SIZE>::STRIDE - 1) / BrCfg<SIZE>::STRIDE;
	const ZeroPlan z = plan_group<SIZE>(s, bins, N, n_pass, true, true, true, n_sh > 1 ? (u64)n_sh * lut_entries : 0,
	                                    rank_fused ? (u64)BrCfg<SIZE>::STRIDE : (sp.local() && !sp.rank ? (u64)BcCfg<SIZE>::STRIDE : (u64)CpCfg<SIZE>::TILE), rank_fused ? 2u : 1u,
	                                    rank_tiles);
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
		rc = rank_group<SIZE>(s, bins, sorted, free_array, P, lut_entries, sp, N, flag, zero_ptr<u32>(s, z.giant), indirect ? (const u64 *)s.recA.p : nullptr);
	else if (sp.local() && !sp.rank && N >= 2)
		rc = count_group<SIZE>(s, bins, sorted, free_array, P, lut_entries, sp, N, flag);
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
		std::vector<const kmc_hip_bin_desc *> ptrs;
		for (const auto &d : groups[i].descs)
			ptrs.push_back(&d);
		const bool timed = s.timed;
		s.timed = false;
		const int rc = run_group_device(s, groups[i].P, ptrs.data(), (u32)ptrs.size(), groups[i].lut_entries, true);
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
			snprintf(buf + n, sizeof buf - (size_t)n, "; first time-out: kernel bits 0x%x, lane/digit %u, tile %u of %u waited for tile %d, last word read 0x%08x%08x, %u polls over %.3f s",
			         g_diag[2] & 0xFFFFu, g_diag[2] >> 16, g_diag[3], g_diag[10], (int)g_diag[4], g_diag[6], g_diag[5], g_diag[7],
			         (double)(((u64)g_diag[9] << 32) | g_diag[8]) / 1e8);
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

void kmc_hip_destroy(kmc_hip_ctx *ctx)
{
	if (!ctx)
		return;
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
	if (s.pending || s.hb_pending)
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
		const void *src = superkmers;
		if (!host_ptr_is_pinned(superkmers)) { /* pageable caller (the drop-in's arena): through the slot's pinned staging buffer */
			if ((rc = ensure_pinned(s.h_stage_in, s.h_stage_in_cap, size)))
				return rc;
			memcpy(s.h_stage_in, superkmers, size);
			src = s.h_stage_in;
		}
		HIPCHK(hipMemcpyAsync(s.in.p, src, size, hipMemcpyHostToDevice, s.stream));
		HIPCHK(hipMemsetAsync((char *)s.in.p + size, 0, 256, s.stream));
		HIPCHK(hipMemcpyAsync(s.pack_start.p, ps.data(), (np + 1) * 8, hipMemcpyHostToDevice, s.stream));
	}
	s.out_staged = !P.without_output && (out_capacity || lut_entries) && !host_ptr_is_pinned(out_capacity ? (const void *)out_suffix : (const void *)lut);
	s.timed = true;
	s.sub_P = P;
	s.sub_size = size;
	s.sub_n_rec = n_rec;
	s.sub_np = np;
	s.out_capacity = out_capacity;
	s.lut_entries = lut_entries;
	if ((rc = enqueue_host_bin(s, false)))
		return rc;
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
	if (!s.pending)
		return fail(KMC_HIP_EINVAL, "no bin in flight on this slot");
	s.pending = false;
	HIPCHK(hipEventSynchronize(s.done_ev)); /* blocks in the kernel driver instead of spinning */
	if (int rc = harvest(s))
		return rc;
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
		if (s.out_staged) {
			if (r.out_bytes)
				memcpy(s.h_out, dst_out, r.out_bytes);
			if (lut_bytes)
				memcpy(s.h_lut, dst_lut, lut_bytes);
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
	if (s.pending || s.hb_pending)
		return fail(KMC_HIP_EINVAL, "slot already has a bin in flight");
	const u64 lut_entries = P.kff ? 0 : kmc_hip_lut_entries(params);
	const u64 lut_pitch = up256(lut_entries * 8);
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
	for (u32 i = 0; i < n_bins; ++i) {
		const kmc_hip_host_bin &b = bins[i];
		uint8_t *d_img = (uint8_t *)s.in.p + in_off[i];
		if (b.size) {
			const void *src = b.superkmers;
			if (stage_in) {
				memcpy((char *)s.h_stage_in + in_off[i], b.superkmers, b.size);
				src = (char *)s.h_stage_in + in_off[i];
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
	if (!s.hb_pending)
		return fail(KMC_HIP_EINVAL, "no group of bins in flight on this slot");
	s.hb_pending = false;
	HIPCHK(hipEventSynchronize(s.done_ev));
	if (int rc = harvest(s))
		return rc;
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
		if (s.out_staged)
			for (size_t i = 0; i < n; ++i) {
				if (r.w[i][0])
					memcpy(s.hb[i].h_out, (uint8_t *)s.h_stage_out + off[i] + lut_bytes, r.w[i][0]);
				if (lut_bytes)
					memcpy(s.hb[i].h_lut, (uint8_t *)s.h_stage_out + off[i], lut_bytes);
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
	g_hybrid_override.store(mode, std::memory_order_relaxed);
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
