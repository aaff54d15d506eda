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
#include <chrono>
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
#include "arena_sort.hip.h"
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
	bool carved = false; /* p lies inside a slot's slab (kmc_hip_reserve_slot): never freed on its own */
};

/* A slot's slab: ONE device allocation made ahead of time (kmc_hip_reserve_slot: the drop-in's loader calls it while KMC's stage 1 runs) from which the slot's grow-only
 * buffers are carved. Why: the drop-in's workers allocate ~10 buffers each when stage 2 starts, through a runtime that takes the process's mmap lock — and the reference's
 * reader holds that lock while it unmaps every bin part it has copied (mem_disk_file.cpp:84-100): 16 workers spent 6.6 of 11.9 engine-seconds in those allocations on an
 * 8 Gbp input (profiles/r06/e2e_sweep_8gbp_session_n.jsonl). A buffer that does not fit what is left of the slab is allocated on its own, as before. */
struct Slab {
	void *p = nullptr;
	size_t cap = 0, used = 0;
};
thread_local Slab *t_slab = nullptr; /* the slab of the slot whose mutex this thread holds (SlabScope), if it has one */
struct SlabScope {
	Slab *prev;
	explicit SlabScope(Slab &sl) : prev(t_slab) { t_slab = sl.p ? &sl : nullptr; }
	~SlabScope() { t_slab = prev; }
};

int ensure(DBuf &b, size_t bytes)
{
	if (bytes <= b.cap)
		return 0;
	const size_t want = (bytes + 255) & ~(size_t)255;
	if (t_slab && t_slab->used + want <= t_slab->cap) { /* (what the buffer held before stays where it was: a little of the slab lost when a buffer grows) */
		if (b.p && !b.carved)
			HIPCHK(hipFree(b.p));
		b.p = (char *)t_slab->p + t_slab->used;
		b.cap = want;
		b.carved = true;
		t_slab->used += want;
		return 0;
	}
	if (b.p && !b.carved)
		HIPCHK(hipFree(b.p));
	b.p = nullptr;
	b.cap = 0;
	b.carved = false;
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
constexpr size_t SM_REDO = 60;       /* u32: set by the LDS finisher when a tile of the hybrid sort did not fit: the host sorts the group again, LSD over all bytes */
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
	size_t arena = 0; /* rank groups of one-word records: the arena's AR_DYN_WORDS words, then its AR_MAX_PASS x 256 digit counters (arena_sort.hip.h); 0: no arena */
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
	Slab slab; /* kmc_hip_reserve_slot */
	std::mutex mtx; /* serialises enqueueing on this slot (asynchronous device-resident calls may come from several threads) */
	u64 portion = PORTION_MAX;
	DBuf in, pack_start;
	DBuf recA, recB, recC, pairA, pairB, zero, dbase, out, lut, sticky;
	DBuf bounds;   /* hybrid sort: tile boundaries of k_bucket_bounds, u64[windows + 1] */
	DBuf arena_work; /* rank groups of one-word records: entries, offsets, bucket numbers, heavy chunks, digit bases and look-back rows of the arena's passes (the arenas: pairA / pairB) */
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
	std::vector<u64> sc_cnt; /* records of the launch; bit 63: the pair brackets the LDS sort (k_bucket_bounds + k_bucket_rank), not a scatter pass */
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
	if (rs_lds_bytes<SIZE>() + 32 * 1024 > 65536 && rs_lds_bytes<SIZE>() + 32 * 1024 <= 160 * 1024) /* + room for $KMC_HIP_SCATTER_LDS_PAD */
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_onesweep<SIZE>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                           (int)rs_lds_bytes<SIZE>() + 32 * 1024));
	else if (rs_lds_bytes<SIZE>() > 65536)
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_onesweep<SIZE>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                           (int)rs_lds_bytes<SIZE>()));
	if (br_lds_bytes<SIZE>() > 65536) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bucket_rank<SIZE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes<SIZE>() + 32 * 1024)); /* + room for $KMC_HIP_RANK_LDS_PAD */
		if (SIZE == 1) {
			HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bucket_rank<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)br_lds_bytes<1>()));
		}
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
	for (DBuf *b : {&s.in, &s.pack_start, &s.recA, &s.recB, &s.recC, &s.pairA, &s.pairB, &s.zero, &s.dbase, &s.out, &s.lut, &s.sticky, &s.bounds, &s.arena_work, &s.redo_log, &s.hb_res})
		if (b->p && !b->carved)
			(void)hipFree(b->p);
	if (s.slab.p)
		(void)hipFree(s.slab.p);
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

/* The rest of this translation unit, in the order it is compiled (round 5: one 3 400-line file split along its entry families; no kernel and no statement changed).
 * tests/emu.py build_hostlib inlines these files again before it rewrites the kernel launches for the CPU emulation. */
#include "host_plan_and_groups.hip.h" /* the sort planner, the HBM passes, the groups of bins (front end, LSD / bucket-count / rank finishers), the redo of flagged groups */
#include "host_hooks_and_order.hip.h" /* stage-isolating test hooks (device side) and the globally ordered database */
#include "host_multi_device_bin.hip.h" /* one bin over all devices of the context */
#include "host_cabi_stage2.hip.h" /* the C-ABI of stage 2: lifetime, narrow boundary, full boundary (one bin / several bins per call, device-resident batches) */
#include "host_cabi_stage1.hip.h" /* the C-ABI test hooks and stage 1 on the device (reads -> bins in HBM; one part of input text -> bin records) */
#include "host_cabi_collective.hip.h" /* the tally all-reduce over the devices of a context (RCCL) and the instrumentation entries */
