/*
 * include/kmc_hip.h — C-ABI of libkmc_hip.so: KMC stage-2 "bin sort & count" on MI355X (gfx950).
 *
 * This is the drop-in boundary. Plain C: pointers + sizes, no C++/torch types.
 * Every entry point names the reference interface it replaces (paths relative to the KMC 3.2.4
 * tree, /root/reference). The C++ worker that binds these inside kmc_core is
 * kmc_amd/host/kb_sorter_plugin.h; INTEGRATION.md shows the maintainer-side change.
 *
 * Conventions
 *   - return 0 on success, a negative KMC_HIP_E* code on failure; never throws, never aborts.
 *     kmc_hip_last_error(ctx) gives the message (the worker forwards it to
 *     CCriticalErrorHandler::HandleCriticalError, critical_error_handler.h:9-90).
 *   - `dev` is an index into the device list given to kmc_hip_init (not a HIP ordinal).
 *   - host threads may call concurrently as long as each uses its own (dev, slot) pair (mirrors one
 *     CWKmerBinSorter thread per sorter, kmc.h:1576-1584); calls on one pair must be serialised by the caller.
 *   - host buffers belong to the caller; device memory, streams and events belong to the library.
 *   - records are CKmer<SIZE> PODs (kmer.h:22-67): `words` x uint64, word 0 least significant.
 */
#ifndef KMC_HIP_H
#define KMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMC_HIP_ABI_VERSION 4 /* 3: + kmc_hip_process_bins_submit/_wait (bound by the worker's loader), kmc_hip_process_bin_multi, kmc_hip_order_database_device;
                               * 4: kmc_hip_split_params.part_kind (long-read parts) */

enum {
	KMC_HIP_OK = 0,
	KMC_HIP_EINVAL = -1,   /* bad argument / unsupported parameter combination */
	KMC_HIP_EDEVICE = -2,  /* HIP runtime error (message has the hipError string) */
	KMC_HIP_ENOMEM = -3,   /* device or pinned-host allocation failed */
	KMC_HIP_ECORRUPT = -4, /* super-k-mer stream is ragged or disagrees with n_rec / pack sizes */
	KMC_HIP_ECAPACITY = -5,/* out_capacity too small for the counted k-mers */
	KMC_HIP_EINTERNAL = -6 /* device-side watchdog tripped (look-back spin bound) */
};

typedef struct kmc_hip_ctx kmc_hip_ctx;

/* Per-run constants the reference sorter copies out of CKMCParams (kb_sorter.h:165-200). */
typedef struct kmc_hip_bin_params {
	uint32_t kmer_len;       /* Params.kmer_len, 1..256 */
	uint32_t both_strands;   /* Params.both_strands: 1 = canonical k-mers */
	uint32_t cutoff_min;     /* Params.cutoff_min */
	uint32_t without_output; /* Params.without_output: tallies only, out_bytes = 0, lut untouched */
	uint64_t cutoff_max;     /* Params.cutoff_max (compared as uint32, kb_sorter.h:186) */
	uint64_t counter_max;    /* Params.counter_max */
	uint32_t lut_prefix_len; /* Params.lut_prefix_len (kmc.h:1434-1469); 0 for KFF */
	uint32_t output_type;    /* 0 = OutputType::KMC, 1 = OutputType::KFF (kb_sorter.h:1043-1049) */
} kmc_hip_bin_params;

/* stats[] order == the four tallies of CKmerQueue::push (queues.h:826, kb_sorter.h:1273) */
enum { KMC_HIP_STAT_UNIQUE = 0, KMC_HIP_STAT_CUTOFF_MIN = 1, KMC_HIP_STAT_CUTOFF_MAX = 2, KMC_HIP_STAT_TOTAL = 3 };

/* ---- lifetime ----------------------------------------------------------------------------- */

/* Create a context on `n_dev` HIP devices (ordinals in device_ids; NULL => {0..n_dev-1}).
 * Replaces: construction of CWKmerBinSorter<SIZE> workers + pmm_radix_buf pool (kmc.h:1510,1576-1584). */
int kmc_hip_init(const int *device_ids, int n_dev, kmc_hip_ctx **out);
void kmc_hip_destroy(kmc_hip_ctx *ctx);
const char *kmc_hip_last_error(kmc_hip_ctx *ctx); /* thread-local message of the calling thread's last failure */
int kmc_hip_abi_version(void);
/* 0 = HIP on a GPU (the product). Test builds say otherwise — 1: this library's source over the CPU emulation of tests/hipemu, 2: the mock of
 * tests/hipemu/mock_hip_lib.cpp — so that whatever measures or certifies (bench.py, __graft_entry__.smoke) can refuse them. */
int kmc_hip_backend_kind(void);
int kmc_hip_num_devices(kmc_hip_ctx *ctx);
int kmc_hip_device_count(void); /* HIP devices visible to the process (0 when the runtime is unusable) */
int kmc_hip_num_slots(void); /* stream slots per device usable with _submit/_wait (independent bins in flight) */

/* Derived sizes, so callers size buffers exactly like kb_reader.h:141-165 does. */
uint32_t kmc_hip_words(uint32_t kmer_len);                                   /* SIZE = ceil(k/32) */
uint32_t kmc_hip_counter_size(uint64_t cutoff_max, uint64_t counter_max);    /* defs.h:154-159 */
uint32_t kmc_hip_out_rec_bytes(const kmc_hip_bin_params *p);                 /* suffix bytes + counter bytes */
uint64_t kmc_hip_lut_entries(const kmc_hip_bin_params *p);                   /* 4^p, 0 when p == 0 */

/* ---- narrow boundary: the sort alone ------------------------------------------------------- */

/* Ascending sort of n records of `words` uint64 by their low `key_bytes` bytes (higher bytes must be
 * zero), result left IN `recs` (host memory).
 * Replaces: SortFunction<CKmer<SIZE>> (raduls.h:19-20) = RadulsSort::RadixSortMSD_* (raduls_impl.h:769-776)
 * / RadixSort::RadixSortMSD (radix.h:845-852) as called at kb_sorter.h:775; key_bytes = rec_len there.
 * (The reference leaves the result in `tmp` when rec_len is odd; this entry always returns it in place — see _into.) */
int kmc_hip_sort_records(kmc_hip_ctx *ctx, int dev, void *recs, uint64_t n, uint32_t words, uint32_t key_bytes);

/* Same, with the sorted records delivered to `dst` (host memory; may equal recs). This is what the SortFunction adapter
 * (kmc_amd/host/hip_sort_function.h) binds: the reference wants the result in `tmp` when key_bytes is odd and in `kmers`
 * when it is even (kb_sorter.h:776-779, raduls_impl.h:552-561), so the adapter passes dst = tmp or kmers. */
int kmc_hip_sort_records_into(kmc_hip_ctx *ctx, int dev, const void *recs, void *dst, uint64_t n, uint32_t words, uint32_t key_bytes);

/* Same, on device-resident records (d_recs, d_tmp: n*words*8 bytes each, 256-B aligned). On return
 * *d_result points at whichever of the two holds the sorted records. Stream-synchronous. */
int kmc_hip_sort_records_device(kmc_hip_ctx *ctx, int dev, void *d_recs, void *d_tmp, uint64_t n, uint32_t words,
                                uint32_t key_bytes, void **d_result);

/* ---- full boundary: one bin, super-k-mer bytes in -> (suffix,count) records + LUT + tallies out --- */

/* Replaces: one iteration of CKmerBinSorter<SIZE>::ProcessBins (kb_sorter.h:210-237) = Expand (:728-752)
 * + Sort (:757-780) + Compact (:1287; CompactKmers :1128-1281 / CompactKxmers :937-1122), i.e. everything
 * between sorters_manager->GetNext (queues.h:2087) and kq->push (queues.h:826).
 *   superkmers/size : the bin image CKmerBinReader read (kb_reader.h:176-190); format kb_collector.cpp:57-71
 *   n_rec           : CBinDesc n_rec of the bin (number of k-mers)
 *   pack_bytes/n_packs : byte length of each expander pack, in order (CExpanderPackDesc::pop, queues.h:390;
 *                     first member of each pair). n_packs == 0 => the library finds pack boundaries itself.
 *   out_suffix/out_capacity : the bin's mba_suffix slot; capacity as kb_reader.h:141-150
 *   out_bytes       : bytes written = the single data pack [0, out_bytes) pushed to kq
 *   lut             : the bin's mba_lut slot, kmc_hip_lut_entries() uint64, per-bin COUNTS (the completer makes
 *                     them cumulative, kb_completer.cpp:193-199); zero-filled by the callee
 *   stats           : {n_unique, n_cutoff_min, n_cutoff_max, n_total}
 * out_suffix / lut MAY alias superkmers (they do in the reference arena when rec_len is even,
 * queues.h:1352-1369): all input is on the device before the first output byte is written.
 * Empty bins (size == 0, n_rec == 0) are valid and produce zeros. */
int kmc_hip_process_bin(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint8_t *superkmers,
                        uint64_t size, uint64_t n_rec, const uint64_t *pack_bytes, uint64_t n_packs,
                        uint8_t *out_suffix, uint64_t out_capacity, uint64_t *out_bytes, uint64_t *lut,
                        uint64_t stats[4]);

/* Several bins per call through the same boundary: the host-side twin of kmc_hip_process_bins_device's grouping. The n_bins (1..16) bins are
 * uploaded together and — as far as the spare bits of the top radix digit can tag them (4 bins at k = 27, 55, 127; see kmc_hip_process_bins_device) —
 * SORTED TOGETHER: launches 4x as large and 4x fewer, which is worth ~10 % at 48 M k-mers per bin and several x on small bins. A worker that finds
 * more than one bin waiting (CBinQueue::pop_if_any, queues.h:751) hands them over in one call; every bin keeps its own buffers and gets its own
 * out_bytes / tallies, the bytes are those of n_bins separate kmc_hip_process_bin calls. _wait: out_bytes[n_bins], stats[n_bins][4], in the order of
 * the descriptors. An error in any bin fails the call (the device error word belongs to the stream). Host buffers must stay valid until _wait returns.
 * Replaces: n_bins iterations of CKmerBinSorter<SIZE>::ProcessBins (kb_sorter.h:210-237). */
typedef struct kmc_hip_host_bin {
	const uint8_t *superkmers;
	uint64_t size, n_rec;
	const uint64_t *pack_bytes;
	uint64_t n_packs;
	uint8_t *out_suffix;
	uint64_t out_capacity;
	uint64_t *lut;
} kmc_hip_host_bin;
int kmc_hip_process_bins_submit(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_bin_params *params, const kmc_hip_host_bin *bins, uint32_t n_bins);
int kmc_hip_process_bins_wait(kmc_hip_ctx *ctx, int dev, int slot, uint64_t *out_bytes, uint64_t *stats);

/* The same bin over ALL devices of the context: the oversized-bin path.
 * Replaces: strict-memory mode's treatment of a bin that does not fit — the reference cuts it into sub-bins by the k-mers' leading symbols, sorts
 * them one after the other through the same sort_func and merges (kmc.h:1607-1692, bkb_sorter.h:187, bkb_subbin.h / bkb_merger.h). Here the cut goes
 * across GPUs: every device expands a share of the expander packs, the records are exchanged by the TOP byte of the key (one all-to-all: RCCL
 * ncclSend/ncclRecv over xGMI between distinct GPUs, peer copies when the context names one GPU more than once), every device sorts and compacts the
 * k-mers of its key range, and the ranges are emitted in order. Arguments, outputs and errors as kmc_hip_process_bin; the result is byte-identical to
 * that call's. Synchronous; takes the first stream slot of every device. */
int kmc_hip_process_bin_multi(kmc_hip_ctx *ctx, const kmc_hip_bin_params *params, const uint8_t *superkmers, uint64_t size, uint64_t n_rec,
                              const uint64_t *pack_bytes, uint64_t n_packs, uint8_t *out_suffix, uint64_t out_capacity, uint64_t *out_bytes,
                              uint64_t *lut, uint64_t stats[4]);

/* Asynchronous pair for double buffering: _submit enqueues H2D + kernels + D2H on the device's stream slot
 * `slot` (0 .. kmc_hip_num_slots()-1) and returns; _wait blocks until that slot's bin is complete and fills out_bytes/stats.
 * Host buffers must stay valid (and out must not alias in) until _wait returns. */
int kmc_hip_process_bin_submit(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_bin_params *params,
                               const uint8_t *superkmers, uint64_t size, uint64_t n_rec, const uint64_t *pack_bytes,
                               uint64_t n_packs, uint8_t *out_suffix, uint64_t out_capacity, uint64_t *lut);
int kmc_hip_process_bin_wait(kmc_hip_ctx *ctx, int dev, int slot, uint64_t *out_bytes, uint64_t stats[4]);

/* Device-resident variant (inputs already in HBM; used by bench.py so the timed region excludes PCIe, and by
 * callers that produce bins on the GPU). All d_* pointers are device memory on `dev`:
 *   d_superkmers[size (+16 B readable slack)], d_pack_start[n_packs+1] = byte offsets of pack starts, last = size
 *   d_out[out_capacity], d_lut[kmc_hip_lut_entries()], d_stats[4] (uint64, written by the device)
 *   d_out_bytes: 1 uint64 written by the device.
 * Returns after enqueueing unless `sync` != 0. Asynchronous calls are spread round-robin over several internal streams
 * (bins are independent), so consecutive small bins overlap (a bin with more than 2 GiB of record arrays always takes
 * the first stream: it fills the GPU alone); kmc_hip_synchronize(dev) waits for all of them and reports any deferred
 * device error (errors are kept in a per-stream sticky word that only a synchronising call clears). Output buffers of calls in flight must be distinct. */
int kmc_hip_process_bin_device(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params,
                               const uint8_t *d_superkmers, uint64_t size, uint64_t n_rec,
                               const uint64_t *d_pack_start, uint64_t n_packs, uint8_t *d_out, uint64_t out_capacity,
                               uint64_t *d_out_bytes, uint64_t *d_lut, uint64_t *d_stats, int sync);

/* Many device-resident bins in one call (per-GPU bin queue): bin i is enqueued on internal stream
 * (i mod n_streams), in index order per stream, by one host thread per stream — a group of bins is 13-14 launches, so hundreds of
 * small bins are bound by the host's launch rate unless several threads submit (KMC's default is 512 bins per run,
 * kmc.h n_bins). n_streams <= 0 picks the default (8), capped at kmc_hip_num_slots(). Returns after enqueueing;
 * kmc_hip_synchronize(dev) waits and reports deferred device errors. Output buffers of all bins must be distinct.
 * Consecutive bins of a stream are SORTED TOGETHER when the top radix digit has spare bits (8 ceil(k/4) - 2k of them: groups of 4 at
 * k = 27, 55, 127): the bin's number inside the group is kept in those bits while the records are on the device, so one set of passes
 * orders the group bin-major; front end and compaction stay per bin. $KMC_HIP_GROUP caps the group size (1 = every bin on its own).
 * Replaces: the hand-out of bins to n_sorters CWKmerBinSorter threads (kmc.h:1576-1584, queues.h:2087-2128). */
typedef struct kmc_hip_bin_desc {
	const uint8_t *d_superkmers;
	uint64_t size, n_rec;
	const uint64_t *d_pack_start;
	uint64_t n_packs;
	uint8_t *d_out;
	uint64_t out_capacity;
	uint64_t *d_out_bytes, *d_lut, *d_stats;
} kmc_hip_bin_desc;
int kmc_hip_process_bins_device(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const kmc_hip_bin_desc *bins,
                                uint64_t n_bins, int n_streams);

/* ---- a globally ordered database ---------------------------------------------
 * The bins of a run — as kmc_hip_process_bin[s]_device left them: per bin `d_out` (suffix records, ascending inside the bin), `d_out_bytes`, `d_lut`
 * (records per lut_prefix_len-symbol prefix) — merged into ONE ascending sequence of all counted k-mers: records of (kmer_len - out_lut_prefix_len) / 4
 * suffix bytes (most significant first) + counter bytes (least significant first) into d_out, and into d_lut_out[4^out_lut_prefix_len] the number of
 * records with a prefix BELOW each entry — the body of the database `kmc_tools transform <db> sort <out>` writes (kmc_tools/kmc1_db_writer.h:368-395;
 * the header and the 'KMCP'/'KMCS' markers around it are the caller's, :309-370). Every k-mer is in exactly one bin (bins partition by signature), so
 * this is a sort, not a merge of counts: records unpacked to (k-mer, count), ordered by the library's LSD passes, packed again. Synchronous; *n_kmers =
 * records written. params: the parameters the bins were made with (KMC output, lut_prefix_len > 0); kmer_len <= 224.
 * Replaces: kmc_tools' sort operation on the CPU (CKMC1DbWriter fed by a priority queue over the bins). */
int kmc_hip_order_database_device(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const kmc_hip_bin_desc *bins, uint64_t n_bins,
                                  uint32_t out_lut_prefix_len, uint8_t *d_out, uint64_t out_capacity, uint64_t *d_lut_out, uint64_t *n_kmers);

/* ---- end-of-run tallies ---------------------------------------------------------------------- */

/* Sum stats[4] over the context's devices with one RCCL all-reduce (ncclUint64 x 4, ncclSum) over xGMI.
 * in/out: per_dev_stats[n_dev][4] host array -> every row holds the sum. With one device this is a device
 * round-trip through the same code path. Replaces: the completer's running sums n_unique.. (kb_completer.cpp:206-209)
 * when bins are sharded over GPUs. */
int kmc_hip_allreduce_stats(kmc_hip_ctx *ctx, uint64_t *per_dev_stats);

/* ---- instrumentation --------------------------------------------------------------------------- */

/* Per-phase device time of the last completed bin on (dev): ms[0]=index (pack scan), [1]=expand, [2]=histogram,
 * [3]=radix scatter passes, [4]=compact, [5]=total enqueue->done; measured with hipEvents on the bin's stream.
 * Replaces: USE_TIMERS / MEASURE_TIMES compile-time probes (raduls_impl.h:30,567-657). */
int kmc_hip_last_timings(kmc_hip_ctx *ctx, int dev, float ms[6]);
/* Radix scatter (k_onesweep) launches of every bin completed on `dev` since the last reset: how many, their summed
 * duration (HIP events around each launch, on the launch's own stream) and the records they moved — the roofline
 * input of bench.py. Waits for the device's streams. */
int kmc_hip_scatter_totals(kmc_hip_ctx *ctx, int dev, int reset, uint64_t *n_launches, double *total_ms, uint64_t *total_records);
/* The LDS half of the hybrid sort (k_bucket_bounds + k_bucket_rank, kmc_amd/csrc/bucket_sort.hip.h; replaces the small-bucket recursion and
 * CSmallSort of raduls_impl.h:133-141,497-510 / small_sort.h:29-179): launches, summed duration (HIP events on the launch's stream) and records
 * since the last reset; plus, process-wide, how many groups of bins took the hybrid sort and how many of them had to be sorted again with LSD
 * passes over every byte because a bucket did not fit a tile. Any pointer may be NULL. Waits for the device's streams. */
int kmc_hip_local_sort_totals(kmc_hip_ctx *ctx, int dev, int reset, uint64_t *n_launches, double *total_ms, uint64_t *total_records,
                              uint64_t *n_hybrid_groups, uint64_t *n_redo_groups);
/* Which sort stage 2 runs (process-wide; overrides $KMC_HIP_HYBRID; tests and tuning): 0 = 8-bit LSD passes over every key byte + k_compact (rounds 1-2);
 * 1 = default: LSD passes over the top key bytes only, then every bucket-aligned tile ranked and counted inside LDS by k_bucket_rank (round 4: every record
 * width; $KMC_HIP_RANK_FUSE=0: rank-in-place + k_compact for k <= 32, LSD passes for wider records); -h = `h` top bytes forced. (2 — round 3's k_bucket_count /
 * k_bucket_sort, which left the library in round 6 — is taken as 1.) Also clears the group counters of kmc_hip_local_sort_totals and
 * kmc_hip_path_counters. Returns the mode that was in force. */
int kmc_hip_set_hybrid(int mode);
/* Groups of bins (a bin on its own is a group of one) by the path their sort + compaction took, process-wide since the last kmc_hip_set_hybrid: [0] top bytes
 * through HBM, tiles ranked AND counted inside LDS (k_bucket_rank fused: the default since round 4, every record width); [1] ranked in place, then k_compact
 * (one-word records whose output may outgrow a tile's span); [2] unused since round 6 (0; was k_bucket_count); [3] 8-bit
 * LSD passes over every key byte + k_compact (rounds 1-2; redo runs; tiny groups). With a context (ctx != NULL; waits for the device's streams): [4] tiles whose
 * largest bucket did not fit LDS and that k_giant_tiles sorted on their own (k-mers repeated thousands of times) and [5] the records in them, since the context
 * was made. [6] the groups of [0] (records of three words and more) whose HBM passes moved one word per record — the key's top four bytes above the record's
 * number — with the records gathered by number inside k_bucket_rank (process-wide, like [0..3]). [7] reserved (0). What replaces raduls_impl.h:216-520,
 * :680-737 / small_sort.h for a group. */
int kmc_hip_path_counters(kmc_hip_ctx *ctx, int dev, uint64_t counters[8]);
/* Diagnostics (process-wide, since the library was loaded): where the host-boundary calls (kmc_hip_process_bin_submit/_wait, kmc_hip_process_bins_submit/_wait) spent
 * their wall time, in seconds summed over all slots: [0] pack starts + device buffers, [1] staging copy in (callers whose images are in ordinary memory), [2] enqueueing
 * copies and launches, [3] waiting for the kernels (includes the H2D copy in front of them and whatever other slots queued before), [4] D2H of the exact-size results,
 * [5] staging copy out (callers whose output buffers are ordinary memory); [6] calls, [7] redo rounds (counts). What the drop-in's worker report prints under
 * $KMC_HIP_VERBOSE; no reference counterpart (the CPU worker has no host link). */
int kmc_hip_host_boundary_times(double seconds[8]);
/* Optional, once per (device, slot), before the slot's first bin: ONE device allocation of `bytes` from which the host-boundary entries (kmc_hip_process_bin_submit/_wait,
 * kmc_hip_process_bins_submit/_wait) carve the slot's grow-only buffers; a buffer that does not fit what is left is allocated on its own, as without a slab. Lets a caller
 * whose process is busy mapping and unmapping memory while bins arrive (the reference's RAM-only stage 2: mem_disk_file.cpp:84-100) pay for device allocations ahead of time —
 * the drop-in's loader reserves while KMC's stage 1 runs. What the CPU worker gets from CMemoryBins, reserved by the reader before the worker sees the bin
 * (queues.h:1288-1340). Refused (KMC_HIP_ECAPACITY, nothing allocated) when it would leave the device with less than half of its memory free. Freed with the context. */
int kmc_hip_reserve_slot(kmc_hip_ctx *ctx, int dev, int slot, uint64_t bytes);
/* Device memory helpers so non-HIP callers (ctypes tests, the C++ worker) need not link HIP themselves. */
int kmc_hip_malloc(kmc_hip_ctx *ctx, int dev, uint64_t bytes, void **d_ptr);
int kmc_hip_free(kmc_hip_ctx *ctx, int dev, void *d_ptr);
int kmc_hip_memcpy_h2d(kmc_hip_ctx *ctx, int dev, void *d_dst, const void *src, uint64_t bytes);
int kmc_hip_memcpy_d2h(kmc_hip_ctx *ctx, int dev, void *dst, const void *d_src, uint64_t bytes);
int kmc_hip_host_register(kmc_hip_ctx *ctx, void *ptr, uint64_t bytes);   /* pin the caller's arena (CMemoryBins buffer) */
int kmc_hip_host_unregister(kmc_hip_ctx *ctx, void *ptr);
int kmc_hip_host_alloc(kmc_hip_ctx *ctx, uint64_t bytes, void **ptr);      /* pinned host memory (hipHostMalloc) for _submit/_wait callers */
int kmc_hip_host_free(kmc_hip_ctx *ctx, void *ptr);
int kmc_hip_synchronize(kmc_hip_ctx *ctx, int dev);

/* ---- stage-isolating test hooks (not used by the worker): run only index+expand, or only compaction ---- */
int kmc_hip_debug_expand(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint8_t *superkmers, uint64_t size,
                         uint64_t n_rec, const uint64_t *pack_bytes, uint64_t n_packs, uint64_t *out_recs /* n_rec*words */);
int kmc_hip_debug_compact(kmc_hip_ctx *ctx, int dev, const kmc_hip_bin_params *params, const uint64_t *sorted_recs, uint64_t n,
                          uint8_t *out_suffix, uint64_t out_capacity, uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4]);

/* ---- stage 1, first kernels (groundwork, NOT part of the drop-in) ----
 * codes: one symbol per byte, 0..3 = A C G T, anything negative = N or a read boundary (join reads with one such byte). Computes on the
 * device what CSplitter::ProcessReads (splitter.cpp:557-672) computes read by read: the minimizer signature (CMmer, kmc_api/mmer.h:25-110,
 * signature_len 5..11) of the k-mer starting at every position -> sig[n] (0xFFFFFFFF where no valid k-mer starts), and the super-k-mers in
 * position order -> sk_pos / sk_len (symbols: kmer_len + extra, extra <= 255) / sk_sig, *n_sk of them (KMC_HIP_ECAPACITY beyond sk_cap).
 * The bin of a super-k-mer is s_mapper->get_bin_id(signature) (s_mapper.h, stays with the reference), its record kb_collector.cpp:57-71. */
int kmc_hip_debug_split_reads(kmc_hip_ctx *ctx, int dev, const int8_t *codes, uint64_t n, uint32_t kmer_len, uint32_t signature_len, uint32_t *sig,
                              uint64_t *sk_pos, uint32_t *sk_len, uint32_t *sk_sig, uint64_t sk_cap, uint64_t *n_sk);

/* ---- stage 1 on the device (groundwork, NOT yet wired into the reference's stage 1) ----
 * Reads already in HBM as codes (as above) -> the signature bins of CSplitter::ProcessReads + CKmerBinCollector (splitter.cpp:557-672,
 * kb_collector.cpp:57-106), left IN HBM in the layout kmc_hip_process_bins_device takes: no trip through the host, no temporary files
 * (CKmerBinStorer, kb_storer.cpp) between the stages.
 *   _plan : signatures, super-k-mers, per-bin totals. d_sig_to_bin[4^signature_len + 1] (device, int32) is the reference's signature map
 *           (CSignatureMapper::get_bin_id, s_mapper.h:232; the map itself is built from stage-0 statistics and stays with the reference).
 *           Fills the caller's HOST arrays: bin_base[n_bins + 1] (byte offset of each bin image in the buffer to allocate, 256-byte aligned;
 *           the last entry is the buffer size incl. the readable slack stage 2 wants), bin_bytes / bin_superkmers / bin_kmers[n_bins],
 *           pack_base[n_bins + 1] (first entry of each bin in the pack-start array; the last entry is the array length).
 *   _emit : writes the bin images into d_bins[bin_base[n_bins]] and the pack boundaries into d_pack_start[pack_base[n_bins]]; d_codes and
 *           d_sig_to_bin of the plan must still be valid. Afterwards bin b is
 *           kmc_hip_bin_desc{d_bins + bin_base[b], bin_bytes[b], bin_kmers[b], d_pack_start + pack_base[b], pack_base[b+1] - pack_base[b] - 1, ...}.
 *           Super-k-mers of a bin are NOT in read order (neither are the reference's with several splitter threads); stage 2 is order-blind.
 *   _free : releases the plan (always call it, also after a failed _emit).
 * One call handles up to 2^41 symbols; the plan keeps 16 bytes per super-k-mer (one per 10-40 symbols of real reads at k = 27) until _free. */
typedef struct kmc_hip_s1_plan kmc_hip_s1_plan;
int kmc_hip_split_reads_plan(kmc_hip_ctx *ctx, int dev, const int8_t *d_codes, uint64_t n, uint32_t kmer_len, uint32_t signature_len, const int32_t *d_sig_to_bin,
                             uint32_t n_bins, kmc_hip_s1_plan **plan, uint64_t *bin_base, uint64_t *bin_bytes, uint64_t *bin_superkmers, uint64_t *bin_kmers,
                             uint64_t *pack_base);
int kmc_hip_split_reads_emit(kmc_hip_ctx *ctx, kmc_hip_s1_plan *plan, uint8_t *d_bins, uint64_t *d_pack_start);
void kmc_hip_split_reads_free(kmc_hip_ctx *ctx, kmc_hip_s1_plan *plan);

/* ---- stage 1, one part of input text (the engine behind kmc_amd/host/kb_splitter_plugin.h; emulation-validated) ----
 * Replaces, for one part the reference's readers cut (fastq_reader.cpp), CSplitter::ProcessReads and the n_bins CKmerBinCollectors up to the
 * bin-part buffers (splitter.cpp:557-672, kb_collector.cpp:34-106): text (host) -> the part's bin records (host, bin b at recs + bin_off[b],
 * bin_bytes[b] bytes; bins are 256-byte aligned. *recs_bytes = bytes of `recs` the part needs: about 0.3 bytes per symbol + 256 per bin for
 * real reads, but up to 1 + k/4 bytes per k-mer when every k-mer is its own super-k-mer — KMC_HIP_ECAPACITY asks for a second call with a
 * larger buffer) and per bin the three sums a
 * collector keeps: bin_kmers (n_recs), bin_superkmers (n_super_kmers), bin_plus_x (n_plus_x_recs, kb_collector.h:72-118); *n_reads = titles
 * in the part. kmc_hip_split_set_map uploads CSignatureMapper's map (s_mapper.h:232; 4^signature_len + 1 entries) once per device.
 * part_kind 1 = a part the reader labelled ReadType::long_read (queues.h:40: a record longer than the reader's buffer, handed out in parts that
 * overlap by k - 1 symbols): an optional title, then symbols only (CSplitter::GetSeqLongRead, splitter.cpp:70-86). Lines of mem_part_pmm_reads
 * symbols or more inside an ordinary part, and long-read parts, reach the reference's super-k-mer loop in pieces that overlap by k - 1 symbols
 * (splitter.cpp:141-145, :226-231, :80-84); the kernels cut super-k-mers at the same piece starts, so the three sums agree with the reference's.
 * Returns 0, a negative KMC_HIP_E* code, or KMC_HIP_UNCOVERED: the text is MALFORMED in a way CSplitter::GetSeq tolerates and the kernels do not
 * reproduce (blank lines, quality of another length than its sequence, a lone '\r', control characters) — nothing was produced; the stage-1 worker
 * stops the run with an error (it has no path into the reference splitter unless built with -DKMC_HIP_S1_REFERENCE_FALLBACK). Calls on one
 * (dev, slot) are serialised. */
#define KMC_HIP_UNCOVERED 1
typedef struct kmc_hip_split_params {
	uint32_t kmer_len, signature_len, n_bins, max_x; /* max_x: CKMCParams::max_x (0..3) */
	uint32_t both_strands;
	uint32_t file_type;                              /* 0 = FASTA (one line per sequence), 1 = FASTQ */
	uint64_t line_cap;                               /* CKMCParams::mem_part_pmm_reads */
	uint32_t part_kind;                              /* 0 = whole records (ReadType::normal_read), 1 = ReadType::long_read */
	uint32_t reserved;                               /* 0 */
} kmc_hip_split_params;
int kmc_hip_split_set_map(kmc_hip_ctx *ctx, int dev, const int32_t *sig_to_bin, uint32_t signature_len);
int kmc_hip_split_part(kmc_hip_ctx *ctx, int dev, int slot, const kmc_hip_split_params *p, const uint8_t *text, uint64_t size, uint8_t *recs,
                       uint64_t recs_capacity, uint64_t *recs_bytes, uint64_t *bin_off, uint64_t *bin_bytes, uint64_t *bin_kmers, uint64_t *bin_superkmers, uint64_t *bin_plus_x,
                       uint64_t *n_reads);

#ifdef __cplusplus
}
#endif
#endif /* KMC_HIP_H */
