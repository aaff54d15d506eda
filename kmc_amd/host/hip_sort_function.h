/*
 * kmc_amd/host/hip_sort_function.h — the NARROW drop-in boundary: KMC's CPU radix sorters replaced by
 * the MI355X sort behind include/kmc_hip.h, with the rest of the reference's stage 2 (CKmerBinSorter's Expand.. and
 * Compact.. steps, kb_sorter.h) left exactly as it is.
 *
 * The reference selects its sorter as a
 *     SortFunction<CKmer<SIZE>> = std::function<void(KMER_T* kmers, KMER_T* tmp, uint64 n_recs, uint32 byte, uint32 n_threads,
 *                                                    CMemoryPool* pmm_radix_buf)>                         (raduls.h:19-20)
 * out of RadulsSort::RadixSortMSD_{SSE2,SSE41,AVX,AVX2,NEON} (raduls.h:22-40, raduls_impl.h:769-776) or
 * RadixSort::RadixSortMSD<KMER_T,SIZE> (radix.h:845-852) at kmc.h:1521-1562, and calls it at kb_sorter.h:775 (per bin) and
 * bkb_sorter.h:187 (strict-memory mode). This header takes the place of raduls.h AND radix.h (their include guards are
 * taken here; compile kmc_runner.cpp with -include hip_sort_function.h and link without raduls_*.o): every one of those
 * entry points keeps its name and signature and forwards to kmc_hip_sort(), so whichever branch kmc.h:1521-1562 takes on
 * the host CPU, `sort_func` is the GPU sort. No reference source is modified.
 *
 * Contract kept (kb_sorter.h:757-780, raduls_impl.h:552-561): ascending by bytes `byte`..0 (higher bytes are zero by
 * construction); the sorted records are left in `tmp` when (byte + 1) is odd and in `kmers` when it is even; both arrays
 * belong to the caller; may be called from several sorter threads at once; errors surface as CCriticalErrorHandler errors.
 * n_threads and pmm_radix_buf (the CPU sorters' thread count and write-combining buffer pool) are not needed.
 */
#ifndef KMC_AMD_HIP_SORT_FUNCTION_H
#define KMC_AMD_HIP_SORT_FUNCTION_H

#if defined(RADULS_H) || defined(_RADIX_H)
#error "hip_sort_function.h must be included before (instead of) raduls.h / radix.h"
#endif
#define RADULS_H
#define _RADIX_H

#include <functional>
#include <string>
#include "defs.h"
#include "kmer.h"
#include "critical_error_handler.h"

#define MAGIC_NUMBER 8 /* raduls.h:17 — kmc.h:374-376 sizes pmm_radix_buf with it */

class CMemoryPool;

template <typename KMER_T> using SortFunction = std::function<void(KMER_T *, KMER_T *, uint64, uint32, uint32, CMemoryPool *)>;

/* hip_loader.cpp: sorts n records of `words` uint64 from `recs` into `dst` (dst may be recs) on the GPU through the C-ABI
 * (kmc_hip_sort_records_into); returns 0 or a KMC_HIP_E* code and the message */
int kmc_hip_host_sort(const void *recs, void *dst, uint64_t n, uint32_t words, uint32_t key_bytes, std::string &err);

template <typename KMER_T> inline void kmc_hip_sort(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 /*n_threads*/, CMemoryPool * /*pmm_radix_buf*/)
{
	static_assert(sizeof(KMER_T) % 8 == 0 && sizeof(KMER_T) <= 64, "CKmer<SIZE> is SIZE x uint64 (kmer.h:22-67)");
	const uint32 key_bytes = byte + 1;
	KMER_T *dst = (key_bytes & 1) ? tmp : kmers; /* result placement rule of the reference sorters */
	std::string err;
	int rc = kmc_hip_host_sort(kmers, dst, n_recs, (uint32)(sizeof(KMER_T) / 8), key_bytes, err);
	if (rc != 0)
		CCriticalErrorHandler::Inst().HandleCriticalError("Error: GPU sort failed (code " + std::to_string(rc) + "): " + err);
}

namespace RadulsSort
{
template <typename KMER_T> void RadixSortMSD_SSE2(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 n_threads, CMemoryPool *pmm_radix_buf) { kmc_hip_sort(kmers, tmp, n_recs, byte, n_threads, pmm_radix_buf); }
template <typename KMER_T> void RadixSortMSD_SSE41(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 n_threads, CMemoryPool *pmm_radix_buf) { kmc_hip_sort(kmers, tmp, n_recs, byte, n_threads, pmm_radix_buf); }
template <typename KMER_T> void RadixSortMSD_AVX(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 n_threads, CMemoryPool *pmm_radix_buf) { kmc_hip_sort(kmers, tmp, n_recs, byte, n_threads, pmm_radix_buf); }
template <typename KMER_T> void RadixSortMSD_AVX2(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 n_threads, CMemoryPool *pmm_radix_buf) { kmc_hip_sort(kmers, tmp, n_recs, byte, n_threads, pmm_radix_buf); }
template <typename KMER_T> void RadixSortMSD_NEON(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 n_threads, CMemoryPool *pmm_radix_buf) { kmc_hip_sort(kmers, tmp, n_recs, byte, n_threads, pmm_radix_buf); }
} // namespace RadulsSort

namespace RadixSort
{
template <typename KMER_T, unsigned SIZE> void RadixSortMSD(KMER_T *kmers, KMER_T *tmp, uint64 n_recs, uint32 byte, uint32 n_threads, CMemoryPool *pmm_radix_buf) { kmc_hip_sort(kmers, tmp, n_recs, byte, n_threads, pmm_radix_buf); }
} // namespace RadixSort

#endif
