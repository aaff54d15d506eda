// tests/host/kmer_ops_host.cpp — host build of kmc_amd/csrc/kmer_ops.h (the arithmetic every kernel uses),
// exported through a tiny C interface so the CPU test-suite can compare it with the oracle without a GPU.
#include "../../kmc_amd/csrc/kmer_ops.h"
#include <stdint.h>

template <int SIZE>
static uint64_t expand_t(uint32_t k, int both, const uint8_t* data, uint64_t size, uint64_t* out)
{
	uint64_t pos = 0, n = 0;
	while (pos < size) {
		uint32_t e = data[pos];
		for (uint32_t off = 0; off <= e; ++off) {
			kmc_u64 v[SIZE];
			kmc_canonical_at<SIZE>(data + pos + 1, off, k, both != 0, v);
			for (int w = 0; w < SIZE; ++w) out[n * SIZE + w] = v[w];
			++n;
		}
		pos += 1 + (k + e + 3) / 4;
	}
	return n;
}

template <int SIZE>
static void emit_t(const uint64_t* recs, uint64_t n, const uint32_t* counts, uint32_t sbytes, uint32_t cbytes, int kff, uint8_t* out,
                   uint32_t k, uint32_t p, uint64_t* prefixes)
{
	for (uint64_t i = 0; i < n; ++i) {
		kmc_u64 v[SIZE];
		for (int w = 0; w < SIZE; ++w) v[w] = recs[i * SIZE + w];
		kmc_emit_record<SIZE>(out + i * (sbytes + cbytes), v, counts[i], sbytes, cbytes, kff != 0);
		prefixes[i] = p ? kmc_remove_suffix<SIZE>(v, 2 * (k - p)) : 0;
	}
}

extern "C" uint64_t kmer_ops_expand(uint32_t k, int both, const uint8_t* data, uint64_t size, uint64_t* out)
{
	switch ((k + 31) / 32) {
	case 1: return expand_t<1>(k, both, data, size, out);
	case 2: return expand_t<2>(k, both, data, size, out);
	case 3: return expand_t<3>(k, both, data, size, out);
	case 4: return expand_t<4>(k, both, data, size, out);
	case 5: return expand_t<5>(k, both, data, size, out);
	case 6: return expand_t<6>(k, both, data, size, out);
	case 7: return expand_t<7>(k, both, data, size, out);
	default: return expand_t<8>(k, both, data, size, out);
	}
}

extern "C" void kmer_ops_emit(uint32_t k, uint32_t p, const uint64_t* recs, uint64_t n, const uint32_t* counts, uint32_t sbytes,
                              uint32_t cbytes, int kff, uint8_t* out, uint64_t* prefixes)
{
	switch ((k + 31) / 32) {
	case 1: emit_t<1>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	case 2: emit_t<2>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	case 3: emit_t<3>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	case 4: emit_t<4>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	case 5: emit_t<5>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	case 6: emit_t<6>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	case 7: emit_t<7>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	default: emit_t<8>(recs, n, counts, sbytes, cbytes, kff, out, k, p, prefixes); break;
	}
}
extern "C" uint32_t kmer_ops_counter_bytes(uint64_t cx, uint64_t cs) { return kmc_counter_bytes(cx, cs); }
extern "C" uint32_t kmer_ops_suffix_bytes(uint32_t k, uint32_t p) { return kmc_suffix_bytes(k, p); }
