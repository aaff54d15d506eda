/*
 * kmc_amd/csrc/synth_bins.cpp — deterministic synthetic stage-2 INPUT generator (libkmc_synth.so).
 *
 * Produces what KMC's stage 1 would hand to stage 2 — bin images of super-k-mers in the on-disk format
 * (kmc_core/kb_collector.cpp:57-71) plus expander-pack byte lengths (kb_collector.cpp:36-42, <= 4096
 * super-k-mers per pack) — directly from the read model of SURVEY.md §8d (uniform random genome, L-bp reads at
 * uniform starts, per-base substitution probability `err`, random strand), without FASTQ text or the reference.
 * Stage 1 itself is out of scope; this is workload plumbing for bench.py and the large-size
 * property tests. Super-k-mers are cut where the minimizer (smallest canonical m-mer of the k-mer, m = sig_len)
 * changes, as in kmc_core/splitter.cpp:557-672, without KMC's signature blacklist (irrelevant to stage 2);
 * bin = mix(signature) % n_bins. Output is independent of the number of threads.
 */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}
struct Rng { /* counter-based: stream (seed, read) is independent of scheduling */
	uint64_t key, ctr;
	Rng(uint64_t seed, uint64_t stream) : key(mix64(seed ^ mix64(stream))), ctr(0) {}
	uint64_t next() { return mix64(key + 0x632BE59BD9B4E019ull * ++ctr); }
};

constexpr uint64_t READS_PER_CHUNK = 1 << 15;
constexpr uint32_t PACK_SUPERKMERS = 4096; /* kb_collector.h:46 */

/* Two passes over the reads of a chunk, no per-chunk buffers (the first version kept a vector per (chunk, bin): 3 M vectors and
 * 60 GB of allocator traffic for the 30 Gbp workload, which scaled to 256 threads very badly): pass 1 COUNTS what every
 * (chunk, bin) cell will hold, the cells' offsets inside the final images follow from prefix sums, pass 2 regenerates the
 * reads (they are a pure function of (seed, read)) and WRITES the super-k-mers in place. */
struct Cell {
	uint64_t bytes = 0, n_rec = 0;
	uint32_t n_super = 0;
};

struct CountSink {
	Cell *cells; /* [n_bins] of this chunk */
	void put(uint32_t bin, const uint8_t *, uint32_t n, uint32_t k)
	{
		Cell &c = cells[bin];
		c.bytes += 1 + (n + 3) / 4;
		c.n_rec += n - k + 1;
		++c.n_super;
	}
};

struct Cursor {
	uint8_t *p = nullptr;
	uint64_t *pk = nullptr;
	uint64_t cur_pack_bytes = 0;
	uint32_t cur_pack_sk = 0;
};

struct WriteSink {
	Cursor *cur; /* [n_bins], positioned at this chunk's cells */
	void put(uint32_t bin, const uint8_t *sym, uint32_t n, uint32_t k)
	{
		Cursor &c = cur[bin];
		if (c.cur_pack_sk >= PACK_SUPERKMERS) {
			*c.pk++ = c.cur_pack_bytes;
			c.cur_pack_bytes = 0;
			c.cur_pack_sk = 0;
		}
		const uint32_t nb = 1 + (n + 3) / 4;
		uint8_t *p = c.p;
		p[0] = (uint8_t)(n - k);
		for (uint32_t i = 1; i < nb; ++i)
			p[i] = 0;
		for (uint32_t i = 0; i < n; ++i)
			p[1 + (i >> 2)] |= (uint8_t)(sym[i] << (6 - 2 * (i & 3)));
		c.p += nb;
		c.cur_pack_bytes += nb;
		++c.cur_pack_sk;
	}
	void close_packs(uint32_t n_bins)
	{
		for (uint32_t b = 0; b < n_bins; ++b)
			if (cur[b].cur_pack_bytes) { /* every chunk closes its own packs */
				*cur[b].pk++ = cur[b].cur_pack_bytes;
				cur[b].cur_pack_bytes = 0;
				cur[b].cur_pack_sk = 0;
			}
	}
};

/* read r of the model: window at a uniform start, per-base substitution with probability err, random strand */
inline void make_read(uint64_t seed, const std::vector<uint8_t> &genome, uint64_t r, uint32_t L, uint64_t err_thr, std::vector<uint8_t> &rd)
{
	const uint64_t G = genome.size();
	Rng g(seed, r);
	const uint64_t st = g.next() % (G - L + 1);
	const bool flip = g.next() & 1;
	for (uint32_t i = 0; i < L; ++i) {
		uint8_t s = genome[st + i];
		if (g.next() < err_thr)
			s = (uint8_t)((s + 1 + g.next() % 3) & 3);
		rd[i] = s;
	}
	if (flip) {
		std::reverse(rd.begin(), rd.end());
		for (auto &s : rd)
			s = 3 - s;
	}
}

void make_genome(uint64_t seed, uint64_t genome_len, int n_threads, std::vector<uint8_t> &genome)
{
	genome.resize(genome_len);
	std::vector<std::thread> th;
	const uint64_t per = (genome_len + n_threads - 1) / n_threads;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([&, t] {
			const uint64_t a = t * per, b = std::min(genome_len, a + per);
			for (uint64_t i = a; i < b; ++i)
				genome[i] = (uint8_t)(mix64(seed * 0x100000001B3ull + i) & 3);
		});
	for (auto &x : th)
		x.join();
	/* $KMC_SYNTH_REPEATS = "unit:copies[:per_mille]" plants `copies` copies of the genome's first `unit` bases at pseudo-random places, each copy with
	 * per_mille/1000 of its bases substituted (default 0) — the repeat families a real genome has and a uniform random one lacks: their k-mers occur
	 * copies x coverage times, far beyond what a tile of the LDS sort holds (bench.py's skew leg, tests). Deterministic in (seed, genome_len).
	 * Round 5: a comma-separated SPECTRUM of families — family j's unit is the `unit` bases behind the units of the families before it (family 0: as ever, so
	 * the one-family leg of round 4 is unchanged) — and "H<len>": one homopolymer run of `len` A's. E.g. "300:100000:120,6000:5000:20,171:100000:20,H20000":
	 * an Alu-like family, an L1-like one, a satellite and a poly-A run. */
	if (const char *e = getenv("KMC_SYNTH_REPEATS")) {
		uint64_t unit_off = 0;
		unsigned fam = 0;
		for (const char *q = e; *q; ++fam) {
			unsigned long long unit = 0, copies = 0, pm = 0;
			if (*q == 'H') {
				const unsigned long long len = strtoull(q + 1, nullptr, 10);
				if (len >= 1 && len * 2 <= genome_len) {
					const uint64_t at = mix64(seed ^ 0xA11A11ull ^ fam) % (genome_len - len + 1);
					for (uint64_t i = 0; i < len; ++i)
						genome[at + i] = 0;
				}
			} else if (sscanf(q, "%llu:%llu:%llu", &unit, &copies, &pm) >= 2 && unit >= 1 && unit_off + unit * 2 <= genome_len) {
				const std::vector<uint8_t> u(genome.begin() + (ptrdiff_t)unit_off, genome.begin() + (ptrdiff_t)(unit_off + unit));
				const uint64_t fs = seed ^ ((uint64_t)fam * 0xD1B54A32D192ED03ull); /* family 0: the seed itself */
				for (unsigned long long c = 0; c < copies; ++c) {
					const uint64_t at = unit_off + unit + mix64(fs ^ (0xC0FFEEull + c * 0x9E3779B97F4A7C15ull)) % (genome_len - unit_off - 2 * unit + 1);
					for (uint64_t i = 0; i < unit; ++i) {
						uint8_t b = u[i];
						if (pm && mix64(fs + 77 * c + 1315423911ull * i) % 1000 < pm)
							b = (uint8_t)((b + 1 + mix64(fs + c + i) % 3) & 3);
						genome[at + i] = b;
					}
				}
				unit_off += unit;
			}
			while (*q && *q != ',')
				++q;
			if (*q == ',')
				++q;
		}
	}
}

template <typename Sink>
void gen_chunk(uint64_t seed, const std::vector<uint8_t> &genome, uint64_t r0, uint64_t r1, uint32_t L, double err, uint32_t k, uint32_t m,
               uint32_t n_bins, Sink &out)
{
	const uint64_t err_thr = (uint64_t)(err * 18446744073709551615.0);
	std::vector<uint8_t> rd(L);
	std::vector<uint32_t> mm(L), mn(L);
	const uint32_t n_mm = L - m + 1, n_k = L - k + 1, w = k - m + 1;
	const uint32_t mmask = (m < 16) ? ((1u << (2 * m)) - 1) : 0xFFFFFFFFu;
	std::vector<uint32_t> dq(L);
	for (uint64_t r = r0; r < r1; ++r) {
		make_read(seed, genome, r, L, err_thr, rd);
		/* canonical m-mers */
		uint32_t f = 0, rc = 0;
		for (uint32_t i = 0; i < L; ++i) {
			f = ((f << 2) | rd[i]) & mmask;
			rc = (rc >> 2) | ((uint32_t)(3 - rd[i]) << (2 * (m - 1)));
			if (i + 1 >= m)
				mm[i + 1 - m] = f < rc ? f : rc;
		}
		/* sliding-window minimum over w consecutive m-mers: leftmost smallest value */
		uint32_t head = 0, tail = 0;
		for (uint32_t i = 0; i < n_mm; ++i) {
			while (tail > head && mm[dq[tail - 1]] > mm[i])
				--tail;
			dq[tail++] = i;
			if (i + 1 >= w) {
				const uint32_t kpos = i + 1 - w;
				while (dq[head] < kpos)
					++head;
				mn[kpos] = mm[dq[head]];
			}
		}
		/* cut into super-k-mers of equal signature */
		uint32_t start = 0;
		for (uint32_t i = 1; i <= n_k; ++i) {
			if (i == n_k || mn[i] != mn[start] || i - start >= 256) {
				const uint32_t bin = (uint32_t)(mix64(mn[start]) % n_bins);
				out.put(bin, rd.data() + start, (i - start) + k - 1, k);
				start = i;
			}
		}
	}
}

struct Result {
	std::vector<uint8_t *> image;
	std::vector<uint64_t> size, n_rec, n_super, n_packs;
	std::vector<uint64_t *> packs;
};

} // namespace

extern "C" {

void kmc_synth_free(void *handle);

/* Generate `n_bins` bin images from reads [read_begin, read_end) of the model (read r is a pure function of (seed, r), and every
 * READS_PER_CHUNK-aligned chunk of reads closes its own expander packs, so images made from chunk-aligned sub-ranges concatenate,
 * bin by bin, to exactly the image of the whole range: bench.py --gpus N generates 1/N of the reads per rank). Arrays of length
 * n_bins are returned through the out_* pointers (malloc'ed; free everything with kmc_synth_free(handle)). Returns an opaque
 * handle or NULL. */
void *kmc_synth_bins_range(uint64_t seed, uint64_t genome_len, uint64_t read_begin, uint64_t read_end, uint32_t read_len, double err, uint32_t k,
                           uint32_t sig_len, uint32_t n_bins, int n_threads, uint8_t ***out_images, uint64_t **out_sizes, uint64_t **out_n_rec,
                           uint64_t ***out_pack_bytes, uint64_t **out_n_packs, uint64_t **out_n_super)
{
	if (k < sig_len || sig_len < 1 || sig_len > 15 || read_len < k || genome_len < read_len || n_bins < 1 || k > 256 || read_end < read_begin ||
	    read_begin % READS_PER_CHUNK)
		return nullptr;
	if (n_threads < 1)
		n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
	std::vector<uint8_t> genome;
	make_genome(seed, genome_len, n_threads, genome);
	const uint64_t n_reads = read_end - read_begin;
	const uint64_t n_chunks = (n_reads + READS_PER_CHUNK - 1) / READS_PER_CHUNK;
	auto chunk_range = [&](uint64_t c, uint64_t &a, uint64_t &b) {
		a = read_begin + c * READS_PER_CHUNK;
		b = read_begin + std::min(n_reads, (c + 1) * READS_PER_CHUNK);
	};
	auto run_threads = [&](auto body) {
		std::vector<std::thread> th;
		for (int t = 0; t < n_threads; ++t)
			th.emplace_back([&, t] { body(t); });
		for (auto &x : th)
			x.join();
	};
	const bool verbose = getenv("KMC_SYNTH_VERBOSE") != nullptr;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	/* pass 1: count */
	std::vector<Cell> cells((size_t)n_chunks * n_bins);
	run_threads([&](int t) {
		for (uint64_t c = t; c < n_chunks; c += n_threads) {
			uint64_t a, b;
			chunk_range(c, a, b);
			CountSink sink{cells.data() + (size_t)c * n_bins};
			gen_chunk(seed, genome, a, b, read_len, err, k, sig_len, n_bins, sink);
		}
	});
	const double t1 = now();
	Result *R = new Result();
	R->image.assign(n_bins, nullptr);
	R->packs.assign(n_bins, nullptr);
	R->size.assign(n_bins, 0);
	R->n_rec.assign(n_bins, 0);
	R->n_super.assign(n_bins, 0);
	R->n_packs.assign(n_bins, 0);
	/* offsets of every cell inside its bin's image / pack list (exclusive prefix over chunks), kept in the cells */
	std::vector<uint64_t> pack_off((size_t)n_chunks * n_bins);
	bool ok = true;
	for (uint32_t b = 0; b < n_bins; ++b) {
		uint64_t sz = 0, np = 0;
		for (uint64_t c = 0; c < n_chunks; ++c) {
			Cell &cl = cells[(size_t)c * n_bins + b];
			const uint64_t bytes = cl.bytes;
			R->n_rec[b] += cl.n_rec;
			R->n_super[b] += cl.n_super;
			cl.bytes = sz; /* now: byte offset of the cell */
			pack_off[(size_t)c * n_bins + b] = np;
			sz += bytes;
			np += (cl.n_super + PACK_SUPERKMERS - 1) / PACK_SUPERKMERS;
		}
		R->size[b] = sz;
		R->n_packs[b] = np;
		const size_t want = sz + 256;
		if (want >= (8u << 20)) { /* big images: 2 MiB-aligned and advised huge, 256 threads first-touching 4 KiB pages do not scale */
			const size_t al = 2u << 20, rounded = (want + al - 1) / al * al;
			R->image[b] = (uint8_t *)aligned_alloc(al, rounded);
			if (R->image[b])
				(void)madvise(R->image[b], rounded, MADV_HUGEPAGE);
		} else
			R->image[b] = (uint8_t *)malloc(want);
		R->packs[b] = (uint64_t *)malloc((np + 1) * 8);
		if (!R->image[b] || !R->packs[b])
			ok = false;
		else
			memset(R->image[b] + sz, 0, 256);
	}
	if (!ok) {
		kmc_synth_free(R);
		return nullptr;
	}
	/* pass 2: write in place */
	const double t2 = now();
	run_threads([&](int t) {
		std::vector<Cursor> cur(n_bins);
		for (uint64_t c = t; c < n_chunks; c += n_threads) {
			uint64_t a, b;
			chunk_range(c, a, b);
			for (uint32_t bi = 0; bi < n_bins; ++bi) {
				cur[bi].p = R->image[bi] + cells[(size_t)c * n_bins + bi].bytes;
				cur[bi].pk = R->packs[bi] + pack_off[(size_t)c * n_bins + bi];
				cur[bi].cur_pack_bytes = 0;
				cur[bi].cur_pack_sk = 0;
			}
			WriteSink sink{cur.data()};
			gen_chunk(seed, genome, a, b, read_len, err, k, sig_len, n_bins, sink);
			sink.close_packs(n_bins);
		}
	});
	if (verbose)
		fprintf(stderr, "[kmc_synth] %d threads, %llu reads, %u bins: count %.2f s, allocate %.2f s, write %.2f s\n", n_threads,
		        (unsigned long long)n_reads, n_bins, t1 - t0, t2 - t1, now() - t2);
	*out_images = R->image.data();
	*out_sizes = R->size.data();
	*out_n_rec = R->n_rec.data();
	*out_pack_bytes = R->packs.data();
	*out_n_packs = R->n_packs.data();
	if (out_n_super)
		*out_n_super = R->n_super.data();
	return R;
}

void *kmc_synth_bins(uint64_t seed, uint64_t genome_len, uint64_t n_reads, uint32_t read_len, double err, uint32_t k, uint32_t sig_len,
                     uint32_t n_bins, int n_threads, uint8_t ***out_images, uint64_t **out_sizes, uint64_t **out_n_rec,
                     uint64_t ***out_pack_bytes, uint64_t **out_n_packs, uint64_t **out_n_super)
{
	return kmc_synth_bins_range(seed, genome_len, 0, n_reads, read_len, err, k, sig_len, n_bins, n_threads, out_images, out_sizes, out_n_rec,
	                            out_pack_bytes, out_n_packs, out_n_super);
}

uint64_t kmc_synth_chunk_reads(void) { return READS_PER_CHUNK; }

/* The SAME reads [read_begin, read_end) as 4-line FASTQ ("@r<id>", bases, "+", quality 'I'), so that the reference kmc can be run on
 * exactly the k-mer multiset the synthetic bins hold (bench.py compares its tallies with the GPU's). Returns bytes written, 0 on error. */
uint64_t kmc_synth_fastq(uint64_t seed, uint64_t genome_len, uint64_t read_begin, uint64_t read_end, uint32_t read_len, double err,
                         const char *path, int n_threads)
{
	if (genome_len < read_len || read_end < read_begin || !path)
		return 0;
	if (n_threads < 1)
		n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
	std::vector<uint8_t> genome;
	make_genome(seed, genome_len, n_threads, genome);
	FILE *f = fopen(path, "wb");
	if (!f)
		return 0;
	const uint64_t err_thr = (uint64_t)(err * 18446744073709551615.0);
	const uint32_t L = read_len;
	const uint64_t rec = 1 + 1 + 12 + 1 + L + 1 + 1 + 1 + L + 1; /* "@r" + 12 digits + "\n" + bases + "\n+\n" + qualities + "\n" */
	const uint64_t BATCH = 1 << 20; /* reads per write */
	std::vector<char> buf(BATCH * rec);
	uint64_t total = 0;
	bool ok = true;
	for (uint64_t b0 = read_begin; b0 < read_end && ok; b0 += BATCH) {
		const uint64_t b1 = std::min(read_end, b0 + BATCH), nb = b1 - b0;
		std::vector<std::thread> th;
		const uint64_t per = (nb + n_threads - 1) / n_threads;
		for (int t = 0; t < n_threads; ++t)
			th.emplace_back([&, t] {
				std::vector<uint8_t> rd(L);
				const uint64_t a = b0 + t * per, e = std::min(b1, a + per);
				for (uint64_t r = a; r < e; ++r) {
					make_read(seed, genome, r, L, err_thr, rd);
					char *p = buf.data() + (r - b0) * rec;
					*p++ = '@';
					*p++ = 'r';
					uint64_t id = r;
					for (int d = 11; d >= 0; --d) {
						p[d] = (char)('0' + id % 10);
						id /= 10;
					}
					p += 12;
					*p++ = '\n';
					for (uint32_t i = 0; i < L; ++i)
						*p++ = "ACGT"[rd[i]];
					*p++ = '\n';
					*p++ = '+';
					*p++ = '\n';
					memset(p, 'I', L);
					p += L;
					*p++ = '\n';
				}
			});
		for (auto &x : th)
			x.join();
		ok = fwrite(buf.data(), 1, nb * rec, f) == nb * rec;
		total += nb * rec;
	}
	ok = (fclose(f) == 0) && ok;
	return ok ? total : 0;
}

void kmc_synth_free(void *handle)
{
	Result *R = (Result *)handle;
	if (!R)
		return;
	for (auto p : R->image)
		free(p);
	for (auto p : R->packs)
		free(p);
	delete R;
}

} /* extern "C" */
