// tools/ubench_scatter.hip — what HBM bandwidth does MI355X give a radix-scatter-shaped access pattern?
// Reads 8-byte records coalesced and writes them in runs of R records to pseudo-randomly permuted run slots
// (R*8-byte contiguous bursts, each from one wave/workgroup) — the write pattern of one LSD pass with 256 buckets.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_scatter.hip -o gpurun_out/ubench_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;

template <int ITEMS>
__global__ void __launch_bounds__(512) k_scatter(const u64* __restrict__ in, u64* __restrict__ out, u64 n, u64 run, u64 nruns_mask, u64 A)
{
	const u64 base = (u64)blockIdx.x * (512 * ITEMS);
	u64 v[ITEMS];
#pragma unroll
	for (int i = 0; i < ITEMS; ++i) v[i] = in[base + i * 512 + threadIdx.x];
#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const u64 e = base + i * 512 + threadIdx.x;
		const u64 r = e / run, o = e % run;
		const u64 pr = (r * A + 12345) & nruns_mask;
		out[pr * run + o] = v[i];
	}
}
// Two sub-tiles per workgroup, written one after the other; run r of sub-tile 0 and run r of sub-tile 1 land in ADJACENT
// slots, so the cache line they share is written twice by the same CU a few microseconds apart (does L2 merge them?).
template <int ITEMS>
__global__ void __launch_bounds__(512) k_scatter_pair(const u64* __restrict__ in, u64* __restrict__ out, u64 n, u64 run, u64 npairs_mask, u64 A)
{
	const u64 tile = 512 * ITEMS;
	const u64 runs_per_tile = tile / run;
	for (int sub = 0; sub < 2; ++sub) {
		const u64 base = ((u64)blockIdx.x * 2 + sub) * tile;
		u64 v[ITEMS];
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) v[i] = in[base + i * 512 + threadIdx.x];
#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const u64 el = i * 512 + threadIdx.x;
			const u64 r = el / run, o = el % run;
			const u64 pair = ((u64)blockIdx.x * runs_per_tile + r) * A & npairs_mask;
			out[(pair * 2 + sub) * run + o] = v[i];
		}
		__syncthreads();
	}
}
__global__ void __launch_bounds__(512) k_copy16(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out, u64 n2)
{
	for (u64 i = (u64)blockIdx.x * 512 + threadIdx.x; i < n2; i += (u64)gridDim.x * 512) out[i] = in[i];
}

int main()
{
	const u64 n = 1ull << 29; // 4 GiB of records in, 4 GiB out
	u64 *in, *out;
	hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
	hipMemset(in, 1, n * 8); hipMemset(out, 0, n * 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto timeit = [&](auto f, const char* name, double bytes) {
		f(); hipDeviceSynchronize();
		float best = 1e9;
		for (int it = 0; it < 5; ++it) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
		printf("%-28s %8.3f ms  %8.1f GB/s (read+write)\n", name, best, bytes / best / 1e6);
	};
	timeit([&] { k_copy16<<<256 * 8, 512>>>((const ulonglong2*)in, (ulonglong2*)out, n / 2); }, "copy 16B/lane grid-stride", 2.0 * n * 8);
	for (u64 run : {8ull, 16ull, 32ull, 64ull, 128ull, 256ull, 512ull, 4096ull}) {
		const u64 nruns = n / run;
		char name[64]; snprintf(name, sizeof name, "scatter runs of %5llu B", run * 8);
		timeit([&] { k_scatter<16><<<(unsigned)(n / (512 * 16)), 512>>>(in, out, n, run, nruns - 1, 0x9E3779B1ull); }, name, 2.0 * n * 8);
	}
	// misaligned runs: shift the whole output by 24 bytes so every run straddles cache lines like real bucket boundaries
	for (u64 run : {32ull, 40ull, 64ull, 80ull, 96ull, 128ull, 160ull, 256ull, 512ull}) { /* 40 = k_onesweep<1>'s 320-byte runs; 80 / 160: tiles of 20 K / 40 K records */
		const u64 nruns = n / run - 1;
		u64 m = 1; while (m * 2 <= nruns) m *= 2;
		char name[64]; snprintf(name, sizeof name, "scatter %5llu B, +24 B skew", run * 8);
		timeit([&] { k_scatter<16><<<(unsigned)(n / (512 * 16)), 512>>>(in, out + 3, n, run, m - 1, 0x9E3779B1ull); }, name, 2.0 * n * 8);
	}
	for (u64 run : {32ull, 64ull}) {
		const u64 npairs = n / run / 2 - 1;
		u64 m = 1; while (m * 2 <= npairs) m *= 2;
		char name[64]; snprintf(name, sizeof name, "pair-adjacent %4llu B +24 B", run * 8);
		timeit([&] { k_scatter_pair<16><<<(unsigned)(n / (512 * 16) / 2), 512>>>(in, out + 3, n, run, m - 1, 0x9E3779B1ull); }, name, 2.0 * n * 8);
	}
	// working sets that fit the 256 MB Infinity Cache: records in + out = 2 * nn * 8 bytes
	for (u64 nn : {1ull << 21, 1ull << 22, 1ull << 23, 1ull << 24, 1ull << 25, 1ull << 26}) {
		for (u64 run : {32ull}) {
			const u64 nruns = nn / run - 1;
			u64 m = 1; while (m * 2 <= nruns) m *= 2;
			char name[64]; snprintf(name, sizeof name, "ws %4llu MB, 256 B +24 B", 2 * nn * 8 >> 20);
			auto f = [&] { for (int rep = 0; rep < 8; ++rep) k_scatter<16><<<(unsigned)(nn / (512 * 16)), 512>>>(in, out + 3, nn, run, m - 1, 0x9E3779B1ull); };
			timeit(f, name, 8 * 2.0 * nn * 8);
		}
	}
	return 0;
}
