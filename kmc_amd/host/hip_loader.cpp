/*
 * kmc_amd/host/hip_loader.cpp — the HIP bin engine of the stage-2 worker (kb_sorter_plugin.h): binds the C-ABI
 * of include/kmc_hip.h at run time with dlopen, the way a kmc_core maintainer would ship an optional GPU back end
 * (kmc stays buildable and runnable without ROCm; the GPU path is taken only when the library loads).
 *
 * Environment:
 *   KMC_HIP_LIB      path of libkmc_hip.so (default: <dir of the executable>/../../kmc_amd/libkmc_hip.so, then
 *                    plain "libkmc_hip.so" through the loader path)
 *   KMC_HIP_DEVICES  comma-separated HIP ordinals (default: every visible device, at most $KMC_HIP_MAX_DEVICES of them, minus those that fail to
 *                    initialise; HIP_VISIBLE_DEVICES narrows the set from outside); worker i uses device i % n_devices
 *   KMC_HIP_EAGER_INIT  "0": load the library at the first stage-2 worker instead of at program start
 *   KMC_HIP_VERBOSE  "1": print where the worker spent its time when the last engine is destroyed
 *   KMC_HIP_PINNED_POOL_MB  pinned slab for the bin images (host_pool.h; default 2048, 0 = none); KMC_HIP_POOL_WAIT_MS: how long a reader waits for a buffer of it (default 2000)
 *   KMC_HIP_SLOT_SLAB_MB  device memory reserved per stream slot while stage 1 runs (default 4096, 0 = none): kmc_hip_reserve_slot
 *   KMC_HIP_POOL_REGISTER  "1" (opt-in, round 6): the slab is ordinary memory on huge pages, registered with the runtime, instead of hipHostMalloc'ed
 *   KMC_HIP_TUNE_MALLOC  "1" (opt-in, round 6): re-execute the program once with the allocator tunables below (tune_allocator()); "2": the same without the huge-page heap
 * There is deliberately NO CPU fallback here: if the library or a GPU is missing the engine reports the error and
 * the worker raises it through CCriticalErrorHandler.
 */
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bin_engine.h"
#include "host_pool.h"

namespace {

struct Api {
	void *so = nullptr;
	int (*init)(const int *, int, kmc_hip_ctx **) = nullptr;
	void (*destroy)(kmc_hip_ctx *) = nullptr;
	const char *(*last_error)(kmc_hip_ctx *) = nullptr;
	int (*abi_version)(void) = nullptr;
	int (*submit)(kmc_hip_ctx *, int, int, const kmc_hip_bin_params *, const uint8_t *, uint64_t, uint64_t, const uint64_t *, uint64_t,
	              uint8_t *, uint64_t, uint64_t *) = nullptr;
	int (*wait)(kmc_hip_ctx *, int, int, uint64_t *, uint64_t *) = nullptr;
	int (*submit_bins)(kmc_hip_ctx *, int, int, const kmc_hip_bin_params *, const kmc_hip_host_bin *, uint32_t) = nullptr;
	int (*wait_bins)(kmc_hip_ctx *, int, int, uint64_t *, uint64_t *) = nullptr;
	int (*num_slots)(void) = nullptr;
	int (*sort_into)(kmc_hip_ctx *, int, const void *, void *, uint64_t, uint32_t, uint32_t) = nullptr;
	int (*host_alloc)(kmc_hip_ctx *, uint64_t, void **) = nullptr; /* optional: pinned buffers for the reader plug-in (KmcHostPool) */
	int (*host_free)(kmc_hip_ctx *, void *) = nullptr;
	int (*host_register)(kmc_hip_ctx *, void *, uint64_t) = nullptr;
	int n_slots = 1;
	std::mutex slot_mtx[64][16]; /* the C-ABI wants calls on one (device, slot) serialised; workers may outnumber slots */
	kmc_hip_ctx *ctx = nullptr;
	int n_dev = 0;
	std::string err;
};

Api g_api;
std::once_flag g_once;

/* KMC_HIP_VERBOSE=1: where did the worker spend its time? (printed when the last engine is destroyed) */
std::atomic<long long> g_ns_init{0}, g_ns_bins{0}, g_n_bins{0}, g_n_group_calls{0}, g_bytes_in{0}, g_bytes_out{0}, g_kmers{0}, g_engines{0};
inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::string exe_dir()
{
	char buf[4096];
	ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
	if (n <= 0)
		return ".";
	buf[n] = 0;
	std::string s(buf);
	size_t p = s.rfind('/');
	return p == std::string::npos ? "." : s.substr(0, p);
}

template <typename F> bool sym(void *so, const char *name, F &f, std::string &err)
{
	f = reinterpret_cast<F>(dlsym(so, name));
	if (!f) {
		err = std::string("libkmc_hip.so lacks symbol ") + name;
		return false;
	}
	return true;
}

void load_api_impl();
void load_api()
{
	const long long t0 = now_ns();
	load_api_impl();
	g_ns_init += now_ns() - t0;
}
void load_api_impl()
{
	Api &a = g_api;
	std::vector<std::string> cands;
	if (const char *e = getenv("KMC_HIP_LIB"))
		cands.push_back(e);
	cands.push_back(exe_dir() + "/../libkmc_hip.so");          /* kmc_amd/bin/<exe> -> kmc_amd/libkmc_hip.so */
	cands.push_back(exe_dir() + "/../../kmc_amd/libkmc_hip.so");
	cands.push_back("libkmc_hip.so");
	for (auto &c : cands) {
		a.so = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL);
		if (a.so)
			break;
		a.err = std::string("dlopen ") + c + ": " + dlerror();
	}
	if (!a.so)
		return;
	if (!sym(a.so, "kmc_hip_init", a.init, a.err) || !sym(a.so, "kmc_hip_destroy", a.destroy, a.err) ||
	    !sym(a.so, "kmc_hip_last_error", a.last_error, a.err) || !sym(a.so, "kmc_hip_abi_version", a.abi_version, a.err) ||
	    !sym(a.so, "kmc_hip_process_bin_submit", a.submit, a.err) || !sym(a.so, "kmc_hip_process_bin_wait", a.wait, a.err) ||
	    !sym(a.so, "kmc_hip_process_bins_submit", a.submit_bins, a.err) || !sym(a.so, "kmc_hip_process_bins_wait", a.wait_bins, a.err) ||
	    !sym(a.so, "kmc_hip_num_slots", a.num_slots, a.err) ||
	    !sym(a.so, "kmc_hip_sort_records_into", a.sort_into, a.err)) {
		a.so = nullptr;
		return;
	}
	if (a.abi_version() != KMC_HIP_ABI_VERSION) {
		a.err = "libkmc_hip.so ABI version mismatch";
		a.so = nullptr;
		return;
	}
	/* devices: $KMC_HIP_DEVICES ("0,1,..."), by default EVERY visible device — the stage-2 workers are spread over them (worker i drives device
	 * i mod n_dev), as the reference hands its bins to n_sorters threads (kmc.h:1576-1584, queues.h:2087-2128); HIP_VISIBLE_DEVICES narrows the set
	 * from outside */
	std::vector<int> devs;
	const char *e = getenv("KMC_HIP_DEVICES");
	if (e && *e) {
		std::string s = e;
		size_t pos = 0;
		while (pos <= s.size()) {
			size_t q = s.find(',', pos);
			if (q == std::string::npos)
				q = s.size();
			if (q > pos)
				devs.push_back(atoi(s.substr(pos, q - pos).c_str()));
			pos = q + 1;
		}
	} else {
		int (*device_count)() = nullptr;
		std::string ignore;
		const int n = sym(a.so, "kmc_hip_device_count", device_count, ignore) ? device_count() : 1;
		for (int i = 0; i < (n > 0 ? n : 1); ++i)
			devs.push_back(i);
	}
	if (devs.empty())
		devs.push_back(0);
	const bool chosen_by_user = e && *e;
	if (!chosen_by_user) { /* the default set may be capped from outside ($KMC_HIP_MAX_DEVICES): a run with few sorter threads leaves the other devices alone */
		const char *m = getenv("KMC_HIP_MAX_DEVICES");
		const int cap = m ? atoi(m) : 0;
		if (cap >= 1 && (size_t)cap < devs.size())
			devs.resize((size_t)cap);
	}
	int rc = a.init(devs.data(), (int)devs.size(), &a.ctx);
	if (rc && !chosen_by_user && devs.size() > 1) {
		/* one busy / faulty GPU of a shared node must not fail a run that device 0 alone would have carried (round 2's default): keep the devices that
		 * initialise on their own. A set named in $KMC_HIP_DEVICES is taken literally and fails loudly. */
		const std::string first_error = a.last_error(nullptr);
		std::vector<int> good;
		for (int d : devs) {
			kmc_hip_ctx *probe = nullptr;
			if (a.init(&d, 1, &probe) == 0) {
				good.push_back(d);
				a.destroy(probe);
			} else if (getenv("KMC_HIP_VERBOSE"))
				fprintf(stderr, "[kmc_hip] device %d left out: %s\n", d, a.last_error(nullptr));
		}
		if (!good.empty() && good.size() < devs.size()) {
			if (getenv("KMC_HIP_VERBOSE"))
				fprintf(stderr, "[kmc_hip] %zu of %zu visible devices in use (the full set failed: %s)\n", good.size(), devs.size(), first_error.c_str());
			devs = good;
			rc = a.init(devs.data(), (int)devs.size(), &a.ctx);
		}
	}
	if (rc) {
		a.err = std::string("kmc_hip_init failed: ") + a.last_error(nullptr);
		a.ctx = nullptr;
		return;
	}
	a.n_dev = (int)devs.size();
	{ /* the reader plug-in may read bin images straight into pinned memory of this library (host_pool.h): one slab, allocated here — on the background thread
	   * that brings the library up during stage 1 — so that stage 2 never waits for the runtime to pin anything */
		std::string ignore;
		const size_t want = KmcHostPool::wanted_bytes();
		bool have = false;
		/* opt-in: ordinary anonymous memory on huge pages, pinned by registration — registering is an order of magnitude cheaper than hipHostMalloc (profiles/r02/
		 * ubench_host.json: 126 against 7.7 GB/s) and, measured in round 6, a hipHostMalloc slab of 8 GB added 2 s to the process's exit. Never unmapped: the kernel takes
		 * it back at exit. NOT the default: the two sessions that ran the drop-in with this and the allocator tunables as defaults (r06o, r06s) each lost their GPU box
		 * (host memory, by the look of it) before a result came back; which of the two is at fault was not established */
		const char *reg = getenv("KMC_HIP_POOL_REGISTER");
		if (want && reg && reg[0] == '1' && sym(a.so, "kmc_hip_host_register", a.host_register, ignore)) {
			void *p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
			if (p != MAP_FAILED) {
				(void)madvise(p, want, MADV_HUGEPAGE);
				if (a.host_register(a.ctx, p, want) == 0) {
					KmcHostPool::inst().adopt(p, want, nullptr);
					have = true;
				} else
					munmap(p, want);
			}
		}
		if (want && !have && sym(a.so, "kmc_hip_host_alloc", a.host_alloc, ignore) && sym(a.so, "kmc_hip_host_free", a.host_free, ignore)) {
			void *p = nullptr;
			if (a.host_alloc(a.ctx, want, &p) == 0 && p) {
				KmcHostPool::inst().adopt(p, want, [](void *q) { (void)g_api.host_free(g_api.ctx, q); });
				have = true;
			}
		}
		if (want && !have && getenv("KMC_HIP_VERBOSE"))
			fprintf(stderr, "[kmc_hip] no pinned pool (%zu MB refused): bin images go through the arena\n", want >> 20);
	}
	a.n_slots = a.num_slots();
	if (a.n_slots > 16)
		a.n_slots = 16;
	if (a.n_slots < 1)
		a.n_slots = 1;
	{ /* device memory of the stream slots, ahead of time (include/kmc_hip.h kmc_hip_reserve_slot): one allocation per slot now — on this background thread, during stage 1 —
	   * instead of ~10 per worker at the moment stage 2 starts, when the reference's reader is unmapping bin parts and every allocation of the runtime waits for the
	   * process's mmap lock. $KMC_HIP_SLOT_SLAB_MB per slot (default 4096: a 48-84 M-k-mer bin of a 30 Gbp run wants 2.8-3.2 GB, and ONE buffer
	   * that does not fit and is allocated on demand costs its worker ~1 s at that moment — session w, 3 GB slabs: 17.9 s summed; 0 = off) */
		const char *e = getenv("KMC_HIP_SLOT_SLAB_MB");
		int (*backend_kind)(void) = nullptr;
		std::string ignore;
		/* a test build of the library (tests/hipemu: "device" memory is host memory, filled on allocation) reserves nothing unless told to */
		const bool test_build = sym(a.so, "kmc_hip_backend_kind", backend_kind, ignore) && backend_kind() != 0;
		const uint64_t mb = e ? strtoull(e, nullptr, 10) : (test_build ? 0 : 4096);
		int (*reserve)(kmc_hip_ctx *, int, int, uint64_t) = nullptr;
		if (mb && sym(a.so, "kmc_hip_reserve_slot", reserve, ignore)) {
			int ok = 0;
			for (int d = 0; d < a.n_dev; ++d)
				for (int sl = 0; sl < a.n_slots; ++sl)
					ok += reserve(a.ctx, d, sl, mb << 20) == 0;
			if (getenv("KMC_HIP_VERBOSE"))
				fprintf(stderr, "[kmc_hip] %d of %d stream slots have a slab of %llu MB\n", ok, a.n_dev * a.n_slots, (unsigned long long)mb);
		}
	}
}

void start_main_sampler(); /* below: KMC_HIP_SAMPLE_MAIN diagnostics */

struct HipEngine : KmcBinEngine {
	int dev, slot;
	std::string err;
	HipEngine(int dev, int slot) : dev(dev), slot(slot)
	{
		if (++g_engines == 1)
			start_main_sampler();
	}
	~HipEngine() override
	{
		if (--g_engines == 0 && getenv("KMC_HIP_VERBOSE"))
			fprintf(stderr, "[kmc_hip] init %.3f s; %lld bins (%lld calls with several bins), %.3f s inside the engine (sum over workers), %.1f MB in, %.1f MB out, %lld k-mers\n",
			        g_ns_init.load() * 1e-9, g_n_bins.load(), g_n_group_calls.load(), g_ns_bins.load() * 1e-9, g_bytes_in.load() / 1e6, g_bytes_out.load() / 1e6,
			        g_kmers.load());
		int (*hb_times)(double *) = nullptr;
		std::string ignore;
		double t[8];
		if (getenv("KMC_HIP_VERBOSE") && g_engines == 0 && g_api.so && sym(g_api.so, "kmc_hip_host_boundary_times", hb_times, ignore) && hb_times(t) == 0)
			fprintf(stderr, "[kmc_hip host boundary] %.0f calls (%.0f redo rounds); seconds summed over workers: pack starts + buffers %.3f, staging copy in %.3f, enqueue %.3f, "
			                "wait for the kernels %.3f, D2H of the results %.3f, staging copy out %.3f\n", t[6], t[7], t[0], t[1], t[2], t[3], t[4], t[5]);
	}
	int process_bin(const kmc_hip_bin_params &p, const uint8_t *sk, uint64_t size, uint64_t n_rec, const uint64_t *pack_bytes,
	                uint64_t n_packs, uint8_t *out, uint64_t cap, uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4]) override
	{
		if (!g_api.ctx) {
			err = g_api.err.empty() ? "HIP engine not initialised" : g_api.err;
			return KMC_HIP_EDEVICE;
		}
		/* Several stream slots per device (kmc_hip_num_slots): workers (one per stage-2 sorter thread, kmc.h:1576-1584) are spread over
		 * (device, slot) pairs; a pair is used by one worker at a time, so the copies of one bin overlap the kernels
		 * of another. */
		std::lock_guard<std::mutex> lck(g_api.slot_mtx[dev & 63][slot]);
		const long long t0 = now_ns();
		int rc = g_api.submit(g_api.ctx, dev, slot, &p, sk, size, n_rec, pack_bytes, n_packs, out, cap, lut);
		if (!rc)
			rc = g_api.wait(g_api.ctx, dev, slot, out_bytes, stats);
		if (rc)
			err = g_api.last_error(g_api.ctx);
		g_ns_bins += now_ns() - t0;
		++g_n_bins;
		g_bytes_in += (long long)size;
		g_bytes_out += (long long)*out_bytes;
		g_kmers += (long long)n_rec;
		return rc;
	}
	int process_bins(const kmc_hip_bin_params &p, const kmc_hip_host_bin *bins, uint32_t n_bins, uint64_t *out_bytes, uint64_t *stats) override
	{
		if (!g_api.ctx) {
			err = g_api.err.empty() ? "HIP engine not initialised" : g_api.err;
			return KMC_HIP_EDEVICE;
		}
		std::lock_guard<std::mutex> lck(g_api.slot_mtx[dev & 63][slot]);
		const long long t0 = now_ns();
		int rc = g_api.submit_bins(g_api.ctx, dev, slot, &p, bins, n_bins);
		if (!rc)
			rc = g_api.wait_bins(g_api.ctx, dev, slot, out_bytes, stats);
		if (rc)
			err = g_api.last_error(g_api.ctx);
		g_ns_bins += now_ns() - t0;
		g_n_bins += n_bins;
		++g_n_group_calls;
		for (uint32_t i = 0; i < n_bins; ++i) {
			g_bytes_in += (long long)bins[i].size;
			g_bytes_out += rc ? 0 : (long long)out_bytes[i];
			g_kmers += (long long)bins[i].n_rec;
		}
		return rc;
	}
	std::string last_error() override { return err; }
};

} // namespace

/* HIP start-up (dlopen of the ROCm runtime, context and stream creation) costs 0.13-0.35 s — as much as the GPU needs for
 * the whole stage 2 of a 2 Gbp input — and nothing in kmc_core runs before stage 2 that the worker could hook. So the
 * library is loaded by a background thread started when the program is loaded, i.e. during stage 1; the first worker only
 * waits for it (call_once). KMC_HIP_EAGER_INIT=0 restores the lazy behaviour (load at the first stage-2 worker). */
#ifdef __GLIBC__
/* The allocator under the REFERENCE's RAM-only pipeline (measured in round 6, profiles/r06/e2e_sweep_8gbp.jsonl). Stage 1 keeps every bin as a list of `new uchar[]`
 * parts (mem_disk_file.cpp:111-126), tens of megabytes each: mmap'ed chunks on 4 KB pages. Stage 2's CMemDiskFile::Read copies a part and deletes it
 * (mem_disk_file.cpp:84-100): an munmap of ~12 000 pages under the process's mmap lock, from every reader thread — and everything else that needs that lock waits: the
 * workers' page faults and the runtime's device allocations (16 workers spent 6.6 s in "pack starts + buffers" where 1.8 s suffice). With
 *   glibc.malloc.mmap_max=0, trim_threshold / top_pad large : large blocks come from the heap and go back to malloc's lists, not to the kernel (reader wall 0.66 -> 0.20 s,
 *                                                             "2nd stage" 1.26 -> 0.77 s on 8 Gbp; the pages are returned when the process exits instead)
 *   glibc.malloc.hugetlb=1                                  : the heap on transparent huge pages where the system offers them by madvise (stage 1's storer touches
 *                                                             512x fewer pages: "1st stage" 7.9 -> 6.3 s, process wall 9.7 -> 8.5 s)
 * Tunables are read when the process starts, so the program re-executes itself ONCE, before main() and before any thread exists, with $GLIBC_TUNABLES completed
 * (settings the user gave are kept); a failed exec just carries on. OPT-IN (KMC_HIP_TUNE_MALLOC=1): each tunable was measured on its own at 8 Gbp (session n), but the two
 * sessions that ran them together as the default, at 8 and 30 Gbp, lost their GPU boxes — most likely host memory: with -m512 KMC sizes its pools for 512 GB, a pool part that
 * is touched pins a whole 2 MB page once the heap is on huge pages, and nothing is ever returned to the kernel. Until that is understood the default is the round-5 behaviour. */
__attribute__((constructor(101))) static void tune_allocator(int /*argc*/, char **argv, char ** /*envp*/)
{
	const char *sw = getenv("KMC_HIP_TUNE_MALLOC");
	if (!sw || (sw[0] != '1' && sw[0] != '2') || getenv("KMC_HIP_TUNED") || !argv || !argv[0])
		return;
	const bool huge_pages = sw[0] == '1'; /* 2: without the huge-page heap — the subset that was measured on its own at 8 Gbp ("2nd stage" 1.26 -> 0.77 s) */
	const char *cur = getenv("GLIBC_TUNABLES");
	std::string t = cur ? cur : "";
	auto add = [&t](const char *name, const char *val) {
		if (t.find(name) != std::string::npos)
			return;
		if (!t.empty())
			t += ":";
		t += name;
		t += "=";
		t += val;
	};
	if (huge_pages)
		add("glibc.malloc.hugetlb", "1");
	add("glibc.malloc.mmap_max", "0");
	add("glibc.malloc.trim_threshold", "1099511627776");
	add("glibc.malloc.top_pad", "1073741824");
	setenv("GLIBC_TUNABLES", t.c_str(), 1);
	setenv("KMC_HIP_TUNED", "1", 1);
	execv("/proc/self/exe", argv);
	unsetenv("KMC_HIP_TUNED"); /* could not: untuned, as before */
}
#endif

namespace {
struct EagerInit {
	std::thread th;
	EagerInit()
	{
		const char *e = getenv("KMC_HIP_EAGER_INIT");
		if (!e || e[0] != '0')
			th = std::thread([] { std::call_once(g_once, load_api); });
	}
	~EagerInit()
	{
		if (th.joinable())
			th.join();
	}
} g_eager;

/* KMC_HIP_SAMPLE_MAIN=<ms> (diagnostics): where is the MAIN thread — the one that runs CKMC::Process — while nothing of this repo runs? A sampler thread
 * signals it every <ms> milliseconds, the handler keeps the return addresses (backtrace), and at exit the samples are printed with the second since the
 * program started, symbolised. Answers "where do the 0.17 s after the completer has closed its files go" (round-3 verdict) without touching kmc_core. */
struct MainSampler {
	static constexpr int MAX_SAMPLES = 4096, DEPTH = 10;
	static void *frames[MAX_SAMPLES][DEPTH];
	static int depth[MAX_SAMPLES];
	static long long when_ns[MAX_SAMPLES];
	static std::atomic<int> n;
	static long long t0;
	pthread_t main_thread;
	std::thread th;
	std::atomic<bool> stop{false};
	static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
	static void on_signal(int)
	{
		const int i = n.fetch_add(1);
		if (i < MAX_SAMPLES) {
			when_ns[i] = now_ns();
			depth[i] = backtrace(frames[i], DEPTH);
		}
	}
	MainSampler()
	{
		main_thread = pthread_self(); /* static initialisers run on the main thread */
		t0 = now_ns();
	}
	/* called when the first stage-2 worker is constructed (start-up — the dynamic loader, the HIP runtime coming up on another thread — is over by then) */
	void start()
	{
		const char *e = getenv("KMC_HIP_SAMPLE_MAIN");
		const int ms = e ? atoi(e) : 0;
		if (ms < 1 || th.joinable())
			return;
		void *warm[4];
		(void)backtrace(warm, 4); /* loads libgcc outside the handler */
		signal(SIGUSR2, on_signal);
		th = std::thread([this, ms] {
			while (!stop.load()) {
				std::this_thread::sleep_for(std::chrono::milliseconds(ms));
				pthread_kill(main_thread, SIGUSR2);
			}
		});
	}
	~MainSampler()
	{
		if (!th.joinable())
			return;
		stop.store(true);
		th.join();
		const int m = std::min(n.load(), (int)MAX_SAMPLES);
		fprintf(stderr, "[kmc_hip main-thread samples] %d samples (program start at %.3f s of the steady clock); second of that clock: frames\n", m, t0 * 1e-9);
		for (int i = 0; i < m; ++i) {
			fprintf(stderr, "  %.3f:", when_ns[i] * 1e-9);
			char **sym = backtrace_symbols(frames[i], depth[i]);
			for (int d = 2; d < depth[i] && d < 8; ++d) { /* 0-1: the handler and the signal trampoline */
				std::string sname = sym ? sym[d] : "?";
				const size_t a = sname.find('('), b = sname.find('+', a == std::string::npos ? 0 : a);
				if (a != std::string::npos && b != std::string::npos && b > a + 1)
					sname = sname.substr(a + 1, b - a - 1);
				fprintf(stderr, " %s |", sname.substr(0, 60).c_str());
			}
			fprintf(stderr, "\n");
			free(sym);
		}
	}
} g_main_sampler;
void *MainSampler::frames[MainSampler::MAX_SAMPLES][MainSampler::DEPTH];
int MainSampler::depth[MainSampler::MAX_SAMPLES];
long long MainSampler::when_ns[MainSampler::MAX_SAMPLES];
std::atomic<int> MainSampler::n{0};
long long MainSampler::t0 = 0;
void start_main_sampler() { g_main_sampler.start(); }
} // namespace

/* the narrow boundary (hip_sort_function.h): one GPU sort on behalf of a reference sorter thread. Threads are spread over the
 * configured devices; on one device host sorts take turns (the library serialises them on the device's staging arrays). */
int kmc_hip_host_sort(const void *recs, void *dst, uint64_t n, uint32_t words, uint32_t key_bytes, std::string &err)
{
	std::call_once(g_once, load_api);
	if (!g_api.ctx) {
		err = g_api.err.empty() ? "HIP engine not initialised" : g_api.err;
		return KMC_HIP_EDEVICE;
	}
	static std::atomic<unsigned> next_thread{0};
	thread_local unsigned my_idx = next_thread++;
	const int dev = (int)(my_idx % (unsigned)(g_api.n_dev > 0 ? g_api.n_dev : 1));
	int rc = g_api.sort_into(g_api.ctx, dev, recs, dst, n, words, key_bytes);
	if (rc)
		err = g_api.last_error(g_api.ctx);
	return rc;
}

KmcBinEngine *kmc_make_bin_engine(int worker_idx, int /*n_workers*/)
{
	std::call_once(g_once, load_api);
	const int n = g_api.n_dev > 0 ? g_api.n_dev : 1;
	return new HipEngine(worker_idx % n, (worker_idx / n) % g_api.n_slots);
}

/* for other engines bound to the same library (hip_split_loader.cpp): the loaded handle and the one context of the process */
bool kmc_hip_loader_handles(void *&so, kmc_hip_ctx *&ctx, int &n_dev, int &n_slots, std::string &err)
{
	std::call_once(g_once, load_api);
	so = g_api.so;
	ctx = g_api.ctx;
	n_dev = g_api.n_dev > 0 ? g_api.n_dev : 1;
	n_slots = g_api.n_slots;
	if (!g_api.ctx)
		err = g_api.err.empty() ? "HIP engine not initialised" : g_api.err;
	return g_api.ctx != nullptr;
}
