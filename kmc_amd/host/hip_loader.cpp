/*
 * kmc_amd/host/hip_loader.cpp — the HIP bin engine of the stage-2 worker (kb_sorter_plugin.h): binds the C-ABI
 * of include/kmc_hip.h at run time with dlopen, the way a kmc_core maintainer would ship an optional GPU back end
 * (kmc stays buildable and runnable without ROCm; the GPU path is taken only when the library loads).
 *
 * Environment:
 *   KMC_HIP_LIB      path of libkmc_hip.so (default: <dir of the executable>/../../kmc_amd/libkmc_hip.so, then
 *                    plain "libkmc_hip.so" through the loader path)
 *   KMC_HIP_DEVICES  comma-separated HIP ordinals (default "0"); worker i uses device i % n_devices
 * There is deliberately NO CPU fallback here: if the library or a GPU is missing the engine reports the error and
 * the worker raises it through CCriticalErrorHandler.
 */
#include <dlfcn.h>
#include <unistd.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "bin_engine.h"

namespace {

struct Api {
	void *so = nullptr;
	int (*init)(const int *, int, kmc_hip_ctx **) = nullptr;
	void (*destroy)(kmc_hip_ctx *) = nullptr;
	const char *(*last_error)(kmc_hip_ctx *) = nullptr;
	int (*abi_version)(void) = nullptr;
	int (*process_bin)(kmc_hip_ctx *, int, const kmc_hip_bin_params *, const uint8_t *, uint64_t, uint64_t, const uint64_t *, uint64_t,
	                   uint8_t *, uint64_t, uint64_t *, uint64_t *, uint64_t *) = nullptr;
	int (*host_register)(kmc_hip_ctx *, void *, uint64_t) = nullptr;
	kmc_hip_ctx *ctx = nullptr;
	int n_dev = 0;
	std::string err;
};

Api g_api;
std::once_flag g_once;

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

void load_api()
{
	Api &a = g_api;
	std::vector<std::string> cands;
	if (const char *e = getenv("KMC_HIP_LIB"))
		cands.push_back(e);
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
	    !sym(a.so, "kmc_hip_process_bin", a.process_bin, a.err) || !sym(a.so, "kmc_hip_host_register", a.host_register, a.err)) {
		a.so = nullptr;
		return;
	}
	if (a.abi_version() != KMC_HIP_ABI_VERSION) {
		a.err = "libkmc_hip.so ABI version mismatch";
		a.so = nullptr;
		return;
	}
	std::vector<int> devs;
	const char *e = getenv("KMC_HIP_DEVICES");
	std::string s = e ? e : "0";
	size_t pos = 0;
	while (pos <= s.size()) {
		size_t q = s.find(',', pos);
		if (q == std::string::npos)
			q = s.size();
		if (q > pos)
			devs.push_back(atoi(s.substr(pos, q - pos).c_str()));
		pos = q + 1;
	}
	if (devs.empty())
		devs.push_back(0);
	int rc = a.init(devs.data(), (int)devs.size(), &a.ctx);
	if (rc) {
		a.err = std::string("kmc_hip_init failed: ") + a.last_error(nullptr);
		a.ctx = nullptr;
		return;
	}
	a.n_dev = (int)devs.size();
}

struct HipEngine : KmcBinEngine {
	int dev;
	std::string err;
	explicit HipEngine(int dev) : dev(dev) {}
	int process_bin(const kmc_hip_bin_params &p, const uint8_t *sk, uint64_t size, uint64_t n_rec, const uint64_t *pack_bytes,
	                uint64_t n_packs, uint8_t *out, uint64_t cap, uint64_t *out_bytes, uint64_t *lut, uint64_t stats[4]) override
	{
		if (!g_api.ctx) {
			err = g_api.err.empty() ? "HIP engine not initialised" : g_api.err;
			return KMC_HIP_EDEVICE;
		}
		int rc = g_api.process_bin(g_api.ctx, dev, &p, sk, size, n_rec, pack_bytes, n_packs, out, cap, out_bytes, lut, stats);
		if (rc)
			err = g_api.last_error(g_api.ctx);
		return rc;
	}
	std::string last_error() override { return err; }
	void register_arena(void *ptr, uint64_t bytes) override
	{
		if (g_api.ctx)
			(void)g_api.host_register(g_api.ctx, ptr, bytes);
	}
};

} // namespace

KmcBinEngine *kmc_make_bin_engine(int worker_idx, int /*n_workers*/)
{
	std::call_once(g_once, load_api);
	const int n = g_api.n_dev > 0 ? g_api.n_dev : 1;
	return new HipEngine(worker_idx % n);
}
