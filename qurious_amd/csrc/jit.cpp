// jit.cpp — hiprtc instantiation of the hand-written kernel templates with a generated policy.
//
// source = qhip_status.h + qhip_device.hpp (embedded at build time, device_src.inc) + policy text.
// Compiled code objects are cached in memory per context and on disk (QHIP_KERNEL_CACHE or
// <libdir>/_kcache), keyed by a hash of the full source, so a plan shape is compiled once per machine.
#include "jit.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <unistd.h>

namespace qhip {

static const char* kDeviceSource =
#include "device_src.inc"
    ;

const char* device_source() { return kDeviceSource; }

static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ULL) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ULL; }
  return h;
}

Module::~Module() {
  if (mod) (void)hipModuleUnload(mod);
}

static std::string default_cache_dir() {
  Dl_info info;
  if (dladdr((void*)&default_cache_dir, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.rfind('/');
    if (k != std::string::npos) return p.substr(0, k) + "/_kcache";
  }
  return "";
}

static const char* const kCompileOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};

// what a cached code object depends on besides its source: the compiler (hiprtc version) and the options
static std::string toolchain_tag() {
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  std::string t = "hiprtc " + std::to_string(major) + "." + std::to_string(minor);
  for (const char* o : kCompileOptions) { t += ' '; t += o; }
  return t;
}

std::vector<char> compile_to_code_object(const std::string& full_source, std::string* log_out) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, full_source.c_str(), "qhip_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    fail(QHIP_HIP_ERROR, "hiprtcCreateProgram failed");
  hiprtcResult r = hiprtcCompileProgram(prog, 4, const_cast<const char**>(kCompileOptions));
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  std::string log(ls, '\0');
  if (ls) hiprtcGetProgramLog(prog, &log[0]);
  if (log_out) *log_out = log;
  if (r != HIPRTC_SUCCESS) {
    hiprtcDestroyProgram(&prog);
    fail(QHIP_HIP_ERROR, std::string("hiprtc compile failed: ") + hiprtcGetErrorString(r) + "\n" + log);
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  return code;
}

std::string full_source_for(const std::string& policy_source) { return std::string(kDeviceSource) + "\n" + policy_source; }

std::string cache_path_for(const std::string& dir, const std::string& full_source) {
  char name[64];
  snprintf(name, sizeof name, "/qk_%016llx_%zu.hsaco", (unsigned long long)fnv1a(full_source, fnv1a(toolchain_tag())), full_source.size());
  return dir + name;
}

// tmp file + rename: another process never reads a partial code object
static void write_atomically(const std::string& dir, const std::string& path, const std::vector<char>& code) {
  (void)mkdir(dir.c_str(), 0755);
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  std::ofstream f(tmp, std::ios::binary);
  if (!f) return;
  f.write(code.data(), (std::streamsize)code.size());
  f.close();
  if (rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

std::shared_ptr<Module> get_module(Ctx* ctx, const std::string& policy_source, const std::string& kernel_name) {
  // one loaded module per (source, entry point): a policy source may hold several kernels (the aggregate's partitioned
  // path); the code object is compiled and cached once per source either way
  const std::string mkey = policy_source + "\n//entry:" + kernel_name;
  auto it = ctx->modules.find(mkey);
  if (it != ctx->modules.end()) return it->second;
  const auto t0 = std::chrono::steady_clock::now();
  const std::string src = full_source_for(policy_source);
  std::string dir = ctx->cache_dir.empty() ? default_cache_dir() : ctx->cache_dir;
  std::vector<char> code;
  std::string path;
  if (!dir.empty()) {
    path = cache_path_for(dir, src);
    std::ifstream f(path, std::ios::binary);
    if (f) code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  const bool from_cache = !code.empty();
  auto m = std::make_shared<Module>();
  auto load = [&]() {
    return hipModuleLoadData(&m->mod, code.data()) == hipSuccess && hipModuleGetFunction(&m->fn, m->mod, kernel_name.c_str()) == hipSuccess;
  };
  if (!from_cache || !load()) {
    // nothing cached — or a cached object that does not load (truncated file, another ROCm): drop it and compile once
    if (from_cache) {
      if (m->mod) { (void)hipModuleUnload(m->mod); m->mod = nullptr; }
      (void)unlink(path.c_str());
    }
    code = compile_to_code_object(src, nullptr);
    if (!path.empty()) write_atomically(dir, path, code);
    QHIP_HIP_CHECK(hipModuleLoadData(&m->mod, code.data()));
    QHIP_HIP_CHECK(hipModuleGetFunction(&m->fn, m->mod, kernel_name.c_str()));
  }
  // a long-lived context that keeps seeing new plans does not keep every code object loaded for ever: beyond 512 modules
  // the in-memory cache starts over (callers hold shared_ptrs to what they are running; the disk cache still has the rest)
  if (ctx->modules.size() >= 512) ctx->modules.clear();
  ctx->modules[mkey] = m;
  ctx->stats.jit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return m;
}

}  // namespace qhip

// Ahead-of-time compile of a policy into the on-disk cache (no GPU needed): used by
// __graft_entry__.build() to check that every catalog kernel compiles for gfx950.
extern "C" int qhip_jit_compile_to_cache(const char* policy_source, const char* cache_dir, char* log, size_t log_len) {
  try {
    const std::string src = qhip::full_source_for(policy_source ? policy_source : "");
    std::string l;
    std::vector<char> code = qhip::compile_to_code_object(src, &l);
    if (log && log_len) snprintf(log, log_len, "%s", l.c_str());
    if (cache_dir && *cache_dir) qhip::write_atomically(cache_dir, qhip::cache_path_for(cache_dir, src), code);
    return QHIP_OK;
  } catch (const qhip::Error& e) {
    if (log && log_len) snprintf(log, log_len, "%s", e.what());
    return e.code;
  }
}
