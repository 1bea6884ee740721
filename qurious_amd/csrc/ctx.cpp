// ctx.cpp — context lifecycle, dtype helpers, device buffers.
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <execinfo.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"
#include "device/qhip_status.h"

namespace qhip {

static std::atomic<uint64_t> g_sync_count{0};   // process-wide (contexts on several threads): atomic
uint64_t sync_counter() { return g_sync_count.load(std::memory_order_relaxed); }
void trace_point(const char* what) {
  static const bool on = env_int("QHIP_TRACE", 0) >= 2;
  if (!on) return;
  static const auto t0 = std::chrono::steady_clock::now();
  fprintf(stderr, "[qhip t] %10.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
}
void note_sync() {
  g_sync_count.fetch_add(1, std::memory_order_relaxed);
  static const bool trace = env_int("QHIP_SYNC_TRACE", 0) != 0;
  if (trace) {
    void* frames[6];
    const int n = backtrace(frames, 6);
    fprintf(stderr, "[qhip] host wait #%llu\n", (unsigned long long)sync_counter());
    backtrace_symbols_fd(frames + 1, n - 1, 2);
  }
}

uint32_t* zeroed_block(Ctx* ctx, int n) {
  // two halves of 256 blocks: the half being handed out from was cleared when it was entered; moving on to the other half
  // clears THAT one only, so blocks a call already holds (it may ask for several, one after another) stay intact — a
  // single call would have to take more than 256 blocks to catch up with itself
  constexpr size_t kBlock = 128, kHalf = 256;
  if (n < 1 || (size_t)n > kHalf / 4) fail(QHIP_HIP_ERROR, "zeroed_block: bad block count (internal error)");
  if (!ctx->zero_ring.ptr) {
    ctx->zero_ring.alloc(kBlock * kHalf * 2);
    QHIP_HIP_CHECK(hipMemsetAsync(ctx->zero_ring.ptr, 0, kBlock * kHalf, ctx->stream));
    ctx->zero_next = 0;
  }
  size_t half = ctx->zero_next / kHalf, off = ctx->zero_next % kHalf;
  if (ctx->zero_next == 2 * kHalf || off + (size_t)n > kHalf || (off == 0 && ctx->zero_next != 0)) {
    half = ctx->zero_next == 2 * kHalf ? 0 : (off == 0 ? half : half + 1) % 2;
    QHIP_HIP_CHECK(hipMemsetAsync(ctx->zero_ring.as<uint8_t>() + half * kHalf * kBlock, 0, kBlock * kHalf, ctx->stream));
    off = 0;
  }
  uint32_t* p = (uint32_t*)(ctx->zero_ring.as<uint8_t>() + (half * kHalf + off) * kBlock);
  ctx->zero_next = half * kHalf + off + (size_t)n;
  return p;
}

void verify_pending_sizes(Ctx* ctx) {
  if (ctx->pending_sizes.empty()) return;
  std::vector<Ctx::PendingSize> pend;
  pend.swap(ctx->pending_sizes);
  std::string why;
  for (const Ctx::PendingSize& p : pend) {
    const uint32_t* build = p.slot;                    // [build status | probe status | pair total]
    const uint64_t total = p.slot[2 * QS_WORDS];
    if (p.total_out) *p.total_out = total;
    if (build[QS_MAXCOUNT] > 1) {                      // duplicate build keys after all: remember, run again the careful way
      if (ctx->join_dup_builds.size() > 4096) ctx->join_dup_builds.clear();
      ctx->join_dup_builds.insert(p.dup_hint);
      ctx->join_size_hints.erase(p.key);
      why = "duplicate build keys";
      continue;
    }
    if (build[QS_OVERFLOW]) { ctx->join_size_hints.erase(p.key); why = "build table overflow"; continue; }
    if (total > p.capacity) {
      ctx->join_size_hints.erase(p.key);
      why = std::to_string(total) + " pairs, room for " + std::to_string(p.capacity);
      continue;
    }
    ctx->join_size_hints[p.key] = total;
  }
  if (!why.empty()) fail(QHIP_RETRY, "a hash join that did not wait for its size has to run again (" + why + ")");
  // data-dependent errors of the key / filter expressions surface exactly as they would have in the join's own call
  for (const Ctx::PendingSize& p : pend)
    for (const uint32_t* st : {(const uint32_t*)p.slot, (const uint32_t*)p.slot + QS_WORDS}) {
      if (st[QS_KEY_TOO_LONG]) fail(QHIP_UNSUPPORTED, "Utf8 group/join key longer than its packed key words (expression keys: 7 bytes)");
      if (st[QS_DIV_ZERO]) fail(QHIP_EXEC_ERROR, "Arrow error: Divide by zero error");
      if (st[QS_CAST_OVERFLOW]) fail(QHIP_EXEC_ERROR, "Arrow error: Cast error: value out of range for the target type");
      if (st[QS_ARITH_OVERFLOW]) fail(QHIP_EXEC_ERROR, "Arrow error: Arithmetic overflow: Overflow happened on integer division");
    }
}

static std::mutex g_err_mu;
static std::string g_err;
void set_global_error(const std::string& m) { std::lock_guard<std::mutex> l(g_err_mu); g_err = m; }

std::string dtype_name(const DType& t) {
  switch (t.id) {
    case QHIP_NULL: return "Null";
    case QHIP_BOOL: return "Boolean";
    case QHIP_INT8: return "Int8";
    case QHIP_INT16: return "Int16";
    case QHIP_INT32: return "Int32";
    case QHIP_INT64: return "Int64";
    case QHIP_UINT8: return "UInt8";
    case QHIP_UINT16: return "UInt16";
    case QHIP_UINT32: return "UInt32";
    case QHIP_UINT64: return "UInt64";
    case QHIP_FLOAT32: return "Float32";
    case QHIP_FLOAT64: return "Float64";
    case QHIP_DATE32: return "Date32";
    case QHIP_DATE64: return "Date64";
    case QHIP_TIME32_S: return "Time32(Second)";
    case QHIP_TIME32_MS: return "Time32(Millisecond)";
    case QHIP_TIME64_US: return "Time64(Microsecond)";
    case QHIP_TIME64_NS: return "Time64(Nanosecond)";
    case QHIP_TIMESTAMP_S: return "Timestamp(Second, None)";
    case QHIP_TIMESTAMP_MS: return "Timestamp(Millisecond, None)";
    case QHIP_TIMESTAMP_US: return "Timestamp(Microsecond, None)";
    case QHIP_TIMESTAMP_NS: return "Timestamp(Nanosecond, None)";
    case QHIP_DECIMAL128: return "Decimal128(" + std::to_string(t.precision) + ", " + std::to_string(t.scale) + ")";
    case QHIP_UTF8: return "Utf8";
  }
  return "Unknown(" + std::to_string(t.id) + ")";
}

int dtype_width(const DType& t) {
  switch (t.id) {
    case QHIP_INT8: case QHIP_UINT8: return 1;
    case QHIP_INT16: case QHIP_UINT16: return 2;
    case QHIP_INT32: case QHIP_UINT32: case QHIP_FLOAT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: return 4;
    case QHIP_INT64: case QHIP_UINT64: case QHIP_FLOAT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: return 8;
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: return 8;
    case QHIP_DECIMAL128: return 16;
    default: return 0;
  }
}
bool dtype_is_integer(const DType& t) { return t.id >= QHIP_INT8 && t.id <= QHIP_UINT64; }
bool dtype_is_signed(const DType& t) { return t.id >= QHIP_INT8 && t.id <= QHIP_INT64; }
bool dtype_is_float(const DType& t) { return t.id == QHIP_FLOAT32 || t.id == QHIP_FLOAT64; }

std::string dtype_to_format(const DType& t) {
  switch (t.id) {
    case QHIP_NULL: return "n";
    case QHIP_BOOL: return "b";
    case QHIP_INT8: return "c";
    case QHIP_UINT8: return "C";
    case QHIP_INT16: return "s";
    case QHIP_UINT16: return "S";
    case QHIP_INT32: return "i";
    case QHIP_UINT32: return "I";
    case QHIP_INT64: return "l";
    case QHIP_UINT64: return "L";
    case QHIP_FLOAT32: return "f";
    case QHIP_FLOAT64: return "g";
    case QHIP_DATE32: return "tdD";
    case QHIP_DATE64: return "tdm";
    case QHIP_TIME32_S: return "tts";
    case QHIP_TIME32_MS: return "ttm";
    case QHIP_TIME64_US: return "ttu";
    case QHIP_TIME64_NS: return "ttn";
    case QHIP_TIMESTAMP_S: return "tss:";
    case QHIP_TIMESTAMP_MS: return "tsm:";
    case QHIP_TIMESTAMP_US: return "tsu:";
    case QHIP_TIMESTAMP_NS: return "tsn:";
    case QHIP_DECIMAL128: return "d:" + std::to_string(t.precision) + "," + std::to_string(t.scale);
    case QHIP_UTF8: return "u";
  }
  fail(QHIP_UNSUPPORTED, "no Arrow format for " + dtype_name(t));
}

DType dtype_from_format(const char* f) {
  std::string s(f ? f : "");
  if (s == "n") return DType(QHIP_NULL);
  if (s == "b") return DType(QHIP_BOOL);
  if (s == "c") return DType(QHIP_INT8);
  if (s == "C") return DType(QHIP_UINT8);
  if (s == "s") return DType(QHIP_INT16);
  if (s == "S") return DType(QHIP_UINT16);
  if (s == "i") return DType(QHIP_INT32);
  if (s == "I") return DType(QHIP_UINT32);
  if (s == "l") return DType(QHIP_INT64);
  if (s == "L") return DType(QHIP_UINT64);
  if (s == "f") return DType(QHIP_FLOAT32);
  if (s == "g") return DType(QHIP_FLOAT64);
  if (s == "tdD") return DType(QHIP_DATE32);
  if (s == "tdm") return DType(QHIP_DATE64);
  if (s == "tts") return DType(QHIP_TIME32_S);
  if (s == "ttm") return DType(QHIP_TIME32_MS);
  if (s == "ttu") return DType(QHIP_TIME64_US);
  if (s == "ttn") return DType(QHIP_TIME64_NS);
  if (s == "tss:") return DType(QHIP_TIMESTAMP_S);
  if (s == "tsm:") return DType(QHIP_TIMESTAMP_MS);
  if (s == "tsu:") return DType(QHIP_TIMESTAMP_US);
  if (s == "tsn:") return DType(QHIP_TIMESTAMP_NS);
  if (s.size() > 4 && s[0] == 't' && s[1] == 's' && s[3] == ':')
    fail(QHIP_UNSUPPORTED, "timestamp columns with a timezone (" + s + ") are not supported by the HIP backend");
  if (s == "u") return DType(QHIP_UTF8);
  if (s.rfind("d:", 0) == 0) {
    int p = 0, sc = 0, bits = 128;
    int n = sscanf(s.c_str(), "d:%d,%d,%d", &p, &sc, &bits);
    if (n >= 2 && bits == 128) return DType(QHIP_DECIMAL128, p, sc);
  }
  fail(QHIP_UNSUPPORTED, "Arrow format '" + s + "' is not supported by the HIP backend");
}

// ---------------------------------------------------------------- caching device allocator
// hipMalloc / hipFree cost 10^2 us each and hipFree synchronises the device; an operator call allocates a dozen
// temporaries. Freed blocks are kept per (device, size class) and reused; size classes are 1/8-octave steps (<= 12.5 %
// slack), so the steady state of a repeated query performs no driver allocation at all. 288 GB of HBM per GPU make the
// cache cap (QHIP_POOL_MAX_GB, default 64 GB) a non-issue.
namespace {
struct Pool {
  std::mutex mu;
  std::unordered_map<size_t, std::vector<void*>> free_blocks;
  size_t cached = 0;
};
Pool g_pools[32];
// Blocks are recycled without a synchronisation because everything of a context runs on its one stream (the next user of
// a block is ordered after the previous one). That argument needs ONE stream per device: with a second live context on
// the same device a block is returned to the pool only after the device is idle.
std::atomic<int> g_live_contexts[32];
size_t pool_cap_bytes() {
  static size_t cap = [] { const char* v = getenv("QHIP_POOL_MAX_GB"); return (size_t)(v && *v ? atof(v) : 64.0) * (1ULL << 30); }();
  return cap;
}
size_t size_class(size_t n) {
  if (n <= 4096) return 4096;
  int lg = 63 - __builtin_clzll((unsigned long long)n);
  size_t step = (size_t)1 << (lg - 3);
  return (n + step - 1) / step * step;
}
}  // namespace

void DevBuf::alloc(size_t n) {
  release();
  // 64 bytes of slack: kernels read Utf8 values with one unaligned 8-byte load (qh_pack_str7) and never fault
  // on the last value of a buffer; n == 0 still yields a valid pointer
  const size_t m = size_class(n + 64);
  int dev = 0;
  (void)hipGetDevice(&dev);
  Pool& pool = g_pools[dev & 31];
  {
    std::lock_guard<std::mutex> l(pool.mu);
    auto it = pool.free_blocks.find(m);
    if (it != pool.free_blocks.end() && !it->second.empty()) {
      ptr = it->second.back();
      it->second.pop_back();
      pool.cached -= m;
    }
  }
  if (!ptr) {
    hipError_t e = hipMalloc(&ptr, m);
    if (e == hipErrorOutOfMemory) {
      // give the cache back to the driver and retry once
      std::lock_guard<std::mutex> l(pool.mu);
      for (auto& kv : pool.free_blocks) { for (void* p : kv.second) (void)hipFree(p); kv.second.clear(); }
      pool.cached = 0;
      e = hipMalloc(&ptr, m);
    }
    if (e != hipSuccess) {
      ptr = nullptr;
      fail(e == hipErrorOutOfMemory ? QHIP_OUT_OF_MEMORY : QHIP_HIP_ERROR, "hipMalloc(" + std::to_string(m) + "): " + hipGetErrorString(e));
    }
  }
  bytes = n;
  cap = m;
  device = dev;
}
void DevBuf::release() {
  if (owner) { owner.reset(); ptr = nullptr; bytes = 0; cap = 0; return; }   // a view: the owner's last reference frees the memory
  if (!ptr) return;
  if (g_live_contexts[device & 31].load(std::memory_order_relaxed) > 1) (void)hipDeviceSynchronize();
  Pool& pool = g_pools[device & 31];
  bool kept = false;
  {
    std::lock_guard<std::mutex> l(pool.mu);
    if (pool.cached + cap <= pool_cap_bytes()) {
      pool.free_blocks[cap].push_back(ptr);
      pool.cached += cap;
      kept = true;
    }
  }
  if (!kept) (void)hipFree(ptr);
  ptr = nullptr; bytes = 0; cap = 0;
}

int64_t DevColumn::resident_bytes() const {
  // a column whose upload / gather was owed and has been done through another table sharing it counts as resident
  if (pending_upload) return pending_upload->done ? pending_upload->result.resident_bytes() : 0;
  if (deferred) return deferred->done ? deferred->result.resident_bytes() : 0;
  int64_t b = 0;
  if (values) b += (int64_t)values->bytes;
  if (validity) b += (int64_t)validity->bytes;
  if (data) b += data_bytes;
  return b;
}

}  // namespace qhip

using namespace qhip;

extern "C" {

const char* qhip_version(void) { return "qhip 0.1.0 (gfx950)"; }

int qhip_device_available(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  return (e == hipSuccess && n > 0) ? 1 : 0;
}

int qhip_ctx_create(int device_index, qhip_ctx** out) {
  if (!out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(nullptr, [&] {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
      fail(QHIP_HIP_ERROR, std::string("no HIP device visible (hipGetDeviceCount: ") + hipGetErrorString(e) +
                               "); libqhip has no CPU fallback");
    int dev = device_index;
    if (dev < 0) QHIP_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= n) fail(QHIP_INVALID_ARGUMENT, "device index out of range");
    QHIP_HIP_CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    QHIP_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    std::string arch = prop.gcnArchName;
    if (arch.rfind("gfx950", 0) != 0 && !getenv("QHIP_ALLOW_ANY_ARCH"))
      fail(QHIP_HIP_ERROR, "device " + std::to_string(dev) + " is " + arch + ", libqhip kernels are written for gfx950 (MI355X)");
    std::unique_ptr<qhip_ctx> c(new qhip_ctx());
    c->device = dev;
    std::string pname = prop.name;
    if (pname.empty()) pname = "AMD Instinct MI355X";   // some runtime/driver combinations leave hipDeviceProp_t::name empty
    c->device_name = pname + " (" + arch + ")";
    c->num_cus = prop.multiProcessorCount;
    QHIP_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    g_live_contexts[c->device & 31].fetch_add(1);
    for (auto& ev : c->ev) QHIP_HIP_CHECK(hipEventCreate(&ev));
    c->status.alloc(4 * QS_WORDS * sizeof(uint32_t));   // (a hash join uses three blocks: build status, probe status, pair total)
    c->timing = env_int("QHIP_TIMING", 0) != 0;
    c->pinned_bytes = 256 * 1024;
    QHIP_HIP_CHECK(hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault));
    memset(&c->stats, 0, sizeof(c->stats));
    const char* cd = getenv("QHIP_KERNEL_CACHE");
    c->cache_dir = cd ? cd : "";
    *out = c.release();
  });
}

void qhip_ctx_destroy(qhip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  ctx->modules.clear();
  ctx->plan_cache.clear();
  ctx->status.release();
  ctx->zero_ring.release();
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (int k = 0; k < 2; ++k) {
    if (ctx->up_slot[k]) (void)hipHostFree(ctx->up_slot[k]);
    if (ctx->up_ev[k]) (void)hipEventDestroy(ctx->up_ev[k]);
  }
  for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
  if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); g_live_contexts[ctx->device & 31].fetch_sub(1); }
  for (auto& k : ctx->host_keep) { k.p.reset(); if (k.ev) (void)hipEventDestroy(k.ev); }   // (behind the wait: their copies are done)
  delete ctx;
}

const char* qhip_last_error(const qhip_ctx* ctx) {
  if (ctx) return ctx->last_error.c_str();
  static thread_local std::string copy;
  std::lock_guard<std::mutex> l(g_err_mu);
  copy = g_err;
  return copy.c_str();
}

int qhip_ctx_synchronize(qhip_ctx* ctx) {
  if (!ctx) return QHIP_INVALID_ARGUMENT;
  return guarded(ctx, [&] { QHIP_HIP_CHECK(hipSetDevice(ctx->device)); QHIP_HIP_CHECK(sync_stream(ctx->stream)); });
}

uint64_t qhip_ctx_sync_count(const qhip_ctx*) { return qhip::sync_counter(); }

int qhip_ctx_set_timing(qhip_ctx* ctx, int32_t on) {
  if (!ctx) return QHIP_INVALID_ARGUMENT;
  ctx->timing = on != 0;
  return QHIP_OK;
}

int qhip_ctx_allow_deferred_sizes(qhip_ctx* ctx, int32_t delta) {
  if (!ctx) return QHIP_INVALID_ARGUMENT;
  if (delta == 0) { ctx->allow_deferred_sizes = 0; ctx->pending_sizes.clear(); }   // reset (after an error above a deferred join)
  else ctx->allow_deferred_sizes = std::max(0, ctx->allow_deferred_sizes + (int)delta);
  return QHIP_OK;
}

int qhip_ctx_forget_plans(qhip_ctx* ctx) {
  if (!ctx) return QHIP_INVALID_ARGUMENT;
  return guarded(ctx, [&] {
    QHIP_HIP_CHECK(hipSetDevice(ctx->device));
    QHIP_HIP_CHECK(sync_stream(ctx->stream));   // (plans own device arenas: nothing of theirs may still be in flight)
    // joins of deferred size still in flight: their status slots are read now (the stream is idle) so that the tables they
    // produced get their verified row counts (rows_final); an overflow is of no consequence for a caller that forgets the plans
    try { verify_pending_sizes(ctx); } catch (const Error&) {}
    ctx->plan_cache.clear();
    ctx->join_size_hints.clear();
    ctx->join_dup_builds.clear();
    ctx->agg_group_hints.clear();
    ctx->agg_slot_words.clear();
    ctx->pending_sizes.clear();
  });
}

int qhip_table_forget_statistics(qhip_table* t) {
  if (!t) return QHIP_INVALID_ARGUMENT;
  for (DevColumn& c : t->cols) {
    for (DevColumn* col : {&c, c.deferred && c.deferred->done ? &c.deferred->result : nullptr, c.pending_upload && c.pending_upload->done ? &c.pending_upload->result : nullptr}) {
      if (!col) continue;
      col->value_maxabs = 0;
      col->utf8_max_len = -1;
      col->narrow.reset();
      col->big_reads = 0;
      col->range = std::make_shared<ColRange>();
      col->range_inherited = false;
    }
  }
  return QHIP_OK;
}

int64_t qhip_table_aux_bytes(const qhip_table* t) {
  if (!t) return 0;
  int64_t b = 0;
  for (const DevColumn& c : t->cols) {
    const DevColumn& col = c.pending_upload && c.pending_upload->done ? c.pending_upload->result : c;
    if (col.narrow && col.narrow->buf) b += (int64_t)col.narrow->buf->bytes;
    if (col.range && col.range->narrow_buf && !col.range_inherited) b += (int64_t)col.range->narrow_buf->bytes;   // (made for readers behind an index vector)
    if (col.range && col.range->rec_buf && col.range->rec_offset == 0 && !col.range_inherited) b += (int64_t)col.range->rec_buf->bytes;   // (the record copy, counted with its first field)
  }
  return b;
}

int qhip_ctx_last_stats(const qhip_ctx* ctx, qhip_exec_stats* out) {
  if (!ctx || !out) return QHIP_INVALID_ARGUMENT;
  if (ctx->stats_timing_pending) {
    // operators do not wait for their last kernels just to time them: the events are read here
    qhip_ctx* c = const_cast<qhip_ctx*>(ctx);
    float ms = 0;
    if (sync_event(c->ev[1]) == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->stats.total_device_ms = ms;
    if (c->stats_timing_pending >= 2 && hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->stats.main_kernel_ms = ms;
    if (c->stats_timing_pending == 3 && hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) c->stats.build_ms = ms;   // (partition: pass 1 + scan)
    if (c->stats_timing_pending == 2 && hipEventElapsedTime(&ms, c->ev[0], c->ev[2]) == hipSuccess) c->stats.build_ms = ms;   // (hash join)
    else if (c->stats_timing_pending == 1) c->stats.main_kernel_ms = c->stats.total_device_ms;
    c->stats_timing_pending = 0;
  }
  *out = ctx->stats;
  return QHIP_OK;
}

int qhip_ctx_device_name(const qhip_ctx* ctx, char* buf, size_t buflen) {
  if (!ctx || !buf || !buflen) return QHIP_INVALID_ARGUMENT;
  snprintf(buf, buflen, "%s", ctx->device_name.c_str());
  return QHIP_OK;
}

}  // extern "C"
