// table.cpp — qhip_table: Arrow C Data Interface <-> HBM-resident columns.
//
// A qhip_table stands for the Vec<RecordBatch> flowing between the reference's operators
// (physical/plan/mod.rs:27). Upload concatenates the batches of each column into one device
// buffer (what the reference does with concat_batches, aggregate/hash.rs:150, join/hash_join.rs:154,
// happens here for free as part of the host->HBM copy) and remembers the batch boundaries.
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdlib>

#include <mutex>

#include "common.hpp"
#include "hostcol.hpp"
#include "kernels.hpp"
#include "relops.hpp"

namespace qhip {

static inline bool host_bit(const uint8_t* bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
static inline void host_set_bit(uint8_t* bm, int64_t i) { bm[i >> 3] |= (uint8_t)(1u << (i & 7)); }

// copy `n` bits from src starting at bit `src_off` to dst starting at bit `dst_off` (dst pre-zeroed there)
static void copy_bits(const uint8_t* src, int64_t src_off, uint8_t* dst, int64_t dst_off, int64_t n) {
  if (((src_off | dst_off) & 7) == 0) {
    int64_t nb = n >> 3;
    memcpy(dst + (dst_off >> 3), src + (src_off >> 3), (size_t)nb);
    for (int64_t i = nb << 3; i < n; ++i) if (host_bit(src, src_off + i)) host_set_bit(dst, dst_off + i);
    return;
  }
  for (int64_t i = 0; i < n; ++i) if (host_bit(src, src_off + i)) host_set_bit(dst, dst_off + i);
}
static void set_bits(uint8_t* dst, int64_t dst_off, int64_t n) {
  int64_t i = 0;
  while (i < n && ((dst_off + i) & 7)) { host_set_bit(dst, dst_off + i); ++i; }
  int64_t nb = (n - i) >> 3;
  memset(dst + ((dst_off + i) >> 3), 0xff, (size_t)nb);
  i += nb << 3;
  for (; i < n; ++i) host_set_bit(dst, dst_off + i);
}
static int64_t count_set_bits(const uint8_t* bm, int64_t off, int64_t n) {
  int64_t c = 0;
  int64_t i = 0;
  while (i < n && ((off + i) & 7)) { c += host_bit(bm, off + i); ++i; }
  for (; i + 8 <= n; i += 8) c += __builtin_popcount(bm[(off + i) >> 3]);
  for (; i < n; ++i) c += host_bit(bm, off + i);
  return c;
}

// ---------------------------------------------------------------- host -> HBM streaming
// Arrow buffers live in pageable memory; hipMemcpyAsync streams them through the runtime's page-locked staging at
// ~50 GB/s on this platform (PCIe 5 x16; measured with tools/upload_timing.py). Two alternatives were built and measured
// no faster, and dropped: page-locking the caller's buffers in place (hipHostRegister costs ~11 GB/s, as much as it
// saves) and an own ring of page-locked slots filled by copy threads (48 GB/s). What did matter: no per-row work on the
// host — Utf8 offsets are rebased by a kernel after the copy (the host loop alone held the whole upload at 19 GB/s).
static void h2d(void* dst, const void* src, size_t n, hipStream_t s) {
  if (n) QHIP_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s));
}
static void h2d_stream(Ctx* ctx, void* dst, const void* src, size_t n) { h2d(dst, src, n, ctx->stream); }
static void h2d_flush(Ctx*) {}
static void d2h(void* dst, const void* src, size_t n, hipStream_t s) {
  if (n) QHIP_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s));
}

// Many small batches — the reference's CSV loader produces 1024-row batches (datasource/file/csv.rs:63-66: SF10 lineitem =
// 58.6 k batches) — would mean one copy call per 4-16 KB buffer at ~10 us each. Their bytes are gathered instead into two
// page-locked 8 MB slots (host memcpy) and sent as large asynchronous copies to a CONTIGUOUS device range, the fill of one
// slot overlapping the DMA of the other.
struct Coalescer {
  static constexpr size_t kSlot = 8u << 20;
  Ctx* ctx;
  uint8_t* dst;          // device: next byte to write
  int cur = 0;
  size_t fill = 0;
  Coalescer(Ctx* c, void* device_dst) : ctx(c), dst((uint8_t*)device_dst) {
    for (int k = 0; k < 2; ++k)
      if (!ctx->up_slot[k]) {
        QHIP_HIP_CHECK(hipHostMalloc(&ctx->up_slot[k], kSlot, hipHostMallocDefault));
        QHIP_HIP_CHECK(hipEventCreateWithFlags(&ctx->up_ev[k], hipEventDisableTiming));
        QHIP_HIP_CHECK(hipEventRecord(ctx->up_ev[k], ctx->stream));
      }
    QHIP_HIP_CHECK(sync_event(ctx->up_ev[cur]));   // a previous user's copy out of this slot has finished
  }
  void send() {
    if (!fill) return;
    QHIP_HIP_CHECK(hipMemcpyAsync(dst, ctx->up_slot[cur], fill, hipMemcpyHostToDevice, ctx->stream));
    QHIP_HIP_CHECK(hipEventRecord(ctx->up_ev[cur], ctx->stream));
    dst += fill;
    fill = 0;
    cur ^= 1;
    QHIP_HIP_CHECK(sync_event(ctx->up_ev[cur]));   // the other slot's copy (two sends ago) has finished
  }
  void append(const void* src, size_t n) {
    const uint8_t* p = (const uint8_t*)src;
    while (n) {
      const size_t k = std::min(n, kSlot - fill);
      memcpy((uint8_t*)ctx->up_slot[cur] + fill, p, k);
      fill += k; p += k; n -= k;
      if (fill == kSlot) send();
    }
  }
  void finish() { send(); }
};

// One column of the batches -> HBM, concatenated (values / offsets rebased / bitmaps realigned). Synchronous.
static DevColumn upload_column(Ctx* ctx, const char* format, int64_t c, const ArrowArray* const* batches, int64_t nb,
                               const std::vector<int64_t>& batch_offsets) {
  const int64_t N = batch_offsets.back();
  std::vector<std::vector<uint8_t>> staging;  // host staging kept alive until the stream is drained
  DevColumn col;
  col.type = dtype_from_format(format);
  col.length = N;
  // null count + validity
  int64_t nulls = 0;
  for (int64_t b = 0; b < nb; ++b) {
    const ArrowArray* ca = batches[b]->children[c];
    if (ca->length != batches[b]->length) fail(QHIP_INVALID_ARGUMENT, "column length differs from its batch length");
    if (col.type.id == QHIP_NULL) { nulls += ca->length; continue; }
    const uint8_t* bm = ca->n_buffers > 0 ? (const uint8_t*)ca->buffers[0] : nullptr;
    if (bm && ca->null_count != 0) nulls += ca->length - count_set_bits(bm, ca->offset, ca->length);
  }
  col.null_count = nulls;
  if (nulls > 0 && col.type.id != QHIP_NULL) {
    staging.emplace_back((size_t)((N + 7) / 8 + 8), 0);
    uint8_t* dst = staging.back().data();
    for (int64_t b = 0; b < nb; ++b) {
      const ArrowArray* ca = batches[b]->children[c];
      const uint8_t* bm = (const uint8_t*)ca->buffers[0];
      if (bm && ca->null_count != 0) copy_bits(bm, ca->offset, dst, batch_offsets[(size_t)b], ca->length);
      else set_bits(dst, batch_offsets[(size_t)b], ca->length);
    }
    col.validity = std::make_shared<DevBuf>((size_t)((N + 7) / 8 + 8));
    h2d(col.validity->ptr, dst, col.validity->bytes, ctx->stream);
  }
  const int w = dtype_width(col.type);
  // (the page-locked slots only pay off when a copy call would otherwise move a few KB)
  const bool small_batches = nb > 8 && N / nb < 32768 && env_int("QHIP_UPLOAD_NO_COALESCE", 0) == 0;
  if (w > 0) {
    col.values = std::make_shared<DevBuf>((size_t)N * w);
    std::unique_ptr<Coalescer> co(small_batches ? new Coalescer(ctx, col.values->ptr) : nullptr);
    for (int64_t b = 0; b < nb; ++b) {
      const ArrowArray* ca = batches[b]->children[c];
      if (ca->length == 0) continue;
      if (ca->n_buffers < 2 || !ca->buffers[1]) fail(QHIP_INVALID_ARGUMENT, "missing values buffer");
      const uint8_t* src = (const uint8_t*)ca->buffers[1] + (size_t)ca->offset * w;
      if (co) co->append(src, (size_t)ca->length * w);
      else h2d_stream(ctx, (uint8_t*)col.values->ptr + (size_t)batch_offsets[(size_t)b] * w, src, (size_t)ca->length * w);
    }
    if (co) co->finish();
  } else if (col.type.id == QHIP_BOOL) {
    staging.emplace_back((size_t)((N + 7) / 8 + 8), 0);
    uint8_t* dst = staging.back().data();
    for (int64_t b = 0; b < nb; ++b) {
      const ArrowArray* ca = batches[b]->children[c];
      if (ca->length) copy_bits((const uint8_t*)ca->buffers[1], ca->offset, dst, batch_offsets[(size_t)b], ca->length);
    }
    col.values = std::make_shared<DevBuf>(staging.back().size());
    h2d(col.values->ptr, dst, col.values->bytes, ctx->stream);
  } else if (col.type.id == QHIP_UTF8) {
    // the int32 offsets of every batch are uploaded as they are and rebased onto the concatenated data buffer on the
    // device (a host loop over every row used to cost more than the transfer itself)
    int64_t total = 0;
    for (int64_t b = 0; b < nb; ++b) {
      const ArrowArray* ca = batches[b]->children[c];
      if (ca->length == 0) continue;
      const int32_t* so = (const int32_t*)ca->buffers[1] + ca->offset;
      total += (int64_t)so[ca->length] - so[0];
    }
    if (total > 0x7fffffffLL) fail(QHIP_UNSUPPORTED, "Utf8 column larger than 2 GiB (needs LargeUtf8 offsets)");
    col.data = std::make_shared<DevBuf>((size_t)total);
    col.data_bytes = total;
    col.values = std::make_shared<DevBuf>((size_t)(N + 1) * 4);
    int32_t* off_dev = col.values->as<int32_t>();
    std::vector<int32_t> shifts((size_t)nb, 0);
    int64_t pos = 0;
    if (small_batches) {
      // offsets and bytes each stream into a contiguous range: two passes over the batches through the two slots
      {
        Coalescer co(ctx, off_dev);
        for (int64_t b = 0; b < nb; ++b) {
          const ArrowArray* ca = batches[b]->children[c];
          if (ca->length == 0) continue;
          const int32_t* so = (const int32_t*)ca->buffers[1] + ca->offset;
          shifts[(size_t)b] = (int32_t)pos - so[0];
          pos += (int64_t)so[ca->length] - so[0];
          co.append(so, (size_t)ca->length * 4);
        }
        co.finish();
      }
      {
        Coalescer co(ctx, col.data->ptr);
        for (int64_t b = 0; b < nb; ++b) {
          const ArrowArray* ca = batches[b]->children[c];
          if (ca->length == 0) continue;
          const int32_t* so = (const int32_t*)ca->buffers[1] + ca->offset;
          co.append((const uint8_t*)ca->buffers[2] + so[0], (size_t)((int64_t)so[ca->length] - so[0]));
        }
        co.finish();
      }
      // one rebasing launch for all batches: row -> batch by binary search in the batch starts
      staging.emplace_back((size_t)(nb + 1) * 8 + (size_t)nb * 4, 0);
      uint64_t* starts = (uint64_t*)staging.back().data();
      for (int64_t b = 0; b <= nb; ++b) starts[b] = (uint64_t)batch_offsets[(size_t)b];
      memcpy(starts + nb + 1, shifts.data(), (size_t)nb * 4);
      DevBuf meta(staging.back().size());
      h2d(meta.ptr, staging.back().data(), staging.back().size(), ctx->stream);
      launch_add_i32_batched(off_dev, (uint64_t)N, meta.as<uint64_t>(), (const int32_t*)(meta.as<uint64_t>() + nb + 1), (uint32_t)nb, ctx->stream);
    } else {
      for (int64_t b = 0; b < nb; ++b) {
        const ArrowArray* ca = batches[b]->children[c];
        if (ca->length == 0) continue;
        const int32_t* so = (const int32_t*)ca->buffers[1] + ca->offset;
        const int32_t base = so[0];
        shifts[(size_t)b] = (int32_t)pos - base;
        h2d_stream(ctx, off_dev + batch_offsets[(size_t)b], so, (size_t)ca->length * 4);
        const int64_t nbytes = (int64_t)so[ca->length] - base;
        h2d_stream(ctx, (uint8_t*)col.data->ptr + pos, (const uint8_t*)ca->buffers[2] + base, (size_t)nbytes);
        pos += nbytes;
      }
      h2d_flush(ctx);   // every copy is on the stream: the rebasing kernels below are ordered behind them
      for (int64_t b = 0; b < nb; ++b) {
        const ArrowArray* ca = batches[b]->children[c];
        launch_add_i32(off_dev + batch_offsets[(size_t)b], (uint64_t)ca->length, shifts[(size_t)b], ctx->stream);
      }
    }
    staging.emplace_back(4, 0);
    const int32_t last = (int32_t)total;
    memcpy(staging.back().data(), &last, 4);
    h2d(off_dev + N, staging.back().data(), 4, ctx->stream);
  } else if (col.type.id == QHIP_NULL) {
    col.null_count = N;
  }
  h2d_flush(ctx);                                      // every chunk staged and on the stream
  QHIP_HIP_CHECK(sync_stream(ctx->stream));   // the staging vectors die here
  return col;
}

// The host-side Arrow arrays a lazily uploaded table keeps: moved out of the caller's structs (Arrow C Data Interface
// move semantics: the source's release is set to NULL), released when the last column that needs them is gone.
struct HostBatches {
  std::vector<ArrowArray> arrays;
  std::vector<const ArrowArray*> ptrs;
  ~HostBatches() { for (auto& a : arrays) if (a.release) a.release(&a); }
};

DevColumn upload_host_column(Ctx* ctx, const DeferredUpload& u);   // agg.cpp (HostColumn)

DevColumn materialize_upload(Ctx* ctx, const DeferredUpload& u) {
  if (u.host_col) return upload_host_column(ctx, u);
  auto hb = std::static_pointer_cast<HostBatches>(u.host);
  return upload_column(ctx, u.format.c_str(), u.column, hb->ptrs.data(), (int64_t)hb->ptrs.size(), u.batch_offsets);
}

qhip_table* table_from_arrow(Ctx* ctx, const ArrowSchema* schema, ArrowArray* const* batches, int64_t nb, bool lazy) {
  if (!schema || !schema->format || strcmp(schema->format, "+s") != 0)
    fail(QHIP_INVALID_ARGUMENT, "qhip_table_from_arrow: schema must be a struct ('+s') describing a RecordBatch");
  const int64_t nc = schema->n_children;
  std::unique_ptr<qhip_table> t(new qhip_table());
  t->ctx = ctx;
  t->batch_offsets.push_back(0);
  for (int64_t b = 0; b < nb; ++b) {
    const ArrowArray* a = batches[b];
    if (!a || a->n_children != nc) fail(QHIP_INVALID_ARGUMENT, "batch " + std::to_string(b) + " does not match the schema");
    if (a->null_count > 0) fail(QHIP_UNSUPPORTED, "struct-level nulls in a RecordBatch are not supported");
    t->batch_offsets.push_back(t->batch_offsets.back() + a->length);
  }
  const int64_t N = t->batch_offsets.back();
  t->num_rows = N;
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  std::shared_ptr<HostBatches> hb;
  if (lazy) {
    for (int64_t c = 0; c < nc; ++c) (void)dtype_from_format(schema->children[c]->format);   // unsupported types fail now, not later
    hb = std::make_shared<HostBatches>();
    hb->arrays.resize((size_t)nb);
    for (int64_t b = 0; b < nb; ++b) {
      hb->arrays[(size_t)b] = *batches[b];
      batches[b]->release = nullptr;          // moved
    }
    for (auto& a : hb->arrays) hb->ptrs.push_back(&a);
  }
  for (int64_t c = 0; c < nc; ++c) {
    const ArrowSchema* cs = schema->children[c];
    t->names.push_back(cs->name ? cs->name : "");
    t->nullable.push_back((cs->flags & ARROW_FLAG_NULLABLE) != 0);
    if (!lazy) { t->cols.push_back(upload_column(ctx, cs->format, c, batches, nb, t->batch_offsets)); continue; }
    DevColumn col;
    col.type = dtype_from_format(cs->format);
    col.length = N;
    int64_t maybe_nulls = col.type.id == QHIP_NULL ? N : 0;
    for (int64_t b = 0; b < nb && !maybe_nulls; ++b) if (hb->ptrs[(size_t)b]->children[c]->null_count != 0) maybe_nulls = 1;
    col.null_count = maybe_nulls;   // a may-have-nulls flag until the column is uploaded
    auto u = std::make_shared<DeferredUpload>();
    u->host = hb; u->format = cs->format; u->column = c; u->batch_offsets = t->batch_offsets;
    col.pending_upload = u;
    t->cols.push_back(std::move(col));
  }
  return t.release();
}

// ---------------------------------------------------------------- download
// Page-locked blocks for the big buffers of exported batches (round 4). A result column used to land in freshly malloc'ed
// pageable memory: the runtime stages such a copy through its own page-locked buffers and the fresh pages fault in one by one
// (Q3's 113 k groups, 3.6 MB: ~0.3 ms of a 0.7 ms execute()). A block from this pool is the DMA's destination itself and is handed
// to the consumer as the Arrow buffer; the array's release callback brings it back. Process-wide (arrays may outlive their
// context), size classes of 64 KB << k, at most 512 MB cached and 4 GB handed out — beyond that, and for small buffers, malloc.
struct PinnedPool {
  static constexpr size_t kMin = 64u << 10, kCacheMax = 512u << 20, kOutMax = 4ull << 30;
  std::mutex mu;
  std::vector<void*> free_[16];
  size_t cached = 0, out = 0;
  void* take(size_t n, int& cls) {
    cls = -1;
    if (n < kMin || env_int("QHIP_EXPORT_PAGEABLE", 0) != 0) return nullptr;
    int k = 0;
    size_t c = kMin;
    while (c < n && k < 15) { c <<= 1; ++k; }
    if (c < n) return nullptr;
    {
      std::lock_guard<std::mutex> g(mu);
      if (out + c > kOutMax) return nullptr;
      if (!free_[k].empty()) {
        void* p = free_[k].back();
        free_[k].pop_back();
        cached -= c; out += c; cls = k;
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, c, hipHostMallocDefault) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> g(mu);
    out += c; cls = k;
    return p;
  }
  void give(void* p, int k) {
    const size_t c = kMin << k;
    bool keep;
    {
      std::lock_guard<std::mutex> g(mu);
      out -= c;
      keep = cached + c <= kCacheMax;
      if (keep) { free_[k].push_back(p); cached += c; }
    }
    if (!keep) (void)hipHostFree(p);
  }
};
static PinnedPool& export_pool() { static PinnedPool* pool = new PinnedPool(); return *pool; }   // (never destroyed: release callbacks may run at exit)

struct HostArrayPrivate {
  std::vector<void*> buffers;       // malloc'ed, or a block of the page-locked pool (cls[i] >= 0)
  std::vector<int> cls;             // parallel to `buffers` where set; missing entries = malloc'ed
  std::vector<const void*> buffer_ptrs;
  std::vector<ArrowArray*> children;
  std::vector<ArrowArray> child_storage;
  void push(void* b, int c = -1) { cls.resize(buffers.size(), -1); buffers.push_back(b); cls.push_back(c); }
};
static void release_array(ArrowArray* a) {
  if (!a || !a->release) return;
  HostArrayPrivate* p = (HostArrayPrivate*)a->private_data;
  if (p) {
    for (auto& ch : p->child_storage) if (ch.release) ch.release(&ch);
    for (size_t i = 0; i < p->buffers.size(); ++i) {
      if (i < p->cls.size() && p->cls[i] >= 0) export_pool().give(p->buffers[i], p->cls[i]);
      else free(p->buffers[i]);
    }
    delete p;
  }
  a->release = nullptr;
}
struct HostSchemaPrivate {
  std::string format, name;
  std::vector<ArrowSchema*> children;
  std::vector<ArrowSchema> child_storage;
};
static void release_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  HostSchemaPrivate* p = (HostSchemaPrivate*)s->private_data;
  if (p) {
    for (auto& ch : p->child_storage) if (ch.release) ch.release(&ch);
    delete p;
  }
  s->release = nullptr;
}
static void fill_schema(ArrowSchema* s, const std::string& fmt, const std::string& name, bool nullable, size_t nchildren) {
  HostSchemaPrivate* p = new HostSchemaPrivate();
  p->format = fmt; p->name = name;
  p->child_storage.resize(nchildren);
  for (auto& c : p->child_storage) { memset(&c, 0, sizeof(c)); p->children.push_back(&c); }
  memset(s, 0, sizeof(*s));
  s->format = p->format.c_str();
  s->name = p->name.c_str();
  s->metadata = nullptr;
  s->flags = nullable ? ARROW_FLAG_NULLABLE : 0;
  s->n_children = (int64_t)nchildren;
  s->children = nchildren ? p->children.data() : nullptr;
  s->release = release_schema;
  s->private_data = p;
}

void table_schema_to_arrow(const qhip_table* t, ArrowSchema* out) {
  fill_schema(out, "+s", "", false, t->cols.size());
  for (size_t c = 0; c < t->cols.size(); ++c)
    fill_schema(out->children[c], dtype_to_format(t->cols[c].type), t->names[c], t->nullable[c], 0);
}

static void* xmalloc(size_t n) {
  void* p = nullptr;
  if (posix_memalign(&p, 64, n ? ((n + 63) / 64) * 64 : 64) != 0) throw std::bad_alloc();
  memset(p, 0, n ? ((n + 63) / 64) * 64 : 64);
  return p;
}

void table_batch_to_arrow(Ctx* ctx, const qhip_table* t, int64_t b, ArrowArray* out) {
  if (b < -1 || b >= t->num_batches()) fail(QHIP_INVALID_ARGUMENT, "batch index out of range");
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  // b == -1: every row as ONE array (a caller with thousands of small batches downloads once and slices on the host)
  const int64_t r0 = b < 0 ? 0 : t->offsets()[(size_t)b], r1 = b < 0 ? t->num_rows : t->offsets()[(size_t)b + 1], n = r1 - r0;
  std::unique_ptr<HostArrayPrivate> top(new HostArrayPrivate());
  top->child_storage.resize(t->cols.size());
  for (auto& c : top->child_storage) memset(&c, 0, sizeof(c));
  // Small buffers (a LIMIT's ten rows, a handful of groups) are fetched through the context's page-locked scratch: the
  // copies are queued back to back and land after ONE wait, where a copy into pageable memory is a stream round trip each.
  struct Staged { void* dst; size_t at; size_t n; };
  std::vector<Staged> staged;
  size_t staged_used = 128;   // (the first 128 bytes hold status words)
  const size_t staged_end = ctx->pinned_bytes > 1024 ? ctx->pinned_bytes - 1024 : 0;
  auto land_staged = [&] {
    for (auto& f : staged) memcpy(f.dst, (const uint8_t*)ctx->pinned + f.at, f.n);
    staged.clear();
    staged_used = 128;
  };
  // a big buffer: a block of the page-locked pool when there is one (its class in `cls`), else zeroed pageable memory
  auto big_alloc = [&](size_t nbytes, int& cls) -> void* {
    void* p = export_pool().take(nbytes, cls);
    return p ? p : xmalloc(nbytes);
  };
  auto d2h = [&](void* dst, const void* src, size_t nbytes, hipStream_t st) {
    if (!nbytes) return;
    const size_t room = (nbytes + 15) & ~(size_t)15;
    if (nbytes < PinnedPool::kMin && staged_used + room <= staged_end) {
      QHIP_HIP_CHECK(hipMemcpyAsync((uint8_t*)ctx->pinned + staged_used, src, nbytes, hipMemcpyDeviceToHost, st));
      staged.push_back({dst, staged_used, nbytes});
      staged_used += room;
    } else {
      QHIP_HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, st));
    }
  };
  struct Pending { uint8_t* raw; uint8_t* dst; int64_t bit_off; int64_t nbits; };
  std::vector<Pending> bitfix;                       // bitmaps to realign after the copies land
  struct OffFix { int32_t* off; int64_t n; };
  std::vector<OffFix> offfix;
  std::vector<std::pair<ArrowArray*, uint8_t*>> nullfix;
  for (size_t c = 0; c < t->cols.size(); ++c) {
    // a column an operator assembled on the host and nobody has read on the device (an aggregate's few result rows) is
    // exported from there: uploading it only to copy it back would cost a host -> HBM copy, a wait and an HBM -> host copy
    if (t->cols[c].pending_upload && !t->cols[c].pending_upload->done && t->cols[c].pending_upload->host_col) {
      const HostColumn& hc = *std::static_pointer_cast<HostColumn>(t->cols[c].pending_upload->host_col);
      ArrowArray* ca = &top->child_storage[c];
      HostArrayPrivate* p = new HostArrayPrivate();
      ca->private_data = p;
      ca->release = release_array;
      ca->length = n;
      ca->offset = 0;
      ca->null_count = 0;
      if (hc.type.id == QHIP_NULL) {
        ca->null_count = n; ca->n_buffers = 0; ca->buffers = nullptr;
        top->children.push_back(ca);
        continue;
      }
      uint8_t* validity = nullptr;
      if (hc.null_count > 0 && !hc.validity.empty()) {
        validity = (uint8_t*)xmalloc((size_t)((n + 7) / 8));
        memset(validity, 0, (size_t)((n + 7) / 8));
        if (n) copy_bits(hc.validity.data(), r0, validity, 0, n);
        ca->null_count = n - count_set_bits(validity, 0, n);
        if (ca->null_count == 0) { free(validity); validity = nullptr; }
      }
      p->buffers.push_back(validity);
      const int w = dtype_width(hc.type);
      if (w > 0) {
        uint8_t* v = (uint8_t*)xmalloc((size_t)n * w);
        if (n) memcpy(v, hc.values.data() + (size_t)r0 * w, (size_t)n * w);
        p->buffers.push_back(v);
      } else if (hc.type.id == QHIP_BOOL) {
        uint8_t* v = (uint8_t*)xmalloc((size_t)((n + 7) / 8));
        memset(v, 0, (size_t)((n + 7) / 8));
        if (n) copy_bits(hc.values.data(), r0, v, 0, n);
        p->buffers.push_back(v);
      } else if (hc.type.id == QHIP_UTF8) {
        int32_t* off = (int32_t*)xmalloc((size_t)(n + 1) * 4);
        const int32_t base = hc.offsets.empty() ? 0 : hc.offsets[(size_t)r0];
        for (int64_t i = 0; i <= n; ++i) off[i] = (hc.offsets.empty() ? 0 : hc.offsets[(size_t)(r0 + i)]) - base;
        uint8_t* data = (uint8_t*)xmalloc((size_t)off[n]);
        if (off[n]) memcpy(data, hc.data.data() + base, (size_t)off[n]);
        p->buffers.push_back(off);
        p->buffers.push_back(data);
      }
      for (void* bp : p->buffers) p->buffer_ptrs.push_back(bp);
      ca->n_buffers = (int64_t)p->buffer_ptrs.size();
      ca->buffers = p->buffer_ptrs.data();
      top->children.push_back(ca);
      continue;
    }
    const DevColumn& col = resolved(ctx, t->cols[c]);
    ArrowArray* ca = &top->child_storage[c];
    HostArrayPrivate* p = new HostArrayPrivate();
    ca->private_data = p;
    ca->release = release_array;
    ca->length = n;
    ca->offset = 0;
    ca->null_count = 0;
    uint8_t* validity = nullptr;
    auto fetch_bits = [&](const DevBuf& src) -> uint8_t* {
      uint8_t* dst = (uint8_t*)xmalloc((size_t)((n + 7) / 8));
      if (n == 0) return dst;
      const int64_t byte0 = r0 >> 3, byte1 = (r1 + 7) >> 3;
      if ((r0 & 7) == 0) {
        d2h(dst, (const uint8_t*)src.ptr + byte0, (size_t)((n + 7) / 8), ctx->stream);
      } else {
        uint8_t* raw = (uint8_t*)xmalloc((size_t)(byte1 - byte0));
        d2h(raw, (const uint8_t*)src.ptr + byte0, (size_t)(byte1 - byte0), ctx->stream);
        bitfix.push_back({raw, dst, r0 & 7, n});
      }
      return dst;
    };
    if (col.type.id == QHIP_NULL) {
      ca->null_count = n;
      ca->n_buffers = 0;
      ca->buffers = nullptr;
      top->children.push_back(ca);
      continue;
    }
    if (col.validity && col.null_count > 0) {
      validity = fetch_bits(*col.validity);
      nullfix.push_back({ca, validity});
    }
    p->push(validity);
    const int w = dtype_width(col.type);
    if (w > 0) {
      int cls;
      uint8_t* v = (uint8_t*)big_alloc((size_t)n * w, cls);
      d2h(v, (const uint8_t*)col.values->ptr + (size_t)r0 * w, (size_t)n * w, ctx->stream);
      p->push(v, cls);
    } else if (col.type.id == QHIP_BOOL) {
      p->push(fetch_bits(*col.values));
    } else if (col.type.id == QHIP_UTF8) {
      int ocls, dcls;
      int32_t* off = (int32_t*)big_alloc((size_t)(n + 1) * 4, ocls);
      d2h(off, (const int32_t*)col.values->ptr + r0, (size_t)(n + 1) * 4, ctx->stream);
      QHIP_HIP_CHECK(sync_stream(ctx->stream));   // need the offsets to size the data slice
      land_staged();
      const int32_t base = off[0];
      const int64_t nbytes = (int64_t)off[n] - base;
      // (offsets that run backwards or past the data buffer would turn the copy below into a wild one)
      if (base < 0 || nbytes < 0 || (nbytes > 0 && (!col.data || (size_t)base + (size_t)nbytes > col.data->bytes))) {
        if (ocls >= 0) export_pool().give(off, ocls); else free(off);
        fail(QHIP_INVALID_ARGUMENT, "Utf8 column with offsets outside its data buffer");
      }
      uint8_t* data = (uint8_t*)big_alloc((size_t)nbytes, dcls);
      if (nbytes) d2h(data, (const uint8_t*)col.data->ptr + base, (size_t)nbytes, ctx->stream);
      if (base) offfix.push_back({off, n + 1});
      p->push(off, ocls);
      p->push(data, dcls);
    }
    for (void* bp : p->buffers) p->buffer_ptrs.push_back(bp);
    ca->n_buffers = (int64_t)p->buffer_ptrs.size();
    ca->buffers = p->buffer_ptrs.data();
    top->children.push_back(ca);
  }
  QHIP_HIP_CHECK(sync_stream(ctx->stream));
  land_staged();
  for (auto& f : bitfix) { copy_bits(f.raw, f.bit_off, f.dst, 0, f.nbits); free(f.raw); }
  for (auto& f : offfix) { const int32_t base = f.off[0]; for (int64_t i = 0; i < f.n; ++i) f.off[i] -= base; }
  for (auto& nf : nullfix) nf.first->null_count = n - count_set_bits(nf.second, 0, n);
  memset(out, 0, sizeof(*out));
  out->length = n;
  out->null_count = 0;
  out->offset = 0;
  top->buffers.push_back(nullptr);
  top->buffer_ptrs.push_back(nullptr);
  out->n_buffers = 1;
  out->buffers = top->buffer_ptrs.data();
  out->n_children = (int64_t)top->children.size();
  out->children = top->children.empty() ? nullptr : top->children.data();
  out->release = release_array;
  out->private_data = top.release();
}

}  // namespace qhip

using namespace qhip;
namespace qhip {
qhip_table* table_from_arrow(Ctx*, const ArrowSchema*, ArrowArray* const*, int64_t, bool lazy);
void table_batch_to_arrow(Ctx*, const qhip_table*, int64_t, ArrowArray*);
void table_schema_to_arrow(const qhip_table*, ArrowSchema*);
}

extern "C" {

int qhip_table_from_arrow(qhip_ctx* ctx, const struct ArrowSchema* schema, const struct ArrowArray* const* batches,
                          int64_t n_batches, qhip_table** out) {
  if (!ctx || !out || n_batches < 0 || (n_batches > 0 && !batches)) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = table_from_arrow(ctx, schema, const_cast<ArrowArray* const*>(batches), n_batches, false); });
}

int qhip_table_from_arrow_lazy(qhip_ctx* ctx, const struct ArrowSchema* schema, struct ArrowArray* const* batches, int64_t n_batches,
                               qhip_table** out) {
  if (!ctx || !out || n_batches < 0 || (n_batches > 0 && !batches)) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = table_from_arrow(ctx, schema, batches, n_batches, true); });
}

int qhip_table_to_arrow(qhip_ctx* ctx, const qhip_table* t, int64_t batch_index, struct ArrowArray* out_array,
                        struct ArrowSchema* out_schema) {
  if (!ctx || !t) return QHIP_INVALID_ARGUMENT;
  return guarded(ctx, [&] {
    settle_rows(t);
    if (out_array) table_batch_to_arrow(ctx, t, batch_index, out_array);
    if (out_schema) table_schema_to_arrow(t, out_schema);
  });
}

int qhip_table_batch_offsets(const qhip_table* t, int64_t* out, int64_t n_out) {
  if (!t || !out) return QHIP_INVALID_ARGUMENT;
  try {
    settle_rows(t);
    const std::vector<int64_t>& off = t->offsets();
    if (n_out != (int64_t)off.size()) return QHIP_INVALID_ARGUMENT;
    for (size_t k = 0; k < off.size(); ++k) out[k] = off[k];
    return QHIP_OK;
  } catch (const qhip::Error& e) { return e.code; }
}

int64_t qhip_table_num_batches(const qhip_table* t) {
  if (!t) return -1;
  try { settle_rows(t); return t->num_batches(); } catch (const qhip::Error&) { return -1; }   // (pending boundaries are read from the device here)
}
int64_t qhip_table_num_rows(const qhip_table* t) {
  if (!t) return -1;
  try { settle_rows(t); return t->num_rows; } catch (const qhip::Error&) { return -1; }   // (a join output of deferred size is waited for here)
}
int64_t qhip_table_num_columns(const qhip_table* t) { return t ? (int64_t)t->cols.size() : -1; }
int64_t qhip_table_column_bytes(const qhip_table* t, int64_t col) {
  if (!t || col < 0 || col >= (int64_t)t->cols.size()) return -1;
  return t->cols[(size_t)col].resident_bytes();
}
void qhip_table_destroy(qhip_table* t) {
  if (!t) return;
  if (t->ctx) (void)hipSetDevice(t->ctx->device);
  delete t;
}

}  // extern "C"
