// exchange.cpp — the pieces of the multi-GPU hash-join exchange that run on one GPU:
//   qhip_partition_by_key   split a table into n_parts tables by mix64(join key) so equal keys meet on one rank
//   qhip_table_column_buffer / qhip_table_from_device   raw device buffers for the RCCL all-to-all (moved by torch.distributed)
//   qhip_table_concat       append the tables received from the peers
// The reference has no exchange operator (it is single-process); only the joined RESULT must match it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>

#include "common.hpp"
#include "device/qhip_status.h"
#include "kernels.hpp"
#include "relops.hpp"

using namespace qhip;

namespace {

int log2u(uint32_t x) { int b = 0; while ((1u << b) < x) ++b; return b; }

void partition_by_key(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n_keys, int n_parts,
                      qhip_table** out_parts) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_parts <= 0 || n_parts > 1024 || n_keys <= 0) fail(QHIP_INVALID_ARGUMENT, "qhip_partition_by_key: bad arguments");
  if (in->num_rows >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  hipStream_t s = ctx->stream;
  const uint64_t N = (uint64_t)in->num_rows;
  resolve_all(ctx, in);   // every column travels
  std::vector<InputCol> icols = input_cols_of(in);
  // partition ids must agree across ranks and across the two join sides: Utf8 keys always hash as 4 words (<= 31 bytes)
  for (int k = 0; k < n_keys; ++k)
    if (roots[k] >= 0 && roots[k] < n_exprs && exprs[roots[k]].kind == QHIP_EXPR_COLUMN && exprs[roots[k]].column >= 0 &&
        exprs[roots[k]].column < (int)icols.size() && icols[(size_t)exprs[roots[k]].column].type.id == QHIP_UTF8)
      icols[(size_t)exprs[roots[k]].column].utf8_max_len = 31;
  ExprSet es;
  es.build(exprs, n_exprs, icols);
  KeysPlan kp;
  DevBuf keys, valid;
  uint32_t* const dstat = zeroed_block(ctx);   // the key expressions' status words: read back with the histogram (ONE wait)
  eval_key_words(ctx, in, es, icols, roots, n_keys, kp, keys, valid, -1, true, dstat);
  DevBuf part((N + 1) * 4), hist((size_t)(n_parts + 1) * 4), iota((N + 1) * 4), sorted_part((N + 1) * 4), sorted_rows((N + 1) * 4);
  QHIP_HIP_CHECK(hipMemsetAsync(hist.ptr, 0, hist.bytes, s));
  launch_partition_ids(kp.W, keys.as<uint64_t>(), N, (uint32_t)n_parts, part.as<uint32_t>(), hist.as<uint32_t>(), s);
  launch_iota_u32(iota.as<uint32_t>(), N, s);
  stable_sort_pairs_u32(part.as<uint32_t>(), sorted_part.as<uint32_t>(), iota.as<uint32_t>(), sorted_rows.as<uint32_t>(), N,
                        std::max(1, log2u((uint32_t)n_parts)), s);
  uint32_t* const back = (uint32_t*)ctx->pinned;   // [status words | histogram]
  if ((size_t)(QS_WORDS + n_parts) * 4 > ctx->pinned_bytes) fail(QHIP_UNSUPPORTED, "qhip_partition_by_key: too many parts for the read-back scratch");
  QHIP_HIP_CHECK(hipMemcpyAsync(back, dstat, QS_WORDS * 4, hipMemcpyDeviceToHost, s));
  QHIP_HIP_CHECK(hipMemcpyAsync(back + QS_WORDS, hist.ptr, (size_t)n_parts * 4, hipMemcpyDeviceToHost, s));
  QHIP_HIP_CHECK(sync_stream(s));
  check_status_words(back);
  std::vector<uint32_t> h(back + QS_WORDS, back + QS_WORDS + n_parts);
  // the parts' columns: every plain column (fixed width, no NULLs) of every part goes into batched gather launches (8 gathers
  // each) instead of one launch per (part, column); the others (validity bits, strings) take their own gathers
  GatherBatch gb;
  memset(&gb, 0, sizeof gb);
  int n_batched = 0;
  auto flush = [&] { if (n_batched) { launch_gather_multi(gb, n_batched, s); n_batched = 0; } };
  uint64_t pos = 0;
  for (int p = 0; p < n_parts; ++p) {
    std::unique_ptr<qhip_table> t(new qhip_table());
    t->ctx = ctx;
    t->names = in->names;
    t->nullable = in->nullable;
    const uint64_t m = h[(size_t)p];
    for (auto& c : in->cols) {
      const DevColumn& src = resolved(ctx, c);
      const int w = dtype_width(src.type);
      if (w > 0 && src.null_count == 0 && m > 0) {
        DevColumn oc;
        oc.type = src.type;
        oc.length = (int64_t)m;
        oc.value_maxabs = src.value_maxabs;
        oc.range = src.range; oc.range_inherited = true;
        oc.values = std::make_shared<DevBuf>((size_t)m * (size_t)w);
        gb.d[n_batched++] = GatherDesc{src.values->ptr, sorted_rows.as<uint32_t>() + pos, oc.values->ptr, m, (uint32_t)w, 0u};
        if (n_batched == kGatherBatch) flush();
        t->cols.push_back(std::move(oc));
      } else
        t->cols.push_back(gather_column(ctx, c, sorted_rows.as<uint32_t>() + pos, m, false));
    }
    t->num_rows = (int64_t)m;
    t->batch_offsets = {0, (int64_t)m};
    pos += m;
    out_parts[p] = t.release();
  }
  flush();
}

// One destination column assembled from n source sections on the device (shared by qhip_table_concat and the unpacking of
// the exchange's wire images): fixed-width values are D2D copies, bitmaps are appended bit-exactly at any bit position
// (k_bits_append), Utf8 offsets are copied and rebased by the bytes already placed (every device Utf8 column's offsets
// start at 0: uploads rebase them and every other column is a gather). Nothing goes through the host.
struct ColumnSection {
  int64_t rows = 0, null_count = 0, data_bytes = 0;
  const void* values = nullptr;     // fixed-width values | int32 offsets (rows + 1) | Boolean bits
  const void* validity = nullptr;   // bitmap or null (no NULLs in this section)
  const void* data = nullptr;       // utf8 bytes
};

DevColumn assemble_column(Ctx* ctx, DType type, const std::vector<ColumnSection>& secs) {
  hipStream_t s = ctx->stream;
  DevColumn oc;
  oc.type = type;
  int64_t N = 0, nulls = 0, bytes = 0;
  for (const auto& sc : secs) { N += sc.rows; nulls += sc.null_count; bytes += sc.data_bytes; }
  oc.length = N;
  oc.null_count = nulls;
  if (type.id == QHIP_NULL) { oc.null_count = N; return oc; }
  const int w = dtype_width(type);
  auto bitmap = [&](bool values) {
    auto b = std::make_shared<DevBuf>((size_t)((N + 63) / 64) * 8 + 8);
    QHIP_HIP_CHECK(hipMemsetAsync(b->ptr, 0, b->bytes, s));
    int64_t pos = 0;
    for (const auto& sc : secs) {
      const void* src = values ? sc.values : sc.validity;
      if (values && !src && sc.rows) fail(QHIP_INVALID_ARGUMENT, "Boolean column section without a values buffer");
      launch_bits_append(b->as<uint32_t>(), (uint64_t)pos, (const uint8_t*)src, (uint64_t)sc.rows, s);
      pos += sc.rows;
    }
    return b;
  };
  if (nulls > 0) oc.validity = bitmap(false);
  if (w > 0) {
    oc.values = std::make_shared<DevBuf>((size_t)N * w);
    int64_t pos = 0;
    for (const auto& sc : secs) {
      if (sc.rows) QHIP_HIP_CHECK(hipMemcpyAsync((uint8_t*)oc.values->ptr + (size_t)pos * w, sc.values, (size_t)sc.rows * w, hipMemcpyDeviceToDevice, s));
      pos += sc.rows;
    }
  } else if (type.id == QHIP_BOOL) {
    oc.values = bitmap(true);
  } else if (type.id == QHIP_UTF8) {
    if (bytes > 0x7fffffffLL) fail(QHIP_UNSUPPORTED, "Utf8 column larger than 2 GiB");
    oc.values = std::make_shared<DevBuf>((size_t)(N + 1) * 4);
    oc.data = std::make_shared<DevBuf>((size_t)bytes);
    oc.data_bytes = bytes;
    QHIP_HIP_CHECK(hipMemsetAsync(oc.values->ptr, 0, 4, s));   // N == 0, or leading empty sections: offsets[0] = 0
    int64_t pos = 0, bpos = 0;
    for (const auto& sc : secs) {
      if (!sc.rows) continue;
      // rows + 1 offsets: the last one lands on the next section's first slot, which that section rewrites with the same value
      QHIP_HIP_CHECK(hipMemcpyAsync(oc.values->as<int32_t>() + pos, sc.values, (size_t)(sc.rows + 1) * 4, hipMemcpyDeviceToDevice, s));
      launch_add_i32(oc.values->as<int32_t>() + pos, (uint64_t)sc.rows + 1, (int32_t)bpos, s);
      if (sc.data_bytes) QHIP_HIP_CHECK(hipMemcpyAsync((uint8_t*)oc.data->ptr + bpos, sc.data, (size_t)sc.data_bytes, hipMemcpyDeviceToDevice, s));
      pos += sc.rows;
      bpos += sc.data_bytes;
    }
  }
  return oc;
}

qhip_table* table_concat(Ctx* ctx, const qhip_table* const* ts, int n) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  if (n <= 0) fail(QHIP_INVALID_ARGUMENT, "qhip_table_concat: no tables");
  const qhip_table* first = ts[0];
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->names = first->names;
  out->nullable = first->nullable;
  out->batch_offsets.push_back(0);
  int64_t N = 0;
  for (int k = 0; k < n; ++k) resolve_all(ctx, ts[k]);
  for (int k = 0; k < n; ++k) {
    if (ts[k]->cols.size() != first->cols.size()) fail(QHIP_INVALID_ARGUMENT, "qhip_table_concat: schemas differ");
    for (int64_t b = 0; b < ts[k]->num_batches(); ++b) out->batch_offsets.push_back(N + ts[k]->offsets()[(size_t)b + 1]);
    N += ts[k]->num_rows;
  }
  out->num_rows = N;
  if (N >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  for (size_t c = 0; c < first->cols.size(); ++c) {
    std::vector<ColumnSection> secs((size_t)n);
    for (int k = 0; k < n; ++k) {
      const DevColumn& sc = ts[k]->cols[c];
      if (sc.type != first->cols[c].type) fail(QHIP_INVALID_ARGUMENT, "qhip_table_concat: column types differ");
      ColumnSection& o = secs[(size_t)k];
      o.rows = sc.length;
      o.null_count = sc.null_count;
      o.data_bytes = sc.data_bytes;
      o.values = sc.values ? sc.values->ptr : nullptr;
      o.validity = sc.null_count > 0 && sc.validity ? sc.validity->ptr : nullptr;
      o.data = sc.data ? sc.data->ptr : nullptr;
    }
    out->cols.push_back(assemble_column(ctx, first->cols[c].type, secs));
  }
  QHIP_HIP_CHECK(sync_stream(ctx->stream));
  return out.release();
}

// ---- wire image of a table for the exchange: ONE contiguous buffer per (source, destination) pair, so an exchange is one
// small metadata round plus one payload round whatever the number of columns. Sections in column order — values,
// validity, utf8 data — each aligned to 16 bytes; both sides derive the layout from the schema and the table's metadata
// words [rows, image bytes, (null_count, data_bytes) per column], so the image carries no header of its own.
inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct WireColumn { size_t values_off = 0, values_bytes = 0, validity_off = 0, validity_bytes = 0, data_off = 0, data_bytes = 0; };

size_t wire_layout(const std::vector<DType>& types, int64_t rows, const int64_t* col_meta, std::vector<WireColumn>& out) {
  size_t pos = 0;
  out.assign(types.size(), WireColumn());
  for (size_t c = 0; c < types.size(); ++c) {
    WireColumn& wc = out[c];
    const int w = dtype_width(types[c]);
    const int64_t nulls = col_meta[2 * c], dbytes = col_meta[2 * c + 1];
    if (rows > 0 && types[c].id != QHIP_NULL) {
      wc.values_bytes = w > 0 ? (size_t)rows * w : types[c].id == QHIP_BOOL ? (size_t)((rows + 7) / 8) : (size_t)(rows + 1) * 4;
      wc.validity_bytes = nulls > 0 ? (size_t)((rows + 7) / 8) : 0;
      wc.data_bytes = types[c].id == QHIP_UTF8 ? (size_t)dbytes : 0;
    }
    wc.values_off = pos; pos = align16(pos + wc.values_bytes);
    wc.validity_off = pos; pos = align16(pos + wc.validity_bytes);
    wc.data_off = pos; pos = align16(pos + wc.data_bytes);
  }
  return pos;
}

std::vector<DType> types_of(const qhip_table* t) {
  std::vector<DType> out;
  for (const auto& c : t->cols) out.push_back(c.type);
  return out;
}

void table_wire_meta(Ctx* ctx, const qhip_table* t, int64_t* meta) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  resolve_all(ctx, t);   // every column travels; null counts are exact once gathered
  meta[0] = t->num_rows;
  for (size_t c = 0; c < t->cols.size(); ++c) {
    meta[2 + 2 * c] = t->cols[c].validity ? t->cols[c].null_count : 0;
    meta[3 + 2 * c] = t->cols[c].data_bytes;
  }
  std::vector<WireColumn> lay;
  meta[1] = (int64_t)wire_layout(types_of(t), t->num_rows, meta + 2, lay);
}

void table_pack(Ctx* ctx, const qhip_table* t, void* dst, int64_t dst_bytes, bool sync = true) {
  std::vector<int64_t> meta(2 + 2 * t->cols.size());
  table_wire_meta(ctx, t, meta.data());
  if (dst_bytes < meta[1]) fail(QHIP_INVALID_ARGUMENT, "qhip_table_pack: destination smaller than the wire image");
  std::vector<WireColumn> lay;
  wire_layout(types_of(t), t->num_rows, meta.data() + 2, lay);
  hipStream_t s = ctx->stream;
  auto put = [&](size_t off, size_t nbytes, const std::shared_ptr<DevBuf>& src) {
    if (!nbytes) return;
    if (!src || src->bytes < nbytes) fail(QHIP_INVALID_ARGUMENT, "qhip_table_pack: column buffer missing or short");
    QHIP_HIP_CHECK(hipMemcpyAsync((uint8_t*)dst + off, src->ptr, nbytes, hipMemcpyDeviceToDevice, s));
  };
  for (size_t c = 0; c < t->cols.size(); ++c) {
    put(lay[c].values_off, lay[c].values_bytes, t->cols[c].values);
    put(lay[c].validity_off, lay[c].validity_bytes, t->cols[c].validity);
    put(lay[c].data_off, lay[c].data_bytes, t->cols[c].data);
  }
  if (sync) QHIP_HIP_CHECK(sync_stream(s));   // the transport reads the image on its own stream (RCCL inside libqhip runs on this one)
}

qhip_table* table_unpack_concat(Ctx* ctx, const char* const* names, const qhip_dtype* dtypes, int n_cols, const int64_t* metas,
                                const void* const* images, int n, bool sync = true) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  if (n <= 0 || n_cols < 0) fail(QHIP_INVALID_ARGUMENT, "qhip_table_unpack_concat: bad arguments");
  std::vector<DType> types;
  for (int c = 0; c < n_cols; ++c) types.push_back(DType(dtypes[c]));
  const size_t M = 2 + 2 * (size_t)n_cols;
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->batch_offsets.push_back(0);
  std::vector<std::vector<WireColumn>> lays((size_t)n);
  int64_t N = 0;
  for (int k = 0; k < n; ++k) {
    const int64_t* m = metas + (size_t)k * M;
    if (m[0] < 0 || (size_t)m[1] != wire_layout(types, m[0], m + 2, lays[(size_t)k]) || (m[1] > 0 && !images[k]))
      fail(QHIP_INVALID_ARGUMENT, "qhip_table_unpack_concat: metadata does not describe the wire image");
    N += m[0];
    out->batch_offsets.push_back(N);   // one batch per peer, in rank order
  }
  if (N >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  out->num_rows = N;
  for (int c = 0; c < n_cols; ++c) {
    std::vector<ColumnSection> secs((size_t)n);
    for (int k = 0; k < n; ++k) {
      const int64_t* m = metas + (size_t)k * M;
      const WireColumn& wc = lays[(size_t)k][(size_t)c];
      const uint8_t* base = (const uint8_t*)images[k];
      ColumnSection& o = secs[(size_t)k];
      o.rows = m[0];
      o.null_count = types[(size_t)c].id == QHIP_NULL ? m[0] : m[2 + 2 * c];
      o.data_bytes = (int64_t)wc.data_bytes;
      o.values = wc.values_bytes ? base + wc.values_off : nullptr;
      o.validity = wc.validity_bytes ? base + wc.validity_off : nullptr;
      o.data = wc.data_bytes ? base + wc.data_off : nullptr;
    }
    out->cols.push_back(assemble_column(ctx, types[(size_t)c], secs));
    out->names.push_back(names && names[c] ? names[c] : ("c" + std::to_string(c)));
    out->nullable.push_back(true);
  }
  if (sync) QHIP_HIP_CHECK(sync_stream(ctx->stream));   // the caller frees the images when this returns (images from the stream-ordered pool need no wait)
  return out.release();
}

// Projection pushdown through the exchange: the columns no ancestor of the exchange reads become NULL-typed columns (no
// buffers, zero bytes on the wire) at the same positions, so column indices downstream stay valid; the kept columns share
// their buffers — and their deferred state, so a dropped column's pending gather or upload never happens.
qhip_table* table_keep_columns(Ctx* ctx, const qhip_table* in, const int32_t* keep) {
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->names = in->names;
  out->nullable = in->nullable;
  out->num_rows = in->num_rows;
  // (batch boundaries an operator left on the device stay there: whoever asks the view for them fetches them itself)
  out->batch_offsets = in->batch_offsets;
  out->pending_offsets = in->pending_offsets;
  for (size_t c = 0; c < in->cols.size(); ++c) {
    if (keep[c]) { out->cols.push_back(in->cols[c]); continue; }
    DevColumn nc;
    nc.type = DType{QHIP_NULL, 0, 0};
    nc.length = in->num_rows;
    nc.null_count = in->num_rows;
    out->cols.push_back(std::move(nc));
    out->nullable[c] = true;
  }
  return out.release();
}

qhip_table* table_from_device(Ctx* ctx, const char* const* names, const qhip_device_column* cols, int n_cols, int64_t n_rows) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  std::unique_ptr<qhip_table> t(new qhip_table());
  t->ctx = ctx;
  t->num_rows = n_rows;
  t->batch_offsets = {0, n_rows};
  for (int c = 0; c < n_cols; ++c) {
    const qhip_device_column& dc = cols[c];
    if (dc.length != n_rows) fail(QHIP_INVALID_ARGUMENT, "qhip_table_from_device: column length differs from n_rows");
    DevColumn col;
    col.type = DType(dc.dtype);
    col.length = n_rows;
    col.null_count = dc.null_count;
    auto copy = [&](const void* src, size_t nbytes) {
      auto b = std::make_shared<DevBuf>(nbytes);
      if (nbytes && src) QHIP_HIP_CHECK(hipMemcpyAsync(b->ptr, src, nbytes, hipMemcpyDeviceToDevice, s));
      return b;
    };
    const int w = dtype_width(col.type);
    if (dc.null_count > 0 && dc.validity) col.validity = copy(dc.validity, (size_t)((n_rows + 7) / 8));
    if (w > 0) col.values = copy(dc.values, (size_t)n_rows * w);
    else if (col.type.id == QHIP_BOOL) col.values = copy(dc.values, (size_t)((n_rows + 7) / 8));
    else if (col.type.id == QHIP_UTF8) {
      col.values = copy(dc.values, (size_t)(n_rows + 1) * 4);
      col.data = copy(dc.data, (size_t)dc.data_bytes);
      col.data_bytes = dc.data_bytes;
    }
    t->cols.push_back(std::move(col));
    t->names.push_back(names && names[c] ? names[c] : ("c" + std::to_string(c)));
    t->nullable.push_back(true);
  }
  QHIP_HIP_CHECK(sync_stream(s));
  return t.release();
}


// ================================================================ RCCL inside libqhip
// The exchange of a repartitioned / broadcast hash join behind the C ABI: a host that has no torch (the reference's is
// Rust) creates a communicator from an ncclUniqueId it distributed itself and calls qhip_exchange_tables /
// qhip_all_gather_table. librccl is dlopen'ed at first use (an already loaded copy — torch's — is reused), so the
// library builds, links and runs single-GPU without RCCL on the machine. Everything is enqueued on the context's own
// stream: packing, the metadata all-gather, the grouped ncclSend / ncclRecv (xGMI is point-to-point: one group puts all
// 7 links of a GPU to work at once) and the unpacking are stream-ordered, and the ONE host wait of an exchange is the
// read-back of the received parts' sizes.
typedef struct { char internal[128]; } qh_nccl_unique_id;
typedef void* qh_nccl_comm;
enum { QH_NCCL_UINT8 = 1, QH_NCCL_INT64 = 4 };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(qh_nccl_unique_id*) = nullptr;
  int (*CommInitRank)(qh_nccl_comm*, int, qh_nccl_unique_id, int) = nullptr;
  int (*CommDestroy)(qh_nccl_comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, qh_nccl_comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, qh_nccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, qh_nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  std::string error;
};
Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {getenv("QHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) if (n && *n && !x.lib) x.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already holds
    for (const char* n : names) if (n && *n && !x.lib) x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!x.lib) { x.error = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "not found"); return x; }
    auto sym = [&](const char* name) { void* p = dlsym(x.lib, name); if (!p && x.error.empty()) x.error = std::string("librccl lacks ") + name; return p; };
    x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId");
    x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
    x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
    x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
    x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
    x.Send = (decltype(x.Send))sym("ncclSend");
    x.Recv = (decltype(x.Recv))sym("ncclRecv");
    x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
    x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
    x.GetVersion = (decltype(x.GetVersion))sym("ncclGetVersion");
    return x;
  }();
  if (!r.error.empty()) fail(QHIP_RCCL_ERROR, r.error);
  return r;
}
void rccl_check(int rc, const char* what) {
  if (rc != 0) fail(QHIP_RCCL_ERROR, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
}  // namespace

struct qhip_comm {
  Ctx* ctx = nullptr;
  qh_nccl_comm comm = nullptr;   // nullptr for a world of one (nothing to talk to: no RCCL call is made)
  int rank = 0, world = 1;
  qhip_comm_stats stats;
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool timed = false;            // ev[0] .. ev[1] bracket the last exchange's transfers, not yet added to stats.seconds
};

namespace {
// add the last exchange's transfer time to the statistics; `wait`: block for its end event (qhip_comm_get_stats), else only
// when the event has already happened (the next exchange: never a host wait — a sample that is not ready yet is dropped)
void comm_settle_time(qhip_comm* c, bool wait) {
  if (!c->timed) return;
  float ms = 0;
  const hipError_t ready = wait ? sync_event(c->ev[1]) : hipEventQuery(c->ev[1]);
  if (ready == hipSuccess && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->stats.transfer_seconds += (double)ms * 1e-3;
  else (void)hipGetLastError();
  c->timed = false;
}

// parts[r] goes to rank r (n = world parts) — or, all_gather, the ONE table parts[0] goes to every rank; the result is the
// concatenation, in rank order, of what every rank sent here
qhip_table* comm_exchange(Ctx* ctx, qhip_comm* c, const qhip_table* const* parts, bool all_gather, const char* const* names,
                          const qhip_dtype* dtypes, int n_cols) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int W = c->world, me = c->rank;
  // (QHIP_COMM_SELF_RCCL=1, tests: this rank's own part travels through RCCL too — ncclSend / ncclRecv to itself inside the
  // group, the metadata through ncclAllGather — so that a ONE-rank communicator exercises every RCCL entry point)
  const bool self_rccl = c->comm != nullptr && env_int("QHIP_COMM_SELF_RCCL", 0) != 0;
  // metadata words per part: the wire image's [rows, bytes, (nulls, utf8 bytes) per column] + per column [range known, min,
  // max] — a join above the exchange then knows its integer key's value range (the dense table layout) without a
  // reduction and a host wait of its own over the freshly unpacked column
  const size_t M0 = 2 + 2 * (size_t)n_cols, M = M0 + 3 * (size_t)n_cols;
  const int n_mine = all_gather ? 1 : W;
  for (int p = 0; p < n_mine; ++p) {
    if (!parts[p] || (int)parts[p]->cols.size() != n_cols) fail(QHIP_INVALID_ARGUMENT, "exchange: a part is missing or has a different number of columns");
    settle_rows(parts[p]);
  }
  comm_settle_time(c, false);
  // ---- my parts' metadata and ONE send buffer with their wire images back to back (no wait: everything stays on the stream)
  std::vector<int64_t> meta_out((size_t)W * M);
  for (int p = 0; p < n_mine; ++p) {
    int64_t* m = meta_out.data() + (size_t)p * M;
    table_wire_meta(ctx, parts[p], m);
    for (int col = 0; col < n_cols; ++col) {
      const DevColumn& dc = parts[p]->cols[(size_t)col];
      const bool known = dc.range && dc.range->known;
      m[M0 + 3 * col] = known ? 1 : 0;
      m[M0 + 3 * col + 1] = known ? dc.range->min : 0;
      m[M0 + 3 * col + 2] = known ? dc.range->max : 0;
    }
  }
  if (all_gather) for (int p = 1; p < W; ++p) memcpy(meta_out.data() + (size_t)p * M, meta_out.data(), M * 8);
  std::vector<size_t> send_off((size_t)W + 1, 0);
  for (int p = 0; p < n_mine; ++p) send_off[(size_t)p + 1] = send_off[(size_t)p] + align16((size_t)meta_out[(size_t)p * M + 1]);
  DevBuf send_buf(send_off[(size_t)n_mine]);
  for (int p = 0; p < n_mine; ++p)
    table_pack(ctx, parts[p], send_buf.as<uint8_t>() + send_off[(size_t)p], meta_out[(size_t)p * M + 1], false);
  auto send_ptr = [&](int dst) { return send_buf.as<uint8_t>() + (all_gather ? 0 : send_off[(size_t)dst]); };
  // ---- what will I receive? every rank's metadata rows, all-gathered (fixed size: world x M words per rank); the one wait
  std::vector<int64_t> meta_in((size_t)W * M);   // row r: the part rank r sends HERE
  if (W == 1 && !self_rccl) memcpy(meta_in.data(), meta_out.data(), M * 8);
  else {
    const size_t words = (size_t)W * M;
    DevBuf dmine(words * 8), dall(words * 8 * (size_t)W);
    int64_t* stage = (int64_t*)ctx->pinned;
    if ((words * (size_t)(W + 1)) * 8 > ctx->pinned_bytes) fail(QHIP_UNSUPPORTED, "exchange: metadata of this many columns x ranks exceeds the read-back scratch");
    memcpy(stage, meta_out.data(), words * 8);
    QHIP_HIP_CHECK(hipMemcpyAsync(dmine.ptr, stage, words * 8, hipMemcpyHostToDevice, s));
    rccl_check(rccl().AllGather(dmine.ptr, dall.ptr, words, QH_NCCL_INT64, c->comm, s), "ncclAllGather (exchange metadata)");
    int64_t* all = stage + words;
    QHIP_HIP_CHECK(hipMemcpyAsync(all, dall.ptr, words * 8 * (size_t)W, hipMemcpyDeviceToHost, s));
    QHIP_HIP_CHECK(sync_stream(s));
    ++c->stats.host_waits;
    for (int r = 0; r < W; ++r) memcpy(meta_in.data() + (size_t)r * M, all + ((size_t)r * W + (size_t)me) * M, M * 8);
  }
  // ---- the images: one group of point-to-point operations, my own part by a device copy
  std::vector<size_t> recv_off((size_t)W + 1, 0);
  for (int r = 0; r < W; ++r) recv_off[(size_t)r + 1] = recv_off[(size_t)r] + align16((size_t)meta_in[(size_t)r * M + 1]);
  DevBuf recv_buf(recv_off[(size_t)W]);
  QHIP_HIP_CHECK(hipEventRecord(c->ev[0], s));
  const size_t own = (size_t)meta_in[(size_t)me * M + 1];
  if (own && !self_rccl) QHIP_HIP_CHECK(hipMemcpyAsync(recv_buf.as<uint8_t>() + recv_off[(size_t)me], send_ptr(me), own, hipMemcpyDeviceToDevice, s));
  if (W > 1 || self_rccl) {
    rccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (int r = 0; r < W; ++r) {
      if (r == me && !self_rccl) continue;
      const size_t out_bytes = (size_t)meta_out[(size_t)r * M + 1], in_bytes = (size_t)meta_in[(size_t)r * M + 1];
      if (out_bytes) { rccl_check(rccl().Send(send_ptr(r), out_bytes, QH_NCCL_UINT8, r, c->comm, s), "ncclSend"); c->stats.bytes_sent += out_bytes; }
      if (in_bytes) { rccl_check(rccl().Recv(recv_buf.as<uint8_t>() + recv_off[(size_t)r], in_bytes, QH_NCCL_UINT8, r, c->comm, s), "ncclRecv"); c->stats.bytes_received += in_bytes; }
    }
    rccl_check(rccl().GroupEnd(), "ncclGroupEnd");
  }
  QHIP_HIP_CHECK(hipEventRecord(c->ev[1], s));
  c->timed = true;
  for (int p = 0; p < W; ++p) c->stats.bytes_packed += (uint64_t)meta_out[(size_t)p * M + 1];
  ++c->stats.exchanges;
  // ---- unpack straight into the concatenated table (stream-ordered: the buffers go back to the pool behind it)
  std::vector<const void*> images((size_t)W);
  for (int r = 0; r < W; ++r) images[(size_t)r] = meta_in[(size_t)r * M + 1] ? recv_buf.as<uint8_t>() + recv_off[(size_t)r] : nullptr;
  std::vector<int64_t> wire_meta((size_t)W * M0);
  for (int r = 0; r < W; ++r) memcpy(wire_meta.data() + (size_t)r * M0, meta_in.data() + (size_t)r * M, M0 * 8);
  qhip_table* out = table_unpack_concat(ctx, names, dtypes, n_cols, wire_meta.data(), images.data(), W, false);
  for (int col = 0; col < n_cols; ++col) {   // the union of the parts' ranges (parts without rows say nothing)
    bool all = true, any = false;
    int64_t lo = 0, hi = 0;
    for (int r = 0; r < W; ++r) {
      const int64_t* m = meta_in.data() + (size_t)r * M;
      if (m[0] == 0) continue;
      if (!m[M0 + 3 * col]) { all = false; break; }
      lo = any ? std::min(lo, m[M0 + 3 * col + 1]) : m[M0 + 3 * col + 1];
      hi = any ? std::max(hi, m[M0 + 3 * col + 2]) : m[M0 + 3 * col + 2];
      any = true;
    }
    if (all && any) {
      DevColumn& dc = out->cols[(size_t)col];
      dc.range = std::make_shared<ColRange>();
      dc.range->known = true; dc.range->min = lo; dc.range->max = hi;
      dc.range_inherited = false;
    }
  }
  return out;
}
}  // namespace

extern "C" {

int qhip_comm_unique_id(void* id_out, size_t id_bytes) {
  if (!id_out || id_bytes < 128) return QHIP_INVALID_ARGUMENT;
  return guarded(nullptr, [&] {
    qh_nccl_unique_id id;
    rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, 128);
  });
}

int qhip_comm_create(qhip_ctx* ctx, const void* unique_id, int32_t rank, int32_t world, qhip_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id)) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] {
    QHIP_HIP_CHECK(hipSetDevice(ctx->device));
    std::unique_ptr<qhip_comm> c(new qhip_comm());
    c->ctx = ctx; c->rank = rank; c->world = world;
    memset(&c->stats, 0, sizeof c->stats);
    for (auto& e : c->ev) QHIP_HIP_CHECK(hipEventCreate(&e));
    if (world > 1 || env_int("QHIP_COMM_FORCE_RCCL", 0) != 0) {
      // (QHIP_COMM_FORCE_RCCL=1: a ONE-rank communicator is a real RCCL communicator too — the one-GPU rehearsal of the init path)
      qh_nccl_unique_id id;
      if (unique_id) memcpy(&id, unique_id, 128);
      else rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
      rccl_check(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
      int v = 0;
      if (rccl().GetVersion && rccl().GetVersion(&v) == 0) c->stats.rccl_version = v;
    }
    *out = c.release();
  });
}

void qhip_comm_destroy(qhip_comm* c) {
  if (!c) return;
  if (c->ctx) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  delete c;
}

int qhip_comm_get_stats(qhip_comm* c, qhip_comm_stats* out, int32_t reset) {
  if (!c || !out) return QHIP_INVALID_ARGUMENT;
  comm_settle_time(c, true);
  *out = c->stats;
  out->rank = c->rank; out->world = c->world;
  if (reset) { const int32_t v = c->stats.rccl_version; memset(&c->stats, 0, sizeof c->stats); c->stats.rccl_version = v; }
  return QHIP_OK;
}

int qhip_exchange_tables(qhip_ctx* ctx, qhip_comm* comm, const qhip_table* const* parts, const char* const* names, const qhip_dtype* dtypes,
                         int32_t n_cols, qhip_table** out) {
  if (!ctx || !comm || !parts || !out || n_cols < 0 || (n_cols > 0 && !dtypes) || comm->ctx != ctx) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = comm_exchange(ctx, comm, parts, false, names, dtypes, n_cols); });
}

int qhip_all_gather_table(qhip_ctx* ctx, qhip_comm* comm, const qhip_table* t, const char* const* names, const qhip_dtype* dtypes, int32_t n_cols,
                          qhip_table** out) {
  if (!ctx || !comm || !t || !out || n_cols < 0 || (n_cols > 0 && !dtypes) || comm->ctx != ctx) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = comm_exchange(ctx, comm, &t, true, names, dtypes, n_cols); });
}

}  // extern "C"

namespace {
}  // namespace

extern "C" {

int qhip_partition_by_key(qhip_ctx* ctx, const qhip_table* input, const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots,
                          int32_t n_keys, int32_t n_parts, qhip_table** out_parts) {
  if (!ctx || !input || !out_parts) return QHIP_INVALID_ARGUMENT;
  for (int p = 0; p < n_parts; ++p) out_parts[p] = nullptr;
  int rc = guarded(ctx, [&] { settle_rows(input); partition_by_key(ctx, input, exprs, n_exprs, key_roots, n_keys, n_parts, out_parts); });
  if (rc != QHIP_OK)
    for (int p = 0; p < n_parts; ++p) { if (out_parts[p]) { delete out_parts[p]; out_parts[p] = nullptr; } }
  return rc;
}

int qhip_table_concat(qhip_ctx* ctx, const qhip_table* const* tables, int32_t n, qhip_table** out) {
  if (!ctx || !tables || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { for (int k = 0; k < n; ++k) settle_rows(tables[k]); *out = table_concat(ctx, tables, n); });
}

int qhip_table_keep_columns(qhip_ctx* ctx, const qhip_table* t, const int32_t* keep, int32_t n_cols, qhip_table** out) {
  if (!ctx || !t || !keep || !out || n_cols != (int32_t)t->cols.size()) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { settle_rows(t); *out = table_keep_columns(ctx, t, keep); });
}

int qhip_table_stride_sample(qhip_ctx* ctx, const qhip_table* t, int64_t stride, qhip_table** out) {
  if (!ctx || !t || !out || stride < 1 || stride > 0x7fffffff) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] {
    QHIP_HIP_CHECK(hipSetDevice(ctx->device));
    settle_rows(t);
    const uint64_t m = ((uint64_t)t->num_rows + (uint64_t)stride - 1) / (uint64_t)stride;
    std::unique_ptr<qhip_table> o(new qhip_table());
    o->ctx = ctx;
    o->names = t->names;
    o->nullable = t->nullable;
    o->num_rows = (int64_t)m;
    o->batch_offsets = {0, (int64_t)m};
    auto idx = std::make_shared<DevBuf>((m + 1) * 4);
    launch_iota_stride_u32(idx->as<uint32_t>(), m, (uint32_t)stride, ctx->stream);
    defer_gather(ctx, t->cols, idx, m, false, o->cols);
    *out = o.release();
  });
}

int qhip_table_wire_meta(qhip_ctx* ctx, const qhip_table* t, int64_t* meta, int32_t n_meta) {
  if (!ctx || !t || !meta || n_meta != (int32_t)(2 + 2 * t->cols.size())) return QHIP_INVALID_ARGUMENT;
  return guarded(ctx, [&] { settle_rows(t); table_wire_meta(ctx, t, meta); });
}

int qhip_table_pack(qhip_ctx* ctx, const qhip_table* t, void* device_dst, int64_t dst_bytes) {
  if (!ctx || !t || (!device_dst && dst_bytes > 0)) return QHIP_INVALID_ARGUMENT;
  return guarded(ctx, [&] { settle_rows(t); table_pack(ctx, t, device_dst, dst_bytes); });
}

int qhip_table_unpack_concat(qhip_ctx* ctx, const char* const* names, const qhip_dtype* dtypes, int32_t n_cols, const int64_t* metas,
                             const void* const* device_images, int32_t n, qhip_table** out) {
  if (!ctx || !metas || !device_images || !out || (n_cols > 0 && !dtypes)) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = table_unpack_concat(ctx, names, dtypes, n_cols, metas, device_images, n); });
}

int qhip_table_column_buffer(const qhip_table* t, int64_t col, int32_t which, void** device_ptr, int64_t* n_bytes) {
  if (!t || col < 0 || col >= (int64_t)t->cols.size() || !device_ptr || !n_bytes) return QHIP_INVALID_ARGUMENT;
  if (t->cols[(size_t)col].deferred || t->rows_dev) {
    if (!t->ctx) return QHIP_INVALID_ARGUMENT;
    const int rc = guarded(static_cast<qhip_ctx*>(t->ctx), [&] { QHIP_HIP_CHECK(hipSetDevice(t->ctx->device)); settle_rows(t); (void)resolved(t->ctx, t->cols[(size_t)col]); });
    if (rc != QHIP_OK) return rc;
  }
  const DevColumn& c = t->cols[(size_t)col];
  const std::shared_ptr<DevBuf>& b = which == 0 ? c.values : which == 1 ? c.validity : c.data;
  *device_ptr = b ? b->ptr : nullptr;
  if (!b) *n_bytes = 0;
  else if (which == 2) *n_bytes = c.data_bytes;
  else if (which == 1) *n_bytes = (c.length + 7) / 8;
  else if (c.type.id == QHIP_UTF8) *n_bytes = (c.length + 1) * 4;
  else if (c.type.id == QHIP_BOOL) *n_bytes = (c.length + 7) / 8;
  else *n_bytes = c.length * dtype_width(c.type);
  return QHIP_OK;
}

int qhip_table_from_device(qhip_ctx* ctx, const char* const* names, const qhip_device_column* cols, int32_t n_cols, int64_t n_rows,
                           qhip_table** out) {
  if (!ctx || !cols || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = table_from_device(ctx, names, cols, n_cols, n_rows); });
}

}  // extern "C"
