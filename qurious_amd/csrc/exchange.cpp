// exchange.cpp — the pieces of the multi-GPU hash-join exchange that run on one GPU:
//   qhip_partition_by_key   split a table into n_parts tables by mix64(join key) so equal keys meet on one rank
//   qhip_table_column_buffer / qhip_table_from_device   raw device buffers for the RCCL all-to-all (moved by torch.distributed)
//   qhip_table_concat       append the tables received from the peers
// The reference has no exchange operator (it is single-process); only the joined RESULT must match it.
#include <hip/hip_runtime_api.h>

#include <algorithm>

#include "common.hpp"
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
  eval_key_words(ctx, in, es, icols, roots, n_keys, kp, keys, valid);
  DevBuf part((N + 1) * 4), hist((size_t)(n_parts + 1) * 4), iota((N + 1) * 4), sorted_part((N + 1) * 4), sorted_rows((N + 1) * 4);
  QHIP_HIP_CHECK(hipMemsetAsync(hist.ptr, 0, hist.bytes, s));
  launch_partition_ids(kp.W, keys.as<uint64_t>(), N, (uint32_t)n_parts, part.as<uint32_t>(), hist.as<uint32_t>(), s);
  launch_iota_u32(iota.as<uint32_t>(), N, s);
  stable_sort_pairs_u32(part.as<uint32_t>(), sorted_part.as<uint32_t>(), iota.as<uint32_t>(), sorted_rows.as<uint32_t>(), N,
                        std::max(1, log2u((uint32_t)n_parts)), s);
  std::vector<uint32_t> h((size_t)n_parts);
  copy_sync(s, h.data(), hist.ptr, (size_t)n_parts * 4, hipMemcpyDeviceToHost);
  uint64_t pos = 0;
  for (int p = 0; p < n_parts; ++p) {
    std::unique_ptr<qhip_table> t(new qhip_table());
    t->ctx = ctx;
    t->names = in->names;
    t->nullable = in->nullable;
    const uint64_t m = h[(size_t)p];
    for (auto& c : in->cols) t->cols.push_back(gather_column(ctx, c, sorted_rows.as<uint32_t>() + pos, m, false));
    t->num_rows = (int64_t)m;
    t->batch_offsets = {0, (int64_t)m};
    pos += m;
    out_parts[p] = t.release();
  }
}

qhip_table* table_concat(Ctx* ctx, const qhip_table* const* ts, int n) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  if (n <= 0) fail(QHIP_INVALID_ARGUMENT, "qhip_table_concat: no tables");
  hipStream_t s = ctx->stream;
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
    for (int64_t b = 0; b < ts[k]->num_batches(); ++b) out->batch_offsets.push_back(N + ts[k]->batch_offsets[(size_t)b + 1]);
    N += ts[k]->num_rows;
  }
  out->num_rows = N;
  if (N >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  // concat by gathering through (table, row) -> the simple way that handles every layout incl. bitmaps: build for every
  // source table an index vector and gather it into the destination range. Fixed-width columns use plain D2D copies.
  for (size_t c = 0; c < first->cols.size(); ++c) {
    DevColumn oc;
    oc.type = first->cols[c].type;
    oc.length = N;
    const int w = dtype_width(oc.type);
    int64_t nulls = 0, bytes = 0;
    for (int k = 0; k < n; ++k) {
      if (ts[k]->cols[c].type != oc.type) fail(QHIP_INVALID_ARGUMENT, "qhip_table_concat: column types differ");
      nulls += ts[k]->cols[c].null_count;
      bytes += ts[k]->cols[c].data_bytes;
    }
    oc.null_count = nulls;
    if (oc.type.id == QHIP_NULL) { out->cols.push_back(oc); continue; }
    if (w > 0) {
      oc.values = std::make_shared<DevBuf>((size_t)N * w);
      int64_t pos = 0;
      for (int k = 0; k < n; ++k) {
        const DevColumn& sc = ts[k]->cols[c];
        if (sc.length) QHIP_HIP_CHECK(hipMemcpyAsync((uint8_t*)oc.values->ptr + (size_t)pos * w, sc.values->ptr, (size_t)sc.length * w, hipMemcpyDeviceToDevice, s));
        pos += sc.length;
      }
    }
    // bitmaps (validity, Boolean values) and Utf8 go through the host: exchange outputs are concatenated once per query
    auto concat_bits = [&](bool values) {
      std::vector<uint8_t> host((size_t)((N + 7) / 8 + 8), 0);
      int64_t pos = 0;
      for (int k = 0; k < n; ++k) {
        const DevColumn& sc = ts[k]->cols[c];
        const std::shared_ptr<DevBuf>& src = values ? sc.values : sc.validity;
        std::vector<uint8_t> tmp((size_t)((sc.length + 7) / 8 + 8), values ? 0 : 0xff);
        if (src && sc.length) copy_sync(s, tmp.data(), src->ptr, (size_t)((sc.length + 7) / 8), hipMemcpyDeviceToHost);
        for (int64_t i = 0; i < sc.length; ++i)
          if ((tmp[(size_t)(i >> 3)] >> (i & 7)) & 1) host[(size_t)((pos + i) >> 3)] |= (uint8_t)(1u << ((pos + i) & 7));
        pos += sc.length;
      }
      auto b = std::make_shared<DevBuf>(host.size());
      copy_sync(s, b->ptr, host.data(), host.size(), hipMemcpyHostToDevice);
      return b;
    };
    if (nulls > 0) oc.validity = concat_bits(false);
    if (oc.type.id == QHIP_BOOL) oc.values = concat_bits(true);
    if (oc.type.id == QHIP_UTF8) {
      if (bytes > 0x7fffffffLL) fail(QHIP_UNSUPPORTED, "Utf8 column larger than 2 GiB");
      std::vector<int32_t> off((size_t)N + 1, 0);
      oc.data = std::make_shared<DevBuf>((size_t)bytes);
      oc.data_bytes = bytes;
      int64_t pos = 0, bpos = 0;
      for (int k = 0; k < n; ++k) {
        const DevColumn& sc = ts[k]->cols[c];
        if (!sc.length) continue;
        std::vector<int32_t> so((size_t)sc.length + 1);
        copy_sync(s, so.data(), sc.values->ptr, so.size() * 4, hipMemcpyDeviceToHost);
        for (int64_t i = 0; i < sc.length; ++i) off[(size_t)(pos + i)] = so[(size_t)i] - so[0] + (int32_t)bpos;
        const int64_t nb = so[(size_t)sc.length] - so[0];
        if (nb) QHIP_HIP_CHECK(hipMemcpyAsync((uint8_t*)oc.data->ptr + bpos, (const uint8_t*)sc.data->ptr + so[0], (size_t)nb, hipMemcpyDeviceToDevice, s));
        pos += sc.length;
        bpos += nb;
      }
      off[(size_t)N] = (int32_t)bpos;
      oc.values = std::make_shared<DevBuf>(off.size() * 4);
      copy_sync(s, oc.values->ptr, off.data(), off.size() * 4, hipMemcpyHostToDevice);
    }
    out->cols.push_back(std::move(oc));
  }
  QHIP_HIP_CHECK(hipStreamSynchronize(s));
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
  QHIP_HIP_CHECK(hipStreamSynchronize(s));
  return t.release();
}

}  // namespace

extern "C" {

int qhip_partition_by_key(qhip_ctx* ctx, const qhip_table* input, const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots,
                          int32_t n_keys, int32_t n_parts, qhip_table** out_parts) {
  if (!ctx || !input || !out_parts) return QHIP_INVALID_ARGUMENT;
  for (int p = 0; p < n_parts; ++p) out_parts[p] = nullptr;
  int rc = guarded(ctx, [&] { partition_by_key(ctx, input, exprs, n_exprs, key_roots, n_keys, n_parts, out_parts); });
  if (rc != QHIP_OK)
    for (int p = 0; p < n_parts; ++p) { if (out_parts[p]) { delete out_parts[p]; out_parts[p] = nullptr; } }
  return rc;
}

int qhip_table_concat(qhip_ctx* ctx, const qhip_table* const* tables, int32_t n, qhip_table** out) {
  if (!ctx || !tables || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { *out = table_concat(ctx, tables, n); });
}

int qhip_table_column_buffer(const qhip_table* t, int64_t col, int32_t which, void** device_ptr, int64_t* n_bytes) {
  if (!t || col < 0 || col >= (int64_t)t->cols.size() || !device_ptr || !n_bytes) return QHIP_INVALID_ARGUMENT;
  if (t->cols[(size_t)col].deferred) {
    if (!t->ctx) return QHIP_INVALID_ARGUMENT;
    const int rc = guarded(static_cast<qhip_ctx*>(t->ctx), [&] { QHIP_HIP_CHECK(hipSetDevice(t->ctx->device)); (void)resolved(t->ctx, t->cols[(size_t)col]); });
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
