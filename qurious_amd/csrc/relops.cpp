// relops.cpp — host-side building blocks shared by Filter, HashJoinExec and the exchange: predicate -> keep mask
// (JIT), mask -> selection vector (scan + ballot rank), row gather of whole columns, key-word evaluation (JIT).
#include "relops.hpp"

#include <algorithm>

#include <hip/hip_runtime_api.h>

#include "device/qhip_status.h"
#include "jit.hpp"
#include "kernels.hpp"

namespace qhip {

void run_pred_mask(Ctx* ctx, const qhip_table* t, const ExprSet& es, const std::vector<InputCol>& icols, int root, DevBuf& mask,
                   DevBuf& wave_count, uint32_t* status_dev) {
  const int64_t N = t->num_rows;
  const uint64_t nwords = (uint64_t)(N + 63) / 64;
  mask.alloc(nwords * 8);
  wave_count.alloc((nwords + 1) * 4);
  if (N == 0) return;
  MaskPlan mp;
  plan_predicate_mask(es, icols, root, mp);
  std::shared_ptr<Module> mod = get_module(ctx, mp.source, mp.kernel_name);
  HKArgs ka;
  DevBuf strlit;
  fill_kargs(ctx, t, mp.bind, ka, strlit);
  // status_dev: a pre-zeroed status block of the caller, who reads it back with whatever else it waits for (no memset,
  // no wait here); nullptr: the context's block, checked before returning
  if (!status_dev) QHIP_HIP_CHECK(hipMemsetAsync(ctx->status.ptr, 0, QS_WORDS * 4, ctx->stream));
  void* mptr = mask.ptr;
  void* wptr = wave_count.ptr;
  void* sptr = status_dev ? (void*)status_dev : ctx->status.ptr;
  void* args[] = {&ka, &mptr, &wptr, &sptr};
  // (a wavefront owns a run of tiles of mp.mask_r mask words; 8 workgroups of 4 wavefronts per CU are resident)
  const uint64_t ntiles = (nwords + (uint64_t)mp.mask_r - 1) / (uint64_t)mp.mask_r;
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((ntiles + 3) / 4, (uint64_t)ctx->num_cus * 8));
  QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr));
  if (status_dev) return;   // (the pooled literal buffer may be recycled: any later writer runs on the same stream, i.e. after this kernel)
  uint32_t status[QS_WORDS];
  QHIP_HIP_CHECK(hipMemcpyAsync(status, ctx->status.ptr, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
  QHIP_HIP_CHECK(sync_stream(ctx->stream));
  check_status_words(status);
}

uint32_t count_from_mask(Ctx* ctx, DevBuf& wave_count, int64_t nrows) {
  const uint64_t nwords = (uint64_t)(nrows + 63) / 64;
  if (nrows == 0) return 0;
  DevBuf total(4);
  exclusive_scan_u32(wave_count.as<uint32_t>(), wave_count.as<uint32_t>(), nwords, total.as<uint32_t>(), ctx->stream);
  uint32_t m = 0;
  copy_sync(ctx->stream, &m, total.ptr, 4, hipMemcpyDeviceToHost);
  return m;
}

void indices_from_mask(Ctx* ctx, const DevBuf& mask, const DevBuf& wave_offset, int64_t nrows, uint32_t m, DevBuf& sel) {
  sel.alloc((size_t)m * 4);
  if (nrows) launch_select_indices(mask.as<uint64_t>(), wave_offset.as<uint32_t>(), (uint64_t)nrows, sel.as<uint32_t>(), ctx->stream);
}

uint32_t select_from_mask(Ctx* ctx, const DevBuf& mask, DevBuf& wave_count, int64_t nrows, DevBuf& sel) {
  if (nrows == 0) { sel.alloc(0); return 0; }
  const uint32_t m = count_from_mask(ctx, wave_count, nrows);
  indices_from_mask(ctx, mask, wave_count, nrows, m, sel);
  return m;   // consumers run on the same stream: no synchronisation needed here
}

bool compact_column(Ctx* ctx, const DevColumn& col_in, const DevBuf& mask, const DevBuf& wave_offset, int64_t nrows, uint32_t m, DevColumn& out) {
  const DevColumn& col = resolved(ctx, col_in);
  if (col.null_count > 0 || col.type.id == QHIP_BOOL) return false;
  out = DevColumn();
  out.type = col.type;
  out.length = (int64_t)m;
  out.utf8_max_len = col.utf8_max_len;
  out.value_maxabs = col.value_maxabs;
  out.range = col.range; out.range_inherited = true;   // (a subset's values lie inside its source's range)
  if (col.type.id == QHIP_NULL) { out.null_count = (int64_t)m; return true; }
  const int w = dtype_width(col.type);
  if (w > 0) {
    out.values = std::make_shared<DevBuf>((size_t)m * w);
    launch_compact_fixed(col.values->ptr, mask.as<uint64_t>(), wave_offset.as<uint32_t>(), out.values->ptr, (uint64_t)nrows, w, ctx->stream);
    return true;
  }
  if (col.type.id == QHIP_UTF8) {
    // every value exactly one byte long (TPC-H flags): the data bytes are a 1-byte column, the offsets 0, 1, 2, ...
    if (col.utf8_max_len < 0) {
      DevBuf mx(4);
      QHIP_HIP_CHECK(hipMemsetAsync(mx.ptr, 0, 4, ctx->stream));
      launch_utf8_max_len(col.values->as<int32_t>(), (uint64_t)col.length, mx.as<uint32_t>(), ctx->stream);
      uint32_t v = 0;
      copy_sync(ctx->stream, &v, mx.ptr, 4, hipMemcpyDeviceToHost);
      col.utf8_max_len = (int32_t)v;
      out.utf8_max_len = col.utf8_max_len;
    }
    if (!(col.utf8_max_len == 1 && col.data_bytes == col.length)) return false;
    out.values = std::make_shared<DevBuf>(((size_t)m + 1) * 4);
    launch_iota_u32(out.values->as<uint32_t>(), (uint64_t)m + 1, ctx->stream, 0);
    out.data = std::make_shared<DevBuf>((size_t)m + 64);
    out.data_bytes = (int64_t)m;
    launch_compact_fixed(col.data->ptr, mask.as<uint64_t>(), wave_offset.as<uint32_t>(), out.data->ptr, (uint64_t)nrows, 1, ctx->stream);
    return true;
  }
  return false;
}

void ensure_utf8_key_lengths(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n,
                             std::vector<InputCol>& icols) {
  for (int k = 0; k < n; ++k) {
    if (roots[k] < 0 || roots[k] >= n_exprs) continue;
    const qhip_expr& e = exprs[roots[k]];
    if (e.kind != QHIP_EXPR_COLUMN || e.column < 0 || e.column >= (int)t->cols.size()) continue;
    if (t->cols[(size_t)e.column].type.id != QHIP_UTF8) continue;   // (before resolving: a fixed-width key may be read through its index vector)
    const DevColumn& col = resolved(ctx, t->cols[(size_t)e.column]);
    if (col.utf8_max_len < 0) {
      DevBuf m(4);
      QHIP_HIP_CHECK(hipMemsetAsync(m.ptr, 0, 4, ctx->stream));
      launch_utf8_max_len(col.values->as<int32_t>(), (uint64_t)col.length, m.as<uint32_t>(), ctx->stream);
      uint32_t v = 0;
      copy_sync(ctx->stream, &v, m.ptr, 4, hipMemcpyDeviceToHost);
      col.utf8_max_len = (int32_t)v;
    }
    icols[(size_t)e.column].utf8_max_len = col.utf8_max_len;
    // longest value 1 byte and as many data bytes as rows: every value (NULL slots included) is exactly 1 byte, so the
    // (rebased) offsets are 0, 1, 2, ... and a kernel can address the data bytes by row number — Q1's flag columns
    icols[(size_t)e.column].utf8_fixed1 = col.utf8_max_len == 1 && col.data_bytes == col.length && getenv("QHIP_NO_FIXED1") == nullptr;
  }
}

void ensure_value_bounds(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, std::vector<InputCol>& icols, int64_t min_rows) {
  const bool narrow_ok = env_int("QHIP_NARROW_DECIMALS", 1) != 0;
  const bool narrow_first = env_int("QHIP_NARROW_FIRST_USE", 0) != 0;
  std::vector<char> seen(t->cols.size(), 0);   // (a column referenced by several expression nodes counts as ONE read)
  // Statistics first, for ALL the columns that lack them: one pass per column, ONE host wait for the lot (round 3: a wait per
  // column), and — Decimal128 columns — the 4-byte narrow copy written speculatively by the same pass (round 3: a second pass
  // over the column at its second big read). Adopted below when the maximum fits 31 bits, dropped otherwise.
  {
    struct Pending { const DevColumn* col; std::shared_ptr<DevBuf> spec; size_t at; bool shared; };
    std::vector<Pending> pend;
    std::vector<char> seen0(t->cols.size(), 0);
    const bool narrow_indirect = narrow_ok && env_int("QHIP_NARROW_INDIRECT", 1) != 0;
    for (int k = 0; k < n_exprs; ++k) {
      const qhip_expr& e = exprs[k];
      if (e.kind != QHIP_EXPR_COLUMN || e.column < 0 || e.column >= (int)t->cols.size() || seen0[(size_t)e.column]) continue;
      if (t->cols[(size_t)e.column].type.id != QHIP_DECIMAL128 && t->cols[(size_t)e.column].type.id != QHIP_INT64) continue;
      seen0[(size_t)e.column] = 1;
      if (icols[(size_t)e.column].indirect) {
        // read through an index vector: the statistics and the narrow copy belong to the SOURCE column and live in the object
        // all its copies share (ColRange); made at the second read of a big Decimal128 source
        const DevColumn& src = t->cols[(size_t)e.column].deferred->src;
        ColRange& sh = *src.range;
        if (narrow_indirect && src.type.id == QHIP_DECIMAL128 && src.values && src.length >= min_rows && !src.range_inherited &&
            (sh.maxabs == 0 || (sh.narrow_src != src.values->ptr && sh.maxabs != ~0ULL && sh.maxabs < (1ULL << 31))) && ++sh.reads >= 2)
          pend.push_back(Pending{&src, nullptr, pend.size() * 2, true});
        continue;
      }
      const DevColumn& col = resolved(ctx, t->cols[(size_t)e.column]);
      if (col.value_maxabs == 0 && col.length >= min_rows && col.values) pend.push_back(Pending{&col, nullptr, pend.size() * 2, false});
    }
    if (!pend.empty()) {
      DevBuf out(pend.size() * 16);
      QHIP_HIP_CHECK(hipMemsetAsync(out.ptr, 0, out.bytes, ctx->stream));
      for (Pending& p : pend) {
        const DevColumn& col = *p.col;
        const bool spec = narrow_ok && col.type.id == QHIP_DECIMAL128 && !col.deferred && (p.shared || env_int("QHIP_NARROW_SPECULATIVE", 1) != 0);
        if (spec) p.spec = std::make_shared<DevBuf>((size_t)col.length * 4);
        launch_value_maxabs(col.values->ptr, (uint64_t)col.length, col.type.id == QHIP_DECIMAL128 ? 2 : 1, out.as<uint64_t>() + p.at, ctx->stream,
                            spec ? p.spec->as<uint32_t>() : nullptr);
      }
      std::vector<uint64_t> h(pend.size() * 2, 0);
      copy_sync(ctx->stream, h.data(), out.ptr, h.size() * 8, hipMemcpyDeviceToHost);
      for (Pending& p : pend) {
        const DevColumn& col = *p.col;
        if (p.shared) {
          ColRange& sh = *col.range;
          sh.maxabs = h[p.at + 1] ? ~0ULL : std::max<uint64_t>(h[p.at], 1);
          if (p.spec && sh.maxabs < (1ULL << 31)) { sh.narrow_buf = p.spec; sh.narrow_bytes = 4; sh.narrow_src = col.values->ptr; sh.narrow_rows = col.length; }
          continue;
        }
        col.value_maxabs = h[p.at + 1] ? ~0ULL : std::max<uint64_t>(h[p.at], 1);
        if (p.spec && col.value_maxabs < (1ULL << 31)) {
          auto nc = std::make_shared<DevColumn::NarrowCopy>();
          nc->buf = p.spec; nc->bytes = 4; nc->src = col.values->ptr; nc->rows = col.length;
          col.narrow = nc;
        }
      }
    }
  }
  for (int k = 0; k < n_exprs; ++k) {
    const qhip_expr& e = exprs[k];
    if (e.kind != QHIP_EXPR_COLUMN || e.column < 0 || e.column >= (int)t->cols.size()) continue;
    if (t->cols[(size_t)e.column].type.id != QHIP_DECIMAL128 && t->cols[(size_t)e.column].type.id != QHIP_INT64) continue;
    if (seen[(size_t)e.column]) continue;
    seen[(size_t)e.column] = 1;
    // (a deferred gather the kernel reads through its index vector is not gathered for the statistic: it carries its source's)
    const DevColumn& col = icols[(size_t)e.column].indirect ? t->cols[(size_t)e.column] : resolved(ctx, t->cols[(size_t)e.column]);
    // the bound enters lowered-plan cache keys: rounded up to a whole number of bits so that plans are shared by data
    // of the same magnitude
    uint64_t m = col.value_maxabs;
    if (icols[(size_t)e.column].indirect) {
      const DevColumn& src = col.deferred->src;
      const ColRange& sh = *src.range;
      if (m == 0 && sh.maxabs != 0 && !src.range_inherited) m = sh.maxabs;
      // ... and its source's narrow copy: the gathers then touch a quarter of the address range
      if (narrow_ok && m != 0 && m != ~0ULL && src.type.id == QHIP_DECIMAL128 && src.values && env_int("QHIP_NARROW_INDIRECT", 1) != 0) {
        if (sh.narrow_buf && sh.narrow_src == src.values->ptr && sh.narrow_rows == src.length) icols[(size_t)e.column].narrow_bytes = sh.narrow_bytes;
        // (... or the copy the table's own column had when the gather was deferred: made by a direct reader)
        else if (src.narrow && src.narrow->buf && src.narrow->src == src.values->ptr && src.narrow->rows == src.length) icols[(size_t)e.column].narrow_bytes = src.narrow->bytes;
      }
    }
    if (m != 0 && m != ~0ULL) { int bits = 1; while (bits < 63 && (m >> bits)) ++bits; m = (1ULL << bits) - 1; }
    icols[(size_t)e.column].value_maxabs = m;
    // every value fits 32 / 64 bits: the kernel reads the column's narrow copy (DevColumn::narrow), made here once per column
    if (narrow_ok && m != 0 && m != ~0ULL && col.type.id == QHIP_DECIMAL128 && !icols[(size_t)e.column].indirect && col.values && col.length > 0 &&
        (++col.big_reads >= 2 || col.narrow || narrow_first)) {
      const int nb = m < (1ULL << 31) ? 4 : 8;
      if (!col.narrow || col.narrow->bytes != nb || col.narrow->src != col.values->ptr || col.narrow->rows != col.length) {
        auto nc = std::make_shared<DevColumn::NarrowCopy>();
        nc->buf = std::make_shared<DevBuf>((size_t)col.length * (size_t)nb);
        nc->bytes = nb; nc->src = col.values->ptr; nc->rows = col.length;
        launch_narrow_decimal(col.values->ptr, (uint64_t)col.length, nb, nc->buf->ptr, ctx->stream);
        col.narrow = nc;
      }
      icols[(size_t)e.column].narrow_bytes = nb;
    }
  }
}

// An aggregate over a join output reads each of its arguments through the join's index vectors: source[index[row]] — one random
// 64-byte access per row and COLUMN, and the memory system serves ~35-40 G of those per second whatever their size (DESIGN §9
// item 0): Q3's aggregate reads four columns that way (price and discount of lineitem, date and priority of orders) and spends
// its time there. Columns of ONE source table read through ONE index vector are therefore interleaved into records (4- / 8-byte
// fields, a stride of 8 or 16 bytes: a record never straddles a 64-byte line), built once per table at the second such read
// like the narrow copies and kept in the objects the table's columns share (ColRange): one access per row and TABLE.
void ensure_indirect_records(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, std::vector<InputCol>& icols, int64_t min_rows) {
  // (off by default: the execution that builds the records re-lowers the plan — its key holds the stride — so a repeated query
  // reaches its steady state one execution later, and the measured gain is a tenth of the aggregate kernel)
  const int mode = env_int("QHIP_INDIRECT_RECORDS", 0);
  if (mode == 0) return;
  // mode 2 (not validated on the device yet — DESIGN §9's short list): EVERY sighting of an indirect base column counts, also
  // while it is still 16 bytes wide, so that a group whose members get their narrow copies in the second execution gets its
  // record in that same execution and the plan is lowered once more, not twice
  const bool count_sightings = mode >= 2;
  struct Cand { int column; const DevColumn* src; const void* vals; int width; };
  struct Group { const void* idx; int64_t rows; std::vector<Cand> cols; };
  std::vector<Group> groups;
  std::vector<char> seen(t->cols.size(), 0);
  for (int k = 0; k < n_exprs; ++k) {
    const qhip_expr& e = exprs[k];
    if (e.kind != QHIP_EXPR_COLUMN || e.column < 0 || e.column >= (int)t->cols.size() || seen[(size_t)e.column]) continue;
    seen[(size_t)e.column] = 1;
    if (!icols[(size_t)e.column].indirect) continue;
    const DeferredGather& d = *t->cols[(size_t)e.column].deferred;
    const DevColumn& src = d.src;
    if (src.range_inherited || src.length < min_rows || !src.values) continue;   // (base columns only: an intermediate result lives for one query)
    ColRange& sh = *src.range;
    if (count_sightings) ++sh.rec_reads;
    int w = dtype_width(src.type);
    const void* vals = src.values->ptr;
    if (const int nb = icols[(size_t)e.column].narrow_bytes) {   // the field holds what the kernel would read from the narrow copy
      w = nb;
      if (sh.narrow_buf && sh.narrow_bytes == nb && sh.narrow_src == src.values->ptr && sh.narrow_rows == src.length) vals = sh.narrow_buf->ptr;
      else if (src.narrow && src.narrow->buf && src.narrow->bytes == nb && src.narrow->src == src.values->ptr && src.narrow->rows == src.length) vals = src.narrow->buf->ptr;
      else continue;
    }
    if (w != 4 && w != 8) continue;
    if (sh.rec_buf && sh.rec_src == src.values->ptr && sh.rec_rows == src.length) {
      if (sh.rec_width == w) icols[(size_t)e.column].rec_stride = sh.rec_stride;   // (a record made earlier, maybe with other partners)
      continue;
    }
    Group* g = nullptr;
    for (Group& x : groups) if (x.idx == (const void*)d.idx.get() && x.rows == src.length) g = &x;
    if (!g) { groups.push_back(Group{(const void*)d.idx.get(), src.length, {}}); g = &groups.back(); }
    g->cols.push_back(Cand{e.column, &src, vals, w});
  }
  for (Group& g : groups) {
    if (g.cols.size() < 2) continue;
    int reads = 0;
    if (count_sightings) { for (const Cand& c : g.cols) reads = std::max(reads, c.src->range->rec_reads); }
    else reads = ++g.cols[0].src->range->rec_reads;
    if (reads < 2 && env_int("QHIP_NARROW_FIRST_USE", 0) == 0) continue;   // (the second read earns the copy)
    std::stable_sort(g.cols.begin(), g.cols.end(), [](const Cand& a, const Cand& b) { return a.width > b.width; });
    std::vector<std::pair<const Cand*, int>> fields;
    int end = 0;
    for (const Cand& c : g.cols) if (end + c.width <= 16) { fields.emplace_back(&c, end); end += c.width; }
    if (fields.size() < 2) continue;
    const int stride = end <= 8 ? 8 : 16;
    std::shared_ptr<DevBuf> buf;
    try { buf = std::make_shared<DevBuf>((size_t)g.rows * (size_t)stride); } catch (const Error&) { continue; }   // (no room: the columns' own arrays serve)
    for (auto& f : fields) {
      const DevColumn& src = *f.first->src;
      launch_pack_field(f.first->vals, (uint64_t)g.rows, f.first->width, buf->ptr, (uint32_t)stride, (uint32_t)f.second, ctx->stream);
      ColRange& sh = *src.range;
      sh.rec_buf = buf; sh.rec_stride = stride; sh.rec_offset = f.second; sh.rec_width = f.first->width; sh.rec_src = src.values->ptr; sh.rec_rows = src.length;
      icols[(size_t)f.first->column].rec_stride = stride;
    }
  }
}

// Int64 columns read by a big probe side (join keys: TPC-H's order and customer keys) whose values fit 32 bits get a 4-byte
// narrow copy too (DevColumn::narrow), decided from the column's value RANGE (DevColumn::range: computed once per table column,
// key_range_of) — Q3's lineitem probe then streams 8 instead of 12 bytes per row.
void ensure_narrow_int_columns(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, std::vector<InputCol>& icols, int64_t min_rows) {
  if (env_int("QHIP_NARROW_INTS", 1) == 0 || t->num_rows < min_rows) return;
  std::vector<char> seen(t->cols.size(), 0);
  for (int k = 0; k < n_exprs; ++k) {
    const qhip_expr& e = exprs[k];
    if (e.kind != QHIP_EXPR_COLUMN || e.column < 0 || e.column >= (int)t->cols.size()) continue;
    if (t->cols[(size_t)e.column].type.id != QHIP_INT64 || icols[(size_t)e.column].indirect) continue;
    if (seen[(size_t)e.column]) continue;
    seen[(size_t)e.column] = 1;
    const DevColumn& col = resolved(ctx, t->cols[(size_t)e.column]);
    if (!col.values || col.length < min_rows) continue;
    int64_t mn = 0, mx = 0;
    // (only base columns are worth a copy: a column with an inherited range — a gathered / filtered subset — is an intermediate
    // result; tested BEFORE the range is asked for, which on such a column would cost a reduction and a host wait per query)
    if (col.range_inherited || !key_range_of(ctx, col, mn, mx)) continue;
    if (mn < -(int64_t)0x7fffffff || mx > (int64_t)0x7fffffff) continue;
    if (++col.big_reads < 2 && !col.narrow && env_int("QHIP_NARROW_FIRST_USE", 0) == 0) continue;   // (DevColumn::big_reads: the second read earns the copy)
    if (!col.narrow || col.narrow->bytes != 4 || col.narrow->src != col.values->ptr || col.narrow->rows != col.length) {
      auto nc = std::make_shared<DevColumn::NarrowCopy>();
      nc->buf = std::make_shared<DevBuf>((size_t)col.length * 4);
      nc->bytes = 4; nc->src = col.values->ptr; nc->rows = col.length;
      launch_narrow_decimal(col.values->ptr, (uint64_t)col.length, 4, nc->buf->ptr, ctx->stream, 1);
      col.narrow = nc;
    }
    icols[(size_t)e.column].narrow_bytes = 4;
  }
}

// Value range of an integer-like column (DevColumn::range): known, inherited from the column it was gathered from, or
// computed now — one reduction + one read-back — over the base column's values: for a deferred gather that is its SOURCE
// (the base table's column, whose ColRange object every query's gathers share), so a table's key range is found once.
bool key_range_of(Ctx* ctx, const DevColumn& col_in, int64_t& mn, int64_t& mx) {
  const DevColumn* col = &col_in;
  if (!col->range->known) {
    if (col->pending_upload) col = &resolved(ctx, *col);
    const DevColumn* base = col;
    if (col->deferred && !col->deferred->done) {
      DevColumn& src = col->deferred->src;
      if (src.pending_upload) (void)resolved(ctx, src);
      base = &src;
    } else if (col->deferred) {
      base = &resolved(ctx, *col);
      col = base;
    }
    if (!base->range->known) {
      const int w = dtype_width(base->type);
      if (!base->values || base->length <= 0 || w <= 0 || w > 8 || dtype_is_float(base->type)) return false;
      if (base->range_inherited) {   // a gathered subset: its own (narrower) bounds must not reach its source's object
        base->range = std::make_shared<ColRange>();
        base->range_inherited = false;
      }
      DevBuf out(16);
      QHIP_HIP_CHECK(hipMemsetAsync(out.ptr, 0, 16, ctx->stream));
      const bool is_signed = !(base->type.id == QHIP_UINT8 || base->type.id == QHIP_UINT16 || base->type.id == QHIP_UINT32 || base->type.id == QHIP_UINT64);
      launch_value_range(base->values->ptr, (uint64_t)base->length, w, is_signed, out.as<uint64_t>(), ctx->stream);
      uint64_t h[2] = {0, 0};
      copy_sync(ctx->stream, h, out.ptr, 16, hipMemcpyDeviceToHost);
      base->range->max = (int64_t)(h[0] ^ 0x8000000000000000ULL);
      base->range->min = (int64_t)(~h[1] ^ 0x8000000000000000ULL);
      base->range->known = true;
    }
    if (base != col) { col->range = base->range; col->range_inherited = true; }
  }
  mn = col->range->min;
  mx = col->range->max;
  return true;
}

DevColumn materialize_upload(Ctx* ctx, const DeferredUpload& u);   // table.cpp

const DevColumn& resolved(Ctx* ctx, const DevColumn& col) {
  if (col.pending_upload) {
    DeferredUpload& u = *col.pending_upload;
    if (!u.done) {
      u.result = materialize_upload(ctx, u);
      u.done = true;
      u.host.reset();
    }
    DevColumn r = u.result;
    r.utf8_max_len = col.utf8_max_len >= 0 ? col.utf8_max_len : r.utf8_max_len;
    const_cast<DevColumn&>(col) = std::move(r);
    return col;
  }
  if (!col.deferred) return col;
  DeferredGather& d = *col.deferred;
  if (!d.done) {
    d.result = gather_column(ctx, d.src, d.idx->as<uint32_t>(), d.m, d.idx_may_be_null);
    d.done = true;
    d.src = DevColumn();   // the source buffers and the index vector are no longer needed by this column
    d.idx.reset();
  }
  if (col.utf8_max_len >= 0 && d.result.utf8_max_len < 0) d.result.utf8_max_len = col.utf8_max_len;
  if (col.value_maxabs != 0 && d.result.value_maxabs == 0) d.result.value_maxabs = col.value_maxabs;
  // the table keeps the gathered column from now on (tables are immutable to their users; this only fills in a value
  // that was owed). Other tables sharing the DeferredGather find the cached result.
  DevColumn r = d.result;
  const_cast<DevColumn&>(col) = std::move(r);
  return col;
}

void resolve_all(Ctx* ctx, const qhip_table* t) {
  for (const DevColumn& c : t->cols) (void)resolved(ctx, c);
}

void defer_gather(Ctx* ctx, const std::vector<DevColumn>& cols, const std::shared_ptr<DevBuf>& idx, uint64_t m, bool idx_may_be_null,
                  std::vector<DevColumn>& out) {
  std::vector<std::pair<const DevBuf*, std::shared_ptr<DevBuf>>> composed;   // inner index vector -> inner[idx]
  // the compositions (one per distinct inner index vector) go into ONE launch
  {
    GatherBatch gb;
    memset(&gb, 0, sizeof gb);
    int n = 0;
    for (const DevColumn& c : cols) {
      if (!(c.deferred && !c.deferred->done)) continue;
      const DeferredGather& in = *c.deferred;
      bool seen = false;
      for (auto& e : composed) seen = seen || e.first == in.idx.get();
      if (seen || n >= kGatherBatch) continue;
      auto both = std::make_shared<DevBuf>((m + 1) * 4);
      gb.d[n++] = GatherDesc{in.idx->ptr, idx->as<uint32_t>(), both->ptr, m, 4u, 1u};
      composed.emplace_back(in.idx.get(), both);
    }
    if (n == 1) launch_gather_u32_nullable((const uint32_t*)gb.d[0].in, gb.d[0].idx, (uint32_t*)gb.d[0].out, m, ctx->stream);
    else if (n > 1) launch_gather_multi(gb, n, ctx->stream);
  }
  for (const DevColumn& c : cols) {
    DevColumn o;
    o.type = c.type;
    o.length = (int64_t)m;
    o.utf8_max_len = c.utf8_max_len;
    o.value_maxabs = c.value_maxabs;
    o.range = c.range; o.range_inherited = true;   // (a subset's values lie inside its source's range)
    auto d = std::make_shared<DeferredGather>();
    d->m = m;
    if (c.deferred && !c.deferred->done) {
      const DeferredGather& in = *c.deferred;
      std::shared_ptr<DevBuf> both;
      for (auto& e : composed) if (e.first == in.idx.get()) both = e.second;
      if (!both) {
        both = std::make_shared<DevBuf>((m + 1) * 4);
        launch_gather_u32_nullable(in.idx->as<uint32_t>(), idx->as<uint32_t>(), both->as<uint32_t>(), m, ctx->stream);
        composed.emplace_back(in.idx.get(), both);
      }
      d->src = in.src;
      d->idx = both;
      d->idx_may_be_null = in.idx_may_be_null || idx_may_be_null;
    } else {
      d->src = c.deferred ? c.deferred->result : c;
      d->src.deferred.reset();
      d->idx = idx;
      d->idx_may_be_null = idx_may_be_null;
    }
    o.null_count = (d->src.null_count > 0 || d->idx_may_be_null) ? 1 : 0;
    if (d->src.type.id == QHIP_NULL) o.null_count = (int64_t)m;
    o.deferred = d;
    out.push_back(std::move(o));
  }
}

bool indirect_eligible(const DevColumn& c) {
  if (!c.deferred || c.deferred->done || c.pending_upload) return false;
  const DeferredGather& d = *c.deferred;
  return !d.idx_may_be_null && !d.src.deferred && !d.src.pending_upload && d.src.null_count == 0 && dtype_width(d.src.type) > 0 && d.src.values &&
         d.idx && d.m > 0 && env_int("QHIP_NO_LATE_GATHER", 0) == 0;
}

void resolve_referenced(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, bool keep_indirect) {
  if (keep_indirect) {   // the caller's kernel reads the plain deferred gathers through their index vectors: only the others are gathered
    for (int k = 0; k < n_exprs; ++k)
      if (exprs[k].kind == QHIP_EXPR_COLUMN && exprs[k].column >= 0 && exprs[k].column < (int)t->cols.size() &&
          !indirect_eligible(t->cols[(size_t)exprs[k].column]))
        (void)resolved(ctx, t->cols[(size_t)exprs[k].column]);
    return;
  }
  // the plain cases first, together: deferred gathers of fixed-width columns without NULLs (no validity to gather, nothing
  // to count) go into ONE launch per 8 columns
  {
    std::vector<DeferredGather*> batch;
    std::vector<bool> seen(t->cols.size(), false);
    for (int k = 0; k < n_exprs; ++k) {
      if (exprs[k].kind != QHIP_EXPR_COLUMN || exprs[k].column < 0 || exprs[k].column >= (int)t->cols.size() || seen[(size_t)exprs[k].column]) continue;
      seen[(size_t)exprs[k].column] = true;
      const DevColumn& c = t->cols[(size_t)exprs[k].column];
      if (!c.deferred || c.deferred->done || c.pending_upload) continue;
      DeferredGather& d = *c.deferred;
      if (d.idx_may_be_null || d.src.deferred || d.src.pending_upload || d.src.null_count > 0 || dtype_width(d.src.type) <= 0 || !d.src.values || d.m == 0) continue;
      batch.push_back(&d);
    }
    for (size_t first = 0; batch.size() >= 2 && first < batch.size(); first += (size_t)kGatherBatch) {
      GatherBatch gb;
      memset(&gb, 0, sizeof gb);
      const int n = (int)std::min<size_t>((size_t)kGatherBatch, batch.size() - first);
      for (int j = 0; j < n; ++j) {
        DeferredGather& d = *batch[first + (size_t)j];
        DevColumn out;
        out.type = d.src.type;
        out.length = (int64_t)d.m;
        out.utf8_max_len = d.src.utf8_max_len;
        out.value_maxabs = d.src.value_maxabs;
        out.range = d.src.range; out.range_inherited = true;   // (a subset's values lie inside its source's range)
        const int w = dtype_width(d.src.type);
        out.values = std::make_shared<DevBuf>((size_t)d.m * (size_t)w);
        gb.d[j] = GatherDesc{d.src.values->ptr, d.idx->as<uint32_t>(), out.values->ptr, d.m, (uint32_t)w, 0u};
        d.result = std::move(out);
      }
      launch_gather_multi(gb, n, ctx->stream);
      for (int j = 0; j < n; ++j) {   // (the launch holds raw pointers: the sources may go only now — stream-ordered pool)
        DeferredGather& d = *batch[first + (size_t)j];
        d.done = true;
        d.src = DevColumn();
        d.idx.reset();
      }
    }
  }
  for (int k = 0; k < n_exprs; ++k)
    if (exprs[k].kind == QHIP_EXPR_COLUMN && exprs[k].column >= 0 && exprs[k].column < (int)t->cols.size())
      (void)resolved(ctx, t->cols[(size_t)exprs[k].column]);
}

DevColumn gather_column(Ctx* ctx, const DevColumn& col_in, const uint32_t* idx, uint64_t m, bool idx_may_be_null) {
  const DevColumn& col = resolved(ctx, col_in);
  DevColumn out;
  out.type = col.type;
  out.length = (int64_t)m;
  out.utf8_max_len = col.utf8_max_len;   // an upper bound stays an upper bound under gathering
  out.value_maxabs = col.value_maxabs;
  out.range = col.range; out.range_inherited = true;   // (a subset's values lie inside its source's range)
  if (col.type.id == QHIP_NULL) { out.null_count = (int64_t)m; return out; }
  DevBuf counter(4);
  if (col.null_count > 0 || idx_may_be_null) {
    auto words = std::make_shared<DevBuf>(((m + 63) / 64) * 8 + 8);
    QHIP_HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 4, ctx->stream));
    launch_gather_bits(col.validity ? col.validity->as<uint8_t>() : nullptr, idx, m, words->as<uint64_t>(), counter.as<uint32_t>(), ctx->stream);
    uint32_t set = 0;
    QHIP_HIP_CHECK(hipMemcpyAsync(&set, counter.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
    QHIP_HIP_CHECK(sync_stream(ctx->stream));
    out.null_count = (int64_t)m - (int64_t)set;
    if (out.null_count > 0) out.validity = words;
  }
  const int w = dtype_width(col.type);
  if (w > 0) {
    out.values = std::make_shared<DevBuf>((size_t)m * w);
    launch_gather_fixed(col.values->ptr, idx, out.values->ptr, m, w, ctx->stream);
  } else if (col.type.id == QHIP_BOOL) {
    out.values = std::make_shared<DevBuf>(((m + 63) / 64) * 8 + 8);
    QHIP_HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 4, ctx->stream));
    launch_gather_bits(col.values->as<uint8_t>(), idx, m, out.values->as<uint64_t>(), counter.as<uint32_t>(), ctx->stream);
  } else if (col.type.id == QHIP_UTF8 && !idx_may_be_null && col.utf8_max_len == 1 && col.data_bytes == col.length) {
    // every value exactly one byte long (TPC-H flags; known once the column was a key or went through a Filter): the data
    // bytes gather like a 1-byte column, the offsets are 0, 1, 2, ... — no length scan, no read-back of the byte total
    out.values = std::make_shared<DevBuf>((size_t)(m + 1) * 4);
    launch_iota_u32(out.values->as<uint32_t>(), m + 1, ctx->stream, 0);
    out.data = std::make_shared<DevBuf>((size_t)m + 64);
    out.data_bytes = (int64_t)m;
    launch_gather_fixed(col.data->ptr, idx, out.data->ptr, m, 1, ctx->stream);
  } else if (col.type.id == QHIP_UTF8) {
    out.values = std::make_shared<DevBuf>((size_t)(m + 1) * 4);
    uint32_t* off = out.values->as<uint32_t>();
    launch_gather_utf8_lengths(col.values->as<int32_t>(), idx, m, off, ctx->stream);
    DevBuf total(4);
    exclusive_scan_u32(off, off, m, total.as<uint32_t>(), ctx->stream);
    uint32_t nbytes = 0;
    copy_sync(ctx->stream, &nbytes, total.ptr, 4, hipMemcpyDeviceToHost);
    if (nbytes > 0x7fffffffu) fail(QHIP_UNSUPPORTED, "gathered Utf8 column exceeds 2 GiB");
    QHIP_HIP_CHECK(hipMemcpyAsync(off + m, total.ptr, 4, hipMemcpyDeviceToDevice, ctx->stream));
    out.data = std::make_shared<DevBuf>((size_t)nbytes);
    out.data_bytes = nbytes;
    launch_gather_utf8_bytes(col.values->as<int32_t>(), col.data->as<uint8_t>(), idx, m, off, out.data->as<uint8_t>(), ctx->stream);
  }
  return out;   // stream-ordered: whoever reads the column next runs on ctx->stream or synchronises it first
}

void eval_key_words(Ctx* ctx, const qhip_table* t, const ExprSet& es, const std::vector<InputCol>& icols, const int32_t* roots, int n,
                    KeysPlan& kp, DevBuf& keys, DevBuf& keyvalid, int predicate_root, bool deferred_status, uint32_t* status_dev) {
  plan_keys(es, icols, roots, n, kp, predicate_root);
  if (!status_dev) status_dev = ctx->status.as<uint32_t>();
  const int64_t N = t->num_rows;
  const uint64_t nwords = (uint64_t)(N + 63) / 64;
  keys.alloc((size_t)kp.W * (size_t)N * 8);
  keyvalid.alloc(nwords * 8 + 8);
  if (N == 0) return;
  std::shared_ptr<Module> mod = get_module(ctx, kp.source, kp.kernel_name);
  HKArgs ka;
  DevBuf strlit;
  fill_kargs(ctx, t, kp.bind, ka, strlit);
  if (!deferred_status) QHIP_HIP_CHECK(hipMemsetAsync(status_dev, 0, QS_WORDS * 4, ctx->stream));
  void* kptr = keys.ptr;
  void* vptr = keyvalid.ptr;
  void* sptr = status_dev;
  void* args[] = {&ka, &kptr, &vptr, &sptr};
  const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nwords + 3) / 4, (uint64_t)ctx->num_cus * 8));
  QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr));
  // deferred: the caller reads ctx->status at its next natural synchronisation point. The pooled literal buffer may be
  // recycled after return: any later writer runs on the same stream, i.e. after this kernel.
  if (deferred_status) return;
  uint32_t status[QS_WORDS];
  QHIP_HIP_CHECK(hipMemcpyAsync(status, status_dev, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
  QHIP_HIP_CHECK(sync_stream(ctx->stream));
  check_status_words(status);
}

}  // namespace qhip

// The boundaries are read on the table's stream into the page-locked scratch (one round trip), once.
void qhip::settle_rows(const qhip_table* tc) {
  if (!tc || !tc->rows_dev) return;
  qhip_table* t = const_cast<qhip_table*>(tc);
  Ctx* ctx = t->ctx;
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  QHIP_HIP_CHECK(qhip::sync_stream(ctx->stream));
  try {
    verify_pending_sizes(ctx);   // QHIP_RETRY when the join had too little room: the table is then void
  } catch (...) {
    ctx->pending_sizes.clear();
    throw;
  }
  const int64_t m = t->deferred_count();
  t->num_rows = m;
  for (DevColumn& c : t->cols) {
    c.length = m;
    if (c.type.id == QHIP_NULL) c.null_count = m;
    if (c.deferred) {
      c.deferred->m = (uint64_t)m;
      if (c.deferred->done) {   // gathered over the capacity by an earlier reader: the filler rows' NULLs do not count
        DevColumn& r = c.deferred->result;
        r.length = m;
        if (r.validity && m > 0) {
          DevBuf counter(4);
          QHIP_HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 4, ctx->stream));
          launch_count_bits(r.validity->as<uint64_t>(), (uint64_t)m, counter.as<uint32_t>(), ctx->stream);
          uint32_t set = 0;
          copy_sync(ctx->stream, &set, counter.ptr, 4, hipMemcpyDeviceToHost);
          r.null_count = m - (int64_t)set;
        } else if (m == 0) r.null_count = 0;
      }
    }
  }
  if (t->pending_offsets) { t->pending_offsets->total_rows = m; t->pending_offsets->search_m = (uint64_t)m; }
  t->rows_dev = nullptr;
  t->rows_host = nullptr;
  t->rows_final.reset();
  t->rows_blk.reset();
}

const std::vector<int64_t>& qhip_table::offsets() const {
  if (!pending_offsets) return batch_offsets;
  const qhip::PendingOffsets& p = *pending_offsets;
  if (!ctx) qhip::fail(QHIP_INVALID_ARGUMENT, "table with pending batch boundaries has no context");
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  std::vector<uint32_t> pos_v;
  uint32_t* pos = (uint32_t*)((uint8_t*)ctx->pinned + 128);   // (the first 128 bytes hold status words)
  if (128 + p.n * 4 + 1024 > ctx->pinned_bytes) {
    pos_v.resize(p.n);
    pos = pos_v.data();
  }
  if (p.search_in) {
    qhip::settle_rows(this);   // (a join of deferred size: the number of pairs must be exact before it is searched)
    pending_offsets->pos = std::make_shared<qhip::DevBuf>(p.n * 4);
    if (!p.bounds) {   // the boundary rows were left on the host: uploaded now that somebody asks
      pending_offsets->bounds = std::make_shared<qhip::DevBuf>(p.bounds_host.size() * 8);
      qhip::copy_sync(ctx->stream, p.bounds->ptr, p.bounds_host.data(), p.bounds_host.size() * 8, hipMemcpyHostToDevice);
    }
    qhip::launch_lower_bound_u32(p.search_in->as<uint32_t>(), p.search_m, nullptr, p.bounds->as<uint64_t>(), (uint32_t)p.n, p.pos->as<uint32_t>(), ctx->stream);
  }
  QHIP_HIP_CHECK(hipMemcpyAsync(pos, p.pos->ptr, p.n * 4, hipMemcpyDeviceToHost, ctx->stream));
  QHIP_HIP_CHECK(qhip::sync_stream(ctx->stream));
  batch_offsets.clear();
  if (p.skip_empty) {
    batch_offsets.push_back(0);
    for (size_t b = 1; b < p.n; ++b)
      if ((int64_t)pos[b] > batch_offsets.back()) batch_offsets.push_back((int64_t)pos[b]);
    if (p.tail) batch_offsets.push_back(p.total_rows);
  } else {
    batch_offsets.assign(pos, pos + p.n);
  }
  pending_offsets.reset();
  return batch_offsets;
}

const uint64_t* qhip_table::device_offsets() const {
  if (!offsets_dev) {
    const std::vector<int64_t>& off = offsets();
    std::vector<uint64_t> rows(off.begin(), off.end());
    auto d = std::make_shared<qhip::DevBuf>(rows.size() * 8);
    qhip::copy_sync(ctx->stream, d->ptr, rows.data(), rows.size() * 8, hipMemcpyHostToDevice);
    offsets_dev = d;
  }
  return offsets_dev->as<uint64_t>();
}
