// agg.cpp — qhip_hash_aggregate_execute: HashAggregate / NoGroupingAggregate with the scan filter fused in.
//
// Reference: physical/plan/aggregate/hash.rs:138-170 (execute), :45-87 (GroupAccumulator::update),
// :89-107 (output); aggregate/no_grouping.rs:30-62; accumulators physical/expr/aggregate/*.rs;
// fused predicate = MemoryTable::scan (datasource/memory.rs:90-93) / Filter (physical/plan/filter.rs:28-44).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>

#include "codegen.hpp"
#include "common.hpp"
#include "device/qhip_status.h"
#include "hostcol.hpp"
#include "jit.hpp"
#include "kargs_host.hpp"
#include "kernels.hpp"
#include "relops.hpp"

namespace qhip {
static uint64_t g_learn_tick = 0;   // orders what twin aggregate plans learnt (AggPlan::learnt_at)


// ---------------------------------------------------------------- HostColumn
void HostColumn::init_fixed(const DType& t, int64_t n) {
  type = t; length = n; null_count = 0;
  const int w = dtype_width(t);
  if (w) values.assign((size_t)n * w, 0);
  else if (t.id == QHIP_BOOL) values.assign((size_t)((n + 7) / 8), 0);
  else if (t.id == QHIP_UTF8) offsets.assign((size_t)n + 1, 0);
}
void HostColumn::set_null(int64_t i) {
  if (validity.empty()) validity.assign((size_t)((length + 7) / 8), 0xff);
  validity[(size_t)(i >> 3)] &= (uint8_t)~(1u << (i & 7));
  ++null_count;
}

// A host-assembled column -> HBM (runs when a device operator first reads the column; no host wait: the host vectors are kept
// alive in a small ring until an event recorded behind the copies has passed)
DevColumn upload_host_column(Ctx* ctx, const DeferredUpload& u) {
  HostColumn& hc = *std::static_pointer_cast<HostColumn>(u.host_col);
  const int64_t nrows = hc.length;
  DevColumn dc;
  dc.type = hc.type; dc.length = nrows; dc.null_count = hc.null_count;
  auto up = [&](const void* src, size_t n) {
    auto b = std::make_shared<DevBuf>(n);
    if (n) QHIP_HIP_CHECK(hipMemcpyAsync(b->ptr, src, n, hipMemcpyHostToDevice, ctx->stream));
    return b;
  };
  if (hc.null_count > 0 && hc.type.id != QHIP_NULL) {
    hc.validity.resize((size_t)((nrows + 7) / 8 + 8), 0);
    dc.validity = up(hc.validity.data(), hc.validity.size());
  }
  if (hc.type.id == QHIP_UTF8) {
    dc.values = up(hc.offsets.data(), hc.offsets.size() * 4);
    dc.data = up(hc.data.data(), hc.data.size());
    dc.data_bytes = (int64_t)hc.data.size();
  } else if (hc.type.id == QHIP_BOOL) {
    hc.values.resize((size_t)((nrows + 7) / 8 + 8), 0);
    dc.values = up(hc.values.data(), hc.values.size());
  } else if (hc.type.id != QHIP_NULL) {
    dc.values = up(hc.values.data(), hc.values.size());
  }
  // the host vectors may go away with the DeferredUpload while the copies are still queued: park them behind an event
  Ctx::HostKeep& k = ctx->host_keep[ctx->host_keep_next++ % 32];
  if (!k.ev) QHIP_HIP_CHECK(hipEventCreateWithFlags(&k.ev, hipEventDisableTiming));
  else if (k.p && hipEventQuery(k.ev) != hipSuccess) { (void)hipGetLastError(); QHIP_HIP_CHECK(sync_event(k.ev)); }   // (32 uploads ago: long done)
  k.p = u.host_col;
  QHIP_HIP_CHECK(hipEventRecord(k.ev, ctx->stream));
  return dc;
}

// A single-batch (or zero-batch) table from host columns. The columns STAY on the host (an aggregate's few result rows
// are usually exported next, as the reference's results are host batches) and are uploaded when a device operator reads
// them (Sort / Limit / Projection over an aggregate, a join over a subquery result).
qhip_table* table_from_host(Ctx* ctx, const std::vector<std::string>& names, const std::vector<bool>& nullable,
                            std::vector<HostColumn>& cols, int64_t nrows, bool zero_batches) {
  std::unique_ptr<qhip_table> t(new qhip_table());
  t->ctx = ctx;
  t->names = names;
  t->nullable = nullable;
  t->num_rows = nrows;
  t->batch_offsets.push_back(0);
  if (!zero_batches) t->batch_offsets.push_back(nrows);
  for (auto& hc : cols) {
    DevColumn dc;
    dc.type = hc.type; dc.length = nrows; dc.null_count = hc.type.id == QHIP_NULL ? nrows : hc.null_count;
    hc.length = nrows;
    auto u = std::make_shared<DeferredUpload>();
    u->host_col = std::make_shared<HostColumn>(std::move(hc));
    dc.pending_upload = u;
    t->cols.push_back(std::move(dc));
  }
  return t.release();
}

// ---------------------------------------------------------------- helpers shared with other operators
std::vector<InputCol> input_cols_of(const qhip_table* t, bool mark_indirect) {
  std::vector<InputCol> v;
  for (auto& c0 : t->cols) {
    // a deferred column that nobody resolved is not referenced by the plan being typed: its may-have-nulls flag is enough
    const DevColumn& c = (c0.deferred && c0.deferred->done) ? c0.deferred->result : c0;
    InputCol ic; ic.type = c.type; ic.has_nulls = c.null_count > 0; ic.utf8_max_len = c.utf8_max_len;
    ic.indirect = mark_indirect && indirect_eligible(c0);
    if (ic.indirect) ic.has_nulls = false;   // (eligible = a source without NULLs and an index vector without NULL indices)
    v.push_back(ic);
  }
  return v;
}

void fill_kargs(Ctx* ctx, const qhip_table* t, const KernelBindings& b, HKArgs& a, DevBuf& strlit_dev) {
  memset(&a, 0, sizeof(a));
  for (size_t s = 0; s < b.cols.size(); ++s) {
    const DevColumn& c0 = t->cols[(size_t)b.cols[s]];
    if (s < b.indirect.size() && b.indirect[s]) {   // read through the deferred gather's index vector (InputCol::indirect)
      if (!indirect_eligible(c0)) fail(QHIP_HIP_ERROR, "a column planned as an indirect read has been gathered meanwhile (internal error)");
      a.c[s].v = c0.deferred->src.values->ptr;
      a.c[s].d = (const uint8_t*)c0.deferred->idx->ptr;
      if (s < b.rec.size() && b.rec[s]) {   // ... as a field of the source's record copy (ColRange::rec_buf)
        const DevColumn& src = c0.deferred->src;
        const ColRange& sh = *src.range;
        const int w = s < b.narrow.size() && b.narrow[s] ? (int)b.narrow[s] : dtype_width(src.type);
        if (!sh.rec_buf || sh.rec_stride != (int)b.rec[s] || sh.rec_width != w || sh.rec_src != src.values->ptr || sh.rec_rows != src.length)
          fail(QHIP_HIP_ERROR, "an indirect column planned with a record copy has none of that layout (internal error)");
        a.c[s].v = (const uint8_t*)sh.rec_buf->ptr + sh.rec_offset;
        continue;
      }
      if (s < b.narrow.size() && b.narrow[s]) {   // ... from the source's narrow copy (the object every copy of the column shares)
        const DevColumn& src = c0.deferred->src;
        const ColRange& sh = *src.range;
        if (sh.narrow_buf && sh.narrow_bytes == (int)b.narrow[s] && sh.narrow_src == src.values->ptr && sh.narrow_rows == src.length) a.c[s].v = sh.narrow_buf->ptr;
        else if (src.narrow && src.narrow->buf && src.narrow->bytes == (int)b.narrow[s] && src.narrow->src == src.values->ptr && src.narrow->rows == src.length)
          a.c[s].v = src.narrow->buf->ptr;
        else fail(QHIP_HIP_ERROR, "an indirect column planned with a narrow copy has none of that width (internal error)");
      }
      continue;
    }
    const DevColumn& c = resolved(ctx, c0);
    a.c[s].v = c.values ? c.values->ptr : nullptr;
    if (s < b.narrow.size() && b.narrow[s]) {   // the kernel was generated for the column's narrow copy (InputCol::narrow_bytes)
      if (!c.narrow || c.narrow->bytes != (int)b.narrow[s] || !c.values || c.narrow->src != c.values->ptr || c.narrow->rows != c.length)
        fail(QHIP_HIP_ERROR, "a column planned with a narrow copy has none of that width (internal error)");
      a.c[s].v = c.narrow->buf->ptr;
    }
    a.c[s].n = c.validity ? (const uint8_t*)c.validity->ptr : nullptr;
    a.c[s].d = c.data ? (const uint8_t*)c.data->ptr : nullptr;
  }
  for (size_t l = 0; l < b.lit_lo.size(); ++l) { a.lit_lo[l] = b.lit_lo[l]; a.lit_hi[l] = b.lit_hi[l]; }
  for (size_t l = 0; l < b.stroff.size() && l < (size_t)kMaxLits + 1; ++l) a.stroff[l] = b.stroff[l];
  // (a caller that keeps strlit_dev with its cached plan uploads the literals once: same bindings, same bytes)
  if (!strlit_dev.ptr || strlit_dev.bytes != b.strlits.size()) {
    strlit_dev.alloc(b.strlits.size());
    if (!b.strlits.empty())
      QHIP_HIP_CHECK(hipMemcpyAsync(strlit_dev.ptr, b.strlits.data(), b.strlits.size(), hipMemcpyHostToDevice, ctx->stream));
  }
  a.strlit = (const uint8_t*)strlit_dev.ptr;
  a.nrows = t->num_rows;
  a.nrows_dev = t->rows_dev;   // (a join output whose size the host has not waited for: aggregate / join build only)
}

void check_status_words(const uint32_t* st) {
  if (st[QS_KEY_TOO_LONG]) fail(QHIP_UNSUPPORTED, "Utf8 group/join key longer than its packed key words (expression keys: 7 bytes)");
  if (st[QS_DIV_ZERO]) fail(QHIP_EXEC_ERROR, "Arrow error: Divide by zero error");
  if (st[QS_CAST_OVERFLOW]) fail(QHIP_EXEC_ERROR, "Arrow error: Cast error: value out of range for the target type");
  if (st[QS_ARITH_OVERFLOW]) fail(QHIP_EXEC_ERROR, "Arrow error: Arithmetic overflow: Overflow happened on integer division");
}

static uint32_t pow2_ceil(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return (uint32_t)std::min<uint64_t>(p, 1ULL << 31);
}

static double ord_to_f64(uint64_t k) {
  uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
  double d; memcpy(&d, &b, 8); return d;
}

// ---------------------------------------------------------------- the operator
// The input of a mid-sized many-group aggregate, ordered by key hash into one part per workgroup (qk_filter_agg_parts reads part
// p = rows [runs[p * stride], runs[(p + 1) * stride]) of the view and appends its groups to the dense slots: AggLaunch)
struct AggParts { const uint32_t* runs; uint32_t stride; int n_parts; uint64_t hint_key; };

static uint64_t agg_hint_key(const qhip_expr* exprs, int n_exprs, int pred_root, const int32_t* group_roots, int n_groups, const qhip_agg* aggs, int n_aggs) {
  uint64_t h = 1469598103934665603ULL;
  auto mix = [&](const void* p, size_t n) { for (size_t k = 0; k < n; ++k) { h ^= ((const unsigned char*)p)[k]; h *= 1099511628211ULL; } };
  for (int k = 0; k < n_exprs; ++k) {
    qhip_expr e = exprs[k];
    const char* str = e.lit_str; const int64_t len = e.lit_len;
    e.lit_str = nullptr;
    mix(&e, sizeof e);
    if (str && len > 0 && e.kind == QHIP_EXPR_LITERAL) mix(str, (size_t)len);
  }
  mix(&pred_root, sizeof pred_root);
  mix(group_roots, sizeof(int32_t) * (size_t)n_groups);
  mix(aggs, sizeof(qhip_agg) * (size_t)n_aggs);
  return h;
}

static qhip_table* hash_aggregate(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, int pred_root,
                                  const int32_t* group_roots, int n_groups, const qhip_agg* aggs, int n_aggs,
                                  const char* const* out_names, const AggParts* parts = nullptr) {
  const bool trace = getenv("QHIP_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (trace) fprintf(stderr, "[qhip agg] %-28s %8.1f us\n", what,
                       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
  };
  trace_point("aggregate: entry");
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_groups < 0 || n_aggs < 0 || (n_groups > 0 && !group_roots) || (n_aggs > 0 && !aggs) || (n_exprs > 0 && !exprs))
    fail(QHIP_INVALID_ARGUMENT, "qhip_hash_aggregate_execute: bad arguments");
  for (int k = 0; k < n_groups; ++k)
    if (group_roots[k] < 0 || group_roots[k] >= n_exprs) fail(QHIP_INVALID_ARGUMENT, "group expression index out of range");
  if (pred_root >= n_exprs) fail(QHIP_INVALID_ARGUMENT, "predicate index out of range");
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;

  // ---- mid-sized input, many groups (the same aggregate produced them last time): order the rows by key hash into one part per
  // workgroup first — the exchange's two partition passes over the ROW NUMBERS (nothing but the parts' selection vector is
  // written; the columns are then read through it, composed with whatever index vectors a join below left) — and aggregate
  // every part in its workgroup's LDS table alone: no group is shared between workgroups, so nothing is merged into the HBM
  // table and no LDS table overflows (configs[4]'s per-rank aggregate, 2 M joined rows -> 200 k groups: round 3 spent 46 % of
  // the kernel merging ~1.5 M (workgroup, group) pairs with memory-side atomics). QHIP_AGG_PARTS: 0 never, 1 auto, 2 always.
  const uint64_t hint_key = parts ? parts->hint_key : agg_hint_key(exprs, n_exprs, pred_root, group_roots, n_groups, aggs, n_aggs);
  if (!parts && n_groups > 0 && pred_root < 0 && !in->no_batches()) {
    const int mode = env_int("QHIP_AGG_PARTS", 1);
    const auto hint = ctx->agg_group_hints.find(hint_key);
    const uint32_t groups_hint = hint != ctx->agg_group_hints.end() ? hint->second : 0u;
    const int64_t N0 = in->num_rows;
    // Only over plain columns. A join output read through index vectors pays ~64 bytes of random sector traffic per row and
    // referenced column; ordering the rows first makes the key columns pay it twice (pass 1 and the aggregate) — measured on
    // configs[4]'s per-rank aggregate (2 M joined rows, 5 columns through 2 index vectors, Zipf keys): pass 1 141 us + pass 2 /
    // index composition 67 us + aggregate 347 us against 307 us for the unpartitioned kernel, whose own floor those gathers
    // are (profiles/r04_q3_sf100_slice_parts_timeline.txt). What that input needs is the aggregate's arguments evaluated where
    // the pairs are emitted (a dense record stream), not another pass over the index vectors.
    bool plain_input = !in->rows_dev;
    for (int k = 0; k < n_exprs; ++k)
      if (exprs[k].kind == QHIP_EXPR_COLUMN && exprs[k].column >= 0 && exprs[k].column < (int)in->cols.size() && in->cols[(size_t)exprs[k].column].deferred)
        plain_input = false;
    const bool three_pass_forced = env_int("QHIP_AGG_PARTITION", 1) == 2;   // (tests: the three-pass partitioned path keeps precedence)
    // (from 2^20 rows: below, the five launches in front of the kernel cost what the merges did — Q3 at SF10, 0.34 M rows -> 113 k
    // groups: 118 us against 59)
    if (!three_pass_forced && (mode == 2 ? N0 > 0 : (mode == 1 && plain_input && groups_hint >= 16384 && N0 >= ((int64_t)1 << 20) && N0 <= ((int64_t)1 << 22) &&
                                                      env_int("QHIP_AGG_STATS", 0) == 0))) {
      // parts: a 1 024-thread workgroup's LDS table (128 KB) at a load of ~0.4
      int slot_words_guess = 8;
      { const auto sw = ctx->agg_slot_words.find(hint_key); if (sw != ctx->agg_slot_words.end()) slot_words_guess = sw->second; }
      uint32_t lslots = 16;
      while ((uint64_t)lslots * 2 * (uint64_t)slot_words_guess * 8 <= 128 * 1024) lslots *= 2;
      // (at least ~3/4 of the CUs' worth of parts: a part is one workgroup's work)
      const int np = (int)std::max<uint64_t>(mode == 2 ? 2 : std::min<uint64_t>(255, (uint64_t)ctx->num_cus * 3 / 4),
                                             std::min<uint64_t>(255, ((uint64_t)std::max<uint32_t>(groups_hint, 1) * 5 / 2 + lslots - 1) / lslots));
      PartitionWork w;
      partition_pass1(ctx, in, exprs, n_exprs, group_roots, n_groups, -1, np, w);
      std::vector<MovedColumn> moved;
      std::vector<size_t> odd;
      std::shared_ptr<DevBuf> sel;
      partition_scatter(ctx, in, nullptr, np, w, (uint64_t)N0, moved, odd, sel, true);
      qhip_table view;
      view.ctx = ctx;
      view.names = in->names;
      view.nullable = in->nullable;
      view.num_rows = N0;
      view.batch_offsets = {0, N0};
      defer_gather(ctx, in->cols, sel, (uint64_t)N0, false, view.cols);
      AggParts ap{w.runs.as<uint32_t>(), w.n_units, np, hint_key};
      qhip_table* out = hash_aggregate(ctx, &view, exprs, n_exprs, pred_root, group_roots, n_groups, aggs, n_aggs, out_names, &ap);
      ctx->stats.rows_in = in->rows_dev ? in->deferred_count() : N0;
      // a join of deferred size that turned out to have produced nothing has no output batches (hash_join.rs:363-372)
      if (in->rows_dev && in->deferred_count() == 0 && out->num_rows == 0) { out->batch_offsets.assign(1, 0); out->pending_offsets.reset(); }
      return out;   // (w / sel go back to the stream-ordered pool: whoever gets them next runs behind the kernels that read them)
    }
  }
  // (late materialisation: the plain deferred gathers of a join output are read through their index vectors by the kernel)
  resolve_referenced(ctx, in, exprs, n_exprs, true);
  std::vector<InputCol> icols = input_cols_of(in, true);
  ensure_utf8_key_lengths(ctx, in, exprs, n_exprs, group_roots, n_groups, icols);
  // |value| bounds of the Int64 / Decimal128 columns (cached per column; computed only on inputs big enough to pay for the
  // reduction): the generated code multiplies and accumulates in 32 / 64 bits where the bounds allow
  if (env_int("QHIP_AGG_NO_BOUNDS", 0) == 0) ensure_value_bounds(ctx, in, exprs, n_exprs, icols, (int64_t)env_int("QHIP_STATS_MIN_ROWS", 1 << 22));   // (the switch: tests / the fuzzer run the statistics paths on small tables)
  // columns read through one index vector: one record per row and source table instead of one array per column
  ensure_indirect_records(ctx, in, exprs, n_exprs, icols, (int64_t)env_int("QHIP_STATS_MIN_ROWS", 1 << 22));
  // lowered plans are cached per context: a repeated query (same expression PODs over the same column signature) skips
  // typing and code generation; literal VALUES are part of the key because they are bound into the plan's KernelBindings
  std::string key = "agg|";
  auto put = [&](const void* p, size_t n) { key.append((const char*)p, n); };
  for (auto& ic : icols) {
    const int v[9] = {ic.type.id, ic.type.precision, ic.type.scale, ic.has_nulls ? 1 : 0, ic.utf8_max_len, ic.utf8_fixed1 ? 1 : 0, ic.indirect ? 1 : 0, ic.narrow_bytes, ic.rec_stride};
    put(&ic.value_maxabs, sizeof ic.value_maxabs);   // (a whole number of bits, see ensure_value_bounds)
    put(v, sizeof v);
  }
  for (int k = 0; k < n_exprs; ++k) {
    qhip_expr e = exprs[k];
    const char* str = e.lit_str; const int64_t len = e.lit_len;
    e.lit_str = nullptr;
    put(&e, sizeof e);
    if (str && len > 0 && e.kind == QHIP_EXPR_LITERAL) put(str, (size_t)len);
  }
  put(&pred_root, sizeof pred_root);
  put(group_roots, sizeof(int32_t) * (size_t)n_groups);
  put(aggs, sizeof(qhip_agg) * (size_t)n_aggs);
  // rows per thread: the register-budget rule of plan_aggregate (R = 2..4) suits inputs that fill the chip many times
  // over; a SMALL input (the 0.3 M joined rows Q3 aggregates) is a latency chain per row — fewer rows per thread and more
  // workgroups shorten it (Q3's aggregate kernel: 73 -> 49 us)
  const int64_t small_rows = (int64_t)256 * ctx->num_cus * 8;
  const int r_env = env_int("QHIP_AGG_R", 0) ? env_int("QHIP_AGG_R", 0) : in->num_rows <= small_rows ? 1 : in->num_rows <= 2 * small_rows ? 2 : 0;
  const int kc_env = env_int("QHIP_AGG_KC", -1);
  put(&r_env, sizeof r_env); put(&kc_env, sizeof kc_env);
  const int dev_rows = in->rows_dev ? 1 : 0;   // (a join output of deferred size: the kernel variant that reads the row count on the device)
  std::string sibling_key = key;
  put(&dev_rows, sizeof dev_rows);
  { const int other = 1 - dev_rows; sibling_key.append((const char*)&other, sizeof other); }
  std::shared_ptr<AggPlan> plan_ptr;
  auto cached = ctx->plan_cache.find(key);
  if (cached != ctx->plan_cache.end()) plan_ptr = std::static_pointer_cast<AggPlan>(cached->second);
  else {
    ExprSet es;
    es.build(exprs, n_exprs, icols);
    plan_ptr = std::make_shared<AggPlan>();
    plan_aggregate(es, icols, pred_root, group_roots, n_groups, aggs, n_aggs, r_env, *plan_ptr, dev_rows != 0);

    if (ctx->plan_cache.size() > 4096) ctx->plan_cache.clear();
    ctx->plan_cache[key] = plan_ptr;
  }
  const AggPlan& plan = *plan_ptr;
  {
    // the same aggregate over an input whose row count is / is not on the device is a twin plan (another kernel variant):
    // what either learnt about the data (groups, occupied slots) serves both
    auto sib = ctx->plan_cache.find(sibling_key);
    if (sib != ctx->plan_cache.end()) {
      const AggPlan& o = *std::static_pointer_cast<AggPlan>(sib->second);
      if (o.learnt_at > plan.learnt_at) { plan.last_groups = o.last_groups; plan.last_dense = o.last_dense; plan.learnt_at = o.learnt_at; }
    }
  }
  mark("planned");

  // output schema: keys then aggregates (hash.rs:166-169)
  std::vector<std::string> names;
  std::vector<bool> nullable;
  for (int k = 0; k < n_groups + n_aggs; ++k) {
    names.push_back(out_names && out_names[k] ? out_names[k] : ("col" + std::to_string(k)));
    nullable.push_back(true);
  }
  const bool zero_batches_in = in->no_batches();
  auto no_batches_out = [&] {   // hash.rs:146-148: no input batches -> no output batches
    std::vector<HostColumn> cols((size_t)(n_groups + n_aggs));
    for (int k = 0; k < n_groups; ++k) cols[(size_t)k].init_fixed(plan.keys[(size_t)k].type, 0);
    for (int k = 0; k < n_aggs; ++k) cols[(size_t)(n_groups + k)].init_fixed(plan.aggs[(size_t)k].ret, 0);
    return table_from_host(ctx, names, nullable, cols, 0, true);
  };
  if (n_groups > 0 && zero_batches_in) return no_batches_out();

  if (!plan.module) plan.module = get_module(ctx, plan.source, plan.kernel_name);
  std::shared_ptr<Module> mod = std::static_pointer_cast<Module>(plan.module);
  HKArgs ka;
  if (!plan.strlit) plan.strlit = std::make_shared<DevBuf>();
  DevBuf& strlit = *std::static_pointer_cast<DevBuf>(plan.strlit);
  fill_kargs(ctx, in, plan.bind, ka, strlit);
  mark("module + kargs");

  const int64_t N = in->num_rows;
  // bytes of column data the kernel reads per row (for roofline figures)
  double bytes_per_row = 0;
  for (int c : plan.bind.cols) {
    const DevColumn& dc = in->cols[(size_t)c];
    const int w = dtype_width(dc.type);
    if (w > 0) bytes_per_row += icols[(size_t)c].narrow_bytes ? icols[(size_t)c].narrow_bytes : w;
    else if (dc.type.id == QHIP_BOOL) bytes_per_row += 0.125;
    else if (dc.type.id == QHIP_UTF8) bytes_per_row += icols[(size_t)c].utf8_fixed1 ? 1.0 : 4.0 + (N > 0 ? (double)dc.data_bytes / (double)N : 0.0);
    if (dc.null_count > 0) bytes_per_row += 0.125;
  }
  const int slot_bytes = plan.slot_words * 8;
  // LDS-staged table: as many slots as fit the per-workgroup LDS budget
  uint32_t l_nslots = 0;
  bool wide = false;
  if (plan.W > 0) {
    // Many groups (the plan's previous run says so) on a mid-sized input: every workgroup's LDS table ends up full and is
    // merged slot by slot into the HBM table at the end, the heavy keys by EVERY workgroup — same-slot atomic traffic that
    // grows with the number of workgroups, and with ~1000 workgroups x 512 slots as many HBM updates as the input has
    // rows. One workgroup per CU with a 64 KB table halves the merges and quarters the contention (Q3 with Zipf(1.1)
    // keys, 1.35 M rows -> 175 k groups: kernel 0.456 -> 0.264 ms). On inputs far bigger than the merge (50 M rows ->
    // 1 M groups, tools/highcard_timing.py) the rows that miss the LDS table dominate and more workgroups hide their
    // latency better (4.7 vs 5.3 ms), so the default shape stays.
    const bool merge_heavy = plan.last_groups > 4096 && N > 2 * (int64_t)256 * ctx->num_cus * 8 && N <= (int64_t)1 << 22;
    // ... and since round 3 that one workgroup per CU has 1024 threads (qk_filter_agg_wide) and a table of up to 128 KB: PMC
    // on configs[4]'s slice showed the four wavefronts per CU of the 256-thread shape waiting 77 % of their time (1.6 M HBM
    // atomics and 0.4 GB of random reads in 0.35 ms: neither a throughput limit) — a latency chain per row with too few rows in
    // flight; more 256-thread workgroups hide it but multiply the end-of-kernel merges (1 / 2 / 4 / 8 per CU: 352 / 399 / 557 /
    // 683 us), sixteen wavefronts on ONE table do not
    wide = merge_heavy && env_int("QHIP_AGG_WIDE", 1) != 0;
    const int lds_budget = env_int("QHIP_AGG_LDS_BYTES", wide ? 128 * 1024 : merge_heavy ? 64 * 1024 : 32 * 1024);
    l_nslots = 16;
    while ((uint64_t)l_nslots * 2 * slot_bytes <= (uint64_t)lds_budget) l_nslots *= 2;
    if ((uint64_t)l_nslots * slot_bytes > (wide ? 128 : 64) * 1024) l_nslots = 0;   // slot too wide for LDS staging
    if (l_nslots == 0) wide = false;
  }
  if (parts) {   // one 1 024-thread workgroup per part on the biggest LDS table that fits
    l_nslots = 16;
    while ((uint64_t)l_nslots * 2 * slot_bytes <= 128 * 1024) l_nslots *= 2;
    wide = true;
  }
  ctx->agg_slot_words[hint_key] = plan.slot_words;
  const size_t lds_bytes = (size_t)l_nslots * slot_bytes;
  const int block = wide ? 1024 : 256;
  // the consecutive-rows form (a lane owns RC adjacent rows: one wide load per column; plan.RC > 0 = every referenced column is
  // plain and narrow): a table of at least a few tiles per workgroup, the 256-thread shape, a row count known on the host
  const bool cons = plan.RC > 0 && !parts && !wide && !in->rows_dev && N >= (int64_t)256 * plan.RC * 64 && env_int("QHIP_AGG_CONS", 1) != 0;
  const int64_t tile_rows = (int64_t)block * (cons ? plan.RC : plan.R);
  const int64_t ntiles = (N + tile_rows - 1) / tile_rows;
  const bool merge_heavy_grid = plan.W > 0 && plan.last_groups > 4096 && N > 2 * (int64_t)256 * ctx->num_cus * 8 && N <= (int64_t)1 << 22;
  const int bpc = env_int("QHIP_AGG_BLOCKS_PER_CU", N <= 2 * (int64_t)256 * ctx->num_cus * 8 ? 8 : merge_heavy_grid ? 1 : 4);
  unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)ctx->num_cus * bpc));
  if (parts) grid = (unsigned)parts->n_parts;

  // First attempt: a SMALL table (4096 slots) replicated 32 times, workgroup b merging into replica b % 32. Clearing and
  // compacting cost time proportional to the table size, and with few groups (Q1: 4) the ~1000 workgroups would
  // otherwise all merge into the same handful of slots at the end of the kernel (measured: ~45 us of a 470 us kernel).
  // The host merges the replicas (<= 32 x G slots). More groups than the small table holds -> the kernel bails out
  // early on the overflow flag -> one un-replicated table sized for the row count, x16 until it fits.
  const uint32_t cap_max = plan.W == 0 ? 1 : std::max<uint32_t>(1024, pow2_ceil((uint64_t)std::max<int64_t>(N, 1) * 2));
  uint32_t cap = plan.W == 0 ? 1 : std::min<uint32_t>(cap_max, (uint32_t)env_int("QHIP_AGG_INITIAL_SLOTS", 4096));
  uint32_t replicas = plan.W == 0 ? 1 : (uint32_t)std::max(1, env_int("QHIP_AGG_REPLICAS", 32));
  if (plan.W > 0 && env_int("QHIP_AGG_PARTITION", 1) == 2) {   // tests: the partitioned path on every grouped aggregate
    replicas = 1;
    cap = std::min<uint32_t>(cap_max, std::max<uint32_t>(cap, 4096));
  }
  if (parts) {   // the HBM table only takes what an LDS table cannot hold (a part with more groups than planned)
    replicas = 1;
    cap = std::min<uint32_t>(cap_max, std::max<uint32_t>(1u << 16, pow2_ceil((uint64_t)plan.last_groups / 2 + 1)));   // (+ the sliced parts' groups; grown x16 on overflow like any table)
  } else
  if (plan.W > 0 && plan.last_groups > cap / 4) {
    // the same plan produced many groups last time: go straight to one table with room for them
    cap = std::min<uint32_t>(cap_max, std::max<uint32_t>(1u << 16, pow2_ceil((uint64_t)plan.last_groups * 2)));
    replicas = 1;
  } else if (plan.W > 0 && plan.last_groups > 0 && plan.last_groups * 16 < cap) {
    // ... or very few (Q1: 4): 16 slots per expected group are plenty, and clearing + compacting the replicated table
    // (both proportional to its size, both on the critical path of the call) shrink with it
    cap = std::min<uint32_t>(cap, std::max<uint32_t>(64, pow2_ceil((uint64_t)plan.last_groups * 16)));
  }
  DevBuf gtable, dense;
  uint32_t status[QS_WORDS];
  int retries = 0;
  float main_ms = 0;
  uint32_t G = 0, guess = 0, pre_copied = 0;   // (pre_copied: dense slots that came back with the status words)
  std::vector<uint64_t> slots;
  // dense slots fetched together with the status words (one sync), through the context's page-locked scratch
  // (256 of them, or what the plan produced last time plus a quarter while that stays a host-side result)
  const size_t pre_want = plan.last_groups > 4096 ? 256 : std::max<size_t>(256, std::min<size_t>(4096, (size_t)plan.last_dense + (size_t)plan.last_dense / 4));
  const uint32_t PRE = (uint32_t)std::min<size_t>(pre_want, (ctx->pinned_bytes - 64 - 8 - 1024) / (size_t)slot_bytes);
  uint32_t* status_pinned = (uint32_t*)ctx->pinned;
  uint64_t* pre_host = (uint64_t*)((uint8_t*)ctx->pinned + 64);
  uint32_t* fin_pinned = (uint32_t*)((uint8_t*)ctx->pinned + ctx->pinned_bytes - 1024);   // [status words (8) | null counts (<= 248)]

  // ---- output assembly on the device for many groups (k_agg_finalize): nothing but a few counters crosses PCIe.
  // enqueue: allocate the output columns for up to `cap_rows` groups and launch; the number of groups is either known
  // (g_dev == nullptr) or read by the kernel from the compaction counter (speculative launch right behind the compaction,
  // so that a repeated many-group query needs ONE synchronisation). finish: after the stream has been synchronised.
  struct DevFinal {
    std::unique_ptr<qhip_table> out;
    std::vector<FinCol> fc;
    DevBuf fc_dev;
    const uint32_t* fin_host = nullptr;   // page-locked [status words | null counts] of this finalisation
    std::vector<std::shared_ptr<DevBuf>> valid_bufs;
    bool has_utf8 = false;
  };
  const int cell0 = 1 + plan.W;
  // fin_dev: zeroed device words [status | null count per column] the caller provides and reads back itself (the speculative
  // launch: they sit in the call's own status block and travel in its ONE read-back, fin_host = where they land); nullptr:
  // a block of their own, read back here
  auto enqueue_device_finalize = [&](DevFinal& F, const uint64_t* dense, uint32_t cap_rows, const uint32_t* g_dev, uint32_t* fin_dev,
                                     const uint32_t* fin_host) {
    const int ncols = n_groups + n_aggs;
    F.fc.assign((size_t)ncols, FinCol());
    F.out.reset(new qhip_table());
    F.out->ctx = ctx;
    F.out->names = names;
    F.out->nullable = nullable;
    const size_t vwords = ((size_t)cap_rows + 63) / 64 + 1;
    for (int k = 0; k < ncols; ++k) {
      FinCol& f = F.fc[(size_t)k];
      memset(&f, 0, sizeof f);
      f.cnt_word = -1; f.key_index = -1; f.src_word = 0;
      DevColumn col;
      if (k < n_groups) {
        const KeyDesc& kd = plan.keys[(size_t)k];
        col.type = kd.type;
        f.src_word = 1 + kd.word_off;
        f.key_index = (plan.null_mask_word && kd.nullable) ? k : -1;
        if (kd.type.id == QHIP_UTF8) { f.kind = F_KEY_UTF8_LEN; f.width = 4; f.pad = kd.words; F.has_utf8 = true; }
        else if (kd.type.id == QHIP_DECIMAL128) { f.kind = F_KEY_DEC; f.width = 16; }
        else { f.kind = F_KEY_FIXED; f.width = dtype_width(kd.type); }
      } else {
        const AggDesc& ad = plan.aggs[(size_t)(k - n_groups)];
        col.type = ad.ret;
        const int cntw = cell0 + plan.cells[(size_t)ad.count_cell].off;
        f.src_word = ad.value_cell >= 0 ? cell0 + plan.cells[(size_t)ad.value_cell].off : 0;
        f.width = dtype_width(ad.ret);
        switch (ad.kind) {
          case QHIP_AGG_COUNT: f.kind = F_COUNT; f.cnt_word = cntw; f.width = 8; break;
          case QHIP_AGG_SUM: f.kind = ad.ret.id == QHIP_DECIMAL128 ? F_SUM128 : F_SUM64; f.cnt_word = cntw; break;
          case QHIP_AGG_AVG:
            f.cnt_word = cntw;
            if (ad.ret.id == QHIP_FLOAT64) f.kind = F_AVG_F64;
            else {
              const DType& at = plan.args[(size_t)ad.arg].type;
              if (ad.ret.scale < at.scale) fail(QHIP_EXEC_ERROR, "Internal error: Arithmetic Overflow in DecimalAvgAccumulator");
              const i128 mul = pow10_i128(ad.ret.scale - at.scale), lim = pow10_i128(ad.ret.precision);
              f.kind = F_AVG_DEC;
              f.mul_lo = (uint64_t)(u128)mul; f.mul_hi = (uint64_t)((u128)mul >> 64);
              f.lim_lo = (uint64_t)(u128)lim; f.lim_hi = (uint64_t)((u128)lim >> 64);
            }
            break;
          default:
            f.is_min = ad.kind == QHIP_AGG_MIN;
            if (ad.ret.id == QHIP_DECIMAL128) f.kind = F_MM_DEC;
            else if (ad.ret.id == QHIP_FLOAT64) f.kind = F_MM_F64;
            else if (ad.ret.id == QHIP_FLOAT32) f.kind = F_MM_F32;
            else { f.kind = F_MM_INT; f.is_signed = dtype_is_signed(ad.ret) || ad.ret.id == QHIP_DATE32 || ad.ret.id == QHIP_DATE64 || (ad.ret.id >= QHIP_TIME32_S && ad.ret.id <= QHIP_TIMESTAMP_NS); }
        }
      }
      col.values = std::make_shared<DevBuf>(f.kind == F_KEY_UTF8_LEN ? ((size_t)cap_rows + 1) * 4 : (size_t)cap_rows * f.width);
      auto vb = std::make_shared<DevBuf>(vwords * 8);
      f.out_values = col.values->ptr;
      f.out_valid = vb->as<uint64_t>();
      F.valid_bufs.push_back(vb);
      F.out->cols.push_back(std::move(col));
    }
    // the column descriptors travel as a kernel argument (up to kFinColsByValue of them: no upload), else through a buffer
    const FinCol* fc_dev = nullptr;
    if (ncols > kFinColsByValue) {
      F.fc_dev.alloc(F.fc.size() * sizeof(FinCol));
      QHIP_HIP_CHECK(hipMemcpyAsync(F.fc_dev.ptr, F.fc.data(), F.fc.size() * sizeof(FinCol), hipMemcpyHostToDevice, ctx->stream));
      fc_dev = (const FinCol*)F.fc_dev.ptr;
    }
    // [status words | null count per column], zeroed, from the context's ring; read back together
    if (ncols > 240) fail(QHIP_UNSUPPORTED, "more than 240 output columns in an aggregate assembled on the device");
    const bool own = fin_dev == nullptr;
    if (own) fin_dev = zeroed_block(ctx, (QS_WORDS + ncols + 31) / 32);
    launch_agg_finalize(dense, cap_rows, g_dev, plan.slot_words, plan.null_mask_word ? 1 : 0, F.fc.data(), fc_dev, ncols,
                        fin_dev + QS_WORDS, fin_dev, ctx->stream);
    F.fin_host = own ? fin_pinned : fin_host;
    if (own) QHIP_HIP_CHECK(hipMemcpyAsync(fin_pinned, fin_dev, (size_t)(QS_WORDS + ncols) * 4, hipMemcpyDeviceToHost, ctx->stream));
  };
  // (call after the stream has been synchronised at least up to the read-backs above)
  auto finish_device_finalize = [&](DevFinal& F, const uint64_t* dense, uint32_t groups, bool synced) -> qhip_table* {
    const int ncols = n_groups + n_aggs;
    if (!synced) QHIP_HIP_CHECK(sync_stream(ctx->stream));   // (the speculative launch sat in front of the call's one wait)
    if (F.fin_host[QS_ARITH_OVERFLOW]) fail(QHIP_EXEC_ERROR, "AVG(Decimal128): scaled sum overflows the result type (reference yields a mistyped NULL, avg.rs:105-116)");
    F.out->num_rows = groups;
    F.out->batch_offsets = {0, (int64_t)groups};
    for (int k = 0; k < ncols; ++k) {
      DevColumn& col = F.out->cols[(size_t)k];
      col.length = groups;
      col.null_count = F.fin_host[QS_WORDS + k];
      if (col.null_count > 0) col.validity = F.valid_bufs[(size_t)k];
      if (F.fc[(size_t)k].kind == F_KEY_UTF8_LEN) {
        // lengths -> offsets (exclusive scan) -> bytes
        uint32_t* off = col.values->as<uint32_t>();
        DevBuf total(4);
        exclusive_scan_u32(off, off, groups, total.as<uint32_t>(), ctx->stream);
        uint32_t nbytes = 0;
        copy_sync(ctx->stream, &nbytes, total.ptr, 4, hipMemcpyDeviceToHost);
        QHIP_HIP_CHECK(hipMemcpyAsync(off + groups, total.ptr, 4, hipMemcpyDeviceToDevice, ctx->stream));
        col.data = std::make_shared<DevBuf>((size_t)nbytes);
        col.data_bytes = nbytes;
        launch_agg_utf8_key_bytes(dense, groups, plan.slot_words, F.fc[(size_t)k].src_word, off, col.data->as<uint8_t>(), ctx->stream);
      }
    }
    if (F.has_utf8) QHIP_HIP_CHECK(sync_stream(ctx->stream));   // the dense slots are released on return
    return F.out.release();
  };

  const uint32_t dev_threshold = (uint32_t)env_int("QHIP_AGG_DEVICE_FINALIZE_MIN_GROUPS", 4096);
  DevFinal spec;                    // speculative device-side assembly enqueued behind the compaction
  bool spec_enqueued = false;
  uint64_t* table_dev = nullptr;
  uint64_t* dense_dev = nullptr;    // [counter | dense slots]
  bool ran_partitioned = false;
  // ---- an input whose equal keys are adjacent (qh_agg_runs_body): no table, no compaction — the kernel writes the dense slots.
  // Tried when the plan produced many SHORT runs' worth of groups last time (>= 4096 groups, on average <= 16 rows each) and no
  // execution has found its input unsorted; the kernel verifies the order and the host falls through to the hashed path when
  // it does not hold. QHIP_AGG_RUNS: 0 never, 1 (default) by that rule, 2 whenever the plan has the entry point (tests).
  bool ran_runs = false;
  {
    const int runs_mode = env_int("QHIP_AGG_RUNS", 1);
    const bool eligible = plan.has_runs && !parts && plan.W > 0 && N > 0 && N < ((int64_t)1 << 31) && (uint64_t)N * slot_bytes <= (256ull << 20) && !plan.not_sorted;
    const bool worth = plan.last_groups >= 4096 && (uint64_t)plan.last_groups * 16 >= (uint64_t)N;
    if (eligible && (runs_mode == 2 || (runs_mode == 1 && worth))) {
      guess = (uint32_t)N;                          // one slot per row at worst: nothing can be lost
      dense.alloc((size_t)guess * slot_bytes + 8);
      dense_dev = dense.as<uint64_t>();
      const int ncols = n_groups + n_aggs;
      uint32_t* status_dev = zeroed_block(ctx, (32 + QS_WORDS + ncols + 31) / 32);
      uint32_t* const counter_dev = status_dev + 16;
      HRunsLaunch rl;
      rl.dense_out = dense_dev + 1; rl.counter = counter_dev; rl.status = status_dev; rl.flags = status_dev + 20;
      rl.cap = guess; rl.max_run = (uint32_t)std::max(16, env_int("QHIP_AGG_RUNS_MAX", 256));
      // dynamic LDS: the evaluated rows of a workgroup's four wavefronts + their look-ahead (4 x 320 Row structs; a Row is at
      // most 8 + 8 W + 24 bytes per argument — the kernel sizes the look-ahead from what it gets and gives up below 264 rows)
      const size_t row_bound = 8 + 8 * (size_t)plan.W + 24 * plan.args.size();
      const int lds_kb = env_int("QHIP_AGG_RUNS_LDS_KB", (int)std::min<size_t>(160, std::max<size_t>(32, (4 * 320 * row_bound + 16383) / 16384 * 16)));
      rl.lds_bytes = (uint32_t)std::max(16, std::min(160, lds_kb)) * 1024u;
      void* rargs[] = {&ka, &rl};
      std::shared_ptr<Module> rmod = get_module(ctx, plan.source, "qk_agg_runs");
      const unsigned rgrid = (unsigned)((N + 1023) / 1024);
      time_mark(ctx, 0);
      QHIP_HIP_CHECK(hipModuleLaunchKernel(rmod->fn, rgrid, 1, 1, 256, 1, 1, rl.lds_bytes, ctx->stream, rargs, nullptr));
      time_mark(ctx, 1);
      bool utf8_key = false;
      for (auto& kd : plan.keys) utf8_key = utf8_key || kd.type.id == QHIP_UTF8;
      const bool will_spec = plan.last_groups >= dev_threshold && !utf8_key && env_int("QHIP_AGG_NO_SPECULATIVE_FINALIZE", 0) == 0;
      if (will_spec) {
        spec = DevFinal();
        enqueue_device_finalize(spec, dense_dev + 1, guess, counter_dev, status_dev + 32, status_pinned + 32);
        spec_enqueued = true;
      }
      // ONE read-back: status + counter + the order flags (word 20) + the finalisation's status and null counts
      QHIP_HIP_CHECK(hipMemcpyAsync(status_pinned, status_dev, (size_t)(32 + QS_WORDS + ncols) * 4, hipMemcpyDeviceToHost, ctx->stream));
      QHIP_HIP_CHECK(sync_stream(ctx->stream));
      memcpy(status, status_pinned, sizeof(status));
      verify_pending_sizes(ctx);
      if (in->rows_dev && in->deferred_count() == 0 && n_groups > 0) return no_batches_out();
      if (ctx->timing) QHIP_HIP_CHECK(hipEventElapsedTime(&main_ms, ctx->ev[0], ctx->ev[1]));
      if (status_pinned[20] == 0) {
        check_status_words(status);
        ran_runs = true;
        replicas = 1;
        cap = guess;
        grid = rgrid;
        l_nslots = 0;
      } else {
        // not that kind of input (or a few long runs): remember, and aggregate it through the table
        if (env_int("QHIP_AGG_RUNS_DEBUG", 0)) fprintf(stderr, "[qhip agg runs] rows %lld: flags %u (1 = order, 2 = run too long), runs counted %u\n", (long long)N, status_pinned[20], status_pinned[16]);
        plan.not_sorted = true;
        spec = DevFinal();
        spec_enqueued = false;
        dense = DevBuf();
        dense_dev = nullptr;
        ++retries;
      }
    }
  }
  if (!ran_runs)
  for (;;) {
    const size_t table_bytes = (size_t)cap * replicas * slot_bytes;
    const uint32_t total_slots = cap * replicas;
    // (a plan that produced many groups last time gets a dense buffer that should hold them all at once)
    guess = plan.W == 0 ? 0 : std::min<uint32_t>(total_slots, std::max<uint32_t>(8192, plan.last_groups + plan.last_groups / 4));
    if (parts) guess = (uint32_t)std::max<int64_t>(N, 1);   // (the workgroups append their groups themselves: room for one group per row, nothing can be lost)
    // Small replicated attempt: a persistent arena [status | counter | table | dense slots] that the PREVIOUS call left
    // zeroed, so the kernel launch is the first thing on the stream. (total_slots <= guess there: one compaction always
    // suffices and the table is not needed again after it.)
    const bool use_arena = !parts && plan.W > 0 && replicas > 1 && total_slots <= 8192 && table_bytes <= (1u << 20) && env_int("QHIP_AGG_NO_ARENA", 0) == 0;
    // [status words (64 bytes) | compaction counter]: in the arena, else a zeroed block of the context's ring; read back together
    uint32_t* status_dev = nullptr;
    std::shared_ptr<DevBuf> arena;
    const size_t zero_bytes = 128 + table_bytes;   // status (64) + counter (64) + table
    if (use_arena) {
      const size_t need = zero_bytes + (size_t)guess * slot_bytes + 64;
      arena = std::static_pointer_cast<DevBuf>(plan.arena);
      if (!arena || plan.arena_bytes != need) {
        arena = std::make_shared<DevBuf>(need);
        plan.arena = arena;
        plan.arena_bytes = need;
        plan.arena_clean = false;
      }
      if (!plan.arena_clean) QHIP_HIP_CHECK(hipMemsetAsync(arena->ptr, 0, zero_bytes, ctx->stream));
      plan.arena_clean = false;
      status_dev = arena->as<uint32_t>();
      table_dev = (uint64_t*)(arena->as<uint8_t>() + 128);
    } else {
      gtable.alloc(table_bytes);
      QHIP_HIP_CHECK(hipMemsetAsync(gtable.ptr, 0, table_bytes, ctx->stream));
      // [status words (16) | counter (2) | .. | words 32..: the speculative finalisation's status (8) + null counts]
      status_dev = zeroed_block(ctx, (32 + QS_WORDS + (n_groups + n_aggs) + 31) / 32);
      table_dev = gtable.as<uint64_t>();
    }
    HAggLaunch L;
    L.gtable = table_dev;
    L.g_nslots = cap;
    L.l_nslots = l_nslots;
    L.status = status_dev;
    L.replicas = replicas;
    L.collect_stats = env_int("QHIP_AGG_STATS", 0) ? 1u : 0u;
    void* args[] = {&ka, &L};
    if (!parts) time_mark(ctx, 0);   // (parts: the clock started in front of the partition passes)
    // Many groups on a big input: partition the rows by key hash first, so that every bin's groups fit an LDS table and the
    // HBM table is touched once per GROUP instead of once per row (device/qhip_device.hpp, "partitioned aggregation").
    // QHIP_AGG_PARTITION: 0 never, 1 when the plan's previous run says it pays (default), 2 always (tests).
    const int pa_mode = env_int("QHIP_AGG_PARTITION", 1);
    // (an instrumented run, QHIP_AGG_STATS, measures the fused kernel's LDS table and keeps to it)
    // Mid-sized inputs (2^18 .. 2^22 rows with >= 16 k groups — BASELINE configs[4]'s per-rank aggregate: 2 M joined rows ->
    // 200 k groups, LDS tables 100 % full, 0.27-0.35 ms in the fused kernel) CAN take the same three passes without a host
    // round trip in between (QHIP_AGG_PARTITION_MID=1: the reduce pass derives its work items — bin slices — from the scanned
    // histogram on the device). Measured on that aggregate (round 3, rocprofv3): histogram 71 us + staged pass 168 us + reduce
    // pass 317 us = 0.57 ms against the fused kernel's 0.35 ms — the input is read through the joins' index vectors (twice
    // here) and Zipf's heavy keys serialise the LDS atomics of their bins, which the fused kernel's wave-resident hot keys
    // avoid. Off by default; what this size needs is a combiner in front of the partitioning, not fewer host waits.
    const bool mid = N < ((int64_t)1 << 22);
    const bool mid_on = env_int("QHIP_AGG_PARTITION_MID", 0) != 0;
    bool partitioned = !parts && plan.W > 0 && N > 0 && replicas == 1 && l_nslots >= 64 && !use_arena && !L.collect_stats &&
                             (pa_mode == 2 || (pa_mode == 1 && N >= ((int64_t)1 << 22) && plan.last_groups >= 32768) ||
                              (pa_mode == 1 && mid && N >= ((int64_t)1 << 18) && plan.last_groups >= 16384 && mid_on));
    const bool device_items = partitioned && mid && (mid_on || pa_mode == 2);
    std::vector<uint32_t> item_first;   // (kept alive until the call's next synchronisation)
    // the record buffer is as big as the input's key + argument columns: when HBM cannot hold it the fused kernel runs
    DevBuf pa_records;
    if (partitioned) {
      try {
        pa_records.alloc((size_t)N * (slot_bytes - 8) + 8);
      } catch (const Error& e) {
        if (e.code != QHIP_OUT_OF_MEMORY) throw;
        partitioned = false;
      }
    }
    if (partitioned) {
      ran_partitioned = true;
      hipStream_t s = ctx->stream;
      // (QHIP_AGG_PART_LDS_BYTES: bigger LDS tables in the reduce pass = fewer bins = longer runs per tile in pass 2 — measured:
      // pass 2 gains less than the reduce pass loses with one or two workgroups per CU: 50 M rows -> 1 M groups 1.86 -> 2.23 ms
      // at 64 KB, 2.98 ms at 128 KB; off by default)
      uint32_t l_nslots_p = l_nslots;
      // the reduce pass as 1 024-thread workgroups with 128 KB LDS tables (one per CU, 16 wavefronts): four times the groups
      // per bin = a quarter of the bins = 4x longer runs per tile in pass 2, and the reduce pass itself keeps its occupancy
      // (with 256-thread workgroups bigger tables lost more than pass 2 gained). 50 M rows -> 1 M groups 1.87 -> 1.56 ms,
      // Zipf(1.1) keys 1.69 -> 1.50 ms (QHIP_AGG_PART_WIDE=0: the 256-thread reduce pass)
      const bool wide = env_int("QHIP_AGG_PART_WIDE", 1) != 0 && !getenv("QHIP_AGG_LDS_BYTES") && plan.part_pr > 0 && (uint64_t)slot_bytes * 64 <= 128 * 1024;
      if (!getenv("QHIP_AGG_LDS_BYTES"))
        while ((uint64_t)l_nslots_p * 2 * slot_bytes <= (uint64_t)env_int("QHIP_AGG_PART_LDS_BYTES", wide ? 128 * 1024 : 0)) l_nslots_p *= 2;
      const uint32_t per_bin = std::max<uint32_t>(16, l_nslots_p * (l_nslots_p > l_nslots ? 5 : 3) / 8);   // groups a bin should hold
      uint32_t n_bins = 16;
      while (n_bins < 4096 && (uint64_t)n_bins * per_bin < std::max<uint32_t>(plan.last_groups, 1)) n_bins *= 2;
      if (env_int("QHIP_AGG_PART_BINS", 0) >= 16) n_bins = (uint32_t)pow2_ceil((uint64_t)std::min(4096, env_int("QHIP_AGG_PART_BINS", 0)));
      uint64_t g1 = std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)N + 255) / 256, (uint64_t)ctx->num_cus * (uint64_t)std::max(1, env_int("QHIP_AGG_PART_WGS_PER_CU", 4))));
      const uint64_t rows_per_wg = ((((uint64_t)N + g1 - 1) / g1) + 255) / 256 * 256;
      g1 = ((uint64_t)N + rows_per_wg - 1) / rows_per_wg;
      const uint64_t n_hist = (uint64_t)n_bins * g1;
      DevBuf hist((n_hist + 1) * 4), items_dev;
      DevBuf& records = pa_records;
      std::shared_ptr<Module> m_hist = get_module(ctx, plan.source, "qk_agg_part_hist");
      std::shared_ptr<Module> m_scat = get_module(ctx, plan.source, "qk_agg_part_scatter");
      std::shared_ptr<Module> m_red = get_module(ctx, plan.source, "qk_agg_reduce");
      HPartLaunch pl;
      pl.hist = hist.as<uint32_t>();
      pl.records = records.as<uint64_t>();
      pl.status = status_dev;
      pl.n_bins = n_bins;
      pl.rows_per_wg = (uint32_t)rows_per_wg;
      void* pargs[] = {&ka, &pl};
      QHIP_HIP_CHECK(hipModuleLaunchKernel(m_hist->fn, (unsigned)g1, 1, 1, 256, 1, 1, n_bins * 4, s, pargs, nullptr));
      exclusive_scan_u32(hist.as<uint32_t>(), hist.as<uint32_t>(), n_hist, hist.as<uint32_t>() + n_hist, s);
      // pass 2: LDS-staged (records of a tile ordered by bin, written out as runs) when a tile of records fits LDS
      const bool staged = plan.part_pr > 0 && env_int("QHIP_AGG_PART_STAGE", 1) != 0;
      if (staged) {
        std::shared_ptr<Module> m_stage = get_module(ctx, plan.source, "qk_agg_part_stage");
        const size_t tile = (size_t)1024 * (size_t)plan.part_pr;
        const size_t stage_lds = (size_t)n_bins * 12 + 8 + tile * ((size_t)(plan.slot_words - 1) * 8 + 2) + 16;
        QHIP_HIP_CHECK(hipModuleLaunchKernel(m_stage->fn, (unsigned)g1, 1, 1, 1024, 1, 1, (unsigned)stage_lds, s, pargs, nullptr));
      } else
        QHIP_HIP_CHECK(hipModuleLaunchKernel(m_scat->fn, (unsigned)g1, 1, 1, 256, 1, 1, n_bins * 4, s, pargs, nullptr));
      if (device_items) {
        // the bins' slices as work items, computed by the reduce kernel itself from the scanned histogram: a bin of `cnt`
        // records is cut into `slices` slices of at least 4 096 records (a heavy key's bin is aggregated by several
        // workgroups, each merging its LDS table into the HBM table); empty slices return at once
        HReduceLaunch rl;
        rl.records = records.as<uint64_t>();
        rl.item_first = nullptr;
        rl.slices = (uint32_t)std::max(1, env_int("QHIP_AGG_PART_SLICES", 8));
        rl.n_items = n_bins * rl.slices;
        rl.hist = hist.as<uint32_t>();
        rl.g1 = (uint32_t)g1; rl.n_bins = n_bins; rl.min_slice = 4096;
        HAggLaunch Lp = L;
        Lp.l_nslots = l_nslots_p;
        void* rargs[] = {&rl, &Lp};
        if (wide) {
          std::shared_ptr<Module> m_wide = get_module(ctx, plan.source, "qk_agg_reduce_wide");
          QHIP_HIP_CHECK(hipModuleLaunchKernel(m_wide->fn, std::min<unsigned>(rl.n_items, (unsigned)ctx->num_cus * 8), 1, 1, 1024, 1, 1, (unsigned)((size_t)l_nslots_p * slot_bytes), s, rargs, nullptr));
        } else {
          QHIP_HIP_CHECK(hipModuleLaunchKernel(m_red->fn, std::min<unsigned>(rl.n_items, (unsigned)ctx->num_cus * 16), 1, 1, 256, 1, 1, (unsigned)((size_t)l_nslots_p * slot_bytes), s, rargs, nullptr));
        }
        // (hist / records go back to the stream-ordered pool: no wait; the call's one synchronisation follows below)
      } else {
      // first record of every bin (= of its first workgroup's run) + the record total: one strided read-back
      uint32_t* first = (uint32_t*)((uint8_t*)ctx->pinned + 128);
      QHIP_HIP_CHECK(hipMemcpy2DAsync(first, 4, hist.ptr, (size_t)g1 * 4, 4, n_bins, hipMemcpyDeviceToHost, s));
      QHIP_HIP_CHECK(hipMemcpyAsync(first + n_bins, hist.as<uint32_t>() + n_hist, 4, hipMemcpyDeviceToHost, s));
      QHIP_HIP_CHECK(sync_stream(s));
      verify_pending_sizes(ctx);   // (an input of deferred size: did the joins below have room? — else QHIP_RETRY)
      // work items: a bin, or a slice of a big one (a heavy key's bin is aggregated by several workgroups, each merging
      // its LDS table into the HBM table: the key is merged once per slice, not once per row)
      // (work items of 128 k records for the wide reduce pass: 256 k is better for uniform keys, 64 k for skewed ones)
      const uint32_t max_item = (uint32_t)std::max(4096, env_int("QHIP_AGG_PARTITION_ITEM", wide ? 131072 : 32768));
      for (uint32_t b = 0; b < n_bins; ++b)
        for (uint32_t r = first[b]; r < first[b + 1]; r += max_item) item_first.push_back(r);
      const uint32_t n_items = (uint32_t)item_first.size();
      item_first.push_back(first[n_bins]);
      if (n_items) {
        items_dev.alloc(item_first.size() * 4);
        QHIP_HIP_CHECK(hipMemcpyAsync(items_dev.ptr, item_first.data(), item_first.size() * 4, hipMemcpyHostToDevice, s));
        HReduceLaunch rl;
        rl.records = records.as<uint64_t>();
        rl.item_first = items_dev.as<uint32_t>();
        rl.n_items = n_items;
        HAggLaunch Lp = L;
        Lp.l_nslots = l_nslots_p;
        void* rargs[] = {&rl, &Lp};
        if (wide) {
          std::shared_ptr<Module> m_wide = get_module(ctx, plan.source, "qk_agg_reduce_wide");
          const unsigned rgrid = (unsigned)std::min<uint64_t>(n_items, (uint64_t)ctx->num_cus * 2);
          QHIP_HIP_CHECK(hipModuleLaunchKernel(m_wide->fn, rgrid, 1, 1, 1024, 1, 1, (unsigned)((size_t)l_nslots_p * slot_bytes), s, rargs, nullptr));
        } else {
          const unsigned rgrid = (unsigned)std::min<uint64_t>(n_items, (uint64_t)ctx->num_cus * 4);
          QHIP_HIP_CHECK(hipModuleLaunchKernel(m_red->fn, rgrid, 1, 1, 256, 1, 1, (unsigned)((size_t)l_nslots_p * slot_bytes), s, rargs, nullptr));
        }
        QHIP_HIP_CHECK(sync_stream(s));   // hist / records / items go back to the pool here; item_first is pageable
      }
      }

    } else if (N > 0 && parts) {
      dense.alloc((size_t)guess * slot_bytes + 8);
      dense_dev = dense.as<uint64_t>();
      L.part_runs = parts->runs; L.part_stride = parts->stride;
      L.dense_out = dense_dev + 1; L.dense_counter = status_dev + 16; L.dense_cap = guess;
      // a part of more than 4x the average is sliced (heavy keys): at most n_parts / 4 + 1 slices in all
      L.n_parts = (uint32_t)parts->n_parts;
      L.part_max = (uint32_t)std::max<int64_t>(1024, env_int("QHIP_AGG_PARTS_MAX_FACTOR", 4) * ((N + parts->n_parts - 1) / parts->n_parts));
      const unsigned pgrid = grid + (unsigned)((uint64_t)N / L.part_max) + 1;
      std::shared_ptr<Module> pmod = get_module(ctx, plan.source, "qk_filter_agg_parts");
      QHIP_HIP_CHECK(hipModuleLaunchKernel(pmod->fn, pgrid, 1, 1, 1024, 1, 1, (unsigned)lds_bytes, ctx->stream, args, nullptr));
    } else if (N > 0) {
      if (wide) {
        std::shared_ptr<Module> wmod = get_module(ctx, plan.source, "qk_filter_agg_wide");
        QHIP_HIP_CHECK(hipModuleLaunchKernel(wmod->fn, grid, 1, 1, 1024, 1, 1, (unsigned)lds_bytes, ctx->stream, args, nullptr));
      } else if (cons) {
        std::shared_ptr<Module> cmod = get_module(ctx, plan.source, "qk_filter_agg_cons");
        QHIP_HIP_CHECK(hipModuleLaunchKernel(cmod->fn, grid, 1, 1, 256, 1, 1, (unsigned)lds_bytes, ctx->stream, args, nullptr));
      } else {
        QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, grid, 1, 1, 256, 1, 1, (unsigned)lds_bytes, ctx->stream, args, nullptr));
      }
    }
    time_mark(ctx, 1);
    uint32_t* const counter_dev = status_dev + 16;   // (page-locked mirror: status_pinned = pinned + 0, pre_host[0] = pinned + 64)
    pre_copied = 0;
    if (plan.W == 0) {
      QHIP_HIP_CHECK(hipMemcpyAsync(status_pinned, status_dev, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
      QHIP_HIP_CHECK(hipMemcpyAsync(pre_host, table_dev, (size_t)slot_bytes, hipMemcpyDeviceToHost, ctx->stream));
    } else {
      // speculative compaction right behind the kernel: [counter | dense slots]; the common case (few groups, no
      // overflow) then needs a single synchronisation for status + result
      if (use_arena) {
        // counter at +64 (zeroed with the arena), dense slots behind the table; the 8 bytes in front of the slots are a
        // copy target only in the read-back below, so read counter and slots separately
        dense_dev = (uint64_t*)(arena->as<uint8_t>() + zero_bytes) ;
        // the first pre_copied dense slots land in page-locked host memory straight from the compaction kernel (no
        // device-to-host copy of the slots behind it: that copy goes through the DMA engine, ~25 us with its hand-over gaps)
        pre_copied = std::min(PRE, guess);
        const bool direct = env_int("QHIP_AGG_PINNED_SLOTS", 1) != 0;
        launch_compact_slots(table_dev, total_slots, plan.slot_words, dense_dev + 1, counter_dev, guess, ctx->stream, direct ? pre_host + 1 : nullptr, pre_copied);
        QHIP_HIP_CHECK(hipMemcpyAsync(status_pinned, status_dev, 64 + 8, hipMemcpyDeviceToHost, ctx->stream));   // status + counter
        if (!direct) QHIP_HIP_CHECK(hipMemcpyAsync(pre_host + 1, dense_dev + 1, (size_t)pre_copied * slot_bytes, hipMemcpyDeviceToHost, ctx->stream));
      } else {
        if (!parts) { dense.alloc((size_t)guess * slot_bytes + 8); dense_dev = dense.as<uint64_t>(); }   // (parts: allocated in front of the kernel, which appends to it)
        // a plan that produced many groups last time will most likely do so again: assemble its output columns on the
        // device right away (the kernel reads the group count from the compaction counter) — one synchronisation in all,
        // and no slot crosses PCIe
        bool utf8_key = false;
        for (auto& kd : plan.keys) utf8_key = utf8_key || kd.type.id == QHIP_UTF8;
        spec_enqueued = false;
        const bool will_spec = replicas == 1 && plan.last_groups >= dev_threshold && !utf8_key && env_int("QHIP_AGG_NO_SPECULATIVE_FINALIZE", 0) == 0;
        // (a host-side result: its first dense slots go to page-locked host memory straight from the compaction kernel)
        // (parts: the workgroups appended most slots themselves — the first ones are copied out of the dense buffer instead)
        const bool direct = !will_spec && !parts && env_int("QHIP_AGG_PINNED_SLOTS", 1) != 0;
        launch_compact_slots(table_dev, total_slots, plan.slot_words, dense_dev + 1, counter_dev, guess, ctx->stream, direct ? pre_host + 1 : nullptr,
                             std::min(PRE, guess));
        if (will_spec) {
          spec = DevFinal();
          const int ncols = n_groups + n_aggs;
          enqueue_device_finalize(spec, dense_dev + 1, guess, counter_dev, status_dev + 32, status_pinned + 32);
          spec_enqueued = true;
          // ONE read-back: status + counter + the finalisation's status and null counts
          QHIP_HIP_CHECK(hipMemcpyAsync(status_pinned, status_dev, (size_t)(32 + QS_WORDS + ncols) * 4, hipMemcpyDeviceToHost, ctx->stream));
        } else {
          QHIP_HIP_CHECK(hipMemcpyAsync(status_pinned, status_dev, 64 + 8, hipMemcpyDeviceToHost, ctx->stream));   // status + counter
          pre_copied = std::min(PRE, guess);
          if (!direct) QHIP_HIP_CHECK(hipMemcpyAsync(pre_host + 1, dense_dev + 1, (size_t)pre_copied * slot_bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
      }
    }
    if (use_arena) {
      // wait for the read-backs only; the arena is zeroed for the next call behind them
      QHIP_HIP_CHECK(hipEventRecord(ctx->ev[2], ctx->stream));
      QHIP_HIP_CHECK(hipMemsetAsync(arena->ptr, 0, zero_bytes, ctx->stream));
      plan.arena_clean = true;
      mark("launched");
      QHIP_HIP_CHECK(sync_event(ctx->ev[2]));
    } else {
      mark("launched");
      QHIP_HIP_CHECK(sync_stream(ctx->stream));
    }
    memcpy(status, status_pinned, sizeof(status));
    if (env_int("QHIP_AGG_PROF", 0)) {   // phase timers of the fused kernel (P::PROF): mean cycles per wavefront, in units of 256
      const double waves = (double)grid * (block / 64);
      fprintf(stderr, "[qhip agg prof] rows %lld grid %u x %d: loads+eval %.0f  cache %.0f  table updates %.0f  cached keys -> table %.0f  merge %.0f  (x256 cycles per wavefront)\n",
              (long long)N, grid, block, status_pinned[8] / waves, status_pinned[9] / waves, status_pinned[10] / waves, status_pinned[11] / waves, status_pinned[12] / waves);
    }
    mark("synchronised");
    trace_point("aggregate: back from its wait");
    verify_pending_sizes(ctx);   // (an input of deferred size: did the joins below have room? — else QHIP_RETRY)
    // a join of deferred size that turned out to have produced nothing has no output batches (hash_join.rs:363-372)
    if (in->rows_dev && in->deferred_count() == 0 && n_groups > 0) return no_batches_out();
    if (ctx->timing) QHIP_HIP_CHECK(hipEventElapsedTime(&main_ms, ctx->ev[0], ctx->ev[1]));
    check_status_words(status);
    if (!status[QS_OVERFLOW]) break;
    if (replicas == 1 && cap >= cap_max) fail(QHIP_HIP_ERROR, "group table overflow at maximum capacity (internal error)");
    cap = replicas > 1 ? std::min<uint32_t>(cap_max, std::max<uint32_t>(cap * 16, 1u << 18)) : (uint32_t)std::min<uint64_t>((uint64_t)cap * 16, cap_max);
    replicas = 1;
    ++retries;
  }

  const uint32_t lds_used = status[QS_LDS_USED];          // (the status words are reused by the output assembly below)
  const bool lds_spilled = status[QS_LDS_SPILL] != 0;
  // ---- dense slots -> host (few groups) or kept on the device (many groups)
  DevBuf dense_keep;                                   // [counter | dense slots] when the output is assembled on the device
  if (plan.W == 0) {
    G = 1;
    slots.assign(pre_host, pre_host + plan.slot_words);
  } else {
    G = (uint32_t)pre_host[0];
    plan.last_dense = G;
    const uint32_t total_slots = cap * replicas;
    // (the speculative assembly holds what fitted the buffer it was enqueued with: after a re-compaction its columns are short —
    // a plan whose remembered group count came from a much smaller table, the same plan key at another scale factor)
    const bool spec_fits = G <= guess;
    if (G > guess) {
      // more groups than the speculative buffer holds: compact again with the exact size
      guess = G;
      dense.alloc((size_t)guess * slot_bytes + 8);
      dense_dev = dense.as<uint64_t>();
      launch_compact_slots(table_dev, total_slots, plan.slot_words, dense_dev + 1, zeroed_block(ctx), guess, ctx->stream);
      QHIP_HIP_CHECK(sync_stream(ctx->stream));
      pre_copied = 0;
    }
    if (spec_enqueued && spec_fits && replicas == 1 && G >= dev_threshold) {
      // the speculative device-side assembly is the result
      plan.last_groups = G; plan.learnt_at = ++g_learn_tick;
      if (ctx->agg_group_hints.size() > 4096) ctx->agg_group_hints.clear();
      ctx->agg_group_hints[hint_key] = G;
      qhip_table* result = finish_device_finalize(spec, dense_dev + 1, G, true);
      ctx->stats.main_kernel_ms = main_ms;
      ctx->stats.total_device_ms = main_ms;
      ctx->stats.rows_in = N;
      ctx->stats.rows_out = G;
      ctx->stats.groups = G;
      ctx->stats.table_capacity = (int64_t)cap * replicas;
      ctx->stats.retries = retries;
      ctx->stats.lds_table_slots = (int32_t)l_nslots;
      ctx->stats.bytes_per_row_read = bytes_per_row;
      ctx->stats.workgroups = (int32_t)grid;
      ctx->stats.lds_spilled = lds_spilled ? 1 : 0;
      ctx->stats.hbm_table_load = cap ? (double)G / ((double)cap * replicas) : 0.0;
      ctx->stats.lds_occupancy = (env_int("QHIP_AGG_STATS", 0) && l_nslots) ? (double)lds_used / ((double)grid * l_nslots) : -1.0;
      snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "%s", ran_partitioned ? "qk_agg_part_hist+scatter+reduce" : parts ? "qk_part_ids+qk_part_scatter+qk_filter_agg_parts" : ran_runs ? "qk_agg_runs" : cons ? "qk_filter_agg_cons" : plan.kernel_name.c_str());
      return result;
    }
    if (replicas == 1 && G >= dev_threshold) {
      dense_keep = std::move(dense);
    } else if (G <= pre_copied) {
      slots.assign(pre_host + 1, pre_host + 1 + (size_t)G * plan.slot_words);
    } else {
      slots.resize((size_t)G * plan.slot_words);
      copy_sync(ctx->stream, slots.data(), dense_dev + 1, (size_t)G * slot_bytes, hipMemcpyDeviceToHost);
    }
    if (replicas > 1 && G > 1) {
      // merge the replicas: same key words -> one slot; every cell is a commutative monoid (wrapping adds, max)
      std::map<std::vector<uint64_t>, uint32_t> seen;   // only consulted once there are many distinct keys
      uint32_t out = 0;
      for (uint32_t g = 0; g < G; ++g) {
        uint64_t* src = &slots[(size_t)g * plan.slot_words];
        uint32_t found = out;
        if (out <= 16) {
          for (uint32_t k = 0; k < out; ++k)
            if (!memcmp(&slots[(size_t)k * plan.slot_words + 1], src + 1, (size_t)plan.W * 8)) { found = k; break; }
          if (found == out && out == 16)   // growing past the linear-search regime: index what we have
            for (uint32_t k = 0; k < out; ++k) {
              const uint64_t* ks = &slots[(size_t)k * plan.slot_words + 1];
              seen.emplace(std::vector<uint64_t>(ks, ks + plan.W), k);
            }
        } else {
          auto it = seen.find(std::vector<uint64_t>(src + 1, src + 1 + plan.W));
          if (it != seen.end()) found = it->second;
        }
        if (found == out) {
          if (out >= 16) seen.emplace(std::vector<uint64_t>(src + 1, src + 1 + plan.W), out);
          if (out != g) memcpy(&slots[(size_t)out * plan.slot_words], src, (size_t)slot_bytes);
          ++out;
          continue;
        }
        uint64_t* dst = &slots[(size_t)found * plan.slot_words] + 1 + plan.W;
        const uint64_t* sc = src + 1 + plan.W;
        for (auto& cd : plan.cells) {
          switch (cd.kind) {
            case CELL_ROWS: case CELL_CNT: case CELL_SUM_U64: dst[cd.off] += sc[cd.off]; break;
            case CELL_SUM_I128: {
              const u128 a = ((u128)dst[cd.off + 1] << 64) | dst[cd.off], b2 = ((u128)sc[cd.off + 1] << 64) | sc[cd.off], r = a + b2;
              dst[cd.off] = (uint64_t)r; dst[cd.off + 1] = (uint64_t)(r >> 64);
              break;
            }
            case CELL_SUM_F64: { double x, y; memcpy(&x, &dst[cd.off], 8); memcpy(&y, &sc[cd.off], 8); x += y; memcpy(&dst[cd.off], &x, 8); break; }
            case CELL_MAXORD64: dst[cd.off] = std::max(dst[cd.off], sc[cd.off]); break;
            case CELL_MAXORD128: {
              const u128 a = ((u128)dst[cd.off + 1] << 64) | dst[cd.off], b2 = ((u128)sc[cd.off + 1] << 64) | sc[cd.off];
              if (b2 > a) { dst[cd.off] = sc[cd.off]; dst[cd.off + 1] = sc[cd.off + 1]; }
              break;
            }
          }
        }
      }
      G = out;
      slots.resize((size_t)G * plan.slot_words);
    }
  }

  plan.last_groups = G; plan.learnt_at = ++g_learn_tick;
  if (ctx->agg_group_hints.size() > 4096) ctx->agg_group_hints.clear();
  ctx->agg_group_hints[hint_key] = G;
  auto set_stats = [&]() {
    ctx->stats.main_kernel_ms = main_ms;
    ctx->stats.total_device_ms = main_ms;
    ctx->stats.rows_in = N;
    ctx->stats.rows_out = G;
    ctx->stats.groups = G;
    ctx->stats.table_capacity = (int64_t)cap * replicas;
    ctx->stats.retries = retries;
    ctx->stats.lds_table_slots = (int32_t)l_nslots;
    ctx->stats.bytes_per_row_read = bytes_per_row;
    ctx->stats.workgroups = (int32_t)grid;
    ctx->stats.lds_spilled = lds_spilled ? 1 : 0;
    ctx->stats.hbm_table_load = cap ? (double)G / ((double)cap * replicas) : 0.0;
    ctx->stats.lds_occupancy = (env_int("QHIP_AGG_STATS", 0) && l_nslots) ? (double)lds_used / ((double)grid * l_nslots) : -1.0;
    snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "%s", ran_partitioned ? "qk_agg_part_hist+scatter+reduce" : parts ? "qk_part_ids+qk_part_scatter+qk_filter_agg_parts" : ran_runs ? "qk_agg_runs" : cons ? "qk_filter_agg_cons" : plan.kernel_name.c_str());
  };
  if (dense_keep.ptr) {
    // ---- many groups: assemble the output columns on the device (k_agg_finalize), nothing crosses PCIe
    DevFinal fin;
    enqueue_device_finalize(fin, dense_keep.as<uint64_t>() + 1, G, nullptr, nullptr, nullptr);
    qhip_table* result = finish_device_finalize(fin, dense_keep.as<uint64_t>() + 1, G, false);
    set_stats();
    return result;
  }

  // ---- few groups: assemble the output columns on the host (GroupAccumulator::output, hash.rs:89-107; accumulator evaluate())
  std::vector<HostColumn> cols((size_t)(n_groups + n_aggs));
  for (int k = 0; k < n_groups; ++k) {
    const KeyDesc& kd = plan.keys[(size_t)k];
    HostColumn& hc = cols[(size_t)k];
    hc.init_fixed(kd.type, G);
    for (uint32_t g = 0; g < G; ++g) {
      const uint64_t* slot = &slots[(size_t)g * plan.slot_words];
      const bool is_null = plan.null_mask_word && ((slot[1] >> k) & 1);
      const uint64_t w0 = slot[1 + kd.word_off];
      if (kd.type.id == QHIP_UTF8) {
        const uint64_t* kw = slot + 1 + kd.word_off;
        const int len = is_null ? 0 : (int)(kw[kd.words - 1] >> 56);
        for (int b = 0; b < len; ++b) hc.data.push_back((uint8_t)(kw[b >> 3] >> (8 * (b & 7))));
        hc.offsets[(size_t)g + 1] = (int32_t)hc.data.size();
      } else if (kd.type.id == QHIP_DECIMAL128) {
        hc.as<uint64_t>()[2 * (size_t)g] = w0;
        hc.as<uint64_t>()[2 * (size_t)g + 1] = slot[1 + kd.word_off + 1];
      } else {
        const int w = dtype_width(kd.type);
        memcpy(hc.values.data() + (size_t)g * w, &w0, (size_t)w);   // little-endian truncation of the sign-extended word
      }
      if (is_null) hc.set_null(g);
    }
  }
  for (int k = 0; k < n_aggs; ++k) {
    const AggDesc& ad = plan.aggs[(size_t)k];
    HostColumn& hc = cols[(size_t)(n_groups + k)];
    hc.init_fixed(ad.ret, G);
    for (uint32_t g = 0; g < G; ++g) {
      const uint64_t* cell = &slots[(size_t)g * plan.slot_words + cell0];
      const uint64_t nonnull = cell[plan.cells[(size_t)ad.count_cell].off];
      const uint64_t* vc = ad.value_cell >= 0 ? cell + plan.cells[(size_t)ad.value_cell].off : nullptr;
      switch (ad.kind) {
        case QHIP_AGG_COUNT:   // count.rs:36-48
          hc.as<int64_t>()[g] = (int64_t)nonnull;
          break;
        case QHIP_AGG_SUM:     // sum.rs:71-103: None until a non-null value was seen
          if (!nonnull) { hc.set_null(g); break; }
          if (ad.ret.id == QHIP_DECIMAL128) { hc.as<uint64_t>()[2 * (size_t)g] = vc[0]; hc.as<uint64_t>()[2 * (size_t)g + 1] = vc[1]; }
          else hc.as<uint64_t>()[g] = vc[0];   // Int64 / UInt64 wrapping sum, Float64 bit pattern
          break;
        case QHIP_AGG_AVG: {
          if (!nonnull) { hc.set_null(g); break; }
          if (ad.ret.id == QHIP_FLOAT64) {   // avg.rs:63-78
            double s; memcpy(&s, vc, 8);
            hc.as<double>()[g] = s / (double)nonnull;
            break;
          }
          // avg.rs:91-116 (DecimalAvgAccumulator::evaluate)
          const DType& at = plan.args[(size_t)ad.arg].type;
          const i128 sum = (i128)(((u128)vc[1] << 64) | (u128)vc[0]);
          if (ad.ret.scale < at.scale) fail(QHIP_EXEC_ERROR, "Internal error: Arithmetic Overflow in DecimalAvgAccumulator");
          const i128 mul = pow10_i128(ad.ret.scale - at.scale);
          i128 value;
          if (__builtin_mul_overflow(sum, mul, &value)) fail(QHIP_EXEC_ERROR, "AVG(Decimal128): sum * 10^k overflows i128 (reference yields a mistyped NULL, avg.rs:105-116)");
          const i128 lim = pow10_i128(ad.ret.precision);
          if (value >= lim || value <= -lim)
            fail(QHIP_EXEC_ERROR, "AVG(Decimal128): scaled sum exceeds " + dtype_name(ad.ret) + " (reference yields a mistyped NULL, avg.rs:105-116)");
          const i128 q = value / (i128)nonnull;   // truncating, like i128::div_wrapping
          hc.as<uint64_t>()[2 * (size_t)g] = (uint64_t)(u128)q;
          hc.as<uint64_t>()[2 * (size_t)g + 1] = (uint64_t)((u128)q >> 64);
          break;
        }
        case QHIP_AGG_MIN:
        case QHIP_AGG_MAX: {
          // PrimitiveAccumulator (aggregate/mod.rs:28-84): seeded with NATIVE::MAX / MIN, Some() as soon as
          // accumulate ran once — i.e. for every existing group, and for NoGrouping whenever a batch arrived.
          const bool is_min = ad.kind == QHIP_AGG_MIN;
          if (plan.W == 0 && zero_batches_in) { hc.set_null(g); break; }
          const DType& t = ad.ret;
          if (t.id == QHIP_DECIMAL128) {
            u128 o = ((u128)vc[1] << 64) | (u128)vc[0];
            if (is_min) o = ~o;
            const u128 v = o ^ ((u128)1 << 127);
            hc.as<uint64_t>()[2 * (size_t)g] = (uint64_t)v;
            hc.as<uint64_t>()[2 * (size_t)g + 1] = (uint64_t)(v >> 64);
          } else if (dtype_is_float(t)) {
            uint64_t o = is_min ? ~vc[0] : vc[0];
            double v = (vc[0] == 0) ? (is_min ? DBL_MAX : -DBL_MAX) : ord_to_f64(o);
            if (t.id == QHIP_FLOAT32) {
              const float lim = FLT_MAX;
              float fv = (float)v;
              if (vc[0] == 0 || std::isnan(fv)) fv = is_min ? lim : -lim;
              if (is_min && fv > lim) fv = lim;
              if (!is_min && fv < -lim) fv = -lim;
              hc.as<float>()[g] = fv;
            } else {
              if (std::isnan(v)) v = is_min ? DBL_MAX : -DBL_MAX;
              if (is_min && v > DBL_MAX) v = DBL_MAX;
              if (!is_min && v < -DBL_MAX) v = -DBL_MAX;
              hc.as<double>()[g] = v;
            }
          } else {
            uint64_t o = is_min ? ~vc[0] : vc[0];
            const bool sgn = dtype_is_signed(t) || t.id == QHIP_DATE32 || t.id == QHIP_DATE64 || (t.id >= QHIP_TIME32_S && t.id <= QHIP_TIMESTAMP_NS);
            uint64_t raw = sgn ? (o ^ 0x8000000000000000ULL) : o;
            const int w = dtype_width(t);
            // no non-null value seen (an all-zero cell): the seed of the column's OWN type (i32::MAX, not i64::MAX truncated)
            if (vc[0] == 0 && sgn && w < 8) raw = is_min ? ((1ULL << (8 * w - 1)) - 1) : (1ULL << (8 * w - 1));
            memcpy(hc.values.data() + (size_t)g * w, &raw, (size_t)w);
          }
          break;
        }
      }
    }
  }
  set_stats();
  mark("assembled");
  qhip_table* result = table_from_host(ctx, names, nullable, cols, G, false);
  mark("result table");
  return result;
}

}  // namespace qhip

using namespace qhip;

extern "C" int qhip_hash_aggregate_execute(qhip_ctx* ctx, const qhip_table* input, const qhip_expr* exprs, int32_t n_exprs,
                                           int32_t predicate_root, const int32_t* group_roots, int32_t n_groups, const qhip_agg* aggs,
                                           int32_t n_aggs, const char* const* out_names, qhip_table** out) {
  if (!ctx || !input || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] {
    try {
      std::unique_ptr<qhip_table> r(hash_aggregate(ctx, input, exprs, n_exprs, predicate_root, group_roots, n_groups, aggs, n_aggs, out_names));
      if (!ctx->pending_sizes.empty()) {   // (a path that never waited: the joins of deferred size below are checked all the same)
        QHIP_HIP_CHECK(sync_stream(ctx->stream));
        verify_pending_sizes(ctx);
      }
      *out = r.release();
      trace_point("aggregate: return");
    } catch (...) {
      ctx->pending_sizes.clear();
      throw;
    }
  });
}
