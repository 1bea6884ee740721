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
#include "jit.hpp"
#include "kargs_host.hpp"
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

}  // namespace

namespace qhip {
// ---- the fused path (round 4): scan filter + key -> part byte per row + per-wavefront histogram (qk_part_ids, generated) ->
// scan -> ONE host wait (the parts' sizes + the status words) -> k_part_scatter moves only the columns `keep` names, each into
// ONE buffer over all parts; a part's column is a slice (view) of it. Columns the scatter kernel cannot move itself
// (validity bitmaps, Booleans, strings) are gathered per part through the selection vector the same kernel leaves behind.
// A deferred gather whose source is plain is read THROUGH its index vector in both passes (never materialised).
struct PartitionPlan { KeysPlan kp; std::shared_ptr<Module> mod, mod_wide; DevBuf strlit; };

// pass 2 runs its workgroup-cooperative form (few output streams, bursts of kilobytes) from 2^20 rows on
bool scatter_wg_form(int64_t rows) {
  const int wg_mode = env_int("QHIP_PART_SCATTER_WG", 1);   // 0 never, 1 big inputs, 2 always (tests)
  return env_int("QHIP_PART_SCATTER_JIT", 1) != 0 && (wg_mode == 2 || (wg_mode == 1 && rows >= (1 << 20)));
}


void partition_pass1(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n_keys, int pred_root,
                     int n_parts, PartitionWork& w, const int64_t* range_bounds) {
  hipStream_t s = ctx->stream;
  const uint64_t N = (uint64_t)in->num_rows;
  resolve_referenced(ctx, in, exprs, n_exprs, true);
  std::vector<InputCol> icols = input_cols_of(in, true);
  // partition ids must agree across ranks and across the two join sides: Utf8 keys always hash as 4 words (<= 31 bytes)
  for (int k = 0; k < n_keys; ++k)
    if (roots[k] >= 0 && roots[k] < n_exprs && exprs[roots[k]].kind == QHIP_EXPR_COLUMN && exprs[roots[k]].column >= 0 &&
        exprs[roots[k]].column < (int)icols.size() && icols[(size_t)exprs[roots[k]].column].type.id == QHIP_UTF8)
      icols[(size_t)exprs[roots[k]].column].utf8_max_len = 31;
  // a resident table's Int64 key column whose values fit 32 bits is read through its 4-byte narrow copy (made at the second big
  // read of the column, relops.cpp ensure_narrow_int_columns — TPC-H's order and customer keys)
  ensure_narrow_int_columns(ctx, in, exprs, n_exprs, icols, (int64_t)env_int("QHIP_STATS_MIN_ROWS", 1 << 22));
  std::string pkey = "partition|";
  {
    auto put = [&](const void* p, size_t n) { pkey.append((const char*)p, n); };
    for (auto& ic : icols) {
      const int v[8] = {ic.type.id, ic.type.precision, ic.type.scale, ic.has_nulls ? 1 : 0, ic.utf8_max_len, ic.utf8_fixed1 ? 1 : 0, ic.indirect ? 1 : 0, ic.narrow_bytes};
      put(v, sizeof v);
    }
    put("|", 1);
    for (int k = 0; k < n_exprs; ++k) {
      qhip_expr e = exprs[k];
      const char* str = e.lit_str; const int64_t len = e.lit_len;
      e.lit_str = nullptr;
      put(&e, sizeof e);
      if (str && len > 0 && e.kind == QHIP_EXPR_LITERAL) put(str, (size_t)len);
    }
    put("|", 1);
    put(roots, sizeof(int32_t) * (size_t)n_keys);
    const int v[3] = {pred_root, n_parts, in->rows_dev ? 1 : 0};
    put(v, sizeof v);
  }
  std::shared_ptr<PartitionPlan> pp;
  {
    auto cached = ctx->plan_cache.find(pkey);
    if (cached != ctx->plan_cache.end()) pp = std::static_pointer_cast<PartitionPlan>(cached->second);
    else {
      pp = std::make_shared<PartitionPlan>();
      ExprSet es;
      es.build(exprs, n_exprs, icols);
      if (pred_root >= n_exprs) fail(QHIP_INVALID_ARGUMENT, "qhip_partition_filtered: predicate index out of range");
      plan_keys(es, icols, roots, n_keys, pp->kp, pred_root, KEYS_KERNEL_PARTITION, in->rows_dev != nullptr, n_parts);
      if (ctx->plan_cache.size() > 4096) ctx->plan_cache.clear();
      ctx->plan_cache[pkey] = pp;
    }
  }
  if (!pp->mod) pp->mod = get_module(ctx, pp->kp.source, pp->kp.kernel_name);
  for (size_t k = 0; k < pp->kp.bind.cols.size(); ++k) {
    const int c = pp->kp.bind.cols[k];
    const int wd = dtype_width(icols[(size_t)c].type);
    w.pass1_bytes_per_row += icols[(size_t)c].indirect ? wd + 4 : icols[(size_t)c].narrow_bytes ? icols[(size_t)c].narrow_bytes : wd > 0 ? wd : 8;
  }
  // units of work = wavefronts with a static row range of whole tiles (256 rows in both passes); enough of them for ~8 rounds
  // of the chip's resident wavefronts, few enough that the histogram is scanned by ONE launch (<= 64 k counters) when the
  // table is not huge
  // Big inputs (pass 2 runs its workgroup form): a unit is a WORKGROUP of pass 1 whose four wavefronts count a quarter of its
  // rows each — a quarter of the counters to scan (k_scan_small: one workgroup, ~20 us for 64 k counters).
  w.wg_units = scatter_wg_form(in->num_rows);
  const uint64_t tile = w.wg_units ? 4 * (uint64_t)std::max(64, env_int("QHIP_PART_UNIT_QUANTUM", 256)) : 256;   // (a workgroup unit: four wavefronts x whole tiles)
  uint64_t want = std::max<uint64_t>(1, (uint64_t)ctx->num_cus * (w.wg_units ? 8 : 32));
  if ((uint64_t)n_parts * want > 65536 && 65536 / (uint64_t)n_parts >= (uint64_t)ctx->num_cus * 4) want = 65536 / (uint64_t)n_parts;
  uint64_t rpu = std::max<uint64_t>(w.wg_units ? tile : tile * 4, ((N + want - 1) / want + tile - 1) / tile * tile);
  if (env_int("QHIP_PART_ROWS_PER_UNIT", 0) > 0) rpu = (uint64_t)(env_int("QHIP_PART_ROWS_PER_UNIT", 0) + (int)tile - 1) / tile * tile;
  w.rows_per_unit = (uint32_t)rpu;
  w.n_units = (uint32_t)std::max<uint64_t>(1, (N + rpu - 1) / rpu);
  const size_t nh = (size_t)n_parts * w.n_units;
  DevBuf hist((nh + 4) * 4);
  w.ids.alloc(N + 64);
  w.runs.alloc((nh + 4) * 4);
  w.starts.alloc(((size_t)n_parts + 1) * 4);
  w.dstat = zeroed_block(ctx);
  HKArgs ka;
  fill_kargs(ctx, in, pp->kp.bind, ka, pp->strlit);
  HPartIdsLaunch pl = {w.ids.as<uint8_t>(), hist.as<uint32_t>(), w.dstat, w.n_units, w.rows_per_unit, w.wg_units ? 1u : 0u, 0u, nullptr};
  if (range_bounds && n_parts > 1) {
    // by key range: ONE integer-like key (its word is the sign-extended value)
    if (n_keys != 1) fail(QHIP_INVALID_ARGUMENT, "partitioning by key range takes exactly one key");
    const KeyDesc& kd = pp->kp.keys.at(0);
    if (kd.words != 1 || kd.type.id == QHIP_UTF8 || kd.type.id == QHIP_DECIMAL128) fail(QHIP_UNSUPPORTED, "partitioning by key range needs an integer-like key, not " + dtype_name(kd.type));
    for (int b = 1; b < n_parts - 1; ++b)
      if (range_bounds[b] < range_bounds[b - 1]) fail(QHIP_INVALID_ARGUMENT, "partitioning by key range: the bounds must be ascending");
    w.bounds_host.assign(range_bounds, range_bounds + (n_parts - 1));
    w.bounds_dev.alloc((size_t)(n_parts - 1) * 8);
    QHIP_HIP_CHECK(hipMemcpyAsync(w.bounds_dev.ptr, w.bounds_host.data(), (size_t)(n_parts - 1) * 8, hipMemcpyHostToDevice, s));
    pl.bounds = w.bounds_dev.as<int64_t>();
  }
  void* args[] = {&ka, &pl};
  time_mark(ctx, 0);
  // (the wide form needs whole tiles to shift a partial last tile back over: an exact row count of at least one tile)
  const bool wide = env_int("QHIP_PART_WIDE", 1) != 0 && !in->rows_dev && N >= 256 * 8;
  if (wide && !pp->mod_wide) pp->mod_wide = get_module(ctx, pp->kp.source, "qk_part_ids_wide");
  QHIP_HIP_CHECK(hipModuleLaunchKernel((wide ? pp->mod_wide : pp->mod)->fn, w.wg_units ? w.n_units : (w.n_units + 3) / 4, 1, 1, 256, 1, 1, 0, s, args, nullptr));
  time_mark(ctx, 4);
  exclusive_scan_u32(hist.as<uint32_t>(), w.runs.as<uint32_t>(), nh, w.runs.as<uint32_t>() + nh, s);
  launch_gather_stride_u32(w.runs.as<uint32_t>(), w.n_units, (uint32_t)n_parts + 1, w.starts.as<uint32_t>(), s);   // (entry n_parts = the scan's total)
}

// pass 2's launches: the plain kept columns into `moved` (one buffer each over all parts), the others listed in `odd` with the
// parts' selection vector in `sel` (the caller gathers them per part)
void partition_scatter(Ctx* ctx, const qhip_table* in, const int32_t* keep, int n_parts, PartitionWork& w, uint64_t total,
                       std::vector<MovedColumn>& moved, std::vector<size_t>& odd, std::shared_ptr<DevBuf>& sel, bool rows_only) {
  hipStream_t s = ctx->stream;
  typedef MovedColumn Moved;
  // one generated kernel per group of columns (qh_part_scatter_body: the registers of two tiles' values bound the group: <= 48
  // bytes per row, <= 8 columns); the AOT kernel k_part_scatter (any shape, QHIP_PART_SCATTER_JIT=0) is the plan-independent form
  const bool jit = env_int("QHIP_PART_SCATTER_JIT", 1) != 0;
  const int budget = std::max(16, env_int("QHIP_PART_SCATTER_BYTES", 48));
  struct Pending { const void* src; const uint32_t* idx; void* out; int width; };
  std::vector<Pending> group;
  int group_bytes = 0;
  bool marked = false;
  auto flush = [&] {
    if (group.empty()) return;
    if (total) {
      if (!marked) { time_mark(ctx, 2); marked = true; }
      if (jit) {
        std::vector<int> widths; std::vector<char> ind;
        for (auto& g : group) { widths.push_back(g.src || g.width != 4 ? g.width : 0); ind.push_back(g.idx ? 1 : 0); }
        std::string key = "part_scatter|" + std::to_string(rows_only ? -1 : n_parts <= 8 ? 8 : n_parts <= 16 ? 16 : 0) + (in->rows_dev ? "|d" : "|h");
        for (size_t k = 0; k < widths.size(); ++k) key += "|" + std::to_string(widths[k]) + (ind[k] ? "i" : "");
        struct ScatterPlan { PartScatterPlan sp; std::shared_ptr<Module> mod, mod_wg; };
        std::shared_ptr<ScatterPlan> sp;
        auto cached = ctx->plan_cache.find(key);
        if (cached != ctx->plan_cache.end()) sp = std::static_pointer_cast<ScatterPlan>(cached->second);
        else {
          sp = std::make_shared<ScatterPlan>();
          plan_part_scatter(widths, ind, n_parts, in->rows_dev != nullptr, sp->sp, rows_only);
          sp->mod = get_module(ctx, sp->sp.source, sp->sp.kernel_name);
          ctx->plan_cache[key] = sp;
        }
        HPartScatterLaunch L;
        memset(&L, 0, sizeof L);
        L.ids = w.ids.as<uint8_t>(); L.runs = w.runs.as<uint32_t>();
        L.nrows = in->num_rows; L.nrows_dev = in->rows_dev;
        L.n_units = w.n_units; L.rows_per_unit = w.rows_per_unit; L.n_parts = (uint32_t)n_parts; L.sub = 1;
        L.trash = w.trash.ptr;
        for (size_t k = 0; k < group.size(); ++k) { L.src[k] = group[k].src; L.idx[k] = group[k].idx; L.out[k] = group[k].out; }
        void* args[] = {&L};
        // the workgroup-cooperative form: a workgroup takes `sub` consecutive pass-1 units: ~3 workgroups per CU over the
        // table (measured: 2-3 best, 6 / 12 cost 7-9 % — more streams), at least 4 tiles each
        if (w.wg_units) {
          if (!sp->mod_wg) sp->mod_wg = get_module(ctx, sp->sp.source, "qk_part_scatter_wg");
          const uint64_t tile = (uint64_t)sp->sp.wg_threads * (uint64_t)sp->sp.rows_per_lane;
          const uint64_t want_wgs = (uint64_t)std::max(1, env_int("QHIP_PART_SCATTER_WGS_PER_CU", 3)) * (uint64_t)ctx->num_cus;
          uint64_t rows_wg = std::max<uint64_t>(4 * tile, ((uint64_t)in->num_rows + want_wgs - 1) / want_wgs);
          const uint32_t sub = (uint32_t)std::max<uint64_t>(1, (rows_wg + w.rows_per_unit - 1) / w.rows_per_unit);
          L.sub = sub;
          const unsigned wgs = (w.n_units + sub - 1) / sub;
          QHIP_HIP_CHECK(hipModuleLaunchKernel(sp->mod_wg->fn, wgs, 1, 1, (unsigned)sp->sp.wg_threads, 1, 1, 0, s, args, nullptr));
        } else
        QHIP_HIP_CHECK(hipModuleLaunchKernel(sp->mod->fn, (w.n_units + 3) / 4, 1, 1, 256, 1, 1, 0, s, args, nullptr));
      } else {
        PartScatterArgs A;
        memset(&A, 0, sizeof A);
        A.ids = w.ids.as<uint8_t>(); A.runs = w.runs.as<uint32_t>();
        A.nrows = (uint64_t)in->num_rows; A.nrows_dev = in->rows_dev;
        A.n_units = w.n_units; A.rows_per_unit = w.rows_per_unit; A.n_parts = (uint32_t)n_parts;
        for (auto& g : group) A.cols[A.n_cols++] = PartCol{g.src, g.idx, g.out, (uint32_t)g.width, g.src ? 0u : 1u};
        launch_part_scatter(A, s);
      }
    }
    group.clear();
    group_bytes = 0;
  };
  auto push = [&](const void* src, const uint32_t* idx, void* out, int width) {
    if (!group.empty() && (group_bytes + width > budget || (int)group.size() == (jit ? kPartMaxCols : kPartCols))) flush();
    group.push_back(Pending{src, idx, out, width});
    group_bytes += width;
  };
  for (size_t c = 0; c < in->cols.size() && !rows_only; ++c) {
    const DevColumn& c0 = in->cols[c];
    if ((keep && !keep[c]) || c0.type.id == QHIP_NULL) continue;
    const int width = dtype_width(c0.type);
    if (indirect_eligible(c0)) {
      auto out = std::make_shared<DevBuf>((size_t)total * (size_t)width);
      push(c0.deferred->src.values->ptr, c0.deferred->idx->as<uint32_t>(), out->ptr, width);
      moved.push_back(Moved{c, out, width});
      continue;
    }
    const DevColumn& rc = resolved(ctx, c0);
    if (width > 0 && rc.null_count == 0 && rc.values) {
      auto out = std::make_shared<DevBuf>((size_t)total * (size_t)width);
      push(rc.values->ptr, nullptr, out->ptr, width);
      moved.push_back(Moved{c, out, width});
    } else odd.push_back(c);
  }
  if (!odd.empty() || rows_only) {
    sel = std::make_shared<DevBuf>(((size_t)total + 1) * 4);
    // (rows_only: the consumer reads sel[0 .. capacity) before it knows the row count: no garbage row numbers behind the parts)
    if (rows_only) QHIP_HIP_CHECK(hipMemsetAsync(sel->ptr, 0, sel->bytes, s));
    push(nullptr, nullptr, sel->ptr, 4);   // (no source: the row number itself)
  }
  flush();
  if (!marked) time_mark(ctx, 2);
  time_mark(ctx, 3);
}

}  // namespace qhip

namespace {
// the parts' tables from the parts' first positions (host copy of PartitionWork::starts): pass 2 + the gathers of the odd columns
void partition_pass2(Ctx* ctx, const qhip_table* in, const int32_t* keep, int n_parts, PartitionWork& w, const uint32_t* starts,
                     qhip_table** out_parts) {
  typedef MovedColumn Moved;
  std::vector<Moved> moved;
  std::vector<size_t> odd;   // columns gathered per part through the selection vector
  std::shared_ptr<DevBuf> sel;
  partition_scatter(ctx, in, keep, n_parts, w, starts[n_parts], moved, odd, sel, false);
  for (int p = 0; p < n_parts; ++p) {
    std::unique_ptr<qhip_table> t(new qhip_table());
    t->ctx = ctx;
    t->names = in->names;
    t->nullable = in->nullable;
    const uint64_t at = starts[p], m = (uint64_t)starts[p + 1] - at;
    t->cols.resize(in->cols.size());
    for (size_t c = 0; c < in->cols.size(); ++c) {   // dropped columns: NULL-typed placeholders (qhip_table_keep_columns)
      DevColumn& oc = t->cols[c];
      if ((keep && !keep[c]) || in->cols[c].type.id == QHIP_NULL) {
        oc.type = DType{QHIP_NULL, 0, 0}; oc.length = (int64_t)m; oc.null_count = (int64_t)m;
        t->nullable[c] = true;
      }
    }
    for (const Moved& mv : moved) {
      const DevColumn& src = in->cols[mv.col];
      DevColumn& oc = t->cols[mv.col];
      oc.type = src.type;
      oc.length = (int64_t)m;
      oc.value_maxabs = src.deferred ? src.deferred->src.value_maxabs : src.value_maxabs;
      oc.range = src.deferred ? src.deferred->src.range : src.range; oc.range_inherited = true;
      oc.values = std::make_shared<DevBuf>(mv.out, (size_t)at * (size_t)mv.width, (size_t)m * (size_t)mv.width);
    }
    for (size_t c : odd) t->cols[c] = gather_column(ctx, in->cols[c], sel->as<uint32_t>() + at, m, false);
    t->num_rows = (int64_t)m;
    t->batch_offsets = {0, (int64_t)m};
    out_parts[p] = t.release();
  }
}

void partition_filtered(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n_keys, int pred_root,
                        const int32_t* keep, int n_parts, qhip_table** out_parts, const int64_t* range_bounds = nullptr) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  if (n_parts <= 0 || n_parts > 255 || n_keys <= 0) fail(QHIP_INVALID_ARGUMENT, "qhip_partition_filtered: bad arguments (1 .. 255 parts)");
  if (in->num_rows >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  hipStream_t s = ctx->stream;
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  PartitionWork w;
  partition_pass1(ctx, in, exprs, n_exprs, roots, n_keys, pred_root, n_parts, w, range_bounds);   // (HIP events 0 / 4 around its kernel)
  uint32_t* const back = (uint32_t*)ctx->pinned;   // [status words | parts' first positions + total]
  if ((size_t)(QS_WORDS + n_parts + 1) * 4 > ctx->pinned_bytes) fail(QHIP_UNSUPPORTED, "qhip_partition_filtered: too many parts for the read-back scratch");
  QHIP_HIP_CHECK(hipMemcpyAsync(back, w.dstat, QS_WORDS * 4, hipMemcpyDeviceToHost, s));
  QHIP_HIP_CHECK(hipMemcpyAsync(back + QS_WORDS, w.starts.ptr, ((size_t)n_parts + 1) * 4, hipMemcpyDeviceToHost, s));
  QHIP_HIP_CHECK(sync_stream(s));
  verify_pending_sizes(ctx);
  check_status_words(back);
  std::vector<uint32_t> starts(back + QS_WORDS, back + QS_WORDS + n_parts + 1);
  partition_pass2(ctx, in, keep, n_parts, w, starts.data(), out_parts);   // (HIP events 2 / 3 around its scatter launches)
  time_mark(ctx, 1);
  // statistics (qhip_exec_stats): build_ms = pass 1's kernel, main_kernel_ms = pass 2's kernel(s), total_device_ms = first launch ..
  // last launch incl. the scan and the host wait between the passes; bytes_per_row_read
  // = column bytes pass 2 reads per INPUT row, build_bytes_per_row = column bytes pass 1 reads per input row
  ctx->stats_timing_pending = ctx->timing ? 3 : 0;
  ctx->stats.rows_in = in->num_rows;
  ctx->stats.rows_out = (int64_t)starts[(size_t)n_parts];
  ctx->stats.groups = n_parts;
  ctx->stats.workgroups = (int32_t)((w.n_units + 3) / 4);
  ctx->stats.build_rows = in->num_rows;
  ctx->stats.build_bytes_per_row = w.pass1_bytes_per_row;
  double moved = 0;
  for (size_t c = 0; c < in->cols.size(); ++c)
    if ((!keep || keep[c]) && in->cols[c].type.id != QHIP_NULL) moved += dtype_width(in->cols[c].type);
  ctx->stats.bytes_per_row_read = moved;
  snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "%s",
           w.wg_units ? "qk_part_scatter_wg" : env_int("QHIP_PART_SCATTER_JIT", 1) != 0 ? "qk_part_scatter" : "k_part_scatter");
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
  // qhip_shuffle_tables: the transfers run on the communicator's own stream (behind `ready`, recorded on the context's stream
  // when an input's runs are written; the context's stream waits for `done` before the results are used)
  hipStream_t xstream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
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
// ---- qhip_shuffle_tables (include/qhip.h): the exchange step of a distributed join in one call, one host wait
void comm_shuffle(Ctx* ctx, qhip_comm* c, const qhip_shuffle_input* ins, int n_in, qhip_table** outs) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int W = c->world, me = c->rank;
  if (W > 255) fail(QHIP_UNSUPPORTED, "qhip_shuffle_tables: more than 255 ranks");
  const bool self_rccl = c->comm != nullptr && env_int("QHIP_COMM_SELF_RCCL", 0) != 0;
  // ---- which columns travel, and may they? Decided by TYPES (identical on every rank) before anything collective happens.
  struct Side {
    const qhip_table* in; std::vector<int32_t> keep; int np; PartitionWork w;
    std::vector<size_t> cols;          // the kept, non-NULL-typed columns
    size_t meta_at = 0;                // first metadata word of the side: rows per part [np], then (known, min, max) per kept column
    bool has_nulls = false;
  };
  std::vector<Side> sides((size_t)n_in);
  size_t M = 1;   // word 0: flags (bit 0: a join of deferred size below must run again; bit 1: a kept column holds NULLs here)
  for (int k = 0; k < n_in; ++k) {
    Side& sd = sides[(size_t)k];
    const qhip_shuffle_input& I = ins[k];
    if (!I.table || !I.key_roots || I.n_keys <= 0) fail(QHIP_INVALID_ARGUMENT, "qhip_shuffle_tables: an input without table / keys");
    sd.in = I.table;
    sd.np = (I.all_gather & 1) ? 1 : W;
    sd.keep.assign(sd.in->cols.size(), 1);
    if (I.keep_columns) for (size_t col = 0; col < sd.keep.size(); ++col) sd.keep[col] = I.keep_columns[col] ? 1 : 0;
    if (sd.in->num_rows >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
    for (size_t col = 0; col < sd.in->cols.size(); ++col) {
      if (!sd.keep[col] || sd.in->cols[col].type.id == QHIP_NULL) continue;
      if (dtype_width(sd.in->cols[col].type) <= 0)
        fail(QHIP_UNSUPPORTED, "qhip_shuffle_tables: only fixed-width columns travel this way (" + dtype_name(sd.in->cols[col].type) + ")");
      sd.cols.push_back(col);
    }
    sd.meta_at = M;
    M += (size_t)sd.np + 3 * sd.cols.size();
  }
  comm_settle_time(c, false);
  // ---- pass 1 of every input; the device-side metadata row of this rank
  DevBuf meta_dev(M * 8), all_dev(M * 8 * (size_t)W);
  int64_t* const stage = (int64_t*)ctx->pinned;   // [my host-known words | everybody's rows]
  if ((M * (size_t)(W + 1)) * 8 + 64 > ctx->pinned_bytes) fail(QHIP_UNSUPPORTED, "qhip_shuffle_tables: metadata of this many columns x ranks exceeds the read-back scratch");
  memset(stage, 0, M * 8);
  bool deferred_in = false;
  for (int k = 0; k < n_in; ++k) {
    Side& sd = sides[(size_t)k];
    const qhip_shuffle_input& I = ins[k];
    if (sd.in->rows_dev) deferred_in = true;
    partition_pass1(ctx, sd.in, I.exprs, I.n_exprs, I.key_roots, I.n_keys, I.predicate_root, sd.np, sd.w, sd.np > 1 ? I.range_bounds : nullptr);
    // (NULLs in a kept column: known once the column is resolved — a deferred gather through a nullable index vector is made now)
    for (size_t j = 0; j < sd.cols.size(); ++j) {
      const DevColumn& c0 = sd.in->cols[sd.cols[j]];
      const DevColumn& rc = indirect_eligible(c0) ? c0.deferred->src : resolved(ctx, c0);
      if (rc.null_count > 0) sd.has_nulls = true;
      const DevColumn& stat = c0.deferred && !c0.deferred->done ? c0.deferred->src : rc;
      // the value range of an integer column travels with it (a join behind the exchange addresses its table by the key when
      // the range is small: without it, it would reduce the received column and wait at EVERY execution). Found once per base
      // column (one reduction + one wait, cached on the column and shared by whatever is gathered from it).
      bool is_key = false;
      for (int kk = 0; kk < I.n_keys; ++kk)
        if (I.key_roots[kk] >= 0 && I.key_roots[kk] < I.n_exprs && I.exprs[I.key_roots[kk]].kind == QHIP_EXPR_COLUMN && I.exprs[I.key_roots[kk]].column == (int)sd.cols[j]) is_key = true;
      if ((I.all_gather & 2) && is_key && !(stat.range && stat.range->known) && sd.in->num_rows > 0) {
        const int id = stat.type.id;
        if (id == QHIP_INT64 || id == QHIP_INT32 || id == QHIP_UINT8 || id == QHIP_DATE32 || id == QHIP_DATE64 || (id >= QHIP_TIME32_S && id <= QHIP_TIME64_NS)) {
          int64_t mn = 0, mx = 0;
          (void)key_range_of(ctx, c0, mn, mx);
        }
      }
      const bool known = stat.range && stat.range->known;
      stage[sd.meta_at + (size_t)sd.np + 3 * j] = known ? 1 : 0;
      stage[sd.meta_at + (size_t)sd.np + 3 * j + 1] = known ? stat.range->min : 0;
      stage[sd.meta_at + (size_t)sd.np + 3 * j + 2] = known ? stat.range->max : 0;
    }
    if (sd.has_nulls) stage[0] |= 2;
  }
  QHIP_HIP_CHECK(hipMemcpyAsync(meta_dev.ptr, stage, M * 8, hipMemcpyHostToDevice, s));
  for (int k = 0; k < n_in; ++k) launch_shuffle_meta(sides[(size_t)k].w.starts.as<uint32_t>(), (uint32_t)sides[(size_t)k].np, meta_dev.as<int64_t>() + sides[(size_t)k].meta_at, s);
  if (!ctx->pending_sizes.empty()) {   // joins of deferred size in flight: their verdict becomes part of the metadata (see below)
    PendingSlots ps;
    memset(&ps, 0, sizeof ps);
    for (const Ctx::PendingSize& p : ctx->pending_sizes) { if (ps.n >= 64) break; ps.slot[ps.n] = p.slot; ps.cap[ps.n] = p.capacity; ++ps.n; }
    launch_pending_flags(ps, meta_dev.as<int64_t>(), s);   // (ORs bit 0 into word 0 behind the copy above)
  }
  int64_t* const all = stage + M;
  if (W == 1 && !self_rccl) QHIP_HIP_CHECK(hipMemcpyAsync(all, meta_dev.ptr, M * 8, hipMemcpyDeviceToHost, s));
  else {
    rccl_check(rccl().AllGather(meta_dev.ptr, all_dev.ptr, M, QH_NCCL_INT64, c->comm, s), "ncclAllGather (shuffle metadata)");
    QHIP_HIP_CHECK(hipMemcpyAsync(all, all_dev.ptr, M * 8 * (size_t)W, hipMemcpyDeviceToHost, s));
  }
  // pass 1's status words ride along (one block per input)
  uint32_t* const st_back = (uint32_t*)(all + M * (size_t)W);
  if ((M * (size_t)(W + 1)) * 8 + (size_t)n_in * QS_WORDS * 4 > ctx->pinned_bytes) fail(QHIP_UNSUPPORTED, "qhip_shuffle_tables: read-back scratch too small");
  for (int k = 0; k < n_in; ++k) QHIP_HIP_CHECK(hipMemcpyAsync(st_back + (size_t)k * QS_WORDS, sides[(size_t)k].w.dstat, QS_WORDS * 4, hipMemcpyDeviceToHost, s));
  QHIP_HIP_CHECK(sync_stream(s));   // THE host wait of the exchange
  ++c->stats.host_waits;
  // ---- agreed verdicts first: every rank sees every rank's flags
  bool any_retry = false, any_nulls = false;
  for (int r = 0; r < W; ++r) { any_retry |= (all[(size_t)r * M] & 1) != 0; any_nulls |= (all[(size_t)r * M] & 2) != 0; }
  {
    bool local_retry = false;
    try { verify_pending_sizes(ctx); } catch (const Error& e) { if (e.code != QHIP_RETRY) throw; local_retry = true; }
    if (local_retry && !any_retry) fail(QHIP_HIP_ERROR, "qhip_shuffle_tables: the device-side verdict on a join of deferred size differs from the host's (internal error)");
    if (any_retry) fail(QHIP_RETRY, "a hash join of deferred size below the exchange has to run again on some rank: every rank re-executes its input");
  }
  (void)deferred_in;
  for (int k = 0; k < n_in; ++k) check_status_words(st_back + (size_t)k * QS_WORDS);
  if (any_nulls) fail(QHIP_UNSUPPORTED, "qhip_shuffle_tables: a kept column holds NULLs on some rank (take qhip_partition_filtered + qhip_exchange_tables)");
  // ---- pass 2 of every input, and behind it (on the communicator's stream) its transfers: runs -> final columns
  QHIP_HIP_CHECK(hipEventRecord(c->ev[0], s));
  std::vector<std::shared_ptr<DevBuf>> keep_alive;   // send runs: read by the transfer stream until `done`
  for (int k = 0; k < n_in; ++k) {
    Side& sd = sides[(size_t)k];
    const int64_t* mine = all + (size_t)me * M + sd.meta_at;
    std::vector<uint32_t> starts((size_t)sd.np + 1, 0);
    for (int p = 0; p < sd.np; ++p) starts[(size_t)p + 1] = starts[(size_t)p] + (uint32_t)mine[p];
    std::vector<MovedColumn> moved;
    std::vector<size_t> odd;
    std::shared_ptr<DevBuf> sel;
    partition_scatter(ctx, sd.in, sd.keep.data(), sd.np, sd.w, starts[(size_t)sd.np], moved, odd, sel, false);
    if (!odd.empty()) fail(QHIP_HIP_ERROR, "qhip_shuffle_tables: a column needs a gather of its own (internal error)");
    // rows every rank sends here, in rank order
    std::vector<int64_t> from((size_t)W), at((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) {
      from[(size_t)r] = all[(size_t)r * M + sd.meta_at + (size_t)((ins[k].all_gather & 1) ? 0 : me)];
      at[(size_t)r + 1] = at[(size_t)r] + from[(size_t)r];
    }
    const int64_t N = at[(size_t)W];
    if (N >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
    std::unique_ptr<qhip_table> t(new qhip_table());
    t->ctx = ctx;
    t->names = sd.in->names;
    t->nullable = sd.in->nullable;
    t->num_rows = N;
    t->batch_offsets.push_back(0);
    for (int r = 0; r < W; ++r) t->batch_offsets.push_back(at[(size_t)r + 1]);   // one batch per source rank
    t->cols.resize(sd.in->cols.size());
    for (size_t col = 0; col < sd.in->cols.size(); ++col) {
      DevColumn& oc = t->cols[col];
      oc.type = DType{QHIP_NULL, 0, 0}; oc.length = N; oc.null_count = N;
      if (!sd.keep[col] || sd.in->cols[col].type.id == QHIP_NULL) t->nullable[col] = true;
    }
    QHIP_HIP_CHECK(hipEventRecord(c->ready, s));
    QHIP_HIP_CHECK(hipStreamWaitEvent(c->xstream, c->ready, 0));
    const bool grouped = W > 1 || self_rccl;
    if (grouped) rccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (size_t j = 0; j < moved.size(); ++j) {
      const MovedColumn& mv = moved[j];
      DevColumn& oc = t->cols[mv.col];
      const DevColumn& src = sd.in->cols[mv.col];
      oc.type = src.type; oc.null_count = 0; oc.length = N;
      oc.value_maxabs = src.deferred ? src.deferred->src.value_maxabs : src.value_maxabs;
      oc.values = std::make_shared<DevBuf>((size_t)N * (size_t)mv.width);
      keep_alive.push_back(mv.out);
      for (int r = 0; r < W; ++r) {
        const int p = (ins[k].all_gather & 1) ? 0 : r;   // the run that goes to rank r
        const size_t out_bytes = (size_t)(starts[(size_t)p + 1] - starts[(size_t)p]) * (size_t)mv.width, in_bytes = (size_t)from[(size_t)r] * (size_t)mv.width;
        const uint8_t* sp = mv.out->as<uint8_t>() + (size_t)starts[(size_t)p] * (size_t)mv.width;
        uint8_t* dp = oc.values->as<uint8_t>() + (size_t)at[(size_t)r] * (size_t)mv.width;
        if (r == me && !self_rccl) { if (in_bytes) QHIP_HIP_CHECK(hipMemcpyAsync(dp, sp, in_bytes, hipMemcpyDeviceToDevice, c->xstream)); continue; }
        if (out_bytes) { rccl_check(rccl().Send(sp, out_bytes, QH_NCCL_UINT8, r, c->comm, c->xstream), "ncclSend"); c->stats.bytes_sent += out_bytes; }
        if (in_bytes) { rccl_check(rccl().Recv(dp, in_bytes, QH_NCCL_UINT8, r, c->comm, c->xstream), "ncclRecv"); c->stats.bytes_received += in_bytes; }
      }
      c->stats.bytes_packed += (uint64_t)starts[(size_t)sd.np] * (uint64_t)mv.width;
      // the union of the senders' value ranges (a rank without rows says nothing)
      size_t jm = 0;
      for (; jm < sd.cols.size(); ++jm) if (sd.cols[jm] == mv.col) break;
      bool all_known = true, any = false;
      int64_t lo = 0, hi = 0;
      for (int r = 0; r < W && jm < sd.cols.size(); ++r) {
        if (from[(size_t)r] == 0) continue;
        const int64_t* m = all + (size_t)r * M + sd.meta_at + (size_t)sd.np + 3 * jm;
        if (!m[0]) { all_known = false; break; }
        lo = any ? std::min(lo, m[1]) : m[1];
        hi = any ? std::max(hi, m[2]) : m[2];
        any = true;
      }
      if (all_known && any) { oc.range = std::make_shared<ColRange>(); oc.range->known = true; oc.range->min = lo; oc.range->max = hi; oc.range_inherited = false; }
    }
    if (grouped) rccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    ++c->stats.exchanges;
    outs[k] = t.release();
  }
  QHIP_HIP_CHECK(hipEventRecord(c->done, c->xstream));
  QHIP_HIP_CHECK(hipStreamWaitEvent(s, c->done, 0));   // everything the caller does next is ordered behind the transfers
  QHIP_HIP_CHECK(hipEventRecord(c->ev[1], s));
  c->timed = true;
  // (keep_alive goes back to the pool here: whoever gets those buffers next runs on the context's stream, behind `done`)
}
}  // namespace

extern "C" {

int qhip_shuffle_tables(qhip_ctx* ctx, qhip_comm* comm, const qhip_shuffle_input* inputs, int32_t n_inputs, qhip_table** outs) {
  if (!ctx || !comm || !inputs || !outs || n_inputs <= 0 || n_inputs > 8 || comm->ctx != ctx) return QHIP_INVALID_ARGUMENT;
  for (int k = 0; k < n_inputs; ++k) outs[k] = nullptr;
  int rc = guarded(ctx, [&] { comm_shuffle(ctx, comm, inputs, n_inputs, outs); });
  if (rc != QHIP_OK) {
    if (rc == QHIP_RETRY) ctx->pending_sizes.clear();
    for (int k = 0; k < n_inputs; ++k) { if (outs[k]) { delete outs[k]; outs[k] = nullptr; } }
  }
  return rc;
}

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
    QHIP_HIP_CHECK(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    QHIP_HIP_CHECK(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    QHIP_HIP_CHECK(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
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
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
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
  int rc = guarded(ctx, [&] {
    settle_rows(input);
    if (n_parts <= 255 && env_int("QHIP_PARTITION_FUSED", 1) != 0) partition_filtered(ctx, input, exprs, n_exprs, key_roots, n_keys, -1, nullptr, n_parts, out_parts);
    else partition_by_key(ctx, input, exprs, n_exprs, key_roots, n_keys, n_parts, out_parts);
  });
  if (rc != QHIP_OK)
    for (int p = 0; p < n_parts; ++p) { if (out_parts[p]) { delete out_parts[p]; out_parts[p] = nullptr; } }
  return rc;
}

int qhip_table_column_range(qhip_ctx* ctx, const qhip_table* t, int64_t col, int64_t* out_min, int64_t* out_max) {
  if (!ctx || !t || !out_min || !out_max || col < 0 || col >= (int64_t)t->cols.size()) return QHIP_INVALID_ARGUMENT;
  return guarded(ctx, [&] {
    QHIP_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t mn = 0, mx = 0;
    if (!key_range_of(ctx, t->cols[(size_t)col], mn, mx)) fail(QHIP_UNSUPPORTED, "qhip_table_column_range: not an integer-like column with values (" + dtype_name(t->cols[(size_t)col].type) + ")");
    *out_min = mn; *out_max = mx;
  });
}

int qhip_partition_filtered(qhip_ctx* ctx, const qhip_table* input, const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots,
                            int32_t n_keys, int32_t predicate_root, const int32_t* keep_columns, int32_t n_parts, qhip_table** out_parts) {
  return qhip_partition_filtered_by_range(ctx, input, exprs, n_exprs, key_roots, n_keys, predicate_root, keep_columns, nullptr, n_parts, out_parts);
}

int qhip_partition_filtered_by_range(qhip_ctx* ctx, const qhip_table* input, const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots,
                                     int32_t n_keys, int32_t predicate_root, const int32_t* keep_columns, const int64_t* upper_bounds,
                                     int32_t n_parts, qhip_table** out_parts) {
  if (!ctx || !input || !out_parts || !key_roots) return QHIP_INVALID_ARGUMENT;
  for (int p = 0; p < n_parts; ++p) out_parts[p] = nullptr;
  int rc = guarded(ctx, [&] {
    if (upper_bounds && (n_parts > 255 || env_int("QHIP_PARTITION_FUSED", 1) == 0))
      fail(QHIP_UNSUPPORTED, "partitioning by key range needs the fused path (<= 255 parts)");
    if (n_parts > 255 || env_int("QHIP_PARTITION_FUSED", 1) == 0) {
      // more parts than a byte names (or the switch): the generic sort-based path, which takes no filter and moves every column
      if (predicate_root >= 0) fail(QHIP_UNSUPPORTED, "qhip_partition_filtered: a fused scan filter needs the fused path (<= 255 parts)");
      settle_rows(input);
      if (keep_columns) {
        std::unique_ptr<qhip_table> kept(table_keep_columns(ctx, input, keep_columns));
        partition_by_key(ctx, kept.get(), exprs, n_exprs, key_roots, n_keys, n_parts, out_parts);
      } else partition_by_key(ctx, input, exprs, n_exprs, key_roots, n_keys, n_parts, out_parts);
      return;
    }
    settle_rows(input);
    partition_filtered(ctx, input, exprs, n_exprs, key_roots, n_keys, predicate_root, keep_columns, n_parts, out_parts, upper_bounds);
  });
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
