// filter.cpp — qhip_filter_execute: Filter::execute (physical/plan/filter.rs:28-44) and MemoryTable::scan with a
// pushed-down filter / projection (datasource/memory.rs:69-98).
//   predicate -> keep mask (JIT kernel, wavefront ballot)  ->  exclusive scan of the per-wave popcounts
//   -> selection vector (rank inside the ballot mask)      ->  every column gathered by the selection vector.
// One output batch per input batch, possibly empty, row order preserved.
#include <hip/hip_runtime_api.h>

#include "common.hpp"
#include "kernels.hpp"
#include "relops.hpp"

using namespace qhip;

static qhip_table* filter_execute(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, int root, const int32_t* projection,
                                  int n_projection) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  std::vector<int> cols;
  if (projection && n_projection >= 0) {
    for (int k = 0; k < n_projection; ++k) {
      if (projection[k] < 0 || projection[k] >= (int)in->cols.size()) fail(QHIP_INVALID_ARGUMENT, "projection index out of range");
      cols.push_back(projection[k]);
    }
  } else {
    for (size_t c = 0; c < in->cols.size(); ++c) cols.push_back((int)c);
  }
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  for (int c : cols) { out->names.push_back(in->names[(size_t)c]); out->nullable.push_back(in->nullable[(size_t)c]); }
  if (root < 0) {
    // projection only: the column buffers are shared, nothing is copied (RecordBatch::project, memory.rs:79-88)
    for (int c : cols) out->cols.push_back(in->cols[(size_t)c]);
    out->num_rows = in->num_rows;
    out->batch_offsets = in->offsets();
    return out.release();
  }
  if (root >= n_exprs) fail(QHIP_INVALID_ARGUMENT, "predicate index out of range");
  if (in->num_rows >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  resolve_referenced(ctx, in, exprs, n_exprs);
  std::vector<InputCol> icols = input_cols_of(in);
  ExprSet es;
  es.build(exprs, n_exprs, icols);
  DevBuf mask, wave, sel;
  time_mark(ctx, 0);
  run_pred_mask(ctx, in, es, icols, root, mask, wave);
  // rows kept (the one read-back that sizes the output), then every column compacted:
  //  * a predicate that keeps a good part of the rows (>= 1/8): MASK-DRIVEN — each column is read in row order and its kept
  //    values written in runs (k_compact_fixed); no selection vector exists unless a column's layout needs one (nullable,
  //    Boolean, Utf8 of more than 1 byte);
  //  * a selective predicate: the selection vector (rank inside the ballot word + scanned wave offset) and an index gather
  //    per column, which reads only the kept values.
  const uint32_t m = count_from_mask(ctx, wave, in->num_rows);
  const bool mask_driven = in->num_rows > 0 && (uint64_t)m * 8 >= (uint64_t)in->num_rows && env_int("QHIP_FILTER_NO_COMPACT", 0) == 0;
  bool have_sel = false;
  for (int c : cols) {
    DevColumn oc;
    if (mask_driven && compact_column(ctx, in->cols[(size_t)c], mask, wave, in->num_rows, m, oc)) { out->cols.push_back(std::move(oc)); continue; }
    if (!have_sel) { indices_from_mask(ctx, mask, wave, in->num_rows, m, sel); have_sel = true; }
    {
      // (a Utf8 column's longest value decides whether its gather needs a length scan: found once per column, cached)
      const DevColumn& src = resolved(ctx, in->cols[(size_t)c]);
      if (src.type.id == QHIP_UTF8 && src.utf8_max_len < 0 && src.length > 0) {
        DevBuf mx(4);
        QHIP_HIP_CHECK(hipMemsetAsync(mx.ptr, 0, 4, ctx->stream));
        launch_utf8_max_len(src.values->as<int32_t>(), (uint64_t)src.length, mx.as<uint32_t>(), ctx->stream);
        uint32_t v = 0;
        copy_sync(ctx->stream, &v, mx.ptr, 4, hipMemcpyDeviceToHost);
        src.utf8_max_len = (int32_t)v;
      }
    }
    out->cols.push_back(gather_column(ctx, in->cols[(size_t)c], sel.as<uint32_t>(), m, false));
  }
  time_mark(ctx, 1);
  // output batch boundaries = kept rows before each input batch start
  const size_t nb1 = in->offsets().size();
  out->num_rows = m;
  if (in->num_rows == 0) {
    out->batch_offsets.assign(nb1, 0);
  } else {
    // kept rows before every input batch start, computed on the device and left there until somebody asks
    // (qhip_table::offsets()): an aggregate or a join's build side above this filter never does
    auto pend = std::make_shared<PendingOffsets>();
    pend->pos = std::make_shared<DevBuf>(nb1 * 4);
    pend->n = nb1;
    launch_mask_prefix_at(mask.as<uint64_t>(), wave.as<uint32_t>(), in->device_offsets(), (uint32_t)nb1, (uint64_t)in->num_rows, m,
                          pend->pos->as<uint32_t>(), ctx->stream);
    out->batch_offsets.clear();
    out->pending_offsets = pend;
    if (env_int("QHIP_EAGER_OFFSETS", 0) != 0) (void)out->offsets();
  }
  ctx->stats_timing_pending = ctx->timing ? 1 : 0;   // ev0..ev1, read by qhip_ctx_last_stats
  ctx->stats.rows_in = in->num_rows;
  ctx->stats.rows_out = m;
  snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "qk_pred_mask+gather");
  return out.release();
}

extern "C" int qhip_filter_execute(qhip_ctx* ctx, const qhip_table* input, const qhip_expr* exprs, int32_t n_exprs, int32_t predicate_root,
                                   const int32_t* projection, int32_t n_projection, qhip_table** out) {
  if (!ctx || !input || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { settle_rows(input); *out = filter_execute(ctx, input, exprs, n_exprs, predicate_root, projection, n_projection); });
}
