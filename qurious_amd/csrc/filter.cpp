// filter.cpp — qhip_filter_execute: Filter::execute (physical/plan/filter.rs:28-44) and MemoryTable::scan with a
// pushed-down filter / projection (datasource/memory.rs:69-98).
//   predicate -> keep mask (JIT kernel, wavefront ballot)  ->  exclusive scan of the per-wave popcounts
//   -> selection vector (rank inside the ballot mask)      ->  every column gathered by the selection vector.
// One output batch per input batch, possibly empty, row order preserved.
#include <hip/hip_runtime_api.h>

#include "common.hpp"
#include "device/qhip_status.h"
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
  // ONE host wait: the predicate's status words and the number of kept rows (the read-back that sizes the output) come back
  // together — the status block is a pre-zeroed one from the context's ring, the scan leaves its total in the block's
  // spare word
  uint32_t* const dstat = zeroed_block(ctx);
  run_pred_mask(ctx, in, es, icols, root, mask, wave, dstat);
  uint32_t m = 0;
  if (in->num_rows > 0) {
    const uint64_t nwords = (uint64_t)(in->num_rows + 63) / 64;
    exclusive_scan_u32(wave.as<uint32_t>(), wave.as<uint32_t>(), nwords, dstat + QS_WORDS, ctx->stream);
    uint32_t* const st = (uint32_t*)ctx->pinned;
    QHIP_HIP_CHECK(hipMemcpyAsync(st, dstat, (QS_WORDS + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    QHIP_HIP_CHECK(sync_stream(ctx->stream));
    check_status_words(st);
    m = st[QS_WORDS];
  }
  // every column compacted:
  //  * a predicate that keeps a good part of the rows (>= 1/8): MASK-DRIVEN — each column is read in row order and its kept
  //    values written in runs (k_compact_fixed); no selection vector exists unless a column's layout needs one (nullable,
  //    Boolean, Utf8 of more than 1 byte);
  //  * a selective predicate: the selection vector (rank inside the ballot word + scanned wave offset), then ALL plain
  //    columns — fixed width, no NULLs; Utf8 whose every value is one byte — gathered by ONE launch (k_gather_multi; the
  //    one-byte strings share one 0, 1, 2, ... offsets buffer), which reads only the kept values; nullable / Boolean /
  //    longer Utf8 columns take their own gathers (validity bits and byte totals need passes of their own).
  const bool mask_driven = in->num_rows > 0 && (uint64_t)m * 8 >= (uint64_t)in->num_rows && env_int("QHIP_FILTER_NO_COMPACT", 0) == 0;
  bool have_sel = false;
  auto need_sel = [&] { if (!have_sel) { indices_from_mask(ctx, mask, wave, in->num_rows, m, sel); have_sel = true; } };
  out->cols.resize(cols.size());
  GatherBatch gb;
  memset(&gb, 0, sizeof gb);
  int n_batched = 0;
  std::shared_ptr<DevBuf> iota_offsets;   // offsets 0 .. m of every gathered one-byte Utf8 column
  auto flush = [&] { if (n_batched) { launch_gather_multi(gb, n_batched, ctx->stream); n_batched = 0; } };
  const bool batch_gathers = env_int("QHIP_FILTER_NO_BATCH", 0) == 0;
  for (size_t k = 0; k < cols.size(); ++k) {
    const int c = cols[k];
    DevColumn oc;
    if (mask_driven && compact_column(ctx, in->cols[(size_t)c], mask, wave, in->num_rows, m, oc)) { out->cols[k] = std::move(oc); continue; }
    need_sel();
    const DevColumn& src = resolved(ctx, in->cols[(size_t)c]);
    // (a Utf8 column's longest value decides whether its gather needs a length scan: found once per column, cached)
    if (src.type.id == QHIP_UTF8 && src.utf8_max_len < 0 && src.length > 0) {
      DevBuf mx(4);
      QHIP_HIP_CHECK(hipMemsetAsync(mx.ptr, 0, 4, ctx->stream));
      launch_utf8_max_len(src.values->as<int32_t>(), (uint64_t)src.length, mx.as<uint32_t>(), ctx->stream);
      uint32_t v = 0;
      copy_sync(ctx->stream, &v, mx.ptr, 4, hipMemcpyDeviceToHost);
      src.utf8_max_len = (int32_t)v;
    }
    const int w = dtype_width(src.type);
    const bool plain_fixed = w > 0 && src.null_count == 0;
    const bool plain_str1 = src.type.id == QHIP_UTF8 && src.null_count == 0 && src.utf8_max_len == 1 && src.data_bytes == src.length;
    if (batch_gathers && m > 0 && (plain_fixed || plain_str1)) {
      oc.type = src.type;
      oc.length = (int64_t)m;
      oc.utf8_max_len = src.utf8_max_len;
      oc.value_maxabs = src.value_maxabs;
      oc.range = src.range; oc.range_inherited = true;
      const void* from;
      void* to;
      if (plain_fixed) {
        oc.values = std::make_shared<DevBuf>((size_t)m * (size_t)w);
        from = src.values->ptr; to = oc.values->ptr;
      } else {
        if (!iota_offsets) {
          iota_offsets = std::make_shared<DevBuf>(((size_t)m + 1) * 4);
          launch_iota_u32(iota_offsets->as<uint32_t>(), (uint64_t)m + 1, ctx->stream, 0);
        }
        oc.values = iota_offsets;
        oc.data = std::make_shared<DevBuf>((size_t)m + 64);
        oc.data_bytes = (int64_t)m;
        from = src.data->ptr; to = oc.data->ptr;
      }
      gb.d[n_batched++] = GatherDesc{from, sel.as<uint32_t>(), to, (uint64_t)m, (uint32_t)(plain_fixed ? w : 1), 0u};
      if (n_batched == kGatherBatch) flush();
      out->cols[k] = std::move(oc);
      continue;
    }
    out->cols[k] = gather_column(ctx, in->cols[(size_t)c], sel.as<uint32_t>(), m, false);
  }
  flush();
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
