// projection.cpp — qhip_projection_execute: Projection::execute (physical/plan/projection.rs:27-46), SURVEY §8f rank 2.
// Column expressions share the input column (deferred gathers stay deferred); all other expressions are evaluated by one
// generated kernel (qk_project) in a single pass over the columns they read; a literal is a broadcast of the same kernel.
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

qhip_table* project_table(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n_out,
                          const char* const* out_names) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  if (n_out < 0 || (n_out > 0 && !roots)) fail(QHIP_INVALID_ARGUMENT, "qhip_projection_execute: bad arguments");
  for (int k = 0; k < n_out; ++k)
    if (roots[k] < 0 || roots[k] >= n_exprs) fail(QHIP_INVALID_ARGUMENT, "projection expression index out of range");
  hipStream_t s = ctx->stream;
  const int64_t N = in->num_rows;
  // only the columns that computed expressions read are gathered if deferred; plain Column outputs stay as they are
  std::vector<qhip_expr> computed_only(exprs, exprs + n_exprs);
  std::vector<bool> is_col((size_t)n_out, false);
  std::vector<bool> needed((size_t)n_exprs, false);
  for (int k = 0; k < n_out; ++k) is_col[(size_t)k] = exprs[roots[k]].kind == QHIP_EXPR_COLUMN;
  {
    // mark the nodes reachable from computed roots
    std::vector<int> stack;
    for (int k = 0; k < n_out; ++k) if (!is_col[(size_t)k]) stack.push_back(roots[k]);
    while (!stack.empty()) {
      const int k = stack.back(); stack.pop_back();
      if (k < 0 || k >= n_exprs || needed[(size_t)k]) continue;
      needed[(size_t)k] = true;
      stack.push_back(exprs[k].left); stack.push_back(exprs[k].right);
      if (exprs[k].kind == QHIP_EXPR_IF) stack.push_back(exprs[k].third);
    }
    for (int k = 0; k < n_exprs; ++k)
      if (needed[(size_t)k] && exprs[k].kind == QHIP_EXPR_COLUMN && exprs[k].column >= 0 && exprs[k].column < (int)in->cols.size())
        (void)resolved(ctx, in->cols[(size_t)exprs[k].column]);
  }
  std::vector<InputCol> icols = input_cols_of(in);
  ExprSet es;
  es.build(exprs, n_exprs, icols);

  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->num_rows = N;
  out->batch_offsets = in->offsets();
  out->cols.resize((size_t)n_out);
  std::vector<int32_t> croots;
  std::vector<int> cslot;
  for (int k = 0; k < n_out; ++k) {
    out->names.push_back(out_names && out_names[k] ? out_names[k] : ("col" + std::to_string(k)));
    out->nullable.push_back(true);
    if (is_col[(size_t)k]) out->cols[(size_t)k] = in->cols[(size_t)es.at(roots[k]).column];
    else { croots.push_back(roots[k]); cslot.push_back(k); }
  }
  time_mark(ctx, 0);
  if (!croots.empty()) {
    ProjectionPlan plan;
    plan_projection(es, icols, croots.data(), (int)croots.size(), plan);
    const uint64_t nwords = (uint64_t)(N + 63) / 64;
    HProjOut po;
    memset(&po, 0, sizeof po);
    for (size_t c = 0; c < plan.outs.size(); ++c) {
      DevColumn& col = out->cols[(size_t)cslot[c]];
      col.type = plan.outs[c].type;
      col.length = N;
      const int w = dtype_width(col.type);
      // (a computed Utf8 output: N + 1 int32 words — the kernel's first pass leaves the lengths, the scan makes them offsets)
      col.values = std::make_shared<DevBuf>(col.type.id == QHIP_UTF8 ? ((size_t)N + 1) * 4 : w > 0 ? (size_t)N * w : (size_t)nwords * 8 + 8);
      if (col.type.id == QHIP_UTF8 && N == 0) {   // no rows: one zero offset and an (empty) data buffer, like an uploaded empty column
        QHIP_HIP_CHECK(hipMemsetAsync(col.values->ptr, 0, 4, s));
        col.data = std::make_shared<DevBuf>(1);
        col.data_bytes = 0;
      }
      po.v[c] = col.values->ptr;
      if (plan.outs[c].nullable) {
        col.validity = std::make_shared<DevBuf>((size_t)nwords * 8 + 8);
        po.n[c] = col.validity->as<uint64_t>();
      }
    }
    if (N > 0 && plan.has_utf8) {
      // int32 offsets (Arrow Utf8): the lengths are summed in 32 bits on the device, so the output is bounded on the host first.
      // A row's value is a literal or the value of a Utf8 column AT THAT ROW, so all rows together hold at most rows x (the
      // literals' bytes) + the data bytes of every referenced Utf8 column; only a bound below 2^32 runs, and the exact total is
      // checked below.
      unsigned __int128 bound = (unsigned __int128)plan.bind.strlits.size() * (uint64_t)N;
      for (int ci : plan.bind.cols)
        if (icols[(size_t)ci].type.id == QHIP_UTF8) bound += (uint64_t)std::max<int64_t>(resolved(ctx, in->cols[(size_t)ci]).data_bytes, 0);
      if (bound >= ((unsigned __int128)1 << 32)) fail(QHIP_UNSUPPORTED, "computed Utf8 column that may exceed 2 GiB (needs LargeUtf8 offsets)");
    }
    if (N > 0) {
      std::shared_ptr<Module> mod = get_module(ctx, plan.source, plan.kernel_name);
      HKArgs ka;
      DevBuf strlit;
      fill_kargs(ctx, in, plan.bind, ka, strlit);
      QHIP_HIP_CHECK(hipMemsetAsync(ctx->status.ptr, 0, QS_WORDS * 4, s));
      void* sp = ctx->status.ptr;
      void* args[] = {&ka, &po, &sp};
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nwords + 3) / 4, (uint64_t)ctx->num_cus * 8));
      QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
      // null counts of the nullable outputs (validity is kept only when there are NULLs), then the status words
      std::vector<DevBuf> counters;
      std::vector<uint32_t> set_bits(plan.outs.size(), 0);
      for (size_t c = 0; c < plan.outs.size(); ++c) {
        counters.emplace_back(4);
        if (!plan.outs[c].nullable) continue;
        QHIP_HIP_CHECK(hipMemsetAsync(counters[c].ptr, 0, 4, s));
        launch_count_bits(po.n[c], (uint64_t)N, counters[c].as<uint32_t>(), s);
        QHIP_HIP_CHECK(hipMemcpyAsync(&set_bits[c], counters[c].ptr, 4, hipMemcpyDeviceToHost, s));
      }
      // computed Utf8 outputs: lengths -> offsets (+ the byte total behind the last offset), read back with the status words
      std::vector<uint32_t> utf8_bytes(plan.outs.size(), 0);
      for (size_t c = 0; c < plan.outs.size(); ++c) {
        if (plan.outs[c].type.id != QHIP_UTF8) continue;
        uint32_t* len = out->cols[(size_t)cslot[c]].values->as<uint32_t>();
        exclusive_scan_u32(len, len, (uint64_t)N, len + N, s);
        QHIP_HIP_CHECK(hipMemcpyAsync(&utf8_bytes[c], len + N, 4, hipMemcpyDeviceToHost, s));
      }
      uint32_t st[QS_WORDS];
      copy_sync(s, st, ctx->status.ptr, sizeof st, hipMemcpyDeviceToHost);
      check_status_words(st);
      if (plan.has_utf8) {
        // second pass: the same expressions once more, the bytes copied to their offsets
        for (size_t c = 0; c < plan.outs.size(); ++c) {
          if (plan.outs[c].type.id != QHIP_UTF8) continue;
          DevColumn& col = out->cols[(size_t)cslot[c]];
          if (utf8_bytes[c] > 0x7fffffffu) fail(QHIP_UNSUPPORTED, "computed Utf8 column larger than 2 GiB (needs LargeUtf8 offsets)");
          col.data = std::make_shared<DevBuf>(std::max<size_t>(utf8_bytes[c], 1));
          col.data_bytes = (int64_t)utf8_bytes[c];
          po.d[c] = col.data->as<uint8_t>();
        }
        std::shared_ptr<Module> cmod = get_module(ctx, plan.source, "qk_project_copy");
        QHIP_HIP_CHECK(hipModuleLaunchKernel(cmod->fn, grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
        QHIP_HIP_CHECK(sync_stream(s));   // (`po` and the literals' buffer live on this frame)
      }
      for (size_t c = 0; c < plan.outs.size(); ++c) {
        DevColumn& col = out->cols[(size_t)cslot[c]];
        if (!plan.outs[c].nullable) continue;
        col.null_count = N - (int64_t)set_bits[c];
        if (col.null_count == 0) col.validity.reset();
      }
    } else {
      for (size_t c = 0; c < plan.outs.size(); ++c) out->cols[(size_t)cslot[c]].validity.reset();
    }
  }
  time_mark(ctx, 1);
  ctx->stats_timing_pending = ctx->timing ? 1 : 0;
  ctx->stats.rows_in = N;
  ctx->stats.rows_out = N;
  snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "qk_project");
  return out.release();
}

}  // namespace

extern "C" int qhip_projection_execute(qhip_ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int32_t n_exprs, const int32_t* roots,
                                       int32_t n_out, const char* const* out_names, qhip_table** out) {
  if (!ctx || !in || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { settle_rows(in); *out = project_table(ctx, in, exprs, n_exprs, roots, n_out, out_names); });
}
