// sort.cpp — qhip_sort_execute (Sort::execute, physical/plan/sort.rs:48-82) and qhip_limit_execute (Limit::execute,
// physical/plan/limit.rs:27-58): the operators directly downstream of the aggregate in Q1 / Q3.
//
//   sort    key expressions -> order-preserving unsigned images (JIT, qk_sort_keys) -> lexsort_to_indices as stable LSD
//           radix passes (rocPRIM) over (image word, row) pairs, least significant key word first; the reference's implicit
//           last key — the row number in the concatenated input, sort.rs:62-73 — is the initial order. NULL placement
//           (SortOptions.nulls_first) is one extra 1-bit pass per nullable key, `descending` flips the image bits (ties
//           stay in input order either way, as with arrow's lexsort). limit = keep the first n indices (top-N pushdown
//           of the planner, planner/mod.rs:69-75). Output: ONE batch, every column gathered by the index vector
//           (deferred until read).
//   limit   row window [skip, skip + fetch) over the batch list, batch by batch like the reference — including its
//           quirk of emitting one empty batch when the window ends exactly at a batch boundary and more batches follow.
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

qhip_table* sort_table(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, const int32_t* roots, const int32_t* descending,
                       const int32_t* nulls_first, int n_keys, int64_t limit) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  if (n_keys < 0 || (n_keys > 0 && (!roots || !descending || !nulls_first))) fail(QHIP_INVALID_ARGUMENT, "qhip_sort_execute: bad arguments");
  for (int k = 0; k < n_keys; ++k)
    if (roots[k] < 0 || roots[k] >= n_exprs) fail(QHIP_INVALID_ARGUMENT, "sort key index out of range");
  if (in->num_rows >= (int64_t)kNullIdx) fail(QHIP_UNSUPPORTED, "tables of 2^32 - 1 rows or more are not supported");
  hipStream_t s = ctx->stream;
  const uint64_t N = (uint64_t)in->num_rows;
  time_mark(ctx, 0);

  resolve_referenced(ctx, in, exprs, n_exprs);
  std::vector<InputCol> icols = input_cols_of(in);
  ExprSet es;
  es.build(exprs, n_exprs, icols);
  SortKeysPlan plan;
  plan_sort_keys(es, icols, roots, n_keys, plan);

  auto idx = std::make_shared<DevBuf>((N + 1) * 4);
  launch_iota_u32(idx->as<uint32_t>(), N, s);
  if (N > 1 && n_keys > 0) {
    const uint64_t nwords = (N + 63) / 64;
    DevBuf img((size_t)std::max(1, plan.NW) * N * 8), keyvalid((size_t)n_keys * nwords * 8 + 8), diff_dev((size_t)std::max(1, plan.NW) * 8);
    std::vector<uint64_t> diff((size_t)std::max(1, plan.NW), ~0ULL);   // bits in which the rows differ, per image word
    {
      std::shared_ptr<Module> mod = get_module(ctx, plan.source, plan.kernel_name);
      HKArgs ka;
      DevBuf strlit;
      fill_kargs(ctx, in, plan.bind, ka, strlit);
      QHIP_HIP_CHECK(hipMemsetAsync(ctx->status.ptr, 0, QS_WORDS * 4, s));
      QHIP_HIP_CHECK(hipMemsetAsync(diff_dev.ptr, 0, diff_dev.bytes, s));
      void* ip = img.ptr;
      void* vp = keyvalid.ptr;
      void* dp = diff_dev.ptr;
      void* sp = ctx->status.ptr;
      void* args[] = {&ka, &ip, &vp, &dp, &sp};
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nwords + 3) / 4, (uint64_t)ctx->num_cus * 8));
      QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
      uint32_t st[QS_WORDS];
      if (plan.NW > 0) QHIP_HIP_CHECK(hipMemcpyAsync(diff.data(), diff_dev.ptr, (size_t)plan.NW * 8, hipMemcpyDeviceToHost, s));
      copy_sync(s, st, ctx->status.ptr, sizeof st, hipMemcpyDeviceToHost);
      check_status_words(st);
      if (env_int("QHIP_SORT_FULL_PASSES", 0)) std::fill(diff.begin(), diff.end(), ~0ULL);
    }
    auto idx2 = std::make_shared<DevBuf>((N + 1) * 4);
    DevBuf key_a((N + 1) * 8), key_b((N + 1) * 8);
    auto pass = [&](int bits) {   // key_a holds the images in the current order
      stable_sort_pairs_u64(key_a.as<uint64_t>(), key_b.as<uint64_t>(), idx->as<uint32_t>(), idx2->as<uint32_t>(), N, bits, s);
      std::swap(idx, idx2);
    };
    for (int k = n_keys - 1; k >= 0; --k) {
      const SortKeyDesc& kd = plan.keys[(size_t)k];
      const bool desc = descending[k] != 0;
      if (kd.type.id == QHIP_UTF8) {
        const DevColumn& col = resolved(ctx, in->cols[(size_t)kd.column]);
        if (col.utf8_max_len < 0) {
          DevBuf m(4);
          QHIP_HIP_CHECK(hipMemsetAsync(m.ptr, 0, 4, s));
          launch_utf8_max_len(col.values->as<int32_t>(), (uint64_t)col.length, m.as<uint32_t>(), s);
          uint32_t v = 0;
          copy_sync(s, &v, m.ptr, 4, hipMemcpyDeviceToHost);
          col.utf8_max_len = (int32_t)v;
        }
        const int nchunks = (col.utf8_max_len + 7) / 8;
        const uint8_t* validity = col.validity ? col.validity->as<uint8_t>() : nullptr;
        const uint8_t* data = col.data ? col.data->as<uint8_t>() : nullptr;
        if (nchunks > 0) {
          // least significant first: the length, then the chunks from the last to the first
          launch_sort_utf8_chunk(col.values->as<int32_t>(), data, validity, idx->as<uint32_t>(), N, -1, desc ? 0xFFFFFFFFULL : 0, key_a.as<uint64_t>(), s);
          pass(32);
          for (int c = nchunks - 1; c >= 0; --c) {
            launch_sort_utf8_chunk(col.values->as<int32_t>(), data, validity, idx->as<uint32_t>(), N, c, desc ? ~0ULL : 0, key_a.as<uint64_t>(), s);
            pass(64);
          }
        }
      } else {
        for (int w = 0; w < kd.words; ++w) {
          // only the bits in which some rows differ can change the order (NULL rows carry image 0: the XOR with row 0's
          // image covers them); a word all rows agree on needs no pass at all
          const uint64_t dw = diff[(size_t)(kd.word_off + w)];
          if (dw == 0) continue;
          const int bits = std::min(w == kd.words - 1 ? kd.top_bits : 64, 64 - __builtin_clzll(dw));
          const uint64_t mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
          launch_sort_gather_img(img.as<uint64_t>() + (size_t)(kd.word_off + w) * N, idx->as<uint32_t>(), N, desc ? mask : 0, key_a.as<uint64_t>(), s);
          pass(bits);
        }
      }
      if (kd.nullable) {
        launch_sort_gather_valid(keyvalid.as<uint64_t>() + (size_t)k * nwords, idx->as<uint32_t>(), N, nulls_first[k] != 0, key_a.as<uint64_t>(), s);
        pass(1);
      }
    }
  }
  const uint64_t m = limit >= 0 ? std::min<uint64_t>((uint64_t)limit, N) : N;

  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->names = in->names;
  out->nullable = in->nullable;
  defer_gather(ctx, in->cols, idx, m, false, out->cols);
  out->num_rows = (int64_t)m;
  out->batch_offsets = {0, (int64_t)m};   // sort.rs:81: always exactly one batch
  time_mark(ctx, 1);
  ctx->stats_timing_pending = ctx->timing ? 1 : 0;
  ctx->stats.rows_in = (int64_t)N;
  ctx->stats.rows_out = (int64_t)m;
  snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "qk_sort_keys+radix passes");
  return out.release();
}

qhip_table* limit_table(Ctx* ctx, const qhip_table* in, int64_t skip, int64_t fetch) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  if (skip < 0) fail(QHIP_INVALID_ARGUMENT, "qhip_limit_execute: negative skip");
  // limit.rs:31-55, batch by batch — including its OFFSET quirk: `skip` is decremented by whole skipped batches only
  // (limit.rs:39-42) and never cleared after the batch it was applied to (limit.rs:44), so every later batch loses its
  // first `skip` rows too (or is dropped whole, shrinking `skip`). The output is therefore a list of row segments.
  const uint64_t max_fetch = fetch < 0 ? ~0ULL : (uint64_t)fetch;
  uint64_t fetched = 0, to_skip = (uint64_t)skip;
  struct Seg { int64_t first; uint64_t rows; };
  std::vector<Seg> segs;
  std::vector<int64_t> offs = {0};
  for (int64_t b = 0; b < in->num_batches(); ++b) {
    const uint64_t rows = (uint64_t)(in->offsets()[(size_t)b + 1] - in->offsets()[(size_t)b]);
    if (rows <= to_skip) { to_skip -= rows; continue; }
    const uint64_t new_rows = rows - to_skip;
    const int64_t first = in->offsets()[(size_t)b] + (int64_t)to_skip;
    const uint64_t remaining = max_fetch - fetched;
    const uint64_t take = new_rows <= remaining ? new_rows : remaining;
    if (take) {
      if (!segs.empty() && segs.back().first + (int64_t)segs.back().rows == first) segs.back().rows += take;
      else segs.push_back({first, take});
    }
    fetched += take;
    offs.push_back((int64_t)fetched);   // possibly an empty batch (remaining == 0), like the reference's slice(0, 0)
    if (new_rows > remaining) break;
  }
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->names = in->names;
  out->nullable = in->nullable;
  out->num_rows = (int64_t)fetched;
  out->batch_offsets = offs;
  if ((int64_t)fetched == in->num_rows) {
    out->cols = in->cols;   // the whole table: column buffers are shared
  } else {
    auto idx = std::make_shared<DevBuf>((fetched + 1) * 4);
    uint64_t at = 0;
    for (const Seg& sg : segs) {
      launch_iota_u32(idx->as<uint32_t>() + at, sg.rows, ctx->stream, (uint32_t)sg.first);
      at += sg.rows;
    }
    defer_gather(ctx, in->cols, idx, fetched, false, out->cols);
  }
  ctx->stats.rows_in = in->num_rows;
  ctx->stats.rows_out = (int64_t)fetched;
  return out.release();
}

}  // namespace

extern "C" int qhip_sort_execute(qhip_ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots,
                                 const int32_t* descending, const int32_t* nulls_first, int32_t n_keys, int64_t limit, qhip_table** out) {
  if (!ctx || !in || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { settle_rows(in); *out = sort_table(ctx, in, exprs, n_exprs, key_roots, descending, nulls_first, n_keys, limit); });
}

extern "C" int qhip_limit_execute(qhip_ctx* ctx, const qhip_table* in, int64_t skip, int64_t fetch, qhip_table** out) {
  if (!ctx || !in || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { settle_rows(in); *out = limit_table(ctx, in, skip, fetch); });
}
