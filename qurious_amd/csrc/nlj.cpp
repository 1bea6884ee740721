// nlj.cpp — qhip_nested_loop_join_execute (NestedLoopJoinExec::execute, physical/plan/join/nest_loop_join.rs:79-228) and
// qhip_cross_join_execute (CrossJoin::execute, physical/plan/join/cross_join.rs:121-166): the reference's fallbacks for
// joins without equi-keys. O(|L| x |R|) by definition; here the pair index vectors are generated on
// the device in the reference's order, the optional JoinFilter is evaluated over an intermediate batch of its columns
// exactly like the hash join's residual filter, and every output column is a deferred gather.
#include <hip/hip_runtime_api.h>

#include <algorithm>

#include "common.hpp"
#include "device/qhip_status.h"
#include "kernels.hpp"
#include "relops.hpp"

using namespace qhip;

namespace {

const uint64_t kMaxPairs = 1ULL << 31;

// keep the pairs whose filter evaluates to true (join_filter_indices, nest_loop_join.rs:262-289)
void apply_filter(Ctx* ctx, const qhip_table* L, const qhip_table* R, const qhip_expr* fex, int nfex, int froot, const int32_t* fsides,
                  const int32_t* fcols, int nfcols, std::shared_ptr<DevBuf>& li, std::shared_ptr<DevBuf>& ri, uint64_t& M) {
  if (froot < 0 || M == 0) return;
  hipStream_t s = ctx->stream;
  qhip_table inter;
  inter.ctx = ctx;
  for (int k = 0; k < nfcols; ++k) {
    const qhip_table* src = fsides[k] == 0 ? L : R;
    if (fcols[k] < 0 || fcols[k] >= (int)src->cols.size()) fail(QHIP_INVALID_ARGUMENT, "join filter column index out of range");
    inter.cols.push_back(gather_column(ctx, src->cols[(size_t)fcols[k]], (fsides[k] == 0 ? li : ri)->as<uint32_t>(), M, false));
    inter.names.push_back(src->names[(size_t)fcols[k]]);
    inter.nullable.push_back(true);
  }
  inter.num_rows = (int64_t)M;
  inter.batch_offsets = {0, (int64_t)M};
  std::vector<InputCol> fic = input_cols_of(&inter);
  ExprSet fes;
  fes.build(fex, nfex, fic);
  DevBuf mask, wave, sel;
  run_pred_mask(ctx, &inter, fes, fic, froot, mask, wave);
  const uint32_t m2 = select_from_mask(ctx, mask, wave, (int64_t)M, sel);
  auto l2 = std::make_shared<DevBuf>(((uint64_t)m2 + 1) * 4), r2 = std::make_shared<DevBuf>(((uint64_t)m2 + 1) * 4);
  launch_gather_fixed(li->ptr, sel.as<uint32_t>(), l2->ptr, m2, 4, s);
  launch_gather_fixed(ri->ptr, sel.as<uint32_t>(), r2->ptr, m2, 4, s);
  li = l2; ri = r2;
  M = m2;
}

// rows of `n` whose visited bit equals want_set, ascending
uint64_t select_by_bits(Ctx* ctx, const DevBuf& bits, uint64_t n, int want_set, DevBuf& sel) {
  if (!n) return 0;
  DevBuf tmask(((n + 63) / 64) * 8 + 8), twave((((n + 63) / 64) + 1) * 4);
  launch_mask_from_bits(bits.as<uint32_t>(), n, want_set, tmask.as<uint64_t>(), twave.as<uint32_t>(), ctx->stream);
  return select_from_mask(ctx, tmask, twave, (int64_t)n, sel);
}

qhip_table* nested_loop_join(Ctx* ctx, const qhip_table* L, const qhip_table* R, int join_type, const qhip_expr* fex, int nfex, int froot,
                             const int32_t* fsides, const int32_t* fcols, int nfcols) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  if (join_type < QHIP_JOIN_LEFT || join_type > QHIP_JOIN_LEFT_ANTI) fail(QHIP_INVALID_ARGUMENT, "unknown join type");
  hipStream_t s = ctx->stream;
  const uint64_t NL = (uint64_t)L->num_rows, NR = (uint64_t)R->num_rows;
  const bool semi_anti = join_type == QHIP_JOIN_LEFT_SEMI || join_type == QHIP_JOIN_LEFT_ANTI;
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  auto add_columns = [&](const std::shared_ptr<DevBuf>& li, const std::shared_ptr<DevBuf>& ri, uint64_t m, bool l_null, bool r_null) {
    defer_gather(ctx, L->cols, li, m, l_null, out->cols);
    if (!semi_anti) defer_gather(ctx, R->cols, ri, m, r_null, out->cols);
  };
  for (size_t c = 0; c < L->cols.size(); ++c) { out->names.push_back(L->names[c]); out->nullable.push_back(true); }
  if (!semi_anti) for (size_t c = 0; c < R->cols.size(); ++c) { out->names.push_back(R->names[c]); out->nullable.push_back(true); }
  out->batch_offsets = {0};

  if (NR == 0) {   // nest_loop_join.rs:87-121
    if (join_type == QHIP_JOIN_INNER || join_type == QHIP_JOIN_RIGHT) {
      add_columns(std::make_shared<DevBuf>(4), std::make_shared<DevBuf>(4), 0, false, false);
      out->num_rows = 0;
      return out.release();   // no batches at all
    }
    const uint64_t m = join_type == QHIP_JOIN_LEFT_SEMI ? 0 : NL;
    auto li = std::make_shared<DevBuf>((m + 1) * 4), ri = std::make_shared<DevBuf>((m + 1) * 4);
    launch_iota_u32(li->as<uint32_t>(), m, s);
    launch_fill_u32(ri->as<uint32_t>(), m, kNullIdx, s);
    add_columns(li, ri, m, false, true);
    out->num_rows = (int64_t)m;
    out->batch_offsets.push_back((int64_t)m);
    return out.release();
  }

  // ---- all pairs, right-row major: for every right row the left rows in order (build_join_indices, :232-260)
  if (NL * NR >= kMaxPairs) fail(QHIP_UNSUPPORTED, "nested loop join of more than 2^31 row pairs is not accelerated");
  uint64_t M = NL * NR;
  auto li = std::make_shared<DevBuf>((M + 1) * 4), ri = std::make_shared<DevBuf>((M + 1) * 4);
  launch_pair_indices(li->as<uint32_t>(), ri->as<uint32_t>(), M, (uint32_t)std::max<uint64_t>(NL, 1), 0, 0, 0, s);
  apply_filter(ctx, L, R, fex, nfex, froot, fsides, fcols, nfcols, li, ri, M);

  if (semi_anti) {   // :141-170: one batch of the left rows that have (Semi) / lack (Anti) a surviving pair
    DevBuf visited((((NL + 63) / 64) * 2 + 2) * 4), sel;
    QHIP_HIP_CHECK(hipMemsetAsync(visited.ptr, 0, visited.bytes, s));
    launch_join_mark(li->as<uint32_t>(), li->as<uint32_t>(), M, visited.as<uint32_t>(), nullptr, s);
    const uint64_t T = select_by_bits(ctx, visited, NL, join_type == QHIP_JOIN_LEFT_SEMI ? 1 : 0, sel);
    auto keep = std::make_shared<DevBuf>(std::move(sel));
    if (!keep->ptr) keep->alloc(4);
    add_columns(keep, keep, T, false, false);
    out->num_rows = (int64_t)T;
    out->batch_offsets.push_back((int64_t)T);
    return out.release();
  }

  out->batch_offsets.push_back((int64_t)M);   // matched_batch
  if (join_type == QHIP_JOIN_INNER) {
    add_columns(li, ri, M, false, false);
    out->num_rows = (int64_t)M;
    return out.release();
  }
  // ---- Left / Right / Full: a second batch with the unmatched rows (:172-221): left ones first (NULL right side),
  // then right ones (NULL left side), each ascending
  uint64_t TL = 0, TR = 0;
  DevBuf sel_l, sel_r;
  if (join_type == QHIP_JOIN_LEFT || join_type == QHIP_JOIN_FULL) {
    DevBuf visited((((NL + 63) / 64) * 2 + 2) * 4);
    QHIP_HIP_CHECK(hipMemsetAsync(visited.ptr, 0, visited.bytes, s));
    launch_join_mark(li->as<uint32_t>(), li->as<uint32_t>(), M, visited.as<uint32_t>(), nullptr, s);
    TL = select_by_bits(ctx, visited, NL, 0, sel_l);
  }
  if (join_type == QHIP_JOIN_RIGHT || join_type == QHIP_JOIN_FULL) {
    DevBuf visited((((NR + 63) / 64) * 2 + 2) * 4);
    QHIP_HIP_CHECK(hipMemsetAsync(visited.ptr, 0, visited.bytes, s));
    launch_join_mark(ri->as<uint32_t>(), ri->as<uint32_t>(), M, visited.as<uint32_t>(), nullptr, s);
    TR = select_by_bits(ctx, visited, NR, 0, sel_r);
  }
  const uint64_t total = M + TL + TR;
  if (total >= kNullIdx) fail(QHIP_UNSUPPORTED, "join output of 2^32 - 1 rows or more is not supported");
  auto l_all = std::make_shared<DevBuf>((total + 1) * 4), r_all = std::make_shared<DevBuf>((total + 1) * 4);
  if (M) {
    QHIP_HIP_CHECK(hipMemcpyAsync(l_all->ptr, li->ptr, M * 4, hipMemcpyDeviceToDevice, s));
    QHIP_HIP_CHECK(hipMemcpyAsync(r_all->ptr, ri->ptr, M * 4, hipMemcpyDeviceToDevice, s));
  }
  if (TL) {
    QHIP_HIP_CHECK(hipMemcpyAsync(l_all->as<uint32_t>() + M, sel_l.ptr, TL * 4, hipMemcpyDeviceToDevice, s));
    launch_fill_u32(r_all->as<uint32_t>() + M, TL, kNullIdx, s);
  }
  if (TR) {
    launch_fill_u32(l_all->as<uint32_t>() + M + TL, TR, kNullIdx, s);
    QHIP_HIP_CHECK(hipMemcpyAsync(r_all->as<uint32_t>() + M + TL, sel_r.ptr, TR * 4, hipMemcpyDeviceToDevice, s));
  }
  QHIP_HIP_CHECK(sync_stream(s));   // sel_l / sel_r are about to be released
  add_columns(l_all, r_all, total, TR > 0, TL > 0);
  out->num_rows = (int64_t)total;
  out->batch_offsets.push_back((int64_t)total);   // unmatched_batch, possibly empty
  return out.release();
}

// CrossJoin::execute (cross_join.rs:121-166): for every left batch, for every right batch, for every left row: ONE output
// batch = that left row repeated next to the right batch
qhip_table* cross_join(Ctx* ctx, const qhip_table* L, const qhip_table* R) {
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  hipStream_t s = ctx->stream;
  const uint64_t total = (uint64_t)L->num_rows * (uint64_t)R->num_rows;
  if (total >= kMaxPairs) fail(QHIP_UNSUPPORTED, "cross join of more than 2^31 row pairs is not accelerated");
  auto li = std::make_shared<DevBuf>((total + 1) * 4), ri = std::make_shared<DevBuf>((total + 1) * 4);
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  out->batch_offsets = {0};
  uint64_t pos = 0;
  for (int64_t lb = 0; lb < L->num_batches(); ++lb) {
    const uint64_t l0 = (uint64_t)L->offsets()[(size_t)lb], nl = (uint64_t)L->offsets()[(size_t)lb + 1] - l0;
    for (int64_t rb = 0; rb < R->num_batches(); ++rb) {
      const uint64_t r0 = (uint64_t)R->offsets()[(size_t)rb], nr = (uint64_t)R->offsets()[(size_t)rb + 1] - r0;
      // left-row major inside the (left batch, right batch) block: pair k -> (l0 + k / nr, r0 + k % nr)
      if (nl > 0 && nr > 0) launch_pair_indices(ri->as<uint32_t>() + pos, li->as<uint32_t>() + pos, nl * nr, (uint32_t)nr, (uint32_t)r0, (uint32_t)l0, 0, s);
      for (uint64_t l = 0; l < nl; ++l) { pos += nr; out->batch_offsets.push_back((int64_t)pos); }
    }
  }
  defer_gather(ctx, L->cols, li, total, false, out->cols);
  defer_gather(ctx, R->cols, ri, total, false, out->cols);
  for (size_t c = 0; c < L->cols.size(); ++c) { out->names.push_back(L->names[c]); out->nullable.push_back(L->nullable[c]); }
  for (size_t c = 0; c < R->cols.size(); ++c) { out->names.push_back(R->names[c]); out->nullable.push_back(R->nullable[c]); }
  out->num_rows = (int64_t)total;
  return out.release();
}

}  // namespace

extern "C" int qhip_nested_loop_join_execute(qhip_ctx* ctx, const qhip_table* left, const qhip_table* right, int32_t join_type,
                                             const qhip_expr* filter_exprs, int32_t n_filter_exprs, int32_t filter_root,
                                             const int32_t* filter_sides, const int32_t* filter_cols, int32_t n_filter_cols, qhip_table** out) {
  if (!ctx || !left || !right || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] {
    settle_rows(left);
    settle_rows(right);
    *out = nested_loop_join(ctx, left, right, join_type, filter_exprs, n_filter_exprs, filter_root, filter_sides, filter_cols, n_filter_cols);
  });
}

extern "C" int qhip_cross_join_execute(qhip_ctx* ctx, const qhip_table* left, const qhip_table* right, qhip_table** out) {
  if (!ctx || !left || !right || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] { settle_rows(left); settle_rows(right); *out = cross_join(ctx, left, right); });
}
