// relops.hpp — shared host-side building blocks (see relops.cpp) + helpers defined in agg.cpp.
#pragma once
#include <vector>

#include "codegen.hpp"
#include "common.hpp"
#include "kargs_host.hpp"

namespace qhip {

// agg.cpp
// mark_indirect: plain deferred gathers (indirect_eligible) are typed as read THROUGH their index vector (InputCol::indirect)
std::vector<InputCol> input_cols_of(const qhip_table* t, bool mark_indirect = false);
bool indirect_eligible(const DevColumn& c);
void fill_kargs(Ctx* ctx, const qhip_table* t, const KernelBindings& b, HKArgs& a, DevBuf& strlit_dev);
void check_status_words(const uint32_t* st);

// relops.cpp
// keep-mask of `root` over every row of t: mask word j = ballot of rows 64j..64j+63, wave_count[j] = its popcount
void run_pred_mask(Ctx* ctx, const qhip_table* t, const ExprSet& es, const std::vector<InputCol>& icols, int root, DevBuf& mask,
                   DevBuf& wave_count, uint32_t* status_dev = nullptr);
// wave_count is scanned in place into per-wave output offsets; sel receives the kept row indices; returns their number
uint32_t select_from_mask(Ctx* ctx, const DevBuf& mask, DevBuf& wave_count, int64_t nrows, DevBuf& sel);
// ... in two steps: count_from_mask scans and returns the number of kept rows, indices_from_mask fills `sel` when (and if)
// somebody needs the selection vector
uint32_t count_from_mask(Ctx* ctx, DevBuf& wave_count, int64_t nrows);
void indices_from_mask(Ctx* ctx, const DevBuf& mask, const DevBuf& wave_offset, int64_t nrows, uint32_t m, DevBuf& sel);
// Mask-driven compaction of one column (null when the column's layout needs the selection vector: nullable, Boolean,
// Utf8 with values of more than one byte): out[k] = the k-th kept row's value, read in row order
bool compact_column(Ctx* ctx, const DevColumn& col, const DevBuf& mask, const DevBuf& wave_offset, int64_t nrows, uint32_t m, DevColumn& out);
// out[k] = col[idx[k]] (device u32 indices, kNullIdx -> NULL when idx_may_be_null)
DevColumn gather_column(Ctx* ctx, const DevColumn& col, const uint32_t* idx, uint64_t m, bool idx_may_be_null);
// The column itself, gathered now if it was deferred (the result is cached in the column's DeferredGather).
const DevColumn& resolved(Ctx* ctx, const DevColumn& col);
// Deferred out[k] = col[idx[k]] for every column of `cols`, appended to `out`: nothing is gathered until a column is
// read. Deferred inputs are composed (one u32 gather per distinct inner index vector), never chained.
void defer_gather(Ctx* ctx, const std::vector<DevColumn>& cols, const std::shared_ptr<DevBuf>& idx, uint64_t m, bool idx_may_be_null,
                  std::vector<DevColumn>& out);
void resolve_all(Ctx* ctx, const qhip_table* t);
// gather the columns the expression trees reference (everything else may stay deferred)
void resolve_referenced(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, bool keep_indirect = false);
// make sure every Utf8 column used directly as a key (roots are Column nodes) has its longest-value length cached in the
// table and copied into icols (packed key words are sized from it)
void ensure_utf8_key_lengths(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n,
                             std::vector<InputCol>& icols);
// |value| bounds of the Int64 / Decimal128 columns the expressions reference (DevColumn::value_maxabs -> icols): computed
// (one reduction + one read-back per column, cached on the column) only for inputs of `min_rows` rows or more
void ensure_value_bounds(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, std::vector<InputCol>& icols, int64_t min_rows);
// 4-byte narrow copies of the Int64 columns (join keys) a big probe side's expressions reference, where the value range allows
// (round 4) columns an aggregate reads through ONE index vector get a shared record copy of their source columns (ColRange::rec_buf)
void ensure_indirect_records(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, std::vector<InputCol>& icols, int64_t min_rows);
void ensure_narrow_int_columns(Ctx* ctx, const qhip_table* t, const qhip_expr* exprs, int n_exprs, std::vector<InputCol>& icols, int64_t min_rows);
// [mn, mx] of an integer-like column's values (cached on the column and shared with its source table, DevColumn::range);
// false when the column has no values to look at / is not integer-like
bool key_range_of(Ctx* ctx, const DevColumn& col, int64_t& mn, int64_t& mx);
// key words [W][N] + validity bitmap of the key expressions `roots`
void eval_key_words(Ctx* ctx, const qhip_table* t, const ExprSet& es, const std::vector<InputCol>& icols, const int32_t* roots, int n,
                    KeysPlan& kp, DevBuf& keys, DevBuf& keyvalid, int predicate_root = -1, bool deferred_status = false,
                    uint32_t* status_dev = nullptr);   // (status words: the context's block unless the caller keeps its own)

// exchange.cpp — the two streaming passes of the fused filter + partition (qhip_partition_filtered), also used by the aggregate's
// pre-partitioning (agg.cpp): pass 1 = scan filter + key -> part byte per row + scanned (part, unit) counters; partition_scatter
// = pass 2's launches for the kept columns (rows_only: nothing but the parts' selection vector, ranks in any order)
struct PartitionWork {       // what pass 1 leaves on the device for pass 2
  DevBuf trash{2048};        // where pass 2's unconditional stores of a tile without rows go
  DevBuf ids, runs, starts;  // part byte per row | scanned hist [n_parts * n_units] + total | the parts' first positions (n_parts + 1)
  uint32_t n_units = 0, rows_per_unit = 0;
  bool wg_units = false;     // a unit is a workgroup of pass 1 (pass 2 then runs its workgroup form)
  DevBuf bounds_dev;         // partition by key range: the n_parts - 1 upper bounds (PartIdsLaunch::bounds)
  std::vector<int64_t> bounds_host;
  uint32_t* dstat = nullptr;
  double pass1_bytes_per_row = 0;   // column bytes the filter + key expressions read per row
};
struct MovedColumn { size_t col; std::shared_ptr<DevBuf> out; int width; };   // a column's values of all parts, part after part
void partition_pass1(Ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int n_exprs, const int32_t* roots, int n_keys, int pred_root,
                     int n_parts, PartitionWork& w, const int64_t* range_bounds = nullptr);
void partition_scatter(Ctx* ctx, const qhip_table* in, const int32_t* keep, int n_parts, PartitionWork& w, uint64_t total,
                       std::vector<MovedColumn>& moved, std::vector<size_t>& odd, std::shared_ptr<DevBuf>& sel, bool rows_only);

}  // namespace qhip
