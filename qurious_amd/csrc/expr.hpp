// expr.hpp — typed expression IR built from the qhip_expr POD tree, with the arrow-rs 53 type
// rules the reference relies on (physical/expr/binary.rs:31-70, cast.rs:33-37).
#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace qhip {

struct ENode {
  int kind = 0, op = 0, column = -1, left = -1, right = -1, third = -1;
  DType type;
  bool nullable = false;   // can evaluate to NULL for some row
  // literal payload
  bool lit_null = false;
  uint64_t lo = 0;
  int64_t hi = 0;
  double f = 0;
  std::string s;
  DType cast_to;           // CAST target
  std::string canon;       // structural identity (for common-subexpression sharing)
  // upper bound of |value| over all rows for integer / Decimal128 nodes (the unscaled integer), from the columns' cached
  // statistics, the literals' values and the operators; kUnbounded = nothing better than the type is known
  u128 maxabs = ~(u128)0;
};
constexpr u128 kUnbounded = ~(u128)0;

struct InputCol {
  DType type;
  bool has_nulls = false;  // null_count > 0 in the table actually being executed
  int utf8_max_len = -1;   // longest value of a Utf8 column in bytes when known (sizes packed join / group keys)
  bool utf8_fixed1 = false;   // every value of the Utf8 column is exactly 1 byte long: offsets[i] == i, nothing to load for them
  uint64_t value_maxabs = 0;  // DevColumn::value_maxabs (0 = unknown)
  int narrow_bytes = 0;       // 4 / 8: the kernel reads the column's narrow copy (DevColumn::narrow) instead of its 16-byte values
  // the column is a deferred gather that the kernel reads THROUGH its index vector (value = source[index[row]]) instead of
  // a gathered copy: what an aggregate / a join build reads from a join output are a few short columns, each a gather launch
  // and a write + read of the copy otherwise (KCol::v = the source values, KCol::d = the u32 index vector)
  bool indirect = false;
  // ... from a RECORD copy of its source (ColRange::rec_buf: the source's values interleaved with those of the other columns
  // the same kernel reads through the same index vector — one random 64-byte access per row and table instead of one per
  // column): the stride in bytes, 0 = the column's own array
  int rec_stride = 0;
};

struct ExprSet {
  std::vector<ENode> nodes;
  // Parses and types the POD tree against `input`. Literal casts are folded on the host (the reference
  // re-materialises and re-casts N copies of the literal per batch, literal.rs:20-22 + cast.rs:33-37).
  void build(const qhip_expr* exprs, int n, const std::vector<InputCol>& input);
  const ENode& at(int k) const { return nodes.at((size_t)k); }
};

// host-side scalar cast with arrow's safe=false semantics; throws QHIP_EXEC_ERROR on overflow / parse failure
void fold_literal_cast(const ENode& src, const DType& to, ENode& out);
int32_t parse_date32(const std::string& s);   // "YYYY-MM-DD" -> days since epoch; throws QHIP_EXEC_ERROR
i128 pow10_i128(int e);
// LIKE pattern -> matcher tokens: literal bytes, 0xFF for %, 0xFE for _ (neither byte occurs in UTF-8); `\\x` -> x
std::string canonical_like_pattern(const std::string& pattern);

}  // namespace qhip
