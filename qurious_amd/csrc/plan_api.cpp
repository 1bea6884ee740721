// plan_api.cpp — plan-only entry points (no GPU needed): return the kernel source libqhip would
// instantiate for a plan. __graft_entry__.build() uses them with qhip_jit_compile_to_cache to compile
// the benchmark/test catalog for gfx950 ahead of time; CPU tests use them to check the host logic.
#include <cstdio>
#include <cstdlib>

#include "codegen.hpp"
#include "common.hpp"

using namespace qhip;

static std::vector<InputCol> make_input(const qhip_dtype* t, const int32_t* has_nulls, int n) {
  std::vector<InputCol> v;
  for (int k = 0; k < n; ++k) { InputCol c; c.type = DType(t[k]); c.has_nulls = has_nulls && has_nulls[k]; v.push_back(c); }
  // QHIP_PLAN_VALUE_BITS="3:13,4:24": |value| of column 3 fits 13 bits, ... — the column statistic an execution would find
  // (relops.cpp ensure_value_bounds), so that the catalog can pre-compile the narrow variants of the benchmark kernels
  if (const char* e = getenv("QHIP_PLAN_VALUE_BITS")) {
    int col = 0, bits = 0, used = 0;
    while (*e && sscanf(e, "%d:%d%n", &col, &bits, &used) == 2) {
      if (col >= 0 && col < n && bits >= 1 && bits <= 63) {
        v[(size_t)col].value_maxabs = (1ULL << bits) - 1;
        // (... and the narrow copy an execution makes of such a Decimal128 column, DevColumn::narrow)
        if (v[(size_t)col].type.id == QHIP_DECIMAL128 && env_int("QHIP_NARROW_DECIMALS", 1) != 0) v[(size_t)col].narrow_bytes = bits <= 31 ? 4 : 8;
      }
      e += used;
      if (*e == ',') ++e;
    }
  }
  // QHIP_PLAN_NARROW_INTS="0,2": these Int64 columns are read through 4-byte narrow copies (what a big probe side's key columns
  // get at run time when their value range fits, relops.cpp ensure_narrow_int_columns)
  if (const char* e = getenv("QHIP_PLAN_NARROW_INTS")) {
    int col = 0, used = 0;
    while (*e && sscanf(e, "%d%n", &col, &used) == 1) {
      if (col >= 0 && col < n && v[(size_t)col].type.id == QHIP_INT64) v[(size_t)col].narrow_bytes = 4;
      e += used;
      if (*e == ',') ++e;
    }
  }
  // QHIP_PLAN_UTF8_FIXED1="1,2": every value of these Utf8 columns is exactly one byte long (what an execution finds out on the
  // device for TPC-H's flag columns, relops.cpp ensure_utf8_key_lengths): the kernel addresses their bytes by row number
  if (const char* e = getenv("QHIP_PLAN_UTF8_FIXED1")) {
    int col = 0, used = 0;
    while (*e && sscanf(e, "%d%n", &col, &used) == 1) {
      if (col >= 0 && col < n && v[(size_t)col].type.id == QHIP_UTF8) { v[(size_t)col].utf8_max_len = 1; v[(size_t)col].utf8_fixed1 = true; }
      e += used;
      if (*e == ',') ++e;
    }
  }
  // QHIP_PLAN_INDIRECT=1: every fixed-width column is a deferred gather read through its index vector (what an aggregate /
  // a join build over a join output sees at run time, InputCol::indirect)
  if (env_int("QHIP_PLAN_INDIRECT", 0))
    for (auto& c : v) if (dtype_width(c.type) > 0) { c.indirect = true; c.has_nulls = false; }
  // QHIP_PLAN_RECORDS="5:8,6:8": these indirect columns are read as fields of `stride`-byte record copies of their source
  // tables (relops.cpp ensure_indirect_records, InputCol::rec_stride)
  if (const char* e = getenv("QHIP_PLAN_RECORDS")) {
    int col = 0, stride = 0, used = 0;
    while (*e && sscanf(e, "%d:%d%n", &col, &stride, &used) == 2) {
      if (col >= 0 && col < n && v[(size_t)col].indirect && (stride == 8 || stride == 16)) v[(size_t)col].rec_stride = stride;
      e += used;
      if (*e == ',') ++e;
    }
  }
  return v;
}
static int give(const std::string& s, char* buf, size_t buflen, size_t* needed) {
  if (needed) *needed = s.size() + 1;
  if (buf && buflen) snprintf(buf, buflen, "%s", s.c_str());
  return QHIP_OK;
}
static thread_local std::string g_plan_err;

extern "C" {

const char* qhip_plan_last_error(void) { return g_plan_err.c_str(); }

int qhip_plan_aggregate_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                               int32_t n_exprs, int32_t predicate_root, const int32_t* group_roots, int32_t n_groups,
                               const qhip_agg* aggs, int32_t n_aggs, char* buf, size_t buflen, size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    AggPlan p;
    const char* r = getenv("QHIP_AGG_R");
    plan_aggregate(es, in, predicate_root, group_roots, n_groups, aggs, n_aggs, r && *r ? atoi(r) : 0, p, env_int("QHIP_PLAN_DEV_ROWS", 0) != 0);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_filter_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                            int32_t n_exprs, int32_t predicate_root, char* buf, size_t buflen, size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    MaskPlan p;
    plan_predicate_mask(es, in, predicate_root, p);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_keys_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                          int32_t n_exprs, const int32_t* key_roots, int32_t n_keys, char* buf, size_t buflen, size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    KeysPlan p;
    plan_keys(es, in, key_roots, n_keys, p);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_probe_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                           int32_t n_exprs, const int32_t* key_roots, int32_t n_keys, int32_t predicate_root, char* buf, size_t buflen,
                           size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    KeysPlan p;
    plan_keys(es, in, key_roots, n_keys, p, predicate_root, env_int("QHIP_PLAN_DENSE", 0) ? KEYS_KERNEL_DENSE_PROBE : KEYS_KERNEL_PROBE);   // (QHIP_PLAN_DENSE=1: the dense join layout's kernels)
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_scatter_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                             int32_t n_exprs, const int32_t* key_roots, int32_t n_keys, int32_t predicate_root, char* buf, size_t buflen,
                             size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    KeysPlan p;
    plan_keys(es, in, key_roots, n_keys, p, predicate_root, env_int("QHIP_PLAN_DENSE", 0) ? KEYS_KERNEL_DENSE_BUILD : KEYS_KERNEL_SCATTER, env_int("QHIP_PLAN_DEV_ROWS", 0) != 0);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_partition_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                               int32_t n_exprs, const int32_t* key_roots, int32_t n_keys, int32_t predicate_root, int32_t n_parts, char* buf,
                               size_t buflen, size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    // (like an execution: Utf8 key columns of the exchange always pack into 4 words)
    for (int k = 0; k < n_keys; ++k)
      if (key_roots[k] >= 0 && key_roots[k] < n_exprs && exprs[key_roots[k]].kind == QHIP_EXPR_COLUMN && exprs[key_roots[k]].column >= 0 &&
          exprs[key_roots[k]].column < n_cols && in[(size_t)exprs[key_roots[k]].column].type.id == QHIP_UTF8)
        in[(size_t)exprs[key_roots[k]].column].utf8_max_len = 31;
    ExprSet es; es.build(exprs, n_exprs, in);
    KeysPlan p;
    plan_keys(es, in, key_roots, n_keys, p, predicate_root, KEYS_KERNEL_PARTITION, env_int("QHIP_PLAN_DEV_ROWS", 0) != 0, n_parts);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_part_scatter_source(const int32_t* widths, const int32_t* indirect, int32_t n_cols, int32_t n_parts, char* buf, size_t buflen,
                                  size_t* needed) {
  try {
    std::vector<int> w(widths, widths + (n_cols > 0 ? n_cols : 0));
    std::vector<char> ind;
    for (int k = 0; k < n_cols; ++k) ind.push_back(indirect && indirect[k] ? 1 : 0);
    PartScatterPlan p;
    plan_part_scatter(w, ind, n_parts, env_int("QHIP_PLAN_DEV_ROWS", 0) != 0, p);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_sort_keys_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                               int32_t n_exprs, const int32_t* key_roots, int32_t n_keys, char* buf, size_t buflen, size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    SortKeysPlan p;
    plan_sort_keys(es, in, key_roots, n_keys, p);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

int qhip_plan_projection_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols, const qhip_expr* exprs,
                                int32_t n_exprs, const int32_t* roots, int32_t n_out, char* buf, size_t buflen, size_t* needed) {
  try {
    auto in = make_input(col_types, col_has_nulls, n_cols);
    ExprSet es; es.build(exprs, n_exprs, in);
    std::vector<int32_t> computed;
    for (int k = 0; k < n_out; ++k)
      if (roots[k] >= 0 && roots[k] < n_exprs && exprs[roots[k]].kind != QHIP_EXPR_COLUMN) computed.push_back(roots[k]);
    ProjectionPlan p;
    plan_projection(es, in, computed.data(), (int)computed.size(), p);
    return give(p.source, buf, buflen, needed);
  } catch (const Error& e) { g_plan_err = e.what(); return e.code; }
}

}  // extern "C"
