// mirror_tests.cpp — the reference's own operator unit tests, written against the C++ host mirror (include/qhip_plan.hpp)
// the way they are written in Rust against the reference: hash_join.rs:396-914, sort.rs:97-205, limit.rs:64-95,
// nest_loop_join.rs:356-408, cross_join.rs:181-213, binary.rs:100-251 and the aggregation / group_by .slt files.
// Same inputs, same expected rows in the same order. Needs an MI355X; exits non-zero on the first mismatch.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>

#include "../../include/qhip_plan.hpp"

using namespace qurious_hip;
using Rows = std::vector<std::vector<std::optional<int64_t>>>;

// ---------------------------------------------------------------- Arrow int32 batches without an Arrow library
struct Owned {
  std::vector<std::vector<int32_t>> values;
  std::vector<std::vector<uint8_t>> validity;
  std::vector<const void*> buffers[16];
  std::vector<ArrowArray> children;
  std::vector<ArrowArray*> child_ptrs;
  const void* top_buffers[1] = {nullptr};
};
static void release_owned_array(ArrowArray* a) { delete (Owned*)a->private_data; a->release = nullptr; }
static void release_noop_array(ArrowArray* a) { a->release = nullptr; }
struct OwnedSchema {
  std::vector<std::string> names;
  std::vector<ArrowSchema> children;
  std::vector<ArrowSchema*> child_ptrs;
};
static void release_owned_schema(ArrowSchema* s) { delete (OwnedSchema*)s->private_data; s->release = nullptr; }
static void release_noop_schema(ArrowSchema* s) { s->release = nullptr; }

using Col = std::pair<std::string, std::vector<std::optional<int32_t>>>;
// test_utils::build_table_scan_i32: one batch of Int32 columns
static PlanRef build_table_scan_i32(const ContextRef& ctx, const std::vector<Col>& cols, ExprRef filter = nullptr) {
  auto* os = new OwnedSchema();
  auto* schema = new ArrowSchema();
  memset(schema, 0, sizeof *schema);
  os->children.resize(cols.size());
  for (size_t c = 0; c < cols.size(); ++c) {
    os->names.push_back(cols[c].first);
  }
  for (size_t c = 0; c < cols.size(); ++c) {
    ArrowSchema& cs = os->children[c];
    memset(&cs, 0, sizeof cs);
    cs.format = "i"; cs.name = os->names[c].c_str(); cs.flags = ARROW_FLAG_NULLABLE; cs.release = release_noop_schema;
    os->child_ptrs.push_back(&cs);
  }
  schema->format = "+s"; schema->name = ""; schema->n_children = (int64_t)cols.size(); schema->children = os->child_ptrs.data();
  schema->release = release_owned_schema; schema->private_data = os;

  auto* ow = new Owned();
  auto* arr = new ArrowArray();
  memset(arr, 0, sizeof *arr);
  const int64_t n = cols.empty() ? 0 : (int64_t)cols[0].second.size();
  ow->children.resize(cols.size());
  ow->values.resize(cols.size());
  ow->validity.resize(cols.size());
  for (size_t c = 0; c < cols.size(); ++c) {
    int64_t nulls = 0;
    ow->values[c].resize((size_t)n + 1);
    ow->validity[c].assign((size_t)(n + 7) / 8 + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
      if (cols[c].second[(size_t)i]) { ow->values[c][(size_t)i] = *cols[c].second[(size_t)i]; ow->validity[c][(size_t)i >> 3] |= (uint8_t)(1u << (i & 7)); }
      else ++nulls;
    }
    ow->buffers[c] = {nulls ? (const void*)ow->validity[c].data() : nullptr, (const void*)ow->values[c].data()};
    ArrowArray& ca = ow->children[c];
    memset(&ca, 0, sizeof ca);
    ca.length = n; ca.null_count = nulls; ca.n_buffers = 2; ca.buffers = ow->buffers[c].data(); ca.release = release_noop_array;
    ow->child_ptrs.push_back(&ca);
  }
  arr->length = n; arr->n_buffers = 1; arr->buffers = ow->top_buffers; arr->n_children = (int64_t)cols.size();
  arr->children = ow->child_ptrs.data(); arr->release = release_owned_array; arr->private_data = ow;
  auto table = std::make_shared<MemoryTable>(ctx, schema, std::vector<ArrowArray*>{arr});
  delete arr;
  delete schema;
  return std::make_shared<Scan>(table, std::move(filter));
}

// rows of a result (Int32 / Int64 / Float64-as-integer columns), NULL -> nullopt
static Rows rows_of(const std::vector<RecordBatch>& batches) {
  Rows out;
  for (const auto& b : batches) {
    for (int64_t i = 0; i < b.array.length; ++i) {
      std::vector<std::optional<int64_t>> row;
      for (int64_t c = 0; c < b.array.n_children; ++c) {
        const ArrowArray* ca = b.array.children[c];
        const char* fmt = b.schema.children[c]->format;
        const uint8_t* valid = (const uint8_t*)ca->buffers[0];
        if (valid && !((valid[i >> 3] >> (i & 7)) & 1)) { row.push_back(std::nullopt); continue; }
        if (!strcmp(fmt, "i")) row.push_back(((const int32_t*)ca->buffers[1])[i]);
        else if (!strcmp(fmt, "l")) row.push_back(((const int64_t*)ca->buffers[1])[i]);
        else if (!strcmp(fmt, "g")) row.push_back((int64_t)((const double*)ca->buffers[1])[i]);
        else if (!strcmp(fmt, "b")) row.push_back((((const uint8_t*)ca->buffers[1])[i >> 3] >> (i & 7)) & 1);
        else { fprintf(stderr, "unexpected column format %s\n", fmt); exit(2); }
      }
      out.push_back(std::move(row));
    }
  }
  return out;
}
static std::string show(const Rows& r) {
  std::string s;
  for (auto& row : r) { s += "("; for (auto& v : row) s += (v ? std::to_string(*v) : std::string("NULL")) + ","; s += ") "; }
  return s;
}
static int failures = 0, checks = 0;
static void expect_rows(const char* name, const std::vector<RecordBatch>& got, const Rows& want, int want_batches = -1) {
  ++checks;
  const Rows g = rows_of(got);
  if (g != want || (want_batches >= 0 && (int)got.size() != want_batches)) {
    ++failures;
    fprintf(stderr, "FAIL %s\n  got  (%zu batches) %s\n  want %s\n", name, got.size(), show(g).c_str(), show(want).c_str());
  } else {
    printf("ok   %s\n", name);
  }
}
static std::optional<int64_t> N = std::nullopt;
static ExprRef col(const char* n, int i) { return std::make_shared<Column>(n, i); }
static ExprRef lit_i32(int32_t v) { return std::make_shared<Literal>(ScalarValue::Int32(v)); }

int main() {
  ContextRef ctx;
  try {
    ctx = std::make_shared<Context>();
  } catch (const Error& e) {
    fprintf(stderr, "no usable MI355X: %s\n", e.what());
    return 77;
  }
  using C3 = std::vector<std::optional<int32_t>>;
  auto tbl3 = [&](const char* a, C3 av, const char* b, C3 bv, const char* c, C3 cv) {
    return build_table_scan_i32(ctx, {{a, std::move(av)}, {b, std::move(bv)}, {c, std::move(cv)}});
  };
  std::vector<std::pair<ExprRef, ExprRef>> on_b = {{col("b1", 1), col("b2", 1)}};
  {   // hash_join.rs:396-698, 889-914 — inputs and expected rows (in order) exactly as the reference asserts them
    expect_rows("test_hash_join_inner",
                HashJoinExec::try_new(tbl3("a1", {1, 2, 3}, "b1", {4, 5, 6}, "c1", {7, 8, 9}), tbl3("a2", {10, 20, 30}, "b2", {4, 5, 6}, "c2", {70, 80, 90}),
                                      QHIP_JOIN_INNER, on_b)->execute(),
                {{1, 4, 7, 10, 4, 70}, {2, 5, 8, 20, 5, 80}, {3, 6, 9, 30, 6, 90}});
    expect_rows("test_join_inner_one_no_shared_column_names",
                HashJoinExec::try_new(tbl3("a1", {1, 2, 3}, "b1", {4, 5, 5}, "c1", {7, 8, 9}), tbl3("a2", {10, 20, 30}, "b2", {4, 5, 6}, "c2", {70, 80, 90}),
                                      QHIP_JOIN_INNER, on_b)->execute(),
                {{1, 4, 7, 10, 4, 70}, {2, 5, 8, 20, 5, 80}, {3, 5, 9, 20, 5, 80}});
    expect_rows("test_join_inner_one_randomly_ordered",
                HashJoinExec::try_new(tbl3("a1", {0, 3, 2, 1}, "b1", {4, 5, 5, 4}, "c1", {6, 9, 8, 7}), tbl3("a2", {20, 30, 10}, "b2", {5, 6, 4}, "c2", {80, 90, 70}),
                                      QHIP_JOIN_INNER, on_b)->execute(),
                {{3, 5, 9, 20, 5, 80}, {2, 5, 8, 20, 5, 80}, {0, 4, 6, 10, 4, 70}, {1, 4, 7, 10, 4, 70}});
    expect_rows("test_join_inner_two",
                HashJoinExec::try_new(tbl3("a1", {1, 2, 2}, "b2", {1, 2, 2}, "c1", {7, 8, 9}), tbl3("a1", {1, 2, 3}, "b2", {1, 2, 2}, "c2", {70, 80, 90}),
                                      QHIP_JOIN_INNER, {{col("a1", 0), col("a1", 0)}, {col("b2", 1), col("b2", 1)}})->execute(),
                {{1, 1, 7, 1, 1, 70}, {2, 2, 8, 2, 2, 80}, {2, 2, 9, 2, 2, 80}});
    expect_rows("test_join_left_one",
                HashJoinExec::try_new(tbl3("a1", {1, 2, 3}, "b1", {4, 5, 7}, "c1", {7, 8, 9}), tbl3("a2", {10, 20, 30}, "b1", {4, 5, 6}, "c2", {70, 80, 90}),
                                      QHIP_JOIN_LEFT, {{col("b1", 1), col("b1", 1)}})->execute(),
                {{1, 4, 7, 10, 4, 70}, {2, 5, 8, 20, 5, 80}, {3, 7, 9, N, N, N}});
    expect_rows("test_join_left_empty_right",
                HashJoinExec::try_new(tbl3("a1", {1, 2, 3}, "b1", {4, 5, 7}, "c1", {7, 8, 9}), tbl3("a2", {}, "b2", {}, "c2", {}), QHIP_JOIN_LEFT, on_b)->execute(),
                {{1, 4, 7, N, N, N}, {2, 5, 8, N, N, N}, {3, 7, 9, N, N, N}});
    expect_rows("test_join_full",
                HashJoinExec::try_new(tbl3("a1", {1, 2, 3}, "b1", {4, 5, 7}, "c1", {7, 8, 9}), tbl3("a2", {10, 20, 30}, "b2", {4, 5, 6}, "c2", {70, 80, 90}),
                                      QHIP_JOIN_FULL, on_b)->execute(),
                {{1, 4, 7, 10, 4, 70}, {2, 5, 8, 20, 5, 80}, {N, N, N, 30, 6, 90}, {3, 7, 9, N, N, N}});
    expect_rows("test_join_full_one",
                HashJoinExec::try_new(build_table_scan_i32(ctx, {{"v1", {1, 2, 3}}, {"v2", {1, 2, 3}}}), build_table_scan_i32(ctx, {{"v3", {1, 3, 4}}, {"v4", {100, 300, 400}}}),
                                      QHIP_JOIN_FULL, {{col("v1", 0), col("v3", 0)}})->execute(),
                {{1, 1, 1, 100}, {3, 3, 3, 300}, {N, N, 4, 400}, {2, 2, N, N}});
    expect_rows("test_hash_join_left_semi_distinct_left_rows",
                HashJoinExec::try_new(build_table_scan_i32(ctx, {{"a1", {1, 2, 3}}, {"k1", {10, 20, 30}}}), build_table_scan_i32(ctx, {{"k2", {10, 10, 999}}, {"b2", {1, 2, 3}}}),
                                      QHIP_JOIN_LEFT_SEMI, {{col("k1", 1), col("k2", 0)}})->execute(),
                {{1, 10}});
    bool threw = false;
    try {
      HashJoinExec::try_new(build_table_scan_i32(ctx, {{"x", {1}}}), build_table_scan_i32(ctx, {{"y", {1}}}), QHIP_JOIN_INNER, {});
    } catch (const Error& e) { threw = std::string(e.what()).find("should be non-empty") != std::string::npos; }
    ++checks; if (!threw) { ++failures; fprintf(stderr, "FAIL empty ON clause must be an error\n"); } else printf("ok   join_empty_on_is_an_error\n");
  }
  {   // sort.rs: test_sort / test_sort_is_stable_for_equal_keys / test_sort_with_limit_returns_top_n_only
    PlanRef t1 = build_table_scan_i32(ctx, {{"a", {1, 2, 3, 4}}, {"c", {1, 2, 3, 4}}});
    expect_rows("test_sort", Sort({{col("a", 0), {true, false}}}, t1).execute(), {{4, 4}, {3, 3}, {2, 2}, {1, 1}}, 1);
    PlanRef in = build_table_scan_i32(ctx, {{"a", {1, 4, 2, 3}}, {"b", {10, 40, 20, 30}}});
    expect_rows("test_sort_with_limit_returns_top_n_only", Sort({{col("a", 0), {true, false}}}, in, 2).execute(), {{4, 40}, {3, 30}}, 1);
    PlanRef st = build_table_scan_i32(ctx, {{"a", {1, 1, 1, 2, 2}}, {"b", {10, 11, 12, 20, 21}}});
    expect_rows("test_sort_is_stable_for_equal_keys", Sort({{col("a", 0), {false, false}}}, st).execute(), {{1, 10}, {1, 11}, {1, 12}, {2, 20}, {2, 21}}, 1);
    // order_by.slt with NULLs, planner options (nulls_first = true)
    PlanRef nl = build_table_scan_i32(ctx, {{"v1", {1, 2, N, 2}}, {"v2", {0, 2, 5, N}}});
    expect_rows("order_by_nulls_first", Sort({{col("v1", 0), {false, true}}, {col("v2", 1), {false, true}}}, nl).execute(),
                {{N, 5}, {1, 0}, {2, N}, {2, 2}}, 1);
  }
  {   // limit.rs test_limit: Limit::new(input, Some(3), 1)
    PlanRef in = build_table_scan_i32(ctx, {{"a", {1, 2, 3, 4}}, {"c", {1, 2, 3, 4}}});
    expect_rows("limit_3_skip_1", Limit(in, 3, 1).execute(), {{2, 2}, {3, 3}, {4, 4}});
    expect_rows("limit_0", Limit(in, 0, 0).execute(), {});
  }
  {   // nest_loop_join.rs: left anti with an empty right side; left semi with duplicate matches
    PlanRef l = build_table_scan_i32(ctx, {{"a1", {1, 2, 3}}, {"k1", {10, 20, 30}}});
    PlanRef empty = build_table_scan_i32(ctx, {{"k2", {}}, {"b2", {}}});
    expect_rows("nlj_left_anti_empty_right", NestedLoopJoinExec(l, empty, QHIP_JOIN_LEFT_ANTI).execute(), {{1, 10}, {2, 20}, {3, 30}});
    PlanRef r = build_table_scan_i32(ctx, {{"k2", {10, 10, 999}}, {"b2", {1, 2, 3}}});
    JoinFilter f{std::make_shared<BinaryExpr>(col("k1", 0), QHIP_OP_EQ, col("k2", 1)), {{1, JoinSide::Left}, {0, JoinSide::Right}}};
    expect_rows("nlj_left_semi_distinct_left_rows", NestedLoopJoinExec(l, r, QHIP_JOIN_LEFT_SEMI, f).execute(), {{1, 10}});
  }
  {   // cross_join.rs test_cross_join: 3 output batches
    PlanRef l = build_table_scan_i32(ctx, {{"a1", {1, 2, 3}}, {"b1", {4, 5, 6}}, {"c1", {7, 8, 9}}});
    PlanRef r = build_table_scan_i32(ctx, {{"a2", {10, 11}}, {"b2", {12, 13}}, {"c2", {14, 15}}});
    expect_rows("cross_join", CrossJoin(l, r).execute(),
                {{1, 4, 7, 10, 12, 14}, {1, 4, 7, 11, 13, 15}, {2, 5, 8, 10, 12, 14}, {2, 5, 8, 11, 13, 15}, {3, 6, 9, 10, 12, 14}, {3, 6, 9, 11, 13, 15}}, 3);
  }
  {   // aggregation.slt / group_by.slt shapes: sum(v1) group by v2 over ((1,1),(2,1),(3,2),(4,2)) wait for v2 > 1; filter.slt
    PlanRef t = build_table_scan_i32(ctx, {{"v1", {1, 2, 3, 4}}, {"v2", {1, 1, 2, 2}}});
    auto sum_by = HashAggregate({"v2", "s"}, t, {col("v2", 1)}, {AggregateExpr::Sum(std::make_shared<CastExpr>(col("v1", 0), dtype(QHIP_INT64)), dtype(QHIP_INT64))});
    Rows got = rows_of(sum_by.execute());
    std::sort(got.begin(), got.end());
    ++checks;
    if (got != Rows{{1, 3}, {2, 7}}) { ++failures; fprintf(stderr, "FAIL sum group by: %s\n", show(got).c_str()); } else printf("ok   sum_group_by\n");
    PlanRef scan_f = build_table_scan_i32(ctx, {{"v1", {1, 2, 3, 4}}, {"v2", {1, 1, 2, 2}}}, std::make_shared<BinaryExpr>(col("v1", 0), QHIP_OP_GT, lit_i32(1)));
    expect_rows("count_max_filtered_no_grouping",
                NoGroupingAggregate({"c", "m"}, scan_f, {AggregateExpr::Count(col("v1", 0)), AggregateExpr::Max(col("v2", 1), dtype(QHIP_INT32))}).execute(), {{3, 2}});
    expect_rows("filter_node", Filter(t, std::make_shared<BinaryExpr>(col("v2", 1), QHIP_OP_EQ, lit_i32(2))).execute(), {{3, 2}, {4, 2}});
    // binary.rs arithmetic through a Projection, CASE on top
    auto plus = std::make_shared<BinaryExpr>(col("v1", 0), QHIP_OP_ADD, col("v2", 1));
    auto kase = std::make_shared<CaseExpr>(std::vector<std::pair<ExprRef, ExprRef>>{{std::make_shared<BinaryExpr>(col("v1", 0), QHIP_OP_GT, lit_i32(2)), plus}}, lit_i32(0));
    expect_rows("projection_case", Projection({"v1", "x"}, t, {col("v1", 0), kase}).execute(), {{1, 0}, {2, 0}, {3, 5}, {4, 6}});
  }
  {   // no reference counterpart: a repeated join under an aggregate leaves its size on the device (detail::feed /
      // detail::retrying, qhip_ctx_allow_deferred_sizes) — same rows every time, one host wait from the second run on
    std::vector<std::optional<int32_t>> bk, bv, pk, pv;
    for (int i = 0; i < 3000; ++i) { bk.push_back(i); bv.push_back(i % 7); }
    for (int i = 0; i < 9000; ++i) { pk.push_back((i * 7919) % 4500); pv.push_back(i % 5); }
    PlanRef build = build_table_scan_i32(ctx, {{"bk", bk}, {"bv", bv}}), probe = build_table_scan_i32(ctx, {{"pk", pk}, {"pv", pv}});
    auto join = HashJoinExec::try_new(build, probe, QHIP_JOIN_INNER, {{col("bk", 0), col("pk", 0)}});
    HashAggregate agg({"pv", "n", "s"}, join, {col("pv", 3)},
                      {AggregateExpr::Count(col("bk", 0)), AggregateExpr::Sum(std::make_shared<CastExpr>(col("bv", 1), dtype(QHIP_INT64)), dtype(QHIP_INT64))});
    Rows first;
    uint64_t waits_last = 0;
    for (int run = 0; run < 4; ++run) {
      const uint64_t before = qhip_ctx_sync_count(ctx->raw());
      DeviceTableRef t = agg.execute_device();
      waits_last = qhip_ctx_sync_count(ctx->raw()) - before;
      Rows got = rows_of(agg.execute());
      std::sort(got.begin(), got.end());
      if (run == 0) first = got;
      ++checks;
      if (got != first || got.size() != 5) { ++failures; fprintf(stderr, "FAIL join under aggregate, run %d: %s\n", run, show(got).c_str()); }
    }
    ++checks;
    if (waits_last != 1) { ++failures; fprintf(stderr, "FAIL repeated join under aggregate: %llu host waits, expected 1\n", (unsigned long long)waits_last); }
    else printf("ok   repeated_join_under_aggregate_waits_once\n");
  }
  printf("%d checks, %d failures\n", checks, failures);
  return failures ? 1 : 0;
}
