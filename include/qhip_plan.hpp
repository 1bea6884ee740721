// qhip_plan.hpp — the reference's operator interface for the accelerated path, in C++ above the C ABI of qhip.h.
//
// The reference is compiled code (Rust); its toolchain is not available in this build, so the host side a maintainer
// would write in Rust is written here in C++ with the same shape: `PhysicalPlan { schema / execute /
// children }` (physical/plan/mod.rs:25-29), `PhysicalExpr` nodes (physical/expr/*.rs), the aggregate expressions
// (physical/expr/aggregate/*.rs) and the plan nodes Scan, Filter, Projection, HashAggregate, NoGroupingAggregate,
// HashJoinExec, NestedLoopJoinExec, CrossJoin, Sort, Limit — same names, same constructor arguments, errors surfaced as
// exceptions carrying the reference's messages. Header-only, C++17, depends on qhip.h and libqhip.so only.
// (qurious_amd/*.py is the same mirror in Python; the parity tests use that one because they need pyarrow.)
#pragma once
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "qhip.h"

namespace qurious_hip {

// crate::error::Error::{InternalError, ArrowError} (error.rs:20-31): message + the C ABI status it came with
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

class Context {
 public:
  explicit Context(int device_index = -1) {
    const int rc = qhip_ctx_create(device_index, &ctx_);
    if (rc != QHIP_OK) throw Error(rc, qhip_last_error(nullptr));
  }
  ~Context() { qhip_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  qhip_ctx* raw() const { return ctx_; }
  void check(int rc) const { if (rc != QHIP_OK) throw Error(rc, qhip_last_error(ctx_)); }

 private:
  qhip_ctx* ctx_ = nullptr;
};
using ContextRef = std::shared_ptr<Context>;

// the device-resident Vec<RecordBatch> handed between operators
class DeviceTable {
 public:
  DeviceTable(ContextRef ctx, qhip_table* t) : ctx_(std::move(ctx)), t_(t) {}
  ~DeviceTable() { qhip_table_destroy(t_); }
  DeviceTable(const DeviceTable&) = delete;
  DeviceTable& operator=(const DeviceTable&) = delete;
  qhip_table* raw() const { return t_; }
  const ContextRef& ctx() const { return ctx_; }
  int64_t num_rows() const { return qhip_table_num_rows(t_); }
  int64_t num_batches() const { return qhip_table_num_batches(t_); }
  int64_t num_columns() const { return qhip_table_num_columns(t_); }

 private:
  ContextRef ctx_;
  qhip_table* t_;
};
using DeviceTableRef = std::shared_ptr<DeviceTable>;

// ---------------------------------------------------------------- datatypes (datatypes/{scalar,operator}.rs, common/join_type.rs)
inline qhip_dtype dtype(int id, int precision = 0, int scale = 0) { return qhip_dtype{id, precision, scale}; }
using Operator = qhip_operator;
using JoinType = qhip_join_type;
enum class JoinSide { Left = 0, Right = 1 };

struct ScalarValue {
  qhip_dtype type{QHIP_NULL, 0, 0};
  bool is_null = false;
  uint64_t lo = 0;
  int64_t hi = 0;
  double f64 = 0;
  std::string str;
  static ScalarValue Int32(int32_t v) { ScalarValue s; s.type = dtype(QHIP_INT32); s.lo = (uint64_t)(int64_t)v; return s; }
  static ScalarValue Int64(int64_t v) { ScalarValue s; s.type = dtype(QHIP_INT64); s.lo = (uint64_t)v; return s; }
  static ScalarValue Float64(double v) { ScalarValue s; s.type = dtype(QHIP_FLOAT64); s.f64 = v; return s; }
  static ScalarValue Boolean(bool v) { ScalarValue s; s.type = dtype(QHIP_BOOL); s.lo = v ? 1 : 0; return s; }
  static ScalarValue Date32(int32_t days) { ScalarValue s; s.type = dtype(QHIP_DATE32); s.lo = (uint64_t)(int64_t)days; return s; }
  static ScalarValue Utf8(std::string v) { ScalarValue s; s.type = dtype(QHIP_UTF8); s.str = std::move(v); return s; }
  static ScalarValue Decimal128(__int128 unscaled, int precision, int scale) {
    ScalarValue s; s.type = dtype(QHIP_DECIMAL128, precision, scale);
    s.lo = (uint64_t)(unsigned __int128)unscaled; s.hi = (int64_t)((unsigned __int128)unscaled >> 64); return s;
  }
  static ScalarValue Null(qhip_dtype t) { ScalarValue s; s.type = t; s.is_null = true; return s; }
};

// ---------------------------------------------------------------- physical expressions (physical/expr/*.rs)
class ExprArray;
struct PhysicalExpr {
  virtual ~PhysicalExpr() = default;
  virtual int lower(ExprArray& out) const = 0;   // appends this tree to the flat qhip_expr array, returns its root index
  virtual std::string to_string() const = 0;     // the reference's Display
};
using ExprRef = std::shared_ptr<const PhysicalExpr>;

class ExprArray {
 public:
  int add(qhip_expr e) { nodes_.push_back(e); return (int)nodes_.size() - 1; }
  static qhip_expr node(int kind) {
    qhip_expr e;
    memset(&e, 0, sizeof e);
    e.kind = kind; e.column = -1; e.left = -1; e.right = -1; e.third = -1;
    return e;
  }
  const char* keep(const std::string& s) { strings_.push_back(std::make_unique<std::string>(s)); return strings_.back()->c_str(); }
  const qhip_expr* data() const { return nodes_.empty() ? nullptr : nodes_.data(); }
  int size() const { return (int)nodes_.size(); }

 private:
  std::vector<qhip_expr> nodes_;
  std::vector<std::unique_ptr<std::string>> strings_;
};

struct Column : PhysicalExpr {   // column.rs
  std::string name; int index;
  Column(std::string n, int i) : name(std::move(n)), index(i) {}
  int lower(ExprArray& o) const override { qhip_expr e = ExprArray::node(QHIP_EXPR_COLUMN); e.column = index; return o.add(e); }
  std::string to_string() const override { return name + "@" + std::to_string(index); }
};
struct Literal : PhysicalExpr {   // literal.rs
  ScalarValue value;
  explicit Literal(ScalarValue v) : value(std::move(v)) {}
  int lower(ExprArray& o) const override {
    qhip_expr e = ExprArray::node(QHIP_EXPR_LITERAL);
    e.dtype = value.type; e.lit_is_null = value.is_null ? 1 : 0; e.lit_lo = value.lo; e.lit_hi = value.hi; e.lit_f64 = value.f64;
    if (value.type.id == QHIP_UTF8) { e.lit_str = o.keep(value.str); e.lit_len = (int64_t)value.str.size(); }
    return o.add(e);
  }
  std::string to_string() const override { return value.type.id == QHIP_UTF8 ? value.str : std::to_string((long long)value.lo); }
};
struct BinaryExpr : PhysicalExpr {   // binary.rs:17-70
  ExprRef left, right; Operator op;
  BinaryExpr(ExprRef l, Operator o, ExprRef r) : left(std::move(l)), right(std::move(r)), op(o) {}
  int lower(ExprArray& o) const override {
    qhip_expr e = ExprArray::node(QHIP_EXPR_BINARY);
    e.left = left->lower(o); e.right = right->lower(o); e.op = (int)op;
    return o.add(e);
  }
  std::string to_string() const override { return left->to_string() + " op" + std::to_string((int)op) + " " + right->to_string(); }
};
struct CastExpr : PhysicalExpr {   // cast.rs (CastOptions { safe: false })
  ExprRef expr; qhip_dtype to;
  CastExpr(ExprRef e, qhip_dtype t) : expr(std::move(e)), to(t) {}
  int lower(ExprArray& o) const override { qhip_expr e = ExprArray::node(QHIP_EXPR_CAST); e.left = expr->lower(o); e.dtype = to; return o.add(e); }
  std::string to_string() const override { return "CAST(" + expr->to_string() + ")"; }
};
template <int KIND> struct UnaryExpr : PhysicalExpr {
  ExprRef expr;
  explicit UnaryExpr(ExprRef e) : expr(std::move(e)) {}
  int lower(ExprArray& o) const override { qhip_expr e = ExprArray::node(KIND); e.left = expr->lower(o); return o.add(e); }
  std::string to_string() const override { return "unary" + std::to_string(KIND) + "(" + expr->to_string() + ")"; }
};
using IsNull = UnaryExpr<QHIP_EXPR_IS_NULL>;          // is_null.rs
using IsNotNull = UnaryExpr<QHIP_EXPR_IS_NOT_NULL>;   // is_not_null.rs
using Negative = UnaryExpr<QHIP_EXPR_NEGATIVE>;       // negative.rs
struct CaseExpr : PhysicalExpr {   // case.rs:13-48: nested zip(mask, truthy, falsy), folded from the ELSE branch up
  std::vector<std::pair<ExprRef, ExprRef>> when_then; ExprRef else_expr;
  CaseExpr(std::vector<std::pair<ExprRef, ExprRef>> wt, ExprRef e) : when_then(std::move(wt)), else_expr(std::move(e)) {}
  int lower(ExprArray& o) const override {
    int acc = else_expr->lower(o);
    for (auto it = when_then.rbegin(); it != when_then.rend(); ++it) {
      qhip_expr e = ExprArray::node(QHIP_EXPR_IF);
      e.left = it->first->lower(o); e.right = it->second->lower(o); e.third = acc;
      acc = o.add(e);
    }
    return acc;
  }
  std::string to_string() const override { return "CASE ... END"; }
};
struct Like : PhysicalExpr {   // like.rs:14-43
  bool negated; ExprRef expr, pattern;
  Like(bool n, ExprRef e, ExprRef p) : negated(n), expr(std::move(e)), pattern(std::move(p)) {}
  int lower(ExprArray& o) const override {
    qhip_expr e = ExprArray::node(QHIP_EXPR_LIKE);
    e.left = expr->lower(o); e.right = pattern->lower(o); e.op = negated ? 1 : 0;
    return o.add(e);
  }
  std::string to_string() const override { return expr->to_string() + (negated ? " NOT LIKE " : " LIKE ") + pattern->to_string(); }
};

// physical/expr/aggregate/{sum,avg,count,min,max}.rs
struct AggregateExpr {
  int kind; ExprRef expr; qhip_dtype return_type;
  static AggregateExpr Sum(ExprRef e, qhip_dtype ret) { return {QHIP_AGG_SUM, std::move(e), ret}; }
  static AggregateExpr Avg(ExprRef e, qhip_dtype ret) { return {QHIP_AGG_AVG, std::move(e), ret}; }
  static AggregateExpr Count(ExprRef e) { return {QHIP_AGG_COUNT, std::move(e), dtype(QHIP_INT64)}; }
  static AggregateExpr Min(ExprRef e, qhip_dtype ret) { return {QHIP_AGG_MIN, std::move(e), ret}; }
  static AggregateExpr Max(ExprRef e, qhip_dtype ret) { return {QHIP_AGG_MAX, std::move(e), ret}; }
};

// ---------------------------------------------------------------- Arrow batches on the host
// What PhysicalPlan::execute returns: library-allocated Arrow arrays (one struct array per RecordBatch) + their schema,
// released through the C Data Interface callbacks.
struct RecordBatch {
  ArrowArray array{};
  ArrowSchema schema{};
  RecordBatch() = default;
  RecordBatch(RecordBatch&& o) noexcept : array(o.array), schema(o.schema) { o.array.release = nullptr; o.schema.release = nullptr; }
  RecordBatch& operator=(RecordBatch&& o) noexcept {
    if (this != &o) { reset(); array = o.array; schema = o.schema; o.array.release = nullptr; o.schema.release = nullptr; }
    return *this;
  }
  RecordBatch(const RecordBatch&) = delete;
  RecordBatch& operator=(const RecordBatch&) = delete;
  ~RecordBatch() { reset(); }
  void reset() {
    if (array.release) array.release(&array);
    if (schema.release) schema.release(&schema);
    array.release = nullptr; schema.release = nullptr;
  }
  int64_t num_rows() const { return array.length; }
  int64_t num_columns() const { return array.n_children; }
};

// ---------------------------------------------------------------- plans (physical/plan/*.rs)
struct PhysicalPlan {
  virtual ~PhysicalPlan() = default;
  // the batches stay in HBM between two accelerated operators
  virtual DeviceTableRef execute_device() const = 0;
  virtual std::vector<std::shared_ptr<const PhysicalPlan>> children() const { return {}; }
  // the context of the first table below this node (null when there is none)
  virtual ContextRef context() const {
    for (const auto& c : children()) if (ContextRef x = c->context()) return x;
    return nullptr;
  }
  // PhysicalPlan::execute (physical/plan/mod.rs:27): the whole result as host batches
  std::vector<RecordBatch> execute() const {
    DeviceTableRef t = execute_device();
    std::vector<RecordBatch> out((size_t)t->num_batches());
    for (int64_t b = 0; b < t->num_batches(); ++b)
      t->ctx()->check(qhip_table_to_arrow(t->ctx()->raw(), t->raw(), b, &out[(size_t)b].array, &out[(size_t)b].schema));
    return out;
  }
};
using PlanRef = std::shared_ptr<const PhysicalPlan>;

// datasource/memory.rs:20-98: batches uploaded once, kept in HBM (takes ownership of the Arrow structs it is given)
class MemoryTable {
 public:
  MemoryTable(ContextRef ctx, ArrowSchema* schema, std::vector<ArrowArray*> batches) : ctx_(std::move(ctx)) {
    qhip_table* t = nullptr;
    const int rc = qhip_table_from_arrow(ctx_->raw(), schema, batches.empty() ? nullptr : batches.data(), (int64_t)batches.size(), &t);
    for (ArrowArray* b : batches) if (b->release) b->release(b);
    if (schema->release) schema->release(schema);
    ctx_->check(rc);
    table_ = std::make_shared<DeviceTable>(ctx_, t);
  }
  const DeviceTableRef& device_table() const { return table_; }
  const ContextRef& ctx() const { return ctx_; }

 private:
  ContextRef ctx_;
  DeviceTableRef table_;
};
using MemoryTableRef = std::shared_ptr<const MemoryTable>;

namespace detail {
inline DeviceTableRef wrap(const ContextRef& ctx, int rc, qhip_table* t) {
  ctx->check(rc);
  return std::make_shared<DeviceTable>(ctx, t);
}
// Deferred sizing (qhip.h: qhip_ctx_allow_deferred_sizes). feed(): execute `node` as the input of an operator that reads a
// device-side row count (HashAggregate's input, a hash join's build side); the consumer runs right after, inside retrying().
inline DeviceTableRef feed(const PhysicalPlan& node) {
  struct Allow {
    ContextRef c;
    explicit Allow(ContextRef x) : c(std::move(x)) { if (c) qhip_ctx_allow_deferred_sizes(c->raw(), +1); }
    ~Allow() { if (c) qhip_ctx_allow_deferred_sizes(c->raw(), -1); }
  } allow(node.context());
  return node.execute_device();
}
// run() = execute the input(s), then the operator. QHIP_RETRY: a join of deferred size below had too little room and has
// forgotten its hint — the second run waits for the size.
template <class F> DeviceTableRef retrying(F&& run) {
  for (int attempt = 0;; ++attempt) {
    try { return run(); }
    catch (const Error& e) { if (e.code != QHIP_RETRY || attempt == 2) throw; }
  }
}
inline DeviceTableRef filter(const DeviceTableRef& in, const ExprRef& predicate) {
  ExprArray ea;
  const int root = predicate ? predicate->lower(ea) : -1;
  qhip_table* out = nullptr;
  const int rc = qhip_filter_execute(in->ctx()->raw(), in->raw(), ea.data(), ea.size(), root, nullptr, -1, &out);
  return wrap(in->ctx(), rc, out);
}
}  // namespace detail

struct Scan : PhysicalPlan {   // scan.rs + MemoryTable::scan (memory.rs:69-98): optional pushed-down filter
  MemoryTableRef datasource; ExprRef filter;
  Scan(MemoryTableRef d, ExprRef f = nullptr) : datasource(std::move(d)), filter(std::move(f)) {}
  DeviceTableRef execute_device() const override {
    return filter ? detail::filter(datasource->device_table(), filter) : datasource->device_table();
  }
  ContextRef context() const override { return datasource->ctx(); }
};
struct Filter : PhysicalPlan {   // filter.rs:12-48
  PlanRef input; ExprRef predicate;
  Filter(PlanRef i, ExprRef p) : input(std::move(i)), predicate(std::move(p)) {}
  DeviceTableRef execute_device() const override { return detail::filter(input->execute_device(), predicate); }
  std::vector<PlanRef> children() const override { return {input}; }
};
struct Projection : PhysicalPlan {   // projection.rs:10-51
  PlanRef input; std::vector<ExprRef> exprs; std::vector<std::string> names;
  Projection(std::vector<std::string> n, PlanRef i, std::vector<ExprRef> e) : input(std::move(i)), exprs(std::move(e)), names(std::move(n)) {}
  DeviceTableRef execute_device() const override {
    DeviceTableRef in = input->execute_device();
    ExprArray ea;
    std::vector<int32_t> roots;
    for (auto& e : exprs) roots.push_back(e->lower(ea));
    std::vector<const char*> cn;
    for (size_t k = 0; k < exprs.size(); ++k) cn.push_back(k < names.size() ? names[k].c_str() : nullptr);
    qhip_table* out = nullptr;
    const int rc = qhip_projection_execute(in->ctx()->raw(), in->raw(), ea.data(), ea.size(), roots.data(), (int)roots.size(), cn.data(), &out);
    return detail::wrap(in->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return {input}; }
};
struct HashAggregate : PhysicalPlan {   // aggregate/hash.rs:110-176; a Scan(filter) input is fused into the kernel
  PlanRef input; std::vector<ExprRef> group_exprs; std::vector<AggregateExpr> aggregate_exprs; std::vector<std::string> names;
  HashAggregate(std::vector<std::string> n, PlanRef i, std::vector<ExprRef> g, std::vector<AggregateExpr> a)
      : input(std::move(i)), group_exprs(std::move(g)), aggregate_exprs(std::move(a)), names(std::move(n)) {}
  DeviceTableRef execute_device() const override { return detail::retrying([&] { return execute_once(); }); }
  DeviceTableRef execute_once() const {
    DeviceTableRef in;
    ExprRef pred;
    if (auto scan = dynamic_cast<const Scan*>(input.get()); scan && scan->filter) { in = scan->datasource->device_table(); pred = scan->filter; }
    else in = detail::feed(*input);
    ExprArray ea;
    const int proot = pred ? pred->lower(ea) : -1;
    std::vector<int32_t> groups;
    for (auto& g : group_exprs) groups.push_back(g->lower(ea));
    std::vector<qhip_agg> aggs;
    for (auto& a : aggregate_exprs) aggs.push_back(qhip_agg{a.kind, a.expr->lower(ea), a.return_type});
    std::vector<const char*> cn;
    for (size_t k = 0; k < groups.size() + aggs.size(); ++k) cn.push_back(k < names.size() ? names[k].c_str() : nullptr);
    qhip_table* out = nullptr;
    const int rc = qhip_hash_aggregate_execute(in->ctx()->raw(), in->raw(), ea.data(), ea.size(), proot, groups.data(), (int)groups.size(),
                                               aggs.data(), (int)aggs.size(), cn.data(), &out);
    return detail::wrap(in->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return {input}; }
};
struct NoGroupingAggregate : HashAggregate {   // aggregate/no_grouping.rs:9-66
  NoGroupingAggregate(std::vector<std::string> n, PlanRef i, std::vector<AggregateExpr> a) : HashAggregate(std::move(n), std::move(i), {}, std::move(a)) {}
  std::vector<PlanRef> children() const override { return {}; }   // no_grouping.rs:63-65
};

struct JoinFilter {   // join/mod.rs JoinFilter { expr, schema, column_indices }
  ExprRef expr;
  std::vector<std::pair<int, JoinSide>> column_indices;
};
struct HashJoinExec : PhysicalPlan {   // join/hash_join.rs:110-384 — build = left, probe = right
  PlanRef left, right; JoinType join_type; std::vector<std::pair<ExprRef, ExprRef>> on; std::optional<JoinFilter> filter;
  static std::shared_ptr<HashJoinExec> try_new(PlanRef l, PlanRef r, JoinType jt, std::vector<std::pair<ExprRef, ExprRef>> on,
                                               std::optional<JoinFilter> f = std::nullopt) {
    if (on.empty()) throw Error(QHIP_INVALID_ARGUMENT, "Internal error: On constraints in HashJoinExec should be non-empty");   // hash_join.rs:129-131
    auto j = std::make_shared<HashJoinExec>();
    j->left = std::move(l); j->right = std::move(r); j->join_type = jt; j->on = std::move(on); j->filter = std::move(f);
    return j;
  }
  DeviceTableRef execute_device() const override { return detail::retrying([&] { return execute_once(); }); }
  DeviceTableRef execute_once() const {
    // an Inner join takes Scan(filter) children as (unfiltered table, predicate): the filter is fused into the key kernels
    auto side = [&](const PlanRef& p, ExprRef& pred) -> DeviceTableRef {
      if (auto scan = dynamic_cast<const Scan*>(p.get()); scan && scan->filter && join_type == QHIP_JOIN_INNER) {
        pred = scan->filter;
        return scan->datasource->device_table();
      }
      return p->execute_device();
    };
    ExprRef lpred, rpred;
    // the build side may arrive with a device-side row count when nothing executes between it and this join: the probe
    // side is a table access (a Scan: fused, or without a filter)
    const Scan* rscan = dynamic_cast<const Scan*>(right.get());
    const bool right_is_table = rscan && (!rscan->filter || join_type == QHIP_JOIN_INNER);
    DeviceTableRef lt = right_is_table && !dynamic_cast<const Scan*>(left.get()) ? detail::feed(*left) : side(left, lpred);
    DeviceTableRef rt = side(right, rpred);
    ExprArray le, re, fe;
    std::vector<int32_t> on_l, on_r, fsides, fcols;
    for (auto& kv : on) { on_l.push_back(kv.first->lower(le)); on_r.push_back(kv.second->lower(re)); }
    const int lp = lpred ? lpred->lower(le) : -1, rp = rpred ? rpred->lower(re) : -1;
    int froot = -1;
    if (filter) {
      froot = filter->expr->lower(fe);
      for (auto& ci : filter->column_indices) { fcols.push_back(ci.first); fsides.push_back((int)ci.second); }
    }
    qhip_table* out = nullptr;
    const int rc = qhip_hash_join_execute(lt->ctx()->raw(), lt->raw(), rt->raw(), (int)join_type, le.data(), le.size(), re.data(), re.size(),
                                          on_l.data(), on_r.data(), (int)on.size(), fe.data(), fe.size(), froot, fsides.data(), fcols.data(),
                                          (int)fcols.size(), lp, rp, &out);
    return detail::wrap(lt->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return {left, right}; }
};
struct NestedLoopJoinExec : PhysicalPlan {   // join/nest_loop_join.rs:42-228
  PlanRef left, right; JoinType join_type; std::optional<JoinFilter> filter;
  NestedLoopJoinExec(PlanRef l, PlanRef r, JoinType jt, std::optional<JoinFilter> f = std::nullopt)
      : left(std::move(l)), right(std::move(r)), join_type(jt), filter(std::move(f)) {}
  DeviceTableRef execute_device() const override {
    DeviceTableRef lt = left->execute_device(), rt = right->execute_device();
    ExprArray fe;
    std::vector<int32_t> fsides, fcols;
    int froot = -1;
    if (filter) {
      froot = filter->expr->lower(fe);
      for (auto& ci : filter->column_indices) { fcols.push_back(ci.first); fsides.push_back((int)ci.second); }
    }
    qhip_table* out = nullptr;
    const int rc = qhip_nested_loop_join_execute(lt->ctx()->raw(), lt->raw(), rt->raw(), (int)join_type, fe.data(), fe.size(), froot,
                                                 fsides.data(), fcols.data(), (int)fcols.size(), &out);
    return detail::wrap(lt->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return {left, right}; }
};
struct CrossJoin : PhysicalPlan {   // join/cross_join.rs:56-170
  PlanRef left, right;
  CrossJoin(PlanRef l, PlanRef r) : left(std::move(l)), right(std::move(r)) {}
  DeviceTableRef execute_device() const override {
    DeviceTableRef lt = left->execute_device(), rt = right->execute_device();
    qhip_table* out = nullptr;
    const int rc = qhip_cross_join_execute(lt->ctx()->raw(), lt->raw(), rt->raw(), &out);
    return detail::wrap(lt->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return {left, right}; }
};

struct SortOptions { bool descending = false; bool nulls_first = true; };   // arrow::compute::SortOptions
struct PhysicalSortExpr { ExprRef expr; SortOptions options; };             // sort.rs:12-21
struct Sort : PhysicalPlan {   // sort.rs:23-86
  std::vector<PhysicalSortExpr> exprs; PlanRef input; std::optional<int64_t> limit;
  Sort(std::vector<PhysicalSortExpr> e, PlanRef i, std::optional<int64_t> l = std::nullopt) : exprs(std::move(e)), input(std::move(i)), limit(l) {}
  DeviceTableRef execute_device() const override {
    DeviceTableRef in = input->execute_device();
    ExprArray ea;
    std::vector<int32_t> roots, desc, nf;
    for (auto& e : exprs) { roots.push_back(e.expr->lower(ea)); desc.push_back(e.options.descending); nf.push_back(e.options.nulls_first); }
    qhip_table* out = nullptr;
    const int rc = qhip_sort_execute(in->ctx()->raw(), in->raw(), ea.data(), ea.size(), roots.data(), desc.data(), nf.data(), (int)roots.size(),
                                     limit ? *limit : -1, &out);
    return detail::wrap(in->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return input->children(); }   // sort.rs:83-85
};
struct Limit : PhysicalPlan {   // limit.rs:10-62
  PlanRef input; std::optional<int64_t> fetch; int64_t skip;
  Limit(PlanRef i, std::optional<int64_t> f, int64_t s) : input(std::move(i)), fetch(f), skip(s) {}
  DeviceTableRef execute_device() const override {
    DeviceTableRef in = input->execute_device();
    qhip_table* out = nullptr;
    const int rc = qhip_limit_execute(in->ctx()->raw(), in->raw(), skip, fetch ? *fetch : -1, &out);
    return detail::wrap(in->ctx(), rc, out);
  }
  std::vector<PlanRef> children() const override { return input->children(); }   // limit.rs:59-61
};

}  // namespace qurious_hip
