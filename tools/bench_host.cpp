// bench_host.cpp — the metric's step (TPC-H Q1 at SF10 + Q3 at SF10, BASELINE configs[2] + configs[3]) driven by a COMPILED
// host: the C++ mirror of the reference's operator API (include/qhip_plan.hpp: MemoryTable, Scan, HashJoinExec, HashAggregate,
// the expression nodes — same names, arguments and fusion rules as the Rust nodes they stand in for) over libqhip's C ABI.
// bench.py times the same step through the Python mirror; this binary shows what the host language costs.
// The plans are the ones of qurious_amd/queries.py (q1_full, q3), the tables the counter-based synthetic ones of SURVEY §8d
// (qhip_synth_*), uploaded once; K timed steps between two stream synchronisations after W warm-up steps.
//   make -C tools bench_host && tools/bench_host [steps] [warmup] [sf]      (needs an MI355X)
// Prints one JSON line: ms_per_step, rows/s, and the two queries' group counts (checked by tests/test_gpu_q3.py).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../include/qhip_plan.hpp"
#include "../include/qhip_bench.h"   // the synthetic tables (benchmark support: libqhip_bench.so)

using namespace qurious_hip;

// ---------------------------------------------------------------- one-batch Arrow tables over buffers we own
struct Col {
  std::string name, format;                 // Arrow C format string: "l", "tdD", "u", "d:15,2"
  std::vector<std::vector<uint8_t>> bufs;   // [values] or [offsets, data] (validity absent: the synthetic tables have no NULLs)
};
struct Holder {
  std::vector<Col> cols;
  std::vector<ArrowSchema> cschema;
  std::vector<ArrowSchema*> cschema_p;
  std::vector<ArrowArray> carr;
  std::vector<ArrowArray*> carr_p;
  std::vector<std::vector<const void*>> cbufs;
  const void* top[1] = {nullptr};
};
static void noop_schema(ArrowSchema* s) { s->release = nullptr; }
static void noop_array(ArrowArray* a) { a->release = nullptr; }

static MemoryTableRef make_table(const ContextRef& ctx, std::vector<Col> cols, int64_t rows) {
  auto h = std::make_unique<Holder>();
  h->cols = std::move(cols);
  const size_t n = h->cols.size();
  h->cschema.resize(n); h->carr.resize(n); h->cbufs.resize(n);
  for (size_t c = 0; c < n; ++c) {
    ArrowSchema& s = h->cschema[c];
    memset(&s, 0, sizeof s);
    s.format = h->cols[c].format.c_str(); s.name = h->cols[c].name.c_str(); s.flags = 0; s.release = noop_schema;
    h->cschema_p.push_back(&s);
    ArrowArray& a = h->carr[c];
    memset(&a, 0, sizeof a);
    h->cbufs[c].push_back(nullptr);   // validity
    for (auto& b : h->cols[c].bufs) h->cbufs[c].push_back(b.data());
    a.length = rows; a.null_count = 0; a.n_buffers = (int64_t)h->cbufs[c].size(); a.buffers = h->cbufs[c].data(); a.release = noop_array;
    h->carr_p.push_back(&a);
  }
  ArrowSchema schema;
  memset(&schema, 0, sizeof schema);
  schema.format = "+s"; schema.name = ""; schema.n_children = (int64_t)n; schema.children = h->cschema_p.data(); schema.release = noop_schema;
  ArrowArray arr;
  memset(&arr, 0, sizeof arr);
  arr.length = rows; arr.n_buffers = 1; arr.buffers = h->top; arr.n_children = (int64_t)n; arr.children = h->carr_p.data(); arr.release = noop_array;
  return std::make_shared<MemoryTable>(ctx, &schema, std::vector<ArrowArray*>{&arr});   // uploads; the host buffers die with `h`
}
static std::vector<uint8_t> bytes(size_t n) { return std::vector<uint8_t>(n ? n : 1); }

// ---------------------------------------------------------------- an order-independent checksum of a WHOLE result
// sum over the rows of mix64-chained column values (fixed-width values as one or two 64-bit words, strings as their bytes packed
// into words + the length, NULL as a constant), plus the row count: tests/test_gpu_q3.py computes the same from the Python
// mirror's batches — every group, every column of both hosts' results must agree, not just the group counts.
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
static uint64_t result_checksum(const std::vector<RecordBatch>& batches, int64_t* rows_out) {
  uint64_t sum = 0;
  int64_t rows = 0;
  for (const RecordBatch& b : batches) {
    const int64_t n = b.array.length;
    rows += n;
    std::vector<uint64_t> h((size_t)n, 0);
    for (int64_t c = 0; c < b.array.n_children; ++c) {
      const ArrowArray& a = *b.array.children[c];
      const std::string fmt = b.schema.children[c]->format;
      const uint8_t* valid = a.n_buffers > 0 ? (const uint8_t*)a.buffers[0] : nullptr;
      for (int64_t i = 0; i < n; ++i) {
        const int64_t j = i + a.offset;
        uint64_t& x = h[(size_t)i];
        if (valid && a.null_count != 0 && !((valid[j >> 3] >> (j & 7)) & 1)) { x = mix64(x ^ 0x9E3779B97F4A7C15ULL); continue; }
        if (fmt == "u") {
          const int32_t* off = (const int32_t*)a.buffers[1];
          const uint8_t* data = (const uint8_t*)a.buffers[2];
          const int32_t len = off[j + 1] - off[j];
          for (int32_t k = 0; k < len; k += 8) {
            uint64_t w = 0;
            memcpy(&w, data + off[j] + k, (size_t)std::min<int32_t>(8, len - k));
            x = mix64(x ^ w);
          }
          x = mix64(x ^ (uint64_t)len);
        } else {
          const int width = fmt == "l" || fmt == "L" || fmt == "g" || fmt == "tdm" ? 8 : fmt.rfind("d:", 0) == 0 ? 16 : fmt == "i" || fmt == "I" || fmt == "f" || fmt == "tdD" ? 4 : 0;
          if (!width) throw Error(QHIP_UNSUPPORTED, "result_checksum: column format " + fmt);
          const uint8_t* v = (const uint8_t*)a.buffers[1] + (size_t)j * (size_t)width;
          uint64_t w0 = 0, w1 = 0;
          memcpy(&w0, v, (size_t)std::min(width, 8));
          if (width == 16) memcpy(&w1, v + 8, 8);
          x = mix64(x ^ w0);
          if (width == 16) x = mix64(x ^ w1);
        }
      }
    }
    for (uint64_t x : h) sum += mix64(x);
  }
  if (rows_out) *rows_out = rows;
  return sum;
}

static ExprRef col(const char* name, int i) { return std::make_shared<Column>(name, i); }
static ExprRef date(const char* s) { return std::make_shared<CastExpr>(std::make_shared<Literal>(ScalarValue::Utf8(s)), dtype(QHIP_DATE32)); }
static ExprRef bin(ExprRef l, Operator op, ExprRef r) { return std::make_shared<BinaryExpr>(std::move(l), op, std::move(r)); }

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 20, warmup = argc > 2 ? atoi(argv[2]) : 6;
  const double sf = argc > 3 ? atof(argv[3]) : 10.0;
  try {
    auto ctx = std::make_shared<Context>();
    // ---- tables (SURVEY §8d recipes through libqhip's own generators)
    const int64_t n_li = (int64_t)(5998605.2 * sf + 0.5);   // SF10: 59 986 052 lineitem rows (Q1's table)
    Col shipdate{"l_shipdate", "tdD", {bytes((size_t)n_li * 4)}}, rf{"l_returnflag", "u", {bytes(((size_t)n_li + 1) * 4), bytes((size_t)n_li)}},
        ls{"l_linestatus", "u", {bytes(((size_t)n_li + 1) * 4), bytes((size_t)n_li)}}, qty{"l_quantity", "d:15,2", {bytes((size_t)n_li * 16)}},
        price{"l_extendedprice", "d:15,2", {bytes((size_t)n_li * 16)}}, disc{"l_discount", "d:15,2", {bytes((size_t)n_li * 16)}},
        tax{"l_tax", "d:15,2", {bytes((size_t)n_li * 16)}};
    if (qhip_synth_lineitem(0, n_li, (int32_t*)shipdate.bufs[0].data(), (int32_t*)rf.bufs[0].data(), rf.bufs[1].data(), (int32_t*)ls.bufs[0].data(),
                            ls.bufs[1].data(), qty.bufs[0].data(), price.bufs[0].data(), disc.bufs[0].data(), tax.bufs[0].data()) != QHIP_OK)
      throw Error(QHIP_INVALID_ARGUMENT, "qhip_synth_lineitem failed");
    std::vector<Col> li_cols;
    li_cols.push_back(std::move(shipdate)); li_cols.push_back(std::move(rf)); li_cols.push_back(std::move(ls)); li_cols.push_back(std::move(qty));
    li_cols.push_back(std::move(price)); li_cols.push_back(std::move(disc)); li_cols.push_back(std::move(tax));
    MemoryTableRef lineitem = make_table(ctx, std::move(li_cols), n_li);

    const int64_t n_c = (int64_t)(150000 * sf), n_o = (int64_t)(1500000 * sf);
    Col ckey{"c_custkey", "l", {bytes((size_t)n_c * 8)}}, cseg{"c_mktsegment", "u", {bytes(((size_t)n_c + 1) * 4), bytes((size_t)n_c * 10)}};
    if (qhip_synth_customer(1, n_c, (int64_t*)ckey.bufs[0].data(), (int32_t*)cseg.bufs[0].data(), cseg.bufs[1].data()) != QHIP_OK)
      throw Error(QHIP_INVALID_ARGUMENT, "qhip_synth_customer failed");
    std::vector<Col> c_cols; c_cols.push_back(std::move(ckey)); c_cols.push_back(std::move(cseg));
    MemoryTableRef customer = make_table(ctx, std::move(c_cols), n_c);
    Col okey{"o_orderkey", "l", {bytes((size_t)n_o * 8)}}, ocust{"o_custkey", "l", {bytes((size_t)n_o * 8)}}, odate{"o_orderdate", "tdD", {bytes((size_t)n_o * 4)}},
        oprio{"o_shippriority", "l", {bytes((size_t)n_o * 8)}};
    if (qhip_synth_orders(1, n_o, n_c, (int64_t*)okey.bufs[0].data(), (int64_t*)ocust.bufs[0].data(), (int32_t*)odate.bufs[0].data(),
                          (int64_t*)oprio.bufs[0].data()) != QHIP_OK)
      throw Error(QHIP_INVALID_ARGUMENT, "qhip_synth_orders failed");
    std::vector<Col> o_cols; o_cols.push_back(std::move(okey)); o_cols.push_back(std::move(ocust)); o_cols.push_back(std::move(odate)); o_cols.push_back(std::move(oprio));
    MemoryTableRef orders = make_table(ctx, std::move(o_cols), n_o);
    const int64_t n_l3 = qhip_synth_q3_lineitem_count(1, n_o);
    Col lkey{"l_orderkey", "l", {bytes((size_t)n_l3 * 8)}}, lship{"l_shipdate", "tdD", {bytes((size_t)n_l3 * 4)}},
        lprice{"l_extendedprice", "d:15,2", {bytes((size_t)n_l3 * 16)}}, ldisc{"l_discount", "d:15,2", {bytes((size_t)n_l3 * 16)}};
    if (qhip_synth_q3_lineitem(1, n_o, (int64_t*)lkey.bufs[0].data(), (int32_t*)lship.bufs[0].data(), lprice.bufs[0].data(), ldisc.bufs[0].data()) != QHIP_OK)
      throw Error(QHIP_INVALID_ARGUMENT, "qhip_synth_q3_lineitem failed");
    std::vector<Col> l3_cols; l3_cols.push_back(std::move(lkey)); l3_cols.push_back(std::move(lship)); l3_cols.push_back(std::move(lprice)); l3_cols.push_back(std::move(ldisc));
    MemoryTableRef lineitem3 = make_table(ctx, std::move(l3_cols), n_l3);

    // ---- plans (qurious_amd/queries.py: q1_full, q3 — the shapes the reference's planner builds)
    const qhip_dtype DEC = dtype(QHIP_DECIMAL128, 15, 2), T4 = dtype(QHIP_DECIMAL128, 38, 4), T6 = dtype(QHIP_DECIMAL128, 38, 6),
                     AVG_T = dtype(QHIP_DECIMAL128, 19, 6);
    auto one = [] { return std::make_shared<CastExpr>(std::make_shared<Literal>(ScalarValue::Int64(1)), dtype(QHIP_DECIMAL128, 20, 0)); };
    PlanRef q1;
    {
      auto scan = std::make_shared<Scan>(lineitem, bin(col("l_shipdate", 0), QHIP_OP_LTEQ, date("1998-09-02")));
      ExprRef qty_ = col("l_quantity", 3), price_ = col("l_extendedprice", 4), disc_ = col("l_discount", 5), tax_ = col("l_tax", 6);
      ExprRef disc_price = bin(price_, QHIP_OP_MUL, bin(one(), QHIP_OP_SUB, disc_));
      ExprRef charge = bin(disc_price, QHIP_OP_MUL, bin(one(), QHIP_OP_ADD, tax_));
      std::vector<AggregateExpr> aggs = {AggregateExpr::Sum(qty_, DEC), AggregateExpr::Sum(price_, DEC), AggregateExpr::Sum(disc_price, T4),
                                         AggregateExpr::Sum(charge, T6), AggregateExpr::Avg(qty_, AVG_T), AggregateExpr::Avg(price_, AVG_T),
                                         AggregateExpr::Avg(disc_, AVG_T), AggregateExpr::Count(std::make_shared<Literal>(ScalarValue::Int64(1)))};
      q1 = std::make_shared<HashAggregate>(std::vector<std::string>{"l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge",
                                                                   "avg_qty", "avg_price", "avg_disc", "count_order"},
                                           scan, std::vector<ExprRef>{col("l_returnflag", 1), col("l_linestatus", 2)}, aggs);
    }
    PlanRef q3;
    {
      auto c_scan = std::make_shared<Scan>(customer, bin(col("c_mktsegment", 1), QHIP_OP_EQ, std::make_shared<Literal>(ScalarValue::Utf8("BUILDING"))));
      auto o_scan = std::make_shared<Scan>(orders, bin(col("o_orderdate", 2), QHIP_OP_LT, date("1995-03-15")));
      auto l_scan = std::make_shared<Scan>(lineitem3, bin(col("l_shipdate", 1), QHIP_OP_GT, date("1995-03-15")));
      auto j1 = HashJoinExec::try_new(c_scan, o_scan, QHIP_JOIN_INNER, {{col("c_custkey", 0), col("o_custkey", 1)}});
      auto j2 = HashJoinExec::try_new(j1, l_scan, QHIP_JOIN_INNER, {{col("o_orderkey", 2), col("l_orderkey", 0)}});
      ExprRef revenue = bin(col("l_extendedprice", 8), QHIP_OP_MUL, bin(one(), QHIP_OP_SUB, col("l_discount", 9)));
      q3 = std::make_shared<HashAggregate>(std::vector<std::string>{"l_orderkey", "o_orderdate", "o_shippriority", "revenue"}, j2,
                                           std::vector<ExprRef>{col("l_orderkey", 6), col("o_orderdate", 4), col("o_shippriority", 5)},
                                           std::vector<AggregateExpr>{AggregateExpr::Sum(revenue, T4)});
    }
    // ---- warm up (plans learn join sizes / group counts, columns get their narrow copies), then K timed steps
    int64_t g1 = 0, g3 = 0;
    for (int k = 0; k < warmup; ++k) { g1 = q1->execute_device()->num_rows(); g3 = q3->execute_device()->num_rows(); }
    ctx->check(qhip_ctx_synchronize(ctx->raw()));
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) { (void)q1->execute_device(); (void)q3->execute_device(); }
    ctx->check(qhip_ctx_synchronize(ctx->raw()));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // the two queries once more through execute(): host Arrow batches, every row of both results folded into a checksum
    int64_t r1 = 0, r3 = 0;
    const uint64_t c1 = result_checksum(q1->execute(), &r1), c3 = result_checksum(q3->execute(), &r3);
    printf("{\"host\": \"C++ mirror (include/qhip_plan.hpp)\", \"steps\": %d, \"warmup\": %d, \"sf\": %g, \"ms_per_step\": %.4f, \"rows_per_s\": %.4g, "
           "\"q1_rows\": %lld, \"q1_groups\": %lld, \"q3_lineitem_rows\": %lld, \"q3_groups\": %lld, "
           "\"q1_result_rows\": %lld, \"q1_result_checksum\": \"%llu\", \"q3_result_rows\": %lld, \"q3_result_checksum\": \"%llu\"}\n",
           steps, warmup, sf, secs / steps * 1e3, (double)(n_li + n_l3) * steps / secs, (long long)n_li, (long long)g1, (long long)n_l3, (long long)g3,
           (long long)r1, (unsigned long long)c1, (long long)r3, (unsigned long long)c3);
    return 0;
  } catch (const Error& e) {
    fprintf(stderr, "bench_host: error %d: %s\n", e.code, e.what());
    return 1;
  }
}
