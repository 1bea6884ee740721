// expr.cpp — typing of the expression tree + host-side literal folding.
#include "expr.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <limits>

namespace qhip {

i128 pow10_i128(int e) {
  i128 r = 1;
  for (int k = 0; k < e; ++k) r *= 10;
  return r;
}

static const char* op_symbol(int op) {
  static const char* s[] = {"==", "!=", ">", ">=", "<", "<=", "AND", "OR", "+", "-", "*", "/", "%"};
  return (op >= 0 && op <= QHIP_OP_MOD) ? s[op] : "?";
}
static bool is_cmp(int op) { return op >= QHIP_OP_EQ && op <= QHIP_OP_LTEQ; }
static bool is_logic(int op) { return op == QHIP_OP_AND || op == QHIP_OP_OR; }

static bool leap(int y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }
int32_t parse_date32(const std::string& s) {
  int y = 0, m = 0, d = 0;
  char tail = 0;
  if (sscanf(s.c_str(), "%d-%d-%d%c", &y, &m, &d, &tail) != 3 || m < 1 || m > 12 || d < 1)
    fail(QHIP_EXEC_ERROR, "Cast error: Cannot cast string '" + s + "' to value of Date32 type");
  static const int mdays[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  int dim = mdays[m - 1] + ((m == 2 && leap(y)) ? 1 : 0);
  if (d > dim) fail(QHIP_EXEC_ERROR, "Cast error: Cannot cast string '" + s + "' to value of Date32 type");
  // days from civil (Howard Hinnant's algorithm)
  int yy = y - (m <= 2);
  const int era = (yy >= 0 ? yy : yy - 399) / 400;
  const unsigned yoe = (unsigned)(yy - era * 400);
  const unsigned doy = (153 * (unsigned)(m + (m > 2 ? -3 : 9)) + 2) / 5 + (unsigned)d - 1;
  const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return (int32_t)(era * 146097 + (int)doe - 719468);
}

static i128 lit_i128(const ENode& n) { return (i128)(((u128)(uint64_t)n.hi << 64) | (u128)n.lo); }
static void set_i128(ENode& n, i128 v) { n.lo = (uint64_t)(u128)v; n.hi = (int64_t)((u128)v >> 64); }

static bool int_range(const DType& t, i128& lo, i128& hi) {
  switch (t.id) {
    case QHIP_INT8: lo = -128; hi = 127; return true;
    case QHIP_INT16: lo = -32768; hi = 32767; return true;
    case QHIP_INT32: case QHIP_DATE32: lo = INT32_MIN; hi = INT32_MAX; return true;
    case QHIP_INT64: case QHIP_DATE64: lo = INT64_MIN; hi = INT64_MAX; return true;
    case QHIP_UINT8: lo = 0; hi = 255; return true;
    case QHIP_UINT16: lo = 0; hi = 65535; return true;
    case QHIP_UINT32: lo = 0; hi = UINT32_MAX; return true;
    case QHIP_UINT64: lo = 0; hi = (i128)UINT64_MAX; return true;
    default: return false;
  }
}
static bool is_intlike(const DType& t) { i128 a, b; return int_range(t, a, b); }

static std::string i128_str(i128 v) {
  if (v == 0) return "0";
  bool neg = v < 0;
  u128 u = neg ? (u128)0 - (u128)v : (u128)v;
  std::string s;
  while (u) { s.insert(s.begin(), (char)('0' + (int)(u % 10))); u /= 10; }
  return neg ? "-" + s : s;
}

void fold_literal_cast(const ENode& src, const DType& to, ENode& out) {
  out = ENode();
  out.kind = QHIP_EXPR_LITERAL;
  out.type = to;
  out.lit_null = src.lit_null;
  out.nullable = src.lit_null;
  if (src.lit_null) return;
  const DType& from = src.type;
  auto overflow = [&](const std::string& v) {
    fail(QHIP_EXEC_ERROR, "Cast error: Can't cast value " + v + " to type " + dtype_name(to));
  };
  if (from == to) { out = src; out.kind = QHIP_EXPR_LITERAL; return; }
  // source value classes
  if (from.id == QHIP_UTF8) {
    if (to.id == QHIP_DATE32) { out.lo = (uint64_t)(int64_t)parse_date32(src.s); out.hi = -(int64_t)(out.lo >> 63); return; }
    if (is_intlike(to)) {
      char* end = nullptr;
      long long v = strtoll(src.s.c_str(), &end, 10);
      if (!end || *end || src.s.empty()) fail(QHIP_EXEC_ERROR, "Cast error: Cannot cast string '" + src.s + "' to value of " + dtype_name(to) + " type");
      i128 lo, hi; int_range(to, lo, hi);
      if (v < lo || v > hi) overflow(src.s);
      out.lo = (uint64_t)v; out.hi = v < 0 ? -1 : 0; return;
    }
    if (dtype_is_float(to)) {
      char* end = nullptr;
      double v = strtod(src.s.c_str(), &end);
      if (!end || *end || src.s.empty()) fail(QHIP_EXEC_ERROR, "Cast error: Cannot cast string '" + src.s + "' to value of " + dtype_name(to) + " type");
      out.f = to.id == QHIP_FLOAT32 ? (double)(float)v : v; return;
    }
    fail(QHIP_UNSUPPORTED, "literal cast Utf8 -> " + dtype_name(to));
  }
  if (from.id == QHIP_BOOL) {
    if (is_intlike(to)) { out.lo = src.lo & 1; return; }
    fail(QHIP_UNSUPPORTED, "literal cast Boolean -> " + dtype_name(to));
  }
  if (is_intlike(from)) {
    i128 v = (from.id == QHIP_UINT64) ? (i128)(u128)src.lo : (i128)(int64_t)src.lo;
    if (is_intlike(to)) {
      i128 lo, hi; int_range(to, lo, hi);
      if (v < lo || v > hi) overflow(i128_str(v));
      out.lo = (uint64_t)(u128)v; out.hi = v < 0 ? -1 : 0; return;
    }
    if (dtype_is_float(to)) { out.f = to.id == QHIP_FLOAT32 ? (double)(float)(double)v : (double)v; return; }
    if (to.id == QHIP_DECIMAL128) {
      if (to.scale < 0) fail(QHIP_UNSUPPORTED, "negative decimal scale");
      i128 r = v * pow10_i128(to.scale);
      i128 lim = pow10_i128(to.precision);
      if (r >= lim || r <= -lim) overflow(i128_str(v));
      set_i128(out, r); return;
    }
    if (to.id == QHIP_BOOL) { out.lo = v != 0; return; }
    fail(QHIP_UNSUPPORTED, "literal cast " + dtype_name(from) + " -> " + dtype_name(to));
  }
  if (dtype_is_float(from)) {
    double v = src.f;
    if (dtype_is_float(to)) { out.f = to.id == QHIP_FLOAT32 ? (double)(float)v : v; return; }
    if (is_intlike(to)) {
      i128 lo, hi; int_range(to, lo, hi);
      double tv = std::trunc(v);
      if (!(tv >= (double)lo && tv <= (double)hi)) overflow(std::to_string(v));
      i128 r = (i128)tv;
      out.lo = (uint64_t)(u128)r; out.hi = r < 0 ? -1 : 0; return;
    }
    if (to.id == QHIP_DECIMAL128) {
      double scaled = std::round(v * std::pow(10.0, to.scale));
      if (!(std::fabs(scaled) < 1.7e38)) overflow(std::to_string(v));
      i128 r = (i128)scaled;
      i128 lim = pow10_i128(to.precision);
      if (r >= lim || r <= -lim) overflow(std::to_string(v));
      set_i128(out, r); return;
    }
    fail(QHIP_UNSUPPORTED, "literal cast " + dtype_name(from) + " -> " + dtype_name(to));
  }
  if (from.id == QHIP_DECIMAL128) {
    i128 v = lit_i128(src);
    if (to.id == QHIP_DECIMAL128) {
      i128 r;
      if (to.scale >= from.scale) r = v * pow10_i128(to.scale - from.scale);
      else {
        i128 d = pow10_i128(from.scale - to.scale), q = v / d, rem = v % d, half = d / 2;
        if (rem >= half + (d & 1 ? 1 : 0) || (rem >= half && !(d & 1))) q += 1;
        else if (-rem >= half + (d & 1 ? 1 : 0) || (-rem >= half && !(d & 1))) q -= 1;
        r = q;
      }
      i128 lim = pow10_i128(to.precision);
      if (r >= lim || r <= -lim) overflow(i128_str(v));
      set_i128(out, r); return;
    }
    if (dtype_is_float(to)) { out.f = (double)v / std::pow(10.0, from.scale); if (to.id == QHIP_FLOAT32) out.f = (double)(float)out.f; return; }
    if (is_intlike(to)) {
      i128 q = v / pow10_i128(from.scale);
      i128 lo, hi; int_range(to, lo, hi);
      if (q < lo || q > hi) overflow(i128_str(v));
      out.lo = (uint64_t)(u128)q; out.hi = q < 0 ? -1 : 0; return;
    }
  }
  fail(QHIP_UNSUPPORTED, "literal cast " + dtype_name(from) + " -> " + dtype_name(to));
}

std::string canonical_like_pattern(const std::string& p) {
  std::string out;
  for (size_t k = 0; k < p.size(); ++k) {
    const unsigned char c = (unsigned char)p[k];
    if (c == '\\' && k + 1 < p.size()) out.push_back(p[++k]);   // escaped character stands for itself
    else if (c == '%') out.push_back((char)0xFF);
    else if (c == '_') out.push_back((char)0xFE);
    else out.push_back((char)c);
  }
  return out;
}

static bool device_cast_supported(const DType& from, const DType& to) {
  if (from == to) return true;
  auto numeric = [](const DType& t) { return is_intlike(t) || dtype_is_float(t) || t.id == QHIP_DECIMAL128; };
  if (from.id == QHIP_BOOL) return is_intlike(to) && to.id != QHIP_DATE32 && to.id != QHIP_DATE64;
  if (from.id == QHIP_UTF8 || to.id == QHIP_UTF8 || to.id == QHIP_BOOL || from.id == QHIP_NULL || to.id == QHIP_NULL) return false;
  return numeric(from) && numeric(to);
}

void ExprSet::build(const qhip_expr* ex, int n, const std::vector<InputCol>& input) {
  nodes.assign((size_t)n, ENode());
  std::vector<int> state((size_t)n, 0);
  // children may appear at any index: resolve recursively with cycle detection
  std::function<void(int)> visit = [&](int k) {
    if (k < 0 || k >= n) fail(QHIP_INVALID_ARGUMENT, "expression child index out of range");
    if (state[(size_t)k] == 2) return;
    if (state[(size_t)k] == 1) fail(QHIP_INVALID_ARGUMENT, "expression tree has a cycle");
    state[(size_t)k] = 1;
    const qhip_expr& e = ex[k];
    ENode nd;
    nd.kind = e.kind; nd.op = e.op; nd.column = e.column; nd.left = e.left; nd.right = e.right; nd.third = e.third;
    switch (e.kind) {
      case QHIP_EXPR_COLUMN: {
        if (e.column < 0 || e.column >= (int)input.size())
          fail(QHIP_INVALID_ARGUMENT, "PhysicalExpr Column references column at index " + std::to_string(e.column) +
                                          " (zero-based) but input schema only has " + std::to_string(input.size()) + " columns");
        nd.type = input[(size_t)e.column].type;
        nd.nullable = input[(size_t)e.column].has_nulls || nd.type.id == QHIP_NULL;
        nd.canon = "c" + std::to_string(e.column);
        break;
      }
      case QHIP_EXPR_LITERAL: {
        nd.type = DType(e.dtype);
        nd.lit_null = e.lit_is_null != 0 || nd.type.id == QHIP_NULL;
        nd.nullable = nd.lit_null;
        nd.lo = e.lit_lo; nd.hi = e.lit_hi; nd.f = e.lit_f64;
        if (nd.type.id == QHIP_UTF8 && !nd.lit_null) nd.s.assign(e.lit_str ? e.lit_str : "", (size_t)(e.lit_len > 0 ? e.lit_len : 0));
        if (nd.type.id == QHIP_FLOAT32) nd.f = (double)(float)nd.f;
        if (nd.type.id != QHIP_DECIMAL128 && nd.type.id != QHIP_UINT64) nd.hi = -(int64_t)(nd.lo >> 63);
        break;
      }
      case QHIP_EXPR_BINARY: {
        visit(e.left); visit(e.right);
        const ENode& l = nodes[(size_t)e.left];
        const ENode& r = nodes[(size_t)e.right];
        nd.nullable = l.nullable || r.nullable;
        const std::string where = dtype_name(l.type) + " " + op_symbol(e.op) + " " + dtype_name(r.type);
        if (is_cmp(e.op)) {
          if (l.type != r.type) fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid comparison operation: " + where);
          if (l.type.id == QHIP_NULL) fail(QHIP_UNSUPPORTED, "comparison of Null arrays");
          nd.type = DType(QHIP_BOOL);
        } else if (is_logic(e.op)) {
          if (l.type.id != QHIP_BOOL || r.type.id != QHIP_BOOL)
            fail(QHIP_INVALID_ARGUMENT, "boolean operator on non-boolean operands: " + where);
          nd.type = DType(QHIP_BOOL);
        } else if (e.op >= QHIP_OP_ADD && e.op <= QHIP_OP_MOD) {
          const bool ld = l.type.id == QHIP_DECIMAL128, rd = r.type.id == QHIP_DECIMAL128;
          if (ld || rd) {
            if (e.op == QHIP_OP_DIV) {
              // binary.rs:54-67 — decimal division runs in Float64
              if (!(ld || is_intlike(l.type) || dtype_is_float(l.type)) || !(rd || is_intlike(r.type) || dtype_is_float(r.type)))
                fail(QHIP_INVALID_ARGUMENT, "Invalid arithmetic operation: " + where);
              nd.type = DType(QHIP_FLOAT64);
            } else {
              if (!(ld && rd)) fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid arithmetic operation: " + where);
              const int p1 = l.type.precision, s1 = l.type.scale, p2 = r.type.precision, s2 = r.type.scale;
              if (e.op == QHIP_OP_ADD || e.op == QHIP_OP_SUB) {
                const int s = std::max(s1, s2);
                const int p = std::min(38, std::max(p1 - s1, p2 - s2) + s + 1);
                nd.type = DType(QHIP_DECIMAL128, p, s);
              } else if (e.op == QHIP_OP_MUL) {
                const int s = s1 + s2;
                if (s > 38) fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Output scale of " + where + " would exceed max scale of 38");
                nd.type = DType(QHIP_DECIMAL128, std::min(38, p1 + p2 + 1), s);
              } else {
                fail(QHIP_UNSUPPORTED, "decimal remainder is not accelerated: " + where);
              }
            }
          } else {
            if (l.type != r.type || !(dtype_is_integer(l.type) || dtype_is_float(l.type)))
              fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid arithmetic operation: " + where);
            nd.type = l.type;
          }
        } else {
          fail(QHIP_INVALID_ARGUMENT, "unknown binary operator " + std::to_string(e.op));
        }
        nd.canon = "b" + std::to_string(e.op) + "(" + l.canon + "," + r.canon + ")";
        break;
      }
      case QHIP_EXPR_CAST: {
        visit(e.left);
        const ENode& ch = nodes[(size_t)e.left];
        const DType to(e.dtype);
        if (ch.kind == QHIP_EXPR_LITERAL) {
          fold_literal_cast(ch, to, nd);   // evaluated once on the host, becomes a kernel scalar
        } else {
          if (!device_cast_supported(ch.type, to))
            fail(QHIP_UNSUPPORTED, "CAST(" + dtype_name(ch.type) + " AS " + dtype_name(to) + ") of a column is not accelerated");
          nd.type = to; nd.cast_to = to; nd.nullable = ch.nullable;
          nd.canon = "cast[" + dtype_name(to) + "](" + ch.canon + ")";
        }
        break;
      }
      case QHIP_EXPR_IS_NULL:
      case QHIP_EXPR_IS_NOT_NULL: {
        visit(e.left);
        nd.type = DType(QHIP_BOOL); nd.nullable = false;
        nd.canon = std::string(e.kind == QHIP_EXPR_IS_NULL ? "isnull(" : "notnull(") + nodes[(size_t)e.left].canon + ")";
        break;
      }
      case QHIP_EXPR_NEGATIVE: {
        visit(e.left);
        const ENode& ch = nodes[(size_t)e.left];
        if (!(dtype_is_signed(ch.type) || dtype_is_float(ch.type) || ch.type.id == QHIP_DECIMAL128))
          fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid arithmetic operation: !" + dtype_name(ch.type));
        nd.type = ch.type; nd.nullable = ch.nullable;
        nd.canon = "neg(" + ch.canon + ")";
        break;
      }
      case QHIP_EXPR_IF: {
        visit(e.left); visit(e.right); visit(e.third);
        const ENode &c = nodes[(size_t)e.left], &t = nodes[(size_t)e.right], &f = nodes[(size_t)e.third];
        if (c.type.id != QHIP_BOOL) fail(QHIP_INVALID_ARGUMENT, "Internal error: CASE WHEN must be boolean");
        if (t.type != f.type)   // arrow zip (case.rs:44)
          fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: arguments need to have the same data type: " + dtype_name(t.type) + " vs " + dtype_name(f.type));
        nd.type = t.type; nd.nullable = t.nullable || f.nullable;
        nd.canon = "if(" + c.canon + "," + t.canon + "," + f.canon + ")";
        break;
      }
      case QHIP_EXPR_LIKE: {
        visit(e.left); visit(e.right);
        const ENode &x = nodes[(size_t)e.left], &pat = nodes[(size_t)e.right];
        if (x.type.id != QHIP_UTF8 || pat.type.id != QHIP_UTF8)
          fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid string operation: " + dtype_name(x.type) + " LIKE " + dtype_name(pat.type));
        if (pat.kind != QHIP_EXPR_LITERAL) {
          // the pattern is a column (or any Utf8 expression the generator can address): matched per row, raw (qh_like_raw)
          if (pat.kind != QHIP_EXPR_COLUMN) fail(QHIP_UNSUPPORTED, "LIKE with a computed pattern is not accelerated");
          nd.type = DType(QHIP_BOOL); nd.nullable = x.nullable || pat.nullable; nd.lit_null = false;
          nd.s.clear();
          nd.canon = std::string(e.op ? "nlike(" : "like(") + x.canon + "," + pat.canon + ")";
          break;
        }
        nd.type = DType(QHIP_BOOL); nd.nullable = x.nullable || pat.lit_null; nd.lit_null = pat.lit_null;
        nd.s = canonical_like_pattern(pat.s);
        nd.canon = std::string(e.op ? "nlike(" : "like(") + x.canon + "," + pat.canon + ")";
        break;
      }
      default:
        fail(QHIP_INVALID_ARGUMENT, "unknown expression kind " + std::to_string(e.kind));
    }
    // ---- value bound (integers and decimals; saturating)
    {
      auto sat_mul = [](u128 a, u128 b) -> u128 {
        if (a == kUnbounded || b == kUnbounded) return kUnbounded;
        if (a == 0 || b == 0) return 0;
        if (a > ((u128)1 << 100) / b) return kUnbounded;
        return a * b;
      };
      auto sat_add = [](u128 a, u128 b) -> u128 {
        if (a == kUnbounded || b == kUnbounded) return kUnbounded;
        const u128 r = a + b;
        return r >= ((u128)1 << 100) ? kUnbounded : r;
      };
      auto p10 = [](int e) -> u128 { return e < 0 ? kUnbounded : e > 30 ? kUnbounded : (u128)pow10_i128(e); };
      nd.maxabs = kUnbounded;
      const bool numeric = is_intlike(nd.type) || nd.type.id == QHIP_DECIMAL128;
      if (numeric) {
        if (nd.kind == QHIP_EXPR_COLUMN) {
          const uint64_t m = input[(size_t)nd.column].value_maxabs;
          i128 lo, hi;
          if (m != 0 && m != ~0ULL) nd.maxabs = m;
          else if (int_range(nd.type, lo, hi) && dtype_width(nd.type) <= 4) nd.maxabs = (u128)(hi > -lo ? hi : -lo);
        } else if (nd.kind == QHIP_EXPR_LITERAL) {
          if (!nd.lit_null) {
            const i128 v = nd.type.id == QHIP_DECIMAL128 ? lit_i128(nd) : (nd.type.id == QHIP_UINT64 ? (i128)(u128)nd.lo : (i128)(int64_t)nd.lo);
            nd.maxabs = v < 0 ? (u128)0 - (u128)v : (u128)v;
          } else nd.maxabs = 0;
        } else if (nd.kind == QHIP_EXPR_BINARY && nd.type.id == QHIP_DECIMAL128) {
          const ENode& l = nodes[(size_t)nd.left];
          const ENode& r = nodes[(size_t)nd.right];
          if (nd.op == QHIP_OP_MUL) nd.maxabs = sat_mul(l.maxabs, r.maxabs);
          else if (nd.op == QHIP_OP_ADD || nd.op == QHIP_OP_SUB)
            nd.maxabs = sat_add(sat_mul(l.maxabs, p10(nd.type.scale - l.type.scale)), sat_mul(r.maxabs, p10(nd.type.scale - r.type.scale)));
        } else if (nd.kind == QHIP_EXPR_NEGATIVE) {
          nd.maxabs = nodes[(size_t)nd.left].maxabs;
        } else if (nd.kind == QHIP_EXPR_CAST) {
          const ENode& ch = nodes[(size_t)nd.left];
          if (is_intlike(ch.type) && nd.type.id == QHIP_DECIMAL128) nd.maxabs = sat_mul(ch.maxabs, p10(nd.type.scale));
          else if (ch.type.id == QHIP_DECIMAL128 && nd.type.id == QHIP_DECIMAL128 && nd.type.scale >= ch.type.scale)
            nd.maxabs = sat_mul(ch.maxabs, p10(nd.type.scale - ch.type.scale));
          else if (is_intlike(ch.type) && is_intlike(nd.type)) nd.maxabs = ch.maxabs;
        } else if (nd.kind == QHIP_EXPR_IF) {
          const u128 a = nodes[(size_t)nd.right].maxabs, b = nodes[(size_t)nd.third].maxabs;
          nd.maxabs = a > b ? a : b;
        }
      }
    }
    if (nd.kind == QHIP_EXPR_LITERAL) {
      char buf[96];
      snprintf(buf, sizeof buf, ":%d:%llx:%llx:%a:", nd.lit_null ? 1 : 0, (unsigned long long)nd.lo, (unsigned long long)nd.hi, nd.f);
      nd.canon = "l[" + dtype_name(nd.type) + "]" + buf + nd.s;
    }
    nodes[(size_t)k] = nd;
    state[(size_t)k] = 2;
  };
  for (int k = 0; k < n; ++k) visit(k);
}

}  // namespace qhip
