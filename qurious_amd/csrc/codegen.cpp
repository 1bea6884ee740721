// codegen.cpp — expression tree -> HIP policy struct source.
//
// Only the per-row expression code (load columns, compare, decimal arithmetic, pack keys) is generated;
// every algorithmic piece (wave dedup, DPP reductions, LDS/HBM tables, atomics) is the hand-written
// template code of device/qhip_device.hpp that these policies plug into. Literal VALUES never appear
// in the generated text (they travel in KArgs.lit_*), so `l_shipdate < DATE x` compiles once for all x.
#include "codegen.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>

#include "device/qhip_status.h"

namespace qhip {

static bool is_cmp(int op) { return op >= QHIP_OP_EQ && op <= QHIP_OP_LTEQ; }
static const char* c_cmp(int op) {
  switch (op) {
    case QHIP_OP_EQ: return "==";
    case QHIP_OP_NOTEQ: return "!=";
    case QHIP_OP_GT: return ">";
    case QHIP_OP_GTEQ: return ">=";
    case QHIP_OP_LT: return "<";
    case QHIP_OP_LTEQ: return "<=";
  }
  return "==";
}

std::string ExprGen::ctype(const DType& t) {
  switch (t.id) {
    case QHIP_BOOL: return "bool";
    case QHIP_INT8: return "signed char";
    case QHIP_INT16: return "short";
    case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: return "int";
    case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: return "i64";
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: return "i64";
    case QHIP_UINT8: return "u8";
    case QHIP_UINT16: return "u16";
    case QHIP_UINT32: return "u32";
    case QHIP_UINT64: return "u64";
    case QHIP_FLOAT32: return "float";
    case QHIP_FLOAT64: return "double";
    case QHIP_DECIMAL128: return "i128";
    default: return "int";
  }
}

std::string ExprGen::i128_const(i128 v) {
  char buf[96];
  snprintf(buf, sizeof buf, "qh_mk128(0x%llxULL, (i64)0x%llxULL)", (unsigned long long)(uint64_t)(u128)v,
           (unsigned long long)(uint64_t)((u128)v >> 64));
  return buf;
}

// (Time32 / Time64 are signed integers as far as key words, MIN / MAX images and sort keys go; the typing in expr.cpp keeps
// arithmetic and casts away from them)
// (... and so are the Timestamp types, which are not hash keys at all: check_key_type)
static bool timelike(const DType& t) { return t.id >= QHIP_TIME32_S && t.id <= QHIP_TIMESTAMP_NS; }
static bool intlike(const DType& t) {
  return (t.id >= QHIP_INT8 && t.id <= QHIP_UINT64) || t.id == QHIP_DATE32 || t.id == QHIP_DATE64 || timelike(t);
}
static bool signed_intlike(const DType& t) { return (t.id >= QHIP_INT8 && t.id <= QHIP_INT64) || t.id == QHIP_DATE32 || t.id == QHIP_DATE64 || timelike(t); }
static std::string int_min(const DType& t) {
  switch (t.id) {
    case QHIP_INT8: return "(-128)";
    case QHIP_INT16: return "(-32768)";
    case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: return "(-2147483647-1)";
    case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: return "(-9223372036854775807LL-1)";
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: return "(-9223372036854775807LL-1)";
    default: return "0";
  }
}
static std::string int_max(const DType& t) {
  switch (t.id) {
    case QHIP_INT8: return "127";
    case QHIP_INT16: return "32767";
    case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: return "2147483647";
    case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: return "9223372036854775807LL";
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: return "9223372036854775807LL";
    case QHIP_UINT8: return "255";
    case QHIP_UINT16: return "65535";
    case QHIP_UINT32: return "4294967295U";
    case QHIP_UINT64: return "18446744073709551615ULL";
    default: return "0";
  }
}
// unsigned type wide enough for wrapping arithmetic on t
static std::string wrap_type(const DType& t) { return dtype_width(t) == 8 ? "u64" : "u32"; }

ExprGen::ExprGen(const ExprSet& es, const std::vector<InputCol>& in) : es_(es), in_(in), done_(es.nodes.size(), false) {
  bind.stroff.push_back(0);
  // Column values and Utf8 offsets are read exactly once by every generated kernel: streaming (non-temporal) loads, which
  // on MI355X read ~10 % faster than plain ones (7.0 vs 6.3 TB/s, tools/stream_sweep.py) and keep the L2 for the tables
  // the kernels access at random. QHIP_NO_NT=1 switches them off (A/B measurements).
  nt_ = getenv("QHIP_NO_NT") == nullptr;
}

std::string ExprGen::ok(int k) const {
  return es_.at(k).nullable ? "n" + std::to_string(k) : "true";
}

int ExprGen::col_slot(int table_col) {
  for (size_t s = 0; s < bind.cols.size(); ++s) if (bind.cols[s] == table_col) return (int)s;
  if (bind.cols.size() >= 24) fail(QHIP_UNSUPPORTED, "expression references more than 24 distinct columns");
  bind.cols.push_back(table_col);
  bind.indirect.push_back(table_col >= 0 && table_col < (int)in_.size() && in_[(size_t)table_col].indirect ? 1 : 0);
  bind.narrow.push_back(table_col >= 0 && table_col < (int)in_.size() ? (char)in_[(size_t)table_col].narrow_bytes : (char)0);
  bind.rec.push_back(table_col >= 0 && table_col < (int)in_.size() ? (char)in_[(size_t)table_col].rec_stride : (char)0);
  return (int)bind.cols.size() - 1;
}

int ExprGen::lit_slot(const ENode& n) {
  if (bind.lit_lo.size() >= 24) fail(QHIP_UNSUPPORTED, "more than 24 literals in one kernel");
  uint64_t lo = n.lo; int64_t hi = n.hi;
  if (n.type.id == QHIP_FLOAT64 || n.type.id == QHIP_FLOAT32) { double f = n.f; memcpy(&lo, &f, 8); hi = 0; }
  if (n.type.id == QHIP_UTF8) {   // the first 8 bytes as a kernel scalar: `col = 'BUILDING'` is one masked word compare (qh_streq_lit)
    lo = 0; hi = 0;
    memcpy(&lo, n.s.data(), std::min<size_t>(8, n.s.size()));
  }
  bind.lit_lo.push_back(lo);
  bind.lit_hi.push_back(hi);
  if (n.type.id == QHIP_UTF8) bind.strlits += n.s;
  bind.stroff.push_back((int)bind.strlits.size());
  return (int)bind.lit_lo.size() - 1;
}

int ExprGen::str_slot(const std::string& bytes) {
  if (bind.lit_lo.size() >= 24) fail(QHIP_UNSUPPORTED, "more than 24 literals in one kernel");
  bind.lit_lo.push_back(0);
  bind.lit_hi.push_back(0);
  bind.strlits += bytes;
  bind.stroff.push_back((int)bind.strlits.size());
  return (int)bind.lit_lo.size() - 1;
}

void ExprGen::emit(int k, std::string& out) {
  if (done_[(size_t)k]) return;
  const ENode& n = es_.at(k);
  const std::string K = std::to_string(k);
  const std::string v = "v" + K, nn = "n" + K;
  std::ostringstream o;
  switch (n.kind) {
    case QHIP_EXPR_COLUMN: {
      const int s = col_slot(n.column);
      const std::string S = std::to_string(s);
      // `ld` receives the loads; in raw mode they go to load_code (fields of `Raw w`) and only aliases are emitted here
      std::ostringstream ld;
      const std::string W = raw_ ? "w." : "";
      const std::string decl = raw_ ? "" : "const ";
      auto field = [&](const std::string& type, const std::string& name) { if (raw_) raw_fields += "    " + type + " " + name + ";\n"; };
      if (n.type.id == QHIP_UTF8) {
        field("int", "b" + K); field("int", "l" + K);
        if (in_[(size_t)n.column].utf8_fixed1) {
          // every value is exactly one byte: offsets[i] == i, no offset loads, the data byte is addressed by row number
          ld << "    " << (raw_ ? "" : "const int ") << W << "b" << K << " = (int)(" << row_ << ");\n";
          ld << "    " << (raw_ ? "" : "const int ") << W << "l" << K << " = 1;\n";
        } else {
          const std::string o0 = "((const int*)a.c[" + S + "].v" + base_ + ")[" + idx_ + "]", o1 = "((const int*)a.c[" + S + "].v" + base_ + ")[" + idx_ + " + 1]";
          ld << "    " << (raw_ ? "" : "const int ") << W << "b" << K << " = " << (nt_ ? "__builtin_nontemporal_load(&" + o0 + ")" : o0) << ";\n";
          ld << "    " << (raw_ ? "" : "const int ") << W << "l" << K << " = " << (nt_ ? "__builtin_nontemporal_load(&" + o1 + ")" : o1) << " - " << W << "b" << K << ";\n";
        }
        if (raw_) {
          auto it = utf8_key_words_.find(k);
          if (it != utf8_key_words_.end()) {
            field("u64", "k" + K + "[" + std::to_string(it->second) + "]");
            // key bytes: 8 at a time — or just as many as the column's longest value has (Q1's flags are 1 byte: a 1-byte
            // load per row instead of 64 overlapping 8-byte windows per wavefront)
            const int maxlen = in_[(size_t)n.column].utf8_max_len;
            for (int w2 = 0; w2 < it->second; ++w2) {
              // (one-byte values: addressed by the 64-bit row, so that a lane's consecutive rows are provably adjacent bytes)
              const std::string at = in_[(size_t)n.column].utf8_fixed1 ? "(a.c[" + S + "].d + " + row_ + " + " + std::to_string(8 * w2) + ")"
                                                                       : "(a.c[" + S + "].d + w.b" + K + " + " + std::to_string(8 * w2) + ")";
              const int left = maxlen >= 0 ? maxlen - 8 * w2 : 8;
              if (it->second == 1 && left <= 1) ld << "    w.k" << K << "[0] = (u64)" << (nt_ ? "__builtin_nontemporal_load((const u8*)" + at + ")" : "*(const u8*)" + at) << ";\n";
              else if (it->second == 1 && left <= 2) ld << "    w.k" << K << "[0] = (u64)*(const qh_u16_unaligned*)" << at << ";\n";
              else if (it->second == 1 && left <= 4) ld << "    w.k" << K << "[0] = (u64)*(const qh_u32_unaligned*)" << at << ";\n";
              else ld << "    w.k" << K << "[" << w2 << "] = *(const qh_u64_unaligned*)" << at << ";\n";
            }
          }
          o << "    const int l" << K << " = w.l" << K << "; const u8* p" << K << " = a.c[" << S << "].d + w.b" << K << ";\n";
        } else {
          ld << "    const u8* p" << K << " = a.c[" << S << "].d + b" << K << ";\n";
        }
      } else if (n.type.id == QHIP_BOOL) {
        field("bool", v);
        ld << "    " << (raw_ ? "" : "const bool ") << W << v << " = qh_bit((const u8*)a.c[" << S << "].v, " << row_ << ");\n";
        if (raw_) o << "    const bool " << v << " = w." << v << ";\n";
      } else if (n.type.id == QHIP_NULL) {
        o << "    const int " << v << " = 0;\n";
      } else {
        // a Decimal128 column with a narrow copy (InputCol::narrow_bytes): the kernel loads 4 / 8 bytes per value and widens
        // (an indirect column — read through an index vector — gathers from its SOURCE's narrow copy: ColRange::narrow_buf)
        const int nb = (n.type.id == QHIP_DECIMAL128 || (n.type.id == QHIP_INT64 && in_[(size_t)n.column].narrow_bytes == 4))
                           ? in_[(size_t)n.column].narrow_bytes : 0;
        const std::string LT = nb == 4 ? "int" : nb == 8 ? "i64" : ctype(n.type);   // the type in memory
        field(LT, v);
        {
          // streaming (non-temporal) loads when asked for: column values are read once; Decimal128 goes through a 4 x u32
          // vector (the builtin takes integer / float / vector types)
          // tile-relative indexing (uniform 64-bit tile base + 32-bit lane offset): the lane offset is turned into BYTES in 32
          // bits, so that the load is `global_load v, v_offset32, s[base]` — one VALU instruction for the address instead of a
          // 64-bit shift-add per column and row (a tile is far smaller than 4 GB / 16)
          const std::string T = LT;
          const bool ind = in_[(size_t)n.column].indirect;   // late materialisation: source[index[row]] (random access, no streaming hint)
          const int rec = ind ? in_[(size_t)n.column].rec_stride : 0;   // a field of the source's record copy (KCol::v = records + field offset)
          const std::string addr = rec ? "(*(const " + T + "*)((const char*)a.c[" + S + "].v + (size_t)((const u32*)a.c[" + S + "].d)[" + row_ + "] * " + std::to_string(rec) + "u))"
                                   : ind ? "((const " + T + "*)a.c[" + S + "].v)[((const u32*)a.c[" + S + "].d)[" + row_ + "]]"
                                   : base_.empty() ? "((const " + T + "*)a.c[" + S + "].v)[" + idx_ + "]"
                                                 : "(*(const " + T + "*)((const char*)((const " + T + "*)a.c[" + S + "].v" + base_ + ") + (size_t)((u32)(" + idx_ +
                                                       ") * (u32)sizeof(" + T + "))))";
          std::string rhs = addr;
          if (ind) rhs = addr;
          else if (nt_ && n.type.id == QHIP_DECIMAL128 && nb == 0) rhs = "qh_nt_load_i128(&" + addr + ")";
          else if (nt_) rhs = "__builtin_nontemporal_load(&" + addr + ")";
          if (nb && !raw_) rhs = "(" + ctype(n.type) + ")(" + rhs + ")";   // (sign-extending)
          ld << "    " << (raw_ ? "" : "const " + ctype(n.type) + " ") << W << v << " = " << rhs << ";\n";
        }
        if (raw_) o << "    const " << ctype(n.type) << " " << v << " = (" << ctype(n.type) << ")w." << v << ";\n";
      }
      if (n.nullable) {
        if (n.type.id == QHIP_NULL) o << "    const bool " << nn << " = false;\n";
        else {
          field("bool", nn);
          ld << "    " << (raw_ ? "" : "const bool ") << W << nn << " = qh_bit(a.c[" << S << "].n, " << row_ << ");\n";
          if (raw_) o << "    const bool " << nn << " = w." << nn << ";\n";
        }
      }
      if (raw_) load_code += ld.str(); else o << ld.str();
      break;
    }
    case QHIP_EXPR_LITERAL: {
      if (n.lit_null) {
        if (n.type.id == QHIP_UTF8) o << "    const u8* p" << K << " = a.strlit; const int l" << K << " = 0;\n";
        else o << "    const " << ctype(n.type) << " " << v << " = 0;\n";
        o << "    const bool " << nn << " = false;\n";
        break;
      }
      const int s = lit_slot(n);
      const std::string S = std::to_string(s);
      if (n.type.id == QHIP_UTF8)
        o << "    const u8* p" << K << " = a.strlit + a.stroff[" << S << "]; const int l" << K << " = a.stroff[" << s + 1 << "] - a.stroff[" << S << "];"
          << " const u64 w" << K << " = a.lit_lo[" << S << "];\n";
      else if (n.type.id == QHIP_DECIMAL128)
        o << "    const i128 " << v << " = qh_mk128(a.lit_lo[" << S << "], a.lit_hi[" << S << "]);\n";
      else if (n.type.id == QHIP_FLOAT64)
        o << "    const double " << v << " = qh_f64(a.lit_lo[" << S << "]);\n";
      else if (n.type.id == QHIP_FLOAT32)
        o << "    const float " << v << " = (float)qh_f64(a.lit_lo[" << S << "]);\n";
      else if (n.type.id == QHIP_BOOL)
        o << "    const bool " << v << " = (a.lit_lo[" << S << "] & 1) != 0;\n";
      else
        o << "    const " << ctype(n.type) << " " << v << " = (" << ctype(n.type) << ")a.lit_lo[" << S << "];\n";
      break;
    }
    case QHIP_EXPR_BINARY: {
      emit(n.left, out);
      emit(n.right, out);
      const ENode& l = es_.at(n.left);
      const ENode& r = es_.at(n.right);
      const std::string lv = val(n.left), rv = val(n.right), lo = ok(n.left), ro = ok(n.right);
      if (is_cmp(n.op)) {
        std::string e;
        if (l.type.id == QHIP_UTF8) {
          // (in)equality with a literal: its first 8 bytes travel as a kernel scalar (lit_slot), so values of up to 8 bytes
          // compare as one masked word instead of a byte loop with an early exit
          const bool rlit = r.kind == QHIP_EXPR_LITERAL && !r.lit_null, llit = l.kind == QHIP_EXPR_LITERAL && !l.lit_null;
          std::string eqx = "qh_streq(" + ptr(n.left) + ", " + len(n.left) + ", " + ptr(n.right) + ", " + len(n.right) + ")";
          if (rlit && !llit) eqx = "qh_streq_lit(" + ptr(n.left) + ", " + len(n.left) + ", " + ptr(n.right) + ", " + len(n.right) + ", w" + std::to_string(n.right) + ")";
          else if (llit && !rlit) eqx = "qh_streq_lit(" + ptr(n.right) + ", " + len(n.right) + ", " + ptr(n.left) + ", " + len(n.left) + ", w" + std::to_string(n.left) + ")";
          if (n.op == QHIP_OP_EQ) e = eqx;
          else if (n.op == QHIP_OP_NOTEQ) e = "!" + eqx;
          else e = "(qh_strcmp(" + ptr(n.left) + ", " + len(n.left) + ", " + ptr(n.right) + ", " + len(n.right) + ") " + c_cmp(n.op) + " 0)";
        } else if (dtype_is_float(l.type)) {
          // arrow-rs compares floats in IEEE total order
          e = "(qh_f64_ord((double)" + lv + ") " + c_cmp(n.op) + " qh_f64_ord((double)" + rv + "))";
        } else if (l.type.id == QHIP_BOOL) {
          e = "((int)" + lv + " " + c_cmp(n.op) + " (int)" + rv + ")";
        } else {
          e = "(" + lv + " " + c_cmp(n.op) + " " + rv + ")";
        }
        if (n.nullable) {
          o << "    const bool " << nn << " = " << lo << " && " << ro << ";\n";
          if (l.type.id == QHIP_UTF8) o << "    const bool " << v << " = " << nn << " ? " << e << " : false;\n";
          else o << "    const bool " << v << " = " << e << ";\n";
        } else {
          o << "    const bool " << v << " = " << e << ";\n";
        }
      } else if (n.op == QHIP_OP_AND) {
        // Kleene AND (binary.rs:44-46): false AND NULL = false
        if (n.nullable) {
          o << "    const bool " << nn << " = (" << lo << " && " << ro << ") || (" << lo << " && !" << lv << ") || (" << ro << " && !" << rv << ");\n";
          o << "    const bool " << v << " = (!" << lo << " || " << lv << ") && (!" << ro << " || " << rv << ");\n";
        } else {
          o << "    const bool " << v << " = " << lv << " && " << rv << ";\n";
        }
      } else if (n.op == QHIP_OP_OR) {
        if (n.nullable) {
          o << "    const bool " << nn << " = (" << lo << " && " << ro << ") || (" << lo << " && " << lv << ") || (" << ro << " && " << rv << ");\n";
          o << "    const bool " << v << " = (" << lo << " && " << lv << ") || (" << ro << " && " << rv << ");\n";
        } else {
          o << "    const bool " << v << " = " << lv << " || " << rv << ";\n";
        }
      } else {
        if (n.nullable) o << "    const bool " << nn << " = " << lo << " && " << ro << ";\n";
        const std::string T = ctype(n.type);
        const bool ld = l.type.id == QHIP_DECIMAL128, rd = r.type.id == QHIP_DECIMAL128;
        if (ld || rd) {
          if (n.op == QHIP_OP_DIV) {
            auto tof = [&](const ENode& x, const std::string& xv) {
              if (x.type.id == QHIP_DECIMAL128) {
                char b[64]; snprintf(b, sizeof b, "1e%d", x.type.scale);
                return "((double)" + xv + " / " + b + ")";
              }
              return "(double)" + xv;
            };
            o << "    const double " << v << " = " << tof(l, lv) << " / " << tof(r, rv) << ";\n";
          } else if (n.op == QHIP_OP_MUL) {
            // operand magnitudes known from the columns' statistics (ENode::maxabs): a 32 x 32 -> 64 or 64 x 64 -> 128
            // multiply written out, instead of the generic 128-bit product behind wave-uniform "do all lanes fit" tests
            const u128 b31 = (u128)1 << 31, b63 = (u128)1 << 63;
            if (l.maxabs < b31 && r.maxabs < b31)
              o << "    const i128 " << v << " = (i128)((i64)(int)(u32)(u128)" << lv << " * (i64)(int)(u32)(u128)" << rv << ");\n";
            else if (l.maxabs < b63 && r.maxabs < b63)
              o << "    const i128 " << v << " = qh_mul_i64_i128((i64)(u64)(u128)" << lv << ", (i64)(u64)(u128)" << rv << ");\n";
            else
              o << "    const i128 " << v << " = " << (raw_ ? "qh_mul_i128(" : "qh_mul_i128_plain(") << lv << ", " << rv << ");\n";
          } else {
            const int s = n.type.scale;
            if (n.maxabs < ((u128)1 << 63)) {
              // the result (hence both rescaled operands) fits 63 bits: the sum in 64-bit arithmetic, sign-extended
              auto k64 = [&](int e) { return std::to_string((unsigned long long)(u128)pow10_i128(e)) + "ULL"; };
              std::string a = "(u64)(u128)" + lv, b = "(u64)(u128)" + rv;
              if (s > l.type.scale) a = "(" + a + " * " + k64(s - l.type.scale) + ")";
              if (s > r.type.scale) b = "(" + b + " * " + k64(s - r.type.scale) + ")";
              o << "    const i128 " << v << " = (i128)(i64)(" << a << (n.op == QHIP_OP_ADD ? " + " : " - ") << b << ");\n";
            } else {
              std::string a = "(u128)" + lv, b = "(u128)" + rv;
              if (s > l.type.scale) a = "(" + a + " * (u128)" + i128_const(pow10_i128(s - l.type.scale)) + ")";
              if (s > r.type.scale) b = "(" + b + " * (u128)" + i128_const(pow10_i128(s - r.type.scale)) + ")";
              o << "    const i128 " << v << " = (i128)(" << a << (n.op == QHIP_OP_ADD ? " + " : " - ") << b << ");\n";
            }
          }
        } else if (dtype_is_float(n.type)) {
          if (n.op == QHIP_OP_MOD) o << "    const " << T << " " << v << " = (" << T << ")fmod((double)" << lv << ", (double)" << rv << ");\n";
          else {
            const char* c = n.op == QHIP_OP_ADD ? "+" : n.op == QHIP_OP_SUB ? "-" : n.op == QHIP_OP_MUL ? "*" : "/";
            o << "    const " << T << " " << v << " = " << lv << " " << c << " " << rv << ";\n";
          }
        } else {
          const std::string U = wrap_type(n.type);
          if (n.op == QHIP_OP_ADD || n.op == QHIP_OP_SUB || n.op == QHIP_OP_MUL) {
            const char* c = n.op == QHIP_OP_ADD ? "+" : n.op == QHIP_OP_SUB ? "-" : "*";
            o << "    const " << T << " " << v << " = (" << T << ")((" << U << ")" << lv << " " << c << " (" << U << ")" << rv << ");\n";
          } else {
            // arrow `div`/`rem` are checked: DivideByZero, and MIN / -1 overflows (evaluated on valid rows only)
            o << "    " << T << " " << v << " = 0;\n";
            o << "    if (" << (n.nullable ? nn : std::string("true")) << ") {\n";
            o << "      if (" << rv << " == 0) err |= " << (1u << QS_DIV_ZERO) << "u;\n";
            if (signed_intlike(n.type)) {
              if (n.op == QHIP_OP_DIV)
                o << "      else if (" << lv << " == " << int_min(n.type) << " && " << rv << " == -1) err |= " << (1u << QS_ARITH_OVERFLOW) << "u;\n";
              else
                o << "      else if (" << rv << " == -1) " << v << " = 0;\n";
            }
            o << "      else " << v << " = (" << T << ")(" << lv << (n.op == QHIP_OP_DIV ? " / " : " % ") << rv << ");\n";
            o << "    }\n";
          }
        }
      }
      break;
    }
    case QHIP_EXPR_CAST: {
      emit(n.left, out);
      const ENode& ch = es_.at(n.left);
      const std::string cv = val(n.left), co = ok(n.left);
      const DType& from = ch.type;
      const DType& to = n.type;
      const std::string T = ctype(to);
      if (n.nullable) o << "    const bool " << nn << " = " << co << ";\n";
      const std::string live = n.nullable ? nn : std::string("true");
      auto flag = [&](const std::string& cond) {
        o << "    if (" << live << " && (" << cond << ")) err |= " << (1u << QS_CAST_OVERFLOW) << "u;\n";
      };
      if (from == to) {
        o << "    const " << T << " " << v << " = " << cv << ";\n";
      } else if (from.id == QHIP_BOOL) {
        o << "    const " << T << " " << v << " = " << cv << " ? 1 : 0;\n";
      } else if (intlike(from) && intlike(to)) {
        if (from.id == QHIP_DATE32 && to.id == QHIP_DATE64) o << "    const i64 " << v << " = (i64)" << cv << " * 86400000LL;\n";
        else if (from.id == QHIP_DATE64 && to.id == QHIP_DATE32) o << "    const int " << v << " = (int)(" << cv << " / 86400000LL);\n";
        else {
          // range check in i128 so that every (from, to) pair is handled uniformly
          const std::string wide = std::string(from.id == QHIP_UINT64 ? "(i128)(u128)" : "(i128)") + cv;
          const std::string lo = signed_intlike(to) ? "(i128)" + int_min(to) : "(i128)0";
          flag(wide + " < " + lo + " || " + wide + " > (i128)" + int_max(to));
          o << "    const " << T << " " << v << " = (" << T << ")" << cv << ";\n";
        }
      } else if (intlike(from) && dtype_is_float(to)) {
        o << "    const " << T << " " << v << " = (" << T << ")" << cv << ";\n";
      } else if (dtype_is_float(from) && dtype_is_float(to)) {
        o << "    const " << T << " " << v << " = (" << T << ")" << cv << ";\n";
      } else if (dtype_is_float(from) && intlike(to)) {
        const std::string tr = "trunc((double)" + cv + ")";
        const std::string lo = signed_intlike(to) ? "(double)" + int_min(to) : "0.0";
        flag("!(" + tr + " >= " + lo + " && " + tr + " <= (double)" + int_max(to) + ")");
        o << "    const " << T << " " << v << " = (" << T << ")" << tr << ";\n";
      } else if (intlike(from) && to.id == QHIP_DECIMAL128) {
        const std::string wide = std::string(from.id == QHIP_UINT64 ? "(i128)(u128)" : "(i128)") + cv;
        o << "    const i128 " << v << " = " << wide << " * " << i128_const(pow10_i128(to.scale)) << ";\n";
        const std::string lim = i128_const(pow10_i128(to.precision));
        flag(v + " >= " + lim + " || " + v + " <= -" + lim);
      } else if (from.id == QHIP_DECIMAL128 && to.id == QHIP_DECIMAL128) {
        if (to.scale >= from.scale) {
          o << "    const i128 " << v << " = " << cv << " * " << i128_const(pow10_i128(to.scale - from.scale)) << ";\n";
        } else {
          const std::string d = i128_const(pow10_i128(from.scale - to.scale));
          o << "    i128 " << v << " = " << cv << " / " << d << "; { const i128 rm = " << cv << " % " << d << ", hf = " << d << " / 2;"
            << " if (rm >= hf) " << v << " += 1; else if (-rm >= hf) " << v << " -= 1; }\n";
        }
        const std::string lim = i128_const(pow10_i128(to.precision));
        flag(v + " >= " + lim + " || " + v + " <= -" + lim);
      } else if (from.id == QHIP_DECIMAL128 && dtype_is_float(to)) {
        char b[64]; snprintf(b, sizeof b, "1e%d", from.scale);
        o << "    const " << T << " " << v << " = (" << T << ")((double)" << cv << " / " << b << ");\n";
      } else if (from.id == QHIP_DECIMAL128 && intlike(to)) {
        o << "    const i128 w" << K << " = " << cv << " / " << i128_const(pow10_i128(from.scale)) << ";\n";
        const std::string lo = signed_intlike(to) ? "(i128)" + int_min(to) : "(i128)0";
        flag("w" + K + " < " + lo + " || w" + K + " > (i128)" + int_max(to));
        o << "    const " << T << " " << v << " = (" << T << ")w" << K << ";\n";
      } else if (dtype_is_float(from) && to.id == QHIP_DECIMAL128) {
        char b[64]; snprintf(b, sizeof b, "1e%d", to.scale);
        o << "    const double f" << K << " = round((double)" << cv << " * " << b << ");\n";
        char lim[64]; snprintf(lim, sizeof lim, "1e%d", to.precision);
        flag("!(fabs(f" + K + ") < " + lim + ")");
        o << "    const i128 " << v << " = (fabs(f" << K << ") < 1.7e38) ? (i128)f" << K << " : (i128)0;\n";
      } else {
        fail(QHIP_UNSUPPORTED, "device cast " + dtype_name(from) + " -> " + dtype_name(to));
      }
      break;
    }
    case QHIP_EXPR_IS_NULL:
    case QHIP_EXPR_IS_NOT_NULL: {
      emit(n.left, out);
      o << "    const bool " << v << " = " << (n.kind == QHIP_EXPR_IS_NULL ? "!" : "") << "(" << ok(n.left) << ");\n";
      break;
    }
    case QHIP_EXPR_IF: {
      // arrow zip(mask, truthy, falsy) (case.rs:44): a NULL or false mask selects the falsy side; both sides are evaluated
      // for every row (like the reference's full-array evaluation, so their error flags are raised regardless of the mask)
      emit(n.left, out); emit(n.right, out); emit(n.third, out);
      const std::string sel = "s" + K;
      o << "    const bool " << sel << " = " << ok(n.left) << " && " << val(n.left) << ";\n";
      if (n.nullable) o << "    const bool " << nn << " = " << sel << " ? " << ok(n.right) << " : " << ok(n.third) << ";\n";
      if (n.type.id == QHIP_UTF8) {
        o << "    const u8* p" << K << " = " << sel << " ? " << ptr(n.right) << " : " << ptr(n.third) << ";\n";
        o << "    const int l" << K << " = " << sel << " ? " << len(n.right) << " : " << len(n.third) << ";\n";
      } else {
        o << "    const " << ctype(n.type) << " " << v << " = " << sel << " ? " << val(n.right) << " : " << val(n.third) << ";\n";
      }
      break;
    }
    case QHIP_EXPR_LIKE: {
      emit(n.left, out);
      if (n.right >= 0 && es_.at(n.right).kind == QHIP_EXPR_COLUMN) {
        // a pattern per row (like.rs:28-43 -> arrow `like` over two arrays)
        emit(n.right, out);
        if (n.nullable) o << "    const bool " << nn << " = " << ok(n.left) << " && " << ok(n.right) << ";\n";
        o << "    const bool " << v << " = " << (n.nullable ? nn + " && " : std::string("")) << (n.op ? "!" : "") << "qh_like_raw(" << ptr(n.left) << ", "
          << len(n.left) << ", " << ptr(n.right) << ", " << len(n.right) << ");\n";
        break;
      }
      if (n.lit_null) {
        o << "    const bool " << v << " = false;\n    const bool " << nn << " = false;\n";
        break;
      }
      const int s = str_slot(n.s);
      if (n.nullable) o << "    const bool " << nn << " = " << ok(n.left) << ";\n";
      o << "    const bool " << v << " = " << (n.nullable ? nn + " && " : std::string("")) << (n.op ? "!" : "") << "qh_like(" << ptr(n.left) << ", "
        << len(n.left) << ", a.strlit + a.stroff[" << s << "], a.stroff[" << s + 1 << "] - a.stroff[" << s << "]);\n";
      break;
    }
    case QHIP_EXPR_NEGATIVE: {
      emit(n.left, out);
      if (n.nullable) o << "    const bool " << nn << " = " << ok(n.left) << ";\n";
      const std::string T = ctype(n.type);
      if (dtype_is_float(n.type)) o << "    const " << T << " " << v << " = -" << val(n.left) << ";\n";
      else if (n.type.id == QHIP_DECIMAL128) o << "    const i128 " << v << " = (i128)((u128)0 - (u128)" << val(n.left) << ");\n";
      else o << "    const " << T << " " << v << " = (" << T << ")((" << wrap_type(n.type) << ")0 - (" << wrap_type(n.type) << ")" << val(n.left) << ");\n";
      break;
    }
  }
  out += o.str();
  done_[(size_t)k] = true;
}

// ---------------------------------------------------------------- key packing shared by aggregate / join / partition kernels
// Key types: the reference's create_hashes (utils/array.rs:190-210) hashes Int64, UInt8, Int32, Utf8, Date32 / Date64,
// Time32 / Time64, Decimal128 and Decimal256 and raises an InternalError for anything else. The HIP path takes the same list
// MINUS Decimal256 (out of scope, DESIGN §8: no such type id crosses the C ABI): a plan with such a key gets the reference's
// own error text here and the shim keeps the CPU node — a documented gap, not a drop-in.
static void check_key_type(const DType& t) {
  switch (t.id) {
    case QHIP_INT64: case QHIP_UINT8: case QHIP_INT32: case QHIP_UTF8: case QHIP_DATE32: case QHIP_DATE64: case QHIP_DECIMAL128:
    case QHIP_TIME32_S: case QHIP_TIME32_MS: case QHIP_TIME64_US: case QHIP_TIME64_NS:
      return;
    default:
      fail(QHIP_INVALID_ARGUMENT, "Internal error: Unsupported data type in hasher: " + dtype_name(t));
  }
}

// words of a packed Utf8 key: the bytes little-endian across the words, the length in the top byte of the last word
static int utf8_key_words(const ExprSet& es, const std::vector<InputCol>& input, int root) {
  const ENode& nd = es.at(root);
  int maxlen = 7;
  if (nd.kind == QHIP_EXPR_COLUMN && input[(size_t)nd.column].utf8_max_len >= 0) maxlen = input[(size_t)nd.column].utf8_max_len;
  else if (nd.kind == QHIP_EXPR_LITERAL) maxlen = (int)nd.s.size();
  const int words = std::max(1, (maxlen + 1 + 7) / 8);
  // (up to 7 words = 55 bytes + the length byte per Utf8 key — round 4; rounds 1-3: 4 words — inside the 8 words a whole key may have)
  if (words > 7) fail(QHIP_UNSUPPORTED, "Utf8 group/join key longer than 55 bytes is not accelerated");
  return words;
}

static void layout_keys(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, bool with_null_mask,
                        std::vector<KeyDesc>& keys, int& W, bool& mask_word) {
  keys.clear();
  mask_word = false;
  if (with_null_mask)
    for (int k = 0; k < n; ++k) if (es.at(roots[k]).nullable) mask_word = true;
  int off = mask_word ? 1 : 0;
  for (int k = 0; k < n; ++k) {
    const ENode& nd = es.at(roots[k]);
    check_key_type(nd.type);
    KeyDesc kd;
    kd.root = roots[k]; kd.type = nd.type; kd.nullable = nd.nullable; kd.word_off = off;
    kd.words = nd.type.id == QHIP_DECIMAL128 ? 2 : nd.type.id == QHIP_UTF8 ? utf8_key_words(es, input, roots[k]) : 1;
    off += kd.words;
    keys.push_back(kd);
  }
  W = off;
  if (n > 60) fail(QHIP_UNSUPPORTED, "more than 60 key columns");
}

// statements storing the key words of the current row into `dst[...]`; `valid_all` receives the conjunction of validities
static void emit_key_words(ExprGen& g, const ExprSet& es, const std::vector<KeyDesc>& keys, bool mask_word, const std::string& dst,
                           std::string& out, std::string* valid_all) {
  std::ostringstream o;
  std::string all = "true";
  if (mask_word) o << "    u64 nm = 0;\n";
  for (size_t k = 0; k < keys.size(); ++k) {
    const KeyDesc& kd = keys[k];
    std::string code;
    g.emit(kd.root, code);
    o << code;
    const std::string okx = g.ok(kd.root);
    const std::string w = dst + "[" + std::to_string(kd.word_off) + "]";
    std::string value;
    if (kd.type.id == QHIP_UTF8) {
      if (g.raw_key_prefetched(kd.root))
        o << "    u64 ks" << k << "[" << kd.words << "]; qh_pack_words<" << kd.words << ">(w.k" << kd.root << ", " << g.len(kd.root) << ", ks" << k << ");\n";
      else
        o << "    u64 ks" << k << "[" << kd.words << "]; qh_pack_str<" << kd.words << ">(" << g.ptr(kd.root) << ", " << g.len(kd.root) << ", ks" << k << ");\n";
      o << "    if (" << okx << " && " << g.len(kd.root) << " > " << 8 * kd.words - 1 << ") err |= " << (1u << QS_KEY_TOO_LONG) << "u;\n";
      for (int w2 = 0; w2 < kd.words; ++w2)
        o << "    " << dst << "[" << kd.word_off + w2 << "] = " << (kd.nullable ? okx + " ? " : std::string("")) << "ks" << k << "[" << w2 << "]"
          << (kd.nullable ? " : 0ULL" : "") << ";\n";
      if (kd.nullable) { if (mask_word) o << "    nm |= " << okx << " ? 0ULL : " << (1ULL << k) << "ULL;\n"; all += " && " + okx; }
      continue;
    } else if (kd.type.id == QHIP_DECIMAL128) {
      value = "(u64)(u128)" + g.val(kd.root);
    } else {
      value = "(u64)(i64)" + g.val(kd.root);
    }
    if (kd.nullable) {
      o << "    " << w << " = " << okx << " ? " << value << " : 0ULL;\n";
      if (kd.type.id == QHIP_DECIMAL128)
        o << "    " << dst << "[" << kd.word_off + 1 << "] = " << okx << " ? (u64)((u128)" << g.val(kd.root) << " >> 64) : 0ULL;\n";
      if (mask_word) o << "    nm |= " << okx << " ? 0ULL : " << (1ULL << k) << "ULL;\n";
      all += " && " + okx;
    } else {
      o << "    " << w << " = " << value << ";\n";
      if (kd.type.id == QHIP_DECIMAL128) o << "    " << dst << "[" << kd.word_off + 1 << "] = (u64)((u128)" << g.val(kd.root) << " >> 64);\n";
    }
  }
  if (mask_word) o << "    " << dst << "[0] = nm;\n";
  out += o.str();
  if (valid_all) *valid_all = all;
}

// ---------------------------------------------------------------- aggregate policy
static std::string ord64(const DType& t, const std::string& v) {
  if (dtype_is_float(t)) return "qh_f64_ord((double)" + v + ")";
  if (signed_intlike(t)) return "((u64)(i64)" + v + " ^ 0x8000000000000000ULL)";
  return "(u64)" + v;
}

void plan_aggregate(const ExprSet& es, const std::vector<InputCol>& input, int predicate_root, const int32_t* group_roots, int n_groups,
                    const qhip_agg* aggs, int n_aggs, int rows_per_thread, AggPlan& P, bool dev_rows) {
  const char* dr = dev_rows ? ", true" : "";   // the instantiation that reads the row count from the device (a join output of deferred size)
  P = AggPlan();
  P.R = rows_per_thread;
  if (predicate_root >= 0 && es.at(predicate_root).type.id != QHIP_BOOL)
    fail(QHIP_INVALID_ARGUMENT, "filter predicate must be Boolean, got " + dtype_name(es.at(predicate_root).type));
  layout_keys(es, input, group_roots, n_groups, true, P.keys, P.W, P.null_mask_word);
  if (P.W > 8) fail(QHIP_UNSUPPORTED, "group key wider than 8 words");

  // cells: cell 0 counts the rows of the group
  auto add_cell = [&](int kind, int arg, bool is_min, int words) {
    for (size_t c = 0; c < P.cells.size(); ++c)
      if (P.cells[c].kind == kind && P.cells[c].arg == arg && P.cells[c].is_min == is_min) return (int)c;
    CellDesc cd; cd.kind = kind; cd.arg = arg; cd.is_min = is_min; cd.words = words; cd.off = 0;
    P.cells.push_back(cd);
    return (int)P.cells.size() - 1;
  };
  add_cell(CELL_ROWS, -1, false, 1);
  auto arg_index = [&](int root) {
    const ENode& nd = es.at(root);
    for (size_t a = 0; a < P.args.size(); ++a) if (es.at(P.args[a].root).canon == nd.canon) return (int)a;
    ArgDesc ad; ad.root = root; ad.type = nd.type; ad.nullable = nd.nullable;
    P.args.push_back(ad);
    return (int)P.args.size() - 1;
  };
  for (int k = 0; k < n_aggs; ++k) {
    const qhip_agg& a = aggs[k];
    if (a.expr < 0 || a.expr >= (int)es.nodes.size()) fail(QHIP_INVALID_ARGUMENT, "aggregate argument index out of range");
    const ENode& arg = es.at(a.expr);
    AggDesc ad; ad.kind = a.kind; ad.ret = DType(a.return_type); ad.value_cell = -1; ad.count_cell = 0;
    // a NULL-typed / all-null literal argument never contributes
    ad.arg = arg_index(a.expr);
    const bool nullable = arg.nullable;
    auto count_cell = [&]() { return nullable ? add_cell(CELL_CNT, ad.arg, false, 1) : 0; };
    switch (a.kind) {
      case QHIP_AGG_COUNT:
        ad.ret = DType(QHIP_INT64);
        ad.count_cell = count_cell();
        break;
      case QHIP_AGG_SUM: {
        // sum.rs:36-51: only these return types have an accumulator, and the argument array is downcast to it
        if (!(ad.ret.id == QHIP_UINT64 || ad.ret.id == QHIP_INT64 || ad.ret.id == QHIP_FLOAT64 || ad.ret.id == QHIP_DECIMAL128))
          fail(QHIP_INVALID_ARGUMENT, "Internal error: Sum not supported for " + arg.canon + ": " + dtype_name(ad.ret));
        if (arg.type.id != ad.ret.id)
          fail(QHIP_INVALID_ARGUMENT, "SUM argument type " + dtype_name(arg.type) + " does not match return type " + dtype_name(ad.ret));
        const int kind = ad.ret.id == QHIP_DECIMAL128 ? CELL_SUM_I128 : ad.ret.id == QHIP_FLOAT64 ? CELL_SUM_F64 : CELL_SUM_U64;
        ad.value_cell = add_cell(kind, ad.arg, false, kind == CELL_SUM_I128 ? 2 : 1);
        ad.count_cell = count_cell();
        break;
      }
      case QHIP_AGG_AVG: {
        // avg.rs:36-61
        if (arg.type.id == QHIP_DECIMAL128 && ad.ret.id == QHIP_DECIMAL128) {
          ad.value_cell = add_cell(CELL_SUM_I128, ad.arg, false, 2);
        } else if (arg.type.id == QHIP_FLOAT64 && ad.ret.id == QHIP_FLOAT64) {
          ad.value_cell = add_cell(CELL_SUM_F64, ad.arg, false, 1);
        } else {
          fail(QHIP_INVALID_ARGUMENT, "Internal error: Unsupported data type [" + dtype_name(ad.ret) + "] for AVG aggregate over " + dtype_name(arg.type));
        }
        ad.count_cell = count_cell();
        break;
      }
      case QHIP_AGG_MIN:
      case QHIP_AGG_MAX: {
        if (arg.type != ad.ret) fail(QHIP_INVALID_ARGUMENT, "MIN/MAX argument type differs from return type");
        const bool is_min = a.kind == QHIP_AGG_MIN;
        if (arg.type.id == QHIP_DECIMAL128) ad.value_cell = add_cell(CELL_MAXORD128, ad.arg, is_min, 3);
        else if (intlike(arg.type) || dtype_is_float(arg.type)) ad.value_cell = add_cell(CELL_MAXORD64, ad.arg, is_min, 1);
        else fail(QHIP_UNSUPPORTED, "MIN/MAX over " + dtype_name(arg.type));
        ad.count_cell = count_cell();
        break;
      }
      default:
        fail(QHIP_INVALID_ARGUMENT, "unknown aggregate kind " + std::to_string(a.kind));
    }
    P.aggs.push_back(ad);
  }
  int off = 0;
  for (auto& c : P.cells) { c.off = off; off += c.words; }
  P.slot_words = 1 + P.W + off;
  // Decimal128 arguments whose magnitude is known (ENode::maxabs) travel and accumulate narrower: a value below 2^63 is kept
  // as i64 in the Row; a SUM over values below 2^39 is accumulated per lane in 64 bits (a lane sees at most 2^24 of a
  // table's < 2^32 rows, so the lane sum stays below 2^63) and widened when the lanes are reduced at the end of the kernel
  std::vector<bool> arg64(P.args.size(), false), arg_acc64(P.args.size(), false);
  for (size_t a = 0; a < P.args.size(); ++a) {
    const ENode& nd = es.at(P.args[a].root);
    arg64[a] = nd.type.id == QHIP_DECIMAL128 && nd.maxabs < ((u128)1 << 63);
    arg_acc64[a] = nd.type.id == QHIP_DECIMAL128 && nd.maxabs < ((u128)1 << 39);
  }
  auto cell_acc64 = [&](const CellDesc& c) { return c.kind == CELL_SUM_I128 && arg_acc64[(size_t)c.arg]; };
  // hot-key cache size: lane-private accumulators for KC keys must fit the register file next to R rows
  int part_regs = 0;
  for (auto& c : P.cells) if (c.kind != CELL_ROWS) part_regs += cell_acc64(c) ? 2 : 2 * c.words - (c.kind == CELL_MAXORD128 ? 2 : 0);
  P.KC = P.W == 0 ? 0 : (part_regs * 4 <= 96 ? 4 : part_regs * 2 <= 96 ? 2 : 0);
  // every SUM narrow (64-bit lane accumulators): a cached key then costs its compare plus two VALU instructions per cell and
  // row, and what counts is the number of PASSES over the tile's rows. Measured on TPC-H Q1's list (5 narrow sums, 4 groups
  // of 49 / 25 / 25 / 1 % of the rows; kernel time for SF10's rows): KC 0 0.680 ms, 1 0.648, 2 0.641 (R = 3), 3 0.717,
  // 4 0.719 — two cached keys and the LDS table for the rest beat four passes; a single narrow sum (q1_mini) still wants
  // every group cached (KC 3-4 0.323 ms, KC 2 0.403).
  bool all_narrow = P.W > 0 && !P.cells.empty();
  for (auto& c : P.cells) if (c.kind != CELL_ROWS && !cell_acc64(c)) all_narrow = false;
  if (all_narrow && part_regs > 0) P.KC = std::max(1, std::min(4, 24 / part_regs));
  // ... and when the kernel streams NARROW COPIES of its decimal columns (InputCol::narrow_bytes: 22 instead of 70 bytes per row
  // for Q1) it is no longer HBM that bounds it but the LDS: PMC on Q1 at KC = 2 — the quarter of the rows that miss the two
  // cached keys keep the CU's LDS pipeline busy 76 % of the time (same-address DS atomics: SQ_LDS_BANK_CONFLICT = 26 % of the
  // CU's cycles). Measured then, kernel + host per query: KC 2 0.522 ms, 3 0.409-0.422, 4 0.298 (R = 2; R = 1: 0.355, 3: 0.349).
  bool narrow_copies = false;
  for (auto& ic : input) if (ic.narrow_bytes > 0 && !ic.indirect) narrow_copies = true;
  if (all_narrow && narrow_copies && part_regs > 0) P.KC = std::max(1, std::min(4, 48 / part_regs));
  if (const char* kc = getenv("QHIP_AGG_KC")) { if (*kc && P.W > 0) P.KC = atoi(kc); }   // tuning experiments only
  int row_regs = 1 + 2 * P.W;   // VGPRs of one evaluated row: pass flag, key words, argument dwords
  for (size_t a = 0; a < P.args.size(); ++a) {
    bool value_needed = false;
    for (auto& c : P.cells) if (c.arg == (int)a && c.kind != CELL_CNT) value_needed = true;
    if (value_needed) row_regs += arg64[a] ? 2 : std::max(1, dtype_width(P.args[a].type) / 4);
    if (P.args[a].nullable) row_regs += 1;
  }
  if (P.R <= 0) {
    // rows per thread per tile: all loads of a tile are in flight together, so more rows = more memory-level
    // parallelism, until registers cut the occupancy. One Row costs 1 + 2 W + (argument dwords) VGPRs, the cache
    // KC x part_regs. Measured on MI355X: q1_mini (7 regs/row) best at R = 4, Q1 (25 regs/row, 80 cache regs) at R = 2.
    P.R = std::max(1, std::min(4, ((all_narrow ? 72 : 144) - P.KC * part_regs) / row_regs));   // (narrow Q1: R = 3, 0.641 ms; 2: 0.666; 4: 0.659)
  }

  // ---- source
  ExprGen g(es, input);
  // tile-relative addressing: uniform 64-bit tile base (SGPRs) + 32-bit lane offset, so that a load needs one VALU
  // instruction for its address instead of a 64-bit multiply-add chain per column
  g.set_indexing(" + tb", "o", "(tb + (i64)o)");
  g.set_raw_mode(true);
  for (auto& kd : P.keys)
    if (kd.type.id == QHIP_UTF8 && es.at(kd.root).kind == QHIP_EXPR_COLUMN) g.mark_utf8_key(kd.root, kd.words);
  std::ostringstream s;
  const int KW = P.W > 0 ? P.W : 1;
  s << "struct P {\n";
  const int KC = P.KC;
  s << "  static constexpr int W = " << P.W << ";\n  static constexpr int R = " << P.R << ";\n  static constexpr int SLOT_WORDS = " << P.slot_words << ";\n";
  s << "  static constexpr int PROF = " << (env_int("QHIP_AGG_PROF", 0) != 0 ? 1 : 0) << ";   // phase timers (s_memtime) summed into status words 8..12 (measurements only)\n";
  s << "  static constexpr int PIPE = " << (env_int("QHIP_AGG_PIPE", 0) != 0 ? 1 : 0) << ";   // two register sets: the next tile's loads fly while a tile is evaluated\n";
  // rows per thread of the partitioned path's staged scatter (qh_agg_part_stage_body): what a 1 024-thread workgroup can stage
  // in LDS beside 4 096 bins' counters (160 KB per CU), at most 4; 0 = records too wide, per-lane stores
  P.part_pr = std::min(4, (int)((163840 - 4096 * 12 - 1024) / (1024 * ((P.slot_words - 1) * 8 + 2))));
  if (const char* e = getenv("QHIP_AGG_PART_PR")) P.part_pr = std::min(P.part_pr, std::max(1, atoi(e)));   // (experiments)
  s << "  static constexpr int PART_PR = " << P.part_pr << ";\n";
  // The consecutive-rows form (qh_filter_agg_body<.., CONS>): for inputs whose referenced columns are all plain (no index
  // vectors, no validity bitmaps, no Booleans, strings only as one-byte flags) and at most 8 bytes wide in the layout that is
  // streamed — a lane's RC values of a column are then adjacent bytes and load as one instruction.
  P.RC = 0;
  // QHIP_AGG_CONS: 0 never, 1 (default) when the registers allow, 2 whenever the columns allow. Measured on MI355X, one box
  // (profiles/r04_q1_consecutive_rows.txt): q1_mini (100 M rows, 9 B per row, one narrow SUM: 84-120 VGPRs) rows r*256+tid 0.225 ms,
  // consecutive rows x 4 with two register sets 0.200 ms (0.50 -> 0.56 of the HBM peak); TPC-H Q1's list at SF10 (22 B per row,
  // five SUMs x four cached keys) 0.268 ms against 0.291-0.368 ms — a quarter of the load instructions and twice the bytes in
  // flight per wavefront do not pay for 147-235 VGPRs against 127 (three or two wavefronts per SIMD instead of four). Hence the
  // rule: the cached keys' accumulators + four evaluated rows within 64 VGPRs.
  const int cons_mode = env_int("QHIP_AGG_CONS", 1);
  const bool cons_fits = KC * part_regs + 4 * row_regs <= 64;
  if (!dev_rows && (cons_mode == 2 || (cons_mode == 1 && cons_fits))) {
    bool ok = !input.empty();
    int widest = 0;
    std::vector<char> used(es.nodes.size(), 0);
    std::vector<int> todo;
    if (predicate_root >= 0) todo.push_back(predicate_root);
    for (int k = 0; k < n_groups; ++k) todo.push_back(group_roots[k]);
    for (int k = 0; k < n_aggs; ++k) todo.push_back(aggs[k].expr);
    while (!todo.empty()) {
      const int k = todo.back();
      todo.pop_back();
      if (k < 0 || k >= (int)es.nodes.size() || used[(size_t)k]) continue;
      used[(size_t)k] = 1;
      todo.push_back(es.nodes[(size_t)k].left); todo.push_back(es.nodes[(size_t)k].right); todo.push_back(es.nodes[(size_t)k].third);
    }
    for (size_t c = 0; c < es.nodes.size(); ++c) {
      const ENode& nd = es.nodes[c];
      if (!used[c] || nd.kind != QHIP_EXPR_COLUMN || nd.type.id == QHIP_NULL) continue;
      const InputCol& ic = input[(size_t)nd.column];
      if (ic.indirect || nd.nullable || nd.type.id == QHIP_BOOL) ok = false;
      else if (nd.type.id == QHIP_UTF8) { if (!ic.utf8_fixed1) ok = false; else widest = std::max(widest, 1); }
      else {
        const int nb = (nd.type.id == QHIP_DECIMAL128 || (nd.type.id == QHIP_INT64 && ic.narrow_bytes == 4)) ? ic.narrow_bytes : 0;
        const int w = nb ? nb : dtype_width(nd.type);
        if (w <= 0 || w > 8) ok = false;
        widest = std::max(widest, w);
      }
    }
    if (ok && widest > 0) P.RC = std::max(1, std::min(8, env_int("QHIP_AGG_CONS_R", 4)));
  }
  {
    // how the sorted-run kernel orders two key words (any total order makes "non-decreasing" imply "equal keys adjacent"; these
    // make the usual sort orders — signed integers, bytewise strings — pass): Utf8 words compare byte-swapped (the bytes are packed
    // little-endian), signed integer words with the sign bit flipped
    unsigned long long swap_mask = 0, sign_mask = 0;
    for (auto& kd : P.keys) {
      for (int w2 = 0; w2 < kd.words; ++w2) {
        if (kd.type.id == QHIP_UTF8) swap_mask |= 1ull << (kd.word_off + w2);
        else if (kd.words == 1 && kd.type.id != QHIP_UINT8) sign_mask |= 1ull << (kd.word_off + w2);
      }
    }
    s << "  static constexpr unsigned long long KEY_SWAP_MASK = " << swap_mask << "ULL, KEY_SIGN_MASK = " << sign_mask << "ULL;\n";
  }
  s << "  static constexpr int RC = " << std::max(1, P.RC) << ";\n";
  s << "  static constexpr int CSB = " << std::max(1, std::min(4, env_int("QHIP_AGG_CONS_SB", 4))) << ";\n";
  s << "  static constexpr int CPIPE = " << (env_int("QHIP_AGG_CONS_PIPE", 1) != 0 ? 1 : 0) << ";\n";
  s << "  static constexpr int KC = " << KC << ";\n";
  s << "  struct Row {\n    bool pass;\n    u64 key[" << KW << "];\n";
  for (size_t a = 0; a < P.args.size(); ++a) {
    const ArgDesc& ad = P.args[a];
    bool value_needed = false;
    for (auto& c : P.cells) if (c.arg == (int)a && c.kind != CELL_CNT) value_needed = true;
    if (value_needed) s << "    " << (arg64[a] ? std::string("i64") : ExprGen::ctype(ad.type)) << " a" << a << ";\n";
    if (ad.nullable) s << "    bool h" << a << ";\n";
  }
  s << "  };\n";
  s << "  struct Part {\n";
  for (size_t c = 0; c < P.cells.size(); ++c) {
    const char* t = "u64";
    if (P.cells[c].kind == CELL_SUM_I128) t = "i128";
    else if (P.cells[c].kind == CELL_SUM_F64) t = "double";
    else if (P.cells[c].kind == CELL_MAXORD128) t = "u128";
    s << "    " << t << " c" << c << ";\n";
  }
  s << "  };\n";
  // eval
  // two phases per row: load() is branch-free and only issues the row's loads into `Raw w` (the kernel calls it for all R
  // rows of a tile first, so their memory latencies overlap); eval() computes predicate, keys and aggregate arguments
  // from w and may contain control flow (string compares, the tiered decimal multiply) without serialising any load
  std::ostringstream ev;
  std::string code;
  if (predicate_root >= 0) {
    g.emit(predicate_root, code);
    ev << code;
    ev << "    r.pass = " << g.ok(predicate_root) << " && " << g.val(predicate_root) << ";\n";
  } else {
    ev << "    r.pass = true;\n";
  }
  code.clear();
  emit_key_words(g, es, P.keys, P.null_mask_word, "r.key", code, nullptr);
  ev << code;
  for (size_t a = 0; a < P.args.size(); ++a) {
    code.clear();
    g.emit(P.args[a].root, code);
    ev << code;
    bool value_needed = false;
    for (auto& c : P.cells) if (c.arg == (int)a && c.kind != CELL_CNT) value_needed = true;
    if (value_needed) ev << "    r.a" << a << " = " << (arg64[a] ? "(i64)(u64)(u128)" : "") << g.val(P.args[a].root) << ";\n";
    if (P.args[a].nullable) ev << "    r.h" << a << " = " << g.ok(P.args[a].root) << ";\n";
  }
  s << "  struct Raw {\n" << g.raw_fields << "    int unused_;\n  };\n";
  s << "  __device__ static __forceinline__ void load(const KArgs& a, const i64 tb, const u32 o, Raw& w) {\n" << g.load_code << "    w.unused_ = 0;\n  }\n";
  s << "  __device__ static __forceinline__ void eval(const KArgs& a, const Raw& w, Row& r, u32& err) {\n" << ev.str() << "  }\n";
  // part_init
  s << "  __device__ static __forceinline__ void part_init(Part& p) {\n";
  for (size_t c = 0; c < P.cells.size(); ++c) s << "    p.c" << c << " = 0;\n";
  s << "  }\n";
  // part_add / acc_add
  // ROWS = false: the caller knows the row count from ballot popcounts (part_set_rows) and saves the per-lane adds.
  // Sums are written as `if (take) acc += v` so that the compiler can run the adds under the EXEC mask instead of
  // paying a v_cndmask per dword on top of every add. Acc = the lane-private accumulators of the hot-key cache (and of the
  // ungrouped aggregate): Part with the narrow SUM cells in 64 bits.
  for (int narrow = 0; narrow < 2; ++narrow) {
    if (narrow) {
      s << "  struct Acc {\n";
      for (size_t c = 0; c < P.cells.size(); ++c) {
        const char* t = "u64";
        if (P.cells[c].kind == CELL_SUM_I128) t = cell_acc64(P.cells[c]) ? "i64" : "i128";
        else if (P.cells[c].kind == CELL_SUM_F64) t = "double";
        else if (P.cells[c].kind == CELL_MAXORD128) t = "u128";
        s << "    " << t << " c" << c << ";\n";
      }
      s << "  };\n";
      s << "  __device__ static __forceinline__ void acc_init(Acc& p) {\n";
      for (size_t c = 0; c < P.cells.size(); ++c) s << "    p.c" << c << " = 0;\n";
      s << "  }\n";
      s << "  __device__ static __forceinline__ void acc_to_part(const Acc& a, Part& p) {\n";
      for (size_t c = 0; c < P.cells.size(); ++c) s << "    p.c" << c << " = " << (cell_acc64(P.cells[c]) ? "(i128)" : "") << "a.c" << c << ";\n";
      s << "  }\n";
    }
    s << "  template <bool ROWS> __device__ static __forceinline__ void " << (narrow ? "acc_add(Acc& p" : "part_add(Part& p") << ", const Row& r, const bool m) {\n";
    for (size_t c = 0; c < P.cells.size(); ++c) {
      const CellDesc& cd = P.cells[c];
      const std::string C = "p.c" + std::to_string(c);
      if (cd.kind == CELL_ROWS) { s << "    if (ROWS) " << C << " += m ? 1ULL : 0ULL;\n"; continue; }
      const ArgDesc& ad = P.args[(size_t)cd.arg];
      const std::string A = "r.a" + std::to_string(cd.arg);
      const std::string take = ad.nullable ? "(m && r.h" + std::to_string(cd.arg) + ")" : std::string("m");
      switch (cd.kind) {
        case CELL_SUM_I128:
          if (narrow && cell_acc64(cd)) s << "    if (" << take << ") " << C << " = (i64)((u64)" << C << " + (u64)" << A << ");\n";
          else s << "    if (" << take << ") " << C << " = (i128)((u128)" << C << " + (u128)(i128)" << A << ");\n";
          break;
        case CELL_SUM_U64: s << "    if (" << take << ") " << C << " += (u64)" << A << ";\n"; break;
        case CELL_SUM_F64: s << "    if (" << take << ") " << C << " += (double)" << A << ";\n"; break;
        case CELL_CNT: s << "    " << C << " += " << take << " ? 1ULL : 0ULL;\n"; break;
        case CELL_MAXORD64: {
          const std::string o = std::string(cd.is_min ? "~" : "") + ord64(ad.type, A);
          s << "    { const u64 o = " << take << " ? " << o << " : 0ULL; " << C << " = o > " << C << " ? o : " << C << "; }\n";
          break;
        }
        case CELL_MAXORD128: {
          const std::string o = std::string(cd.is_min ? "~" : "") + "((u128)(i128)" + A + " ^ ((u128)1 << 127))";
          s << "    { const u128 o = " << take << " ? " << o << " : (u128)0; " << C << " = o > " << C << " ? o : " << C << "; }\n";
          break;
        }
      }
    }
    s << "  }\n";
  }
  // part_reduce
  s << "  __device__ static __forceinline__ void part_set_rows(Part& p, const u64 n) { p.c0 = n; }\n";
  s << "  template <bool ROWS> __device__ static __forceinline__ void part_reduce(Part& p) {\n";
  for (size_t c = 0; c < P.cells.size(); ++c) {
    const std::string C = "p.c" + std::to_string(c);
    switch (P.cells[c].kind) {
      case CELL_ROWS: s << "    if (ROWS) " << C << " = qh_wave_sum_u64(" << C << ");\n"; break;
      case CELL_CNT: case CELL_SUM_U64: s << "    " << C << " = qh_wave_sum_u64(" << C << ");\n"; break;
      case CELL_SUM_I128: s << "    " << C << " = qh_wave_sum_i128(" << C << ");\n"; break;
      case CELL_SUM_F64: s << "    " << C << " = qh_wave_sum_f64(" << C << ");\n"; break;
      case CELL_MAXORD64: s << "    " << C << " = qh_wave_max_u64(" << C << ");\n"; break;
      case CELL_MAXORD128: s << "    " << C << " = qh_wave_max_u128(" << C << ");\n"; break;
    }
  }
  s << "  }\n";
  // slot_update
  s << "  template <class M> __device__ static __forceinline__ void slot_update(u64* slot, const Part& p) {\n";
  s << "    u64* cell = slot + " << 1 + P.W << ";\n";
  for (size_t c = 0; c < P.cells.size(); ++c) {
    const CellDesc& cd = P.cells[c];
    const std::string C = "p.c" + std::to_string(c);
    const std::string at = "cell + " + std::to_string(cd.off);
    switch (cd.kind) {
      case CELL_ROWS: case CELL_CNT: case CELL_SUM_U64: s << "    qh_acc_add_u64<M>(" << at << ", " << C << ");\n"; break;
      case CELL_SUM_I128: s << "    qh_acc_add_i128<M>(" << at << ", " << C << ");\n"; break;
      case CELL_SUM_F64: s << "    qh_acc_add_f64<M>(" << at << ", " << C << ");\n"; break;
      case CELL_MAXORD64: s << "    if (" << C << ") qh_acc_max_u64<M>(" << at << ", " << C << ");\n"; break;
      case CELL_MAXORD128: s << "    if (" << C << ") qh_acc_max_u128<M>(" << at << ", " << C << ");\n"; break;
    }
  }
  s << "  }\n";
  // part_from_slot / part_to_slot: a slot-shaped record (plain memory) <-> Part; slot_merge: LDS slot (plain reads, after
  // the workgroup barrier) -> HBM slot
  s << "  __device__ static __forceinline__ void part_from_slot(const u64* ls, Part& q) {\n";
  s << "    const u64* cell = ls + " << 1 + P.W << ";\n";
  for (size_t c = 0; c < P.cells.size(); ++c) {
    const CellDesc& cd = P.cells[c];
    const std::string at = "cell[" + std::to_string(cd.off) + "]";
    const std::string at1 = "cell[" + std::to_string(cd.off + 1) + "]";
    switch (cd.kind) {
      case CELL_SUM_I128: s << "    q.c" << c << " = qh_mk128(" << at << ", (i64)" << at1 << ");\n"; break;
      case CELL_SUM_F64: s << "    q.c" << c << " = qh_f64(" << at << ");\n"; break;
      case CELL_MAXORD128: s << "    q.c" << c << " = ((u128)" << at1 << " << 64) | (u128)" << at << ";\n"; break;
      default: s << "    q.c" << c << " = " << at << ";\n"; break;
    }
  }
  s << "  }\n";
  s << "  __device__ static __forceinline__ void part_to_slot(u64* ls, const Part& q) {\n";
  s << "    u64* cell = ls + " << 1 + P.W << ";\n";
  for (size_t c = 0; c < P.cells.size(); ++c) {
    const CellDesc& cd = P.cells[c];
    const std::string at = "cell[" + std::to_string(cd.off) + "]";
    const std::string at1 = "cell[" + std::to_string(cd.off + 1) + "]";
    const std::string C = "q.c" + std::to_string(c);
    switch (cd.kind) {
      case CELL_SUM_I128: case CELL_MAXORD128: s << "    " << at << " = (u64)(u128)" << C << "; " << at1 << " = (u64)((u128)" << C << " >> 64);\n"; break;
      case CELL_SUM_F64: s << "    " << at << " = (u64)__double_as_longlong(" << C << ");\n"; break;
      default: s << "    " << at << " = " << C << ";\n"; break;
    }
  }
  s << "  }\n";
  s << "  __device__ static __forceinline__ void slot_merge(u64* gs, const u64* ls) {\n    Part q;\n    part_from_slot(ls, q);\n";
  s << "    slot_update<MemHbm>(gs, q);\n  }\n";
  s << "};\n";
  P.kernel_name = "qk_filter_agg";
  {
    const int waves = env_int("QHIP_AGG_WAVES", 0);
    const std::string wattr = waves > 0 ? "__attribute__((amdgpu_waves_per_eu(" + std::to_string(waves) + "))) " : std::string();
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) " << wattr << "void qk_filter_agg(KArgs a, AggLaunch L) { qh_filter_agg_body<P" << dr << ">(a, L); }\n";
    // 1024-thread workgroups, one per CU, sharing one LDS table of up to 128 KB (mid-sized many-group inputs, agg.cpp)
    if (P.W > 0)
      s << "extern \"C\" __global__ __launch_bounds__(1024) void qk_filter_agg_wide(KArgs a, AggLaunch L) { qh_filter_agg_body<P, " << (dev_rows ? "true" : "false")
        << ", 1024>(a, L); }\n";
    // ... an input whose equal keys are adjacent (checked on the device): runs instead of a hash table
    if (P.W > 0 && predicate_root < 0) {
      P.has_runs = true;
      s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_agg_runs(KArgs a, RunsLaunch L) { qh_agg_runs_body<P" << dr << ">(a, L); }\n";
    }
    // ... a lane owning RC consecutive rows (narrow plain layouts: wide loads)
    if (P.RC > 0)
      s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) " << wattr << "void qk_filter_agg_cons(KArgs a, AggLaunch L) { qh_filter_agg_body<P, false, QH_BLOCK, false, true>(a, L); }\n";
    // ... and over an input pre-partitioned by key hash, one part per workgroup (AggLaunch::part_runs; agg.cpp)
    if (P.W > 0 && !dev_rows)
      s << "extern \"C\" __global__ __launch_bounds__(1024) void qk_filter_agg_parts(KArgs a, AggLaunch L) { qh_filter_agg_body<P, false, 1024, true>(a, L); }\n";
  }
  if (P.W > 0) {
    // the partitioned path for many groups on a big input (same policy, three more entry points of the same module)
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_agg_part_hist(KArgs a, PartLaunch L) { qh_agg_part_body<P, false" << dr << ">(a, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_agg_part_scatter(KArgs a, PartLaunch L) { qh_agg_part_body<P, true" << dr << ">(a, L); }\n";
    if (P.part_pr > 0)
      s << "extern \"C\" __global__ __launch_bounds__(QH_STAGE_BLOCK) void qk_agg_part_stage(KArgs a, PartLaunch L) { qh_agg_part_stage_body<P" << dr << ">(a, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_agg_reduce(ReduceLaunch R, AggLaunch L) { qh_agg_reduce_body<P>(R, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(1024) void qk_agg_reduce_wide(ReduceLaunch R, AggLaunch L) { qh_agg_reduce_body<P, 1024>(R, L); }\n";
  }
  P.source = s.str();
  P.bind = g.bind;
}

void plan_predicate_mask(const ExprSet& es, const std::vector<InputCol>& input, int root, MaskPlan& out) {
  if (root < 0 || root >= (int)es.nodes.size()) fail(QHIP_INVALID_ARGUMENT, "predicate root out of range");
  if (es.at(root).type.id != QHIP_BOOL)
    fail(QHIP_INVALID_ARGUMENT, "filter predicate must be Boolean, got " + dtype_name(es.at(root).type));
  ExprGen g(es, input);
  // the mask kernel is software-pipelined over its tiles: a row's column loads (load(): branch-free, tile-relative
  // addressing) are issued two tiles before the predicate is evaluated from them (pred())
  g.set_indexing(" + tb", "o", "(tb + (i64)o)");
  g.set_raw_mode(true);
  std::string code;
  g.emit(root, code);
  std::ostringstream s;
  out.mask_r = std::max(1, std::min(16, env_int("QHIP_MASK_R", 4)));
  s << "struct P {\n  static constexpr int MASK_R = " << out.mask_r << ";\n";
  s << "  struct Raw {\n" << g.raw_fields << "    int unused_;\n  };\n";
  s << "  __device__ static __forceinline__ void load(const KArgs& a, const i64 tb, const u32 o, Raw& w) {\n" << g.load_code << "    w.unused_ = 0;\n  }\n";
  s << "  __device__ static __forceinline__ bool pred(const KArgs& a, const Raw& w, u32& err) {\n" << code;
  s << "    return " << g.ok(root) << " && " << g.val(root) << ";\n  }\n};\n";
  s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_pred_mask(KArgs a, u64* mask, u32* wave_count, u32* status) { "
       "qh_pred_mask_body<P>(a, mask, wave_count, status); }\n";
  out.source = s.str();
  out.kernel_name = "qk_pred_mask";
  out.bind = g.bind;
}

void plan_keys(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, KeysPlan& out, int predicate_root,
               int kernel, bool dev_rows, int n_parts) {
  out = KeysPlan();
  layout_keys(es, input, roots, n, false, out.keys, out.W, out.null_mask_word);
  if (out.W > 8) fail(QHIP_UNSUPPORTED, "join key wider than 8 words");
  ExprGen g(es, input);
  const bool raw = kernel == KEYS_KERNEL_PROBE || kernel == KEYS_KERNEL_DENSE_PROBE || kernel == KEYS_KERNEL_PARTITION;
  if (kernel == KEYS_KERNEL_PARTITION && (n_parts < 1 || n_parts > 255)) fail(QHIP_INVALID_ARGUMENT, "partition kernel: 1 .. 255 parts");
  if (raw) {
    // the probe kernel is software-pipelined over its tiles: a row's column loads are issued one stage (load(), branch-free,
    // tile-relative addressing) before its filter / key words are computed from them (keys())
    g.set_indexing(" + tb", "o", "(tb + (i64)o)");
    g.set_raw_mode(true);
    for (auto& kd : out.keys)
      if (kd.type.id == QHIP_UTF8 && es.at(kd.root).kind == QHIP_EXPR_COLUMN) g.mark_utf8_key(kd.root, kd.words);
  }
  std::string code, all;
  if (predicate_root >= 0) {
    // scan filter fused into the key evaluation: a row the predicate rejects gets an invalid key, i.e. it is never
    // inserted into / probed against the join table — the filtered batch is never materialised
    if (es.at(predicate_root).type.id != QHIP_BOOL)
      fail(QHIP_INVALID_ARGUMENT, "filter predicate must be Boolean, got " + dtype_name(es.at(predicate_root).type));
    g.emit(predicate_root, code);
  }
  emit_key_words(g, es, out.keys, false, "k", code, &all);
  // (the exchange's partition kernel tells "rejected by the filter" (the row is dropped) from "NULL key" (the row travels, hashed
  // as all-zero key words): bit 1 / bit 0 of what keys() returns)
  if (kernel == KEYS_KERNEL_PARTITION)
    all = "(" + (predicate_root >= 0 ? "(" + g.ok(predicate_root) + " && " + g.val(predicate_root) + ")" : std::string("true")) + " ? 2u : 0u) | ((" + all + ") ? 1u : 0u)";
  else
  if (predicate_root >= 0) all = "(" + g.ok(predicate_root) + " && " + g.val(predicate_root) + ") && " + all;
  std::ostringstream s;
  s << "struct P {\n  static constexpr int W = " << out.W << ";\n";
  if (raw) {
    // (dense probe: a lane owns R consecutive rows, 16 bytes of an 8-byte key column at R = 2 — and of its 4-byte narrow copy at
    // R = 4: Q3's lineitem probe 112 -> 104 us)
    bool narrow_key = false;
    for (auto& ic : input) if (ic.narrow_bytes == 4 && ic.type.id == QHIP_INT64 && !ic.indirect) narrow_key = true;
    out.probe_r = std::max(1, std::min(8, kernel == KEYS_KERNEL_DENSE_PROBE ? env_int("QHIP_DENSE_PROBE_R", narrow_key ? 4 : 2) : env_int("QHIP_PROBE_R", 4)));
    s << "  static constexpr int PROBE_R = " << out.probe_r << ";\n";
    if (kernel == KEYS_KERNEL_PARTITION)
      s << "  static constexpr int NP = " << n_parts << ";\n  static constexpr int PART_R = " << std::max(1, std::min(8, env_int("QHIP_PART_R", 4))) << ";\n";
    s << "  struct Raw {\n" << g.raw_fields << "    int unused_;\n  };\n";
    s << "  __device__ static __forceinline__ void load(const KArgs& a, const i64 tb, const u32 o, Raw& w) {\n" << g.load_code << "    w.unused_ = 0;\n  }\n";
    s << "  __device__ static __forceinline__ " << (kernel == KEYS_KERNEL_PARTITION ? "u32" : "bool") << " keys(const KArgs& a, const Raw& w, u64* k, u32& err) {\n" << code;
  } else {
    s << "  __device__ static __forceinline__ bool keys(const KArgs& a, const i64 i, u64* k, u32& err) {\n" << code;
  }
  s << "    return " << all << ";\n  }\n};\n";
  if (kernel == KEYS_KERNEL_PROBE) {
    const int waves = env_int("QHIP_PROBE_WAVES", out.probe_r >= 4 ? 5 : out.probe_r == 3 ? 6 : 8);
    // two entry points: the region layout of the LDS-staged build (the rule) and the one-table legacy layout. Five waves per
    // SIMD (<= 96 VGPRs): the three tiles in flight need ~90; at 98 the kernel fell to four waves and ran 10-20 % slower
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) __attribute__((amdgpu_waves_per_eu(" << waves << "))) void qk_join_probe(KArgs a, ProbeLaunch L) { qh_join_probe_body<P, true>(a, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) __attribute__((amdgpu_waves_per_eu(" << waves << "))) void qk_join_probe_onetable(KArgs a, ProbeLaunch L) { qh_join_probe_body<P, false>(a, L); }\n";
  } else if (kernel == KEYS_KERNEL_DENSE_PROBE) {
    // the dense stages keep ~half the state of the hashed ones (no key words / filter masks / slot images across stages)
    const int waves = env_int("QHIP_DENSE_PROBE_WAVES", 0);   // (0: no occupancy attribute — the kernels need <= 64 VGPRs anyway, and pinning 8 waves caps the SGPRs at 80: ~30 spills)
    const std::string wattr = waves > 0 ? "__attribute__((amdgpu_waves_per_eu(" + std::to_string(waves) + "))) " : std::string();
    // qk_join_probe_dense: any table (rows r * 64 + lane, clamped per row); _wide: a lane owns R consecutive rows (16-byte
    // column loads; tables of at least one tile); _lds / _hybrid: 1 024-thread workgroups with the bitmap (its first
    // L.lds_words words) staged in LDS, wide loads
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) " << wattr << "void qk_join_probe_dense(KArgs a, ProbeLaunch L) { qh_join_probe_dense_body<P, 0, false>(a, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) " << wattr << "void qk_join_probe_dense_wide(KArgs a, ProbeLaunch L) { qh_join_probe_dense_body<P, 0, true>(a, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(1024) void qk_join_probe_dense_lds(KArgs a, ProbeLaunch L) { qh_join_probe_dense_body<P, 1, false>(a, L); }\n";
    s << "extern \"C\" __global__ __launch_bounds__(1024) void qk_join_probe_dense_hybrid(KArgs a, ProbeLaunch L) { qh_join_probe_dense_body<P, 2, false>(a, L); }\n";
  } else if (kernel == KEYS_KERNEL_PARTITION) {
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_part_ids(KArgs a, PartIdsLaunch L) { qh_part_ids_body<P, " << (dev_rows ? "true" : "false") << ", false>(a, L); }\n";
    // (tables of at least one tile: a lane owns PART_R consecutive rows — 16-byte column loads, one 4-byte store of its part bytes)
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_part_ids_wide(KArgs a, PartIdsLaunch L) { qh_part_ids_body<P, " << (dev_rows ? "true" : "false") << ", true>(a, L); }\n";
  } else if (kernel == KEYS_KERNEL_DENSE_BUILD) {
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_join_dense_build(KArgs a, DenseBuildLaunch L) { qh_join_dense_build_body<P" << (dev_rows ? ", true" : "") << ">(a, L); }\n";
  } else if (kernel == KEYS_KERNEL_SCATTER)
    s << "extern \"C\" __global__ __launch_bounds__(QH_SCATTER_BLOCK) void qk_join_scatter(KArgs a, ScatterLaunch L) { qh_join_scatter_body<P" << (dev_rows ? ", true" : "") << ">(a, L); }\n";
  else
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_eval_keys(KArgs a, u64* keys, u64* keyvalid, u32* status) { "
         "qh_eval_keys_body<P>(a, keys, keyvalid, status); }\n";
  out.source = s.str();
  out.kernel_name = kernel == KEYS_KERNEL_PROBE ? "qk_join_probe" : kernel == KEYS_KERNEL_SCATTER ? "qk_join_scatter" :
                    kernel == KEYS_KERNEL_DENSE_PROBE ? "qk_join_probe_dense" : kernel == KEYS_KERNEL_DENSE_BUILD ? "qk_join_dense_build" :
                    kernel == KEYS_KERNEL_PARTITION ? "qk_part_ids" : "qk_eval_keys";
  out.bind = g.bind;
}

// ---------------------------------------------------------------- exchange, pass 2
void plan_part_scatter(const std::vector<int>& widths, const std::vector<char>& indirect, int n_parts, bool dev_rows, PartScatterPlan& out, bool unstable) {
  out = PartScatterPlan();
  if (widths.empty() || widths.size() > 8) fail(QHIP_INVALID_ARGUMENT, "partition scatter: 1 .. 8 columns per launch");
  auto T = [](int w) { return w == 1 ? "u8" : w == 2 ? "u16" : (w == 4 || w == 0) ? "u32" : w == 8 ? "u64" : "qh_v4u"; };
  int maxw = 4;
  for (int w : widths) maxw = std::max(maxw, w);
  const int R = std::max(1, std::min(8, env_int("QHIP_PART_SCATTER_R", 4)));
  out.rows_per_lane = R;
  std::ostringstream s;
  // measurement switches: stores straight from the registers (no LDS staging: every lane writes its own value to its place in
  // the part's run), one tile at a time instead of the two-stage loop, the stores left out altogether (results wrong)
  const bool direct = env_int("QHIP_PART_SCATTER_DIRECT", 0) != 0, no_stores = env_int("QHIP_PART_SCATTER_NOSTORE", 0) != 0;
  const bool nt_store = env_int("QHIP_PART_SCATTER_NT", 0) != 0;
  const int tb = std::max(256, std::min(1024, env_int("QHIP_PART_SCATTER_TB", 512))) / 64 * 64;
  s << "struct P {\n  static constexpr int TB = " << tb << ";\n  static constexpr int NC = " << widths.size() << ", R = " << R << ", NPT = " << (unstable ? -1 : n_parts <= 8 ? 8 : n_parts <= 16 ? 16 : 0)
    << ", MAXW = " << maxw << ", DIRECT = " << (direct ? 1 : 0) << ", PIPE = " << (env_int("QHIP_PART_SCATTER_PIPE", 1) != 0 ? 1 : 0) << ";\n";
  s << "  struct Vals {\n";
  for (size_t c = 0; c < widths.size(); ++c) s << "    " << T(widths[c]) << " c" << c << "[R];\n";
  s << "  };\n";
  s << "  __device__ static __forceinline__ void load(const PartScatterLaunch& L, const i64 tb, const i64 last, const int lane, Vals& v) {\n";
  s << "#pragma unroll\n    for (int r = 0; r < R; ++r) {\n      i64 row = tb + r * 64 + lane;\n      row = row < last ? row : last - 1;\n";
  for (size_t c = 0; c < widths.size(); ++c) {
    const std::string t = T(widths[c]);
    if (widths[c] == 0) s << "      v.c" << c << "[r] = (u32)row;\n";
    else if (indirect[c]) s << "      v.c" << c << "[r] = ((const " << t << "*)L.src[" << c << "])[L.idx[" << c << "][row]];\n";
    else s << "      v.c" << c << "[r] = __builtin_nontemporal_load((const " << t << "*)L.src[" << c << "] + row);\n";
  }
  s << "    }\n  }\n";
  s << "  __device__ static __forceinline__ void move(const PartScatterLaunch& L, const Vals& v, const i64 tb, const u32* id, const u32* pos, const u32* dst,\n"
       "                                              const u32 total, u8* sval, const u32* sdst, const int lane) {\n";
  // every store is UNCONDITIONAL (a lane beyond the tile's rows stores the tile's last row again — same value, same address — and a
  // tile without rows stores into the launch's scratch line): a store behind a branch is one the compiler cannot count, and it
  // then waits for ALL older stores' acknowledgements before the next tile's loads (s_waitcnt vmcnt(loads only))
  if (!direct) s << "    u32 d[R], jj[R];\n#pragma unroll\n    for (int k = 0; k < R; ++k) { const u32 j = (u32)k * 64u + (u32)lane; jj[k] = j < total ? j : (total ? total - 1u : 0u); d[k] = total ? sdst[jj[k]] : (u32)lane; }\n";
  for (size_t c = 0; c < widths.size(); ++c) {
    const std::string t = T(widths[c]);
    if (direct) {
      if (!no_stores && nt_store) s << "#pragma unroll\n    for (int r = 0; r < R; ++r) if (id[r] != 0xFFu) __builtin_nontemporal_store(v.c" << c << "[r], (" << t << "*)L.out[" << c << "] + dst[r]);\n";
      else if (!no_stores) s << "#pragma unroll\n    for (int r = 0; r < R; ++r) if (id[r] != 0xFFu) ((" << t << "*)L.out[" << c << "])[dst[r]] = v.c" << c << "[r];\n";
      else s << "#pragma unroll\n    for (int r = 0; r < R; ++r) if (id[r] == 0xFEu) ((" << t << "*)L.out[" << c << "])[dst[r]] = v.c" << c << "[r];\n";
      continue;
    }
    s << "#pragma unroll\n    for (int r = 0; r < R; ++r) if (id[r] != 0xFFu) ((" << t << "*)sval)[pos[r]] = v.c" << c << "[r];\n";
    s << "    asm volatile(\"\" ::: \"memory\");\n";
    if (env_int("QHIP_PART_SCATTER_TRASH", 0))   // (measurement: every store lands in the 1 KB scratch line — the store path without HBM writes)
      s << "    { " << t << "* const o = (" << t << "*)L.trash;\n#pragma unroll\n    for (int k = 0; k < R; ++k) d[k] &= 63u;\n";
    else
    s << "    { " << t << "* const o = total ? (" << t << "*)L.out[" << c << "] : (" << t << "*)L.trash;\n";
    s << "#pragma unroll\n    for (int k = 0; k < R; ++k) { const " << t << " x = ((const " << t << "*)sval)[jj[k]]; "
      << (no_stores ? "if (lane > 64) " : "") << (nt_store ? "__builtin_nontemporal_store(x, o + d[k])" : "o[d[k]] = x") << "; } }\n";
    s << "    asm volatile(\"\" ::: \"memory\");\n";
  }
  s << "  }\n";
  // workgroup form: the tile of TB * R rows is laid out in LDS column by column and written out by all threads; every store is
  // unconditional here too (see above)
  s << "  __device__ static __forceinline__ void move_wg(const PartScatterLaunch& L, const Vals& v, const u32* id, const u32* pos, u8* sval, const u32* sdst,\n"
       "                                                 const u32* stotal, const int tid) {\n";
  s << "    u32 d[R], jj[R];\n    u32 total = 0;\n";
  for (size_t c = 0; c < widths.size(); ++c) {
    const std::string t = T(widths[c]);
    s << "#pragma unroll\n    for (int r = 0; r < R; ++r) if (id[r] != 0xFFu) ((" << t << "*)sval)[pos[r]] = v.c" << c << "[r];\n";
    s << "    __syncthreads();\n";
    if (c == 0)
      s << "    total = *stotal;\n#pragma unroll\n    for (int k = 0; k < R; ++k) { const u32 j = (u32)k * (u32)TB + (u32)tid; jj[k] = j < total ? j : (total ? total - 1u : 0u); "
           "d[k] = total ? sdst[jj[k]] : (u32)(tid & 63); }\n";
    s << "    { " << t << "* const o = total ? (" << t << "*)L.out[" << c << "] : (" << t << "*)L.trash;\n";
    s << "#pragma unroll\n    for (int k = 0; k < R; ++k) { const " << t << " x = ((const " << t << "*)sval)[jj[k]]; "
      << (no_stores ? "if (tid > 4096) " : "") << (nt_store ? "__builtin_nontemporal_store(x, o + d[k])" : "o[d[k]] = x") << "; } }\n";
    s << "    __syncthreads();\n";
  }
  s << "  }\n};\n";
  s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_part_scatter(PartScatterLaunch L) { qh_part_scatter_body<P" << (dev_rows ? ", true" : "")
    << ">(L); }\n";
  s << "extern \"C\" __global__ __launch_bounds__(" << tb << ") void qk_part_scatter_wg(PartScatterLaunch L) { qh_part_scatter_wg_body<P" << (dev_rows ? ", true" : "")
    << ">(L); }\n";
  out.wg_threads = tb;
  out.source = s.str();
  out.kernel_name = "qk_part_scatter";
}

// ---------------------------------------------------------------- projection
void plan_projection(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, ProjectionPlan& out) {
  out = ProjectionPlan();
  if (n > 24) fail(QHIP_UNSUPPORTED, "more than 24 computed projection expressions");
  ExprGen g(es, input);
  std::ostringstream body;
  for (int k = 0; k < n; ++k) {
    const ENode& nd = es.at(roots[k]);
    ProjOutDesc od;
    od.root = roots[k]; od.type = nd.type; od.nullable = nd.nullable;
    std::string code;
    g.emit(od.root, code);
    body << code;
    const std::string okx = g.ok(od.root);
    if (nd.type.id == QHIP_NULL)
      fail(QHIP_UNSUPPORTED, "projection of a computed " + dtype_name(nd.type) + " expression is not accelerated");
    if (nd.type.id == QHIP_UTF8) {
      // a computed string (CASE over literals / columns, a literal): pass 0 stores its length, the host scans the lengths into
      // offsets, pass 1 copies the bytes (both passes evaluate the same generated code)
      out.has_utf8 = true;
      const std::string lenx = std::string(od.nullable ? "(" + okx + " ? " : "(") + g.len(od.root) + (od.nullable ? " : 0)" : ")");
      body << "    if (MODE == 0 && inb) ((int*)o.v[" << k << "])[row] = " << lenx << ";\n";
      body << "    if (MODE == 1 && inb" << (od.nullable ? " && " + okx : std::string("")) << ") { u8* dst = o.d[" << k << "] + ((const int*)o.v[" << k
           << "])[row]; const u8* src = " << g.ptr(od.root) << "; const int nb = " << g.len(od.root) << "; for (int b = 0; b < nb; ++b) dst[b] = src[b]; }\n";
    } else if (nd.type.id == QHIP_BOOL) {
      body << "    if (MODE == 0) { const u64 m = qh_ballot(inb && " << (od.nullable ? okx + " && " : std::string("")) << g.val(od.root) << "); if (lane == 0) ((u64*)o.v[" << k
           << "])[j] = m; }\n";
    } else {
      const std::string T = ExprGen::ctype(nd.type);
      body << "    if (MODE == 0 && inb) ((" << T << "*)o.v[" << k << "])[row] = " << (od.nullable ? okx + " ? " : std::string("")) << g.val(od.root)
           << (od.nullable ? " : (" + T + ")0" : std::string("")) << ";\n";
    }
    if (od.nullable) body << "    if (MODE == 0) { const u64 m = qh_ballot(inb && " << okx << "); if (lane == 0) o.n[" << k << "][j] = m; }\n";
    out.outs.push_back(od);
  }
  std::ostringstream s;
  s << "struct P {\n";
  s << "  template <int MODE> __device__ static __forceinline__ void row(const KArgs& a, const ProjOut& o, const i64 i, const i64 row, const bool inb, const i64 j, "
       "const int lane, u32& err) {\n" << body.str() << "  }\n};\n";
  s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_project(KArgs a, ProjOut o, u32* status) { qh_project_body<P, 0>(a, o, status); }\n";
  if (out.has_utf8)
    s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_project_copy(KArgs a, ProjOut o, u32* status) { qh_project_body<P, 1>(a, o, status); }\n";
  out.source = s.str();
  out.kernel_name = "qk_project";
  out.bind = g.bind;
}

// ---------------------------------------------------------------- sort key images
void plan_sort_keys(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, SortKeysPlan& out) {
  out = SortKeysPlan();
  if (n > 32) fail(QHIP_UNSUPPORTED, "more than 32 sort keys");
  ExprGen g(es, input);
  std::ostringstream body;
  int off = 0;
  for (int k = 0; k < n; ++k) {
    const ENode& nd = es.at(roots[k]);
    SortKeyDesc kd;
    kd.root = roots[k]; kd.type = nd.type; kd.nullable = nd.nullable; kd.word_off = off; kd.words = 1; kd.top_bits = 64; kd.column = -1;
    std::string code;
    g.emit(kd.root, code);
    body << code;
    const std::string v = g.val(kd.root), okx = g.ok(kd.root);
    auto store = [&](int w, const std::string& image) {
      body << "    img[" << off + w << "] = " << (kd.nullable ? okx + " ? " : std::string("")) << image << (kd.nullable ? " : 0ULL" : "") << ";\n";
    };
    const DType& t = nd.type;
    if (t.id == QHIP_UTF8) {
      if (nd.kind != QHIP_EXPR_COLUMN) fail(QHIP_UNSUPPORTED, "ORDER BY on a computed Utf8 expression is not accelerated");
      kd.words = 0; kd.top_bits = 0; kd.column = nd.column;
    } else if (t.id == QHIP_DECIMAL128) {
      kd.words = 2;
      store(0, "(u64)(u128)" + v);
      store(1, "((u64)((u128)" + v + " >> 64) ^ 0x8000000000000000ULL)");
    } else if (t.id == QHIP_BOOL) {
      kd.top_bits = 1;
      store(0, "(" + v + " ? 1ULL : 0ULL)");
    } else if (dtype_is_float(t)) {
      store(0, "qh_f64_ord((double)" + v + ")");
    } else if (intlike(t)) {
      const int b = dtype_width(t) * 8;
      kd.top_bits = b;
      if (signed_intlike(t)) {
        if (b == 64) store(0, "((u64)(i64)" + v + " ^ 0x8000000000000000ULL)");
        else store(0, "(((u64)(i64)" + v + " + " + std::to_string(1ULL << (b - 1)) + "ULL) & " + std::to_string((1ULL << b) - 1) + "ULL)");
      } else {
        store(0, "(u64)" + v);
      }
    } else if (t.id == QHIP_NULL) {
      kd.top_bits = 1;
      store(0, "0ULL");
    } else {
      fail(QHIP_UNSUPPORTED, "ORDER BY on " + dtype_name(t) + " is not supported");
    }
    if (kd.nullable) body << "    valid |= " << okx << " ? " << (1u << k) << "u : 0u;\n";
    else body << "    valid |= " << (1u << k) << "u;\n";
    off += kd.words;
    out.keys.push_back(kd);
  }
  out.NW = off;
  std::ostringstream s;
  s << "struct P {\n  static constexpr int NW = " << out.NW << ";\n  static constexpr int NK = " << n << ";\n";
  s << "  __device__ static __forceinline__ void images(const KArgs& a, const i64 i, u64* img, u32& valid, u32& err) {\n" << body.str() << "  }\n};\n";
  s << "extern \"C\" __global__ __launch_bounds__(QH_BLOCK) void qk_sort_keys(KArgs a, u64* img, u64* keyvalid, u64* diff, u32* status) { "
       "qh_sort_keys_body<P>(a, img, keyvalid, diff, status); }\n";
  out.source = s.str();
  out.kernel_name = "qk_sort_keys";
  out.bind = g.bind;
}

}  // namespace qhip
