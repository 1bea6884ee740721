/*
 * qoracle.c — CPU restatement of the reference's filter / hash-aggregate / hash-join path.
 * TEST INFRASTRUCTURE ONLY (see qoracle.h). Every function cites the reference lines it follows
 * (paths relative to /root/reference/qurious/src).
 */
#include "qoracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

static __thread char g_err[512];
const char* qo_last_error(void) { return g_err; }
#define QO_FAIL(code, ...)                        \
  do {                                            \
    snprintf(g_err, sizeof g_err, __VA_ARGS__);   \
    return (code);                                \
  } while (0)

void qo_free(void* p) { free(p); }

void qo_col_free(qo_col* c) {
  if (!c) return;
  if (c->owned) { free(c->values); free(c->valid); free(c->offsets); free(c->data); }
  memset(c, 0, sizeof *c);
}

static int type_width(int id) {
  switch (id) {
    case QHIP_BOOL: case QHIP_INT8: case QHIP_UINT8: return 1;
    case QHIP_INT16: case QHIP_UINT16: return 2;
    case QHIP_INT32: case QHIP_UINT32: case QHIP_FLOAT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: return 4;
    case QHIP_INT64: case QHIP_UINT64: case QHIP_FLOAT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: return 8;
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: return 8;   /* aggregate/mod.rs:108-111 */
    case QHIP_DECIMAL128: return 16;
    default: return 0;
  }
}
static int is_intlike(int id) { return (id >= QHIP_INT8 && id <= QHIP_UINT64) || id == QHIP_DATE32 || id == QHIP_DATE64; }
static int is_signed_int(int id) { return (id >= QHIP_INT8 && id <= QHIP_INT64) || id == QHIP_DATE32 || id == QHIP_DATE64; }
static int is_float(int id) { return id == QHIP_FLOAT32 || id == QHIP_FLOAT64; }
static int same_type(qhip_dtype a, qhip_dtype b) {
  return a.id == b.id && (a.id != QHIP_DECIMAL128 || (a.precision == b.precision && a.scale == b.scale));
}
static i128 pow10_128(int e) { i128 r = 1; for (int k = 0; k < e; ++k) r *= 10; return r; }

static int col_alloc(qo_col* c, qhip_dtype t, int64_t n, int with_valid) {
  memset(c, 0, sizeof *c);
  c->type = t; c->n = n; c->owned = 1;
  int w = type_width(t.id);
  if (w) { c->values = calloc((size_t)(n > 0 ? n : 1), (size_t)w); if (!c->values) QO_FAIL(QHIP_OUT_OF_MEMORY, "oom"); }
  if (with_valid) { c->valid = malloc((size_t)(n > 0 ? n : 1)); if (!c->valid) QO_FAIL(QHIP_OUT_OF_MEMORY, "oom"); memset(c->valid, 1, (size_t)n); }
  return 0;
}

/* ================================================================ SipHash-1-3 (Rust std::hash::DefaultHasher)
 * core::hash::sip::Hasher<Sip13Rounds>: 1 compression round per 8-byte word, 3 finalisation rounds, keys 0/0
 * (DefaultHasher::new()). Integers are written as their little-endian bytes, str as bytes followed by 0xFF
 * (Hasher::write_str default). */
#define ROTL(x, b) (uint64_t)(((x) << (b)) | ((x) >> (64 - (b))))
#define SIPROUND(h)                                                                  \
  do {                                                                               \
    h->v0 += h->v1; h->v1 = ROTL(h->v1, 13); h->v1 ^= h->v0; h->v0 = ROTL(h->v0, 32); \
    h->v2 += h->v3; h->v3 = ROTL(h->v3, 16); h->v3 ^= h->v2;                         \
    h->v0 += h->v3; h->v3 = ROTL(h->v3, 21); h->v3 ^= h->v0;                         \
    h->v2 += h->v1; h->v1 = ROTL(h->v1, 17); h->v1 ^= h->v2; h->v2 = ROTL(h->v2, 32); \
  } while (0)

void qo_hasher_init(qo_hasher* h) {
  h->k0 = 0; h->k1 = 0; h->length = 0; h->tail = 0; h->ntail = 0;
  h->v0 = h->k0 ^ 0x736f6d6570736575ULL;
  h->v1 = h->k1 ^ 0x646f72616e646f6dULL;
  h->v2 = h->k0 ^ 0x6c7967656e657261ULL;
  h->v3 = h->k1 ^ 0x7465646279746573ULL;
}
static inline uint64_t load_le(const uint8_t* p, size_t n) {
  uint64_t v = 0;
  for (size_t k = 0; k < n; ++k) v |= (uint64_t)p[k] << (8 * k);
  return v;
}
void qo_hasher_write(qo_hasher* h, const uint8_t* msg, size_t len) {
  h->length += len;
  size_t needed = 0;
  if (h->ntail != 0) {
    needed = 8 - (size_t)h->ntail;
    size_t take = len < needed ? len : needed;
    h->tail |= load_le(msg, take) << (8 * h->ntail);
    if (len < needed) { h->ntail += len; return; }
    h->v3 ^= h->tail; SIPROUND(h); h->v0 ^= h->tail;
    h->ntail = 0; h->tail = 0;
  }
  size_t rem = len - needed;
  size_t left = rem & 7;
  size_t i = needed;
  const size_t end = len - left;
  for (; i < end; i += 8) {
    uint64_t m = load_le(msg + i, 8);
    h->v3 ^= m; SIPROUND(h); h->v0 ^= m;
  }
  h->tail = load_le(msg + i, left);
  h->ntail = left;
}
uint64_t qo_hasher_finish(const qo_hasher* hc) {
  qo_hasher s = *hc;
  qo_hasher* h = &s;
  uint64_t b = ((h->length & 0xff) << 56) | h->tail;
  h->v3 ^= b; SIPROUND(h); h->v0 ^= b;
  h->v2 ^= 0xff;
  SIPROUND(h); SIPROUND(h); SIPROUND(h);
  return h->v0 ^ h->v1 ^ h->v2 ^ h->v3;
}

/* utils/array.rs:171-210 — hash_array!: NULL values feed nothing; supported key types only */
int qo_create_hashes(const qo_col* cols, int ncols, int64_t nrows, uint64_t* out) {
  /* hash.rs:46-47 / hash_join.rs:161: one DefaultHasher (72 bytes) per row */
  qo_hasher* hs = (qo_hasher*)malloc(sizeof(qo_hasher) * (size_t)(nrows > 0 ? nrows : 1));
  if (!hs) QO_FAIL(QHIP_OUT_OF_MEMORY, "oom");
  for (int64_t r = 0; r < nrows; ++r) qo_hasher_init(&hs[r]);
  for (int c = 0; c < ncols; ++c) {
    const qo_col* col = &cols[c];
    int w = 0;
    switch (col->type.id) {
      case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: w = 8; break;   /* utils/array.rs:193-202 */
      case QHIP_UINT8: w = 1; break;
      case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: w = 4; break;
      case QHIP_DECIMAL128: w = 16; break;
      case QHIP_UTF8: w = -1; break;
      default:
        free(hs);
        QO_FAIL(QHIP_INVALID_ARGUMENT, "Internal error: Unsupported data type in hasher: type id %d", col->type.id);
    }
    for (int64_t r = 0; r < nrows; ++r) {
      if (col->valid && !col->valid[r]) continue;
      if (w > 0) qo_hasher_write(&hs[r], (const uint8_t*)col->values + (size_t)r * w, (size_t)w);
      else {
        qo_hasher_write(&hs[r], col->data + col->offsets[r], (size_t)(col->offsets[r + 1] - col->offsets[r]));
        const uint8_t ff = 0xff;
        qo_hasher_write(&hs[r], &ff, 1);
      }
    }
  }
  for (int64_t r = 0; r < nrows; ++r) out[r] = qo_hasher_finish(&hs[r]);
  free(hs);
  return 0;
}

/* ================================================================ expression evaluation
 * physical/expr/{column,literal,binary,cast,is_null,is_not_null,negative}.rs over arrow-rs kernels.
 * Every node materialises a full-length array (that is what the reference's arrow calls do). */
static int parse_date32(const uint8_t* s, int len, int32_t* out) {
  char buf[32];
  if (len <= 0 || len >= (int)sizeof buf) return -1;
  memcpy(buf, s, (size_t)len); buf[len] = 0;
  int y, m, d; char tail;
  if (sscanf(buf, "%d-%d-%d%c", &y, &m, &d, &tail) != 3 || m < 1 || m > 12 || d < 1) return -1;
  static const int md[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  int leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
  if (d > md[m - 1] + (m == 2 && leap)) return -1;
  int yy = y - (m <= 2);
  int era = (yy >= 0 ? yy : yy - 399) / 400;
  unsigned yoe = (unsigned)(yy - era * 400);
  unsigned doy = (153u * (unsigned)(m + (m > 2 ? -3 : 9)) + 2) / 5 + (unsigned)d - 1;
  unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  *out = (int32_t)(era * 146097 + (int)doe - 719468);
  return 0;
}

/* widen any int-like / decimal value to i128, floats to double */
static i128 get_int(const qo_col* c, int64_t i) {
  switch (c->type.id) {
    case QHIP_BOOL: return ((const uint8_t*)c->values)[i];
    case QHIP_INT8: return ((const int8_t*)c->values)[i];
    case QHIP_INT16: return ((const int16_t*)c->values)[i];
    case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: return ((const int32_t*)c->values)[i];
    case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: return ((const int64_t*)c->values)[i];
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: return ((const int64_t*)c->values)[i];
    case QHIP_UINT8: return ((const uint8_t*)c->values)[i];
    case QHIP_UINT16: return ((const uint16_t*)c->values)[i];
    case QHIP_UINT32: return ((const uint32_t*)c->values)[i];
    case QHIP_UINT64: return (i128)((const uint64_t*)c->values)[i];
    case QHIP_DECIMAL128: return ((const i128*)c->values)[i];
  }
  return 0;
}
static double get_f64(const qo_col* c, int64_t i) {
  if (c->type.id == QHIP_FLOAT32) return ((const float*)c->values)[i];
  if (c->type.id == QHIP_FLOAT64) return ((const double*)c->values)[i];
  return (double)get_int(c, i);
}
static void put_int(qo_col* c, int64_t i, i128 v) {
  switch (c->type.id) {
    case QHIP_BOOL: ((uint8_t*)c->values)[i] = v != 0; break;
    case QHIP_INT8: ((int8_t*)c->values)[i] = (int8_t)v; break;
    case QHIP_INT16: ((int16_t*)c->values)[i] = (int16_t)v; break;
    case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: ((int32_t*)c->values)[i] = (int32_t)v; break;
    case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: ((int64_t*)c->values)[i] = (int64_t)v; break;
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: ((int64_t*)c->values)[i] = (int64_t)v; break;
    case QHIP_UINT8: ((uint8_t*)c->values)[i] = (uint8_t)v; break;
    case QHIP_UINT16: ((uint16_t*)c->values)[i] = (uint16_t)v; break;
    case QHIP_UINT32: ((uint32_t*)c->values)[i] = (uint32_t)v; break;
    case QHIP_UINT64: ((uint64_t*)c->values)[i] = (uint64_t)v; break;
    case QHIP_DECIMAL128: ((i128*)c->values)[i] = v; break;
    case QHIP_FLOAT32: ((float*)c->values)[i] = (float)v; break;
    case QHIP_FLOAT64: ((double*)c->values)[i] = (double)v; break;
  }
}
static void int_limits(int id, i128* lo, i128* hi) {
  switch (id) {
    case QHIP_INT8: *lo = -128; *hi = 127; break;
    case QHIP_INT16: *lo = -32768; *hi = 32767; break;
    case QHIP_INT32: case QHIP_DATE32: case QHIP_TIME32_S: case QHIP_TIME32_MS: *lo = INT32_MIN; *hi = INT32_MAX; break;
    case QHIP_INT64: case QHIP_DATE64: case QHIP_TIME64_US: case QHIP_TIME64_NS: *lo = INT64_MIN; *hi = INT64_MAX; break;
    case QHIP_TIMESTAMP_S: case QHIP_TIMESTAMP_MS: case QHIP_TIMESTAMP_US: case QHIP_TIMESTAMP_NS: *lo = INT64_MIN; *hi = INT64_MAX; break;
    case QHIP_UINT8: *lo = 0; *hi = 255; break;
    case QHIP_UINT16: *lo = 0; *hi = 65535; break;
    case QHIP_UINT32: *lo = 0; *hi = UINT32_MAX; break;
    default: *lo = 0; *hi = (i128)UINT64_MAX; break;
  }
}
static uint64_t f64_total_key(double d) {
  uint64_t b; memcpy(&b, &d, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
static int str_cmp(const qo_col* a, int64_t i, const qo_col* b, int64_t j) {
  int la = a->offsets[i + 1] - a->offsets[i], lb = b->offsets[j + 1] - b->offsets[j];
  int n = la < lb ? la : lb;
  int c = memcmp(a->data + a->offsets[i], b->data + b->offsets[j], (size_t)n);
  return c ? c : la - lb;
}

static int eval_node(const qhip_expr* ex, int n_exprs, int k, const qo_col* cols, int ncols, int64_t n, qo_col* out);

/* ScalarValue::to_array(n) (datatypes/scalar.rs:166-191): `vec![v; n]` */
static int eval_literal(const qhip_expr* e, int64_t n, qo_col* out) {
  int isnull = e->lit_is_null || e->dtype.id == QHIP_NULL;
  if (e->dtype.id == QHIP_UTF8) {
    memset(out, 0, sizeof *out);
    out->type = e->dtype; out->n = n; out->owned = 1;
    int64_t len = isnull ? 0 : e->lit_len;
    out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    out->data = (uint8_t*)malloc((size_t)(len * n + 1));
    for (int64_t i = 0; i < n; ++i) { out->offsets[i] = (int32_t)(i * len); if (len) memcpy(out->data + i * len, e->lit_str, (size_t)len); }
    out->offsets[n] = (int32_t)(n * len);
    if (isnull) { out->valid = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1); }
    return 0;
  }
  int rc = col_alloc(out, e->dtype, n, isnull);
  if (rc) return rc;
  if (isnull) { memset(out->valid, 0, (size_t)n); return 0; }
  if (is_float(e->dtype.id)) { for (int64_t i = 0; i < n; ++i) { if (e->dtype.id == QHIP_FLOAT32) ((float*)out->values)[i] = (float)e->lit_f64; else ((double*)out->values)[i] = e->lit_f64; } return 0; }
  i128 v;
  if (e->dtype.id == QHIP_DECIMAL128) v = (i128)(((u128)(uint64_t)e->lit_hi << 64) | (u128)e->lit_lo);
  else if (e->dtype.id == QHIP_UINT64) v = (i128)e->lit_lo;
  else v = (i128)(int64_t)e->lit_lo;
  for (int64_t i = 0; i < n; ++i) put_int(out, i, v);
  return 0;
}

static uint8_t* merge_valid(const qo_col* a, const qo_col* b, int64_t n) {
  if (!a->valid && !b->valid) return NULL;
  uint8_t* v = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) v[i] = (uint8_t)((!a->valid || a->valid[i]) && (!b->valid || b->valid[i]));
  return v;
}

/* arrow_cast::cast_with_options(.., safe=false) (physical/expr/cast.rs:33-37) */
static int eval_cast(const qo_col* in, qhip_dtype to, int64_t n, qo_col* out) {
  qhip_dtype from = in->type;
  if (from.id == QHIP_UTF8) {
    if (to.id != QHIP_DATE32) QO_FAIL(QHIP_UNSUPPORTED, "oracle: cast Utf8 -> type %d", to.id);
    int rc = col_alloc(out, to, n, in->valid != NULL);
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i) {
      if (in->valid && !in->valid[i]) { out->valid[i] = 0; continue; }
      int32_t d;
      if (parse_date32(in->data + in->offsets[i], in->offsets[i + 1] - in->offsets[i], &d))
        QO_FAIL(QHIP_EXEC_ERROR, "Cast error: Cannot cast string to value of Date32 type");
      ((int32_t*)out->values)[i] = d;
    }
    return 0;
  }
  int rc = col_alloc(out, to, n, in->valid != NULL);
  if (rc) return rc;
  if (in->valid) memcpy(out->valid, in->valid, (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (in->valid && !in->valid[i]) continue;
    if (same_type(from, to)) { if (is_float(from.id)) { if (from.id == QHIP_FLOAT32) ((float*)out->values)[i] = ((float*)in->values)[i]; else ((double*)out->values)[i] = ((double*)in->values)[i]; } else put_int(out, i, get_int(in, i)); continue; }
    if (is_float(to.id)) {
      double f;
      if (from.id == QHIP_DECIMAL128) f = (double)get_int(in, i) / pow(10.0, from.scale);
      else f = get_f64(in, i);
      if (to.id == QHIP_FLOAT32) ((float*)out->values)[i] = (float)f; else ((double*)out->values)[i] = f;
      continue;
    }
    if (to.id == QHIP_DECIMAL128) {
      i128 r;
      if (is_float(from.id)) {
        double sc = round(get_f64(in, i) * pow(10.0, to.scale));
        if (!(fabs(sc) < 1.7e38)) QO_FAIL(QHIP_EXEC_ERROR, "Cast error: float value out of Decimal128 range");
        r = (i128)sc;
      } else if (from.id == QHIP_DECIMAL128) {
        i128 v = get_int(in, i);
        if (to.scale >= from.scale) { if (__builtin_mul_overflow(v, pow10_128(to.scale - from.scale), &r)) QO_FAIL(QHIP_EXEC_ERROR, "Cast error: decimal overflow"); }
        else {
          i128 d = pow10_128(from.scale - to.scale), q = v / d, rm = v % d, half = d / 2;
          if (rm >= half) q += 1; else if (-rm >= half) q -= 1;
          r = q;
        }
      } else {
        if (__builtin_mul_overflow(get_int(in, i), pow10_128(to.scale), &r)) QO_FAIL(QHIP_EXEC_ERROR, "Cast error: decimal overflow");
      }
      i128 lim = pow10_128(to.precision);
      if (r >= lim || r <= -lim) QO_FAIL(QHIP_EXEC_ERROR, "Cast error: value too large to store in a Decimal128 of precision %d", to.precision);
      ((i128*)out->values)[i] = r;
      continue;
    }
    if (is_intlike(to.id)) {
      i128 v;
      if (is_float(from.id)) {
        double t = trunc(get_f64(in, i));
        i128 lo, hi; int_limits(to.id, &lo, &hi);
        if (!(t >= (double)lo && t <= (double)hi)) QO_FAIL(QHIP_EXEC_ERROR, "Cast error: Can't cast value to type %d", to.id);
        v = (i128)t;
      } else if (from.id == QHIP_DECIMAL128) {
        v = get_int(in, i) / pow10_128(from.scale);
      } else if (from.id == QHIP_DATE32 && to.id == QHIP_DATE64) {
        v = get_int(in, i) * 86400000;
      } else if (from.id == QHIP_DATE64 && to.id == QHIP_DATE32) {
        v = get_int(in, i) / 86400000;
      } else {
        v = get_int(in, i);
      }
      i128 lo, hi; int_limits(to.id, &lo, &hi);
      if (v < lo || v > hi) QO_FAIL(QHIP_EXEC_ERROR, "Cast error: Can't cast value to type %d", to.id);
      put_int(out, i, v);
      continue;
    }
    QO_FAIL(QHIP_UNSUPPORTED, "oracle: cast %d -> %d", from.id, to.id);
  }
  return 0;
}

/* physical/expr/binary.rs:31-70 */
static int eval_binary(int op, const qo_col* l, const qo_col* r, int64_t n, qo_col* out) {
  if (op >= QHIP_OP_EQ && op <= QHIP_OP_LTEQ) {
    /* arrow_ord::cmp: identical types required; result NULL where either side is NULL; floats in total order */
    if (!same_type(l->type, r->type)) QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid comparison operation: %d vs %d", l->type.id, r->type.id);
    qhip_dtype bt = {QHIP_BOOL, 0, 0};
    int rc = col_alloc(out, bt, n, 0);
    if (rc) return rc;
    out->valid = merge_valid(l, r, n);
    uint8_t* o = (uint8_t*)out->values;
    for (int64_t i = 0; i < n; ++i) {
      if (out->valid && !out->valid[i]) continue;
      int c;
      if (l->type.id == QHIP_UTF8) c = str_cmp(l, i, r, i);
      else if (is_float(l->type.id)) { uint64_t a = f64_total_key(get_f64(l, i)), b = f64_total_key(get_f64(r, i)); c = a < b ? -1 : a > b; }
      else { i128 a = get_int(l, i), b = get_int(r, i); c = a < b ? -1 : a > b; }
      switch (op) {
        case QHIP_OP_EQ: o[i] = c == 0; break;
        case QHIP_OP_NOTEQ: o[i] = c != 0; break;
        case QHIP_OP_GT: o[i] = c > 0; break;
        case QHIP_OP_GTEQ: o[i] = c >= 0; break;
        case QHIP_OP_LT: o[i] = c < 0; break;
        default: o[i] = c <= 0; break;
      }
    }
    return 0;
  }
  if (op == QHIP_OP_AND || op == QHIP_OP_OR) {
    /* and_kleene / or_kleene (binary.rs:44-49) */
    if (l->type.id != QHIP_BOOL || r->type.id != QHIP_BOOL) QO_FAIL(QHIP_INVALID_ARGUMENT, "boolean operator on non-boolean operands");
    qhip_dtype bt = {QHIP_BOOL, 0, 0};
    int rc = col_alloc(out, bt, n, (l->valid || r->valid));
    if (rc) return rc;
    const uint8_t *a = (const uint8_t*)l->values, *b = (const uint8_t*)r->values;
    uint8_t* o = (uint8_t*)out->values;
    for (int64_t i = 0; i < n; ++i) {
      int la = !l->valid || l->valid[i], lb = !r->valid || r->valid[i];
      if (op == QHIP_OP_AND) {
        int valid = (la && lb) || (la && !a[i]) || (lb && !b[i]);
        o[i] = (uint8_t)((!la || a[i]) && (!lb || b[i]));
        if (out->valid) out->valid[i] = (uint8_t)valid;
      } else {
        int valid = (la && lb) || (la && a[i]) || (lb && b[i]);
        o[i] = (uint8_t)((la && a[i]) || (lb && b[i]));
        if (out->valid) out->valid[i] = (uint8_t)valid;
      }
    }
    return 0;
  }
  /* arithmetic: add_wrapping / sub_wrapping / mul_wrapping / div / rem (binary.rs:51-68) */
  int ld = l->type.id == QHIP_DECIMAL128, rd = r->type.id == QHIP_DECIMAL128;
  if (ld || rd) {
    if (op == QHIP_OP_DIV) {
      /* binary.rs:54-67: both sides cast to Float64, IEEE division */
      qhip_dtype ft = {QHIP_FLOAT64, 0, 0};
      int rc = col_alloc(out, ft, n, 0);
      if (rc) return rc;
      out->valid = merge_valid(l, r, n);
      for (int64_t i = 0; i < n; ++i) {
        if (out->valid && !out->valid[i]) continue;
        double a = ld ? (double)get_int(l, i) / pow(10.0, l->type.scale) : get_f64(l, i);
        double b = rd ? (double)get_int(r, i) / pow(10.0, r->type.scale) : get_f64(r, i);
        ((double*)out->values)[i] = a / b;
      }
      return 0;
    }
    if (!(ld && rd)) QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid arithmetic operation: decimal with non-decimal");
    int p1 = l->type.precision, s1 = l->type.scale, p2 = r->type.precision, s2 = r->type.scale;
    qhip_dtype t = {QHIP_DECIMAL128, 0, 0};
    if (op == QHIP_OP_ADD || op == QHIP_OP_SUB) {
      int s = s1 > s2 ? s1 : s2;
      int m = (p1 - s1) > (p2 - s2) ? (p1 - s1) : (p2 - s2);
      t.scale = s; t.precision = m + s + 1 > 38 ? 38 : m + s + 1;
    } else if (op == QHIP_OP_MUL) {
      t.scale = s1 + s2;
      if (t.scale > 38) QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: Output scale would exceed max scale of 38");
      t.precision = p1 + p2 + 1 > 38 ? 38 : p1 + p2 + 1;
    } else QO_FAIL(QHIP_UNSUPPORTED, "oracle: decimal remainder");
    int rc = col_alloc(out, t, n, 0);
    if (rc) return rc;
    out->valid = merge_valid(l, r, n);
    i128 lm = pow10_128(t.scale - s1 > 0 && op != QHIP_OP_MUL ? t.scale - s1 : 0), rm = pow10_128(t.scale - s2 > 0 && op != QHIP_OP_MUL ? t.scale - s2 : 0);
    for (int64_t i = 0; i < n; ++i) {
      if (out->valid && !out->valid[i]) continue;
      i128 a = get_int(l, i), b = get_int(r, i), v;
      /* arrow-arith decimal_op evaluates decimals with checked i128 arithmetic */
      int of = 0;
      if (op == QHIP_OP_MUL) of = __builtin_mul_overflow(a, b, &v);
      else {
        i128 x, y;
        of = __builtin_mul_overflow(a, lm, &x) | __builtin_mul_overflow(b, rm, &y);
        if (!of) of = op == QHIP_OP_ADD ? __builtin_add_overflow(x, y, &v) : __builtin_sub_overflow(x, y, &v);
      }
      if (of) QO_FAIL(QHIP_EXEC_ERROR, "Arrow error: Arithmetic overflow: Overflow happened on decimal arithmetic");
      ((i128*)out->values)[i] = v;
    }
    return 0;
  }
  if (!same_type(l->type, r->type) || !(is_float(l->type.id) || (l->type.id >= QHIP_INT8 && l->type.id <= QHIP_UINT64)))
    QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid arithmetic operation: %d op %d", l->type.id, r->type.id);
  int rc = col_alloc(out, l->type, n, 0);
  if (rc) return rc;
  out->valid = merge_valid(l, r, n);
  for (int64_t i = 0; i < n; ++i) {
    if (out->valid && !out->valid[i]) continue;
    if (is_float(l->type.id)) {
      double a = get_f64(l, i), b = get_f64(r, i), v;
      if (l->type.id == QHIP_FLOAT32) {
        float fa = (float)a, fb = (float)b, fv;
        switch (op) { case QHIP_OP_ADD: fv = fa + fb; break; case QHIP_OP_SUB: fv = fa - fb; break; case QHIP_OP_MUL: fv = fa * fb; break; case QHIP_OP_DIV: fv = fa / fb; break; default: fv = fmodf(fa, fb); }
        ((float*)out->values)[i] = fv;
        continue;
      }
      switch (op) { case QHIP_OP_ADD: v = a + b; break; case QHIP_OP_SUB: v = a - b; break; case QHIP_OP_MUL: v = a * b; break; case QHIP_OP_DIV: v = a / b; break; default: v = fmod(a, b); }
      ((double*)out->values)[i] = v;
    } else {
      i128 a = get_int(l, i), b = get_int(r, i), v = 0, lo, hi;
      int_limits(l->type.id, &lo, &hi);
      switch (op) {
        case QHIP_OP_ADD: v = a + b; break;
        case QHIP_OP_SUB: v = a - b; break;
        case QHIP_OP_MUL: v = (i128)((u128)a * (u128)b); break;
        case QHIP_OP_DIV:
          if (b == 0) QO_FAIL(QHIP_EXEC_ERROR, "Arrow error: Divide by zero error");
          v = a / b;
          if (v < lo || v > hi) QO_FAIL(QHIP_EXEC_ERROR, "Arrow error: Arithmetic overflow: Overflow happened on integer division");
          break;
        default:
          if (b == 0) QO_FAIL(QHIP_EXEC_ERROR, "Arrow error: Divide by zero error");
          v = (b == -1) ? 0 : a % b;
      }
      put_int(out, i, v); /* truncation to the type's width == two's-complement wrapping */
    }
  }
  return 0;
}

/* arrow_select::zip::zip(mask, truthy, falsy) as CaseExpr::evaluate uses it (physical/expr/case.rs:33-48): row i comes
 * from `truthy` where the mask is valid and true, else from `falsy`; the two sides must have the same type. */
static int eval_zip(const qo_col* mask, const qo_col* t, const qo_col* f, int64_t n, qo_col* out) {
  if (mask->type.id != QHIP_BOOL) QO_FAIL(QHIP_INVALID_ARGUMENT, "Internal error: CASE WHEN must be boolean");
  if (t->type.id != f->type.id || t->type.precision != f->type.precision || t->type.scale != f->type.scale)
    QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: arguments need to have the same data type");
  const uint8_t* mv = (const uint8_t*)mask->values;
  const int with_valid = t->valid || f->valid || t->type.id == QHIP_NULL;
  if (t->type.id == QHIP_UTF8) {
    memset(out, 0, sizeof *out);
    out->type = t->type; out->n = n; out->owned = 1;
    out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
      const qo_col* src = ((!mask->valid || mask->valid[i]) && mv[i]) ? t : f;
      total += src->offsets[i + 1] - src->offsets[i];
    }
    out->data = (uint8_t*)malloc((size_t)(total + 1));
    if (with_valid) out->valid = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    int64_t pos = 0;
    for (int64_t i = 0; i < n; ++i) {
      const qo_col* src = ((!mask->valid || mask->valid[i]) && mv[i]) ? t : f;
      const int len = src->offsets[i + 1] - src->offsets[i];
      out->offsets[i] = (int32_t)pos;
      if (len) memcpy(out->data + pos, src->data + src->offsets[i], (size_t)len);
      pos += len;
      if (with_valid) out->valid[i] = (uint8_t)(!src->valid || src->valid[i]);
    }
    out->offsets[n] = (int32_t)pos;
    return 0;
  }
  int rc = col_alloc(out, t->type, n, with_valid);
  if (rc) return rc;
  const int w = type_width(t->type.id);
  for (int64_t i = 0; i < n; ++i) {
    const qo_col* src = ((!mask->valid || mask->valid[i]) && mv[i]) ? t : f;
    if (w) memcpy((uint8_t*)out->values + (size_t)i * (size_t)w, (const uint8_t*)src->values + (size_t)i * (size_t)w, (size_t)w);
    if (with_valid) out->valid[i] = (uint8_t)(t->type.id == QHIP_NULL ? 0 : (!src->valid || src->valid[i]));
  }
  return 0;
}

/* arrow_string::like::like (physical/expr/like.rs:28-43): `%` any sequence of characters, `_` exactly one character
 * (a UTF-8 code point), backslash escapes the next pattern character. Recursive on purpose (the HIP matcher is an
 * iterative backtracker: two independent formulations). */
static int utf8_char_len(const uint8_t* s, int n) {
  int k = 1;
  while (k < n && (s[k] & 0xC0) == 0x80) ++k;
  return k;
}
static int like_match(const uint8_t* s, int n, const uint8_t* p, int m) {
  if (m == 0) return n == 0;
  if (p[0] == '%') {
    while (m > 1 && p[1] == '%') { ++p; --m; }
    for (int k = 0;; k += utf8_char_len(s + k, n - k)) {
      if (like_match(s + k, n - k, p + 1, m - 1)) return 1;
      if (k >= n) return 0;
    }
  }
  if (n == 0) return 0;
  if (p[0] == '_') { const int c = utf8_char_len(s, n); return like_match(s + c, n - c, p + 1, m - 1); }
  if (p[0] == '\\' && m >= 2) return s[0] == p[1] && like_match(s + 1, n - 1, p + 2, m - 2);
  return s[0] == p[0] && like_match(s + 1, n - 1, p + 1, m - 1);
}

static int eval_node(const qhip_expr* ex, int n_exprs, int k, const qo_col* cols, int ncols, int64_t n, qo_col* out) {
  if (k < 0 || k >= n_exprs) QO_FAIL(QHIP_INVALID_ARGUMENT, "expression index out of range");
  const qhip_expr* e = &ex[k];
  switch (e->kind) {
    case QHIP_EXPR_COLUMN: {
      /* column.rs:24-34: bounds check, Arc clone (no copy) */
      if (e->column < 0 || e->column >= ncols)
        QO_FAIL(QHIP_INVALID_ARGUMENT, "PhysicalExpr Column references column at index %d (zero-based) but input schema only has %d columns", e->column, ncols);
      *out = cols[e->column];
      out->owned = 0;
      out->n = n;
      return 0;
    }
    case QHIP_EXPR_LITERAL: return eval_literal(e, n, out);
    case QHIP_EXPR_BINARY: {
      qo_col l, r;
      int rc = eval_node(ex, n_exprs, e->left, cols, ncols, n, &l);
      if (rc) return rc;
      rc = eval_node(ex, n_exprs, e->right, cols, ncols, n, &r);
      if (rc) { qo_col_free(&l); return rc; }
      rc = eval_binary(e->op, &l, &r, n, out);
      qo_col_free(&l); qo_col_free(&r);
      return rc;
    }
    case QHIP_EXPR_CAST: {
      qo_col c;
      int rc = eval_node(ex, n_exprs, e->left, cols, ncols, n, &c);
      if (rc) return rc;
      rc = eval_cast(&c, e->dtype, n, out);
      qo_col_free(&c);
      return rc;
    }
    case QHIP_EXPR_IS_NULL:
    case QHIP_EXPR_IS_NOT_NULL: {
      qo_col c;
      int rc = eval_node(ex, n_exprs, e->left, cols, ncols, n, &c);
      if (rc) return rc;
      qhip_dtype bt = {QHIP_BOOL, 0, 0};
      rc = col_alloc(out, bt, n, 0);
      if (!rc) for (int64_t i = 0; i < n; ++i) { int v = !c.valid || c.valid[i]; if (c.type.id == QHIP_NULL) v = 0; ((uint8_t*)out->values)[i] = (uint8_t)(e->kind == QHIP_EXPR_IS_NULL ? !v : v); }
      qo_col_free(&c);
      return rc;
    }
    case QHIP_EXPR_NEGATIVE: {
      qo_col c;
      int rc = eval_node(ex, n_exprs, e->left, cols, ncols, n, &c);
      if (rc) return rc;
      if (!(is_signed_int(c.type.id) || is_float(c.type.id) || c.type.id == QHIP_DECIMAL128)) { qo_col_free(&c); QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid arithmetic operation: negation"); }
      rc = col_alloc(out, c.type, n, c.valid != NULL);
      if (!rc) {
        if (c.valid) memcpy(out->valid, c.valid, (size_t)n);
        for (int64_t i = 0; i < n; ++i) {
          if (c.valid && !c.valid[i]) continue;
          if (c.type.id == QHIP_FLOAT64) ((double*)out->values)[i] = -((double*)c.values)[i];
          else if (c.type.id == QHIP_FLOAT32) ((float*)out->values)[i] = -((float*)c.values)[i];
          else put_int(out, i, (i128)((u128)0 - (u128)get_int(&c, i)));
        }
      }
      qo_col_free(&c);
      return rc;
    }
    case QHIP_EXPR_IF: {
      /* case.rs:36-46: every WHEN, THEN and the accumulated ELSE are evaluated over the whole batch, then zipped */
      qo_col c, t, f;
      int rc = eval_node(ex, n_exprs, e->third, cols, ncols, n, &f);
      if (rc) return rc;
      rc = eval_node(ex, n_exprs, e->left, cols, ncols, n, &c);
      if (rc) { qo_col_free(&f); return rc; }
      rc = eval_node(ex, n_exprs, e->right, cols, ncols, n, &t);
      if (rc) { qo_col_free(&f); qo_col_free(&c); return rc; }
      rc = eval_zip(&c, &t, &f, n, out);
      qo_col_free(&c); qo_col_free(&t); qo_col_free(&f);
      return rc;
    }
    case QHIP_EXPR_LIKE: {
      qo_col x, p;
      int rc = eval_node(ex, n_exprs, e->left, cols, ncols, n, &x);
      if (rc) return rc;
      rc = eval_node(ex, n_exprs, e->right, cols, ncols, n, &p);
      if (rc) { qo_col_free(&x); return rc; }
      if (x.type.id != QHIP_UTF8 || p.type.id != QHIP_UTF8) { qo_col_free(&x); qo_col_free(&p); QO_FAIL(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid string operation: LIKE on non-Utf8 operands"); }
      qhip_dtype bt = {QHIP_BOOL, 0, 0};
      rc = col_alloc(out, bt, n, x.valid != NULL || p.valid != NULL);
      if (!rc)
        for (int64_t i = 0; i < n; ++i) {
          const int ok = (!x.valid || x.valid[i]) && (!p.valid || p.valid[i]);
          if (out->valid) out->valid[i] = (uint8_t)ok;
          if (!ok) continue;
          const int m = like_match(x.data + x.offsets[i], x.offsets[i + 1] - x.offsets[i], p.data + p.offsets[i], p.offsets[i + 1] - p.offsets[i]);
          ((uint8_t*)out->values)[i] = (uint8_t)(e->op ? !m : m);
        }
      qo_col_free(&x); qo_col_free(&p);
      return rc;
    }
  }
  QO_FAIL(QHIP_INVALID_ARGUMENT, "unknown expression kind %d", e->kind);
}

int qo_eval(const qhip_expr* exprs, int n_exprs, int root, const qo_col* cols, int ncols, int64_t nrows, qo_col* out) {
  return eval_node(exprs, n_exprs, root, cols, ncols, nrows, out);
}

/* filter_record_batch(batch, mask): keep rows whose mask is valid and true (filter.rs:34, memory.rs:92) */
int64_t qo_filter_indices(const qo_col* mask, int64_t* sel) {
  if (mask->type.id != QHIP_BOOL) { snprintf(g_err, sizeof g_err, "filter predicate is not Boolean (as_boolean() would panic, filter.rs:34)"); return -1; }
  int64_t m = 0;
  const uint8_t* v = (const uint8_t*)mask->values;
  for (int64_t i = 0; i < mask->n; ++i) if ((!mask->valid || mask->valid[i]) && v[i]) sel[m++] = i;
  return m;
}

/* ================================================================ u64 -> u64 hash map
 * Stand-in for std::collections::HashMap<u64, _> (hashbrown + RandomState SipHash-1-3 of the key):
 * open addressing, the u64 key is re-hashed with SipHash-1-3 like RandomState does (cost parity). */
typedef struct u64map {
  uint64_t* keys; uint64_t* vals; uint8_t* used; uint64_t cap; uint64_t len; uint64_t k0, k1;
} u64map;
static uint64_t u64map_hash(const u64map* m, uint64_t key) {
  qo_hasher h; qo_hasher_init(&h);
  h.v0 ^= m->k0; h.v1 ^= m->k1; h.v2 ^= m->k0; h.v3 ^= m->k1;
  qo_hasher_write(&h, (const uint8_t*)&key, 8);
  return qo_hasher_finish(&h);
}
static int u64map_init(u64map* m, uint64_t expect) {
  uint64_t cap = 16;
  while (cap < expect * 2) cap <<= 1;
  m->keys = (uint64_t*)malloc(cap * 8); m->vals = (uint64_t*)malloc(cap * 8); m->used = (uint8_t*)calloc(cap, 1);
  m->cap = cap; m->len = 0; m->k0 = 0x0123456789abcdefULL; m->k1 = 0xfedcba9876543210ULL;
  return (m->keys && m->vals && m->used) ? 0 : -1;
}
static void u64map_free(u64map* m) { free(m->keys); free(m->vals); free(m->used); memset(m, 0, sizeof *m); }
static uint64_t* u64map_find(const u64map* m, uint64_t key) {
  uint64_t s = u64map_hash(m, key) & (m->cap - 1);
  while (m->used[s]) { if (m->keys[s] == key) return &m->vals[s]; s = (s + 1) & (m->cap - 1); }
  return NULL;
}
static void u64map_grow(u64map* m);
static uint64_t* u64map_insert(u64map* m, uint64_t key, uint64_t val) {
  if ((m->len + 1) * 2 > m->cap) u64map_grow(m);
  uint64_t s = u64map_hash(m, key) & (m->cap - 1);
  while (m->used[s]) { if (m->keys[s] == key) { m->vals[s] = val; return &m->vals[s]; } s = (s + 1) & (m->cap - 1); }
  m->used[s] = 1; m->keys[s] = key; m->vals[s] = val; m->len++;
  return &m->vals[s];
}
static void u64map_grow(u64map* m) {
  u64map n = *m;
  n.cap = m->cap * 2;
  n.keys = (uint64_t*)malloc(n.cap * 8); n.vals = (uint64_t*)malloc(n.cap * 8); n.used = (uint8_t*)calloc(n.cap, 1); n.len = 0;
  for (uint64_t s = 0; s < m->cap; ++s) if (m->used[s]) {
    uint64_t t = u64map_hash(&n, m->keys[s]) & (n.cap - 1);
    while (n.used[t]) t = (t + 1) & (n.cap - 1);
    n.used[t] = 1; n.keys[t] = m->keys[s]; n.vals[t] = m->vals[s]; n.len++;
  }
  free(m->keys); free(m->vals); free(m->used);
  *m = n;
}

/* ================================================================ JoinHashMap (hash_join.rs:39-107) */
struct qo_join_map { u64map map; uint64_t* next; int64_t n; };

qo_join_map* qo_join_map_with_capacity(int64_t capacity) {
  qo_join_map* m = (qo_join_map*)calloc(1, sizeof *m);
  u64map_init(&m->map, (uint64_t)capacity);
  m->next = (uint64_t*)calloc((size_t)(capacity > 0 ? capacity : 1), 8);   /* vec![0; capacity] */
  m->n = capacity;
  return m;
}
void qo_join_map_free(qo_join_map* m) { if (m) { u64map_free(&m->map); free(m->next); free(m); } }
/* hash_join.rs:52-64 */
void qo_join_map_update(qo_join_map* m, const uint64_t* hashes, const int64_t* rows, int64_t nrows, int64_t delete_offset) {
  for (int64_t k = 0; k < nrows; ++k) {
    int64_t row = rows[k];
    uint64_t hash = hashes[row];
    uint64_t* index = u64map_find(&m->map, hash);
    if (index) {
      uint64_t pre = *index;
      *index = (uint64_t)(row + 1);
      m->next[row - delete_offset] = pre;
    } else {
      u64map_insert(&m->map, hash, (uint64_t)(row + 1));
    }
  }
}
int qo_join_map_is_distinct(const qo_join_map* m) { return (int64_t)m->map.len == m->n; }   /* :66-68 */
int64_t qo_join_map_len(const qo_join_map* m) { return (int64_t)m->map.len; }
int64_t qo_join_map_get(const qo_join_map* m, uint64_t hash) { uint64_t* v = u64map_find(&m->map, hash); return v ? (int64_t)*v : 0; }
const uint64_t* qo_join_map_next(const qo_join_map* m, int64_t* n) { if (n) *n = m->n; return m->next; }

/* hash_join.rs:70-107 */
int64_t qo_join_map_get_matches(const qo_join_map* m, const uint64_t* ph, int64_t np, uint32_t** input_indices, uint64_t** match_indices) {
  int64_t cap = np > 16 ? np : 16, cnt = 0;
  uint32_t* in = (uint32_t*)malloc((size_t)cap * 4);
  uint64_t* mt = (uint64_t*)malloc((size_t)cap * 8);
#define PUSH(a, b) do { if (cnt == cap) { cap *= 2; in = (uint32_t*)realloc(in, (size_t)cap * 4); mt = (uint64_t*)realloc(mt, (size_t)cap * 8); } in[cnt] = (a); mt[cnt] = (b); ++cnt; } while (0)
  if (qo_join_map_is_distinct(m)) {
    for (int64_t row = 0; row < np; ++row) {
      uint64_t* v = u64map_find(&m->map, ph[row]);
      if (v) PUSH((uint32_t)row, *v - 1);
    }
  } else {
    for (int64_t row = 0; row < np; ++row) {
      uint64_t* v = u64map_find(&m->map, ph[row]);
      if (!v) continue;
      uint64_t matched = *v - 1;
      for (;;) {
        PUSH((uint32_t)row, matched);
        uint64_t nx = m->next[matched];
        if (nx == 0) break;
        matched = nx - 1;
      }
    }
  }
#undef PUSH
  *input_indices = in; *match_indices = mt;
  return cnt;
}

static int key_equal(const qo_col* a, int64_t i, const qo_col* b, int64_t j, int* is_null) {
  if ((a->valid && !a->valid[i]) || (b->valid && !b->valid[j])) { *is_null = 1; return 0; }
  *is_null = 0;
  if (a->type.id == QHIP_UTF8) return str_cmp(a, i, b, j) == 0;
  if (is_float(a->type.id)) return f64_total_key(get_f64(a, i)) == f64_total_key(get_f64(b, j));
  return get_int(a, i) == get_int(b, j);
}

/* probe_hash_table (hash_join.rs:177-216): candidates by hash, then take + eq + and + filter on the keys.
 * A NULL on either side makes eq NULL, which the filter drops. */
int64_t qo_probe_hash_table(const qo_join_map* m, const qo_col* bk, const qo_col* pk, int nkeys, int64_t nprobe, uint64_t** build_idx,
                            uint32_t** probe_idx) {
  uint64_t* hashes = (uint64_t*)malloc((size_t)(nprobe > 0 ? nprobe : 1) * 8);
  if (qo_create_hashes(pk, nkeys, nprobe, hashes)) { free(hashes); return -1; }
  uint32_t* in; uint64_t* mt;
  int64_t cand = qo_join_map_get_matches(m, hashes, nprobe, &in, &mt);
  free(hashes);
  for (int c = 0; c < nkeys; ++c)
    if (!same_type(bk[c].type, pk[c].type)) { free(in); free(mt); snprintf(g_err, sizeof g_err, "Invalid argument error: Invalid comparison operation between join keys"); return -1; }
  int64_t out = 0;
  for (int64_t k = 0; k < cand; ++k) {
    int keep = 1;
    for (int c = 0; c < nkeys && keep; ++c) {
      int isnull;
      int eq = key_equal(&bk[c], (int64_t)mt[k], &pk[c], (int64_t)in[k], &isnull);
      if (isnull || !eq) keep = 0;
    }
    if (keep) { mt[out] = mt[k]; in[out] = in[k]; ++out; }
  }
  *build_idx = mt; *probe_idx = in;
  return out;
}

/* adjust_right_indices (join/mod.rs:176-207) */
int64_t qo_adjust_right_indices(const uint64_t* bi, const uint32_t* pi, int64_t n, int64_t right_rows, int64_t** ob, int64_t** op) {
  int64_t cap = n + right_rows + 1, cnt = 0;
  int64_t* b = (int64_t*)malloc((size_t)cap * 8);
  int64_t* p = (int64_t*)malloc((size_t)cap * 8);
  int64_t last = 0;
  for (int64_t k = 0; k < n; ++k) {
    for (int64_t v = last; v < (int64_t)pi[k]; ++v) { p[cnt] = v; b[cnt] = -1; ++cnt; }
    p[cnt] = pi[k]; b[cnt] = (int64_t)bi[k]; ++cnt;
    last = (int64_t)pi[k] + 1;
  }
  for (int64_t v = last; v < right_rows; ++v) { p[cnt] = v; b[cnt] = -1; ++cnt; }
  *ob = b; *op = p;
  return cnt;
}

/* ================================================================ accumulators (physical/expr/aggregate/{sum,avg,count,min,max}.rs) */
typedef struct acc {
  int kind; qhip_dtype ret; qhip_dtype arg;
  int has_sum; i128 isum; double fsum; uint64_t count;   /* SumAccumulator / Avg accumulators */
  int has_res; i128 ires; double fres;                   /* PrimitiveAccumulator (min/max) */
} acc;

/* arrow::compute::{sum,min,max} over `idx` rows of col: returns 0 when there is no non-null value */
static int reduce_rows(const qo_col* c, const int64_t* idx, int64_t n, int what /*0 sum,1 min,2 max*/, i128* iv, double* fv, uint64_t* nonnull) {
  int any = 0;
  uint64_t nn = 0;
  i128 is = 0; double fs = 0;
  int fl = is_float(c->type.id);
  for (int64_t k = 0; k < n; ++k) {
    int64_t r = idx ? idx[k] : k;
    if (c->type.id == QHIP_NULL || (c->valid && !c->valid[r])) continue;
    ++nn;
    if (fl) {
      double v = get_f64(c, r);
      if (!any) fs = v;
      else if (what == 0) fs += v;
      else if (what == 1) { if (f64_total_key(v) < f64_total_key(fs)) fs = v; }
      else { if (f64_total_key(v) > f64_total_key(fs)) fs = v; }
    } else {
      i128 v = get_int(c, r);
      if (!any) is = v;
      else if (what == 0) is = (i128)((u128)is + (u128)v);
      else if (what == 1) { if (v < is) is = v; }
      else { if (v > is) is = v; }
    }
    any = 1;
  }
  *iv = is; *fv = fs; *nonnull = nn;
  return any;
}
static i128 wrap_to(int id, i128 v) {
  switch (id) {
    case QHIP_INT64: return (i128)(int64_t)v;
    case QHIP_UINT64: return (i128)(uint64_t)v;
    default: return v;
  }
}
static void native_extreme(qhip_dtype t, int want_max, i128* iv, double* fv) {
  *fv = want_max ? (t.id == QHIP_FLOAT32 ? FLT_MAX : DBL_MAX) : -(t.id == QHIP_FLOAT32 ? FLT_MAX : DBL_MAX);
  if (t.id == QHIP_DECIMAL128) { u128 mx = ((u128)1 << 127) - 1; *iv = want_max ? (i128)mx : -(i128)mx - 1; return; }
  i128 lo, hi; int_limits(t.id, &lo, &hi);
  *iv = want_max ? hi : lo;
}
/* Accumulator::accumluate(&ArrayRef) on the rows idx[0..n) of the argument column */
static int acc_accumulate(acc* a, const qo_col* c, const int64_t* idx, int64_t n) {
  i128 iv; double fv; uint64_t nn;
  switch (a->kind) {
    case QHIP_AGG_COUNT: /* count.rs:41-44 */
      reduce_rows(c, idx, n, 0, &iv, &fv, &nn);
      a->count += nn;
      return 0;
    case QHIP_AGG_SUM:   /* sum.rs:71-81 */
    case QHIP_AGG_AVG: { /* avg.rs:119-130 */
      int any = reduce_rows(c, idx, n, 0, &iv, &fv, &nn);
      if (a->kind == QHIP_AGG_AVG) a->count += nn;
      if (any) {
        if (!a->has_sum) { a->has_sum = 1; a->isum = 0; a->fsum = 0; }
        a->isum = wrap_to(a->kind == QHIP_AGG_AVG ? QHIP_DECIMAL128 : a->ret.id, (i128)((u128)a->isum + (u128)iv));
        a->fsum += fv;
      }
      return 0;
    }
    case QHIP_AGG_MIN:   /* min.rs:12-28 + PrimitiveAccumulator (aggregate/mod.rs:60-84) */
    case QHIP_AGG_MAX: { /* max.rs:12-28 */
      int is_min = a->kind == QHIP_AGG_MIN;
      int any = reduce_rows(c, idx, n, is_min ? 1 : 2, &iv, &fv, &nn);
      i128 ci; double cf;
      if (a->has_res) { ci = a->ires; cf = a->fres; } else native_extreme(a->ret, is_min, &ci, &cf);
      if (any) {
        if (is_float(c->type.id)) { if (is_min ? (cf > fv) : (cf < fv)) cf = fv; }   /* PartialOrd: false for NaN */
        else { if (is_min ? (ci > iv) : (ci < iv)) ci = iv; }
      }
      a->has_res = 1; a->ires = ci; a->fres = cf;
      return 0;
    }
  }
  QO_FAIL(QHIP_INVALID_ARGUMENT, "unknown aggregate kind %d", a->kind);
}
/* Accumulator::evaluate() -> ScalarValue, written into row g of the result column */
static int acc_evaluate(const acc* a, qo_col* out, int64_t g) {
  switch (a->kind) {
    case QHIP_AGG_COUNT: ((int64_t*)out->values)[g] = (int64_t)a->count; return 0;
    case QHIP_AGG_SUM:
      if (!a->has_sum) { out->valid[g] = 0; return 0; }
      if (a->ret.id == QHIP_FLOAT64) ((double*)out->values)[g] = a->fsum; else put_int(out, g, a->isum);
      return 0;
    case QHIP_AGG_AVG:
      if (!a->has_sum) { out->valid[g] = 0; return 0; }
      if (a->ret.id == QHIP_FLOAT64) { ((double*)out->values)[g] = a->fsum / (double)a->count; return 0; }
      {
        /* avg.rs:91-116 */
        if (a->ret.scale < a->arg.scale) QO_FAIL(QHIP_EXEC_ERROR, "Internal error: Arithmetic Overflow in DecimalAvgAccumulator");
        i128 value;
        if (__builtin_mul_overflow(a->isum, pow10_128(a->ret.scale - a->arg.scale), &value))
          QO_FAIL(QHIP_EXEC_ERROR, "AVG(Decimal128): sum * 10^k overflows i128 (reference yields a mistyped NULL, avg.rs:105-116)");
        i128 lim = pow10_128(a->ret.precision);
        if (value >= lim || value <= -lim)
          QO_FAIL(QHIP_EXEC_ERROR, "AVG(Decimal128): scaled sum exceeds the result precision (reference yields a mistyped NULL, avg.rs:105-116)");
        ((i128*)out->values)[g] = value / (i128)a->count;
        return 0;
      }
    case QHIP_AGG_MIN:
    case QHIP_AGG_MAX:
      if (!a->has_res) { out->valid[g] = 0; return 0; }
      if (a->ret.id == QHIP_FLOAT64) ((double*)out->values)[g] = a->fres;
      else if (a->ret.id == QHIP_FLOAT32) ((float*)out->values)[g] = (float)a->fres;
      else put_int(out, g, a->ires);
      return 0;
  }
  return 0;
}
static int acc_check(const qhip_agg* ag, const qo_col* arg) {
  qhip_dtype rt = ag->return_type;
  switch (ag->kind) {
    case QHIP_AGG_SUM:
      if (!(rt.id == QHIP_UINT64 || rt.id == QHIP_INT64 || rt.id == QHIP_FLOAT64 || rt.id == QHIP_DECIMAL128))
        QO_FAIL(QHIP_INVALID_ARGUMENT, "Internal error: Sum not supported for return type %d", rt.id);
      if (arg->type.id != rt.id) QO_FAIL(QHIP_INVALID_ARGUMENT, "SUM argument type does not match return type");
      return 0;
    case QHIP_AGG_AVG:
      if (!((arg->type.id == QHIP_DECIMAL128 && rt.id == QHIP_DECIMAL128) || (arg->type.id == QHIP_FLOAT64 && rt.id == QHIP_FLOAT64)))
        QO_FAIL(QHIP_INVALID_ARGUMENT, "Internal error: Unsupported data type for AVG aggregate");
      return 0;
    case QHIP_AGG_MIN: case QHIP_AGG_MAX:
      if (!same_type(arg->type, rt)) QO_FAIL(QHIP_INVALID_ARGUMENT, "MIN/MAX argument type differs from return type");
      return 0;
    default: return 0;
  }
}

/* GroupAccumulator::update / output (hash.rs:45-107); NoGroupingAggregate::execute (no_grouping.rs:30-62) */
int qo_hash_aggregate(const qo_col* keys, int n_keys, const qo_col* args, const qhip_agg* aggs, int n_aggs, int64_t nrows,
                      const int64_t* batch_offsets, int64_t n_batches, qo_agg_result* out) {
  memset(out, 0, sizeof *out);
  out->n_aggs = n_aggs;
  for (int a = 0; a < n_aggs; ++a) { int rc = acc_check(&aggs[a], &args[a]); if (rc) return rc; }
  int64_t G = 0;
  acc* accs = NULL;
  int64_t* first_row = NULL;
  if (n_keys == 0) {
    /* no_grouping.rs: one accumulator set; one accumulate call per input batch */
    G = 1;
    accs = (acc*)calloc((size_t)(n_aggs > 0 ? n_aggs : 1), sizeof(acc));
    first_row = (int64_t*)calloc(1, 8);
    for (int a = 0; a < n_aggs; ++a) {
      accs[a].kind = aggs[a].kind; accs[a].ret = aggs[a].return_type; accs[a].arg = args[a].type;
      for (int64_t b = 0; b < n_batches; ++b) {
        int64_t r0 = batch_offsets[b], r1 = batch_offsets[b + 1];
        int64_t* idx = (int64_t*)malloc((size_t)(r1 - r0 > 0 ? r1 - r0 : 1) * 8);
        for (int64_t r = r0; r < r1; ++r) idx[r - r0] = r;
        int rc = acc_accumulate(&accs[a], &args[a], idx, r1 - r0);
        free(idx);
        if (rc) { free(accs); free(first_row); return rc; }
      }
    }
  } else {
    /* hash.rs:46-49: hashes_buffer + create_hashes */
    uint64_t* hashes = (uint64_t*)malloc((size_t)(nrows > 0 ? nrows : 1) * 8);
    int rc = qo_create_hashes(keys, n_keys, nrows, hashes);
    if (rc) { free(hashes); return rc; }
    /* hash.rs:50-71: map hash -> group (identified by the first row); per-group row-index lists */
    u64map map; u64map_init(&map, 1024);
    int64_t gcap = 1024;
    first_row = (int64_t*)malloc((size_t)gcap * 8);
    int64_t** lists = (int64_t**)malloc((size_t)gcap * sizeof(int64_t*));
    int64_t* llen = (int64_t*)malloc((size_t)gcap * 8);
    int64_t* lcap = (int64_t*)malloc((size_t)gcap * 8);
    for (int64_t row = 0; row < nrows; ++row) {
      uint64_t* gi = u64map_find(&map, hashes[row]);
      int64_t g;
      if (gi) g = (int64_t)*gi;
      else {
        if (G == gcap) {
          gcap *= 2;
          first_row = (int64_t*)realloc(first_row, (size_t)gcap * 8);
          lists = (int64_t**)realloc(lists, (size_t)gcap * sizeof(int64_t*));
          llen = (int64_t*)realloc(llen, (size_t)gcap * 8);
          lcap = (int64_t*)realloc(lcap, (size_t)gcap * 8);
        }
        g = G++;
        u64map_insert(&map, hashes[row], (uint64_t)g);
        first_row[g] = row;
        lcap[g] = 4; llen[g] = 0;
        lists[g] = (int64_t*)malloc(4 * 8);
      }
      if (llen[g] == lcap[g]) { lcap[g] *= 2; lists[g] = (int64_t*)realloc(lists[g], (size_t)lcap[g] * 8); }
      lists[g][llen[g]++] = row;
    }
    free(hashes);
    u64map_free(&map);
    /* hash.rs:73-84: per group, per aggregate: take(values, indices) then accumulate */
    accs = (acc*)calloc((size_t)(G * n_aggs > 0 ? G * n_aggs : 1), sizeof(acc));
    for (int64_t g = 0; g < G; ++g) {
      for (int a = 0; a < n_aggs; ++a) {
        acc* A = &accs[g * n_aggs + a];
        A->kind = aggs[a].kind; A->ret = aggs[a].return_type; A->arg = args[a].type;
        /* UInt64Array::from_iter(indices.clone()) — re-cloned per aggregate in the reference */
        int64_t* idx = (int64_t*)malloc((size_t)llen[g] * 8);
        memcpy(idx, lists[g], (size_t)llen[g] * 8);
        rc = acc_accumulate(A, &args[a], idx, llen[g]);
        free(idx);
        if (rc) return rc;
      }
    }
    for (int64_t g = 0; g < G; ++g) free(lists[g]);
    free(lists); free(llen); free(lcap);
  }
  out->n_groups = G;
  out->first_row = first_row;
  out->agg_cols = (qo_col*)calloc((size_t)(n_aggs > 0 ? n_aggs : 1), sizeof(qo_col));
  for (int a = 0; a < n_aggs; ++a) {
    qhip_dtype rt = aggs[a].kind == QHIP_AGG_COUNT ? (qhip_dtype){QHIP_INT64, 0, 0} : aggs[a].return_type;
    int rc = col_alloc(&out->agg_cols[a], rt, G, 1);
    if (rc) return rc;
    for (int64_t g = 0; g < G; ++g) {
      rc = acc_evaluate(&accs[g * n_aggs + a], &out->agg_cols[a], g);
      if (rc) { free(accs); return rc; }
    }
  }
  free(accs);
  return 0;
}

void qo_agg_result_free(qo_agg_result* r) {
  if (!r) return;
  free(r->first_row);
  for (int a = 0; a < r->n_aggs; ++a) qo_col_free(&r->agg_cols[a]);
  free(r->agg_cols);
  memset(r, 0, sizeof *r);
}

/* ================================================================ whole-pipeline restatement (timed CPU baseline)
 * Scan(filter) -> HashAggregate exactly as the reference runs it for Q1-shaped plans:
 *   per stored batch: evaluate the predicate (literal broadcast + cast + compare, memory.rs:90-91),
 *                     filter_record_batch over EVERY column (memory.rs:92; no projection pushdown, planner/mod.rs:251-256)
 *   concat_batches of all filtered batches (hash.rs:150)
 *   evaluate group and aggregate-argument expressions over the concatenated batch (hash.rs:152-162)
 *   GroupAccumulator::update + output (hash.rs:45-107)
 * batches[b * ncols + c] is column c of batch b. pred_root < 0: no filter. */
static void gather_col(const qo_col* in, const int64_t* sel, int64_t m, qo_col* out) {
  memset(out, 0, sizeof *out);
  out->type = in->type; out->n = m; out->owned = 1;
  int w = type_width(in->type.id);
  if (in->valid) { out->valid = (uint8_t*)malloc((size_t)(m > 0 ? m : 1)); for (int64_t k = 0; k < m; ++k) out->valid[k] = in->valid[sel[k]]; }
  if (in->type.id == QHIP_UTF8) {
    out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m + 1));
    int64_t total = 0;
    for (int64_t k = 0; k < m; ++k) total += in->offsets[sel[k] + 1] - in->offsets[sel[k]];
    out->data = (uint8_t*)malloc((size_t)(total > 0 ? total : 1));
    int32_t pos = 0;
    for (int64_t k = 0; k < m; ++k) {
      int32_t b = in->offsets[sel[k]], len = in->offsets[sel[k] + 1] - b;
      out->offsets[k] = pos;
      memcpy(out->data + pos, in->data + b, (size_t)len);
      pos += len;
    }
    out->offsets[m] = pos;
  } else if (w) {
    out->values = malloc((size_t)(m > 0 ? m : 1) * (size_t)w);
    const uint8_t* src = (const uint8_t*)in->values;
    uint8_t* dst = (uint8_t*)out->values;
    switch (w) {
      case 4: for (int64_t k = 0; k < m; ++k) ((uint32_t*)dst)[k] = ((const uint32_t*)src)[sel[k]]; break;
      case 8: for (int64_t k = 0; k < m; ++k) ((uint64_t*)dst)[k] = ((const uint64_t*)src)[sel[k]]; break;
      case 16: for (int64_t k = 0; k < m; ++k) ((u128*)dst)[k] = ((const u128*)src)[sel[k]]; break;
      default: for (int64_t k = 0; k < m; ++k) memcpy(dst + (size_t)k * w, src + (size_t)sel[k] * w, (size_t)w);
    }
  }
}

static void concat_cols(const qo_col* parts, int64_t nparts, int stride, qo_col* out) {
  /* parts[k * stride] for k in 0..nparts */
  memset(out, 0, sizeof *out);
  if (nparts == 0) return;
  out->type = parts[0].type; out->owned = 1;
  int64_t n = 0, bytes = 0;
  int any_valid = 0;
  for (int64_t k = 0; k < nparts; ++k) {
    const qo_col* p = &parts[k * stride];
    n += p->n;
    if (p->valid) any_valid = 1;
    if (p->type.id == QHIP_UTF8) bytes += p->offsets[p->n] - p->offsets[0];
  }
  out->n = n;
  int w = type_width(out->type.id);
  if (any_valid) out->valid = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
  if (out->type.id == QHIP_UTF8) { out->offsets = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1)); out->data = (uint8_t*)malloc((size_t)(bytes > 0 ? bytes : 1)); }
  else if (w) out->values = malloc((size_t)(n > 0 ? n : 1) * (size_t)w);
  int64_t pos = 0; int32_t bpos = 0;
  for (int64_t k = 0; k < nparts; ++k) {
    const qo_col* p = &parts[k * stride];
    if (any_valid) { if (p->valid) memcpy(out->valid + pos, p->valid, (size_t)p->n); else memset(out->valid + pos, 1, (size_t)p->n); }
    if (out->type.id == QHIP_UTF8) {
      int32_t base = p->offsets[0];
      for (int64_t i = 0; i < p->n; ++i) out->offsets[pos + i] = p->offsets[i] - base + bpos;
      memcpy(out->data + bpos, p->data + base, (size_t)(p->offsets[p->n] - base));
      bpos += p->offsets[p->n] - base;
    } else if (w) memcpy((uint8_t*)out->values + (size_t)pos * w, p->values, (size_t)p->n * w);
    pos += p->n;
  }
  if (out->type.id == QHIP_UTF8) out->offsets[n] = bpos;
}

int qo_scan_filter_aggregate(const qo_col* batches, int64_t nbatches, int ncols, const int64_t* batch_rows, const qhip_expr* exprs,
                             int n_exprs, int pred_root, const int32_t* group_roots, int n_groups, const qhip_agg* aggs, int n_aggs,
                             qo_agg_result* out, qo_col* out_keys /* n_groups result key columns, caller frees */, int64_t* rows_after_filter) {
  int rc = 0;
  qo_col* filtered = (qo_col*)calloc((size_t)(nbatches * ncols > 0 ? nbatches * ncols : 1), sizeof(qo_col));
  int64_t total = 0;
  for (int64_t b = 0; b < nbatches && !rc; ++b) {
    const qo_col* bc = &batches[b * ncols];
    const int64_t n = batch_rows[b];
    if (pred_root >= 0) {
      qo_col mask;
      rc = eval_node(exprs, n_exprs, pred_root, bc, ncols, n, &mask);
      if (rc) break;
      int64_t* sel = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
      int64_t m = qo_filter_indices(&mask, sel);
      qo_col_free(&mask);
      if (m < 0) { free(sel); rc = QHIP_INVALID_ARGUMENT; break; }
      for (int c = 0; c < ncols; ++c) gather_col(&bc[c], sel, m, &filtered[b * ncols + c]);
      free(sel);
      total += m;
    } else {
      for (int c = 0; c < ncols; ++c) { filtered[b * ncols + c] = bc[c]; filtered[b * ncols + c].owned = 0; filtered[b * ncols + c].n = n; }
      total += n;
    }
  }
  qo_col* cat = (qo_col*)calloc((size_t)(ncols > 0 ? ncols : 1), sizeof(qo_col));
  if (!rc) for (int c = 0; c < ncols; ++c) concat_cols(&filtered[c], nbatches, ncols, &cat[c]);
  for (int64_t k = 0; k < nbatches * ncols; ++k) qo_col_free(&filtered[k]);
  free(filtered);
  if (rows_after_filter) *rows_after_filter = total;
  qo_col* keys = (qo_col*)calloc((size_t)(n_groups > 0 ? n_groups : 1), sizeof(qo_col));
  qo_col* args = (qo_col*)calloc((size_t)(n_aggs > 0 ? n_aggs : 1), sizeof(qo_col));
  for (int k = 0; k < n_groups && !rc; ++k) rc = eval_node(exprs, n_exprs, group_roots[k], cat, ncols, total, &keys[k]);
  for (int a = 0; a < n_aggs && !rc; ++a) rc = eval_node(exprs, n_exprs, aggs[a].expr, cat, ncols, total, &args[a]);
  if (!rc) {
    int64_t offs[2] = {0, total};
    rc = qo_hash_aggregate(keys, n_groups, args, aggs, n_aggs, total, offs, nbatches > 0 ? 1 : 0, out);
  }
  if (!rc && out_keys) for (int k = 0; k < n_groups; ++k) gather_col(&keys[k], out->first_row, out->n_groups, &out_keys[k]);
  for (int k = 0; k < n_groups; ++k) qo_col_free(&keys[k]);
  for (int a = 0; a < n_aggs; ++a) qo_col_free(&args[a]);
  for (int c = 0; c < ncols; ++c) qo_col_free(&cat[c]);
  free(keys); free(args); free(cat);
  return rc;
}
