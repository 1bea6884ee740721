// synth.cpp — counter-based synthetic TPC-H-shaped inputs.
// Row i of column c is a pure function of (seed, c, i): u(i,c) = splitmix64((seed ^ (c << 56)) + i), so
// Python, the CPU oracle and this library produce identical data for any row range without shipping files.
#include <cstdint>
#include <cstring>

#include "../../include/qhip_bench.h"

enum { QHIP_OK = 0, QHIP_INVALID_ARGUMENT = 1 };   // (the values of include/qhip.h; this library does not depend on it)

namespace {
constexpr uint64_t kSeed = 0x515552494F555301ULL;   // "QURIOUS\x01"
inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline uint64_t u(uint64_t i, uint64_t c) { return splitmix64((kSeed ^ (c << 56)) + i); }
inline void put_i128(void* base, int64_t k, int64_t v) {
  int64_t* p = (int64_t*)base + 2 * k;
  p[0] = v;
  p[1] = v < 0 ? -1 : 0;
}
}  // namespace

extern "C" int qhip_synth_lineitem(int64_t first_row, int64_t n_rows, int32_t* l_shipdate, int32_t* l_returnflag_offsets,
                                   uint8_t* l_returnflag_data, int32_t* l_linestatus_offsets, uint8_t* l_linestatus_data,
                                   void* l_quantity, void* l_extendedprice, void* l_discount, void* l_tax) {
  if (first_row < 0 || n_rows < 0) return QHIP_INVALID_ARGUMENT;
  for (int64_t k = 0; k < n_rows; ++k) {
    const uint64_t i = (uint64_t)(first_row + k);
    const int32_t ship = 8036 + (int32_t)(u(i, 0) % 2526);          // 1992-01-02 .. 1998-12-01
    if (l_shipdate) l_shipdate[k] = ship;
    const bool late = ship > 9298;                                   // 1995-06-17
    if (l_returnflag_data) {
      const uint64_t r = u(i, 1);
      l_returnflag_data[k] = late ? 'N' : (((r >> 8) % 64 == 0) ? 'N' : ((r & 1) ? 'A' : 'R'));
    }
    if (l_returnflag_offsets) l_returnflag_offsets[k] = (int32_t)k;
    if (l_linestatus_data) l_linestatus_data[k] = late ? 'O' : 'F';
    if (l_linestatus_offsets) l_linestatus_offsets[k] = (int32_t)k;
    if (l_quantity) put_i128(l_quantity, k, 100 * (1 + (int64_t)(u(i, 2) % 50)));
    if (l_extendedprice) put_i128(l_extendedprice, k, 90100 + (int64_t)(u(i, 3) % 10404900));
    if (l_discount) put_i128(l_discount, k, (int64_t)(u(i, 4) % 11));
    if (l_tax) put_i128(l_tax, k, (int64_t)(u(i, 5) % 9));
  }
  if (l_returnflag_offsets) l_returnflag_offsets[n_rows] = (int32_t)n_rows;
  if (l_linestatus_offsets) l_linestatus_offsets[n_rows] = (int32_t)n_rows;
  return QHIP_OK;
}

// ---------------------------------------------------------------- Q3 tables: customer, orders, lineitem-of-orders
namespace {
const char* const kSegments[5] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"};
inline int64_t sparse_orderkey(int64_t k) { return ((k - 1) / 8) * 32 + (k - 1) % 8 + 1; }   // TPC-H's sparse o_orderkey
inline int lines_of(int64_t k) { return 1 + (int)(u((uint64_t)k, 11) % 7); }
inline int32_t orderdate_of(int64_t k) { return 8035 + (int32_t)(u((uint64_t)k, 10) % 2406); }
}  // namespace

extern "C" {

// customers first_key .. first_key + n - 1 (c_custkey is 1-based); seg_data must hold 10 * n bytes
int qhip_synth_customer(int64_t first_key, int64_t n, int64_t* c_custkey, int32_t* seg_offsets, uint8_t* seg_data) {
  if (first_key < 1 || n < 0) return QHIP_INVALID_ARGUMENT;
  int32_t pos = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t key = first_key + i;
    c_custkey[i] = key;
    const char* s = kSegments[u((uint64_t)key, 8) % 5];
    seg_offsets[i] = pos;
    const size_t len = strlen(s);
    memcpy(seg_data + pos, s, len);
    pos += (int32_t)len;
  }
  seg_offsets[n] = pos;
  return QHIP_OK;
}

// orders with ordinal first_k .. first_k + n - 1 (1-based); o_custkey uniform over 1..n_customers
int qhip_synth_orders(int64_t first_k, int64_t n, int64_t n_customers, int64_t* o_orderkey, int64_t* o_custkey, int32_t* o_orderdate,
                      int64_t* o_shippriority) {
  if (first_k < 1 || n < 0 || n_customers < 1) return QHIP_INVALID_ARGUMENT;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t k = first_k + i;
    o_orderkey[i] = sparse_orderkey(k);
    o_custkey[i] = 1 + (int64_t)(u((uint64_t)k, 9) % (uint64_t)n_customers);
    o_orderdate[i] = orderdate_of(k);
    o_shippriority[i] = 0;
  }
  return QHIP_OK;
}

int64_t qhip_synth_q3_lineitem_count(int64_t first_k, int64_t n_orders) {
  int64_t c = 0;
  for (int64_t i = 0; i < n_orders; ++i) c += lines_of(first_k + i);
  return c;
}

// the 1..7 lineitems of orders first_k .. first_k + n_orders - 1, in order; buffers sized by qhip_synth_q3_lineitem_count
int qhip_synth_q3_lineitem(int64_t first_k, int64_t n_orders, int64_t* l_orderkey, int32_t* l_shipdate, void* l_extendedprice,
                           void* l_discount) {
  if (first_k < 1 || n_orders < 0) return QHIP_INVALID_ARGUMENT;
  int64_t r = 0;
  for (int64_t i = 0; i < n_orders; ++i) {
    const int64_t k = first_k + i;
    const int nl = lines_of(k);
    const int64_t okey = sparse_orderkey(k);
    const int32_t od = orderdate_of(k);
    for (int l = 0; l < nl; ++l, ++r) {
      const uint64_t id = (uint64_t)k * 8 + (uint64_t)l;
      l_orderkey[r] = okey;
      l_shipdate[r] = od + 1 + (int32_t)(u(id, 12) % 121);
      put_i128(l_extendedprice, r, 90100 + (int64_t)(u(id, 13) % 10404900));
      put_i128(l_discount, r, (int64_t)(u(id, 14) % 11));
    }
  }
  return QHIP_OK;
}

}  // extern "C"
