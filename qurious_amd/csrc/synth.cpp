// synth.cpp — counter-based synthetic TPC-H-shaped inputs.
// Row i of column c is a pure function of (seed, c, i): u(i,c) = splitmix64((seed ^ (c << 56)) + i), so
// Python, the CPU oracle and this library produce identical data for any row range without shipping files.
#include <cstdint>
#include <cstring>

#include "../../include/qhip.h"

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
