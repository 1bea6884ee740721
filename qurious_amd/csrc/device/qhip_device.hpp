// qhip_device.hpp — hand-written gfx950 (CDNA4, wave64) device code of the qurious-hip backend.
//
// Everything here is plan-independent: 128-bit decimal arithmetic, wavefront
// ballot/DPP primitives, the two-level (LDS-staged + HBM) open-addressing group
// table, and the kernel templates. A plan (predicate, key and aggregate-argument
// expressions) enters as a small generated policy struct `P` that codegen.cpp
// emits from the qhip_expr tree; the kernel bodies below are instantiated with it
// by hiprtc at operator-execute time (and by hipcc at build time for the catalog
// in kernels_aot.hip). No MFMA anywhere: this path is integer/hash/gather work
// bounded by HBM bandwidth.
//
// The header is self-contained (no #include) so that hiprtc can compile it from a
// string; hipcc builds include <hip/hip_runtime.h> before it.
#pragma once

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;
typedef __int128 i128;
typedef unsigned __int128 u128;

#define QH_MAXC 24   // distinct input columns one kernel may reference
#define QH_MAXL 24   // literal slots
#define QH_WAVE 64
#define QH_BLOCK 256

// One input column as the kernel sees it (Arrow layout, concatenated over batches).
struct KCol {
  const void* v;  // fixed-width values, or int32 offsets (n+1) for Utf8, or bit-packed values for Boolean
  const u8* n;    // validity bitmap (LSB order) or nullptr
  const u8* d;    // Utf8 data bytes
};

struct KArgs {
  KCol c[QH_MAXC];
  u64 lit_lo[QH_MAXL];   // integer / date / bool literals (sign-extended), f64 bit patterns, Decimal128 low half
  i64 lit_hi[QH_MAXL];   // Decimal128 high half
  const u8* strlit;      // concatenated Utf8 literals
  int stroff[QH_MAXL + 1];
  i64 nrows;
  const u32* nrows_dev;  // when set: the table's row count lives on the device (a hash join whose output size the host has
                         // not waited for) and nrows is its capacity — the kernels that accept such inputs (qh_rows) stop there
};
// rows of the input: nrows, or the device-side count of a join output the host did not wait for (never above nrows)
__device__ __forceinline__ i64 qh_rows(const KArgs& a) {
  i64 n = a.nrows;
  if (a.nrows_dev) { const i64 d = (i64)*a.nrows_dev; n = d < n ? d : n; }
  return n;
}

// status word indices (QS_*): qhip_status.h, prepended to this file when it is embedded for hiprtc

// ------------------------------------------------------------------ scalar helpers
__device__ __forceinline__ u64 qh_mix64(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__device__ __forceinline__ bool qh_bit(const u8* bm, i64 i) { return (bm[i >> 3] >> (i & 7)) & 1; }
__device__ __forceinline__ int qh_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ i128 qh_mk128(u64 lo, i64 hi) { return (i128)(((u128)(u64)hi << 64) | (u128)lo); }
typedef unsigned int qh_v4u __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-byte load of a Decimal128 value: +10 % read bandwidth over plain loads (measured, MI355X)
__device__ __forceinline__ i128 qh_nt_load_i128(const i128* p) {
  const qh_v4u v = __builtin_nontemporal_load((const qh_v4u*)p);
  return (i128)(((u128)(((u64)v.w << 32) | v.z) << 64) | (u128)(((u64)v.y << 32) | v.x));
}
__device__ __forceinline__ double qh_f64(u64 bits) { return __longlong_as_double((i64)bits); }
// f64 <-> u64 whose unsigned order is the IEEE total order (arrow's min/max kernels compare floats that way)
__device__ __forceinline__ u64 qh_f64_ord(double d) { u64 b = (u64)__double_as_longlong(d); return (b >> 63) ? ~b : (b | 0x8000000000000000ULL); }
__device__ __forceinline__ double qh_ord_f64(u64 k) { u64 b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k; return __longlong_as_double((i64)b); }

// Utf8 value of at most 7 bytes packed injectively into one key word: byte 7 = length, bytes 0..len-1 = data.
// Branch-free: one unaligned 8-byte load (every Utf8 data buffer is allocated with >= 8 bytes of slack), masked to
// the value's length; a loop of byte loads would put control flow between the loads of a tile's rows.
typedef u64 __attribute__((aligned(1))) qh_u64_unaligned;
typedef u32 __attribute__((aligned(1))) qh_u32_unaligned;
typedef u16 __attribute__((aligned(1))) qh_u16_unaligned;
__device__ __forceinline__ u64 qh_pack_str7(const u8* p, int len) {
  const u64 raw = *(const qh_u64_unaligned*)p;
  const int l = len > 7 ? 7 : len;
  const u64 mask = l ? (~0ULL >> (64 - 8 * l)) : 0ULL;
  return (raw & mask) | ((u64)l << 56);
}
// N-word form for values of up to 8 N - 1 bytes: bytes little-endian across the words, length in the top byte of the last
template <int N> __device__ __forceinline__ void qh_pack_str(const u8* p, int len, u64* out) {
  const int l = len > 8 * N - 1 ? 8 * N - 1 : len;
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const u64 raw = *(const qh_u64_unaligned*)(p + 8 * w);
    const int nb = l - 8 * w;   // bytes of the value inside word w
    const u64 mask = nb <= 0 ? 0ULL : (nb >= 8 ? ~0ULL : (~0ULL >> (64 - 8 * nb)));
    out[w] = raw & mask;
  }
  out[N - 1] |= (u64)l << 56;
}
// the same packing from bytes already loaded as N unaligned words (load phase of a split load/eval policy)
template <int N> __device__ __forceinline__ void qh_pack_words(const u64* raw, int len, u64* out) {
  const int l = len > 8 * N - 1 ? 8 * N - 1 : len;
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const int nb = l - 8 * w;
    const u64 mask = nb <= 0 ? 0ULL : (nb >= 8 ? ~0ULL : (~0ULL >> (64 - 8 * nb)));
    out[w] = raw[w] & mask;
  }
  out[N - 1] |= (u64)l << 56;
}
// bytewise compare like arrow's Utf8 ordering: <0, 0, >0
__device__ __forceinline__ int qh_strcmp(const u8* a, int la, const u8* b, int lb) {
  int n = la < lb ? la : lb;
  for (int k = 0; k < n; ++k) { int d = (int)a[k] - (int)b[k]; if (d) return d; }
  return la - lb;
}
// SQL LIKE (like.rs:28-43 -> arrow `like`): pat holds literal bytes, 0xFF = % (any sequence of characters), 0xFE = _ (one
// character, i.e. one UTF-8 code point). Iterative wildcard match with backtracking to the last %.
__device__ __forceinline__ int qh_next_char(const u8* s, int i, int n) {
  ++i;
  while (i < n && (s[i] & 0xC0) == 0x80) ++i;
  return i;
}
// ... with the pattern as it stands in a COLUMN (round 4: like.rs:28-43 evaluates the pattern per row — arrow's `like` over two
// arrays): % and _ are wildcards, a backslash makes the next byte stand for itself (a trailing one is a plain backslash).
__device__ __forceinline__ bool qh_like_raw(const u8* s, int n, const u8* pat, int m) {
  int i = 0, p = 0, star = -1, mark = 0;
  while (i < n) {
    bool adv = false;
    if (p < m) {
      const u8 c = pat[p];
      if (c == (u8)'%') { star = p; mark = i; ++p; adv = true; }
      else if (c == (u8)'_') { i = qh_next_char(s, i, n); ++p; adv = true; }
      else if (c == (u8)'\\' && p + 1 < m) { if (pat[p + 1] == s[i]) { ++i; p += 2; adv = true; } }
      else if (c == s[i]) { ++i; ++p; adv = true; }
    }
    if (!adv) {
      if (star < 0) return false;
      p = star + 1; mark = qh_next_char(s, mark, n); i = mark;
    }
  }
  while (p < m && pat[p] == (u8)'%') ++p;
  return p == m;
}
__device__ __forceinline__ bool qh_like(const u8* s, int n, const u8* pat, int m) {
  int i = 0, p = 0, star = -1, mark = 0;
  while (i < n) {
    if (p < m && pat[p] == 0xFE) { i = qh_next_char(s, i, n); ++p; }
    else if (p < m && pat[p] == 0xFF) { star = p; mark = i; ++p; }
    else if (p < m && pat[p] == s[i]) { ++i; ++p; }
    else if (star >= 0) { p = star + 1; mark = qh_next_char(s, mark, n); i = mark; }
    else return false;
  }
  while (p < m && pat[p] == 0xFF) ++p;
  return p == m;
}
__device__ __forceinline__ bool qh_streq(const u8* a, int la, const u8* b, int lb) {
  if (la != lb) return false;
  for (int k = 0; k < la; ++k) if (a[k] != b[k]) return false;
  return true;
}
// value == literal, the literal's first 8 bytes given as a word (wave-uniform): literals of up to 8 bytes are ONE unaligned
// 8-byte load masked to the literal's length (every Utf8 data buffer carries >= 8 bytes of slack) instead of a byte loop
__device__ __forceinline__ bool qh_streq_lit(const u8* a, int la, const u8* lit, int llit, u64 lit_word) {
  if (llit > 8) return qh_streq(a, la, lit, llit);
  const u64 raw = *(const qh_u64_unaligned*)a;
  const u64 mask = llit >= 8 ? ~0ULL : ((1ULL << (8 * llit)) - 1ULL);
  return la == llit && ((raw ^ lit_word) & mask) == 0;
}
// Wrapping 128-bit product with wave-uniform fast paths: Decimal128 operands of real tables are small (TPC-H money
// fits 32..40 bits), and a generic 128 x 128 multiply costs ~45 VALU instructions. If every active lane's operands
// fit i32 the product is one 32 x 32 -> 64 multiply; if they fit i64 it is a 64 x 64 -> 128 multiply; else the full one.
__device__ __forceinline__ i128 qh_mul_i128(i128 a, i128 b) {
  const i64 alo = (i64)(u64)(u128)a, blo = (i64)(u64)(u128)b;
  const bool fit64 = ((i128)alo == a) && ((i128)blo == b);
  const bool fit32 = fit64 && ((i64)(int)alo == alo) && ((i64)(int)blo == blo);
  if (__all(fit32)) return (i128)((i64)(int)alo * (i64)(int)blo);
  if (__all(fit64)) {
    const u64 lo = (u64)alo * (u64)blo;
    const i64 hi = __mul64hi(alo, blo);
    return qh_mk128(lo, hi);
  }
  return (i128)((u128)a * (u128)b);
}
__device__ __forceinline__ i128 qh_mul_i128_plain(i128 a, i128 b) { return (i128)((u128)a * (u128)b); }   // branch-free form
// 64 x 64 -> 128 signed product (operands known to fit 63 bits from the columns' statistics)
__device__ __forceinline__ i128 qh_mul_i64_i128(i64 a, i64 b) { return qh_mk128((u64)a * (u64)b, __mul64hi(a, b)); }
__device__ __forceinline__ i128 qh_pow10(int e) { i128 r = 1; for (int k = 0; k < e; ++k) r *= 10; return r; }

// ------------------------------------------------------------------ wavefront primitives (wave64)
__device__ __forceinline__ u64 qh_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ u32 qh_readlane32(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }
// Raise the status flags in `err` (bit b -> status word b). One atomic per wavefront and flag, and none when the flag is
// already up: thousands of waves OR-ing the same word serialise (measured ~10 ns per atomic on one address).
__device__ __forceinline__ void qh_report(u32* status, u32 err) {
  if (!__builtin_amdgcn_readfirstlane((int)(qh_ballot(err != 0) != 0))) return;
  for (int b = 0; b < QS_WORDS; ++b) {
    const u64 m = qh_ballot((err >> b) & 1u);
    if (m && qh_lane() == __builtin_ctzll(m) && !__hip_atomic_load(&status[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&status[b], 1u);
  }
}
__device__ __forceinline__ u64 qh_readlane64(u64 v, int lane) {
  return ((u64)qh_readlane32((u32)(v >> 32), lane) << 32) | qh_readlane32((u32)v, lane);
}
__device__ __forceinline__ i128 qh_readlane128(i128 v, int lane) {
  u128 u = (u128)v;
  return (i128)(((u128)qh_readlane64((u64)(u >> 64), lane) << 64) | qh_readlane64((u64)u, lane));
}
// number of set bits of `m` below this lane (the lane's rank inside a ballot mask)
__device__ __forceinline__ int qh_rank(u64 m) {
  return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
}

// DPP lane moves: out-of-row / masked-off destinations read 0, so the moves compose to a sum reduction.
template <int CTRL, int ROWMASK> __device__ __forceinline__ u32 qh_dpp0(u32 v) {
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ i128 qh_dpp128(i128 x) {
  u128 u = (u128)x;
  u32 a0 = qh_dpp0<CTRL, ROWMASK>((u32)u), a1 = qh_dpp0<CTRL, ROWMASK>((u32)(u >> 32));
  u32 a2 = qh_dpp0<CTRL, ROWMASK>((u32)(u >> 64)), a3 = qh_dpp0<CTRL, ROWMASK>((u32)(u >> 96));
  return (i128)(((u128)a3 << 96) | ((u128)a2 << 64) | ((u128)a1 << 32) | (u128)a0);
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ u64 qh_dpp64(u64 x) {
  return ((u64)qh_dpp0<CTRL, ROWMASK>((u32)(x >> 32)) << 32) | qh_dpp0<CTRL, ROWMASK>((u32)x);
}
// Wrapping 128-bit sum over the 64 lanes; every lane must be active. Result is wave-uniform.
// row_shr:1/2/4/8 build an inclusive scan inside each 16-lane row, row_bcast:15 / row_bcast:31
// carry the row totals upward so that lane 63 holds the wave total.
__device__ __forceinline__ i128 qh_wave_sum_i128(i128 x) {
  x += qh_dpp128<0x111, 0xf>(x);
  x += qh_dpp128<0x112, 0xf>(x);
  x += qh_dpp128<0x114, 0xf>(x);
  x += qh_dpp128<0x118, 0xf>(x);
  x += qh_dpp128<0x142, 0xa>(x);
  x += qh_dpp128<0x143, 0xc>(x);
  return qh_readlane128(x, 63);
}
__device__ __forceinline__ u64 qh_wave_sum_u64(u64 x) {
  x += qh_dpp64<0x111, 0xf>(x);
  x += qh_dpp64<0x112, 0xf>(x);
  x += qh_dpp64<0x114, 0xf>(x);
  x += qh_dpp64<0x118, 0xf>(x);
  x += qh_dpp64<0x142, 0xa>(x);
  x += qh_dpp64<0x143, 0xc>(x);
  return qh_readlane64(x, 63);
}
// OR over the 64 lanes (every lane active; the DPP moves read 0 outside a row, the identity of OR). Wave-uniform result.
__device__ __forceinline__ u32 qh_wave_or_u32(u32 x) {
  x |= qh_dpp0<0x111, 0xf>(x);
  x |= qh_dpp0<0x112, 0xf>(x);
  x |= qh_dpp0<0x114, 0xf>(x);
  x |= qh_dpp0<0x118, 0xf>(x);
  x |= qh_dpp0<0x142, 0xa>(x);
  x |= qh_dpp0<0x143, 0xc>(x);
  return qh_readlane32(x, 63);
}
// inclusive prefix sum over the lanes of the wavefront
__device__ __forceinline__ u32 qh_wave_incl_scan_u32(u32 v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)v, d, 64); if (lane >= d) v += t; }
  return v;
}
__device__ __forceinline__ u64 qh_shfl_xor64(u64 v, int m) {
  u32 lo = (u32)__shfl_xor((int)(u32)v, m, 64), hi = (u32)__shfl_xor((int)(u32)(v >> 32), m, 64);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double qh_wave_sum_f64(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __longlong_as_double((i64)qh_shfl_xor64((u64)__double_as_longlong(x), m));
  return x;
}
__device__ __forceinline__ u64 qh_wave_max_u64(u64 x) { for (int m = 32; m >= 1; m >>= 1) { u64 y = qh_shfl_xor64(x, m); x = y > x ? y : x; } return x; }
__device__ __forceinline__ u128 qh_wave_max_u128(u128 x) {
  for (int m = 32; m >= 1; m >>= 1) {
    u128 y = ((u128)qh_shfl_xor64((u64)(x >> 64), m) << 64) | qh_shfl_xor64((u64)x, m);
    x = y > x ? y : x;
  }
  return x;
}

// ------------------------------------------------------------------ memory policies for the group table
// LDS level: workgroup scope, DS atomics. HBM level: agent scope; the 8 XCD L2s are not coherent with
// each other, so every access to the shared table is an 8-byte agent-scope atomic (sc1), never a plain
// load/store (MI355X_MICROARCH "Valid forms": 8-B agent atomics on both sides).
struct MemLds {
  static constexpr int SCOPE = __HIP_MEMORY_SCOPE_WORKGROUP;
};
struct MemHbm {
  static constexpr int SCOPE = __HIP_MEMORY_SCOPE_AGENT;
};
template <class M> __device__ __forceinline__ u64 qh_ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, M::SCOPE); }
template <class M> __device__ __forceinline__ void qh_st64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, M::SCOPE); }
template <class M> __device__ __forceinline__ u64 qh_fadd64(u64* p, u64 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, M::SCOPE); }
template <class M> __device__ __forceinline__ void qh_add64(u64* p, u64 v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, M::SCOPE); }
template <class M> __device__ __forceinline__ bool qh_cas64(u64* p, u64 expect, u64 desired) {
  return __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, M::SCOPE);
}
// Exact wrapping 128-bit accumulate from two 64-bit atomics: the low-half fetch-add returns the old
// value, so each adder sees precisely whether ITS add wrapped; the wraps are then added to the high half.
// Addition commutes, so the final (hi:lo) equals the sum mod 2^128 whatever the interleaving.
template <class M> __device__ __forceinline__ void qh_acc_add_i128(u64* cell, i128 v) {
  u128 u = (u128)v; u64 lo = (u64)u, hi = (u64)(u >> 64);
  if (lo) { u64 old = qh_fadd64<M>(cell, lo); hi += ((u64)(old + lo) < old) ? 1ULL : 0ULL; }
  if (hi) qh_add64<M>(cell + 1, hi);
}
template <class M> __device__ __forceinline__ void qh_acc_add_u64(u64* cell, u64 v) { if (v) qh_add64<M>(cell, v); }
template <class M> __device__ __forceinline__ void qh_acc_add_f64(u64* cell, double v) {
  (void)__hip_atomic_fetch_add((double*)cell, v, __ATOMIC_RELAXED, M::SCOPE);
}
// MIN and MAX cells both store an order-preserving unsigned image of the value (MIN stores its complement),
// so a zero-filled cell is the identity and one atomic max serves both (codegen.cpp ord64()).
template <class M> __device__ __forceinline__ void qh_acc_max_u64(u64* cell, u64 v) { (void)__hip_atomic_fetch_max(cell, v, __ATOMIC_RELAXED, M::SCOPE); }
// 128-bit max has no hardware atomic: the cell carries a third word used as a spin lock. All accesses are atomics of
// the table's scope (HBM level: sc1, L2-coherent).
template <class M> __device__ __forceinline__ void qh_acc_max_u128(u64* cell, u128 v) {
  // The lanes of this wavefront that reached here take the lock ONE AT A TIME (wave-uniform loop over the ballot): a
  // lane spinning on a lock held by another lane of its own wavefront would never let that lane release it (lockstep
  // execution), whereas a holder in another wavefront always makes progress.
  u64 todo = qh_ballot(true);
  const int lane = qh_lane();
  while (todo) {
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1;
    if (lane == l) {
      while (!qh_cas64<M>(cell + 2, 0ULL, 1ULL)) {}
      const u128 cur = ((u128)qh_ld64<M>(cell + 1) << 64) | (u128)qh_ld64<M>(cell);
      if (v > cur) { qh_st64<M>(cell, (u64)v); qh_st64<M>(cell + 1, (u64)(v >> 64)); }
      if (M::SCOPE == __HIP_MEMORY_SCOPE_AGENT) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
      qh_st64<M>(cell + 2, 0ULL);
    }
  }
}

// ------------------------------------------------------------------ open-addressing group table
// Slot = [state][key words W][cells]. state: 0 empty, 1 being written, 2 ready. The claim/publish
// protocol lets multi-word keys be inserted concurrently: the claimer writes the key words and only
// then publishes state=2; everybody else compares key words only on ready slots. A lane never spins
// inside an iteration, so lanes of one wave racing for one slot make progress.
enum { QH_EMPTY = 0, QH_BUSY = 1, QH_READY = 2 };

template <class M, int W>
__device__ __forceinline__ u64* qh_find_or_insert(u64* table, u32 nslots /*pow2*/, int slot_words, const u64* key, u64 h,
                                                   int max_probe, bool* inserted) {
  u32 s = (u32)h & (nslots - 1);
  int probes = 0;
  *inserted = false;
  while (probes < max_probe) {
    u64* slot = table + (size_t)s * slot_words;
    u64 st = qh_ld64<M>(slot);
    if (st == QH_READY) {
      bool eq = true;
#pragma unroll
      for (int w = 0; w < W; ++w) eq &= (qh_ld64<M>(slot + 1 + w) == key[w]);
      if (eq) return slot;
      s = (s + 1) & (nslots - 1); ++probes;
    } else if (st == QH_EMPTY) {
      if (qh_cas64<M>(slot, QH_EMPTY, QH_BUSY)) {
#pragma unroll
        for (int w = 0; w < W; ++w) qh_st64<M>(slot + 1 + w, key[w]);
        // key words must be visible before the slot reads as ready
        if (M::SCOPE == __HIP_MEMORY_SCOPE_AGENT) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
        qh_st64<M>(slot, QH_READY);
        *inserted = true;
        return slot;
      }
      // lost the claim: look at the same slot again
    }
    // QH_BUSY: the claimer is between claim and publish; retry the same slot
  }
  return nullptr;
}

// ------------------------------------------------------------------ fused filter + hash-aggregate kernel
// Replaces MemoryTable::scan's predicate + filter_record_batch (datasource/memory.rs:90-93) and
// HashAggregate::execute / GroupAccumulator::update (physical/plan/aggregate/hash.rs:45-87,138-170)
// in ONE pass over the referenced columns: no mask array, no compacted batch, no concat copy, no
// per-group take — each needed input byte is read once (SURVEY §8d algorithmic bytes).
//
// Policy P (generated):
//   W            key words per group (0 = NoGroupingAggregate, aggregate/no_grouping.rs:30-62)
//   R            rows per thread per tile (all loads of a tile are issued before any is consumed)
//   SLOT_WORDS   u64 words per slot: 1 state + W key + cells
//   struct Row   { bool pass; u64 key[W]; <per-argument value + validity> }
//   struct Part  per-cell partial aggregate of one (thread, key)
//   load(a, tb, o, raw)                 issue the loads of row tb + o into `raw` (branch-free)
//   eval(a, raw, row, err)              predicate / key words / aggregate arguments from raw; status bits into err
//   part_init(p) / part_add<ROWS>(p, row, m)  thread-local accumulate of rows with m == true (ROWS: count them too)
//   part_reduce<ROWS>(p)                wavefront reduction (all lanes active); part_set_rows(p, n) sets the row count
//   (every cell's identity is all-zero bits, so a zero-filled table needs no per-slot initialisation)
//   slot_update<M>(slot, p)             atomically merge a partial into a slot
//   slot_merge(gslot, lslot)            merge an LDS slot into the HBM table slot
//
// Structure: persistent grid (a few workgroups per CU), grid-stride over tiles of 256*R rows. Three levels:
//  (1) wave-resident hot keys: each wave keeps up to KC keys (wave-uniform, SGPRs) with LANE-PRIVATE accumulators
//      in VGPRs. A row whose key is cached costs a compare and an add under the EXEC mask; nothing crosses lanes
//      and no table is touched until the kernel ends (one DPP reduction + one table update per wave and key per
//      KERNEL). Keys are admitted first-come while the wave's rows keep showing duplicates (ballot/readlane
//      discovery); the cache is flushed and switched off if it serves < 1/4 of the rows. This is what makes
//      low-cardinality GROUP BYs (Q1: 4 groups) and skewed keys stream at memory speed.
//  (2) the workgroup's LDS-staged open-addressing table, updated one lane per row with DS atomics (many
//      distinct keys => little contention);
//  (3) the HBM table for keys that do not fit the LDS table; at the end every workgroup merges its LDS table
//      into the HBM table.
#define QH_MAX_PEELS 8
#define QH_LDS_MAX_PROBE 8
#define QH_HBM_MAX_PROBE 128

extern __shared__ __attribute__((aligned(16))) u8 qh_dyn_lds[];

struct AggLaunch {
  u64* gtable;      // HBM table: `replicas` tables of g_nslots * SLOT_WORDS words each, zero-initialised
  u32 g_nslots;     // slots per replica, power of two
  u32 l_nslots;     // LDS slots per workgroup (power of two, 0 = no LDS level)
  u32* status;      // QS_WORDS words
  u32 replicas;     // workgroup b uses replica b % replicas (few groups: spreads the end-of-kernel merge of ~1000
                    // workgroups over many cache lines instead of one slot per group; the host merges replicas)
  u32 collect_stats;  // != 0: every workgroup adds its count of occupied LDS slots to status[QS_LDS_USED]
  // PARTS form (qh_filter_agg_body<.., PARTS = true>: the input was split by key hash, one part per workgroup — the groups of
  // two workgroups are disjoint): workgroup p takes rows [part_runs[p * part_stride], part_runs[(p + 1) * part_stride]) and
  // APPENDS its LDS table's groups to dense_out (one atomic per workgroup on dense_counter) instead of merging them into the
  // HBM table with a find-or-insert and the cells' atomics per group
  const u32* part_runs;
  u64* dense_out;
  u32* dense_counter;
  u32 part_stride, dense_cap;
  u32 n_parts, part_max;   // parts (<= 255); rows a workgroup takes of one part before the part is sliced
};

template <class P, class M>
__device__ __forceinline__ void qh_apply(u64* table, u32 nslots, int max_probe, const u64* key, const typename P::Part& part,
                                         u64*& slot_out) {
  bool inserted;
  u64 h = 0;
#pragma unroll
  for (int w = 0; w < P::W; ++w) h = qh_mix64(h ^ key[w]);
  u64* slot = qh_find_or_insert<M, P::W>(table, nslots, P::SLOT_WORDS, key, h, max_probe, &inserted);
  slot_out = slot;
  if (slot) P::template slot_update<M>(slot, part);
}

template <class P>
__device__ __forceinline__ void qh_update_group(u64* ltable, const AggLaunch& L, const u64* key, const typename P::Part& part, u32& err) {
  u64* slot = nullptr;
  if (L.l_nslots) qh_apply<P, MemLds>(ltable, L.l_nslots, QH_LDS_MAX_PROBE, key, part, slot);
  if (!slot) {
    qh_apply<P, MemHbm>(L.gtable, L.g_nslots, QH_HBM_MAX_PROBE, key, part, slot);
    if (!slot) atomicOr(&L.status[QS_OVERFLOW], 1u);            // prompt: the other workgroups stop streaming on it
    else if (L.l_nslots) err |= 1u << QS_LDS_SPILL;             // informational: reported once, at the end of the kernel
  }
}

// Every ready slot of the workgroup's LDS table is merged into the HBM table (one find-or-insert + the cells' atomics per
// slot); returns this thread's count of merged slots. Callers synchronise the workgroup before.
template <class P, int TB = QH_BLOCK>
__device__ __forceinline__ u32 qh_merge_lds_table(u64* ltable, const AggLaunch& L) {
  constexpr int W = P::W;
  u32 used = 0;
  for (u32 s = threadIdx.x; s < L.l_nslots; s += TB) {
    u64* ls = ltable + (size_t)s * P::SLOT_WORDS;
    if (ls[0] == QH_READY) {
      ++used;
      u64 key[W > 0 ? W : 1];
      u64 h = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) { key[w] = ls[1 + w]; h = qh_mix64(h ^ key[w]); }
      bool inserted;
      u64* gs = qh_find_or_insert<MemHbm, W>(L.gtable, L.g_nslots, P::SLOT_WORDS, key, h, QH_HBM_MAX_PROBE, &inserted);
      if (!gs) atomicOr(&L.status[QS_OVERFLOW], 1u);
      else P::slot_merge(gs, ls);
    }
  }
  return used;
}

// DEVROWS: the input is a join output whose row count lives on the device (KArgs::nrows_dev) — a separate instantiation, so
// that the kernels over ordinary tables keep a.nrows a plain kernel argument (two more live scalars cost the TPC-H Q1 kernel,
// which spills SGPRs, 8 % of its time)
// TB: threads per workgroup — 256 (QH_BLOCK), or 1024 for the one-workgroup-per-CU shape of mid-sized many-group inputs
// (qk_filter_agg_wide: sixteen wavefronts share ONE LDS table of up to 128 KB, so the chains of dependent loads and table
// updates of four times as many rows overlap while the number of end-of-kernel merges stays that of one table per CU)
// PARTS: see AggLaunch. The rows of a mid-sized many-group input are first ordered by key hash into as many parts as there
// are workgroups (the exchange's partition kernels over the row numbers, agg.cpp); what made that size slow was the END of the
// kernel — every workgroup merging its few thousand LDS groups into the shared HBM table, ~9 memory-side operations per
// (workgroup, group), a group living in as many workgroups as it has rows — and the rows that found their LDS table full.
// CONS (round 4, qk_filter_agg_cons): a lane owns RC CONSECUTIVE rows of a tile (row = tile + tid * RC + r) instead of rows
// r * TB + tid. Over a narrow layout (Q1's 22 bytes per row in 1- and 4-byte columns) the rows-r*TB+tid form issues one 4- or
// 1-byte load per lane, column and row — 7.5 load instructions per 64 rows, 2.8 KB in flight per wavefront — and the fraction of
// the HBM peak fell with the bytes per row (70 B: 0.78, 22 B: 0.63, 9 B: 0.50). Consecutive rows make a column's RC values ONE
// 16-byte (4-byte columns) or 4-byte (1-byte columns) load: a quarter of the load instructions, twice the bytes in flight out of
// the same registers. Plain tables of at least one tile only (the last, partial tile is shifted back to end at the last row and
// the rows it shares with the tile before are masked out, like qh_part_load's).
template <class P, bool DEVROWS = false, int TB = QH_BLOCK, bool PARTS = false, bool CONS = false>
__device__ __forceinline__ void qh_filter_agg_body(const KArgs& a, const AggLaunch& L0) {
  constexpr int W = P::W;
  constexpr int R = CONS ? P::RC : P::R;
  u64* ltable = (u64*)qh_dyn_lds;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  AggLaunch L = L0;
  L.gtable = L0.gtable + (size_t)(blockIdx.x % L0.replicas) * L0.g_nslots * P::SLOT_WORDS;

  if (W > 0) {
    // LDS table: zero = empty slots and identity cells
    const u32 lwords = L.l_nslots * (u32)P::SLOT_WORDS;
    for (u32 k = tid; k < lwords; k += TB) ltable[k] = 0;
    __syncthreads();
  }

  typename P::Acc acc;    // W == 0: whole-kernel per-thread accumulator
  if (W == 0) P::acc_init(acc);
  u32 err = 0;            // QS_* bits raised by this thread, reported once at the end
  // wave-resident hot-key cache (wave-uniform bookkeeping lives in SGPRs)
  constexpr int KC = P::KC;
  u64 ck[KC > 0 ? KC : 1][W > 0 ? W : 1];
  typename P::Acc cacc[KC > 0 ? KC : 1];   // (narrow SUM cells in 64 bits, widened at the end of the kernel)
  u64 crows[KC > 0 ? KC : 1];
  int nc = 0, singles = 0;
  bool cache_on = KC > 0, use_cache = true;
  u64 seen_pass = 0, seen_hits = 0;

  const i64 tile_rows = (i64)TB * R;
  // PARTS: workgroup p < n_parts takes part p — or, when the part holds more than part_max rows (a heavy key's part: hashing
  // sends every row of one key to one part), its first part_max rows; the rest of such a part is cut into slices of part_max
  // rows for the workgroups behind the first n_parts (at most rows / part_max of them in all). The slices of one part share
  // their groups, so THEY merge into the HBM table like the unpartitioned kernel; everybody else appends.
  i64 part_lo = 0, part_hi = 0;
  bool part_shared = false;
  if (PARTS) {
    const u32 np = L0.n_parts, pmax = L0.part_max;
    if (blockIdx.x < np) {
      part_lo = (i64)L0.part_runs[(size_t)blockIdx.x * L0.part_stride];
      part_hi = (i64)L0.part_runs[(size_t)(blockIdx.x + 1) * L0.part_stride];
      part_shared = part_hi - part_lo > (i64)pmax;
      if (part_shared) part_hi = part_lo + (i64)pmax;
    } else {
      // slice j of the oversized parts, in part order: every wavefront finds it with one scan over the parts' sizes
      const u32 j = blockIdx.x - np;
      u32 ext[4], first[4], lo4[4], n4[4];
      u32 mine = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 p = (u32)lane * 4u + (u32)k;
        lo4[k] = p < np ? L0.part_runs[(size_t)p * L0.part_stride] : 0u;
        n4[k] = p < np ? L0.part_runs[(size_t)(p + 1) * L0.part_stride] - lo4[k] : 0u;
        ext[k] = n4[k] ? (n4[k] - 1u) / pmax : 0u;
        first[k] = mine;
        mine += ext[k];
      }
      const u32 before = qh_wave_incl_scan_u32(mine, lane) - mine;
      u32 found_lo = 0, found_hi = 0;
      bool found = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 f = before + first[k];
        if (j >= f && j < f + ext[k]) {
          const u32 sl = j - f + 1u;
          found_lo = lo4[k] + sl * pmax;
          const u32 end = lo4[k] + n4[k];
          found_hi = found_lo + pmax < end ? found_lo + pmax : end;
          found = true;
        }
      }
      const u64 who = qh_ballot(found);
      if (!who) return;   // (workgroup-uniform: every wavefront scans the same sizes)
      const int src = __builtin_ctzll(who);
      part_lo = (i64)qh_readlane32(found_lo, src);
      part_hi = (i64)qh_readlane32(found_hi, src);
      part_shared = true;
    }
  }
  const i64 nrows_dr = PARTS ? part_hi : DEVROWS ? qh_rows(a) : 0;
#define QH_NROWS ((DEVROWS || PARTS) ? nrows_dr : a.nrows)
  const i64 ntiles = PARTS ? (part_hi - part_lo + tile_rows - 1) / tile_rows : (QH_NROWS + tile_rows - 1) / tile_rows;
  // phase timers (P::PROF, measurements only): cycles per wavefront in [0] loads + evaluation, [1] hot-key cache, [2] table
  // updates of a tile, [3] cached keys -> table at the end, [4] LDS table -> HBM table; summed into status words 8..12
  u64 prof[5] = {0, 0, 0, 0, 0};
  u64 tmark = P::PROF ? __builtin_readcyclecounter() : 0;
#define QH_PROF_MARK(K) if (P::PROF) { const u64 now_ = __builtin_readcyclecounter(); prof[K] += now_ - tmark; tmark = now_; }
  // the HBM table overflowed somewhere: the host will retry with a larger one, stop streaming (QH_OVERFLOWED: the load is
  // issued with a tile's loads and consumed at the end of the trip, wave-uniform)
#define QH_OVERFLOWED() (W > 0 ? __hip_atomic_load(&L.status[QS_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u)
  // phase 1 of a tile: issue the loads of all R rows (branch-free). Out-of-range lanes re-read the table's last row and are
  // masked out afterwards; addressing is (uniform 64-bit tile base) + (32-bit lane offset)
#define QH_ISSUE(RAW, TBASE, NROWS)                                         \
  _Pragma("unroll") for (int r = 0; r < R; ++r) {                           \
    const u32 o = (u32)r * TB + (u32)tid;                                   \
    const bool inb = (TBASE) + (i64)o < (NROWS);                            \
    P::load(a, (TBASE), inb ? o : (u32)((NROWS) - 1 - (TBASE)), RAW[r]);    \
  }
  if (CONS) {
    // the R rows of a lane are evaluated and accumulated SB at a time out of the raw registers (all R at once would hold R Row
    // structs: Q1's four rows 60 VGPRs)
    constexpr int SB = P::CSB < R ? P::CSB : R;
    // every load unconditional and at (shifted tile base) + tid * R + r: the compiler merges a column's R loads into one
#define QH_CONS_BASE(T) ((T) * tile_rows + tile_rows <= a.nrows ? (T) * tile_rows : a.nrows - tile_rows)
#define QH_CONS_ISSUE(RAW, TBS)                                             \
  _Pragma("unroll") for (int r = 0; r < R; ++r) P::load(a, (TBS), (u32)tid * (u32)R + (u32)r, RAW[r]);
    if ((i64)blockIdx.x < ntiles) {
      if (P::CPIPE == 0) {
        for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
          const u32 overflowed = QH_OVERFLOWED();
          const i64 tbn = t * tile_rows, tbs = QH_CONS_BASE(t);
          typename P::Raw raw[R];
          QH_CONS_ISSUE(raw, tbs)
#define QH_TILE_R SB
#define QH_TILE_RAW (raw + sb)
#define QH_TILE_INB(r) (tbs + (i64)((u32)tid * (u32)R + (u32)(sb + (r))) >= tbn)
          _Pragma("unroll") for (int sb = 0; sb < R; sb += SB) {
#include "qhip_agg_tile.inc"
            __builtin_amdgcn_sched_barrier(0);   // (keeps the sub-batches one after the other: their Row structs share registers)
          }
#undef QH_TILE_RAW
#undef QH_TILE_INB
#undef QH_TILE_R
          if (__builtin_amdgcn_readfirstlane((int)overflowed)) break;
        }
      } else {
        typename P::Raw rawA[R], rawB[R];
        i64 t = blockIdx.x;
        i64 tbsA = QH_CONS_BASE(t);
        QH_CONS_ISSUE(rawA, tbsA)
        for (;;) {
          u32 overflowed = QH_OVERFLOWED();
          const i64 tbnA = t * tile_rows;
          const i64 t1 = t + gridDim.x;
          const bool more1 = t1 < ntiles;
          const i64 tbnB = t1 * tile_rows;
          const i64 tbsB = more1 ? QH_CONS_BASE(t1) : 0;     // (behind the last tile: the table's first tile once more, never evaluated)
          QH_CONS_ISSUE(rawB, tbsB)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#define QH_TILE_R SB
#define QH_TILE_RAW (rawA + sb)
#define QH_TILE_INB(r) (tbsA + (i64)((u32)tid * (u32)R + (u32)(sb + (r))) >= tbnA)
          _Pragma("unroll") for (int sb = 0; sb < R; sb += SB) {
#include "qhip_agg_tile.inc"
            __builtin_amdgcn_sched_barrier(0);   // (keeps the sub-batches one after the other: their Row structs share registers)
          }
#undef QH_TILE_RAW
#undef QH_TILE_INB
#undef QH_TILE_R
          if (!more1 || __builtin_amdgcn_readfirstlane((int)overflowed)) break;
          overflowed = QH_OVERFLOWED();
          const i64 t2 = t1 + gridDim.x;
          const bool more2 = t2 < ntiles;
          tbsA = more2 ? QH_CONS_BASE(t2) : 0;
          QH_CONS_ISSUE(rawA, tbsA)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#define QH_TILE_R SB
#define QH_TILE_RAW (rawB + sb)
#define QH_TILE_INB(r) (tbsB + (i64)((u32)tid * (u32)R + (u32)(sb + (r))) >= tbnB)
          _Pragma("unroll") for (int sb = 0; sb < R; sb += SB) {
#include "qhip_agg_tile.inc"
            __builtin_amdgcn_sched_barrier(0);   // (keeps the sub-batches one after the other: their Row structs share registers)
          }
#undef QH_TILE_RAW
#undef QH_TILE_INB
#undef QH_TILE_R
          if (!more2 || __builtin_amdgcn_readfirstlane((int)overflowed)) break;
          t = t2;
        }
      }
    }
#undef QH_CONS_BASE
#undef QH_CONS_ISSUE
  } else if (P::PIPE == 0 || PARTS) {
    for (i64 t = PARTS ? 0 : blockIdx.x; t < ntiles; t += PARTS ? 1 : gridDim.x) {
      const u32 overflowed = QH_OVERFLOWED();
      const i64 tb = part_lo + t * tile_rows;
      typename P::Raw raw[R];
      QH_ISSUE(raw, tb, QH_NROWS)
#define QH_TILE_RAW raw
#define QH_TILE_INB(r) (tb + (i64)((u32)(r) * TB + (u32)tid) < QH_NROWS)
#define QH_TILE_R R
#include "qhip_agg_tile.inc"
#undef QH_TILE_R
#undef QH_TILE_RAW
#undef QH_TILE_INB
      if (__builtin_amdgcn_readfirstlane((int)overflowed)) break;
    }
  } else if ((i64)blockIdx.x < ntiles) {
    // Two register sets of raw column values: while a tile is evaluated, the loads of the workgroup's NEXT tile are in flight
    // (before: issue, wait, evaluate — a wavefront has nothing in flight while it computes). Every issue is unconditional —
    // behind the last tile it reads row 0 with every lane (one line per column) — and the loop is unrolled over the two sets, so
    // that the compiler waits with vmcnt(N > 0) for exactly the older set (qh_join_probe_dense_body, same discipline).
    typename P::Raw rawA[R], rawB[R];
    i64 t = blockIdx.x;
    QH_ISSUE(rawA, t * tile_rows, QH_NROWS)
    for (;;) {
      u32 overflowed = QH_OVERFLOWED();
      const i64 tbA = t * tile_rows;
      const i64 t1 = t + gridDim.x;
      const bool more1 = t1 < ntiles;
      const i64 tbB = more1 ? t1 * tile_rows : 0;
      QH_ISSUE(rawB, tbB, (more1 ? QH_NROWS : (i64)1))
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#define QH_TILE_RAW rawA
#define QH_TILE_INB(r) (tbA + (i64)((u32)(r) * TB + (u32)tid) < QH_NROWS)
#define QH_TILE_R R
#include "qhip_agg_tile.inc"
#undef QH_TILE_R
#undef QH_TILE_RAW
#undef QH_TILE_INB
      if (!more1 || __builtin_amdgcn_readfirstlane((int)overflowed)) break;
      overflowed = QH_OVERFLOWED();
      const i64 t2 = t1 + gridDim.x;
      const bool more2 = t2 < ntiles;
      const i64 tbA2 = more2 ? t2 * tile_rows : 0;
      QH_ISSUE(rawA, tbA2, (more2 ? QH_NROWS : (i64)1))
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#define QH_TILE_RAW rawB
#define QH_TILE_INB(r) (tbB + (i64)((u32)(r) * TB + (u32)tid) < QH_NROWS)
#define QH_TILE_R R
#include "qhip_agg_tile.inc"
#undef QH_TILE_R
#undef QH_TILE_RAW
#undef QH_TILE_INB
      if (!more2 || __builtin_amdgcn_readfirstlane((int)overflowed)) break;
      t = t2;
    }
  }
#undef QH_ISSUE
#undef QH_OVERFLOWED
#undef QH_NROWS

  // hand the cached groups of this wave to the workgroup's table: one reduction + one update per (wave, key) per KERNEL
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    if (k < nc) {
      typename P::Part part;
      P::acc_to_part(cacc[k], part);
      P::template part_reduce<false>(part);
      P::part_set_rows(part, crows[k]);
      if (lane == 0) qh_update_group<P>(ltable, L, ck[k], part, err);
    }
  }
  QH_PROF_MARK(3)
  qh_report(L.status, err);
  if (W == 0) {
    typename P::Part part;
    P::acc_to_part(acc, part);
    P::template part_reduce<true>(part);
    if (lane == 0) P::template slot_update<MemHbm>(L.gtable, part);
    return;
  }
  // ---- merge this workgroup's LDS table into the HBM table
  __syncthreads();
  if (PARTS && !part_shared) {
    // nobody else holds these groups: the ready slots go out as they are, behind ONE reservation per workgroup
    __shared__ u32 wsum[TB / 64 + 1];
    __shared__ u32 wg_base;
    u32 mine = 0;
    for (u32 s0 = tid; s0 < L.l_nslots; s0 += TB) mine += ltable[(size_t)s0 * P::SLOT_WORDS] == QH_READY ? 1u : 0u;
    const u32 incl = qh_wave_incl_scan_u32(mine, lane);
    if (lane == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
      u32 all = 0;
      for (int w = 0; w < TB / 64; ++w) { const u32 c = wsum[w]; wsum[w] = all; all += c; }
      wg_base = all ? atomicAdd(L.dense_counter, all) : 0u;
    }
    __syncthreads();
    u32 at = wg_base + wsum[tid >> 6] + (incl - mine);
    for (u32 s0 = tid; s0 < L.l_nslots; s0 += TB) {
      const u64* ls = ltable + (size_t)s0 * P::SLOT_WORDS;
      if (ls[0] == QH_READY) {
        if (at < L.dense_cap) {
          u64* o = L.dense_out + (size_t)at * P::SLOT_WORDS;
#pragma unroll
          for (int k = 0; k < P::SLOT_WORDS; ++k) o[k] = ls[k];
        }
        ++at;
      }
    }
    return;
  }
  u32 used = qh_merge_lds_table<P, TB>(ltable, L);
  QH_PROF_MARK(4)
  if (P::PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) atomicAdd(&L.status[8 + k], (u32)(prof[k] >> 8));
  }
#undef QH_PROF_MARK
  if (L.collect_stats && L.l_nslots) {
    // LDS-table occupancy (statistics runs only): one global atomic per workgroup
    __syncthreads();
    u32* total = (u32*)ltable;
    if (tid == 0) *total = 0;
    __syncthreads();
    used = (u32)qh_wave_sum_u64(used);
    if (lane == 0 && used) atomicAdd(total, used);
    __syncthreads();
    if (tid == 0 && *total) atomicAdd(&L.status[QS_LDS_USED], *total);
  }
}

// ------------------------------------------------------------------ partitioned aggregation (many groups on a big input)
// Groups that do not fit the workgroup's LDS table cost one random HBM line and two or three HBM atomics per ROW in the
// kernel above (~30 G operations/s: 50 M rows -> 1 M groups in 4.7 ms). Here the rows are first split by key hash into
// bins whose groups DO fit an LDS table — sequential traffic only — and every bin is then aggregated in LDS and merged
// into the HBM table once per GROUP:
//   pass 1  qh_agg_part_body<P, false>  per-workgroup histogram of the passing rows over the bins
//           (exclusive scan of hist[bin][workgroup], bin-major -> every (bin, workgroup) run's first record)
//   pass 2  qh_agg_part_body<P, true>   the same rows again (same static row range per workgroup): each becomes a
//                                       record [key words | its partial cells] (a slot minus the state word) in its run
//   pass 3  qh_agg_reduce_body<P>       work items = record ranges of one bin (big bins are cut): LDS table, then the
//                                       merge into the HBM table; the usual compaction / finalisation follows
struct PartLaunch {
  u32* hist;        // [n_bins][gridDim.x] counts (pass 1 out) / exclusive scan = first record of the run (pass 2 in)
  u64* records;     // pass 2 out: SLOT_WORDS - 1 words per record
  u32* status;
  u32 n_bins;       // power of two, <= 4096
  u32 rows_per_wg;  // static row range of a workgroup (multiple of QH_BLOCK)
};

__device__ __forceinline__ u32 qh_part_bin(u64 h, u32 n_bins) { return (u32)(h >> 40) & (n_bins - 1); }   // the HBM table uses the LOW bits

template <class P, bool SCATTER, bool DEVROWS = false>
__device__ __forceinline__ void qh_agg_part_body(const KArgs& a, const PartLaunch& L) {
  constexpr int W = P::W;
  u32* cnt = (u32*)qh_dyn_lds;              // [n_bins] counts (pass 1) / next free record of the bin's run (pass 2)
  const u32 tid = threadIdx.x;
  for (u32 b = tid; b < L.n_bins; b += QH_BLOCK) cnt[b] = SCATTER ? L.hist[(size_t)b * gridDim.x + blockIdx.x] : 0u;
  __syncthreads();
  const i64 first = (i64)blockIdx.x * L.rows_per_wg;
  const i64 nrows = DEVROWS ? qh_rows(a) : a.nrows;
  const i64 last = first + L.rows_per_wg < nrows ? first + L.rows_per_wg : nrows;
  u32 err = 0;
  // PR rows per thread and iteration, in phases like the fused kernel: all loads of the tile first (their latencies
  // overlap), then evaluation and the LDS rank, then the record stores
  constexpr int PR = 4;
  for (i64 tb = first; tb < last; tb += (i64)QH_BLOCK * PR) {
    typename P::Raw raw[PR];
    typename P::Row row[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const u32 o = (u32)r * QH_BLOCK + tid;
      P::load(a, tb, tb + (i64)o < last ? o : (u32)(last - 1 - tb), raw[r]);
    }
    u32 pos[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const bool inb = tb + (i64)((u32)r * QH_BLOCK + tid) < last;
      u32 e = 0;
      P::eval(a, raw[r], row[r], e);
      err |= inb ? e : 0u;
      row[r].pass = row[r].pass && inb;
      pos[r] = 0;
      if (row[r].pass) {
        u64 h = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) h = qh_mix64(h ^ row[r].key[w]);
        pos[r] = atomicAdd(&cnt[qh_part_bin(h, L.n_bins)], 1u);
      }
    }
    if (SCATTER) {
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        if (row[r].pass) {
          typename P::Part part;
          P::part_init(part);
          P::template part_add<true>(part, row[r], true);
          // a record is a slot without its state word: [key words | cells] (the slot-shaped view starts one word earlier)
          u64* rec = L.records + (size_t)pos[r] * (P::SLOT_WORDS - 1) - 1;
#pragma unroll
          for (int w = 0; w < W; ++w) rec[1 + w] = row[r].key[w];
          P::part_to_slot(rec, part);
        }
      }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (u32 b = tid; b < L.n_bins; b += QH_BLOCK) L.hist[(size_t)b * gridDim.x + blockIdx.x] = cnt[b];
    qh_report(L.status, err);   // (pass 2 sees the same rows: errors are reported once)
  }
}

// pass 2, staged (the default when a tile of records fits LDS): per-lane 8-byte stores of 24-byte records into 2 048 bins ran
// at 1.2 TB/s — every store instruction touches 64 different lines. Here a workgroup of 1 024 threads takes a TILE of
// 1 024 * PART_PR rows, ranks its records per bin in LDS (one returning DS atomic per record), scans the bin counts, lays the
// records out in LDS ORDERED BY BIN, and then writes them out word by word in that order: consecutive lanes store
// consecutive words of a bin's run (a tile holds 2-8 records per bin: 48-192 contiguous bytes), so a store instruction
// touches a few lines instead of 64. Same runs, same record order inside a run up to the tile-local rank, same histogram.
#define QH_STAGE_BLOCK 1024
template <class P, bool DEVROWS = false>
__device__ __forceinline__ void qh_agg_part_stage_body(const KArgs& a, const PartLaunch& L) {
  constexpr int W = P::W, TB = QH_STAGE_BLOCK, PR = P::PART_PR > 0 ? P::PART_PR : 1, RW = P::SLOT_WORDS - 1, TILE = TB * PR;
  const u32 nb = L.n_bins, tid = threadIdx.x;
  u32* cur = (u32*)qh_dyn_lds;                   // [nb] next record (global index) of the bin's run of this workgroup
  u32* tcnt = cur + nb;                          // [nb] records of the current tile per bin
  u32* tfirst = tcnt + nb;                       // [nb] first staged record of the bin inside the tile
  u64* stage = (u64*)(tfirst + nb) + 1;          // [TILE][RW] the tile's records, ordered by bin (+ 1: a record's slot-shaped view starts one word earlier)
  unsigned short* sbin = (unsigned short*)(stage + (size_t)TILE * RW);   // [TILE] bin of the staged record
  __shared__ u32 wsum[TB / 64];
  __shared__ u32 tile_total;
  for (u32 b = tid; b < nb; b += TB) { cur[b] = L.hist[(size_t)b * gridDim.x + blockIdx.x]; tcnt[b] = 0; }
  __syncthreads();
  const i64 first = (i64)blockIdx.x * L.rows_per_wg;
  const i64 nrows = DEVROWS ? qh_rows(a) : a.nrows;
  const i64 last = first + L.rows_per_wg < nrows ? first + L.rows_per_wg : nrows;
  const u32 per = nb > (u32)TB ? nb / (u32)TB : 1u;   // bins per thread in the scan (nb is a power of two)
  typename P::Raw raw[PR];
  if (first < last) {
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const u32 o = (u32)r * TB + tid;
      P::load(a, first, first + (i64)o < last ? o : (u32)(last - 1 - first), raw[r]);
    }
  }
  for (i64 tb = first; tb < last; tb += TILE) {
    typename P::Row row[PR];
    u32 bin[PR], rank[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const bool inb = tb + (i64)((u32)r * TB + tid) < last;
      u32 e = 0;
      P::eval(a, raw[r], row[r], e);   // (errors were reported by pass 1, which saw the same rows)
      row[r].pass = row[r].pass && inb;
      bin[r] = 0; rank[r] = 0;
      if (row[r].pass) {
        u64 h = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) h = qh_mix64(h ^ row[r].key[w]);
        bin[r] = qh_part_bin(h, nb);
        rank[r] = atomicAdd(&tcnt[bin[r]], 1u);
      }
    }
    // the next tile's column loads are issued now: they fly while this tile is scanned, staged and written out (one
    // workgroup per CU holds the LDS: nothing else would hide their latency)
    if (tb + TILE < last) {
      const i64 nt = tb + TILE;
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        const u32 o = (u32)r * TB + tid;
        P::load(a, nt, nt + (i64)o < last ? o : (u32)(last - 1 - nt), raw[r]);
      }
    }
    __syncthreads();
    // exclusive scan of the tile's bin counts (thread t owns `per` consecutive bins)
    {
      const u32 b0 = tid * per;
      u32 sum = 0;
      if (b0 < nb) for (u32 j = 0; j < per; ++j) sum += tcnt[b0 + j];
      u32 incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)incl, d, 64); if ((int)(tid & 63) >= d) incl += t; }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      u32 run = incl - sum;
      for (u32 w = 0; w < (tid >> 6); ++w) run += wsum[w];
      if (b0 < nb) for (u32 j = 0; j < per; ++j) { tfirst[b0 + j] = run; run += tcnt[b0 + j]; }
      if (tid == TB - 1) tile_total = run;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      if (row[r].pass) {
        const u32 pos = tfirst[bin[r]] + rank[r];
        typename P::Part part;
        P::part_init(part);
        P::template part_add<true>(part, row[r], true);
        u64* rec = stage + (size_t)pos * RW - 1;   // [key words | cells] = a slot without its state word
#pragma unroll
        for (int w = 0; w < W; ++w) rec[1 + w] = row[r].key[w];
        P::part_to_slot(rec, part);
        sbin[pos] = (unsigned short)bin[r];
      }
    }
    __syncthreads();
    // write the tile out in bin order: lane i stores word i of the ordered record stream
    const u32 nwords = tile_total * (u32)RW;
    for (u32 wd = tid; wd < nwords; wd += TB) {
      const u32 r = wd / (u32)RW, part = wd - r * (u32)RW, b = sbin[r];
      L.records[(size_t)(cur[b] + (r - tfirst[b])) * RW + part] = stage[wd];
    }
    __syncthreads();
    for (u32 b = tid; b < nb; b += TB) { cur[b] += tcnt[b]; tcnt[b] = 0; }
    __syncthreads();
  }
}

// ---- aggregation of an input whose equal keys are ADJACENT (round 4: qk_agg_runs). A join output in probe order over a probe
// side stored in key order — TPC-H Q3: lineitem is in l_orderkey order, so the three rows of an order arrive side by side — needs
// no hash table: a group is a RUN of rows. A thread evaluates four consecutive rows and the row in front of them; where a row's key
// differs from its predecessor's a run starts, and the thread that owns the start folds the run (on into the next threads' rows
// when it is longer: they skip rows that do not start a run) and writes the group as a dense slot [READY | key words | cells] —
// the format the compaction of the hashed path produces, so everything behind it is shared. Whether the input IS of that kind is
// checked here, not assumed: the key-word tuples must be non-decreasing (under the unsigned word order — any total order will do)
// from every row to the next, which makes equal keys adjacent; a violation, or a run longer than L.max_run (a few-group input:
// one thread would fold millions of rows), raises a flag and the host runs the hashed kernel instead and remembers (agg.cpp).
// No scan filter (a rejected row between two rows of one key would split its run), no NULL handling beyond the key's mask word.
struct RunsLaunch {
  u64* dense_out;     // cap slots
  u32* counter;       // += runs written
  u32* status;        // QS_WORDS words
  u32* flags;         // |= 1: keys not non-decreasing, |= 2: a run longer than max_run, |= 4: a Row too big for the LDS given
  u32 cap, max_run;
  u32 lds_bytes, pad_;   // dynamic LDS of the launch
};
// Every load of the kernel is issued in ONE batch: a wavefront's 256 rows (four per lane), the row in front of each lane's four,
// and one LOOK-AHEAD row per lane (the rows behind the wavefront's 256). The evaluated rows go to LDS; the lane that owns a run's
// start folds the run out of LDS — its own rows, the next lanes', the look-ahead rows — so that no load depends on a compare.
// (The first version followed a run with a load per row: 54 us for Q3's 0.3 M rows, every step a chain of index load -> gather
// over tables of GB — TLB misses in series.) Only a run that outlives the look-ahead is followed with loads.
template <class P, bool DEVROWS = false>
__device__ __forceinline__ void qh_agg_runs_body(const KArgs& a, const RunsLaunch& L) {
  constexpr int W = P::W > 0 ? P::W : 1, R = 4, WROWS = 64 * R;
  typedef typename P::Row Row;
  const i64 nrows = DEVROWS ? qh_rows(a) : a.nrows;
  const i64 tb = (i64)blockIdx.x * (QH_BLOCK * R);
  if (tb >= nrows) return;                       // (workgroup-uniform)
  const int lane = qh_lane();
  const int wv = (int)(threadIdx.x >> 6);
  // rows of a wavefront staged in LDS: its 256 + a look-ahead of up to 64, as many as the launch's LDS holds
  const u32 per_wave = (L.lds_bytes / (QH_BLOCK / 64)) / (u32)sizeof(Row);
  const u32 LA = per_wave >= (u32)WROWS + 64u ? 64u : per_wave > (u32)WROWS ? per_wave - (u32)WROWS : 0u;   // (uniform)
  if (per_wave < (u32)WROWS + 8u) { if (threadIdx.x == 0) atomicOr(L.flags, 4u); return; }
  Row* srow = (Row*)qh_dyn_lds + (size_t)wv * per_wave;
  const i64 wb = tb + (i64)wv * WROWS;           // the wavefront's first row
  const u32 o0 = (u32)wv * WROWS + (u32)lane * (u32)R;
  const i64 base = tb + (i64)o0;
  u32 err = 0, bad = 0;
  typename P::Raw raw[R], rawp, rawx;
  Row row[R], rowp, rowx;
  const bool have_prev = base > 0 && base < nrows;
  const i64 xrow = wb + WROWS + lane;            // this lane's look-ahead row
  const bool have_x = (u32)lane < LA && xrow < nrows;
  P::load(a, have_prev ? base - 1 : 0, 0u, rawp);
#pragma unroll
  for (int r = 0; r < R; ++r) P::load(a, tb, base + r < nrows ? o0 + (u32)r : (u32)(nrows - 1 - tb), raw[r]);
  P::load(a, tb, have_x ? (u32)(xrow - tb) : (u32)(nrows - 1 - tb), rawx);
  {
    u32 e = 0;
    P::eval(a, rawp, rowp, e);
    err |= have_prev ? e : 0u;
    e = 0;
    P::eval(a, rawx, rowx, e);     // (its errors are reported by the wavefront that owns the row)
  }
  bool start[R];
  u32 nstart = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool inb = base + r < nrows;
    u32 e = 0;
    P::eval(a, raw[r], row[r], e);
    err |= inb ? e : 0u;
    const u64* pk = r == 0 ? rowp.key : row[r - 1].key;
    const bool has = r == 0 ? have_prev : true;
    bool same = has, less = false, decided = false;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const u64 x = row[r].key[w], y = pk[w];
      same = same && x == y;
      if (!decided && x != y) {
        const bool sw = (P::KEY_SWAP_MASK >> w) & 1ULL, sg = (P::KEY_SIGN_MASK >> w) & 1ULL;   // (compile-time per word)
        const u64 xo = sw ? __builtin_bswap64(x) : sg ? x ^ (1ULL << 63) : x, yo = sw ? __builtin_bswap64(y) : sg ? y ^ (1ULL << 63) : y;
        less = xo < yo;
        decided = true;
      }
    }
    if (inb && has && less) bad |= 1u;
    start[r] = inb && !same;
    nstart += start[r] ? 1u : 0u;
    srow[(u32)lane * R + r] = row[r];
  }
  if ((u32)lane < LA) srow[WROWS + lane] = rowx;
  __syncthreads();                               // (the wavefront's rows are in LDS; a workgroup barrier keeps the code simple)
  // one reservation per wavefront for the runs its lanes start
  const u32 incl = qh_wave_incl_scan_u32(nstart, lane);
  const u32 wave_total = qh_readlane32(incl, 63);
  u32 wave_base = 0;
  if (wave_total) {
    if (lane == 0) wave_base = atomicAdd(L.counter, wave_total);
    wave_base = qh_readlane32(wave_base, 0);
  }
  u32 at = wave_base + incl - nstart;
  // (statically unrolled: rows in front of this thread's first start belong to a run of a thread before it)
  typename P::Part part;
  u64 fk[W];
  bool open = false;
  auto flush = [&]() {
    if (at < L.cap) {
      u64* o = L.dense_out + (size_t)at * P::SLOT_WORDS;
      o[0] = QH_READY;
#pragma unroll
      for (int w = 0; w < W; ++w) o[1 + w] = fk[w];
      P::part_to_slot(o, part);
    }
    ++at;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (start[r]) {
      if (open) flush();
      P::part_init(part);
#pragma unroll
      for (int w = 0; w < W; ++w) fk[w] = row[r].key[w];
      open = true;
      P::template part_add<true>(part, row[r], true);
    } else if (open && base + r < nrows) {
      P::template part_add<true>(part, row[r], true);
    }
  }
  if (open) {
    // the run may go on behind this lane's rows: the next lanes' rows and the look-ahead rows, out of LDS ...
    u32 k = (u32)lane * R + R, steps = 0;
    bool ended = false;
    const u32 staged = (u32)WROWS + LA;
    for (; k < staged && wb + (i64)k < nrows; ++k) {
      const Row& qx = srow[k];
      bool same = true;
#pragma unroll
      for (int w = 0; w < W; ++w) same = same && qx.key[w] == fk[w];
      if (!same) { ended = true; break; }
      P::template part_add<true>(part, qx, true);
      ++steps;
    }
    // ... and with a load per row behind those (a run of more than the look-ahead)
    if (!ended) {
      for (i64 x = wb + (i64)k; x < nrows; ++x) {
        typename P::Raw rx;
        Row qx;
        P::load(a, tb, (u32)(x - tb), rx);
        u32 e = 0;
        P::eval(a, rx, qx, e);
        bool same = true;
#pragma unroll
        for (int w = 0; w < W; ++w) same = same && qx.key[w] == fk[w];
        if (!same) break;
        err |= e;
        P::template part_add<true>(part, qx, true);
        if (++steps > L.max_run) { bad |= 2u; break; }
      }
    }
    flush();
  }
  qh_report(L.status, err);
  if (qh_ballot(bad != 0) != 0) {
    u32 all = bad;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) all |= (u32)__shfl_xor((int)all, d, 64);
    if (lane == 0) atomicOr(L.flags, all);
  }
}

struct ReduceLaunch {
  const u64* records;
  const u32* item_first;   // work item k = records [item_first[k], item_first[k + 1]), all of one bin — or nullptr:
  u32 n_items;
  // ... the items are derived ON THE DEVICE from the scanned histogram (mid-sized inputs: no host round trip between the
  // passes): item k = slice k % slices of bin k / slices; bin b = records [hist[b * g1], hist[(b + 1) * g1]) (the last bin
  // ends at hist[n_bins * g1], the record total); a bin is cut into `slices` slices of at least min_slice records
  u32 slices;
  const u32* hist;
  u32 g1, n_bins, min_slice, pad_;
};

// (TB: threads per workgroup — 256, or 1 024 with a 128 KB LDS table: one workgroup per CU still runs 16 wavefronts, and
// FOUR times the groups per bin mean a quarter of the bins, i.e. 4x longer runs per tile in pass 2)
template <class P, int TB = QH_BLOCK>
__device__ __forceinline__ void qh_agg_reduce_body(const ReduceLaunch& R, const AggLaunch& L) {
  constexpr int W = P::W;
  u64* ltable = (u64*)qh_dyn_lds;
  const u32 tid = threadIdx.x;
  const u32 lwords = L.l_nslots * (u32)P::SLOT_WORDS;
  u32 err = 0;
  for (u32 item = blockIdx.x; item < R.n_items; item += gridDim.x) {
    u32 r0, r1;
    if (R.item_first) { r0 = R.item_first[item]; r1 = R.item_first[item + 1]; }
    else {
      const u32 b = item / R.slices, sl = item - b * R.slices;
      const u32 b0 = R.hist[(size_t)b * R.g1], b1 = R.hist[(size_t)(b + 1) * R.g1];   // (bin-major: the next bin's first run)
      const u32 cnt = b1 - b0;
      u32 per = (cnt + R.slices - 1) / R.slices;
      per = per < R.min_slice ? R.min_slice : per;
      r0 = b0 + sl * per;
      r1 = r0 + per < b1 ? r0 + per : b1;
      if (sl * per >= cnt) continue;   // (workgroup-uniform: this slice of the bin is empty)
    }
    for (u32 k = tid; k < lwords; k += TB) ltable[k] = 0;
    __syncthreads();
    constexpr int RR = 4;   // records per thread and iteration: their loads are issued together
    for (u32 i0 = r0; i0 < r1; i0 += TB * RR) {
      u64 key[RR][W > 0 ? W : 1];
      typename P::Part part[RR];
      bool live[RR];
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const u32 i = i0 + (u32)r * TB + tid;
        live[r] = i < r1;
        const u64* rec = R.records + (size_t)(live[r] ? i : r1 - 1) * (P::SLOT_WORDS - 1) - 1;
#pragma unroll
        for (int w = 0; w < W; ++w) key[r][w] = rec[1 + w];
        P::part_from_slot(rec, part[r]);
      }
#pragma unroll
      for (int r = 0; r < RR; ++r)
        if (live[r]) qh_update_group<P>(ltable, L, key[r], part[r], err);   // LDS table; a bin with more groups than it holds spills to HBM
    }
    __syncthreads();
    (void)qh_merge_lds_table<P, TB>(ltable, L);
    __syncthreads();
  }
  qh_report(L.status, err);
}

// Hash of a join key: ONE 64-bit multiply and a fold per key word. The probe kernel hashes every probing row, and a 64-bit
// multiply is four quarter-rate 32-bit ones — the two multiplies of qh_mix64 were about a third of that kernel's VALU time.
// The high half of the product depends on every key bit (it picks the region / the legacy filter word); folding it into
// the low half does the same for the slot and filter-bit fields taken from there (a bare product's low bits would only
// depend on the key's low bits: TPC-H order keys use 8 of every 32 values).
__device__ __forceinline__ u64 qh_fold_mul(u64 x) { x *= 0x9E3779B97F4A7C15ULL; return x ^ (x >> 32); }
template <int W> __device__ __forceinline__ u64 qh_key_hash(const u64* k) {
  u64 h = qh_fold_mul(k[0]);
#pragma unroll
  for (int w = 1; w < W; ++w) h = qh_fold_mul(h ^ k[w]) + (u64)w;
  return h;
}

// ------------------------------------------------------------------ exchange, pass 1: fused scan filter + key -> part of every row
// The multi-GPU hash-join exchange (the reference is a single process and has no counterpart) splits a rank's
// slice of a join input by mix64(key words) into one part per rank. Round 3 did that as five generic steps (Filter operator,
// key words to a buffer, part ids, a stable radix sort of (part, row) pairs, per-part per-column gathers: 0.12 of the HBM
// peak); since round 4 it is two streaming passes:
//   pass 1  qh_part_ids_body (here, generated per plan): the scan filter and the key expressions are evaluated straight from
//           the table's columns; every row gets ONE BYTE — its part, or 0xFF when the filter rejects it — and every wavefront
//           counts its rows per part (ballots; the counts never leave the SGPRs until the wavefront is done);
//           an exclusive scan of hist[part][wavefront] (part-major) then gives every (part, wavefront) run its place;
//   pass 2  k_part_scatter (kernels_rel.hip, plan-independent): the same wavefront re-reads its row range — the part bytes and
//           ONLY the columns the plan above the exchange reads — ranks the rows of a tile per part (ballots again: stable, so a
//           part keeps the input's row order), orders the tile by part in LDS and stores it: consecutive lanes write
//           consecutive values of a part's run.
// A unit of work is a WAVEFRONT with a static, contiguous row range: no workgroup barrier anywhere, no atomics at all.
// Policy P (generated, KEYS_KERNEL_PARTITION): NP parts (compile time: the per-part counters are an unrolled SGPR array),
// W key words, Raw / load() as for the probe kernels, keys() returns bit 0 = every key column is non-null, bit 1 = the row
// passes the scan filter. A row with a NULL key hashes as all-zero key words (like k_partition_ids).
struct PartIdsLaunch {
  u8* ids;            // out: part of row i, 0xFF = rejected by the filter
  u32* hist;          // out: [NP][n_units] rows per (part, unit)
  u32* status;
  u32 n_units;        // units that own rows; unit u owns rows [u * rows_per_unit, (u + 1) * rows_per_unit)
  u32 rows_per_unit;  // a multiple of 4 tiles (a tile = 64 * PART_R rows)
  u32 wg_units;       // 0: a unit is a WAVEFRONT (pass 2's wavefront form needs every wavefront's runs); 1: a unit is a WORKGROUP
  u32 pad_;           //    whose four wavefronts take a quarter of its rows each (big inputs: a quarter of the counters to scan)
  // Partition by KEY RANGE instead of by hash (round 4, §7 "routing by key range"): NP - 1 ascending upper bounds of ONE integer
  // key — part p takes the keys in (bounds[p - 1], bounds[p]], the last part everything above. Both sides of a join and every rank
  // use the same bounds, so whatever they are the join stays correct; bounds taken from the ranks' own key ranges make the rows of
  // tables sliced in key order (TPC-H's orders and lineitem) stay where they are. nullptr: by hash.
  const i64* bounds;
};
template <int NP_>
__device__ __forceinline__ u32 qh_part_range(u64 key_word, bool valid, const i64* bounds) {
  const i64 k = valid ? (i64)key_word : (i64)0;     // (integer key words are sign-extended; a NULL key goes where 0 goes: it matches nothing)
  u32 p = 0;
#pragma unroll 1
  for (int b = 0; b < NP_ - 1; ++b) p += k > bounds[b] ? 1u : 0u;   // (wave-uniform loads)
  return p;
}
template <class P>
struct QhPartTile {
  typename P::Raw raw[P::PART_R];
  i64 tb;      // wave-uniform: first row the tile's loads read (WIDE: shifted back for a partial last tile)
  i64 nominal; // wave-uniform: first row of the tile (rows in front of it belong to the previous tile)
  bool live;   // wave-uniform
};
// WIDE: a lane owns R CONSECUTIVE rows of the tile (row = tile base + lane * R + r): the R loads of a column are adjacent and
// merge into 16-byte loads, and the lane's R part bytes are ONE 4-byte store (per-row byte stores wrote 64-byte half lines).
// Needs a table of at least one tile: a partial last tile is shifted back to end at the range's last row and the rows in
// front of its nominal start are masked. Pass 1 only COUNTS per part, so the row order inside a tile is free.
template <class P, bool WIDE>
__device__ __forceinline__ u32 qh_part_row_off(int r, int lane) { return WIDE ? (u32)lane * P::PART_R + (u32)r : (u32)r * 64u + (u32)lane; }
template <class P, bool WIDE>
__device__ __forceinline__ void qh_part_load(const KArgs& a, QhPartTile<P>& x, i64 first, i64 last, i64 j, int lane) {
  constexpr int R = P::PART_R, TILE = 64 * R;
  const i64 tb = first + j * TILE;
  x.live = tb < last;
  x.nominal = x.live ? tb : first;
  x.tb = WIDE ? (x.nominal + TILE <= last ? x.nominal : last - TILE) : x.nominal;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const u32 o = qh_part_row_off<P, WIDE>(r, lane);
    if (WIDE) P::load(a, x.tb, o, x.raw[r]);
    else P::load(a, x.tb, x.tb + (i64)o < last ? o : (u32)(last - 1 - x.tb), x.raw[r]);
  }
}
// THE partition function of the exchange: both join sides and every rank must agree on it (k_partition_ids computes the same
// from key-word arrays, oracle/ holds the numpy mirror the tests check against). One 64-bit multiply + fold per key word
// (qh_key_hash, the join's hash; the murmur finaliser of rounds 1-3 — two 64-bit multiplies per word, four quarter-rate
// 32-bit multiplies each — was 60 % of pass 1's time), the high half picks the part. A NULL key hashes as all-zero words.
template <int W> __device__ __forceinline__ u32 qh_part_hash(const u64* k, const bool valid, const u32 n_parts) {
  u64 z[W];
#pragma unroll
  for (int w = 0; w < W; ++w) z[w] = valid ? k[w] : 0ULL;
  const u64 h = qh_key_hash<W>(z);
  return (u32)(((h >> 32) * (u64)n_parts) >> 32);
}
template <class P>
__device__ __forceinline__ u32 qh_part_of(const u64* k, bool valid) { return qh_part_hash<P::W>(k, valid, (u32)P::NP); }
template <class P, bool DEVROWS = false, bool WIDE = false>
__device__ __forceinline__ void qh_part_ids_body(const KArgs& a, const PartIdsLaunch& L) {
  constexpr int R = P::PART_R, TILE = 64 * R, NP = P::NP;
  constexpr bool SMALL = NP <= 16;   // counters in SGPRs (unrolled); more parts: a per-wavefront LDS histogram
  __shared__ u32 lh[SMALL ? 1 : QH_BLOCK / 64][SMALL ? 1 : 256];
  __shared__ u32 wg_cnt[SMALL ? 16 : 256];
  const int lane = qh_lane();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool wgu = L.wg_units != 0;   // (uniform over the grid)
  const u32 unit = wgu ? blockIdx.x : blockIdx.x * (QH_BLOCK / 64) + (u32)wv;
  if (unit >= L.n_units) return;
  if (wgu) { for (int p = (int)threadIdx.x; p < (SMALL ? 16 : 256); p += QH_BLOCK) wg_cnt[p] = 0; __syncthreads(); }
  const i64 nrows = DEVROWS ? qh_rows(a) : a.nrows;
  const i64 quarter = (i64)(L.rows_per_unit / (QH_BLOCK / 64));
  const i64 ufirst = (i64)unit * L.rows_per_unit;
  const i64 first = wgu ? ufirst + (i64)wv * quarter : ufirst;
  const i64 uend = wgu ? first + quarter : ufirst + L.rows_per_unit;
  const i64 last = uend < nrows ? uend : nrows;
  u32 cnt[SMALL ? NP : 1];
#pragma unroll
  for (int p = 0; p < (SMALL ? NP : 1); ++p) cnt[p] = 0;
  if (!SMALL) { for (int p = lane; p < 256; p += 64) lh[wv][p] = 0; }
  u32 err = 0;
  if (first < last) {
    const i64 mine = (last - first + TILE - 1) / TILE;
    QhPartTile<P> A, B;
    qh_part_load<P, WIDE>(a, A, first, last, 0, lane);
    // one trip: the next tile's column loads are issued, then the current tile is evaluated out of registers (two register
    // sets, the loop unrolled over them; the scheduling barrier keeps the evaluation behind the issue, qh_pred_mask_body)
#define QH_PART_TRIP(X, Y, J)                                                          \
    qh_part_load<P, WIDE>(a, Y, first, last, (J) + 1, lane);                           \
    asm volatile("" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if (X.live) {                                                                      \
      u32 idr[R];                                                                      \
      _Pragma("unroll") for (int r = 0; r < R; ++r) {                                  \
        const i64 row = X.tb + (i64)qh_part_row_off<P, WIDE>(r, lane);                 \
        const bool inb = row < last && row >= X.nominal;                               \
        u64 k[P::W];                                                                   \
        u32 e = 0;                                                                     \
        const u32 code = P::keys(a, X.raw[r], k, e);                                   \
        err |= inb ? e : 0u;                                                           \
        const u32 id = (inb && (code & 2u)) ? (L.bounds ? qh_part_range<NP>(k[0], (code & 1u) != 0, L.bounds) : qh_part_of<P>(k, (code & 1u) != 0)) : 0xFFu;   \
        idr[r] = id;                                                                   \
        if (!WIDE && inb) L.ids[row] = (u8)id;                                         \
        if (SMALL) {                                                                   \
          _Pragma("unroll") for (int p = 0; p < (SMALL ? NP : 1); ++p) cnt[p] += (u32)__builtin_popcountll(qh_ballot(id == (u32)p)); \
        } else if (id != 0xFFu) atomicAdd(&lh[wv][id], 1u);                            \
      }                                                                                \
      if (WIDE) {                                                                      \
        const i64 row0 = X.tb + (i64)lane * R;                                         \
        if (R == 4 && X.tb == X.nominal && X.tb + TILE <= last) {   /* wave-uniform: a whole, aligned tile */ \
          *(u32*)(L.ids + row0) = idr[0] | (idr[1] << 8) | (idr[2] << 16) | (idr[3] << 24);  \
        } else {                                                                       \
          _Pragma("unroll") for (int r = 0; r < R; ++r) if (row0 + r < last && row0 + r >= X.nominal) L.ids[row0 + r] = (u8)idr[r]; \
        }                                                                              \
      }                                                                                \
    }
    for (i64 j = 0; j < mine; j += 2) {   // wave-uniform
      QH_PART_TRIP(A, B, j)
      QH_PART_TRIP(B, A, j + 1)
    }
#undef QH_PART_TRIP
  }
  if (SMALL) {
    u32 mine_cnt = 0;
#pragma unroll
    for (int p = 0; p < (SMALL ? NP : 1); ++p) mine_cnt = lane == p ? cnt[p] : mine_cnt;
    if (wgu) { if (lane < NP && mine_cnt) atomicAdd(&wg_cnt[lane], mine_cnt); }
    else if (lane < NP) L.hist[(size_t)lane * L.n_units + unit] = mine_cnt;
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int p = lane; p < NP; p += 64) {
      if (wgu) { if (lh[wv][p]) atomicAdd(&wg_cnt[p], lh[wv][p]); }
      else L.hist[(size_t)p * L.n_units + unit] = lh[wv][p];
    }
  }
  if (wgu) {
    __syncthreads();
    for (int p = (int)threadIdx.x; p < NP; p += QH_BLOCK) L.hist[(size_t)p * L.n_units + unit] = wg_cnt[p];
  }
  qh_report(L.status, err);
}

// ------------------------------------------------------------------ exchange, pass 2: rows -> per-part runs
// The wavefront that counted a row range in pass 1 reads it again: the part bytes and the columns that travel. A tile of
// 64 * R rows is ranked per part with ballots (stable: a part keeps the input's row order), ordered by part in the wavefront's
// OWN LDS area and written out so that consecutive lanes store consecutive values of a part's run. Every column is ONE buffer
// over all parts (part p = positions [runs[p * n_units], runs[(p + 1) * n_units]) of it): a part's column is a slice, never a
// copy. No workgroup barrier, no atomic; a wavefront's DS operations execute in order, so its LDS traffic needs no waits
// beyond the register dependencies (the empty asm statements only keep the COMPILER from reordering them).
// The kernel is generated per column shape (policy P: Vals = the registers of one tile's values, load() = all column loads of
// a tile, move() = LDS staging + stores, column by column) because everything about it must be static: the loads of tile
// t + 1 are issued before tile t is ranked and stored, and vector-memory operations return in order — only with a known
// number of loads and stores per trip can the compiler wait with s_waitcnt vmcnt(N > 0) for exactly the older tile's loads
// while this tile's stores are still in flight. (The first version, a plan-independent kernel with a run-time loop over
// column descriptors, drained the queue at every column: 3.1 TB/s on Q3's lineitem side.)
// NPT = 8 / 16: the parts' counters are unrolled SGPR arrays; NPT = 0: up to 255 parts, the distinct parts of a tile row are
// walked with readfirstlane + ballot and the counters live in LDS.
#define QH_PART_MAXC 8
struct PartScatterLaunch {
  const u8* ids;         // pass 1: part of every row, 0xFF = dropped
  const u32* runs;       // exclusive scan of pass 1's hist[part][unit] (part-major)
  i64 nrows;
  const u32* nrows_dev;  // the input's row count lives on the device (a join output of deferred size), nrows = capacity
  u32 n_units, rows_per_unit, n_parts;
  u32 sub;               // workgroup form: a workgroup owns the rows of `sub` consecutive pass-1 units (wavefront form: 1)
  void* trash;           // 1 KB nobody reads: where the (unconditional) stores of a tile without rows go
  const void* src[QH_PART_MAXC];   // the columns' values ...
  const u32* idx[QH_PART_MAXC];    // ... read through this index vector where the policy says so (a deferred gather never materialised)
  void* out[QH_PART_MAXC];
};
// ranks of a tile's rows inside their parts (q), the tile's first position per part (s_tf), rows in the tile (total) and the
// count this lane is responsible for when the running positions advance (cnt4: NPT > 0: part `lane`; NPT = 0: parts 4 lane ..)
// (NPT < 0: up to 255 parts, UNSTABLE — a row's rank inside its part is what a returning DS atomic on the part's counter hands
// out, 64 cycles per 64 rows where the stable walk over the distinct parts costs up to 64 trips; for consumers that do not care
// about the order inside a part: the aggregate's pre-partitioning, agg.cpp)
template <int NPT, int R>
__device__ __forceinline__ void qh_part_rank(const u32 (&id)[R], u32 (&q)[R], u32& total, u32 (&cnt4)[4], u32* s_tf, const u32 np, const int lane) {
  total = 0;
  cnt4[0] = cnt4[1] = cnt4[2] = cnt4[3] = 0;
  if (NPT < 0) {
    for (u32 p = (u32)lane; p < np; p += 64) s_tf[p] = 0;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r) q[r] = id[r] != 0xFFu ? atomicAdd(&s_tf[id[r]], 1u) : 0u;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; cnt4[k] = p < np ? s_tf[p] : 0u; }
    const u32 mine = cnt4[0] + cnt4[1] + cnt4[2] + cnt4[3];
    const u32 incl = qh_wave_incl_scan_u32(mine, lane);
    total = qh_readlane32(incl, 63);
    u32 at = incl - mine;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_tf[p] = at; at += cnt4[k]; }
  } else if (NPT > 0) {
    u32 run[NPT > 0 ? NPT : 1];
#pragma unroll
    for (int p = 0; p < NPT; ++p) run[p] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      q[r] = 0;
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const u64 m = qh_ballot(id[r] == (u32)p);
        q[r] = id[r] == (u32)p ? run[p] + (u32)qh_rank(m) : q[r];
        run[p] += (u32)__builtin_popcountll(m);
      }
    }
    u32 my_tf = 0;
#pragma unroll
    for (int p = 0; p < NPT; ++p) { my_tf = lane == p ? total : my_tf; cnt4[0] = lane == p ? run[p] : cnt4[0]; total += run[p]; }
    if (lane < NPT) s_tf[lane] = my_tf;
  } else {
    for (u32 p = (u32)lane; p < np; p += 64) s_tf[p] = 0;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r) {
      q[r] = 0;
      u64 todo = qh_ballot(id[r] != 0xFFu);
      while (todo) {   // wave-uniform
        const int l = __builtin_ctzll(todo);
        const u32 p = qh_readlane32(id[r], l);
        const u64 m = qh_ballot(id[r] == p);
        const u32 base = s_tf[p];
        if (id[r] == p) q[r] = base + (u32)qh_rank(m);
        asm volatile("" ::: "memory");
        if (lane == l) s_tf[p] = base + (u32)__builtin_popcountll(m);
        asm volatile("" ::: "memory");
        todo &= ~m;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; cnt4[k] = p < np ? s_tf[p] : 0u; }
    const u32 mine = cnt4[0] + cnt4[1] + cnt4[2] + cnt4[3];
    const u32 incl = qh_wave_incl_scan_u32(mine, lane);
    total = qh_readlane32(incl, 63);
    u32 at = incl - mine;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_tf[p] = at; at += cnt4[k]; }
  }
  asm volatile("" ::: "memory");
}
template <class P>
struct QhScatterTile {
  u32 id[P::R];
  typename P::Vals v;
  i64 tb;      // wave-uniform
  bool live;   // wave-uniform
};
template <class P>
__device__ __forceinline__ void qh_scatter_load(const PartScatterLaunch& L, QhScatterTile<P>& x, const i64 first, const i64 last, const i64 j, const int lane) {
  constexpr int R = P::R, TILE = 64 * R;
  const i64 tb = first + j * TILE;
  x.live = tb < last;
  x.tb = x.live ? tb : first;
#pragma unroll
  for (int r = 0; r < R; ++r) { const i64 row = x.tb + r * 64 + lane; x.id[r] = (u32)L.ids[row < last ? row : last - 1]; }
  P::load(L, x.tb, last, lane, x.v);
}
template <class P, bool DEVROWS = false>
__device__ __forceinline__ void qh_part_scatter_body(const PartScatterLaunch& L) {
  constexpr int R = P::R, TILE = 64 * R, NW = QH_BLOCK / 64, NPT = P::NPT, NPL = NPT > 0 ? NPT : 256;
  __shared__ __attribute__((aligned(16))) u8 s_val[NW][TILE * P::MAXW];
  __shared__ u32 s_dst[NW][TILE];
  __shared__ u32 s_cur[NW][NPL];   // per part: position (in the column buffers) of this wavefront's next row of the part
  __shared__ u32 s_tf[NW][NPL];    // per part: first position of the part inside the ordered tile
  const int lane = qh_lane();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const u32 unit = blockIdx.x * NW + (u32)wv;
  if (unit >= L.n_units) return;
  i64 nrows = L.nrows;
  if (DEVROWS) { const i64 d = (i64)*L.nrows_dev; nrows = d < nrows ? d : nrows; }
  const i64 first = (i64)unit * L.rows_per_unit;
  const i64 last = first + L.rows_per_unit < nrows ? first + L.rows_per_unit : nrows;
  if (first >= last) return;
  const u32 np = L.n_parts;
  for (u32 p = (u32)lane; p < np; p += 64) s_cur[wv][p] = L.runs[(size_t)p * L.n_units + unit];
  const i64 mine = (last - first + TILE - 1) / TILE;
  QhScatterTile<P> A, B;
  qh_scatter_load<P>(L, A, first, last, 0, lane);
#define QH_SCATTER_TRIP(X, Y, J)                                                                          \
  qh_scatter_load<P>(L, Y, first, last, (J) + 1, lane);                                                    \
  asm volatile("" ::: "memory");                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                       \
  {   /* unconditional on every path (a trip behind the last tile moves nothing: every id is 0xFF): the number of   */  \
      /* memory operations per trip is then static, which is what lets the compiler wait with vmcnt(N > 0)         */  \
    u32 id[R], q[R], pos[R], cnt4[4], total;                                                               \
    _Pragma("unroll") for (int r = 0; r < R; ++r) id[r] = (X.live && X.tb + r * 64 + lane < last) ? X.id[r] : 0xFFu;  \
    qh_part_rank<NPT, R>(id, q, total, cnt4, s_tf[wv], np, lane);                                          \
    u32 dst[R];                                                                                            \
    _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                        \
      pos[r] = 0; dst[r] = 0;                                                                              \
      if (id[r] != 0xFFu) { pos[r] = s_tf[wv][id[r]] + q[r]; dst[r] = s_cur[wv][id[r]] + q[r]; if (!P::DIRECT) s_dst[wv][pos[r]] = dst[r]; } \
    }                                                                                                      \
    asm volatile("" ::: "memory");                                                                         \
    if (NPT > 0) { if (lane < NPT) s_cur[wv][lane] += cnt4[0]; }                                           \
    else { _Pragma("unroll") for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_cur[wv][p] += cnt4[k]; } } \
    asm volatile("" ::: "memory");                                                                         \
    P::move(L, X.v, X.tb, id, pos, dst, total, s_val[wv], s_dst[wv], lane);                                \
  }                                                                                                        \
  asm volatile("" ::: "memory");                                                                           \
  __builtin_amdgcn_sched_barrier(0);
  if (P::PIPE) {
    for (i64 j = 0; j < mine; j += 2) {   // wave-uniform
      QH_SCATTER_TRIP(A, B, j)
      QH_SCATTER_TRIP(B, A, j + 1)
    }
  } else {   // (measurements: one tile at a time, one register set)
    for (i64 j = 0; j < mine; ++j) {
      if (j) qh_scatter_load<P>(L, A, first, last, j, lane);
      u32 id[R], q[R], pos[R], dst[R], cnt4[4], total;
#pragma unroll
      for (int r = 0; r < R; ++r) id[r] = A.tb + r * 64 + lane < last ? A.id[r] : 0xFFu;
      qh_part_rank<NPT, R>(id, q, total, cnt4, s_tf[wv], np, lane);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        pos[r] = 0; dst[r] = 0;
        if (id[r] != 0xFFu) { pos[r] = s_tf[wv][id[r]] + q[r]; dst[r] = s_cur[wv][id[r]] + q[r]; if (!P::DIRECT) s_dst[wv][pos[r]] = dst[r]; }
      }
      asm volatile("" ::: "memory");
      if (NPT > 0) { if (lane < NPT) s_cur[wv][lane] += cnt4[0]; }
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_cur[wv][p] += cnt4[k]; }
      }
      asm volatile("" ::: "memory");
      P::move(L, A.v, A.tb, id, pos, dst, total, s_val[wv], s_dst[wv], lane);
      asm volatile("" ::: "memory");
    }
  }
#undef QH_SCATTER_TRIP
}

// Pass 2, WORKGROUP-cooperative form (the default for big inputs). What bounds the wavefront form above is not its instruction
// count or its loads (with the stores switched off it moves Q3's lineitem side at 5.8 TB/s) but the WRITES: every wavefront
// keeps parts x columns output streams open and feeds each a few hundred bytes per tile — ~100 k concurrent streams of
// ~270-byte bursts over the chip, which HBM serves at ~2 TB/s (fewer, fatter units were faster although they leave most of
// the chip idle: the measured evidence that the stream count, not the parallelism, is the limit). Here a unit is a WORKGROUP
// of P::TB threads that owns the row range of `sub` consecutive pass-1 units and moves it in tiles of TB * R rows: the
// wavefronts rank their rows with ballots as before, their per-part counts meet in LDS, ONE barrier later every wavefront
// knows where its rows go inside the tile and inside the parts' runs; then column by column the tile is laid out in LDS
// ordered by part and written out by all threads — bursts of kilobytes per part and column, parts x columns streams per
// WORKGROUP. Same policy P as the wavefront form (+ TB), same stable order, same runs.
template <class P, bool DEVROWS = false>
__device__ __forceinline__ void qh_part_scatter_wg_body(const PartScatterLaunch& L) {
  constexpr int R = P::R, TB = P::TB, NW = TB / 64, TILE = TB * R, NPT = P::NPT, NPL = NPT > 0 ? NPT : 256;
  __shared__ __attribute__((aligned(16))) u8 s_val[TILE * P::MAXW];
  __shared__ u32 s_dst[TILE];
  __shared__ u32 s_wcnt[NW][NPL];   // rows of the tile per (wavefront, part)
  __shared__ u32 s_pos[NW][NPL];    // first position inside the ordered tile of the wavefront's rows of the part
  __shared__ u32 s_out[NW][NPL];    // ... and inside the column buffers
  __shared__ u32 s_tf[NW][NPL];     // (scratch of qh_part_rank: per-wavefront first positions, unused here)
  __shared__ u32 s_cur[NPL];        // per part: position in the column buffers of the workgroup's next row of the part
  __shared__ u32 s_total;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const u32 unit = blockIdx.x;
  i64 nrows = L.nrows;
  if (DEVROWS) { const i64 d = (i64)*L.nrows_dev; nrows = d < nrows ? d : nrows; }
  const i64 rows_per_wg = (i64)L.rows_per_unit * L.sub;
  const i64 first = (i64)unit * rows_per_wg;
  const i64 last = first + rows_per_wg < nrows ? first + rows_per_wg : nrows;
  if (first >= last) return;   // (workgroup-uniform)
  const u32 np = L.n_parts;
  for (u32 p = (u32)tid; p < np; p += TB) s_cur[p] = L.runs[(size_t)p * L.n_units + (size_t)unit * L.sub];
  const i64 ntiles = (last - first + TILE - 1) / TILE;
  // the wavefront's rows of tile j: [first + j * TILE + wv * 64 * R, + 64 * R) — row order = (wavefront, r, lane)
  QhScatterTile<P> A, B;
  const i64 wfirst = first + (i64)wv * 64 * R;
  auto load = [&](QhScatterTile<P>& x, const i64 j) {
    const i64 tb = wfirst + j * TILE;
    x.live = j < ntiles;
    x.tb = x.live ? tb : wfirst;
#pragma unroll
    for (int r = 0; r < R; ++r) { const i64 row = x.tb + r * 64 + lane; x.id[r] = (u32)L.ids[row < last ? row : last - 1]; }
    P::load(L, x.tb, last, lane, x.v);
  };
  load(A, 0);
#define QH_SCATTER_WG_TRIP(X, Y, J)                                                                        \
  load(Y, (J) + 1);                                                                                        \
  asm volatile("" ::: "memory");                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                       \
  {                                                                                                        \
    u32 id[R], q[R], pos[R], cnt4[4], wtotal;                                                              \
    _Pragma("unroll") for (int r = 0; r < R; ++r) id[r] = (X.live && X.tb + r * 64 + lane < last) ? X.id[r] : 0xFFu;  \
    qh_part_rank<NPT, R>(id, q, wtotal, cnt4, s_tf[wv], np, lane);                                         \
    if (NPT > 0) { if (lane < NPT) s_wcnt[wv][lane] = cnt4[0]; }                                           \
    else { _Pragma("unroll") for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_wcnt[wv][p] = cnt4[k]; } } \
    __syncthreads();                                                                                       \
    /* every wavefront, for its own row of the tables: rows of earlier wavefronts per part, the tile's rows per part */ \
    {                                                                                                      \
      u32 before[4] = {0, 0, 0, 0}, all[4] = {0, 0, 0, 0};                                                 \
      _Pragma("unroll") for (int k = 0; k < (NPT > 0 ? 1 : 4); ++k) {                                      \
        const u32 p = NPT > 0 ? (u32)lane : (u32)lane * 4u + (u32)k;                                       \
        if (p < np && p < (u32)NPL) {                                                                      \
          for (int w2 = 0; w2 < NW; ++w2) { const u32 c = s_wcnt[w2][p]; all[k] += c; before[k] += w2 < wv ? c : 0u; } \
        }                                                                                                  \
      }                                                                                                    \
      const u32 mine = all[0] + all[1] + all[2] + all[3];                                                  \
      const u32 incl = qh_wave_incl_scan_u32(mine, lane);                                                  \
      u32 at = incl - mine;                                                                                \
      _Pragma("unroll") for (int k = 0; k < (NPT > 0 ? 1 : 4); ++k) {                                      \
        const u32 p = NPT > 0 ? (u32)lane : (u32)lane * 4u + (u32)k;                                       \
        if (p < np && p < (u32)NPL) { s_pos[wv][p] = at + before[k]; s_out[wv][p] = s_cur[p] + before[k]; } \
        at += all[k];                                                                                      \
      }                                                                                                    \
      if (tid == TB - 1) s_total = incl;   /* (lane 63 of the last wavefront: the tile's rows) */          \
      asm volatile("" ::: "memory");                                                                       \
      _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                      \
        pos[r] = 0;                                                                                        \
        if (id[r] != 0xFFu) { pos[r] = s_pos[wv][id[r]] + q[r]; s_dst[pos[r]] = s_out[wv][id[r]] + q[r]; } \
      }                                                                                                    \
      P::move_wg(L, X.v, id, pos, s_val, s_dst, &s_total, tid);                                            \
      /* (move_wg ends behind a barrier: everybody has read s_cur / s_wcnt) */                             \
      if (wv == 0) {                                                                                       \
        _Pragma("unroll") for (int k = 0; k < (NPT > 0 ? 1 : 4); ++k) {                                    \
          const u32 p = NPT > 0 ? (u32)lane : (u32)lane * 4u + (u32)k;                                     \
          if (p < np && p < (u32)NPL) s_cur[p] += all[k];                                                  \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
  }                                                                                                        \
  asm volatile("" ::: "memory");                                                                           \
  __builtin_amdgcn_sched_barrier(0);
  for (i64 j = 0; j < ntiles; j += 2) {   // workgroup-uniform
    QH_SCATTER_WG_TRIP(A, B, j)
    QH_SCATTER_WG_TRIP(B, A, j + 1)
  }
#undef QH_SCATTER_WG_TRIP
}

// ------------------------------------------------------------------ predicate -> selection mask kernel
// Filter::execute (physical/plan/filter.rs:28-44): mask word j holds the keep bits of rows 64j..64j+63
// (wavefront ballot), wave_count[j] their popcount; the exclusive scan of wave_count gives every
// wavefront its output offset for the column compaction kernels (selection-vector compaction).
// Software-pipelined like the join probe (round 3): a wavefront owns a run of consecutive TILES of 64 * MASK_R rows; every trip
// evaluates tile t out of registers while the column loads of tiles t + 1 and t + 2 are in flight (three register sets that
// rotate, the loop unrolled three trips, every load unconditional — the trips behind the last tile re-read the run's first
// tile and store nothing). The one-trip-at-a-time loop before kept 1 KB per wavefront in flight — 8 MB over the chip, where
// 8 TB/s x ~2 us of loaded latency wants 16 MB: a 4-byte predicate column streamed at 3.2 TB/s (60 M rows: 78 us).
// The stores of a tile (lane r: mask word r and its popcount — two store instructions) are YOUNGER than the loads the
// next trip waits for, so the in-order return of vector-memory operations never makes a trip wait for a store.
template <class P>
struct QhMaskTile {
  typename P::Raw raw[P::MASK_R];
  i64 tile;    // wave-uniform
  bool live;   // wave-uniform: a tile of this wavefront's run (not a drain trip's)
};
template <class P>
__device__ __forceinline__ void qh_mask_load(const KArgs& a, QhMaskTile<P>& x, i64 first, i64 mine, i64 j, int lane) {
  constexpr int R = P::MASK_R, TILE = 64 * R;
  x.live = j < mine;
  x.tile = first + (x.live ? j : 0);
  const i64 tb = x.tile * TILE;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const u32 o = (u32)r * 64u + (u32)lane;
    P::load(a, tb, tb + (i64)o < a.nrows ? o : (u32)(a.nrows - 1 - tb), x.raw[r]);
  }
}
template <class P>
__device__ __forceinline__ void qh_mask_finish(const KArgs& a, const QhMaskTile<P>& x, u64* mask, u32* wave_count, i64 nwords, int lane, u32& err) {
  constexpr int R = P::MASK_R, TILE = 64 * R;
  u64 word = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool inb = x.tile * TILE + r * 64 + lane < a.nrows;
    u32 e = 0;
    const bool keep = P::pred(a, x.raw[r], e) && inb;
    err |= (inb && x.live) ? e : 0u;
    const u64 m = qh_ballot(keep);
    word = lane == r ? m : word;
  }
  const i64 w = x.tile * R + lane;
  if (x.live && lane < R && w < nwords) { mask[w] = word; wave_count[w] = (u32)__builtin_popcountll(word); }
}
template <class P>
__device__ __forceinline__ void qh_pred_mask_body(const KArgs& a, u64* mask, u32* wave_count, u32* status) {
  constexpr int R = P::MASK_R, TILE = 64 * R;
  const i64 nwords = (a.nrows + 63) / 64;
  const i64 ntiles = (a.nrows + TILE - 1) / TILE;
  const i64 wave = (i64)blockIdx.x * (QH_BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: the loop is wave-uniform)
  const i64 nwaves = (i64)gridDim.x * (QH_BLOCK / 64);
  const int lane = qh_lane();
  const i64 per = (ntiles + nwaves - 1) / nwaves;
  const i64 first = wave * per;
  if (first >= ntiles) return;
  const i64 mine = ntiles - first < per ? ntiles - first : per;
  u32 err = 0;
  QhMaskTile<P> A, B, C;
  qh_mask_load<P>(a, A, first, mine, 0, lane);
  qh_mask_load<P>(a, B, first, mine, 1, lane);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // (the scheduling barrier ends every trip: without it the compares of the NEXT trip's tile were hoisted above this trip's
  // loads — a wait for that tile with nothing else in flight; seen in the ISA)
#define QH_MASK_TRIP(X3, X1, J)                                    \
  qh_mask_finish<P>(a, X3, mask, wave_count, nwords, lane, err);   \
  qh_mask_load<P>(a, X1, first, mine, (J), lane);                  \
  asm volatile("" ::: "memory");                                   \
  __builtin_amdgcn_sched_barrier(0);
  for (i64 j = 0; j < mine; j += 3) {   // wave-uniform
    QH_MASK_TRIP(A, C, j + 2)
    QH_MASK_TRIP(B, A, j + 3)
    QH_MASK_TRIP(C, B, j + 4)
  }
#undef QH_MASK_TRIP
  qh_report(status, err);
}

// ------------------------------------------------------------------ hash join: fused scan-filter + key + probe + ordered emit
// read-only lookup in the distinct-key table of the build side (slot = [state][key words W]) after the build kernels
// have completed (plain cached loads)
// Slot addressing of the join table. Legacy layout: ONE open-addressing table of `mask + 1` slots (base = 0). Region layout
// (LDS-staged build, k_join_region_build): the table is cut into regions of 2^slot_bits slots, a key lives in region
// qh_region(h) and its probe sequence wraps inside that region, so a region is assembled in LDS by one workgroup and stored
// as whole lines. slot index = base + (s & mask).
__device__ __forceinline__ u32 qh_region(u64 h, u32 n_regions) { return (u32)(((h >> 32) * (u64)n_regions) >> 32); }
template <int W>
__device__ __forceinline__ u32 qh_join_find(const u64* table, u32 base, u32 mask, const u64* k, u32 s) {
  for (u32 probes = 0; probes <= mask; ++probes) {
    const u32 at = base + (s & mask);
    const u64* slot = table + (size_t)at * (1 + W);
    if (slot[0] < QH_READY) return 0xFFFFFFFFu;
    bool eq = true;
#pragma unroll
    for (int w = 0; w < W; ++w) eq &= slot[1 + w] == k[w];
    if (eq) return at;
    ++s;
  }
  return 0xFFFFFFFFu;
}
struct ProbeLaunch {
  const u64* table;      // distinct build keys
  const u64* bloom;      // hash filter of 64-bit words: qh_filter_mask() of the key all set in its word, else the key is not in
                         // the table (the filter stays in L2)
  const u32* count;      // build rows per slot (unused when start == nullptr)
  const u32* start;      // first position of a slot's rows in `rows`; nullptr: unique build keys (the slot's state word - 2 is the row)
  const u32* rows;       // build rows grouped by slot, ascending inside a slot
  u32* ent_slot;         // out, per CHUNK (the tiles_per_wave consecutive tiles of 64 * PROBE_R probe rows one wavefront owns): the
  u32* ent_row;          //      matching rows, compacted in row order from the chunk's first row position on: slot of the key
                         //      (unique build keys: the build row itself) and probe row. Rows without a match write nothing.
  u32* tile_nent;        // out, per chunk: number of matching probe rows
  u32* tile_total;       // out, per chunk: number of (build, probe) pairs
  u32* visited;          // build-row bitmap to mark here (LeftSemi / LeftAnti without a residual filter) or nullptr
  u32* status;
  u32 nslots, bloom_mask;  // legacy layout: slots of the one table (power of two), 64-bit words of the filter - 1
  u32 n_regions;           // region layout (0 = legacy): regions of 2^slot_bits slots, each with a filter slice of
  u32 slot_bits, bword_bits, stage_cap;   // 2^bword_bits 64-bit words; dense layout: LDS staging entries per wavefront
  u32 tiles_per_wave, lds_words;     // wavefront w of the grid owns tiles [w * tiles_per_wave, (w + 1) * tiles_per_wave); (dense
                                     // layout, hybrid: the first lds_words 32-bit words of the bitmap are staged in LDS)
  // dense (direct-address) layout, qh_join_probe_dense_body: `bloom` = an EXACT bitmap over the build keys' value range
  // [dense_min, dense_min + dense_n) as 32-bit words, `table` = u32 row_of[key - dense_min] (defined only where the bit is set)
  u64 dense_min;
  u32 dense_n, dense_words;          // keys in the range (<= 2^30); 32-bit words of the bitmap
};

// The build's hash filter is a blocked Bloom filter: FOUR bits per key inside ONE 64-bit word (two in each half), sized at
// 8 filter bits per table slot = 16 or more per key. A probe row pays one 8-byte L2 access; what the filter lets through
// by mistake (~0.3 % of the probing rows) costs a random 128-byte line of the table and — since such a key usually finds
// its home slot taken by another key — a dependent walk along the probe sequence, which stalls the software pipeline of
// qh_join_probe_body. (The first filter had two bits per key in a 32-bit word at ~10 bits per key: 3-5 % false positives,
// i.e. a walk in nearly every 256-row tile.)
__device__ __forceinline__ u64 qh_filter_mask(u32 x) {
  const u32 lo = (1u << (x & 31u)) | (1u << ((x >> 5) & 31u)), hi = (1u << ((x >> 10) & 31u)) | (1u << ((x >> 15) & 31u));
  return ((u64)hi << 32) | lo;
}
// legacy layout (one table): word from the high hash half, bits from the low one
__device__ __forceinline__ u32 qh_filter_word(u64 h, u32 word_mask) { return (u32)(h >> 32) & word_mask; }
// region layout: the region comes from the HIGH 32 hash bits (qh_region), everything inside a region from disjoint fields
// above each other from bit 0: slot = h[0, sb), filter word of the region's slice = h[sb, sb + wb), the mask = the next 20
__device__ __forceinline__ u32 qh_rfilter_word(u64 h, u32 sb, u32 wb) { return ((u32)h >> sb) & ((1u << wb) - 1u); }
__device__ __forceinline__ u64 qh_rfilter_mask(u64 h, u32 sb, u32 wb) { return qh_filter_mask((u32)(h >> (sb + wb))); }

// probe rows per thread and tile: P::PROBE_R (4; the key policy carries it so that the shape can be varied per plan)
// Probe pass 1 (hash_join.rs:218-275 for every probe batch at once): evaluate the fused scan filter and the key words
// straight from the probe table's columns, look the key up, keep (slot, probe row) of the matching rows only — compacted
// per 256-row tile with ballot/popcount ranks — and the pair count per tile. Pass 2 (k_join_emit) turns the entries into
// ordered (build row, probe row) pairs once the tile totals have been scanned.
//
// A lookup is a chain of dependent reads — columns -> filter word -> home slot -> (rarely) the rest of the probe sequence —
// and a wavefront that walks the chain tile by tile spends its life waiting (measured: 56 % of the wave cycles in
// s_waitcnt, HBM and L2 both far from saturated). So the chain is SOFTWARE-PIPELINED over the wavefront's tiles: every
// trip of the loop finishes tile t (stage 4: compare the slot, write the entries), tests the filter words of tile t + 1 and
// issues its slot loads (stage 3), hashes the keys of tile t + 2 and issues its filter loads (stage 2), and issues the
// column loads of tile t + 3 (stage 1). Loads return in order, so each stage waits only for what was issued a whole trip
// earlier, and three tiles' worth of loads are in flight per wavefront all the time. Every stage is a branch-free pass
// over the thread's R rows. Policy P: struct Raw, load(a, tile base, lane offset, raw) issues a row's column loads,
// keys(a, raw, k, err) computes filter + key words from them.
// The tiles in flight keep their state in three register sets that rotate through the stages (the loop is unrolled three
// trips deep): a tile's key words stay where stage 2 put them until stage 4 has used them, and a tile's column loads land in
// the set whose tile has just finished — no register copies between the stages. (Written as one loop with loop-carried
// stage variables the compiler places the copies of the freshly loaded registers at the loop head, behind an
// s_waitcnt vmcnt(0): the pipeline drains every trip.)
template <class P>
struct QhProbeTile {
  typename P::Raw raw[P::PROBE_R];                       // stage 1 -> 2
  u64 k[P::PROBE_R][P::W];                               // stage 2 -> 4
  u64 fw[P::PROBE_R], fm[P::PROBE_R];                    // stage 2 -> 3: filter word, the key's mask
  u32 at[P::PROBE_R];                                    // stage 2 -> 4: the key's home slot (base + slot inside the region)
  u64 st[P::PROBE_R], kw[P::PROBE_R][P::W];              // stage 3 -> 4
  bool ok[P::PROBE_R];
  i64 tile;                                              // wave-uniform
  bool live;                                             // wave-uniform: a real tile of this wavefront (not a drain trip's)
};
struct QhProbeCtx { bool regions; u32 smask; int lane; i64 first, mine; u32 nent, total; };   // (nent / total: the chunk so far, wave-uniform)

// stage 1: issue the column loads of the wavefront's j-th tile (j beyond its last tile: tile 0 once more, not live)
template <class P>
__device__ __forceinline__ void qh_probe_stage1(const KArgs& a, QhProbeTile<P>& x, const QhProbeCtx& c, i64 j) {
  constexpr int R = P::PROBE_R, TILE = 64 * R;
  x.live = j < c.mine;
  x.tile = x.live ? c.first + j : 0;   // (a drain trip reads tile 0: the same L2-resident lines for every wavefront)
  const i64 tb = x.tile * TILE;
  // row r * 64 + lane: the lanes of one load / lookup instruction hold 64 CONSECUTIVE rows. Fact tables are usually stored
  // in foreign-key order (lineitem by order key), so the filter and table lookups of an instruction fall into few cache
  // lines. (Measured alternative: a lane owning 4 consecutive rows reads each column with 16-byte loads, but spreads an
  // instruction's lookups over 4x as many keys — Q3's lineitem probe 217 -> 312 us; the kernel is VALU-bound, not
  // load-width-bound.)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const u32 o = (u32)r * 64 + (u32)c.lane;
    P::load(a, tb, tb + (i64)o < a.nrows ? o : (u32)(a.nrows - 1 - tb), x.raw[r]);
  }
}
// stage 2: filter + key words from the columns, hash, issue the filter-word loads
// (REGIONS — the table layout — is a template parameter: as a run-time flag both layouts' slot / filter arithmetic was
// evaluated for every row and one of them thrown away)
template <class P, bool REGIONS>
__device__ __forceinline__ void qh_probe_stage2(const KArgs& a, const ProbeLaunch& L, QhProbeTile<P>& x, const QhProbeCtx& c, u32& err) {
  constexpr int R = P::PROBE_R, TILE = 64 * R, W = P::W;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool inb = x.tile * TILE + r * 64 + c.lane < a.nrows;
    u32 e = 0;
    x.ok[r] = P::keys(a, x.raw[r], x.k[r], e) && inb;
    err |= (inb && x.live) ? e : 0u;
    const u64 h = qh_key_hash<W>(x.k[r]);
    const u32 reg = REGIONS ? qh_region(h, L.n_regions) : 0u;
    x.at[r] = (REGIONS ? reg << L.slot_bits : 0u) + ((u32)h & c.smask);
    x.fm[r] = REGIONS ? qh_rfilter_mask(h, L.slot_bits, L.bword_bits) : qh_filter_mask((u32)h);
    const u32 word = REGIONS ? (reg << L.bword_bits) + qh_rfilter_word(h, L.slot_bits, L.bword_bits) : qh_filter_word(h, L.bloom_mask);
    // (32-bit byte offset from the scalar base: the filter holds < 2^29 words)
    x.fw[r] = *(const u64*)((const char*)L.bloom + (size_t)((x.ok[r] ? word : 0u) << 3));
  }
}
// stage 3: filter test, issue the home-slot loads of the rows that pass
template <class P>
__device__ __forceinline__ void qh_probe_stage3(const ProbeLaunch& L, QhProbeTile<P>& x, const QhProbeCtx& c) {
  constexpr int R = P::PROBE_R, W = P::W;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    x.ok[r] = x.ok[r] && (x.fw[r] & x.fm[r]) == x.fm[r];
    const u64* slot = L.table + (size_t)(x.ok[r] ? x.at[r] : 0u) * (1 + W);
    if (W == 1) {   // [state | key]: one 16-byte load
      typedef u64 qh_v2u64 __attribute__((ext_vector_type(2)));
      const qh_v2u64 sk = *(const qh_v2u64*)slot;
      x.st[r] = sk.x;
      x.kw[r][0] = sk.y;
    } else {
      x.st[r] = slot[0];
#pragma unroll
      for (int w = 0; w < W; ++w) x.kw[r][w] = slot[1 + w];
    }
  }
}
// stage 4: compare the home slot, walk on after a collision (rare: the filter), write the tile's entries and counts
template <class P>
__device__ __forceinline__ void qh_probe_stage4(const ProbeLaunch& L, QhProbeTile<P>& x, QhProbeCtx& c) {
  constexpr int R = P::PROBE_R, TILE = 64 * R, W = P::W;
  const int lane = c.lane;
  u32 sid[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    sid[r] = 0xFFFFFFFFu;
    if (x.ok[r] && x.st[r] >= QH_READY) {
      bool eq = true;
#pragma unroll
      for (int w = 0; w < W; ++w) eq &= x.kw[r][w] == x.k[r][w];
      const u32 home = x.at[r] & c.smask, base = x.at[r] & ~c.smask;
      sid[r] = eq ? x.at[r] : qh_join_find<W>(L.table, base, c.smask, x.k[r], home + 1);   // collision: walk on from the next slot
      // unique build keys: the slot's state word carries its one build row, which is all pass 2 needs
      if (!L.start && sid[r] != 0xFFFFFFFFu) sid[r] = (u32)((eq ? x.st[r] : L.table[(size_t)sid[r] * (1 + W)]) - 2);
    }
  }
  if (!x.live) return;   // wave-uniform: a drain trip writes nothing
  u32 total = 0, nent = c.nent;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool found = sid[r] != 0xFFFFFFFFu;
    u32 cnt = 1u;
    if (L.start) cnt = L.count[found ? sid[r] : 0u];   // (wave-uniform branch: duplicated build keys only)
    total += found ? cnt : 0u;
    const u64 m = qh_ballot(found);
    if (found) {
      // the chunk's entries are appended tile after tile: nent is the chunk's running count (<= the rows seen so far)
      const size_t pos = (size_t)c.first * TILE + nent + (u32)__builtin_popcountll(m & ((1ULL << lane) - 1));
      L.ent_slot[pos] = sid[r];
      L.ent_row[pos] = (u32)(x.tile * TILE + r * 64 + lane);
    }
    nent += (u32)__builtin_popcountll(m);
  }
  c.nent = nent;
  if (L.visited) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (sid[r] != 0xFFFFFFFFu) {
        const u32 s0 = L.start ? L.start[sid[r]] : 0u, cnt = L.start ? L.count[sid[r]] : 1u;
        for (u32 q = 0; q < cnt; ++q) { const u32 b = L.start ? L.rows[s0 + q] : sid[r]; atomicOr(&L.visited[b >> 5], 1u << (b & 31)); }
      }
  }
  c.total += (u32)qh_wave_sum_u64(total);   // (published once per chunk, at the end of the kernel)
}

template <class P, bool REGIONS>
__device__ __forceinline__ void qh_join_probe_body(const KArgs& a, const ProbeLaunch& L) {
  constexpr int R = P::PROBE_R, TILE = 64 * R, NW = QH_BLOCK / 64;
  const i64 ntiles = (a.nrows + TILE - 1) / TILE;
  QhProbeCtx c;
  // where a key lives: legacy = one table (base 0, mask nslots - 1); region layout = its region's slot range
  c.regions = REGIONS;
  c.smask = REGIONS ? (1u << L.slot_bits) - 1u : L.nslots - 1u;
  c.lane = qh_lane();
  // the wavefront's chunk: tiles first .. first + mine - 1 (wave-uniform numbers, kept in SGPRs: the pipeline's control flow is
  // scalar). A chunk's matches form ONE run of entries and one (count, pair total) — pass 2 and the scan in front of it work
  // per chunk, not per tile (Q3's lineitem probe: 19.5 k chunks instead of 234 k tiles with 1.4 matches each)
  const i64 wave = (i64)blockIdx.x * NW + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  c.first = wave * (i64)L.tiles_per_wave;
  c.nent = 0; c.total = 0;
  if (c.first >= ntiles) {
    if (c.lane == 0) { L.tile_total[wave] = 0; L.tile_nent[wave] = 0; }
    return;
  }
  c.mine = ntiles - c.first < (i64)L.tiles_per_wave ? ntiles - c.first : (i64)L.tiles_per_wave;
  u32 err = 0;
  QhProbeTile<P> A, B, C;
  // fill the pipeline. EVERY stage call below and in the loop is unconditional and works on a real tile (the trips behind
  // the wavefront's last tile run on tile 0 without writing): the number of memory operations between a load and its use is
  // then the same on every path, which is what lets the compiler wait with s_waitcnt vmcnt(N > 0) for exactly the loads
  // issued a trip earlier. (With `if (tile valid)` around the stages it must assume the shortest path and drains.)
  qh_probe_stage1<P>(a, A, c, 0);
  qh_probe_stage2<P, REGIONS>(a, L, A, c, err);
  qh_probe_stage1<P>(a, B, c, 1);
  qh_probe_stage3<P>(L, A, c);
  qh_probe_stage2<P, REGIONS>(a, L, B, c, err);
  qh_probe_stage1<P>(a, C, c, 2);
  // one trip: finish X4's tile, filter-test X3's, hash X2's, then give X4 (free now) the wavefront's next tile
#define QH_PROBE_TRIP(X4, X3, X2, J)          \
  qh_probe_stage4<P>(L, X4, c);               \
  qh_probe_stage3<P>(L, X3, c);               \
  qh_probe_stage2<P, REGIONS>(a, L, X2, c, err);       \
  qh_probe_stage1<P>(a, X4, c, (J));
  for (i64 j = 3; j < c.mine + 3; j += 3) {   // wave-uniform; stage 4 has run for tiles 0 .. j - 1 after the trip group
    QH_PROBE_TRIP(A, B, C, j)
    QH_PROBE_TRIP(B, C, A, j + 1)
    QH_PROBE_TRIP(C, A, B, j + 2)
  }
#undef QH_PROBE_TRIP
  if (c.lane == 0) { L.tile_total[wave] = c.total; L.tile_nent[wave] = c.nent; }
  qh_report(L.status, err);
}

// ------------------------------------------------------------------ hash join over a DENSE integer key: direct addressing
// When the join key is ONE integer column whose values on the build side span a small range [min, max] (TPC-H's keys:
// c_custkey 1 .. 1.5 M, o_orderkey < 60 M at SF10 — known from the column's cached value range), hashing is pointless:
// the table is an exact bitmap over the range (one bit per possible key: 188 KB / 7.5 MB — L2 / Infinity-Cache resident,
// and read in key order by a probe side that is stored in foreign-key order) plus row_of[key - min] (u32), which is read
// only where the bit is set. A probing row costs a subtract, a range test, a shift and a bit test instead of a 64-bit
// multiply, a 4-bit blocked-filter mask and region arithmetic (~26 VALU); there are no false positives, no probe
// sequences, and a hit costs the 32-byte sector of row_of it needs, not a 128-byte line of a half-empty slot table.
// JoinHashMap's semantics (hash_join.rs:39-108, 177-216) are unchanged: unique build keys are assumed and checked (a
// second row with an equal key raises QS_MAXCOUNT and the join runs again with the hash / CSR layout, which yields the
// reference's ascending chains); NULL keys and rows rejected by a fused scan filter have no valid key and are never
// inserted or probed.
struct DenseBuildLaunch {
  u32* bits;        // zero-filled bitmap, 32-bit words (byte-map form: written afterwards by k_bytes_to_bits, not touched here)
  u32* row_of;      // uninitialised; row_of[key - kmin] = build row for every inserted key
  u32* status;
  u64 kmin;
  u32 n;            // keys in the range
  u32 gen;          // byte-map form: the stamp of this execution (1..255)
  u8* bytes;        // byte-map form (round 4), else null: bytes[key - kmin] = gen — a PLAIN store where the bitmap needs an atomic
  u32* counters;    // byte-map form: [0] += rows inserted (k_bytes_to_bits counts the stamped bytes: fewer = a duplicate key)
};
template <class P, bool DEVROWS = false>
__device__ __forceinline__ void qh_join_dense_build_body(const KArgs& a, const DenseBuildLaunch& L) {
  constexpr int R = 4;
  const i64 nrows = DEVROWS ? qh_rows(a) : a.nrows;
  const i64 tile = (i64)QH_BLOCK * R;
  u32 err = 0, dup = 0, out_of_range = 0, inserted = 0;
  const bool bytemap = L.bytes != nullptr;   // (uniform over the grid)
  for (i64 tb = (i64)blockIdx.x * tile; tb < nrows; tb += (i64)gridDim.x * tile) {
    u64 k[R][P::W];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const i64 i = tb + (i64)r * QH_BLOCK + threadIdx.x;
      const bool inb = i < nrows;
      u32 e = 0;
      ok[r] = P::keys(a, inb ? i : nrows - 1, k[r], e) && inb;
      err |= inb ? e : 0u;
    }
    // (Tried in round 4: the lanes of one bitmap word OR their bits together (DPP) and one lane issues the atomic. Q3's build
    // sides do not have the locality it needs — join 2's keys, a fifth of the orders in order-key order, touch ~40 words per 64
    // rows, join 1's are random — so the atomics stayed: 56 vs 52-62 us.)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u64 idx = k[r][0] - L.kmin;
      if (ok[r]) {
        if (idx < (u64)L.n) {
          if (bytemap) {
            // no atomic: a stamp byte per key value; two rows with one key stamp the same byte, which the packing kernel's count
            // of stamped bytes against the rows inserted here reveals (the memory side serves ~32 G scattered atomics per second —
            // join 2's 1.46 M bitmap atomics were its whole 52 us — but plain stores are absorbed by the L2s)
            L.bytes[(u32)idx] = (u8)L.gen;
            ++inserted;
          } else {
            const u32 bit = 1u << ((u32)idx & 31u);
            const u32 old = __hip_atomic_fetch_or(&L.bits[(u32)idx >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dup |= (old & bit) ? 1u : 0u;
          }
          L.row_of[(u32)idx] = (u32)(tb + (i64)r * QH_BLOCK + threadIdx.x);
        } else out_of_range = 1u;   // (cannot happen: the range comes from the column's own values; reported, never ignored)
      }
    }
  }
  qh_report(L.status, err | (out_of_range << QS_OVERFLOW));
  if (bytemap) {
    const u32 wave_rows = (u32)qh_wave_sum_u64((u64)inserted);
    if (qh_lane() == 0 && wave_rows) atomicAdd(&L.counters[0], wave_rows);
    return;
  }
  // duplicate build keys: status[QS_MAXCOUNT] = 2 (one atomic per wavefront that saw one, none when it is already up)
  if (__builtin_amdgcn_readfirstlane((int)(qh_ballot(dup != 0) != 0)) && qh_lane() == 0 &&
      __hip_atomic_load(&L.status[QS_MAXCOUNT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u)
    atomicMax(&L.status[QS_MAXCOUNT], 2u);
}

// Probe pass 1 over the dense layout — software-pipelined over the wavefront's tiles like qh_join_probe_body, but with
// THREE stages and nothing but column and bitmap loads in its steady state:
//   stage 1  issue the column loads of tile t + 2
//   stage 2  fused scan filter + key of tile t + 1, idx = key - min, range test, issue the bitmap-word loads
//   stage 3  bit test of tile t; the matching rows' (idx, probe row) go, ballot-ranked, into the wavefront's LDS staging area
// The staging area (L.stage_cap entries per wavefront) is flushed to the chunk's entry run when the next tile might not fit
// and at the end of the chunk: with the few matches of a selective join that is ONE burst of coalesced stores per chunk.
// What is NOT in the loop any more, and why (tools/micro/probe_like.hip, 60 M clustered rows, 12 B/row, MI355X): columns +
// predicate alone stream at 6.6-6.8 TB/s, with the dependent bitmap lookup 6.1 — and 4.9 once row_of[idx] is loaded per tile,
// 4.3 with the entry stores per tile, 3.7-4.0 with both, although only 5 % of the rows match: vector-memory operations return
// in order, so every trip's wait for its loads also waits for the acknowledgement of the stores and the return of the
// lookups issued in front of them, however few lanes they carry. So the build row is looked up by PASS 2 (k_join_emit reads
// row_of[idx] for the matching entries only — 0.3 M of Q3's 60 M lineitem rows), and stores leave the loop through LDS.
// WIDE: a lane owns R CONSECUTIVE rows of the tile (row = tile base + lane * R + r) instead of rows r * 64 + lane, so that
// the R loads of a column are adjacent and merge into wider loads. Needs a table of at least one tile: the last, partial
// tile is read as the table's LAST whole tile (wave-uniform base shift, its already-seen rows masked), so that no load is
// clamped per row — a per-row select on the index is what keeps the loads from merging.
// LDSBITS: 1 = every workgroup first copies the whole bitmap into LDS (1 024-thread workgroups so that the one workgroup a
// CU holds still runs 16 wavefronts) and stage 2 reads it with DS loads instead of going through L1 / L2; 2 = hybrid: the
// first L.lds_words words live in LDS, keys beyond them are looked up in L2.
// LeftSemi / LeftAnti joins (L.visited set, no pairs wanted): the flush looks the build rows up itself and marks them.
template <class P>
struct QhDenseTile {
  typename P::Raw raw[P::PROBE_R];   // stage 1 -> 2
  u32 idx[P::PROBE_R];               // stage 2 -> 3
  u32 bw[P::PROBE_R];                // stage 2 -> 3: the key's bitmap word (from L2 / from LDS when LDSBITS == 1)
  u32 bl[P::PROBE_R];                // stage 2 -> 3: ... from LDS (LDSBITS == 2: which one counts is decided in stage 3)
  bool ok[P::PROBE_R];
  i64 tile;                          // wave-uniform
  i64 tb;                            // wave-uniform: first row the tile's loads read (WIDE: shifted back for the last tile)
  bool live;                         // wave-uniform: a real tile of this wavefront (not a drain trip's)
};
struct QhDenseStage { u32* idx; u32* row; u32 cap, nbuf, nflushed; };   // the wavefront's staging area (LDS) and its counters (wave-uniform)
template <class P, bool WIDE>
__device__ __forceinline__ u32 qh_dense_row_off(int r, int lane) { return WIDE ? (u32)lane * P::PROBE_R + (u32)r : (u32)r * 64 + (u32)lane; }
template <class P, bool WIDE>
__device__ __forceinline__ void qh_dense_stage1(const KArgs& a, QhDenseTile<P>& x, const QhProbeCtx& c, i64 j) {
  constexpr int R = P::PROBE_R, TILE = 64 * R;
  x.live = j < c.mine;
  x.tile = x.live ? c.first + j : 0;   // (a drain trip reads tile 0: the same L2-resident lines for every wavefront)
  const i64 tb = x.tile * TILE;
  x.tb = WIDE ? (tb + TILE <= a.nrows ? tb : a.nrows - TILE) : tb;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const u32 o = qh_dense_row_off<P, WIDE>(r, c.lane);
    if (WIDE) P::load(a, x.tb, o, x.raw[r]);
    else P::load(a, tb, tb + (i64)o < a.nrows ? o : (u32)(a.nrows - 1 - tb), x.raw[r]);
  }
}
template <class P, int LDSBITS, bool WIDE>
__device__ __forceinline__ void qh_dense_stage2(const KArgs& a, const ProbeLaunch& L, QhDenseTile<P>& x, const QhProbeCtx& c, u32& err) {
  constexpr int R = P::PROBE_R, TILE = 64 * R;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const i64 row = x.tb + qh_dense_row_off<P, WIDE>(r, c.lane);
    const bool inb = WIDE ? row >= x.tile * TILE : row < a.nrows;   // (WIDE: rows in front of the tile's nominal start were the previous tile's)
    u32 e = 0;
    u64 k[P::W];
    x.ok[r] = P::keys(a, x.raw[r], k, e) && inb;
    err |= (inb && x.live) ? e : 0u;
    const u64 idx = k[0] - L.dense_min;
    x.ok[r] = x.ok[r] && idx < (u64)L.dense_n;
    x.idx[r] = x.ok[r] ? (u32)idx : 0u;
    const u32 w = x.idx[r] >> 5;
    // (32-bit byte offset from the scalar base: the bitmap holds <= 2^25 words)
    if (LDSBITS == 1) x.bw[r] = ((const u32*)qh_dyn_lds)[w];
    else if (LDSBITS == 2) {
      const bool in_lds = w < L.lds_words;
      x.bl[r] = ((const u32*)qh_dyn_lds)[in_lds ? w : 0u];
      x.bw[r] = *(const u32*)((const char*)L.bloom + (size_t)((in_lds ? 0u : w) << 2));   // (word 0 for the LDS-served lanes: one line)
    } else x.bw[r] = *(const u32*)((const char*)L.bloom + (size_t)(w << 2));
  }
}
// the staged entries -> the chunk's run in ent_slot / ent_row (coalesced), behind what was flushed before
__device__ __forceinline__ void qh_dense_flush(const ProbeLaunch& L, QhDenseStage& st, u64 chunk_base, int lane) {
  for (u32 j = (u32)lane; j < st.nbuf; j += 64) {
    const u32 idx = st.idx[j];
    L.ent_slot[chunk_base + st.nflushed + j] = idx;
    L.ent_row[chunk_base + st.nflushed + j] = st.row[j];
    if (L.visited) { const u32 b = ((const u32*)L.table)[idx]; atomicOr(&L.visited[b >> 5], 1u << (b & 31)); }
  }
  st.nflushed += st.nbuf;
  st.nbuf = 0;
}
template <class P, int LDSBITS, bool WIDE>
__device__ __forceinline__ void qh_dense_stage3(const ProbeLaunch& L, QhDenseTile<P>& x, QhProbeCtx& c, QhDenseStage& st) {
  constexpr int R = P::PROBE_R, TILE = 64 * R;
  if (!x.live) return;   // wave-uniform: a drain trip writes nothing
  const int lane = c.lane;
  const u64 below = (1ULL << lane) - 1;
  if (st.nbuf + (u32)TILE > st.cap) qh_dense_flush(L, st, (u64)c.first * TILE, lane);   // wave-uniform, rare
  bool hit[R];
  u64 m[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const u32 word = (LDSBITS == 2 && (x.idx[r] >> 5) < L.lds_words) ? x.bl[r] : x.bw[r];
    hit[r] = x.ok[r] && ((word >> (x.idx[r] & 31u)) & 1u);
    m[r] = qh_ballot(hit[r]);
  }
  if (WIDE) {
    // a lane's rows are consecutive: its entries go, in row order, behind the entries of all lower lanes
    u32 before = 0, total = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) { before += (u32)__builtin_popcountll(m[r] & below); total += (u32)__builtin_popcountll(m[r]); }
    u32 pos = st.nbuf + before;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (hit[r]) { st.idx[pos] = x.idx[r]; st.row[pos] = (u32)(x.tb + (i64)((u32)lane * R + (u32)r)); ++pos; }
    st.nbuf += total;
  } else {
    u32 n = st.nbuf;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (hit[r]) {
        const u32 pos = n + (u32)__builtin_popcountll(m[r] & below);
        st.idx[pos] = x.idx[r];
        st.row[pos] = (u32)(x.tile * TILE + r * 64 + lane);
      }
      n += (u32)__builtin_popcountll(m[r]);
    }
    st.nbuf = n;
  }
}
template <class P, int LDSBITS, bool WIDE>
__device__ __forceinline__ void qh_join_probe_dense_body(const KArgs& a, const ProbeLaunch& L) {
  constexpr int R = P::PROBE_R, TILE = 64 * R;
  const int NW = (int)(blockDim.x >> 6);
  const u32 nbits = LDSBITS == 1 ? L.dense_words : LDSBITS == 2 ? L.lds_words : 0u;
  if (LDSBITS) {
    u32* lbits = (u32*)qh_dyn_lds;
    for (u32 w = threadIdx.x; w < nbits; w += blockDim.x) lbits[w] = ((const u32*)L.bloom)[w];
    __syncthreads();
  }
  const i64 ntiles = (a.nrows + TILE - 1) / TILE;
  QhProbeCtx c;
  c.regions = false; c.smask = 0;
  c.lane = qh_lane();
  const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const i64 wave = (i64)blockIdx.x * NW + wave_in_wg;
  c.first = wave * (i64)L.tiles_per_wave;
  c.nent = 0; c.total = 0;
  if (c.first >= ntiles) {
    if (c.lane == 0) { L.tile_total[wave] = 0; L.tile_nent[wave] = 0; }
    return;
  }
  c.mine = ntiles - c.first < (i64)L.tiles_per_wave ? ntiles - c.first : (i64)L.tiles_per_wave;
  // the wavefront's staging area: [idx | row] x stage_cap, behind the bitmap words of the LDS variants
  QhDenseStage st;
  st.cap = L.stage_cap;
  st.idx = (u32*)qh_dyn_lds + ((nbits + 1u) & ~1u) + (u32)wave_in_wg * 2u * L.stage_cap;
  st.row = st.idx + L.stage_cap;
  st.nbuf = 0; st.nflushed = 0;
  u32 err = 0;
  QhDenseTile<P> A, B, C;
  // fill the pipeline. EVERY stage call below and in the loop is unconditional and works on a real tile (the trips behind the
  // wavefront's last tile run on tile 0 without writing), and a compiler barrier ends every trip: the loads stay where they
  // are issued (without it the machine-sink pass moved loads into the next trip's conditional blocks, right in front of
  // their use and behind an s_waitcnt vmcnt(0); seen in the ISA)
  qh_dense_stage1<P, WIDE>(a, A, c, 0);
  qh_dense_stage1<P, WIDE>(a, B, c, 1);
  asm volatile("" ::: "memory");
  qh_dense_stage2<P, LDSBITS, WIDE>(a, L, A, c, err);
  asm volatile("" ::: "memory");
  // one trip: finish X3's tile, key + bitmap loads of X2's, column loads of the tile two ahead into X1
#define QH_DENSE_TRIP(X3, X2, X1, J)                   \
  qh_dense_stage3<P, LDSBITS, WIDE>(L, X3, c, st);     \
  qh_dense_stage2<P, LDSBITS, WIDE>(a, L, X2, c, err); \
  qh_dense_stage1<P, WIDE>(a, X1, c, (J));             \
  asm volatile("" ::: "memory");
  for (i64 j = 0; j < c.mine; j += 3) {   // wave-uniform; stage 3 runs for tiles j, j + 1, j + 2 (not live beyond the last: no writes)
    QH_DENSE_TRIP(A, B, C, j + 2)
    QH_DENSE_TRIP(B, C, A, j + 3)
    QH_DENSE_TRIP(C, A, B, j + 4)
  }
#undef QH_DENSE_TRIP
  qh_dense_flush(L, st, (u64)c.first * TILE, c.lane);
  // unique build keys: every matching probe row is exactly one pair
  if (c.lane == 0) { L.tile_total[wave] = st.nflushed; L.tile_nent[wave] = st.nflushed; }
  qh_report(L.status, err);
}

// ------------------------------------------------------------------ hash join build, step 1: keys -> entries grouped by region
// LDS-staged build of the join table (build_hash_table, hash_join.rs:148-175). The table is cut into regions of
// 2^slot_bits slots (a key's region comes from the high half of its hash). This kernel evaluates the fused scan filter
// and the key words of the build rows straight from the build table's columns. A workgroup owns a contiguous row range
// and leaves its valid rows as ENTRIES [row + 2 | key words] — the shape of a slot — in `entries` at the range's own
// position, grouped by region (a counting sort inside the workgroup: count per region in LDS, exclusive scan, re-read the
// rows L2-warm and place them), plus the first entry of every region in `first[workgroup][region]` (n_regions + 1 values).
// k_join_region_build then collects a region's entries from all workgroups, assembles the region's open-addressing
// image in LDS and stores it as whole lines. No global atomic anywhere, no memset of the table. (A first version reserved
// room in the regions with one global atomic per (workgroup, region): 512 workgroups x 16 counters per 64-byte line
// serialise at the memory side — 40 us for any build size.)
struct ScatterLaunch {
  u64* entries;      // nrows entries of (1 + W) words; workgroup g writes [g * rows_per_wg, ...) from the front
  u32* first;        // [n_regions + 1][workgroups]: where workgroup g's entries of a region start inside g's range
  u32* status;
  u32 n_regions;
  u32 rows_per_wg;   // static row range of a workgroup
};

#define QH_SCATTER_BLOCK 1024   // 16 wavefronts per workgroup: a workgroup's row range is two or three rows per thread, so each
                                // of its phases is ONE round of loads (the phases are latency chains, not bandwidth)
template <class P, bool DEVROWS = false>
__device__ __forceinline__ void qh_join_scatter_body(const KArgs& a, const ScatterLaunch& L) {
  constexpr int R = 4, W = P::W, TB = QH_SCATTER_BLOCK;
  u32* cnt = (u32*)qh_dyn_lds;         // [n_regions + 1] rows of this workgroup per region, then the next free place of the region's run
  __shared__ u32 wsum[TB / 64];
  const u32 tid = threadIdx.x, nr = L.n_regions;
  for (u32 r = tid; r <= nr; r += TB) cnt[r] = 0;
  __syncthreads();
  const i64 first = (i64)blockIdx.x * L.rows_per_wg;
  const i64 nrows = DEVROWS ? qh_rows(a) : a.nrows;
  const i64 last = first + L.rows_per_wg < nrows ? first + L.rows_per_wg : nrows;
  u32 err = 0;
  // a range of at most R rows per thread (the rule: 2-3) is evaluated ONCE: the placing pass reuses the counting pass's key
  // words instead of reading the rows again (two dependent chains of column loads per workgroup become one)
  const bool single = last - first <= (i64)TB * R;   // workgroup-uniform
  u64 k[R][W];
  bool ok[R];
  for (int pass = 0; pass < 2; ++pass) {
    for (i64 tb = first; tb < last; tb += (i64)TB * R) {
      if (!single || pass == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const i64 i = tb + (i64)r * TB + tid;
          const bool inb = i < last;
          u32 e = 0;
          ok[r] = P::keys(a, inb ? i : last - 1, k[r], e) && inb;
          if (pass == 0) err |= inb ? e : 0u;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (ok[r]) {
          const u32 reg = qh_region(qh_key_hash<W>(k[r]), nr);
          const u32 pos = atomicAdd(&cnt[reg], 1u);
          if (pass == 1) {
            u64* e = L.entries + ((size_t)first + pos) * (1 + W);
            e[0] = (u64)(tb + (i64)r * TB + tid) + 2;
#pragma unroll
            for (int w = 0; w < W; ++w) e[1 + w] = k[r][w];
          }
        }
      }
    }
    __syncthreads();
    if (pass == 0) {
      // exclusive scan of the per-region counts (thread t owns ch consecutive regions), published to first[workgroup][]
      const u32 ch = (nr + TB) / TB;   // ceil((nr + 1) / TB)
      const u32 r0 = tid * ch;
      u32 sum = 0;
      for (u32 j = 0; j < ch; ++j) sum += r0 + j <= nr ? cnt[r0 + j] : 0u;
      u32 incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)incl, d, 64); if ((int)(tid & 63) >= d) incl += t; }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      u32 run = incl - sum;
      for (u32 w = 0; w < (tid >> 6); ++w) run += wsum[w];
      // first[region][workgroup] (region-major): k_join_region_build reads one region's row of it, contiguously
      for (u32 j = 0; j < ch; ++j)
        if (r0 + j <= nr) { const u32 c = cnt[r0 + j]; cnt[r0 + j] = run; L.first[(size_t)(r0 + j) * gridDim.x + blockIdx.x] = run; run += c; }
      __syncthreads();
    }
  }
  qh_report(L.status, err);
}

// ------------------------------------------------------------------ expression -> key words kernel (hash join keys, partition keys)
// Evaluates the W key words of every row into word-major arrays keys[w * nrows + i]; keyvalid bit i is
// set when every key column of the row is non-null (NULL keys never match, hash_join.rs:191-215).
template <class P>
__device__ __forceinline__ void qh_eval_keys_body(const KArgs& a, u64* keys, u64* keyvalid, u32* status) {
  const i64 nwords = (a.nrows + 63) / 64;
  const i64 wave_global = ((i64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  u32 err = 0;
  for (i64 j = wave_global; j < nwords; j += nwaves) {
    const i64 i = j * 64 + lane;
    const bool inb = i < a.nrows;
    u64 k[P::W];
    u32 e = 0;
    const bool ok = P::keys(a, inb ? i : a.nrows - 1, k, e) && inb;
    err |= inb ? e : 0u;
    if (inb) {
#pragma unroll
      for (int w = 0; w < P::W; ++w) keys[(size_t)w * a.nrows + i] = ok ? k[w] : 0;
    }
    u64 m = qh_ballot(ok);
    if (lane == 0) keyvalid[j] = m;
  }
  qh_report(status, err);
}

// ------------------------------------------------------------------ expression -> sort key images (physical/plan/sort.rs:51-60)
// img[w * nrows + i] = word w of row i's order-preserving key images (NULL rows: 0); keyvalid[k * nwords + j] = validity
// bits of sort key k for rows 64j..64j+63.
// diff[w] receives the OR over all rows of (image word w XOR row 0's image word w): the bits in which the rows differ at
// all. The host sorts only those (a radix pass over bits every row agrees on cannot change the order): a Decimal128 SUM
// whose values fit 34 bits costs 5 digit passes instead of 16, a constant key none.
template <class P>
__device__ __forceinline__ void qh_sort_keys_body(const KArgs& a, u64* img, u64* keyvalid, u64* diff, u32* status) {
  const i64 nwords = (a.nrows + 63) / 64;
  const i64 wave_global = ((i64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  u32 err = 0;
  u64 ref[P::NW > 0 ? P::NW : 1], d[P::NW > 0 ? P::NW : 1];
  {
    u32 v0 = 0, e0 = 0;
    P::images(a, 0, ref, v0, e0);
#pragma unroll
    for (int k = 0; k < P::NW; ++k) d[k] = 0;
  }
  for (i64 j = wave_global; j < nwords; j += nwaves) {
    const i64 i = j * 64 + lane;
    const bool inb = i < a.nrows;
    u64 w[P::NW > 0 ? P::NW : 1];
    u32 valid = 0, e = 0;
    P::images(a, inb ? i : a.nrows - 1, w, valid, e);
    err |= inb ? e : 0u;
    if (inb) {
#pragma unroll
      for (int k = 0; k < P::NW; ++k) { img[(size_t)k * a.nrows + i] = w[k]; d[k] |= w[k] ^ ref[k]; }
    }
#pragma unroll
    for (int k = 0; k < P::NK; ++k) {
      const u64 m = qh_ballot(inb && ((valid >> k) & 1u));
      if (lane == 0) keyvalid[(size_t)k * nwords + j] = m;
    }
  }
#pragma unroll
  for (int k = 0; k < P::NW; ++k) {
    u64 x = d[k];
    for (int m = 32; m >= 1; m >>= 1) x |= qh_shfl_xor64(x, m);
    // one atomic per wavefront and word, and none when it adds no new bit
    if (lane == 0 && (x & ~__hip_atomic_load(&diff[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) (void)__hip_atomic_fetch_or(&diff[k], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  qh_report(status, err);
}

// ------------------------------------------------------------------ expressions -> output columns (physical/plan/projection.rs:27-46)
struct ProjOut {
  void* v[QH_MAXC];   // per computed expression: values (fixed width), value bits (Boolean) or — Utf8 — int32 lengths, then offsets
  u64* n[QH_MAXC];    // validity words of the nullable ones
  u8* d[QH_MAXC];     // Utf8 outputs (round 4): the data bytes, written by the second pass at the scanned offsets
};
// P::row evaluates every expression for row `i` (clamped to the table for the lanes beyond its end, `inb` false there)
// and stores values at `row`, bit-packed outputs at word `j` through wavefront ballots.
// MODE 0: every output's values / bits / validity, and the LENGTH of a computed Utf8 value; MODE 1 (only when there is a Utf8
// output, after the lengths have been scanned into offsets): the expressions once more, the bytes copied to their offsets —
// a CASE over string literals or columns (the reference's type.slt:51: case x when 1 then 'a' else 'b' end).
template <class P, int MODE = 0>
__device__ __forceinline__ void qh_project_body(const KArgs& a, const ProjOut& o, u32* status) {
  const i64 nwords = (a.nrows + 63) / 64;
  const i64 wave_global = ((i64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  u32 err = 0;
  for (i64 j = wave_global; j < nwords; j += nwaves) {
    const i64 row = j * 64 + lane;
    const bool inb = row < a.nrows;
    u32 e = 0;
    P::template row<MODE>(a, o, inb ? row : a.nrows - 1, row, inb, j, lane, e);
    err |= inb ? e : 0u;
  }
  if (MODE == 0) qh_report(status, err);
}
