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
};

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
template <class P>
__device__ __forceinline__ u32 qh_merge_lds_table(u64* ltable, const AggLaunch& L) {
  constexpr int W = P::W;
  u32 used = 0;
  for (u32 s = threadIdx.x; s < L.l_nslots; s += QH_BLOCK) {
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

template <class P>
__device__ __forceinline__ void qh_filter_agg_body(const KArgs& a, const AggLaunch& L0) {
  constexpr int W = P::W;
  constexpr int R = P::R;
  u64* ltable = (u64*)qh_dyn_lds;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  AggLaunch L = L0;
  L.gtable = L0.gtable + (size_t)(blockIdx.x % L0.replicas) * L0.g_nslots * P::SLOT_WORDS;

  if (W > 0) {
    // LDS table: zero = empty slots and identity cells
    const u32 lwords = L.l_nslots * (u32)P::SLOT_WORDS;
    for (u32 k = tid; k < lwords; k += QH_BLOCK) ltable[k] = 0;
    __syncthreads();
  }

  typename P::Part acc;   // W == 0: whole-kernel per-thread accumulator
  if (W == 0) P::part_init(acc);
  u32 err = 0;            // QS_* bits raised by this thread, reported once at the end
  // wave-resident hot-key cache (wave-uniform bookkeeping lives in SGPRs)
  constexpr int KC = P::KC;
  u64 ck[KC > 0 ? KC : 1][W > 0 ? W : 1];
  typename P::Part cacc[KC > 0 ? KC : 1];
  u64 crows[KC > 0 ? KC : 1];
  int nc = 0, singles = 0;
  bool cache_on = KC > 0, use_cache = true;
  u64 seen_pass = 0, seen_hits = 0;

  const i64 tile_rows = (i64)QH_BLOCK * R;
  const i64 ntiles = (a.nrows + tile_rows - 1) / tile_rows;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // the HBM table overflowed somewhere: the host will retry with a larger one, stop streaming (the load is
    // issued with the tile's loads and consumed at the end of the iteration, wave-uniform)
    const u32 overflowed = W > 0 ? __hip_atomic_load(&L.status[QS_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    typename P::Row row[R];
    const i64 tb = t * tile_rows;
    typename P::Raw raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // phase 1: issue the loads of all R rows (branch-free). Out-of-range lanes re-read the table's last row and are
      // masked out afterwards; addressing is (uniform 64-bit tile base) + (32-bit lane offset)
      const u32 o = (u32)r * QH_BLOCK + (u32)tid;
      const bool inb = tb + (i64)o < a.nrows;
      P::load(a, tb, inb ? o : (u32)(a.nrows - 1 - tb), raw[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // phase 2: predicate, key words and aggregate arguments of each row (may branch)
      const bool inb = tb + (i64)((u32)r * QH_BLOCK + (u32)tid) < a.nrows;
      u32 e = 0;
      P::eval(a, raw[r], row[r], e);
      row[r].pass = row[r].pass && inb;
      err |= inb ? e : 0u;
    }
    if (W == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) P::template part_add<true>(acc, row[r], row[r].pass);
      continue;
    }
    // ---- wave-resident hot-key accumulators
    bool pend[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pend[r] = row[r].pass;
    u32 tile_pass = 0, tile_hits = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) tile_pass += (u32)__builtin_popcountll(qh_ballot(pend[r]));
    // (1) rows whose key is already cached: lane-private add under the EXEC mask, no cross-lane traffic, no table
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (k < nc && use_cache) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          bool m = pend[r];
#pragma unroll
          for (int w = 0; w < W; ++w) m = m && (row[r].key[w] == ck[k][w]);
          P::template part_add<false>(cacc[k], row[r], m);
          const u32 c = (u32)__builtin_popcountll(qh_ballot(m));
          crows[k] += c;
          tile_hits += c;
          pend[r] = pend[r] && !m;
        }
      }
    }
    // (2) admit new keys while the cache has room and the data keeps showing duplicates inside a wave
    if (cache_on && nc < KC) {
      int tile_misses = 0;
#pragma unroll
      for (int r0 = 0; r0 < R; ++r0) {
        u64 act = qh_ballot(pend[r0]);
        while (act != 0 && cache_on && nc < KC && tile_misses < 4) {
          const int leader = __builtin_ctzll(act);
          u64 lk[W > 0 ? W : 1];
#pragma unroll
          for (int w = 0; w < W; ++w) lk[w] = qh_readlane64(row[r0].key[w], leader);
          u32 cnt = 0;
          bool mm[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            bool m = pend[r];
#pragma unroll
            for (int w = 0; w < W; ++w) m = m && (row[r].key[w] == lk[w]);
            mm[r] = m;
            cnt += (u32)__builtin_popcountll(qh_ballot(m));
          }
          if (cnt <= 1) {
            // Nobody shares this key inside the wave. Uniform high-cardinality data shows nothing but such keys and the
            // cache is given up after a few tiles; a skewed distribution (Zipf: a long tail AND a few heavy keys) shows
            // them between its heavy keys, so a wave that has already admitted a key keeps looking 8x longer. At most 4
            // such misses per tile bound the cost of looking.
            act &= ~(1ULL << leader);
            ++singles;
            if (singles >= (nc > 0 ? 256 : 32)) cache_on = false;
            if (++tile_misses >= 4) break;
            continue;
          }
#pragma unroll
          for (int k = 0; k < KC; ++k) {
            if (k == nc) {
#pragma unroll
              for (int w = 0; w < W; ++w) ck[k][w] = lk[w];
              P::part_init(cacc[k]);
              crows[k] = cnt;
#pragma unroll
              for (int r = 0; r < R; ++r) { P::template part_add<false>(cacc[k], row[r], mm[r]); pend[r] = pend[r] && !mm[r]; }
            }
          }
          ++nc;
          tile_hits += cnt;
          act = qh_ballot(pend[r0]);
        }
      }
    }
    // the cache must earn its compares: a full cache that serves < 1/4 of the rows is no longer consulted (what it
    // holds is merged at the end of the kernel like any other cached group)
    seen_pass += tile_pass; seen_hits += tile_hits;
    if (nc == KC && seen_pass >= 4096 && seen_hits * 4 < seen_pass) { use_cache = false; cache_on = false; }
    // (3) everything else: one lane per row straight to the LDS-staged table (contention is low when keys are many)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (pend[r]) {
        typename P::Part part;
        P::part_init(part);
        P::template part_add<true>(part, row[r], true);
        qh_update_group<P>(ltable, L, row[r].key, part, err);
      }
    }
    if (__builtin_amdgcn_readfirstlane((int)overflowed)) break;
  }

  // hand the cached groups of this wave to the workgroup's table: one reduction + one update per (wave, key) per KERNEL
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    if (k < nc) {
      P::template part_reduce<false>(cacc[k]);
      P::part_set_rows(cacc[k], crows[k]);
      if (lane == 0) qh_update_group<P>(ltable, L, ck[k], cacc[k], err);
    }
  }
  qh_report(L.status, err);
  if (W == 0) {
    P::template part_reduce<true>(acc);
    if (lane == 0) P::template slot_update<MemHbm>(L.gtable, acc);
    return;
  }
  // ---- merge this workgroup's LDS table into the HBM table
  __syncthreads();
  u32 used = qh_merge_lds_table<P>(ltable, L);
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

template <class P, bool SCATTER>
__device__ __forceinline__ void qh_agg_part_body(const KArgs& a, const PartLaunch& L) {
  constexpr int W = P::W;
  u32* cnt = (u32*)qh_dyn_lds;              // [n_bins] counts (pass 1) / next free record of the bin's run (pass 2)
  const u32 tid = threadIdx.x;
  for (u32 b = tid; b < L.n_bins; b += QH_BLOCK) cnt[b] = SCATTER ? L.hist[(size_t)b * gridDim.x + blockIdx.x] : 0u;
  __syncthreads();
  const i64 first = (i64)blockIdx.x * L.rows_per_wg;
  const i64 last = first + L.rows_per_wg < a.nrows ? first + L.rows_per_wg : a.nrows;
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

struct ReduceLaunch {
  const u64* records;
  const u32* item_first;   // work item k = records [item_first[k], item_first[k + 1]), all of one bin
  u32 n_items;
};

template <class P>
__device__ __forceinline__ void qh_agg_reduce_body(const ReduceLaunch& R, const AggLaunch& L) {
  constexpr int W = P::W;
  u64* ltable = (u64*)qh_dyn_lds;
  const u32 tid = threadIdx.x;
  const u32 lwords = L.l_nslots * (u32)P::SLOT_WORDS;
  u32 err = 0;
  for (u32 item = blockIdx.x; item < R.n_items; item += gridDim.x) {
    for (u32 k = tid; k < lwords; k += QH_BLOCK) ltable[k] = 0;
    __syncthreads();
    const u32 r0 = R.item_first[item], r1 = R.item_first[item + 1];
    constexpr int RR = 4;   // records per thread and iteration: their loads are issued together
    for (u32 i0 = r0; i0 < r1; i0 += QH_BLOCK * RR) {
      u64 key[RR][W > 0 ? W : 1];
      typename P::Part part[RR];
      bool live[RR];
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const u32 i = i0 + (u32)r * QH_BLOCK + tid;
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
    (void)qh_merge_lds_table<P>(ltable, L);
    __syncthreads();
  }
  qh_report(L.status, err);
}

// ------------------------------------------------------------------ predicate -> selection mask kernel
// Filter::execute (physical/plan/filter.rs:28-44): mask word j holds the keep bits of rows 64j..64j+63
// (wavefront ballot), wave_count[j] their popcount; the exclusive scan of wave_count gives every
// wavefront its output offset for the column compaction kernels (selection-vector compaction).
template <class P>
__device__ __forceinline__ void qh_pred_mask_body(const KArgs& a, u64* mask, u32* wave_count, u32* status) {
  const i64 nwords = (a.nrows + 63) / 64;
  const i64 wave_global = ((i64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  u32 err = 0;
  for (i64 j = wave_global; j < nwords; j += nwaves) {
    const i64 i = j * 64 + lane;
    const bool inb = i < a.nrows;
    u32 e = 0;
    const bool keep = P::pred(a, inb ? i : a.nrows - 1, e) && inb;
    err |= inb ? e : 0u;
    u64 m = qh_ballot(keep);
    if (lane == 0) { mask[j] = m; wave_count[j] = (u32)__builtin_popcountll(m); }
  }
  qh_report(status, err);
}

// ------------------------------------------------------------------ hash join: fused scan-filter + key + probe + ordered emit
// read-only lookup in the distinct-key table of the build side (slot = [state][key words W]) after the build kernels
// have completed (plain cached loads)
template <int W>
__device__ __forceinline__ u32 qh_join_find(const u64* table, u32 nslots, const u64* k, u64 h) {
  u32 s = (u32)h & (nslots - 1);
  for (u32 probes = 0; probes < nslots; ++probes) {
    const u64* slot = table + (size_t)s * (1 + W);
    if (slot[0] < QH_READY) return 0xFFFFFFFFu;
    bool eq = true;
#pragma unroll
    for (int w = 0; w < W; ++w) eq &= slot[1 + w] == k[w];
    if (eq) return s;
    s = (s + 1) & (nslots - 1);
  }
  return 0xFFFFFFFFu;
}
template <int W> __device__ __forceinline__ u64 qh_key_hash(const u64* k) {
  u64 h = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) h = qh_mix64(h ^ k[w]);
  return h;
}

struct ProbeLaunch {
  const u64* table;      // distinct build keys
  const u32* bloom;      // hash filter: qh_bloom_bits(h) all set in word ((h >> 32) & bloom_mask) >> 5, else the key is not
                         // in the table (the filter stays in L2)
  const u32* count;      // build rows per slot (unused when start == nullptr)
  const u32* start;      // first position of a slot's rows in `rows`; nullptr: unique build keys (the slot's state word - 2 is the row)
  const u32* rows;       // build rows grouped by slot, ascending inside a slot
  u32* ent_slot;         // out, per tile of 64 * QH_PROBE_R consecutive probe rows (one wavefront's share): the matching rows,
  u32* ent_row;          //      compacted in row order at [tile * TILE, tile * TILE + tile_nent[tile]): slot of the key (unique
                         //      build keys: the build row itself) and probe row. Rows without a match write nothing.
  u32* tile_nent;        // out, per tile: number of matching probe rows
  u32* tile_total;       // out, per tile: number of (build, probe) pairs
  u32* visited;          // build-row bitmap to mark here (LeftSemi / LeftAnti without a residual filter) or nullptr
  u32* status;
  u32 nslots, bloom_mask;
};

// The build's hash filter is a blocked Bloom filter with TWO bits per key inside ONE 32-bit word (word and first bit from
// hash bits 32.., second bit from the top five hash bits): the probe still pays one L2 access per row, but with ~11 bits
// per key the false positives — each of which is a random 128-byte HBM line of the 64 MB table fetched for nothing — drop
// from ~9 % to ~3 % of the probing rows (Q3 J2: 32 M probing rows, 0.3 M true matches).
__device__ __forceinline__ u32 qh_bloom_bits(u64 h) { return (1u << ((u32)(h >> 32) & 31u)) | (1u << (u32)(h >> 59)); }

#define QH_PROBE_R 4   // probe rows per thread and tile
// Probe pass 1 (hash_join.rs:218-275 for every probe batch at once): evaluate the fused scan filter and the key words
// straight from the probe table's columns, look the key up, keep (slot, probe row) of the matching rows only — compacted
// per 256-row tile with ballot/popcount ranks — and the pair count per tile. Pass 2 (k_join_emit) turns the entries into
// ordered (build row, probe row) pairs once the tile totals have been scanned.
// Phases, each a branch-free pass over the thread's R rows so that their loads are in flight together (a lookup is a
// chain of dependent random reads; R independent chains per thread and many waves per CU hide its latency):
// key words -> filter bit -> home slot of the table -> (rarely) the rest of the probe sequence -> row count.
template <class P>
__device__ __forceinline__ void qh_join_probe_body(const KArgs& a, const ProbeLaunch& L) {
  constexpr int R = QH_PROBE_R, TILE = 64 * R, NW = QH_BLOCK / 64;
  const int lane = qh_lane();
  const i64 ntiles = (a.nrows + TILE - 1) / TILE;
  u32 err = 0;
  for (i64 tile = (i64)blockIdx.x * NW + (threadIdx.x >> 6); tile < ntiles; tile += (i64)gridDim.x * NW) {
    u32 sid[R];
    u64 k[R][P::W], h[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const i64 i = tile * TILE + r * 64 + lane;
      const bool inb = i < a.nrows;
      u32 e = 0;
      ok[r] = P::keys(a, inb ? i : a.nrows - 1, k[r], e) && inb;
      err |= inb ? e : 0u;
      h[r] = qh_key_hash<P::W>(k[r]);
    }
    u32 fw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) fw[r] = L.bloom[ok[r] ? (((u32)(h[r] >> 32) & L.bloom_mask) >> 5) : 0u];
    u64 st[R], kw[R][P::W];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u32 bits = qh_bloom_bits(h[r]);
      ok[r] = ok[r] && (fw[r] & bits) == bits;
      const u64* slot = L.table + (size_t)(ok[r] ? ((u32)h[r] & (L.nslots - 1)) : 0u) * (1 + P::W);
      st[r] = slot[0];
#pragma unroll
      for (int w = 0; w < P::W; ++w) kw[r][w] = slot[1 + w];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      sid[r] = 0xFFFFFFFFu;
      if (ok[r] && st[r] >= QH_READY) {
        bool eq = true;
#pragma unroll
        for (int w = 0; w < P::W; ++w) eq &= kw[r][w] == k[r][w];
        const u32 home = (u32)h[r] & (L.nslots - 1);
        sid[r] = eq ? home : qh_join_find<P::W>(L.table, L.nslots, k[r], (u64)home + 1);   // collision: walk on from the next slot
        // unique build keys: the slot's state word carries its one build row, which is all pass 2 needs
        if (!L.start && sid[r] != 0xFFFFFFFFu) sid[r] = (u32)((eq ? st[r] : L.table[(size_t)sid[r] * (1 + P::W)]) - 2);
      }
    }
    u32 total = 0, nent = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool found = sid[r] != 0xFFFFFFFFu;
      const u32 c = L.start ? L.count[found ? sid[r] : 0u] : 1u;
      total += found ? c : 0u;
      const u64 m = qh_ballot(found);
      if (found) {
        const size_t pos = (size_t)tile * TILE + nent + (u32)__builtin_popcountll(m & ((1ULL << lane) - 1));
        L.ent_slot[pos] = sid[r];
        L.ent_row[pos] = (u32)(tile * TILE + r * 64 + lane);
      }
      nent += (u32)__builtin_popcountll(m);
    }
    if (L.visited) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (sid[r] != 0xFFFFFFFFu) {
          const u32 s0 = L.start ? L.start[sid[r]] : 0u, c = L.start ? L.count[sid[r]] : 1u;
          for (u32 q = 0; q < c; ++q) { const u32 b = L.start ? L.rows[s0 + q] : sid[r]; atomicOr(&L.visited[b >> 5], 1u << (b & 31)); }
        }
    }
    total = (u32)qh_wave_sum_u64(total);
    if (lane == 0) { L.tile_total[tile] = total; L.tile_nent[tile] = nent; }
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
  void* v[QH_MAXC];   // per computed expression: values (fixed width) or value bits (Boolean), Arrow layout
  u64* n[QH_MAXC];    // validity words of the nullable ones
};
// P::row evaluates every expression for row `i` (clamped to the table for the lanes beyond its end, `inb` false there)
// and stores values at `row`, bit-packed outputs at word `j` through wavefront ballots.
template <class P>
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
    P::row(a, o, inb ? row : a.nrows - 1, row, inb, j, lane, e);
    err |= inb ? e : 0u;
  }
  qh_report(status, err);
}
