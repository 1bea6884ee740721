// join.cpp — qhip_hash_join_execute: HashJoinExec::execute (physical/plan/join/hash_join.rs:354-384).
//
//   build (left, hash_join.rs:148-175)    key words (JIT) -> open-addressing table of distinct keys + a one-bit-per-hash
//                                          filter (stays in L2) -> unique keys: slot -> row; duplicated keys: rows grouped by
//                                          slot with a stable radix sort (ascending build row inside a key, the order the
//                                          reference's reverse-built chains yield) -> CSR start/count per slot
//   probe (right, hash_join.rs:218-275)   pass 1 (JIT): fused scan filter + key words + lookup straight from the probe
//                                          table's columns -> slot per probe row, pair count per 256-row tile -> scan
//                                          -> pass 2: (build, probe) pairs in probe-row order
//                                          [-> residual JoinFilter on an intermediate batch, join/mod.rs:125-154]
//                                          -> visited bitmap -> Right/Full NULL padding (join/mod.rs:176-207)
//   output (utils/batch.rs:18-61)         every column gathered by the index vectors; one batch per non-empty probe batch,
//                                          then the unmatched-build / semi tail batch (hash_join.rs:277-343, 374-381)
// All probe batches are probed in one launch; batch boundaries are recovered from the pairs' (ascending) probe rows.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <map>

#include "common.hpp"
#include "device/qhip_status.h"
#include "jit.hpp"
#include "kargs_host.hpp"
#include "kernels.hpp"
#include "relops.hpp"

using namespace qhip;

namespace {

uint32_t pow2_ceil32(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return (uint32_t)std::min<uint64_t>(p, 1ULL << 31);
}
int log2u(uint32_t x) { int b = 0; while ((1u << b) < x) ++b; return b; }

uint64_t fnv1a64(const std::string& str, uint64_t h = 1469598103934665603ULL) {
  for (unsigned char c : str) { h ^= c; h *= 1099511628211ULL; }
  return h;
}

uint32_t read_u32(hipStream_t s, const void* dev) {
  uint32_t v = 0;
  copy_sync(s, &v, dev, 4, hipMemcpyDeviceToHost);
  return v;
}

thread_local int g_dense_off = 0;   // > 0: the dense (direct-address) layout is switched off for the join being (re-)run

qhip_table* hash_join(Ctx* ctx, const qhip_table* L, const qhip_table* R, int join_type, const qhip_expr* lex, int nlex, const qhip_expr* rex,
                      int nrex, const int32_t* on_l, const int32_t* on_r, int n_on, const qhip_expr* fex, int nfex, int froot,
                      const int32_t* fsides, const int32_t* fcols, int nfcols, int lpred, int rpred) {
  trace_point("join: entry");
  QHIP_HIP_CHECK(hipSetDevice(ctx->device));
  memset(&ctx->stats, 0, sizeof(ctx->stats));
  ctx->stats_timing_pending = 0;
  if (n_on <= 0) fail(QHIP_INVALID_ARGUMENT, "Internal error: On constraints in HashJoinExec should be non-empty");
  if (join_type < QHIP_JOIN_LEFT || join_type > QHIP_JOIN_LEFT_ANTI) fail(QHIP_INVALID_ARGUMENT, "unknown join type");
  if (L->num_rows >= (int64_t)kNullIdx - 1 || R->num_rows >= (int64_t)kNullIdx - 1)
    fail(QHIP_UNSUPPORTED, "join inputs of 2^32 - 2 rows or more are not supported");
  hipStream_t s = ctx->stream;
  // a probe side of deferred size is made exact first; a build side of deferred size is read as it is when the region
  // build takes it (qk_join_scatter stops at the device-side row count), else it is made exact too (below)
  settle_rows(R);
  const uint64_t B = (uint64_t)L->num_rows, P = (uint64_t)R->num_rows;
  const bool semi_anti = join_type == QHIP_JOIN_LEFT_SEMI || join_type == QHIP_JOIN_LEFT_ANTI;
  const bool pad_right = join_type == QHIP_JOIN_RIGHT || join_type == QHIP_JOIN_FULL;

  // ---- key words of both sides (only the columns the key / scan-filter expressions read are gathered if deferred)
  // (measured on Q3's second join: reading the build key through join 1's index vector inside qk_join_scatter costs 17 us
  // more than the separate gather it saves — 92 vs 75 us — so the build side gathers; QHIP_LATE_GATHER_BUILD=1 switches it on)
  // ---- dense (direct-address) layout? ONE integer key column whose build-side values span a small range [kmin, kmax]
  // (DevColumn::range, found once per base table: looked at BEFORE the key column is gathered, while a deferred gather
  // still names its source): the table is then an exact bitmap over the range + row_of[key - kmin]
  // (qh_join_dense_build_body / qh_join_probe_dense_body). QHIP_JOIN_DENSE: 0 never, 1 when the range is within 256x the
  // build rows (default), 2 whenever the key qualifies (tests).
  // (g_dense_off: this join is being re-run after its dense table did not fit the device memory, see below)
  const int dense_mode = g_dense_off > 0 ? 0 : env_int("QHIP_JOIN_DENSE", 1);
  bool dense_candidate = false;
  int64_t kmin = 0, kmax = 0;
  uint64_t dense_n = 0;
  if (dense_mode != 0 && n_on == 1 && B > 0 && on_l[0] >= 0 && on_l[0] < nlex && lex[on_l[0]].kind == QHIP_EXPR_COLUMN && lex[on_l[0]].column >= 0 &&
      lex[on_l[0]].column < (int)L->cols.size()) {
    const DevColumn& kc = L->cols[(size_t)lex[on_l[0]].column];
    const int id = kc.type.id;
    if ((id == QHIP_INT64 || id == QHIP_INT32 || id == QHIP_UINT8 || id == QHIP_DATE32 || id == QHIP_DATE64 || (id >= QHIP_TIME32_S && id <= QHIP_TIME64_NS)) &&
        key_range_of(ctx, kc, kmin, kmax)) {
      const uint64_t span = (uint64_t)kmax - (uint64_t)kmin;   // (kmax >= kmin; the difference fits 64 unsigned bits)
      // automatic mode: the bitmap at most 32 bytes per build row AND the table bounded in absolute terms — row_of is 4 bytes per
      // VALUE of the range (span < 2^28: at most 1 GB beside a 32 MB bitmap; Q3's join 2 at SF100: 600 M values would be 2.4 GB
      // for 15 M build rows — the hashed layouts take over there). Whatever the mode, an allocation that fails falls back to
      // the hashed layouts instead of failing the query.
      const uint64_t max_span = (uint64_t)1 << std::max(16, std::min(30, env_int("QHIP_JOIN_DENSE_MAX_SPAN_BITS", 28)));
      if (span < (1ULL << 30) && (dense_mode == 2 || (span < 256 * B + 65536 && span < max_span))) { dense_candidate = true; dense_n = span + 1; }
    }
  }
  // the build key of a join over a join's output is a deferred gather: the dense build kernel reads it THROUGH the index
  // vector (one dependent load more per row, no gather launch + write + re-read of the key: Q3's join 2 24 + 38 -> 54 us);
  // the region build's scatter kernel lost by that (92 vs 75 us), so the hashed layouts gather first
  const bool late_build = env_int("QHIP_LATE_GATHER_BUILD", dense_candidate ? 1 : 0) != 0;
  resolve_referenced(ctx, L, lex, nlex, late_build);
  resolve_referenced(ctx, R, rex, nrex);
  std::vector<InputCol> lcols = input_cols_of(L, late_build), rcols = input_cols_of(R);
  ensure_utf8_key_lengths(ctx, L, lex, nlex, on_l, n_on, lcols);
  ensure_utf8_key_lengths(ctx, R, rex, nrex, on_r, n_on, rcols);
  ensure_narrow_int_columns(ctx, R, rex, nrex, rcols, (int64_t)env_int("QHIP_STATS_MIN_ROWS", 1 << 22));   // (the probe side streams its key column: 4 bytes where they do)
  // both sides must pack a Utf8 key into the same number of words
  for (int k = 0; k < n_on; ++k) {
    if (on_l[k] < 0 || on_l[k] >= nlex || on_r[k] < 0 || on_r[k] >= nrex) fail(QHIP_INVALID_ARGUMENT, "join key index out of range");
    const qhip_expr &le = lex[on_l[k]], &re = rex[on_r[k]];
    if (le.kind == QHIP_EXPR_COLUMN && re.kind == QHIP_EXPR_COLUMN && le.column >= 0 && le.column < (int)lcols.size() && re.column >= 0 &&
        re.column < (int)rcols.size() && lcols[(size_t)le.column].type.id == QHIP_UTF8 && rcols[(size_t)re.column].type.id == QHIP_UTF8) {
      const int m = std::max(lcols[(size_t)le.column].utf8_max_len, rcols[(size_t)re.column].utf8_max_len);
      lcols[(size_t)le.column].utf8_max_len = m;
      rcols[(size_t)re.column].utf8_max_len = m;
    }
  }
  DevBuf lkeys, lvalid;
  time_mark(ctx, 0);
  if ((lpred >= 0 || rpred >= 0) && join_type != QHIP_JOIN_INNER)
    fail(QHIP_INVALID_ARGUMENT, "fused scan filters are only defined for Inner joins (rows rejected by a filter must not surface as unmatched rows)");
  if (lpred >= nlex || rpred >= nrex) fail(QHIP_INVALID_ARGUMENT, "scan filter index out of range");
  // The build's status (key-evaluation errors, duplicate keys?) is needed before the probe only to choose between the
  // unique-key and the CSR layout. Unique keys are the rule (every FK -> PK join), so unless this build side is known to
  // have had duplicates the probe is launched on that assumption and the build status is read together with the probe's:
  // one host round trip less per join. A wrong guess is memory-safe (a slot's state word always names a valid build row),
  // is detected below, remembered, and the join runs again the careful way.
  // (the hint identifies the build side by its key / filter expressions and row count)
  auto fold_exprs = [](uint64_t h, const qhip_expr* ex, int n) {
    for (int k = 0; k < n; ++k) {
      qhip_expr e = ex[k];
      const char* str = e.lit_str;
      e.lit_str = nullptr;
      h = fnv1a64(std::string((const char*)&e, sizeof e), h);
      if (str && e.lit_len > 0 && e.kind == QHIP_EXPR_LITERAL) h = fnv1a64(std::string(str, (size_t)e.lit_len), h);
    }
    return h;
  };
  uint64_t dup_hint = fold_exprs(B * 0x9E3779B97F4A7C15ULL + (uint64_t)(lpred + 1), lex, nlex);
  for (int k = 0; k < n_on; ++k) dup_hint = dup_hint * 1099511628211ULL + (uint64_t)on_l[k];
  const bool speculate = env_int("QHIP_JOIN_FORCE_CSR", 0) == 0 && env_int("QHIP_JOIN_NO_SPECULATION", 0) == 0 && !ctx->join_dup_builds.count(dup_hint);
  const int region_mode = env_int("QHIP_JOIN_REGION", 1);   // LDS-staged region build: 0 never, 1 when it pays, 2 always (tests)
  const bool dense = dense_candidate && speculate;   // (unique build keys assumed and checked, like the region build)
  const bool want_regions = !dense && speculate && B > 0 && (region_mode == 2 || (region_mode == 1 && B >= 2048));
  // Deferred sizing (qhip.h: qhip_ctx_allow_deferred_sizes): this join as a whole is identified by both sides' expressions,
  // its type and the probe rows (the build rows too unless they are themselves a capacity)
  uint64_t size_key = (L->rows_dev ? 0 : B) * 0x9E3779B97F4A7C15ULL + P * 0xD6E8FEB86659FD93ULL + ((uint64_t)(join_type + 1) << 56) +
                      ((uint64_t)(lpred + 1) << 20) + (uint64_t)(rpred + 1);
  size_key = fold_exprs(fold_exprs(size_key, lex, nlex), rex, nrex);
  for (int k = 0; k < n_on; ++k) size_key = (size_key * 1099511628211ULL + (uint64_t)on_l[k]) * 1099511628211ULL + (uint64_t)on_r[k];
  const auto hint = ctx->join_size_hints.find(size_key);
  const bool defer = ctx->allow_deferred_sizes > 0 && speculate && join_type == QHIP_JOIN_INNER && froot < 0 && B > 0 && P > 0 &&
                     hint != ctx->join_size_hints.end() && env_int("QHIP_JOIN_NO_DEFER", 0) == 0;
  // room for what the join produced last time + 1/8 + 1024 (an FK -> PK join cannot exceed its probe rows)
  const uint64_t defer_cap = defer ? std::min<uint64_t>(P, hint->second + hint->second / 8 + 1024) : 0;
  // device status block of the call: [build status words | probe status words | pair total], read back once
  // (zeroed_block: handed out clean from the context's ring — no memset launch per join)
  uint32_t* const dstat = zeroed_block(ctx);
  // the lowered key plans of both sides (typing + generated source + loaded kernels) are cached per context, keyed by the
  // column signatures and the expression PODs: a repeated join costs no typing, code generation or module lookup
  std::string pkey = "join|";
  {
    auto put = [&](const void* p, size_t n) { pkey.append((const char*)p, n); };
    for (const std::vector<InputCol>* cols : {&lcols, &rcols}) {
      for (auto& ic : *cols) {
        const int v[8] = {ic.type.id, ic.type.precision, ic.type.scale, ic.has_nulls ? 1 : 0, ic.utf8_max_len, ic.utf8_fixed1 ? 1 : 0, ic.indirect ? 1 : 0, ic.narrow_bytes};
        put(v, sizeof v);
      }
      put("|", 1);
    }
    auto put_exprs = [&](const qhip_expr* ex, int n) {
      for (int k = 0; k < n; ++k) {
        qhip_expr e = ex[k];
        const char* str = e.lit_str; const int64_t len = e.lit_len;
        e.lit_str = nullptr;
        put(&e, sizeof e);
        if (str && len > 0 && e.kind == QHIP_EXPR_LITERAL) put(str, (size_t)len);
      }
      put("|", 1);
    };
    put_exprs(lex, nlex);
    put_exprs(rex, nrex);
    put(on_l, sizeof(int32_t) * (size_t)n_on);
    put(on_r, sizeof(int32_t) * (size_t)n_on);
    const int v[5] = {lpred, rpred, want_regions ? 1 : 0, L->rows_dev ? 1 : 0, dense ? 1 : 0};
    put(v, sizeof v);
  }
  struct JoinPlan {
    KeysPlan lkp, rkp; std::shared_ptr<Module> lmod; std::map<std::string, std::shared_ptr<Module>> rmods; DevBuf lstr, rstr;   // (+ the string literals, uploaded once)
    // the dense build's byte map, kept between executions: bytes[key - min] = the execution's stamp (1..255), so that it is
    // cleared once in 255 executions instead of every time (8x the bitmap's bytes: 60 MB for Q3's join 2)
    std::shared_ptr<DevBuf> bytemap; uint64_t bytemap_n = 0; uint32_t bytemap_gen = 0;
  };
  std::shared_ptr<JoinPlan> jp;
  {
    auto cached = ctx->plan_cache.find(pkey);
    if (cached != ctx->plan_cache.end()) jp = std::static_pointer_cast<JoinPlan>(cached->second);
    else {
      jp = std::make_shared<JoinPlan>();
      ExprSet les, res;
      les.build(lex, nlex, lcols);
      res.build(rex, nrex, rcols);
      plan_keys(les, lcols, on_l, n_on, jp->lkp, lpred, dense ? KEYS_KERNEL_DENSE_BUILD : want_regions ? KEYS_KERNEL_SCATTER : KEYS_KERNEL_EVAL,
                (dense || want_regions) && L->rows_dev);
      plan_keys(res, rcols, on_r, n_on, jp->rkp, rpred, dense ? KEYS_KERNEL_DENSE_PROBE : KEYS_KERNEL_PROBE);   // the probe side's keys are evaluated inside the probe kernel
      for (int k = 0; k < n_on; ++k)
        if (jp->lkp.keys[(size_t)k].type != jp->rkp.keys[(size_t)k].type)   // arrow's eq (hash_join.rs:203) needs identical types
          fail(QHIP_INVALID_ARGUMENT, "Invalid argument error: Invalid comparison operation: " + dtype_name(jp->lkp.keys[(size_t)k].type) +
                                          " == " + dtype_name(jp->rkp.keys[(size_t)k].type));
      if (jp->rkp.W != jp->lkp.W) fail(QHIP_HIP_ERROR, "join key layouts of the two sides differ (internal error)");
      if (ctx->plan_cache.size() > 4096) ctx->plan_cache.clear();
      ctx->plan_cache[pkey] = jp;
    }
  }
  const KeysPlan &lkp = jp->lkp, &rkp = jp->rkp;
  const int W = lkp.W;

  // ---- build. Two layouts of the distinct-key table (slot = [state | key words], state - 2 = the build row):
  //  * region layout, LDS-staged (the default while unique build keys are assumed): the table is cut into regions of 2^sb
  //    slots at load <= 1/2; qk_join_scatter (JIT: fused scan filter + key words) drops every build row as an entry into its
  //    region, k_join_region_build assembles each region and its slice of the hash filter in LDS and stores them as whole
  //    lines. No memset, no per-row HBM atomic. Duplicate keys / an overfull region are detected there -> legacy layout.
  //  * legacy layout: ONE open-addressing table filled with agent-scope atomics (k_join_build_insert), per-slot counts and,
  //    for duplicated keys, the CSR of the build rows of every key.
  uint32_t n_regions = 0, slot_bits = 0, bword_bits = 0;
  if (want_regions) {
    // region = the biggest power of two of slots within 32 KB of LDS (W = 1: 2048 slots); 64 KB when that keeps the
    // number of regions (two LDS counters each in qk_join_scatter) within 8192
    slot_bits = 4;
    while (((size_t)16 * (1 + W) << slot_bits) <= 32 * 1024) ++slot_bits;
    // table load: 1/3 when every build row enters the table, 1/2 of the rows when a fused scan filter keeps only part of them
    // (measured on Q3's second join, 1.46 M rows: load 1/2 -> 1/3 build 88 -> 79 us, probe 221 -> 209 us — fewer collisions
    // in the LDS inserts, fewer filter bits set; 0.6 / 0.7 cost 101 / 141 and 248 / 325 us)
    uint64_t load_pct = (uint64_t)std::max(25, std::min(80, env_int("QHIP_JOIN_REGION_LOAD", lpred >= 0 ? 50 : 33)));
    auto regions_for = [&](uint32_t sb) { const uint64_t per = std::max<uint64_t>(1, ((1ull << sb) * load_pct) / 100); return (uint32_t)((B + per - 1) / per); };   // load <= load_pct %
    if (load_pct < 50 && regions_for(slot_bits) > 8192) load_pct = 50;   // (a big build side: rather load 1/2 than more than 8192 regions)
    if (env_int("QHIP_JOIN_REGION_SLOT_BITS", 0) >= 4) slot_bits = (uint32_t)env_int("QHIP_JOIN_REGION_SLOT_BITS", 0);
    else if (regions_for(slot_bits) > 8192 && ((size_t)16 * (1 + W) << slot_bits) <= 60 * 1024) ++slot_bits;
    n_regions = regions_for(slot_bits);
    bword_bits = slot_bits - 3;   // 64-bit filter words per region: 8 filter bits per slot = >= 16 per key
    if (n_regions > 8192 || (((size_t)8 * (1 + W)) << slot_bits) + ((size_t)8 << bword_bits) > 64 * 1024) n_regions = 0;
  }
  const bool region_build = n_regions > 0;
  // only qk_join_scatter reads a device-side row count: otherwise wait, shrink, start over. A join type with a build-side
  // tail (Left / Full / LeftSemi / LeftAnti) needs the EXACT build row count as well: the visited bitmap is scanned over
  // B rows, and the pad rows [count, capacity) of a deferred table are never inserted, hence never visited — they would
  // surface as unmatched build rows.
  if (L->rows_dev && ((!region_build && !dense) || (join_type != QHIP_JOIN_INNER && join_type != QHIP_JOIN_RIGHT))) {
    settle_rows(L);
    return hash_join(ctx, L, R, join_type, lex, nlex, rex, nrex, on_l, on_r, n_on, fex, nfex, froot, fsides, fcols, nfcols, lpred, rpred);
  }
  const uint32_t dense_words = dense ? (uint32_t)((dense_n + 31) / 32) : 0;
  const uint32_t nslots = dense ? 0 : region_build ? n_regions << slot_bits : std::max<uint32_t>(16, pow2_ceil32(B * 2));
  // hash filter: 64-bit words, 8 bits per slot (region layout: 2^bword_bits words per region)
  const uint32_t filter_words = dense ? 0 : region_build ? n_regions << bword_bits : std::max<uint32_t>(16, nslots / 8);
  // one arena: [table | count | filter] (legacy: zero-filled; region layout: every byte of table and filter is stored by
  // k_join_region_build, the counts are not used)
  // (dense layout: table = u32 row_of[dense_n], never initialised — read only where the bitmap says a key is there; "filter"
  // = the exact bitmap, the only part that is cleared)
  const size_t table_bytes = dense ? ((size_t)dense_n * 4 + 127) / 128 * 128 : (size_t)nslots * (1 + W) * 8;
  const size_t count_bytes = (region_build || dense) ? 0 : (((size_t)nslots + 2) * 4 + 7) / 8 * 8;
  const size_t bloom_bytes = dense ? ((size_t)dense_words * 4 + 127) / 128 * 128 : (size_t)filter_words * 8;
  DevBuf arena;
  try {
    arena.alloc(table_bytes + count_bytes + bloom_bytes);
  } catch (const Error& e) {
    if (e.code != QHIP_OUT_OF_MEMORY || !dense) throw;
    // the direct-address table does not fit: the hashed layouts need 16-24 bytes per build ROW, not 4 per key VALUE
    struct Off { Off() { ++g_dense_off; } ~Off() { --g_dense_off; } } off;
    return hash_join(ctx, L, R, join_type, lex, nlex, rex, nrex, on_l, on_r, n_on, fex, nfex, froot, fsides, fcols, nfcols, lpred, rpred);
  }
  uint64_t* table = arena.as<uint64_t>();
  uint32_t* count = (uint32_t*)(arena.as<uint8_t>() + table_bytes);
  uint64_t* bloom = (uint64_t*)(arena.as<uint8_t>() + table_bytes + count_bytes);
  DevBuf start, row_slot, sorted_rows;
  if (dense) {
    if (!jp->lmod) jp->lmod = get_module(ctx, lkp.source, lkp.kernel_name);
    HKArgs ka;
    fill_kargs(ctx, L, lkp.bind, ka, jp->lstr);
    // Byte-map form (round 4, QHIP_JOIN_DENSE_BYTEMAP=1; OFF by default): the build stamps one byte per key with a plain store,
    // a second kernel packs the bytes into the bitmap the probe reads and detects duplicate keys by counting — built to get
    // rid of the bitmap's scattered atomics (~32 G/s at the memory side) and measured SIX TIMES SLOWER on Q3 at SF10 (A/B on one
    // box, profiles/r04_dense_build_bytemap.txt): join 2's build 61 -> 352 us, join 1's 39 -> 86 us. A scattered ONE-BYTE store is
    // the worst thing one can ask of this memory system: every one becomes a read-modify-write of its ECC word at the memory
    // side (~5 G per second), where a scattered 4-byte atomic OR is served at ~32 G/s. Kept as a tested switch, not as a path.
    bool bytemap = env_int("QHIP_JOIN_DENSE_BYTEMAP", 0) != 0 && dense_n <= (1ull << std::max(10, std::min(30, env_int("QHIP_JOIN_DENSE_BYTEMAP_MAX_BITS", 27))));
    HDenseBuildLaunch dl;
    dl.bits = (uint32_t*)bloom; dl.row_of = (uint32_t*)table; dl.status = dstat;
    dl.kmin = (uint64_t)kmin; dl.n = (uint32_t)dense_n;
    if (bytemap) {
      const size_t map_bytes = (size_t)dense_words * 32;   // (whole bitmap words: the packing kernel reads 32 bytes per word)
      if (!jp->bytemap || jp->bytemap_n != dense_n || jp->bytemap_gen >= 255) {
        if (!jp->bytemap || jp->bytemap_n != dense_n) {
          jp->bytemap.reset();
          jp->bytemap_n = 0;
          try {
            jp->bytemap = std::make_shared<DevBuf>(map_bytes);
          } catch (const Error& e) {
            if (e.code != QHIP_OUT_OF_MEMORY) throw;
            bytemap = false;   // (the map stays with the plan between executions: when HBM is short, the atomic form needs no such buffer)
          }
        }
        if (bytemap) {
          QHIP_HIP_CHECK(hipMemsetAsync(jp->bytemap->ptr, 0, map_bytes, s));
          jp->bytemap_n = dense_n;
          jp->bytemap_gen = 0;
        }
      }
    }
    if (bytemap) {
      dl.gen = ++jp->bytemap_gen;
      dl.bytes = jp->bytemap->as<uint8_t>();
      dl.counters = zeroed_block(ctx);
    } else {
      QHIP_HIP_CHECK(hipMemsetAsync(bloom, 0, bloom_bytes, s));
    }
    void* args[] = {&ka, &dl};
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((B + 1023) / 1024, (uint64_t)ctx->num_cus * 8));
    trace_point("join: first launch");
    QHIP_HIP_CHECK(hipModuleLaunchKernel(jp->lmod->fn, grid, 1, 1, 256, 1, 1, 0, s, args, nullptr));
    if (bytemap) launch_bytes_to_bits(dl.bytes, dl.gen, dl.bits, dense_words, dl.counters, dstat, s);
    trace_point("join: first launch done");
  } else if (region_build) {
    if (!jp->lmod) jp->lmod = get_module(ctx, lkp.source, lkp.kernel_name);
    const std::shared_ptr<Module>& mod = jp->lmod;
    HKArgs ka;
    fill_kargs(ctx, L, lkp.bind, ka, jp->lstr);
    // a step-1 workgroup (1024 threads) owns a row range of two or three rows per thread when the rows allow (2
    // workgroups per CU); its entries stay inside the range, so nothing is shared between workgroups
    uint64_t wgs = std::max<uint64_t>(1, std::min<uint64_t>((B + 2047) / 2048, (uint64_t)std::max(1, env_int("QHIP_JOIN_SCATTER_WGS", ctx->num_cus * 2))));
    const uint64_t rows_per_wg = (((B + wgs - 1) / wgs) + 63) / 64 * 64;
    wgs = (B + rows_per_wg - 1) / rows_per_wg;
    DevBuf entries(B * (1 + (size_t)W) * 8), first(wgs * ((size_t)n_regions + 1) * 4);
    HScatterLaunch sl;
    sl.entries = entries.as<uint64_t>(); sl.first = first.as<uint32_t>(); sl.status = dstat;
    sl.n_regions = n_regions; sl.rows_per_wg = (uint32_t)rows_per_wg;
    void* args[] = {&ka, &sl};
    trace_point("join: first launch");
    QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, (unsigned)wgs, 1, 1, 1024, 1, 1, (n_regions + 1) * 4, s, args, nullptr));
    trace_point("join: first launch done");
    launch_join_region_build(W, entries.as<uint64_t>(), first.as<uint32_t>(), (uint32_t)wgs, (uint32_t)rows_per_wg, table, bloom, n_regions,
                             slot_bits, bword_bits, dstat, s);
    // (entries / first go back to the pool here; whoever gets them next runs on the same stream, i.e. afterwards)
  } else {
    QHIP_HIP_CHECK(hipMemsetAsync(arena.ptr, 0, arena.bytes, s));
    row_slot.alloc((B + 1) * 4);
    ExprSet les;
    les.build(lex, nlex, lcols);
    KeysPlan lkp_eval;   // (eval_key_words plans by itself: the same layout as lkp)
    eval_key_words(ctx, L, les, lcols, on_l, n_on, lkp_eval, lkeys, lvalid, lpred, true, dstat);
    launch_join_build_insert(W, lkeys.as<uint64_t>(), lvalid.as<uint64_t>(), B, table, nslots, row_slot.as<uint32_t>(), count, bloom,
                             filter_words - 1, dstat, s);
  }
  uint32_t max_count = 0;
  // read-backs land in the context's page-locked scratch: a D2H copy into pageable memory is a stream round trip of its
  // own, so two of them plus the synchronize cost three waits where one does
  uint32_t* const st_build = (uint32_t*)ctx->pinned;    // pinned mirror of the device status block: [build | probe | pair total]
  uint32_t* const st = st_build + QS_WORDS;
  auto check_build_status = [&] {
    check_status_words(st_build);
    if (st_build[QS_OVERFLOW]) fail(QHIP_HIP_ERROR, "join build table overflow (internal error)");
    max_count = st_build[QS_MAXCOUNT];
  };
  if (!speculate) {
    QHIP_HIP_CHECK(hipMemcpyAsync(st_build, dstat, QS_WORDS * 4, hipMemcpyDeviceToHost, s));   // key evaluation + build
    QHIP_HIP_CHECK(sync_stream(s));
    verify_pending_sizes(ctx);
    check_build_status();
  }
  // Unique build keys (every FK -> PK join): each slot's state word names its one build row and nothing else is needed.
  // Otherwise group the build rows by slot with a stable radix sort (ascending build row inside a key: the order the
  // reference's reverse-built chains yield) into a CSR.
  const bool unique_keys = speculate || (max_count <= 1 && env_int("QHIP_JOIN_FORCE_CSR", 0) == 0);
  const uint32_t* start_ptr = nullptr;
  const uint32_t* rows_ptr = nullptr;
  if (!unique_keys) {
    launch_join_full_counts(W, table, nslots, count, s);
    DevBuf sorted_slot((B + 1) * 4), iota((B + 1) * 4);
    sorted_rows.alloc((B + 1) * 4);
    start.alloc(((size_t)nslots + 1) * 4);
    launch_iota_u32(iota.as<uint32_t>(), B, s);
    stable_sort_pairs_u32(row_slot.as<uint32_t>(), sorted_slot.as<uint32_t>(), iota.as<uint32_t>(), sorted_rows.as<uint32_t>(), B,
                          log2u(nslots) + 1, s);
    exclusive_scan_u32(count, start.as<uint32_t>(), nslots, nullptr, s);
    // (the sort temporaries go back to the pool here; whoever gets them next runs on the same stream, i.e. afterwards)
    start_ptr = start.as<uint32_t>();
    rows_ptr = sorted_rows.as<uint32_t>();
  }

  const bool has_tail = join_type == QHIP_JOIN_LEFT || join_type == QHIP_JOIN_FULL || semi_anti;
  const uint64_t vwords = ((B + 63) / 64) * 2 + 2;
  DevBuf visited(vwords * 4);
  if (has_tail) QHIP_HIP_CHECK(hipMemsetAsync(visited.ptr, 0, visited.bytes, s));
  bool visited_done = false;

  // ---- probe. Pass 1 (JIT: fused scan filter + key words + lookup straight from the probe table's columns) leaves one
  // (slot, probe row) entry per MATCHING probe row and one pair count per 256-row tile; the tile counts are scanned; pass 2 writes the pairs in
  // probe-row order. LeftSemi / LeftAnti without a residual filter only need the visited bits: pass 1 sets them.
  DevBuf ent_slot((P + 1) * 4), ent_row((P + 1) * 4), cnt, pair_off, b_idx, p_idx;
  std::string probe_name = dense ? "qk_join_probe_dense" : "qk_join_probe";   // (the entry point launched, for the statistics)
  uint64_t M = 0;
  const uint32_t* deferred_slot = nullptr;
  // The join key of every output row, written by pass 2 for nothing (dense layout: a matching entry holds key - min): an Inner
  // join on ONE Int64 column with unique build keys hands its key columns on as PLAIN columns instead of deferred gathers, so
  // that a GROUP BY over them (Q3 groups by l_orderkey) streams 8 bytes per row instead of gathering through the index vector —
  // one of the aggregate's five random reads per row gone (QHIP_JOIN_KEY_MATERIALIZE=0: gathers like every other column)
  const bool mat_key = dense && unique_keys && join_type == QHIP_JOIN_INNER && froot < 0 && n_on == 1 && rex[on_r[0]].kind == QHIP_EXPR_COLUMN &&
                       rex[on_r[0]].column >= 0 && rex[on_r[0]].column < (int)R->cols.size() && R->cols[(size_t)rex[on_r[0]].column].type.id == QHIP_INT64 &&
                       env_int("QHIP_JOIN_KEY_MATERIALIZE", 1) != 0;
  std::shared_ptr<DevBuf> key_vals;
  std::shared_ptr<DevBuf> rows_blk;
  std::shared_ptr<uint64_t> rows_final;
  bool probe_timed = false;
  const bool want_pairs = !(semi_anti && froot < 0);
  const bool mark_in_probe = has_tail && froot < 0;             // with a residual filter only surviving pairs mark
  if (P > 0) {
    // which probe kernel (the table layout is a template parameter of the generated kernel). Dense layout: probe sides of
    // less than one tile run the general kernel; else a lane owns consecutive rows (16-byte column loads,
    // QHIP_JOIN_DENSE_WIDE=0: not), and QHIP_JOIN_DENSE_LDS stages the bitmap in LDS: 1 = when all of it fits the CU's
    // 160 KB, the hybrid kernel (first 160 KB in LDS, the rest through L2) up to four times that; 2 = hybrid always
    const uint64_t kProbeTileRows = (uint64_t)64 * (uint64_t)rkp.probe_r;   // one wavefront's tile: 64 * P::PROBE_R consecutive probe rows
    // LDS staging of the dense probe's entries: per wavefront, at least one tile's worth behind the flush threshold
    const uint32_t stage_extra = (uint32_t)std::max(0, env_int("QHIP_DENSE_STAGE_EXTRA", 256));
    const uint32_t stage_cap_small = (uint32_t)kProbeTileRows + stage_extra, stage_cap_lds = (uint32_t)kProbeTileRows + 128;
    const uint32_t kLdsWords = (160 * 1024 - 16 * 2 * 4 * stage_cap_lds) / 4 - 2;   // what the 16 wavefronts' staging areas leave of the CU's 160 KB
    // QHIP_JOIN_DENSE_LDS: 0 never, 1 (default) when at least half of the bitmap fits the CU's LDS and the probe side is
    // big enough to pay for every workgroup's copy of it, 2 always (tests). Measured on Q3 SF10's first join (15 M orders
    // probing 1.5 M customer keys at random, 188 KB of bitmap, 60 % of it staged): 55-59 us against 72-80 us through L1 / L2
    // — a random 4-byte lookup drags a 128-byte line from L2 into L1; the second join's bitmap (7.5 MB, read in key order
    // by lineitem) gains nothing from LDS.
    const int lds_mode = env_int("QHIP_JOIN_DENSE_LDS", 1);
    const char* probe_kernel = region_build ? "qk_join_probe" : "qk_join_probe_onetable";
    uint32_t lds_words = 0;
    if (dense) {
      probe_kernel = "qk_join_probe_dense";
      if (P >= kProbeTileRows) {
        if (env_int("QHIP_JOIN_DENSE_WIDE", 1) != 0) probe_kernel = "qk_join_probe_dense_wide";
        const bool pays = lds_mode == 2 || (lds_mode == 1 && P >= (1u << 20) && P >= 32ull * std::min(dense_words, kLdsWords) && dense_words <= 2 * kLdsWords);
        if (pays && dense_words <= kLdsWords) { probe_kernel = "qk_join_probe_dense_lds"; lds_words = dense_words; }
        else if (pays) { probe_kernel = "qk_join_probe_dense_hybrid"; lds_words = kLdsWords; }
      }
    }
    const bool dense_lds = lds_words > 0;
    probe_name = probe_kernel;
    std::shared_ptr<Module>& rmod = jp->rmods[probe_kernel];
    if (!rmod) rmod = get_module(ctx, rkp.source, probe_kernel);
    const std::shared_ptr<Module>& mod = rmod;
    HKArgs ka;
    fill_kargs(ctx, R, rkp.bind, ka, jp->rstr);
    const uint64_t ntiles = (P + kProbeTileRows - 1) / kProbeTileRows;
    // A wavefront owns a CHUNK of consecutive tiles; a chunk's matches are one run of entries with one count (the scan and
    // pass 2 work per chunk). The grid is sized in whole ROUNDS of the wavefronts the chip holds at once (occupancy of the
    // loaded kernel x CUs), each wavefront with at most QHIP_PROBE_TILES_PER_WAVE (32) tiles: a wavefront pays ~two extra
    // trips to fill its pipeline whatever its chunk, and a last round that fills a third of the chip costs a whole round
    // (round 2 gave every wavefront 12 tiles: Q3's lineitem probe ran 2.4 rounds at 12 / 14 pipeline efficiency — 63 % of the
    // 6.5 TB/s a bare kernel with the same access pattern reaches, tools/micro/stream_widths.hip).
    const unsigned waves_per_wg = dense_lds ? 16 : 4;
    // dynamic LDS: [bitmap words of the LDS variants | per wavefront: stage_cap x (idx, row)] (dense layout only)
    const size_t dyn_lds = !dense ? 0 : (((size_t)lds_words + 1) & ~(size_t)1) * 4 + (size_t)waves_per_wg * 2 * 4 * (dense_lds ? stage_cap_lds : stage_cap_small);
    // (asked once per module AND dynamic-LDS size: the LDS / hybrid variants' bitmap part changes with the data under one cached
    // plan, and with it the workgroups a CU holds — ADVICE r03)
    if (mod->wgs_per_cu == 0 || mod->wgs_dyn_lds != dyn_lds) {
      int nb = 0;
      if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mod->fn, (int)waves_per_wg * 64, dyn_lds) != hipSuccess || nb < 1) nb = 1;
      mod->wgs_per_cu = nb;
      mod->wgs_dyn_lds = dyn_lds;
    }
    const uint64_t resident = (uint64_t)ctx->num_cus * (uint64_t)mod->wgs_per_cu * waves_per_wg;   // wavefronts the chip runs at once
    const uint64_t tpw_max = (uint64_t)std::max(1, env_int("QHIP_PROBE_TILES_PER_WAVE", 32));
    const uint64_t rounds = std::max<uint64_t>(1, (ntiles + resident * tpw_max - 1) / (resident * tpw_max));
    const uint64_t waves_wanted = std::min<uint64_t>(ntiles, rounds * resident);
    const uint64_t tiles_per_wave = (ntiles + waves_wanted - 1) / waves_wanted;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, ((ntiles + tiles_per_wave - 1) / tiles_per_wave + waves_per_wg - 1) / waves_per_wg);
    const uint64_t nchunks = (uint64_t)grid * waves_per_wg;
    DevBuf tile_tot((nchunks + 1) * 4), tile_nent((nchunks + 1) * 4);
    HProbeLaunch pl;
    pl.table = table; pl.bloom = bloom; pl.count = count; pl.start = start_ptr; pl.rows = rows_ptr;
    pl.ent_slot = ent_slot.as<uint32_t>();
    pl.ent_row = ent_row.as<uint32_t>();
    pl.tile_nent = tile_nent.as<uint32_t>();
    pl.tile_total = tile_tot.as<uint32_t>();
    pl.visited = (mark_in_probe && !want_pairs) ? visited.as<uint32_t>() : nullptr;
    pl.status = dstat + QS_WORDS;
    pl.nslots = nslots; pl.bloom_mask = filter_words - 1;
    pl.n_regions = n_regions; pl.slot_bits = slot_bits; pl.bword_bits = bword_bits;
    pl.tiles_per_wave = (uint32_t)tiles_per_wave;
    if (dense) {
      pl.count = nullptr;
      pl.stage_cap = dense_lds ? stage_cap_lds : stage_cap_small;
      pl.dense_min = (uint64_t)kmin; pl.dense_n = (uint32_t)dense_n; pl.dense_words = dense_words; pl.lds_words = lds_words;
    }
    void* args[] = {&ka, &pl};
    time_mark(ctx, 2);
    QHIP_HIP_CHECK(hipModuleLaunchKernel(mod->fn, grid, 1, 1, waves_per_wg * 64, 1, 1, (unsigned)dyn_lds, s, args, nullptr));
    time_mark(ctx, 3);
    probe_timed = true;
    if (want_pairs) exclusive_scan_u32(tile_tot.as<uint32_t>(), tile_tot.as<uint32_t>(), nchunks, dstat + 2 * QS_WORDS, s);
    if (defer) {
      // no read-back: the status block + total go to a page-locked slot that the consumer's synchronisation checks
      if (!ctx->size_slots) QHIP_HIP_CHECK(hipHostMalloc((void**)&ctx->size_slots, (size_t)kSizeSlots * 32 * 4, hipHostMallocDefault));
      if (ctx->pending_sizes.size() >= (size_t)kSizeSlots) fail(QHIP_HIP_ERROR, "too many joins of deferred size in flight (internal error)");
      uint32_t* slot = ctx->size_slots + (size_t)(ctx->size_slot_next++ % kSizeSlots) * 32;
      auto total_out = std::make_shared<uint64_t>(~0ull);
      ctx->pending_sizes.push_back({slot, size_key, defer_cap, dup_hint, total_out});
      rows_final = total_out;
      M = defer_cap;
      b_idx.alloc((M + 1) * 4);
      p_idx.alloc((M + 1) * 4);
      rows_blk = std::make_shared<DevBuf>(64);   // the output table's device-side row count
      if (mat_key) key_vals = std::make_shared<DevBuf>((M + 1) * 8);
      // pass 2 also pads the index vectors up to the capacity, publishes the status block to `slot` and the total to rows_blk
      launch_join_emit(ent_slot.as<uint32_t>(), ent_row.as<uint32_t>(), tile_nent.as<uint32_t>(), tile_tot.as<uint32_t>(), count, start_ptr, rows_ptr, dense ? (const uint32_t*)table : nullptr, nchunks, tiles_per_wave * kProbeTileRows, b_idx.as<uint32_t>(), p_idx.as<uint32_t>(),
                       nullptr, nullptr, nullptr, (uint32_t)M, dstat, slot, rows_blk->as<uint32_t>(), s, key_vals ? key_vals->as<int64_t>() : nullptr, (uint64_t)kmin);
      deferred_slot = slot;
    } else {
    // ONE read-back: build status (needed only now under speculation), probe status and the pair total
    QHIP_HIP_CHECK(hipMemcpyAsync(st_build, dstat, (2 * QS_WORDS + 1) * 4, hipMemcpyDeviceToHost, s));
    QHIP_HIP_CHECK(sync_stream(s));
    verify_pending_sizes(ctx);   // (a build side of deferred size: did ITS join have room?)
    if (speculate) {
      check_build_status();
      if (max_count > 1) {   // duplicate build keys after all: remember, and run again with the CSR layout
        if (ctx->join_dup_builds.size() > 4096) ctx->join_dup_builds.clear();   // a hint, not a record
        ctx->join_dup_builds.insert(dup_hint);
        return hash_join(ctx, L, R, join_type, lex, nlex, rex, nrex, on_l, on_r, n_on, fex, nfex, froot, fsides, fcols, nfcols, lpred, rpred);
      }
    }
    check_status_words(st);
    M = st[QS_WORDS];
    if (join_type == QHIP_JOIN_INNER && froot < 0) {
      if (ctx->join_size_hints.size() > 4096) ctx->join_size_hints.clear();   // hints, not records
      ctx->join_size_hints[size_key] = M;
    }
    if (want_pairs) {
      b_idx.alloc((M + 1) * 4);
      p_idx.alloc((M + 1) * 4);
      if (pad_right) {
        cnt.alloc((P + 1) * 4);
        pair_off.alloc((P + 1) * 4);
        QHIP_HIP_CHECK(hipMemsetAsync(cnt.ptr, 0, cnt.bytes, s));   // pass 2 only visits matching probe rows
      }
      if (mat_key) key_vals = std::make_shared<DevBuf>((M + 1) * 8);
      launch_join_emit(ent_slot.as<uint32_t>(), ent_row.as<uint32_t>(), tile_nent.as<uint32_t>(), tile_tot.as<uint32_t>(), count, start_ptr, rows_ptr, dense ? (const uint32_t*)table : nullptr, nchunks, tiles_per_wave * kProbeTileRows, b_idx.as<uint32_t>(), p_idx.as<uint32_t>(),
                       pad_right ? pair_off.as<uint32_t>() : nullptr, pad_right ? cnt.as<uint32_t>() : nullptr,
                       mark_in_probe ? visited.as<uint32_t>() : nullptr, 0xFFFFFFFFu, nullptr, nullptr, nullptr, s, key_vals ? key_vals->as<int64_t>() : nullptr, (uint64_t)kmin);
    }
    }
    visited_done = mark_in_probe;
  } else if (speculate) {
    QHIP_HIP_CHECK(hipMemcpyAsync(st_build, dstat, QS_WORDS * 4, hipMemcpyDeviceToHost, s));
    QHIP_HIP_CHECK(sync_stream(s));
    verify_pending_sizes(ctx);
    check_build_status();   // (duplicates do not matter without probe rows)
  }
  if (!probe_timed) { time_mark(ctx, 2); time_mark(ctx, 3); }

  // ---- residual JoinFilter (join/mod.rs:125-154): evaluate over an intermediate batch of the filter's columns, keep true rows
  bool filtered = false;
  if (froot >= 0 && M > 0) {
    qhip_table inter;
    inter.ctx = ctx;
    for (int k = 0; k < nfcols; ++k) {
      const qhip_table* src = fsides[k] == 0 ? L : R;
      if (fcols[k] < 0 || fcols[k] >= (int)src->cols.size()) fail(QHIP_INVALID_ARGUMENT, "join filter column index out of range");
      inter.cols.push_back(gather_column(ctx, src->cols[(size_t)fcols[k]], fsides[k] == 0 ? b_idx.as<uint32_t>() : p_idx.as<uint32_t>(), M, false));
      inter.names.push_back(src->names[(size_t)fcols[k]]);
      inter.nullable.push_back(true);
    }
    inter.num_rows = (int64_t)M;
    inter.batch_offsets = {0, (int64_t)M};
    std::vector<InputCol> fic = input_cols_of(&inter);
    ExprSet fes;
    fes.build(fex, nfex, fic);
    DevBuf mask, wave, sel;
    run_pred_mask(ctx, &inter, fes, fic, froot, mask, wave);
    const uint32_t m2 = select_from_mask(ctx, mask, wave, (int64_t)M, sel);
    DevBuf b2(((uint64_t)m2 + 1) * 4), p2(((uint64_t)m2 + 1) * 4);
    launch_gather_fixed(b_idx.ptr, sel.as<uint32_t>(), b2.ptr, m2, 4, s);
    launch_gather_fixed(p_idx.ptr, sel.as<uint32_t>(), p2.ptr, m2, 4, s);
    QHIP_HIP_CHECK(sync_stream(s));
    b_idx = std::move(b2);
    p_idx = std::move(p2);
    M = m2;
    filtered = true;
  }

  // ---- visited bitmap (when the probe kernel has not marked it already); per-probe-row surviving counts for the
  // Right / Full padding when the residual filter changed them
  DevBuf cnt2;
  if (filtered && pad_right) {
    cnt2.alloc((P + 1) * 4);
    QHIP_HIP_CHECK(hipMemsetAsync(cnt2.ptr, 0, cnt2.bytes, s));
  }
  if ((has_tail && !visited_done) || cnt2.ptr)
    launch_join_mark(b_idx.as<uint32_t>(), p_idx.as<uint32_t>(), M, visited.as<uint32_t>(), cnt2.ptr ? cnt2.as<uint32_t>() : nullptr, s);
  const uint32_t* final_cnt = cnt2.ptr ? cnt2.as<uint32_t>() : cnt.as<uint32_t>();
  DevBuf off2;                                  // exclusive scan of final_cnt (position of a probe row's first pair)
  const uint32_t* final_off = pair_off.as<uint32_t>();
  if (cnt2.ptr) {
    off2.alloc((P + 1) * 4);
    exclusive_scan_u32(final_cnt, off2.as<uint32_t>(), P, nullptr, s);
    final_off = off2.as<uint32_t>();
  }

  // ---- Right / Full: probe rows without a surviving pair appear once with a NULL build index, in probe order
  DevBuf out_off;
  if (pad_right) {
    DevBuf out_cnt((P + 1) * 4), tot(4);
    out_off.alloc((P + 1) * 4);
    launch_join_out_counts(final_cnt, P, out_cnt.as<uint32_t>(), s);
    exclusive_scan_u32(out_cnt.as<uint32_t>(), out_off.as<uint32_t>(), P, tot.as<uint32_t>(), s);
    const uint64_t M2 = P ? read_u32(s, tot.ptr) : 0;
    DevBuf b3((M2 + 1) * 4), p3((M2 + 1) * 4);
    launch_join_adjust_right(b_idx.as<uint32_t>(), final_cnt, final_off, out_off.as<uint32_t>(), P, b3.as<uint32_t>(), p3.as<uint32_t>(), s);
    b_idx = std::move(b3);
    p_idx = std::move(p3);
    M = M2;
    final_off = out_off.as<uint32_t>();
  }
  if (semi_anti) M = 0;   // hash_join.rs:260-262: nothing is emitted while probing

  // ---- tail: unmatched build rows (Left / Full / LeftAnti) or matched ones (LeftSemi), ascending build index
  uint64_t T = 0;
  DevBuf tail_sel;
  if (has_tail && B > 0) {
    DevBuf tmask(((B + 63) / 64) * 8 + 8), twave((((B + 63) / 64) + 1) * 4);
    launch_mask_from_bits(visited.as<uint32_t>(), B, join_type == QHIP_JOIN_LEFT_SEMI ? 1 : 0, tmask.as<uint64_t>(), twave.as<uint32_t>(), s);
    T = select_from_mask(ctx, tmask, twave, (int64_t)B, tail_sel);
  }
  const uint64_t total_rows = M + T;
  if (total_rows >= kNullIdx) fail(QHIP_UNSUPPORTED, "join output of 2^32 - 1 rows or more is not supported");

  // ---- index vectors of the whole output: [pairs | tail]
  auto b_all = std::make_shared<DevBuf>(), p_all = std::make_shared<DevBuf>();
  if (T == 0) {
    *b_all = std::move(b_idx);   // no tail rows: the pair vectors are the output index vectors
    *p_all = std::move(p_idx);
  } else {
    b_all->alloc((total_rows + 1) * 4);
    p_all->alloc((total_rows + 1) * 4);
    if (M) {
      QHIP_HIP_CHECK(hipMemcpyAsync(b_all->ptr, b_idx.ptr, M * 4, hipMemcpyDeviceToDevice, s));
      QHIP_HIP_CHECK(hipMemcpyAsync(p_all->ptr, p_idx.ptr, M * 4, hipMemcpyDeviceToDevice, s));
    }
    QHIP_HIP_CHECK(hipMemcpyAsync(b_all->as<uint32_t>() + M, tail_sel.ptr, T * 4, hipMemcpyDeviceToDevice, s));
    launch_fill_u32(p_all->as<uint32_t>() + M, T, kNullIdx, s);
  }

  // ---- output columns (build_batch_from_indices, utils/batch.rs:18-61): the reference gathers every column of both
  // sides; here the gathers are deferred until a column is read (a parent join / aggregate touches only a few)
  std::unique_ptr<qhip_table> out(new qhip_table());
  out->ctx = ctx;
  const bool left_nullable = pad_right;
  const bool right_nullable = has_tail;
  const bool eager = env_int("QHIP_EAGER_GATHER", 0) != 0;
  auto add_side = [&](const qhip_table* src, const std::shared_ptr<DevBuf>& idx, bool side_nullable) {
    if (eager) for (auto& c : src->cols) out->cols.push_back(gather_column(ctx, c, idx->as<uint32_t>(), total_rows, side_nullable));
    else defer_gather(ctx, src->cols, idx, total_rows, side_nullable, out->cols);
    for (size_t c = 0; c < src->cols.size(); ++c) {
      out->names.push_back(src->names[c]);
      out->nullable.push_back(src->nullable[c] || side_nullable);
    }
  };
  add_side(L, b_all, left_nullable);
  if (!semi_anti) add_side(R, p_all, right_nullable);
  if (key_vals) {
    auto plain = [&](const DevColumn& src) {
      DevColumn c;
      c.type = src.type;
      c.length = (int64_t)total_rows;
      c.values = key_vals;
      c.value_maxabs = src.value_maxabs;
      c.range = src.range;             // (a subset of the source's values: its bounds hold)
      c.range_inherited = true;
      return c;
    };
    const size_t rc = (size_t)rex[on_r[0]].column;
    out->cols[L->cols.size() + rc] = plain(R->cols[rc]);
    // ... and the build side's key column holds the same values row for row
    if (lex[on_l[0]].kind == QHIP_EXPR_COLUMN && lex[on_l[0]].column >= 0 && lex[on_l[0]].column < (int)L->cols.size() &&
        L->cols[(size_t)lex[on_l[0]].column].type.id == QHIP_INT64)
      out->cols[(size_t)lex[on_l[0]].column] = plain(L->cols[(size_t)lex[on_l[0]].column]);
  }
  out->num_rows = (int64_t)total_rows;
  if (deferred_slot) {   // total_rows is the capacity; the count is what pass 2 left in rows_blk
    out->rows_blk = rows_blk;
    out->rows_dev = rows_blk->as<uint32_t>();
    out->rows_host = deferred_slot + 2 * QS_WORDS;
    out->rows_final = rows_final;
  }

  // ---- output batches: one per non-empty probe batch (hash_join.rs:363-372), then the tail batch
  out->batch_offsets.clear();
  out->batch_offsets.push_back(0);
  if (!semi_anti && R->num_batches() > 0 && M > 0) {
    const size_t nb1 = R->offsets().size();
    // first output row of every probe batch, computed on the device and LEFT there (qhip_table::offsets() fetches it when
    // somebody asks: a download, a Filter / Limit / probe side above; a parent's build side or an aggregate never does)
    auto pend = std::make_shared<PendingOffsets>();
    pend->n = nb1;
    pend->skip_empty = true;
    pend->tail = has_tail;
    pend->total_rows = (int64_t)total_rows;
    if (pad_right) {
      pend->pos = std::make_shared<DevBuf>(nb1 * 4);
      launch_lookup_u32(final_off, R->device_offsets(), (uint32_t)nb1, P, (uint32_t)M, pend->pos->as<uint32_t>(), s);
    } else {   // the pairs' probe rows ascend: a binary search per batch boundary, run when somebody asks (PendingOffsets)
      pend->search_in = p_all;
      pend->search_m = M;
      // (the probe side's batch boundaries go along on the host, or as the device copy another operator already made:
      // uploading them here would be a host wait per join for something a parent join or an aggregate never asks for)
      if (R->offsets_dev) pend->bounds = R->offsets_dev;
      else pend->bounds_host.assign(R->offsets().begin(), R->offsets().end());
    }
    out->batch_offsets.clear();
    out->pending_offsets = pend;
    if (env_int("QHIP_EAGER_OFFSETS", 0) != 0 && !deferred_slot) (void)out->offsets();
  } else if (has_tail) {
    out->batch_offsets.push_back((int64_t)total_rows);   // always present, possibly empty (hash_join.rs:374-381)
  }
  time_mark(ctx, 1);
  ctx->stats_timing_pending = ctx->timing ? 2 : 0;   // total = ev0..ev1, probe = ev2..ev3, read by qhip_ctx_last_stats
  ctx->stats.rows_in = (int64_t)P;
  ctx->stats.rows_out = (int64_t)total_rows;
  ctx->stats.groups = (int64_t)M;
  ctx->stats.table_capacity = dense ? (int64_t)dense_n : nslots;
  snprintf(ctx->stats.main_kernel_name, sizeof ctx->stats.main_kernel_name, "%s", probe_name.c_str());
  // bytes of column data the probe kernel / the build's key evaluation read per row (roofline figures)
  auto bytes_per_row = [&](const qhip_table* t, const KernelBindings& b) {
    double sum = 0;
    for (size_t k = 0; k < b.cols.size(); ++k) {
      const int c = b.cols[k];
      const DevColumn& dc = t->cols[(size_t)c];
      const int w = dtype_width(dc.type);
      if (w > 0) sum += k < b.narrow.size() && b.narrow[k] ? (int)b.narrow[k] : w;   // (a key column read as its 4-byte narrow copy)
      else if (dc.type.id == QHIP_BOOL) sum += 0.125;
      else if (dc.type.id == QHIP_UTF8) sum += 4.0 + (t->num_rows > 0 ? (double)dc.data_bytes / (double)t->num_rows : 0.0);
      if (dc.null_count > 0) sum += 0.125;
    }
    return sum;
  };
  ctx->stats.bytes_per_row_read = bytes_per_row(R, rkp.bind);
  ctx->stats.build_bytes_per_row = bytes_per_row(L, lkp.bind);
  ctx->stats.build_rows = (int64_t)B;
  return out.release();
}

}  // namespace

extern "C" int qhip_hash_join_execute(qhip_ctx* ctx, const qhip_table* left, const qhip_table* right, int32_t join_type,
                                      const qhip_expr* left_exprs, int32_t n_left_exprs, const qhip_expr* right_exprs, int32_t n_right_exprs,
                                      const int32_t* on_left, const int32_t* on_right, int32_t n_on, const qhip_expr* filter_exprs,
                                      int32_t n_filter_exprs, int32_t filter_root, const int32_t* filter_sides, const int32_t* filter_cols,
                                      int32_t n_filter_cols, int32_t left_scan_filter_root, int32_t right_scan_filter_root, qhip_table** out) {
  if (!ctx || !left || !right || !out) return QHIP_INVALID_ARGUMENT;
  *out = nullptr;
  return guarded(ctx, [&] {
    try {
      *out = hash_join(ctx, left, right, join_type, left_exprs, n_left_exprs, right_exprs, n_right_exprs, on_left, on_right, n_on, filter_exprs,
                       n_filter_exprs, filter_root, filter_sides, filter_cols, n_filter_cols, left_scan_filter_root, right_scan_filter_root);
    } catch (...) {
      ctx->pending_sizes.clear();   // (joins of deferred size below: whatever they left is void with this call's input)
      throw;
    }
  });
}
