// common.hpp — host-side core types of libqhip: context, errors, device buffers, tables.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/qhip.h"

namespace qhip {

typedef unsigned __int128 u128;
typedef __int128 i128;

// ---------------------------------------------------------------- errors
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] inline void fail(int code, const std::string& msg) { throw Error(code, msg); }

#define QHIP_HIP_CHECK(expr)                                                                           \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      ::qhip::fail(_e == hipErrorOutOfMemory ? QHIP_OUT_OF_MEMORY : QHIP_HIP_ERROR,                    \
                   std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
    }                                                                                                  \
  } while (0)

// ---------------------------------------------------------------- dtypes
struct DType {
  int id = QHIP_NULL;
  int precision = 0;
  int scale = 0;
  DType() {}
  DType(int i, int p = 0, int s = 0) : id(i), precision(p), scale(s) {}
  DType(const qhip_dtype& d) : id(d.id), precision(d.precision), scale(d.scale) {}
  bool operator==(const DType& o) const {
    return id == o.id && (id != QHIP_DECIMAL128 || (precision == o.precision && scale == o.scale));
  }
  bool operator!=(const DType& o) const { return !(*this == o); }
  qhip_dtype pod() const { qhip_dtype d; d.id = id; d.precision = precision; d.scale = scale; return d; }
};
std::string dtype_name(const DType& t);       // arrow-rs Display spelling: Int64, Decimal128(15, 2), Utf8 ...
int dtype_width(const DType& t);              // bytes per value of a fixed-width type; 0 for Utf8/Bool/Null
bool dtype_is_integer(const DType& t);
bool dtype_is_signed(const DType& t);
bool dtype_is_float(const DType& t);
std::string dtype_to_format(const DType& t);  // Arrow C Data Interface format string
DType dtype_from_format(const char* fmt);     // throws QHIP_UNSUPPORTED

// ---------------------------------------------------------------- device memory
struct Ctx;
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  size_t cap = 0;     // size class actually allocated (caching allocator, ctx.cpp)
  int device = 0;
  // a VIEW of `bytes` bytes inside another buffer (the parts of a partitioned column are slices of one allocation, exchange.cpp):
  // the view keeps the owner alive and returns nothing to the pool itself
  std::shared_ptr<DevBuf> owner;
  DevBuf() {}
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const std::shared_ptr<DevBuf>& whole, size_t offset, size_t n) : ptr((uint8_t*)whole->ptr + offset), bytes(n), cap(0), device(whole->device), owner(whole) {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : ptr(o.ptr), bytes(o.bytes), cap(o.cap), device(o.device), owner(std::move(o.owner)) { o.ptr = nullptr; o.bytes = 0; o.cap = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); ptr = o.ptr; bytes = o.bytes; cap = o.cap; device = o.device; owner = std::move(o.owner); o.ptr = nullptr; o.bytes = 0; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t n);   // n == 0 still yields a valid (tiny) allocation so kernels never see nullptr
  void release();
  template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
};

// Every host wait on the device goes through these two (counted: qhip_ctx_sync_count, the "host round trips" of a plan)
uint64_t sync_counter();
void note_sync();   // counts; QHIP_SYNC_TRACE=1 prints who waits (ctx.cpp)
void trace_point(const char* what);   // QHIP_TRACE=2: absolute host time of a named point (where does the time between two queries go)
// (polling hipStreamQuery / hipEventQuery instead of blocking was measured: no gain for Q3, Q1's step 0.71 -> 0.86 ms)
inline hipError_t sync_stream(hipStream_t s) { note_sync(); return hipStreamSynchronize(s); }
inline hipError_t sync_event(hipEvent_t e) { note_sync(); return hipEventSynchronize(e); }

// Copy ordered after everything queued on `s` (the context's stream is non-blocking: a plain hipMemcpy on the null stream
// would NOT wait for it), and complete on return.
inline void copy_sync(hipStream_t s, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  if (n) QHIP_HIP_CHECK(hipMemcpyAsync(dst, src, n, kind, s));
  QHIP_HIP_CHECK(sync_stream(s));
}

struct ColRange {
  bool known = false; int64_t min = 0, max = 0;
  // (round 4) ... and, for readers that reach the column through a deferred gather's COPY of it (DeferredGather::src — an
  // aggregate over a join output reads lineitem's prices through the join's index vector and never sees the table's own column
  // object): the magnitude bound and the 4- / 8-byte narrow copy of a Decimal128 column, made at the second such read, valid for
  // exactly the buffer and length they were made from. Same meaning as DevColumn::value_maxabs / ::narrow.
  uint64_t maxabs = 0;
  int reads = 0;
  std::shared_ptr<DevBuf> narrow_buf; int narrow_bytes = 0; const void* narrow_src = nullptr; int64_t narrow_rows = 0;
  // ... and the RECORD copy (relops.cpp ensure_indirect_records): this column's 4- / 8-byte values (narrow where it has a narrow
  // copy) as field `rec_offset` of `rec_stride`-byte records shared with other columns of the same table, so that a kernel
  // that reads several of them through ONE index vector touches one 64-byte line per row instead of one per column. Valid for
  // exactly the buffer and length it was made from; once made it is kept (a column belongs to at most one record).
  std::shared_ptr<DevBuf> rec_buf; int rec_stride = 0, rec_offset = 0, rec_width = 0; const void* rec_src = nullptr; int64_t rec_rows = 0;
  int rec_reads = 0;
};
// One column of a device table, concatenated over all batches, Arrow layout.
struct DevColumn {
  DType type;
  int64_t length = 0;
  int64_t null_count = 0;
  std::shared_ptr<DevBuf> values;    // fixed-width values | int32 offsets (length+1) | bit-packed booleans
  std::shared_ptr<DevBuf> validity;  // bitmap, present iff null_count > 0
  std::shared_ptr<DevBuf> data;      // utf8 bytes
  int64_t data_bytes = 0;
  mutable int32_t utf8_max_len = -1;   // cached longest value (bytes), computed on first use as a key column
  // cached upper bound of |value| of an Int64 / Decimal128 column (0 = not computed, ~0 = some value needs more than 63
  // bits), computed on first use as an aggregate argument of a big input: lets the generated code multiply and accumulate
  // in 32 / 64 bits where the data allows (an upper bound stays one under gathering, like utf8_max_len)
  mutable uint64_t value_maxabs = 0;
  // NARROW COPY of a Decimal128 column whose every value fits 32 / 64 bits (value_maxabs < 2^31 / 2^63): the same values as
  // 4- or 8-byte integers, built once per column when its statistics are collected (relops.cpp ensure_value_bounds) and read
  // by the generated aggregate kernels INSTEAD of the 16-byte values — TPC-H's quantities, prices, discounts and taxes all
  // fit 32 bits, so Q1 streams 22 instead of 70 bytes per row. The 16-byte Arrow layout stays the column's canonical form
  // (every other operator, the exports and the exchange read `values`); the copy is valid for exactly the buffer and length
  // it was made from (src / rows are checked before use).
  struct NarrowCopy { std::shared_ptr<DevBuf> buf; int bytes = 0; const void* src = nullptr; int64_t rows = 0; };
  mutable std::shared_ptr<NarrowCopy> narrow;
  // ... made the SECOND time a big operator streams the column (big_reads): a copy costs a pass over the column, which only a
  // column that lives on — a resident table's — earns back; the columns of an intermediate result (an exchange's received
  // part, a materialised join output) are read once and never get one. QHIP_NARROW_FIRST_USE=1: at the first read (tests).
  mutable int big_reads = 0;
  // Value range [min, max] of an integer-like column, computed on first use as a hash join's build key (one reduction +
  // one read-back; decides whether the join addresses its table by the key itself: join.cpp, dense layout). The object is
  // SHARED by every copy of the column (a base table's column and the `src` of the deferred gathers made from it), so the
  // statistic is computed once per table, not once per query. A gathered column INHERITS its source's object
  // (range_inherited: a superset's bounds hold for the subset; the subset never writes its own, narrower, bounds into it).
  mutable std::shared_ptr<struct ColRange> range = std::make_shared<struct ColRange>();
  mutable bool range_inherited = false;
  // A join / filter output column may be DEFERRED: (source column, row index vector), gathered only when somebody reads
  // it (an expression that references it, an export, an exchange). The reference gathers every column of every join
  // output (utils/batch.rs:18-61) although most are never looked at downstream (Q3: c_mktsegment, o_custkey, ...).
  // While deferred, values / validity / data are empty and null_count is a may-have-nulls flag (0 / 1).
  std::shared_ptr<struct DeferredGather> deferred;
  // ... or not UPLOADED yet: a lazily ingested table (qhip_table_from_arrow_lazy) moves a column host -> HBM when it is
  // first read, so columns no query touches never cross PCIe (the projection pushdown the reference's Scan lacks).
  std::shared_ptr<struct DeferredUpload> pending_upload;
  int64_t resident_bytes() const;
};
struct DeferredUpload {
  std::shared_ptr<void> host_col;       // ... or a HostColumn an operator assembled (small results stay on the host until a
                                        // device operator reads them; the reference's results are host batches anyway)
  std::shared_ptr<void> host;           // HostBatches (table.cpp): the Arrow arrays, kept alive
  std::string format;                   // Arrow C format string of the column
  int64_t column = 0;
  std::vector<int64_t> batch_offsets;
  bool done = false;
  DevColumn result;
};
struct DeferredGather {
  DevColumn src;                  // never itself deferred (index vectors are composed instead)
  std::shared_ptr<DevBuf> idx;    // u32 row numbers into src; kNullIdx -> NULL when idx_may_be_null
  uint64_t m = 0;
  bool idx_may_be_null = false;
  bool done = false;
  DevColumn result;
};

}  // namespace qhip

// C handle types
namespace qhip {
// Batch boundaries an operator computed on the device and nobody has read yet (qhip_table::offsets() reads them)
struct PendingOffsets {
  std::shared_ptr<DevBuf> pos;     // n uint32 output positions: where input batch b's first output row is
  size_t n = 0;
  bool skip_empty = false;         // hash join (hash_join.rs:363-372): only non-empty probe batches produce a batch
  bool tail = false;               // ... and a final batch (unmatched / semi rows), possibly empty
  int64_t total_rows = 0;
  // ... or not even searched for yet (an Inner / Left join's pairs ascend by probe row: batch b starts at the first of the
  // search_m pairs whose probe row is >= bounds[b]): the search runs when somebody asks (a parent's build side or an
  // aggregate never does — one launch less per join)
  std::shared_ptr<DevBuf> search_in, bounds;
  std::vector<uint64_t> bounds_host;   // the boundary rows when `bounds` has not been uploaded (nobody may ever ask)
  uint64_t search_m = 0;
};
}  // namespace qhip
struct qhip_table {
  qhip::Ctx* ctx = nullptr;
  std::vector<std::string> names;
  std::vector<bool> nullable;             // schema-level nullability flag
  std::vector<qhip::DevColumn> cols;
  int64_t num_rows = 0;
  // size = num_batches + 1; batch b = rows [off[b], off[b+1]). WRITE through this member when building a table; READ an
  // input's boundaries through offsets(): a hash join leaves them on the device (pending_offsets) until somebody asks —
  // a parent join's build side or an aggregate never does, which saves them a host round trip (Q3: two per query).
  mutable std::vector<int64_t> batch_offsets;
  mutable std::shared_ptr<qhip::PendingOffsets> pending_offsets;
  const std::vector<int64_t>& offsets() const;      // relops.cpp
  const uint64_t* device_offsets() const;           // relops.cpp: offsets() as uint64 on the device, uploaded once per table
  mutable std::shared_ptr<qhip::DevBuf> offsets_dev;
  int64_t num_batches() const { return (int64_t)offsets().size() - 1; }
  bool no_batches() const { return !pending_offsets && batch_offsets.size() <= 1; }   // the empty Vec<RecordBatch>
  // Deferred sizing (hash join, join.cpp): the join did not wait for its pair total. num_rows is then a CAPACITY, the real
  // count is *rows_dev on the device (*rows_host once the stream has been synchronised) and rows [count, num_rows) repeat
  // row 0 of both sides (valid to gather, never counted). HashAggregate and a hash join's build side read such a table as
  // it is (their kernels stop at *rows_dev); every other reader calls settle_rows() first.
  mutable std::shared_ptr<qhip::DevBuf> rows_blk;   // owns the device-side count
  mutable const uint32_t* rows_dev = nullptr;       // -> the pair total the join's pass 2 left there
  mutable const uint32_t* rows_host = nullptr;      // -> its page-locked copy (a ring slot: valid until verify_pending_sizes has read it)
  mutable std::shared_ptr<uint64_t> rows_final;     // the verified total, written by verify_pending_sizes (~0 = not yet)
  // the exact row count of a table of deferred size; call only after a stream synchronisation + verify_pending_sizes
  int64_t deferred_count() const {
    const uint64_t m = (rows_final && *rows_final != ~0ull) ? *rows_final : (rows_host ? (uint64_t)*rows_host : (uint64_t)num_rows);
    return (int64_t)(m < (uint64_t)num_rows ? m : (uint64_t)num_rows);
  }
};
namespace qhip {
// Make num_rows exact: wait for the stream, check the deferred joins' status words (verify_pending_sizes; may throw
// QHIP_RETRY) and shrink the table to its real row count. No-op for ordinary tables.
void settle_rows(const qhip_table* t);
}

namespace qhip {

struct Module;  // jit.cpp

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // call begin / end, dominant kernel begin / end, end of the first phase (partition: pass 1)
  std::string last_error;
  std::string device_name;
  int num_cus = 256;
  qhip_exec_stats stats;
  // HIP events around an operator's phases (qhip_exec_stats timings). OFF by default: every event record is a packet of its
  // own on the stream (~5 us of stream time; Q3 recorded 10 per query) — qhip_ctx_set_timing / QHIP_TIMING=1 switch them on
  bool timing = false;
  mutable int stats_timing_pending = 0;   // 1: total = ev0..ev1; 2: also main kernel = ev2..ev3 (3: and build_ms = ev0..ev4) — read when the stats are asked for
  std::unordered_map<std::string, std::shared_ptr<Module>> modules;  // kernel cache keyed by generated source
  DevBuf status;       // QS_WORDS u32 status words
  // 128-byte blocks of zeros for status words / counters of ONE operator call (zeroed_block): handed out in turn from a ring
  // that is cleared in bulk every 512 blocks — a memset launch per call costs ~5 us of stream time, more than many of the
  // kernels it precedes. Single stream: whoever held a block 512 hand-outs ago has long finished.
  DevBuf zero_ring;
  size_t zero_next = 0;
  std::unordered_set<uint64_t> join_dup_builds;   // build sides (key policy x row count) seen with duplicate keys: no speculation
  // Deferred sizing. A hash join remembers how many pairs it produced (keyed by its expressions, type and probe rows, NOT
  // by the data). The next time the same join runs under a consumer that can read a device-side row count
  // (allow_deferred_sizes > 0: HashAggregate's input, a hash join's build side) it does not wait for its pair total: the
  // output is allocated for that many pairs plus headroom, the count stays on the device, and the status block + total
  // are copied to a page-locked slot (PendingSize) that the plan's next natural synchronisation checks
  // (verify_pending_sizes): more pairs than the capacity / duplicate build keys -> the hint is dropped and the consumer
  // returns QHIP_RETRY (its input is re-executed, this time waiting).
  std::unordered_map<uint64_t, uint64_t> join_size_hints;
  // groups an aggregate (identified by its expressions, whatever its input's layout) produced the last time it ran, and the words
  // of its table slots: decide whether a mid-sized input is ordered by key hash first (agg.cpp, AggParts)
  std::unordered_map<uint64_t, uint32_t> agg_group_hints;
  std::unordered_map<uint64_t, int> agg_slot_words;
  // (total_out: the output table's own host word — the page-locked slot is part of a ring and is reused by later joins, so
  // the verified total is copied where the table can still find it however long it stays unsettled)
  struct PendingSize { uint32_t* slot; uint64_t key; uint64_t capacity; uint64_t dup_hint; std::shared_ptr<uint64_t> total_out; };
  std::vector<PendingSize> pending_sizes;
  uint32_t* size_slots = nullptr;   // page-locked ring: kSizeSlots x 32 words
  int size_slot_next = 0;
  int allow_deferred_sizes = 0;
  void* pinned = nullptr;            // small page-locked scratch for status / result read-backs (truly asynchronous D2H)
  size_t pinned_bytes = 0;
  // two page-locked 8 MB slots through which uploads of MANY SMALL batches are coalesced (table.cpp), made on first use
  void* up_slot[2] = {nullptr, nullptr};
  hipEvent_t up_ev[2] = {nullptr, nullptr};
  std::string cache_dir;
  std::unordered_map<std::string, std::shared_ptr<void>> plan_cache;   // lowered plans keyed by their POD description
  // host-assembled columns whose upload is in flight (agg.cpp upload_host_column): kept alive until an event behind their
  // copies has passed, instead of a host wait per column (an aggregate's few result rows feeding the next device operator)
  struct HostKeep { hipEvent_t ev = nullptr; std::shared_ptr<void> p; };
  HostKeep host_keep[32];
  size_t host_keep_next = 0;
};

// wraps a C entry point: runs f(), converts exceptions to status codes + last_error
template <class F> int guarded(qhip_ctx* c, F&& f);

}  // namespace qhip

struct qhip_ctx : qhip::Ctx {};

namespace qhip {
void set_global_error(const std::string& m);
template <class F> int guarded(qhip_ctx* c, F&& f) {
  try {
    f();
    return QHIP_OK;
  } catch (const Error& e) {
    if (c) c->last_error = e.what(); else set_global_error(e.what());
    return e.code;
  } catch (const std::bad_alloc&) {
    if (c) c->last_error = "host allocation failed"; else set_global_error("host allocation failed");
    return QHIP_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    if (c) c->last_error = e.what(); else set_global_error(e.what());
    return QHIP_INVALID_ARGUMENT;
  }
}
constexpr int kSizeSlots = 64;
// After a stream synchronisation: check what the deferred-size joins left behind. Throws QHIP_RETRY (and forgets the
// hint) when a join produced more pairs than it had room for or met duplicate build keys — what was computed from its
// output is then garbage and the consumer's input runs again; key / filter errors surface as they would have in the join.
void verify_pending_sizes(Ctx* ctx);
inline void time_mark(Ctx* ctx, int k) { if (ctx->timing) (void)hipEventRecord(ctx->ev[k], ctx->stream); }   // ev[k], only when timings are wanted
uint32_t* zeroed_block(Ctx* ctx, int n = 1);   // n x 32 contiguous zeroed u32 words, valid for the current operator call (ctx.cpp)
inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace qhip
