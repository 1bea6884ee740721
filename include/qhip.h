/*
 * qhip.h — C ABI of libqhip.so, the MI355X (gfx950) execution backend for the
 * filter / hash-aggregate / hash-join path of holicc/qurious.
 *
 * This header is the drop-in boundary. Every entry point replaces one reference
 * operator (cited as file:line under /root/reference/qurious/src) and is what a
 * Rust `extern "C"` block would bind. Only plain pointers,
 * sizes and PODs cross the boundary; column data crosses as Arrow C Data
 * Interface structs (arrow-rs side: arrow::ffi::{to_ffi, from_ffi}).
 *
 * Conventions
 *  - every function returns a qhip_status (0 = ok); qhip_last_error(ctx) gives the
 *    message. The reference's Result<T, Error> (error.rs:20-31) maps as
 *    non-zero -> Error::InternalError(msg); QHIP_UNSUPPORTED lets a caller fall
 *    back to the CPU node. The library never aborts the process.
 *  - a context is used by one thread at a time (the reference executor is
 *    single-threaded, physical/plan/mod.rs:25-29); all calls are synchronous.
 *  - a qhip_table is the device-resident equivalent of the Vec<RecordBatch> that
 *    PhysicalPlan::execute returns (physical/plan/mod.rs:27): columns live
 *    concatenated in HBM, the batch boundaries are remembered so that per-batch
 *    semantics (Filter: one output batch per input batch; HashJoinExec: one
 *    output batch per non-empty probe batch) are reproduced on download.
 */
#ifndef QHIP_H
#define QHIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- Arrow C Data Interface */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

/* ---------------------------------------------------------------- status */
typedef enum qhip_status {
  QHIP_OK = 0,
  QHIP_INVALID_ARGUMENT = 1, /* malformed descriptor / type mismatch (reference: arrow_err!/internal_err!) */
  QHIP_UNSUPPORTED = 2,      /* valid plan the backend does not accelerate: fall back to the CPU node */
  QHIP_HIP_ERROR = 3,        /* HIP runtime / hiprtc failure, or no usable gfx950 device */
  QHIP_OUT_OF_MEMORY = 4,
  QHIP_EXEC_ERROR = 5,       /* data-dependent failure the reference also reports (divide by zero, cast overflow, AVG overflow) */
  QHIP_RCCL_ERROR = 6,
  QHIP_RETRY = 7             /* a hash join below ran without waiting for its output size (qhip_ctx_allow_deferred_sizes) and the
                              * room it had assumed did not hold: the hint has been dropped, execute the input plan again */
} qhip_status;

typedef struct qhip_ctx qhip_ctx;
typedef struct qhip_table qhip_table;
typedef struct qhip_join qhip_join;

/* ---------------------------------------------------------------- types */
/* arrow DataType subset reachable on the path (utils/array.rs:190-210, aggregate/mod.rs:87-116) */
typedef enum qhip_type_id {
  QHIP_NULL = 0,
  QHIP_BOOL = 1,
  QHIP_INT8 = 2, QHIP_INT16 = 3, QHIP_INT32 = 4, QHIP_INT64 = 5,
  QHIP_UINT8 = 6, QHIP_UINT16 = 7, QHIP_UINT32 = 8, QHIP_UINT64 = 9,
  QHIP_FLOAT32 = 10, QHIP_FLOAT64 = 11,
  QHIP_DATE32 = 12, QHIP_DATE64 = 13,
  QHIP_DECIMAL128 = 14,
  QHIP_UTF8 = 15,
  /* Time32(Second | Millisecond) = i32, Time64(Microsecond | Nanosecond) = i64: key / comparison / MIN-MAX types of
     create_hashes (utils/array.rs:199-202); no arithmetic and no casts are defined on them here */
  QHIP_TIME32_S = 16, QHIP_TIME32_MS = 17, QHIP_TIME64_US = 18, QHIP_TIME64_NS = 19,
  /* Timestamp(Second | Millisecond | Microsecond | Nanosecond, None) = i64: comparison / MIN-MAX / sort-key types
     (aggregate/mod.rs:108-111); NOT hash keys — the reference's create_hashes rejects them (utils/array.rs:205) and so does
     this library, with the reference's error text. A timezone-qualified timestamp column is QHIP_UNSUPPORTED at upload. */
  QHIP_TIMESTAMP_S = 20, QHIP_TIMESTAMP_MS = 21, QHIP_TIMESTAMP_US = 22, QHIP_TIMESTAMP_NS = 23
} qhip_type_id;

typedef struct qhip_dtype {
  int32_t id;        /* qhip_type_id */
  int32_t precision; /* Decimal128 only */
  int32_t scale;     /* Decimal128 only */
} qhip_dtype;

/* datatypes/operator.rs:4-20 — same order */
typedef enum qhip_operator {
  QHIP_OP_EQ = 0, QHIP_OP_NOTEQ, QHIP_OP_GT, QHIP_OP_GTEQ, QHIP_OP_LT, QHIP_OP_LTEQ,
  QHIP_OP_AND, QHIP_OP_OR,
  QHIP_OP_ADD, QHIP_OP_SUB, QHIP_OP_MUL, QHIP_OP_DIV, QHIP_OP_MOD
} qhip_operator;

/* physical/expr/{column,literal,binary,cast,is_null,is_not_null,negative,case,like}.rs */
typedef enum qhip_expr_kind {
  QHIP_EXPR_COLUMN = 0,   /* column.rs:24-34   : `column` = index into the input schema */
  QHIP_EXPR_LITERAL = 1,  /* literal.rs:20-22  : ScalarValue broadcast; value in lit_* by dtype */
  QHIP_EXPR_BINARY = 2,   /* binary.rs:31-70   : `op`, children `left`,`right` */
  QHIP_EXPR_CAST = 3,     /* cast.rs:33-37     : child `left`, target `dtype`, safe=false */
  QHIP_EXPR_IS_NULL = 4,  /* is_null.rs        : child `left` */
  QHIP_EXPR_IS_NOT_NULL = 5,
  QHIP_EXPR_NEGATIVE = 6, /* negative.rs       : child `left` */
  QHIP_EXPR_IF = 7,       /* case.rs:33-48     : zip(mask = `left`, truthy = `right`, falsy = `third`); CASE WHEN c1 THEN v1
                           *                      WHEN c2 THEN v2 ELSE e END is lowered as IF(c1, v1, IF(c2, v2, e)) */
  QHIP_EXPR_LIKE = 8      /* like.rs:28-43     : `left` LIKE `right` (a Utf8 literal pattern: % _ and \ escapes);
                           *                      op != 0 = NOT LIKE */
} qhip_expr_kind;

/* One node of an expression tree stored as a flat array; children are indices
 * into the same array (-1 = none). Literals: integers/dates/bools in lit_lo
 * (sign-extended), Decimal128 in (lit_hi:lit_lo), floats in lit_f64, Utf8 in
 * lit_str/lit_len; lit_is_null marks a typed NULL literal. */
typedef struct qhip_expr {
  int32_t kind;
  int32_t op;
  int32_t column;
  int32_t left;
  int32_t right;
  int32_t third;        /* QHIP_EXPR_IF only (-1 otherwise) */
  qhip_dtype dtype;
  int32_t lit_is_null;
  uint64_t lit_lo;
  int64_t lit_hi;
  double lit_f64;
  const char* lit_str;
  int64_t lit_len;
} qhip_expr;

/* physical/expr/aggregate/{sum,avg,count,min,max}.rs */
typedef enum qhip_agg_kind {
  QHIP_AGG_SUM = 0, QHIP_AGG_AVG = 1, QHIP_AGG_COUNT = 2, QHIP_AGG_MIN = 3, QHIP_AGG_MAX = 4
} qhip_agg_kind;

typedef struct qhip_agg {
  int32_t kind;           /* qhip_agg_kind */
  int32_t expr;           /* root index of the argument expression in the expr array */
  qhip_dtype return_type; /* SumAggregateExpr::return_type (sum.rs:15), AvgAggregateExpr::return_type (avg.rs:19), Min/Max return_type */
} qhip_agg;

/* common/join_type.rs:4-11 — same order */
typedef enum qhip_join_type {
  QHIP_JOIN_LEFT = 0, QHIP_JOIN_RIGHT, QHIP_JOIN_INNER, QHIP_JOIN_FULL, QHIP_JOIN_LEFT_SEMI, QHIP_JOIN_LEFT_ANTI
} qhip_join_type;

/* per-call execution statistics: device time of the dominant kernel, measured with
 * HIP events on the stream it was launched on (bench.py roofline.achieved) */
typedef struct qhip_exec_stats {
  double main_kernel_ms;     /* dominant kernel of the last call (fused filter+aggregate, or join probe) */
  double total_device_ms;    /* first launch .. last launch of the call */
  double jit_ms;             /* hiprtc time spent by this call (0 when the kernel cache hit) */
  int64_t rows_in;           /* rows scanned by the dominant kernel */
  int64_t rows_out;
  int64_t groups;            /* groups / matched pairs */
  int64_t table_capacity;    /* hash-table slots used by the call */
  int32_t retries;           /* hash-table growth retries */
  int32_t lds_table_slots;   /* LDS-staged table slots per workgroup */
  char main_kernel_name[64];
  double lds_occupancy;      /* aggregate: mean fraction of a workgroup's LDS table in use when it is merged into the HBM
                              * table (BASELINE configs[4] asks for it); collected only when QHIP_AGG_STATS=1 is set in the
                              * environment (one extra atomic per workgroup), -1 otherwise */
  double hbm_table_load;     /* groups / table_capacity of the attempt that succeeded */
  int32_t lds_spilled;       /* 1 = some keys bypassed the LDS table (it was full around their home slot) */
  int32_t workgroups;        /* grid size of the dominant kernel */
  double bytes_per_row_read; /* aggregate: bytes of column data the fused kernel reads per input row (value, offset and data
                              * buffers of the columns its expressions reference; the offsets of a Utf8 column whose every
                              * value is 1 byte long are not read) — what a roofline figure must be computed from.
                              * Hash join: bytes the probe kernel reads per PROBE row (key + fused-filter columns). */
  double build_ms;           /* hash join: key evaluation + table build, first launch .. probe kernel (main_kernel_ms is the
                              * probe kernel alone, total_device_ms the whole call incl. pair emission) */
  int64_t build_rows;        /* hash join: rows of the build side */
  double build_bytes_per_row;/* hash join: bytes of build-side columns its key / fused-filter expressions read per row */
} qhip_exec_stats;

/* ---------------------------------------------------------------- context */
/* device_index < 0: current HIP device. Fails with QHIP_HIP_ERROR when no gfx950
 * device is visible — there is no CPU fallback behind this library. */
int qhip_ctx_create(int device_index, qhip_ctx** out);
void qhip_ctx_destroy(qhip_ctx* ctx);
const char* qhip_last_error(const qhip_ctx* ctx);  /* ctx may be NULL: last create error */
const char* qhip_version(void);
/* 1 if a HIP device is visible to this process (does not create a context) */
int qhip_device_available(void);
/* Operators return as soon as their result is ordered on the context's stream; wait for it explicitly before handing raw
 * device pointers (qhip_table_column_buffer) to another stream or library. Downloads (qhip_table_to_arrow) wait by themselves. */
int qhip_ctx_synchronize(qhip_ctx* ctx);
int qhip_ctx_last_stats(const qhip_ctx* ctx, qhip_exec_stats* out);
/* Host waits on the device (stream / event synchronisations) this process has made through the library so far: the
 * "host round trips" a plan costs = the difference around its execution. */
uint64_t qhip_ctx_sync_count(const qhip_ctx* ctx);
/* Timings in qhip_exec_stats (main_kernel_ms, total_device_ms, build_ms) come from HIP events recorded around an operator's
 * phases. Every event record is a packet of its own on the stream (~5 us of stream time each; a two-join query recorded
 * ten), so they are OFF by default (the fields then read 0): switch them on for instrumented runs (or QHIP_TIMING=1). */
int qhip_ctx_set_timing(qhip_ctx* ctx, int32_t on);
/* Deferred sizing. A hash join normally waits for its pair total to size its output (one host round trip per join). It
 * also remembers, per join (expressions, join type, probe rows — not the data), how many pairs it produced. While
 * deferred sizes are allowed, an Inner join with such a hint does NOT wait: its output is allocated for the remembered
 * count plus headroom, the real count stays on the device (the kernels of HashAggregate and of a hash join's build side
 * read it there) and the join's status words + total go to page-locked memory, checked by the plan's next natural
 * synchronisation (the aggregate / join above). More pairs than the room, or duplicate build keys: the hint is dropped and
 * THAT call returns QHIP_RETRY — discard its input and execute the input plan again (it will wait this time). Because
 * the check happens in the CONSUMER's call, allow it only around the execution of a child whose output goes straight
 * into qhip_hash_aggregate_execute (input) or qhip_hash_join_execute (left / build side) with nothing executed in
 * between: delta = +1 before the child, -1 after it (the host mirrors do this); delta = 0 resets (after an error).
 * Any other consumer of such a table (Filter, Sort, export, ...) first waits and makes the row count exact by itself.
 * No reference counterpart. */
int qhip_ctx_allow_deferred_sizes(qhip_ctx* ctx, int32_t delta);
int qhip_ctx_device_name(const qhip_ctx* ctx, char* buf, size_t buflen);
/* Forget what the context has LEARNT about the plans it ran — lowered plans (with them the aggregates' remembered group
 * counts and pre-zeroed tables, which live in the lowered plan), remembered join output sizes, duplicate-key flags — while
 * compiled kernels stay loaded. Joins of deferred size still in flight are checked first (their verdict is discarded with
 * the hints: nothing may consume such a table across this call): the next execution of a
 * query is then a first execution with a warm code cache (what the reference's caller sees for every new
 * `session.sql`, execution/session.rs:74-104: it builds a new plan each time). Statistics cached on TABLE columns (value
 * ranges, longest strings) are properties of the data and stay. bench.py measures first_execution_ms this way. */
int qhip_ctx_forget_plans(qhip_ctx* ctx);
/* Forget what the library has learnt about a TABLE's columns: value statistics (max |value|, value range, longest string) and
 * the narrow copies made from them. The next big operator over the table collects them again — what a query over a
 * freshly uploaded table pays once (bench.py: records.*.cold_first_query_ms / table_prepare_ms). */
int qhip_table_forget_statistics(qhip_table* t);
/* Bytes of HBM the table's auxiliary copies occupy beside its Arrow-layout buffers (the narrow copies of Decimal128 / Int64
 * columns); qhip_table_column_bytes reports the Arrow-layout bytes. */
int64_t qhip_table_aux_bytes(const qhip_table* t);

/* ---------------------------------------------------------------- tables (Vec<RecordBatch> in HBM) */
/* Upload n_batches struct-typed ArrowArrays (one per RecordBatch, children = columns)
 * sharing `schema` (a struct ArrowSchema). Host buffers are only read during the call
 * and are not released by the library. Replaces MemoryTable::try_new + the batches a
 * child node's execute() returned (datasource/memory.rs:29-35). */
int qhip_table_from_arrow(qhip_ctx* ctx, const struct ArrowSchema* schema,
                          const struct ArrowArray* const* batches, int64_t n_batches,
                          qhip_table** out);
/* Like qhip_table_from_arrow, but nothing is copied yet: the library takes OWNERSHIP of the batches (Arrow C Data
 * Interface move: each `batches[b]->release` is set to NULL) and uploads a column the first time an operator or an
 * export reads it. Columns no query touches never cross PCIe — the projection pushdown the reference's Scan node has a
 * slot for but never uses (physical/plan/scan.rs:12-17, planner/mod.rs:251-256). */
int qhip_table_from_arrow_lazy(qhip_ctx* ctx, const struct ArrowSchema* schema, struct ArrowArray* const* batches,
                               int64_t n_batches, qhip_table** out);
/* Download batch `batch_index` as a struct ArrowArray (+ schema if out_schema != NULL).
 * Buffers are library-owned host memory freed by the release callbacks.
 * batch_index == -1: every row of the table as ONE array — a caller facing thousands of small batches (the reference's
 * CSV loader makes 1024-row batches, datasource/file/csv.rs:63-66) downloads once and slices on the host at the
 * boundaries qhip_table_batch_offsets returns (n_out = qhip_table_num_batches + 1 entries: first 0, last = rows). */
int qhip_table_to_arrow(qhip_ctx* ctx, const qhip_table* t, int64_t batch_index,
                        struct ArrowArray* out_array, struct ArrowSchema* out_schema);
int64_t qhip_table_num_batches(const qhip_table* t);
int qhip_table_batch_offsets(const qhip_table* t, int64_t* out, int64_t n_out);
int64_t qhip_table_num_rows(const qhip_table* t);
int64_t qhip_table_num_columns(const qhip_table* t);
/* bytes resident in HBM for column `col` (all buffers) — used for the roofline's algorithmic bytes */
int64_t qhip_table_column_bytes(const qhip_table* t, int64_t col);
void qhip_table_destroy(qhip_table* t);

/* ---------------------------------------------------------------- operators */
/* Filter::execute (physical/plan/filter.rs:28-44) and MemoryTable::scan with a pushed-down
 * filter (datasource/memory.rs:69-98): keep rows whose predicate is valid AND true, all
 * columns compacted, one output batch per input batch (possibly empty), row order kept.
 * `projection` (n_projection >= 0 column indices, or NULL) is memory.rs:79-88. */
int qhip_filter_execute(qhip_ctx* ctx, const qhip_table* input,
                        const qhip_expr* exprs, int32_t n_exprs, int32_t predicate_root,
                        const int32_t* projection, int32_t n_projection,
                        qhip_table** out);

/* HashAggregate::execute (physical/plan/aggregate/hash.rs:138-170) with the scan filter of
 * its input fused in (predicate_root < 0: none). Output: one batch of G rows, columns =
 * group keys then aggregates in declaration order; zero input batches -> zero output
 * batches (hash.rs:146-148). n_groups == 0 runs NoGroupingAggregate::execute
 * (aggregate/no_grouping.rs:30-62): always exactly one row. */
int qhip_hash_aggregate_execute(qhip_ctx* ctx, const qhip_table* input,
                                const qhip_expr* exprs, int32_t n_exprs, int32_t predicate_root,
                                const int32_t* group_roots, int32_t n_groups,
                                const qhip_agg* aggs, int32_t n_aggs,
                                const char* const* out_names, /* n_groups + n_aggs names, or NULL */
                                qhip_table** out);

/* HashJoinExec::execute (physical/plan/join/hash_join.rs:354-384): build = left, probe =
 * right per batch. on_left[i]/on_right[i] are expression roots over the left/right schema
 * (JoinOn, join/mod.rs:20). filter_root >= 0: residual JoinFilter (join/mod.rs:125-154)
 * evaluated over an intermediate schema whose column k is
 * (filter_sides[k] == 0 ? left : right).column(filter_cols[k]). Output: left ++ right
 * columns (left only for semi/anti), one batch per non-empty probe batch in probe order,
 * matches of a probe row in ascending build-row order, then the unmatched-build tail
 * (hash_join.rs:277-343).
 * left/right_scan_filter_root >= 0 (Inner joins only): the child is a Scan with a pushed-down filter
 * (datasource/memory.rs:90-93); pass the UNFILTERED table and the predicate (a root in left_exprs / right_exprs):
 * rejected rows are never inserted / probed, so the filtered batch is never materialised. The result equals
 * joining the filtered tables (one output batch per probe batch either way, Filter keeps batch boundaries). */
int qhip_hash_join_execute(qhip_ctx* ctx, const qhip_table* left, const qhip_table* right,
                           int32_t join_type,
                           const qhip_expr* left_exprs, int32_t n_left_exprs,
                           const qhip_expr* right_exprs, int32_t n_right_exprs,
                           const int32_t* on_left, const int32_t* on_right, int32_t n_on,
                           const qhip_expr* filter_exprs, int32_t n_filter_exprs, int32_t filter_root,
                           const int32_t* filter_sides, const int32_t* filter_cols, int32_t n_filter_cols,
                           int32_t left_scan_filter_root, int32_t right_scan_filter_root,
                           qhip_table** out);

/* ---------------------------------------------------------------- exchange (multi-GPU hash-join repartition) */
/* Split `input` into n_parts tables by mix64(key) of the join-key expressions so that equal
 * keys land in the same part on every rank; part p is sent to rank p by the caller's
 * all-to-all (RCCL over xGMI). Rows keep their relative order inside a part. */
int qhip_partition_by_key(qhip_ctx* ctx, const qhip_table* input,
                          const qhip_expr* exprs, int32_t n_exprs,
                          const int32_t* key_roots, int32_t n_keys,
                          int32_t n_parts, qhip_table** out_parts /* n_parts entries */);
/* The same split FUSED with what surrounds it in a repartitioned join (round 4; SURVEY §8e "scan+filter, radix-partition
 * surviving rows into n send buffers"): predicate_root >= 0 is the scan filter of the join side (MemoryTable::scan's,
 * datasource/memory.rs:90-93 — a row whose predicate is false or NULL is in no part), keep_columns (one int32 per input
 * column, NULL = all) names the columns the plan above the exchange reads: only those are moved, the others become
 * QHIP_NULL placeholders exactly as qhip_table_keep_columns makes them. Two streaming passes, one host wait (the parts'
 * sizes); every kept plain column lands in ONE allocation over all parts and a part's column is a slice of it. n_parts <= 255
 * (more: the generic path of qhip_partition_by_key, which takes no filter). */
int qhip_partition_filtered(qhip_ctx* ctx, const qhip_table* input,
                            const qhip_expr* exprs, int32_t n_exprs,
                            const int32_t* key_roots, int32_t n_keys,
                            int32_t predicate_root, const int32_t* keep_columns,
                            int32_t n_parts, qhip_table** out_parts /* n_parts entries */);
/* ... with the part of a row chosen by KEY RANGE instead of by hash (round 4; DESIGN.md §7 "routing by key range"): ONE
 * integer-like key; upper_bounds = n_parts - 1 ascending values, part p takes the keys in (upper_bounds[p - 1], upper_bounds[p]],
 * the last part everything above (a NULL key goes where 0 goes). Both sides of a join and every rank must use the SAME bounds —
 * then the join is correct whatever they are; bounds taken from the ranks' own key ranges (qhip_table_column_range) keep the
 * rows of tables sliced in key order where they are. upper_bounds == NULL: by hash, i.e. qhip_partition_filtered. */
int qhip_partition_filtered_by_range(qhip_ctx* ctx, const qhip_table* input,
                                     const qhip_expr* exprs, int32_t n_exprs,
                                     const int32_t* key_roots, int32_t n_keys,
                                     int32_t predicate_root, const int32_t* keep_columns,
                                     const int64_t* upper_bounds /* n_parts - 1, or NULL */,
                                     int32_t n_parts, qhip_table** out_parts /* n_parts entries */);
/* [min, max] of an integer-like column's values (NULL slots included as stored), computed once per base column and cached; a
 * column that is a deferred gather answers with its SOURCE's range (a superset). QHIP_UNSUPPORTED for other column types. */
int qhip_table_column_range(qhip_ctx* ctx, const qhip_table* t, int64_t col, int64_t* out_min, int64_t* out_max);
/* Concatenate tables with identical schemas (batches appended in order). */
int qhip_table_concat(qhip_ctx* ctx, const qhip_table* const* tables, int32_t n, qhip_table** out);

/* Raw device access for the exchange plumbing (torch.distributed moves the bytes):
 * buffer k of column c: 0 = values (or utf8 offsets), 1 = validity bitmap, 2 = utf8 data. */
int qhip_table_column_buffer(const qhip_table* t, int64_t col, int32_t which,
                             void** device_ptr, int64_t* n_bytes);
/* Build a table from device buffers received from peers; the library copies them. */
typedef struct qhip_device_column {
  qhip_dtype dtype;
  int64_t length;
  int64_t null_count;
  const void* values;        /* device: fixed-width values, or int32 offsets (length+1) for utf8 */
  const void* validity;      /* device bitmap or NULL */
  const void* data;          /* device utf8 bytes or NULL */
  int64_t data_bytes;
} qhip_device_column;
int qhip_table_from_device(qhip_ctx* ctx, const char* const* names, const qhip_device_column* cols,
                           int32_t n_cols, int64_t n_rows, qhip_table** out);

/* Wire image of a table: ONE contiguous device buffer per (source, destination) pair, so that an exchange is one small
 * metadata round plus one payload round whatever the number of columns. Sections in column order — values (or utf8
 * offsets / Boolean bits), validity, utf8 data — each aligned to 16 bytes. The image has no header: both sides derive
 * the layout from the schema and the table's metadata words
 *     meta[0] = rows, meta[1] = image bytes, meta[2 + 2c] = null count of column c, meta[3 + 2c] = its utf8 data bytes
 * (n_meta = 2 + 2 * columns), which travel in the metadata round.
 * qhip_table_pack writes the image into caller-owned device memory (e.g. a torch tensor) and returns once it is complete.
 * qhip_table_unpack_concat builds the concatenation, in the order given, of the tables n images describe (one batch per
 * image); it has copied what it needs when it returns. No reference counterpart (the reference is a single process). */
int qhip_table_wire_meta(qhip_ctx* ctx, const qhip_table* t, int64_t* meta, int32_t n_meta);
/* Projection pushdown through the exchange: a view of `t` in which column c is kept (keep[c] != 0, buffers shared) or
 * replaced by a NULL-typed column of the same length — zero bytes on the wire, column positions unchanged. A column the
 * plan above the exchange never reads is neither gathered nor sent (the reference's join gathers every column,
 * utils/batch.rs:18-61; on one GPU the deferred gathers already avoid that). */
int qhip_table_keep_columns(qhip_ctx* ctx, const qhip_table* t, const int32_t* keep, int32_t n_cols, qhip_table** out);
/* Every `stride`-th row of `t` (rows 0, stride, 2 stride, ...) as a one-batch table of deferred gathers: the sample a
 * repartitioned join counts its probe keys on to find heavy hitters before it decides where rows go. */
int qhip_table_stride_sample(qhip_ctx* ctx, const qhip_table* t, int64_t stride, qhip_table** out);
int qhip_table_pack(qhip_ctx* ctx, const qhip_table* t, void* device_dst, int64_t dst_bytes);
int qhip_table_unpack_concat(qhip_ctx* ctx, const char* const* names, const qhip_dtype* dtypes, int32_t n_cols,
                             const int64_t* metas /* n x (2 + 2 * n_cols) */, const void* const* device_images,
                             int32_t n, qhip_table** out);

/* ---------------------------------------------------------------- RCCL inside the library */
/* The transport of the exchange for a host without torch (the reference's host is Rust): a communicator over the ranks of
 * the job, one per process and GPU, and the two collective steps of a partitioned hash join. librccl.so is dlopen'ed at
 * first use (QHIP_RCCL_LIB overrides the name; a copy the process already holds is reused), failures are QHIP_RCCL_ERROR.
 * All transfers run on the context's own stream, grouped ncclSend / ncclRecv (xGMI is point-to-point: one group drives
 * all of a GPU's links at once); the one host wait of an exchange is the read-back of the incoming parts' sizes.
 * No reference counterpart (the reference is a single process).
 *   rank 0:      qhip_comm_unique_id(id, 128)  -> distribute the 128 bytes to every rank (the host's own channel)
 *   every rank:  qhip_comm_create(ctx, id, rank, world, &comm)     (world == 1: no RCCL call is made at all)
 *   per join:    qhip_partition_by_key -> qhip_exchange_tables (repartition)   or   qhip_all_gather_table (broadcast) */
typedef struct qhip_comm qhip_comm;
typedef struct qhip_comm_stats {
  uint64_t bytes_sent;        /* to other ranks, since the last reset */
  uint64_t bytes_received;    /* from other ranks */
  uint64_t bytes_packed;      /* wire images built, incl. the part that stayed on this rank */
  uint64_t exchanges;         /* qhip_exchange_tables + qhip_all_gather_table calls */
  uint64_t host_waits;        /* stream synchronisations made inside them (one per exchange: the sizes) */
  double transfer_seconds;    /* device time between the first and the last transfer of every exchange (HIP events) */
  int32_t rank, world;
  int32_t rccl_version;       /* ncclGetVersion, 0 for a world of one */
  int32_t reserved;
} qhip_comm_stats;
int qhip_comm_unique_id(void* id_out, size_t id_bytes /* >= 128 */);
int qhip_comm_create(qhip_ctx* ctx, const void* unique_id /* 128 bytes */, int32_t rank, int32_t world, qhip_comm** out);
void qhip_comm_destroy(qhip_comm* comm);
int qhip_comm_get_stats(qhip_comm* comm, qhip_comm_stats* out, int32_t reset);
/* parts[r] (world tables of one schema, e.g. from qhip_partition_by_key) is sent to rank r; *out = the concatenation, in
 * rank order, of the parts every rank sent to this one (one batch per rank). names / dtypes: the schema, n_cols columns
 * (a column dropped by qhip_table_keep_columns is QHIP_NULL on both sides). */
int qhip_exchange_tables(qhip_ctx* ctx, qhip_comm* comm, const qhip_table* const* parts, const char* const* names,
                         const qhip_dtype* dtypes, int32_t n_cols, qhip_table** out);
/* Every rank ends up with the concatenation (rank order) of all ranks' `t`: the build side of a broadcast join. */
int qhip_all_gather_table(qhip_ctx* ctx, qhip_comm* comm, const qhip_table* t, const char* const* names,
                          const qhip_dtype* dtypes, int32_t n_cols, qhip_table** out);

/* The whole exchange step of a distributed hash join in ONE call with ONE host wait (round 4): for every input — the two sides
 * of a repartitioned join, or the build side of a broadcast join (all_gather != 0: every rank receives every kept row) — the
 * scan filter and the key are evaluated and the rows counted per destination (pass 1 of qhip_partition_filtered), the counts
 * of all inputs are all-gathered as device-resident metadata, the host waits ONCE (everybody's sizes), then the kept columns
 * are moved into per-destination runs (pass 2) and sent with one group of ncclSend / ncclRecv per input STRAIGHT from those
 * runs INTO the final columns of the result (no wire image, no pack / unpack copies). The transfers run on the
 * communicator's own stream: input k's transfers overlap input k + 1's pass 2.
 * An input may be a join output of deferred size (qhip_ctx_allow_deferred_sizes): whether such a join had too little room is
 * part of the metadata, so that EVERY rank returns QHIP_RETRY together (and re-executes its input) — never one rank alone.
 * Only tables whose kept columns are fixed-width without NULLs travel this way; QHIP_UNSUPPORTED (decided by column types,
 * identically on every rank, before any collective) or — when NULLs turn up on some rank — an agreed QHIP_UNSUPPORTED after the
 * metadata round tells the caller to take qhip_partition_filtered + qhip_exchange_tables instead.
 * *outs[k]: one batch per source rank, columns as the input's (dropped columns QHIP_NULL). */
typedef struct qhip_shuffle_input {
  const qhip_table* table;
  const qhip_expr* exprs; int32_t n_exprs;
  const int32_t* key_roots; int32_t n_keys;     /* the join keys (also for all_gather: they are not hashed then) */
  int32_t predicate_root;                        /* the side's scan filter or -1 */
  int32_t all_gather;                            /* bit 0: 0 = row -> rank mix(key) (repartition), 1 = every row to every rank
                                                  * (broadcast); bit 1: a JOIN on these keys follows — the value range of a key
                                                  * column travels with it (found once per base column: one reduction + wait) so
                                                  * that the join can address its table by the key without reducing what it received */
  const int32_t* keep_columns;                   /* per input column, NULL = all */
  const int64_t* range_bounds;                   /* NULL: rows go to rank hash(key) % world; else world - 1 ascending upper bounds of
                                                  * ONE integer-like key, identical on every rank and for both sides of the join
                                                  * (qhip_partition_filtered_by_range) */
} qhip_shuffle_input;
int qhip_shuffle_tables(qhip_ctx* ctx, qhip_comm* comm, const qhip_shuffle_input* inputs, int32_t n_inputs, qhip_table** outs);

/* ---------------------------------------------------------------- plan-only entry points (no GPU needed) */
/* Return (snprintf-style; *needed = bytes incl. NUL) the policy source libqhip instantiates the kernel
 * templates of csrc/device/qhip_device.hpp with, for an input whose column c has type col_types[c] and
 * (col_has_nulls[c] != 0) a validity bitmap. qhip_jit_compile_to_cache compiles such a source for gfx950
 * with hiprtc and stores the code object in cache_dir, where contexts pick it up instead of compiling. */
int qhip_plan_aggregate_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                               const qhip_expr* exprs, int32_t n_exprs, int32_t predicate_root,
                               const int32_t* group_roots, int32_t n_groups, const qhip_agg* aggs, int32_t n_aggs,
                               char* buf, size_t buflen, size_t* needed);
int qhip_plan_filter_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                            const qhip_expr* exprs, int32_t n_exprs, int32_t predicate_root,
                            char* buf, size_t buflen, size_t* needed);
int qhip_plan_keys_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                          const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots, int32_t n_keys,
                          char* buf, size_t buflen, size_t* needed);
/* The fused probe kernel of qhip_hash_join_execute (scan filter + key words + table lookup + ordered pair emit) for the
 * probe side's key expressions; predicate_root = -1 for no fused scan filter. */
int qhip_plan_probe_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                           const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots, int32_t n_keys,
                           int32_t predicate_root, char* buf, size_t buflen, size_t* needed);
/* The build-side kernel of qhip_hash_join_execute's LDS-staged build (scan filter + key words -> region entries of the
 * join table, hash_join.rs:148-175); predicate_root = -1 for no fused scan filter. */
int qhip_plan_scatter_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                             const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots, int32_t n_keys,
                             int32_t predicate_root, char* buf, size_t buflen, size_t* needed);
/* Pass 1 of qhip_partition_filtered (scan filter + key -> part of every row + per-wavefront histogram) for n_parts parts. */
int qhip_plan_partition_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                               const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots, int32_t n_keys,
                               int32_t predicate_root, int32_t n_parts, char* buf, size_t buflen, size_t* needed);
/* Pass 2 of qhip_partition_filtered for one group of columns: widths[c] = bytes per value (1, 2, 4, 8, 16; 0 = the row number
 * as u32), indirect[c] != 0 = read through an index vector (a deferred gather). */
int qhip_plan_part_scatter_source(const int32_t* widths, const int32_t* indirect, int32_t n_cols, int32_t n_parts,
                                  char* buf, size_t buflen, size_t* needed);
/* The image kernel of qhip_sort_execute for the key expressions, and the kernel of qhip_projection_execute for the
 * computed (non-Column) expressions among `roots`. */
int qhip_plan_sort_keys_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                               const qhip_expr* exprs, int32_t n_exprs, const int32_t* key_roots, int32_t n_keys,
                               char* buf, size_t buflen, size_t* needed);
int qhip_plan_projection_source(const qhip_dtype* col_types, const int32_t* col_has_nulls, int32_t n_cols,
                                const qhip_expr* exprs, int32_t n_exprs, const int32_t* roots, int32_t n_out,
                                char* buf, size_t buflen, size_t* needed);
const char* qhip_plan_last_error(void);
int qhip_jit_compile_to_cache(const char* policy_source, const char* cache_dir, char* log, size_t log_len);

/* ---------------------------------------------------------------- joins without equi-keys */
/* NestedLoopJoinExec::execute (physical/plan/join/nest_loop_join.rs:79-228): every (left, right) row pair in right-row
 * major order, kept when the optional JoinFilter (same encoding as qhip_hash_join_execute; filter_root < 0 = none) is
 * true. Output batches like the reference: [matched] for Inner, [matched, unmatched] for Left / Right / Full (unmatched
 * left rows first, then unmatched right rows), one batch of left rows for LeftSemi / LeftAnti; an empty right side gives
 * no batch (Inner / Right), the left rows with a NULL right side (Left / Full / LeftAnti) or one empty batch (LeftSemi).
 * Inputs whose row-pair count reaches 2^31 return QHIP_UNSUPPORTED. */
int qhip_nested_loop_join_execute(qhip_ctx* ctx, const qhip_table* left, const qhip_table* right, int32_t join_type,
                                  const qhip_expr* filter_exprs, int32_t n_filter_exprs, int32_t filter_root,
                                  const int32_t* filter_sides, const int32_t* filter_cols, int32_t n_filter_cols,
                                  qhip_table** out);
/* CrossJoin::execute (physical/plan/join/cross_join.rs:121-166): for every left batch, right batch and left row one
 * output batch (that left row next to the right batch). */
int qhip_cross_join_execute(qhip_ctx* ctx, const qhip_table* left, const qhip_table* right, qhip_table** out);

/* ---------------------------------------------------------------- projection */
/* Projection::execute (physical/plan/projection.rs:27-46): one output column per expression, evaluated over every
 * input batch (batch structure kept). A plain Column expression shares the input column's buffers; everything else is
 * computed by ONE generated kernel for all expressions together — run twice when an expression is of Utf8 type (a CASE over
 * strings, tests/sql/type.slt:51): lengths, a scan into int32 offsets, then the bytes. A computed Utf8 column that could exceed
 * 2 GiB returns QHIP_UNSUPPORTED (LargeUtf8 does not cross this ABI). */
int qhip_projection_execute(qhip_ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int32_t n_exprs,
                            const int32_t* roots, int32_t n_out, const char* const* out_names, qhip_table** out);

/* ---------------------------------------------------------------- sort / limit */
/* Sort::execute (physical/plan/sort.rs:48-82): concatenates the input, orders it by the key expressions with arrow's
 * lexsort semantics (per key: descending[k], nulls_first[k]; floats in IEEE total order; ties keep the input order,
 * sort.rs:62-73) and returns ONE batch. limit >= 0 keeps only the first `limit` rows (Sort::new_with_limit, the
 * planner's top-N pushdown planner/mod.rs:69-75); limit < 0 = no limit. */
int qhip_sort_execute(qhip_ctx* ctx, const qhip_table* in, const qhip_expr* exprs, int32_t n_exprs,
                      const int32_t* key_roots, const int32_t* descending, const int32_t* nulls_first, int32_t n_keys,
                      int64_t limit, qhip_table** out);
/* Limit::execute (physical/plan/limit.rs:27-58): rows [skip, skip + fetch) of the batch list, batch structure kept
 * (fetch < 0 = no fetch limit). */
int qhip_limit_execute(qhip_ctx* ctx, const qhip_table* in, int64_t skip, int64_t fetch, qhip_table** out);

/* (The synthetic-table generators and the streaming-read yardstick bench.py uses are NOT part of this boundary: include/qhip_bench.h,
 * built into libqhip_bench.so.) */

#ifdef __cplusplus
}
#endif
#endif /* QHIP_H */
