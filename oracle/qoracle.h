/*
 * qoracle.h — CPU restatement of the reference's filter / hash-aggregate / hash-join path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (qurious_amd/, libqhip.so) includes, links or calls
 * this. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker /
 * the timed CPU baseline ("port"). It restates, function by function, the algorithm of
 * /root/reference/qurious/src (file:line cited at each function) and the documented behaviour of the
 * third-party kernels it calls: arrow = "53.2.0" (Cargo.toml:20, no Cargo.lock => 53.x) and Rust std
 * (std::hash::DefaultHasher = SipHash-1-3 with zero keys).
 *
 * Pinning: tests/test_oracle_golden.py replays every golden vector the reference's own tests hold for
 * this path (JoinHashMap vectors, HashJoinExec ordered outputs, binary.rs expression
 * vectors incl. the Decimal128 case, the .slt goldens, Q1 SF0.01 algebraic checks) against this code.
 * The SipHash VALUES themselves are not pinned by any reference test (they never leave the operators;
 * only their equivalence classes matter) — tests pin the implementation against the SipHash-1-3
 * definition with an independent pure-Python restatement.
 *
 * "Faithful cost": besides semantics the code keeps the reference's algorithmic structure (literal
 * broadcast + cast per batch, full-length temporaries per expression node, one 72-byte hasher state per
 * row, hash -> group map, per-group index lists, per-group per-aggregate gather + reduce, chained join
 * map built in reverse, candidate gather + equality re-check) because bench.py times it as the stand-in
 * for the reference's single-threaded CPU executor.
 */
#ifndef QORACLE_H
#define QORACLE_H
#include <stddef.h>
#include <stdint.h>

#include "../include/qhip.h" /* only for the POD descriptors (qhip_expr, qhip_agg, qhip_dtype, enums) */

#ifdef __cplusplus
extern "C" {
#endif

/* a column (or an expression result): validity is one BYTE per row (1 = valid) or NULL = all valid */
typedef struct qo_col {
  qhip_dtype type;
  int64_t n;
  void* values;          /* fixed width: typed array; Boolean: one byte per row; Utf8: unused */
  uint8_t* valid;
  int32_t* offsets;      /* Utf8: n + 1 */
  uint8_t* data;         /* Utf8 */
  int owned;             /* buffers malloc'ed by the oracle (freed by qo_col_free) */
} qo_col;

void qo_col_free(qo_col* c);
const char* qo_last_error(void);

/* ---- std::hash::DefaultHasher::new(): SipHash-1-3, k0 = k1 = 0, streaming (72-byte state like Rust's) */
typedef struct qo_hasher {
  uint64_t k0, k1;
  uint64_t length;
  uint64_t v0, v2, v1, v3;
  uint64_t tail;
  uint64_t ntail;
} qo_hasher;
void qo_hasher_init(qo_hasher* h);
void qo_hasher_write(qo_hasher* h, const uint8_t* msg, size_t len);
uint64_t qo_hasher_finish(const qo_hasher* h);

/* utils/array.rs:190-210 create_hashes: per-row hash of the key columns; returns 0 or an error code */
int qo_create_hashes(const qo_col* cols, int ncols, int64_t nrows, uint64_t* out);

/* physical/expr/{binary,cast,literal,column,case,like}.rs evaluate(): column-at-a-time evaluation with full-length temporaries */
int qo_eval(const qhip_expr* exprs, int n_exprs, int root, const qo_col* batch_cols, int ncols, int64_t nrows, qo_col* out);

/* filter_record_batch semantics (physical/plan/filter.rs:34): rows whose mask is valid AND true.
 * sel must hold nrows entries; returns the number selected or < 0 */
int64_t qo_filter_indices(const qo_col* mask, int64_t* sel);

/* ---- JoinHashMap (physical/plan/join/hash_join.rs:39-107) */
typedef struct qo_join_map qo_join_map;
qo_join_map* qo_join_map_with_capacity(int64_t capacity);
void qo_join_map_free(qo_join_map* m);
/* update(): rows are visited in the order given (rows[k], hashes[rows[k]]); hash_join.rs:52-64 */
void qo_join_map_update(qo_join_map* m, const uint64_t* hashes, const int64_t* rows, int64_t nrows, int64_t delete_offset);
int qo_join_map_is_distinct(const qo_join_map* m);
int64_t qo_join_map_len(const qo_join_map* m);
int64_t qo_join_map_get(const qo_join_map* m, uint64_t hash); /* stored value (row + 1) or 0 */
const uint64_t* qo_join_map_next(const qo_join_map* m, int64_t* n);
/* get_matches_indices (hash_join.rs:70-107): returns count; arrays malloc'ed, free with qo_free */
int64_t qo_join_map_get_matches(const qo_join_map* m, const uint64_t* probe_hashes, int64_t nprobe, uint32_t** input_indices,
                                uint64_t** match_indices);
void qo_free(void* p);

/* ---- HashJoinExec (hash_join.rs:148-384) on already concatenated build side + one probe batch.
 * Key columns are the evaluated `on` expressions. Output: candidate filtering by key equality
 * (probe_hash_table :177-216) -> (build_idx, probe_idx) pairs, no join-type adjustment yet. */
int64_t qo_probe_hash_table(const qo_join_map* m, const qo_col* build_keys, const qo_col* probe_keys, int nkeys, int64_t nprobe,
                            uint64_t** build_idx, uint32_t** probe_idx);
/* adjust_right_indices (join/mod.rs:176-207): build index -1 = NULL. Arrays malloc'ed. */
int64_t qo_adjust_right_indices(const uint64_t* build_idx, const uint32_t* probe_idx, int64_t n, int64_t right_rows,
                                int64_t** out_build, int64_t** out_probe);

/* ---- GroupAccumulator::update + output (physical/plan/aggregate/hash.rs:45-107), accumulators of
 * physical/expr/aggregate/{sum,avg,count,min,max}.rs. keys/args are the evaluated expression arrays
 * over the concatenated batch. Groups come out in first-seen order; first_row[g] = row that created
 * the group (its key values are the group's key, hash.rs:60-68). n_keys == 0 runs the
 * NoGroupingAggregate protocol instead (no_grouping.rs:30-62) with `n_batches` accumulate calls. */
typedef struct qo_agg_result {
  int64_t n_groups;
  int64_t* first_row;   /* n_groups */
  qo_col* agg_cols;     /* n_aggs result columns of n_groups rows, typed per return_type */
  int n_aggs;
} qo_agg_result;
int qo_hash_aggregate(const qo_col* keys, int n_keys, const qo_col* args, const qhip_agg* aggs, int n_aggs, int64_t nrows,
                      const int64_t* batch_offsets, int64_t n_batches, qo_agg_result* out);
void qo_agg_result_free(qo_agg_result* r);

/* ---- faithful-cost whole pipeline Scan(filter) -> HashAggregate, timed by bench.py as cpu_baseline.
 * batches[b * ncols + c] = column c of stored batch b. See qoracle.c for the steps mirrored. */
int qo_scan_filter_aggregate(const qo_col* batches, int64_t nbatches, int ncols, const int64_t* batch_rows, const qhip_expr* exprs,
                             int n_exprs, int pred_root, const int32_t* group_roots, int n_groups, const qhip_agg* aggs, int n_aggs,
                             qo_agg_result* out, qo_col* out_keys, int64_t* rows_after_filter);

#ifdef __cplusplus
}
#endif
#endif
