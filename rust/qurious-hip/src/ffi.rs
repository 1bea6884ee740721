//! `extern "C"` declarations of libqhip.so — one per entry point of `include/qhip.h` that a query host binds, the structs
//! `#[repr(C)]` copies of the header's PODs (same field order). `tests/test_rust_shim.py` parses this file and the header
//! and checks names, argument counts and field order against each other.
//!
//! `FFI_ArrowArray` / `FFI_ArrowSchema` are arrow-rs' Arrow C Data Interface structs (`arrow::ffi`, arrow 53 — already a
//! dependency of the reference, `Cargo.toml:20`).
#![allow(non_camel_case_types)]

use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use std::os::raw::{c_char, c_int, c_void};

// ---------------------------------------------------------------- status codes (qhip_status)
pub const QHIP_OK: c_int = 0;
pub const QHIP_INVALID_ARGUMENT: c_int = 1;
pub const QHIP_UNSUPPORTED: c_int = 2;
pub const QHIP_HIP_ERROR: c_int = 3;
pub const QHIP_OUT_OF_MEMORY: c_int = 4;
pub const QHIP_EXEC_ERROR: c_int = 5;
pub const QHIP_RCCL_ERROR: c_int = 6;
pub const QHIP_RETRY: c_int = 7;

// ---------------------------------------------------------------- qhip_type_id
pub const QHIP_NULL: i32 = 0;
pub const QHIP_BOOL: i32 = 1;
pub const QHIP_INT8: i32 = 2;
pub const QHIP_INT16: i32 = 3;
pub const QHIP_INT32: i32 = 4;
pub const QHIP_INT64: i32 = 5;
pub const QHIP_UINT8: i32 = 6;
pub const QHIP_UINT16: i32 = 7;
pub const QHIP_UINT32: i32 = 8;
pub const QHIP_UINT64: i32 = 9;
pub const QHIP_FLOAT32: i32 = 10;
pub const QHIP_FLOAT64: i32 = 11;
pub const QHIP_DATE32: i32 = 12;
pub const QHIP_DATE64: i32 = 13;
pub const QHIP_DECIMAL128: i32 = 14;
pub const QHIP_UTF8: i32 = 15;
pub const QHIP_TIME32_S: i32 = 16;
pub const QHIP_TIME32_MS: i32 = 17;
pub const QHIP_TIME64_US: i32 = 18;
pub const QHIP_TIME64_NS: i32 = 19;
pub const QHIP_TIMESTAMP_S: i32 = 20;
pub const QHIP_TIMESTAMP_MS: i32 = 21;
pub const QHIP_TIMESTAMP_US: i32 = 22;
pub const QHIP_TIMESTAMP_NS: i32 = 23;

// ---------------------------------------------------------------- qhip_expr_kind
pub const QHIP_EXPR_COLUMN: i32 = 0;
pub const QHIP_EXPR_LITERAL: i32 = 1;
pub const QHIP_EXPR_BINARY: i32 = 2;
pub const QHIP_EXPR_CAST: i32 = 3;
pub const QHIP_EXPR_IS_NULL: i32 = 4;
pub const QHIP_EXPR_IS_NOT_NULL: i32 = 5;
pub const QHIP_EXPR_NEGATIVE: i32 = 6;
pub const QHIP_EXPR_IF: i32 = 7;
pub const QHIP_EXPR_LIKE: i32 = 8;

// ---------------------------------------------------------------- qhip_agg_kind
pub const QHIP_AGG_SUM: i32 = 0;
pub const QHIP_AGG_AVG: i32 = 1;
pub const QHIP_AGG_COUNT: i32 = 2;
pub const QHIP_AGG_MIN: i32 = 3;
pub const QHIP_AGG_MAX: i32 = 4;

// opaque handles
#[repr(C)]
pub struct qhip_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct qhip_table {
    _private: [u8; 0],
}
#[repr(C)]
pub struct qhip_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct qhip_dtype {
    pub id: i32,
    pub precision: i32,
    pub scale: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct qhip_expr {
    pub kind: i32,
    pub op: i32,
    pub column: i32,
    pub left: i32,
    pub right: i32,
    pub third: i32,
    pub dtype: qhip_dtype,
    pub lit_is_null: i32,
    pub lit_lo: u64,
    pub lit_hi: i64,
    pub lit_f64: f64,
    pub lit_str: *const c_char,
    pub lit_len: i64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct qhip_agg {
    pub kind: i32,
    pub expr: i32,
    pub return_type: qhip_dtype,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct qhip_exec_stats {
    pub main_kernel_ms: f64,
    pub total_device_ms: f64,
    pub jit_ms: f64,
    pub rows_in: i64,
    pub rows_out: i64,
    pub groups: i64,
    pub table_capacity: i64,
    pub retries: i32,
    pub lds_table_slots: i32,
    pub main_kernel_name: [c_char; 64],
    pub lds_occupancy: f64,
    pub hbm_table_load: f64,
    pub lds_spilled: i32,
    pub workgroups: i32,
    pub bytes_per_row_read: f64,
    pub build_ms: f64,
    pub build_rows: i64,
    pub build_bytes_per_row: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct qhip_comm_stats {
    pub bytes_sent: u64,
    pub bytes_received: u64,
    pub bytes_packed: u64,
    pub exchanges: u64,
    pub host_waits: u64,
    pub transfer_seconds: f64,
    pub rank: i32,
    pub world: i32,
    pub rccl_version: i32,
    pub reserved: i32,
}

/// one input of `qhip_shuffle_tables`: a join side (or the build side of a broadcast join) with its scan filter, keys and the
/// columns the plan above the exchange reads
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct qhip_shuffle_input {
    pub table: *const qhip_table,
    pub exprs: *const qhip_expr,
    pub n_exprs: i32,
    pub key_roots: *const i32,
    pub n_keys: i32,
    pub predicate_root: i32,
    pub all_gather: i32,
    pub keep_columns: *const i32,
    /// NULL: rows go to rank hash(key) % world; else world - 1 ascending upper bounds of ONE integer-like key
    pub range_bounds: *const i64,
}

extern "C" {
    // ---- context
    pub fn qhip_ctx_create(device_index: c_int, out: *mut *mut qhip_ctx) -> c_int;
    pub fn qhip_ctx_destroy(ctx: *mut qhip_ctx);
    pub fn qhip_last_error(ctx: *const qhip_ctx) -> *const c_char;
    pub fn qhip_version() -> *const c_char;
    pub fn qhip_device_available() -> c_int;
    pub fn qhip_ctx_synchronize(ctx: *mut qhip_ctx) -> c_int;
    pub fn qhip_ctx_last_stats(ctx: *const qhip_ctx, out: *mut qhip_exec_stats) -> c_int;
    pub fn qhip_ctx_sync_count(ctx: *const qhip_ctx) -> u64;
    pub fn qhip_ctx_set_timing(ctx: *mut qhip_ctx, on: i32) -> c_int;
    pub fn qhip_ctx_allow_deferred_sizes(ctx: *mut qhip_ctx, delta: i32) -> c_int;
    pub fn qhip_ctx_forget_plans(ctx: *mut qhip_ctx) -> c_int;

    // ---- tables (Vec<RecordBatch> in HBM)
    pub fn qhip_table_from_arrow(
        ctx: *mut qhip_ctx,
        schema: *const FFI_ArrowSchema,
        batches: *const *const FFI_ArrowArray,
        n_batches: i64,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_table_from_arrow_lazy(
        ctx: *mut qhip_ctx,
        schema: *const FFI_ArrowSchema,
        batches: *const *mut FFI_ArrowArray,
        n_batches: i64,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_table_to_arrow(
        ctx: *mut qhip_ctx,
        t: *const qhip_table,
        batch_index: i64,
        out_array: *mut FFI_ArrowArray,
        out_schema: *mut FFI_ArrowSchema,
    ) -> c_int;
    pub fn qhip_table_num_batches(t: *const qhip_table) -> i64;
    pub fn qhip_table_batch_offsets(t: *const qhip_table, out: *mut i64, n_out: i64) -> c_int;
    pub fn qhip_table_num_rows(t: *const qhip_table) -> i64;
    pub fn qhip_table_num_columns(t: *const qhip_table) -> i64;
    pub fn qhip_table_destroy(t: *mut qhip_table);

    // ---- operators
    pub fn qhip_filter_execute(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        predicate_root: i32,
        projection: *const i32,
        n_projection: i32,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_hash_aggregate_execute(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        predicate_root: i32,
        group_roots: *const i32,
        n_groups: i32,
        aggs: *const qhip_agg,
        n_aggs: i32,
        out_names: *const *const c_char,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_hash_join_execute(
        ctx: *mut qhip_ctx,
        left: *const qhip_table,
        right: *const qhip_table,
        join_type: i32,
        left_exprs: *const qhip_expr,
        n_left_exprs: i32,
        right_exprs: *const qhip_expr,
        n_right_exprs: i32,
        on_left: *const i32,
        on_right: *const i32,
        n_on: i32,
        filter_exprs: *const qhip_expr,
        n_filter_exprs: i32,
        filter_root: i32,
        filter_sides: *const i32,
        filter_cols: *const i32,
        n_filter_cols: i32,
        left_scan_filter_root: i32,
        right_scan_filter_root: i32,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_nested_loop_join_execute(
        ctx: *mut qhip_ctx,
        left: *const qhip_table,
        right: *const qhip_table,
        join_type: i32,
        filter_exprs: *const qhip_expr,
        n_filter_exprs: i32,
        filter_root: i32,
        filter_sides: *const i32,
        filter_cols: *const i32,
        n_filter_cols: i32,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_cross_join_execute(ctx: *mut qhip_ctx, left: *const qhip_table, right: *const qhip_table, out: *mut *mut qhip_table) -> c_int;
    pub fn qhip_projection_execute(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        roots: *const i32,
        n_out: i32,
        out_names: *const *const c_char,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_sort_execute(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        key_roots: *const i32,
        descending: *const i32,
        nulls_first: *const i32,
        n_keys: i32,
        limit: i64,
        out: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_limit_execute(ctx: *mut qhip_ctx, input: *const qhip_table, skip: i64, fetch: i64, out: *mut *mut qhip_table) -> c_int;

    // ---- multi-GPU: partitioning + the exchange through RCCL inside the library (one process per GPU)
    pub fn qhip_partition_by_key(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        key_roots: *const i32,
        n_keys: i32,
        n_parts: i32,
        out_parts: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_partition_filtered(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        key_roots: *const i32,
        n_keys: i32,
        predicate_root: i32,
        keep_columns: *const i32,
        n_parts: i32,
        out_parts: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_partition_filtered_by_range(
        ctx: *mut qhip_ctx,
        input: *const qhip_table,
        exprs: *const qhip_expr,
        n_exprs: i32,
        key_roots: *const i32,
        n_keys: i32,
        predicate_root: i32,
        keep_columns: *const i32,
        upper_bounds: *const i64,
        n_parts: i32,
        out_parts: *mut *mut qhip_table,
    ) -> c_int;
    pub fn qhip_table_column_range(ctx: *mut qhip_ctx, t: *const qhip_table, col: i64, out_min: *mut i64, out_max: *mut i64) -> c_int;
    pub fn qhip_table_forget_statistics(t: *mut qhip_table) -> c_int;
    pub fn qhip_table_aux_bytes(t: *const qhip_table) -> i64;
    pub fn qhip_table_concat(ctx: *mut qhip_ctx, tables: *const *const qhip_table, n: i32, out: *mut *mut qhip_table) -> c_int;
    pub fn qhip_table_keep_columns(ctx: *mut qhip_ctx, t: *const qhip_table, keep: *const i32, n_cols: i32, out: *mut *mut qhip_table) -> c_int;
    pub fn qhip_table_stride_sample(ctx: *mut qhip_ctx, t: *const qhip_table, stride: i64, out: *mut *mut qhip_table) -> c_int;
    pub fn qhip_comm_unique_id(id_out: *mut c_void, id_bytes: usize) -> c_int;
    pub fn qhip_comm_create(ctx: *mut qhip_ctx, unique_id: *const c_void, rank: i32, world: i32, out: *mut *mut qhip_comm) -> c_int;
    pub fn qhip_comm_destroy(comm: *mut qhip_comm);
    pub fn qhip_comm_get_stats(comm: *mut qhip_comm, out: *mut qhip_comm_stats, reset: i32) -> c_int;
    pub fn qhip_exchange_tables(
        ctx: *mut qhip_ctx,
        comm: *mut qhip_comm,
        parts: *const *const qhip_table,
        names: *const *const c_char,
        dtypes: *const qhip_dtype,
        n_cols: i32,
        out: *mut *mut qhip_table,
    ) -> c_int;
    /// the whole exchange step of a distributed join — both sides of a repartitioned join, or a broadcast join's build side — in
    /// one call with one host wait; QHIP_RETRY is returned by EVERY rank when a join of deferred size below must run again
    pub fn qhip_shuffle_tables(ctx: *mut qhip_ctx, comm: *mut qhip_comm, inputs: *const qhip_shuffle_input, n_inputs: i32, outs: *mut *mut qhip_table) -> c_int;
    pub fn qhip_all_gather_table(
        ctx: *mut qhip_ctx,
        comm: *mut qhip_comm,
        t: *const qhip_table,
        names: *const *const c_char,
        dtypes: *const qhip_dtype,
        n_cols: i32,
        out: *mut *mut qhip_table,
    ) -> c_int;
}
