//! The operator nodes: each implements the reference's `trait PhysicalPlan` (qurious/src/physical/plan/mod.rs:25-29)
//! unchanged — `schema()`, `execute() -> Result<Vec<RecordBatch>>`, `children()` — and, for its HIP parents, `HipNode`,
//! whose `execute_device()` hands over the HBM-resident table without a download. A child is either another HIP node or any
//! `Arc<dyn PhysicalPlan>` of the reference (its batches are uploaded).
//!
//! Expressions were lowered at plan time (lower.rs); a node is immutable after construction like the reference's.
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_char;
use std::sync::{Arc, Mutex};

use arrow::array::{Array, StructArray};
use arrow::datatypes::{Field, Schema, SchemaRef};
use arrow::ffi::{from_ffi, to_ffi, FFI_ArrowArray, FFI_ArrowSchema};
use arrow::record_batch::RecordBatch;
use qurious::common::join_type::JoinType;
use qurious::error::{Error, Result};
use qurious::physical::plan::PhysicalPlan;
use qurious::provider::table::TableProvider;

use crate::ffi::*;
use crate::lower::{ExprArray, HipError, HipResult};

// ---------------------------------------------------------------- context
/// One `qhip_ctx` (one GPU, one stream). The reference executor is single-threaded; the mutex makes the handle `Sync` for
/// `QueryPlanner: Send + Sync` without promising concurrent execution.
pub struct HipContext {
    raw: *mut qhip_ctx,
    lock: Mutex<()>,
    /// device copies of scanned tables: a table is uploaded once, like `MemoryTable` keeps its batches in host memory
    /// (datasource/memory.rs:20-35) — and again whenever its DATA changes. `MemoryTable::insert` appends batches and
    /// `delete` rewrites or clears them (datasource/memory.rs:104-137), so a provider's address, batch count and row count
    /// can all stay the same over different data (DELETE everything, INSERT as many rows): the entry therefore remembers the
    /// batches it was made from and is valid only while the provider hands out the SAME arrays (`Arc::ptr_eq` on every
    /// column of every batch — the stored batches are immutable `Arc`s, and `scan` clones those `Arc`s); holding them keeps
    /// their addresses from being reused, and the `Weak` tells a dead provider from a new one at the same address.
    tables: Mutex<HashMap<usize, CachedTable>>,
}
struct CachedTable {
    provider: std::sync::Weak<dyn TableProvider>,
    batches: Vec<RecordBatch>,
    table: Arc<DeviceTable>,
}
/// the same stored data: as many batches, every batch the same rows and the very same column arrays
fn same_batches(a: &[RecordBatch], b: &[RecordBatch]) -> bool {
    a.len() == b.len()
        && a.iter().zip(b).all(|(x, y)| {
            x.num_rows() == y.num_rows()
                && x.num_columns() == y.num_columns()
                && x.columns().iter().zip(y.columns()).all(|(p, q)| Arc::ptr_eq(p, q))
        })
}
unsafe impl Send for HipContext {}
unsafe impl Sync for HipContext {}

impl std::fmt::Debug for HipContext {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "HipContext({:p})", self.raw)
    }
}

impl HipContext {
    /// Fails when no gfx950 device is visible: there is no CPU fallback inside libqhip (the planner falls back per node).
    pub fn new(device_index: i32) -> Result<Arc<Self>> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { qhip_ctx_create(device_index, &mut raw) };
        if rc != QHIP_OK {
            let msg = unsafe { CStr::from_ptr(qhip_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(Error::InternalError(msg));
        }
        Ok(Arc::new(Self { raw, lock: Mutex::new(()), tables: Mutex::new(HashMap::new()) }))
    }
    pub fn raw(&self) -> *mut qhip_ctx {
        self.raw
    }
    /// status code -> the shim's error: UNSUPPORTED lets the caller fall back, RETRY repeats the input (deferred sizes),
    /// everything else is the reference's `Error::InternalError(message)` (error.rs:20-31)
    pub fn check(&self, rc: i32) -> HipResult<()> {
        match rc {
            QHIP_OK => Ok(()),
            QHIP_RETRY => Err(HipError::Retry),
            _ => {
                let msg = unsafe { CStr::from_ptr(qhip_last_error(self.raw)) }.to_string_lossy().into_owned();
                if rc == QHIP_UNSUPPORTED {
                    Err(HipError::Unsupported(msg))
                } else {
                    Err(HipError::Failed(Error::InternalError(msg)))
                }
            }
        }
    }
    /// `Vec<RecordBatch>` -> HBM (qhip_table_from_arrow: one struct-typed ArrowArray per batch)
    pub fn upload(self: &Arc<Self>, schema: &SchemaRef, batches: &[RecordBatch]) -> HipResult<Arc<DeviceTable>> {
        let _guard = self.lock.lock().unwrap();
        let ffi_schema = FFI_ArrowSchema::try_from(schema.as_ref())?;
        let mut arrays: Vec<FFI_ArrowArray> = Vec::with_capacity(batches.len());
        for b in batches {
            let data = StructArray::from(b.clone()).into_data();
            let (array, _schema) = to_ffi(&data)?;
            arrays.push(array);
        }
        let ptrs: Vec<*const FFI_ArrowArray> = arrays.iter().map(|a| a as *const _).collect();
        let mut out = std::ptr::null_mut();
        self.check(unsafe { qhip_table_from_arrow(self.raw, &ffi_schema, ptrs.as_ptr(), ptrs.len() as i64, &mut out) })?;
        Ok(Arc::new(DeviceTable { ctx: self.clone(), raw: out }))
    }
    /// the device copy of a provider's batches: uploaded on first use and whenever the provider's data is no longer what the
    /// copy was made from (INSERT / DELETE / a new provider at a freed one's address) — see `tables`
    pub fn table_of(self: &Arc<Self>, source: &Arc<dyn TableProvider>) -> HipResult<Arc<DeviceTable>> {
        let batches = source.scan(None, None)?;
        let key = Arc::as_ptr(source) as *const () as usize;
        {
            let mut tables = self.tables.lock().unwrap();
            tables.retain(|_, e| e.provider.strong_count() > 0);   // dropped providers give their device tables back
            if let Some(e) = tables.get(&key) {
                let alive = e.provider.upgrade().map_or(false, |p| Arc::ptr_eq(&p, source));
                if alive && same_batches(&e.batches, &batches) {
                    return Ok(e.table.clone());
                }
            }
        }
        let t = self.upload(&source.schema(), &batches)?;
        let entry = CachedTable { provider: Arc::downgrade(source), batches, table: t.clone() };
        self.tables.lock().unwrap().insert(key, entry);
        Ok(t)
    }
    /// +1 around the execution of a child whose table goes straight into an aggregate / a join's build side: a hash join at
    /// its top may then leave its output size on the device (include/qhip.h: qhip_ctx_allow_deferred_sizes)
    pub fn allow_deferred_sizes(self: &Arc<Self>) -> DeferGuard {
        unsafe { qhip_ctx_allow_deferred_sizes(self.raw, 1) };
        DeferGuard { ctx: self.clone() }
    }
}
impl Drop for HipContext {
    fn drop(&mut self) {
        self.tables.lock().unwrap().clear();
        unsafe { qhip_ctx_destroy(self.raw) }
    }
}
pub struct DeferGuard {
    ctx: Arc<HipContext>,
}
impl Drop for DeferGuard {
    fn drop(&mut self) {
        unsafe { qhip_ctx_allow_deferred_sizes(self.ctx.raw, -1) };
    }
}

// ---------------------------------------------------------------- device tables
/// A `qhip_table`: the HBM-resident form of `Vec<RecordBatch>`.
pub struct DeviceTable {
    ctx: Arc<HipContext>,
    raw: *mut qhip_table,
}
unsafe impl Send for DeviceTable {}
unsafe impl Sync for DeviceTable {}
impl DeviceTable {
    pub fn raw(&self) -> *const qhip_table {
        self.raw
    }
    pub fn from_raw(ctx: &Arc<HipContext>, raw: *mut qhip_table) -> Arc<Self> {
        Arc::new(Self { ctx: ctx.clone(), raw })
    }
    /// download as `Vec<RecordBatch>` with the batch structure the reference's operator would have produced; `schema`
    /// re-attaches field names and the `qurious.field_qualifiers` metadata the C side is agnostic of
    /// (common/table_schema.rs:18,58-77; join merge rule physical/plan/join/mod.rs:84-122)
    pub fn download(&self, schema: &SchemaRef) -> HipResult<Vec<RecordBatch>> {
        let _guard = self.ctx.lock.lock().unwrap();
        let n = unsafe { qhip_table_num_batches(self.raw) };
        let mut out = Vec::with_capacity(n.max(0) as usize);
        for b in 0..n {
            let mut array = FFI_ArrowArray::empty();
            let mut ffi_schema = FFI_ArrowSchema::empty();
            self.ctx.check(unsafe { qhip_table_to_arrow(self.ctx.raw, self.raw, b, &mut array, &mut ffi_schema) })?;
            let data = unsafe { from_ffi(array, &ffi_schema) }?;
            let columns = StructArray::from(data).columns().to_vec();
            let fields: Vec<Field> = schema
                .fields()
                .iter()
                .zip(columns.iter())
                .map(|(f, c)| Field::new(f.name(), c.data_type().clone(), f.is_nullable() || c.null_count() > 0).with_metadata(f.metadata().clone()))
                .collect();
            let typed = Arc::new(Schema::new_with_metadata(fields, schema.metadata().clone()));
            out.push(RecordBatch::try_new(typed, columns)?);
        }
        Ok(out)
    }
}
impl Drop for DeviceTable {
    fn drop(&mut self) {
        unsafe { qhip_table_destroy(self.raw) }
    }
}

// ---------------------------------------------------------------- nodes
/// A node that can hand its result over on the device.
pub trait HipNode: PhysicalPlan + Send + Sync {
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>>;
    fn ctx(&self) -> &Arc<HipContext>;
    /// `Some((unfiltered table, predicate))` for a table scan with a pushed-down filter: an aggregate / an Inner join fuses the
    /// predicate into its kernel instead of materialising the filtered batches (datasource/memory.rs:90-93)
    fn fusable_scan(&self) -> Option<(HipResult<Arc<DeviceTable>>, &ExprArray, i32)> {
        None
    }
}

/// The input of a HIP node.
#[derive(Clone)]
pub enum Child {
    Hip(Arc<dyn HipNode>),
    Cpu(Arc<dyn PhysicalPlan>),
}
impl Child {
    pub fn schema(&self) -> SchemaRef {
        match self {
            Child::Hip(n) => n.schema(),
            Child::Cpu(p) => p.schema(),
        }
    }
    pub fn as_physical(&self) -> Arc<dyn PhysicalPlan> {
        match self {
            Child::Hip(n) => upcast(n.clone()),
            Child::Cpu(p) => p.clone(),
        }
    }
    fn execute_device(&self, ctx: &Arc<HipContext>) -> HipResult<Arc<DeviceTable>> {
        match self {
            Child::Hip(n) => n.execute_device(),
            Child::Cpu(p) => ctx.upload(&p.schema(), &p.execute()?),
        }
    }
}
/// `Arc<dyn HipNode>` -> `Arc<dyn PhysicalPlan>` (a supertrait upcast, spelled out for compilers without trait upcasting)
pub fn upcast(n: Arc<dyn HipNode>) -> Arc<dyn PhysicalPlan> {
    struct AsPlan(Arc<dyn HipNode>);
    impl PhysicalPlan for AsPlan {
        fn schema(&self) -> SchemaRef {
            self.0.schema()
        }
        fn execute(&self) -> Result<Vec<RecordBatch>> {
            self.0.execute()
        }
        fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
            self.0.children()
        }
    }
    Arc::new(AsPlan(n))
}

fn finish(node: &dyn HipNode) -> Result<Vec<RecordBatch>> {
    match node.execute_device().and_then(|t| t.download(&node.schema())) {
        Ok(batches) => Ok(batches),
        Err(HipError::Failed(e)) => Err(e),
        Err(HipError::Unsupported(msg)) => Err(Error::InternalError(format!("qurious-hip: {msg} (found at execution time)"))),
        Err(HipError::Retry) => Err(Error::InternalError("qurious-hip: QHIP_RETRY left the shim (internal error)".into())),
    }
}

/// run `f` (execute the input(s) with deferral allowed, then the operator); QHIP_RETRY = a join of deferred size below had
/// too little room and has dropped its hint: run again, it will wait this time
fn retrying<T>(mut f: impl FnMut() -> HipResult<T>) -> HipResult<T> {
    for attempt in 0..3 {
        match f() {
            Err(HipError::Retry) if attempt < 2 => continue,
            other => return other,
        }
    }
    unreachable!()
}

fn names_of(schema: &SchemaRef) -> (Vec<std::ffi::CString>, Vec<*const c_char>) {
    let owned: Vec<std::ffi::CString> = schema.fields().iter().map(|f| std::ffi::CString::new(f.name().replace('\0', "")).unwrap()).collect();
    let ptrs = owned.iter().map(|s| s.as_ptr()).collect();
    (owned, ptrs)
}

// ---- Scan (physical/plan/scan.rs:12-47) with its pushed-down filter (datasource/memory.rs:69-98)
pub struct HipScan {
    pub ctx: Arc<HipContext>,
    pub schema: SchemaRef,
    pub source: Arc<dyn TableProvider>,
    pub exprs: ExprArray,
    pub predicate: i32, // root in `exprs`, -1 = no filter
}
impl PhysicalPlan for HipScan {
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        None
    }
}
impl HipNode for HipScan {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let table = self.ctx.table_of(&self.source)?;
        if self.predicate < 0 {
            return Ok(table);
        }
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe {
            qhip_filter_execute(self.ctx.raw(), table.raw(), self.exprs.as_ptr(), self.exprs.len(), self.predicate, std::ptr::null(), -1, &mut out)
        })?;
        Ok(DeviceTable::from_raw(&self.ctx, out))
    }
    fn fusable_scan(&self) -> Option<(HipResult<Arc<DeviceTable>>, &ExprArray, i32)> {
        (self.predicate >= 0).then(|| (self.ctx.table_of(&self.source), &self.exprs, self.predicate))
    }
}

// ---- Filter (physical/plan/filter.rs:12-48)
pub struct HipFilter {
    pub ctx: Arc<HipContext>,
    pub input: Child,
    pub exprs: ExprArray,
    pub predicate: i32,
}
impl PhysicalPlan for HipFilter {
    fn schema(&self) -> SchemaRef {
        self.input.schema()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        Some(vec![self.input.as_physical()])
    }
}
impl HipNode for HipFilter {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let table = self.input.execute_device(&self.ctx)?;
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe {
            qhip_filter_execute(self.ctx.raw(), table.raw(), self.exprs.as_ptr(), self.exprs.len(), self.predicate, std::ptr::null(), -1, &mut out)
        })?;
        Ok(DeviceTable::from_raw(&self.ctx, out))
    }
}

// ---- HashAggregate / NoGroupingAggregate (physical/plan/aggregate/hash.rs:110-176, no_grouping.rs:9-66)
pub struct HipAggregate {
    pub ctx: Arc<HipContext>,
    pub schema: SchemaRef,
    pub input: Child,
    /// group keys and aggregate arguments over the input schema; when the input is a fusable scan its predicate is
    /// appended to this array at plan time and `fused_predicate` is its root
    pub exprs: ExprArray,
    pub fused_predicate: i32,
    pub groups: Vec<i32>,
    pub aggs: Vec<qhip_agg>,
}
impl PhysicalPlan for HipAggregate {
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        if self.groups.is_empty() {
            None // no_grouping.rs:63-65
        } else {
            Some(vec![self.input.as_physical()])
        }
    }
}
impl HipNode for HipAggregate {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let (_names, name_ptrs) = names_of(&self.schema);
        retrying(|| {
            let table = match (&self.input, self.fused_predicate >= 0) {
                (Child::Hip(n), true) => n.fusable_scan().expect("planned as a fused scan").0?,
                (child, _) => {
                    let _defer = self.ctx.allow_deferred_sizes(); // the aggregate kernel reads a device-side row count
                    child.execute_device(&self.ctx)?
                }
            };
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe {
                qhip_hash_aggregate_execute(
                    self.ctx.raw(),
                    table.raw(),
                    self.exprs.as_ptr(),
                    self.exprs.len(),
                    self.fused_predicate,
                    self.groups.as_ptr(),
                    self.groups.len() as i32,
                    self.aggs.as_ptr(),
                    self.aggs.len() as i32,
                    name_ptrs.as_ptr(),
                    &mut out,
                )
            })?;
            Ok(DeviceTable::from_raw(&self.ctx, out))
        })
    }
}

// ---- HashJoinExec (physical/plan/join/hash_join.rs:110-384): build = left, probe = right
pub struct HipHashJoin {
    pub ctx: Arc<HipContext>,
    pub schema: SchemaRef,
    pub left: Child,
    pub right: Child,
    pub join_type: JoinType,
    pub left_exprs: ExprArray,
    pub right_exprs: ExprArray,
    pub on_left: Vec<i32>,
    pub on_right: Vec<i32>,
    pub filter: ExprArray,
    pub filter_root: i32, // -1 = no residual JoinFilter
    pub filter_sides: Vec<i32>,
    pub filter_cols: Vec<i32>,
    /// fused scan filters of Inner joins: roots in left_exprs / right_exprs, -1 = the side is executed as it is
    pub left_scan_filter: i32,
    pub right_scan_filter: i32,
}
/// common/join_type.rs:4-11 — qhip_join_type has the same order
pub fn join_type_code(t: JoinType) -> i32 {
    match t {
        JoinType::Left => 0,
        JoinType::Right => 1,
        JoinType::Inner => 2,
        JoinType::Full => 3,
        JoinType::LeftSemi => 4,
        JoinType::LeftAnti => 5,
    }
}
impl PhysicalPlan for HipHashJoin {
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        Some(vec![self.left.as_physical(), self.right.as_physical()])
    }
}
impl HipHashJoin {
    fn side(&self, child: &Child, fused: i32, may_defer: bool) -> HipResult<Arc<DeviceTable>> {
        match (child, fused >= 0) {
            (Child::Hip(n), true) => n.fusable_scan().expect("planned as a fused scan").0,
            (c, _) if may_defer => {
                let _defer = self.ctx.allow_deferred_sizes();
                c.execute_device(&self.ctx)
            }
            (c, _) => c.execute_device(&self.ctx),
        }
    }
}
impl HipNode for HipHashJoin {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        // the build side may arrive with its row count on the device when nothing executes between it and this join, i.e.
        // when the probe side is a table access (the reference's order — left first — is kept)
        let right_is_table = matches!(&self.right, Child::Hip(n) if n.children().is_none() && (n.fusable_scan().is_none() || self.right_scan_filter >= 0));
        retrying(|| {
            let left = self.side(&self.left, self.left_scan_filter, right_is_table)?;
            let right = self.side(&self.right, self.right_scan_filter, false)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe {
                qhip_hash_join_execute(
                    self.ctx.raw(),
                    left.raw(),
                    right.raw(),
                    join_type_code(self.join_type),
                    self.left_exprs.as_ptr(),
                    self.left_exprs.len(),
                    self.right_exprs.as_ptr(),
                    self.right_exprs.len(),
                    self.on_left.as_ptr(),
                    self.on_right.as_ptr(),
                    self.on_left.len() as i32,
                    self.filter.as_ptr(),
                    self.filter.len(),
                    self.filter_root,
                    self.filter_sides.as_ptr(),
                    self.filter_cols.as_ptr(),
                    self.filter_cols.len() as i32,
                    self.left_scan_filter,
                    self.right_scan_filter,
                    &mut out,
                )
            })?;
            Ok(DeviceTable::from_raw(&self.ctx, out))
        })
    }
}

// ---- NestedLoopJoinExec / CrossJoin (physical/plan/join/nest_loop_join.rs:42-228, cross_join.rs:56-170)
pub struct HipNestedLoopJoin {
    pub ctx: Arc<HipContext>,
    pub schema: SchemaRef,
    pub left: Child,
    pub right: Child,
    pub join_type: JoinType,
    pub filter: ExprArray,
    pub filter_root: i32,
    pub filter_sides: Vec<i32>,
    pub filter_cols: Vec<i32>,
    pub cross: bool,
}
impl PhysicalPlan for HipNestedLoopJoin {
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        Some(vec![self.left.as_physical(), self.right.as_physical()])
    }
}
impl HipNode for HipNestedLoopJoin {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let left = self.left.execute_device(&self.ctx)?;
        let right = self.right.execute_device(&self.ctx)?;
        let mut out = std::ptr::null_mut();
        let rc = if self.cross {
            unsafe { qhip_cross_join_execute(self.ctx.raw(), left.raw(), right.raw(), &mut out) }
        } else {
            unsafe {
                qhip_nested_loop_join_execute(
                    self.ctx.raw(),
                    left.raw(),
                    right.raw(),
                    join_type_code(self.join_type),
                    self.filter.as_ptr(),
                    self.filter.len(),
                    self.filter_root,
                    self.filter_sides.as_ptr(),
                    self.filter_cols.as_ptr(),
                    self.filter_cols.len() as i32,
                    &mut out,
                )
            }
        };
        self.ctx.check(rc)?;
        Ok(DeviceTable::from_raw(&self.ctx, out))
    }
}

// ---- Projection (physical/plan/projection.rs:10-51)
pub struct HipProjection {
    pub ctx: Arc<HipContext>,
    pub schema: SchemaRef,
    pub input: Child,
    pub exprs: ExprArray,
    pub roots: Vec<i32>,
}
impl PhysicalPlan for HipProjection {
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        Some(vec![self.input.as_physical()])
    }
}
impl HipNode for HipProjection {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let table = self.input.execute_device(&self.ctx)?;
        let (_names, name_ptrs) = names_of(&self.schema);
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe {
            qhip_projection_execute(
                self.ctx.raw(),
                table.raw(),
                self.exprs.as_ptr(),
                self.exprs.len(),
                self.roots.as_ptr(),
                self.roots.len() as i32,
                name_ptrs.as_ptr(),
                &mut out,
            )
        })?;
        Ok(DeviceTable::from_raw(&self.ctx, out))
    }
}

// ---- Sort (+ top-N) and Limit (physical/plan/sort.rs:23-86, limit.rs:10-62)
pub struct HipSort {
    pub ctx: Arc<HipContext>,
    pub input: Child,
    pub exprs: ExprArray,
    pub keys: Vec<i32>,
    pub descending: Vec<i32>,
    pub nulls_first: Vec<i32>,
    pub limit: i64, // Sort::new_with_limit's top-N, -1 = none
}
impl PhysicalPlan for HipSort {
    fn schema(&self) -> SchemaRef {
        self.input.schema()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        self.input.as_physical().children() // sort.rs:83-85
    }
}
impl HipNode for HipSort {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let table = self.input.execute_device(&self.ctx)?;
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe {
            qhip_sort_execute(
                self.ctx.raw(),
                table.raw(),
                self.exprs.as_ptr(),
                self.exprs.len(),
                self.keys.as_ptr(),
                self.descending.as_ptr(),
                self.nulls_first.as_ptr(),
                self.keys.len() as i32,
                self.limit,
                &mut out,
            )
        })?;
        Ok(DeviceTable::from_raw(&self.ctx, out))
    }
}

pub struct HipLimit {
    pub ctx: Arc<HipContext>,
    pub input: Child,
    pub skip: i64,
    pub fetch: i64, // -1 = none
}
impl PhysicalPlan for HipLimit {
    fn schema(&self) -> SchemaRef {
        self.input.schema()
    }
    fn execute(&self) -> Result<Vec<RecordBatch>> {
        finish(self)
    }
    fn children(&self) -> Option<Vec<Arc<dyn PhysicalPlan>>> {
        self.input.as_physical().children() // limit.rs:59-61
    }
}
impl HipNode for HipLimit {
    fn ctx(&self) -> &Arc<HipContext> {
        &self.ctx
    }
    fn execute_device(&self) -> HipResult<Arc<DeviceTable>> {
        let table = self.input.execute_device(&self.ctx)?;
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { qhip_limit_execute(self.ctx.raw(), table.raw(), self.skip, self.fetch, &mut out) })?;
        Ok(DeviceTable::from_raw(&self.ctx, out))
    }
}
