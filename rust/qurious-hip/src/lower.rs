//! Lowering: the reference's `LogicalExpr` trees -> the flat `qhip_expr` arrays of the C ABI (include/qhip.h), the way
//! `DefaultQueryPlanner::create_physical_expr` turns them into `Arc<dyn PhysicalExpr>` (qurious/src/planner/mod.rs:102-153,
//! 352-400). The physical expression objects themselves cannot be lowered: `trait PhysicalExpr` (physical/expr/mod.rs:33-35)
//! has no `as_any`, so the shim starts from the logical plan, where every node is a public enum.
//!
//! Anything the backend does not accelerate (`Function`, sub-queries, `Decimal256`, ...) is `HipError::Unsupported`: the
//! planner then builds the reference's CPU node for that part of the plan.
use std::ffi::CString;
use std::os::raw::c_char;
use std::sync::Arc;

use arrow::datatypes::{DataType, Schema, SchemaRef, TimeUnit};
use qurious::common::table_schema::FIELD_QUALIFIERS_META_KEY;
use qurious::datatypes::operator::Operator;
use qurious::datatypes::scalar::ScalarValue;
use qurious::error::Error;
use qurious::logical::expr::{AggregateExpr, AggregateOperator, BinaryExpr, Column, LogicalExpr};

use crate::ffi::*;

/// What can go wrong between the logical plan and a device result.
#[derive(Debug)]
pub enum HipError {
    /// a valid plan the HIP backend does not accelerate: build the CPU node instead (QHIP_UNSUPPORTED, or found while lowering)
    Unsupported(String),
    /// the consumer of a hash join of deferred size must execute its input again (QHIP_RETRY; never leaves the shim)
    Retry,
    /// everything else, already in the reference's error type (error.rs:42-53)
    Failed(Error),
}
pub type HipResult<T> = std::result::Result<T, HipError>;

impl From<Error> for HipError {
    fn from(e: Error) -> Self {
        HipError::Failed(e)
    }
}
impl From<arrow::error::ArrowError> for HipError {
    fn from(e: arrow::error::ArrowError) -> Self {
        HipError::Failed(Error::ArrowError(e, None))
    }
}
pub fn unsupported<T>(what: impl Into<String>) -> HipResult<T> {
    Err(HipError::Unsupported(what.into()))
}

/// arrow DataType -> qhip_dtype (the subset of include/qhip.h `qhip_type_id`)
pub fn dtype_of(t: &DataType) -> HipResult<qhip_dtype> {
    let id = match t {
        DataType::Null => QHIP_NULL,
        DataType::Boolean => QHIP_BOOL,
        DataType::Int8 => QHIP_INT8,
        DataType::Int16 => QHIP_INT16,
        DataType::Int32 => QHIP_INT32,
        DataType::Int64 => QHIP_INT64,
        DataType::UInt8 => QHIP_UINT8,
        DataType::UInt16 => QHIP_UINT16,
        DataType::UInt32 => QHIP_UINT32,
        DataType::UInt64 => QHIP_UINT64,
        DataType::Float32 => QHIP_FLOAT32,
        DataType::Float64 => QHIP_FLOAT64,
        DataType::Date32 => QHIP_DATE32,
        DataType::Date64 => QHIP_DATE64,
        DataType::Utf8 => QHIP_UTF8,
        DataType::Time32(TimeUnit::Second) => QHIP_TIME32_S,
        DataType::Time32(TimeUnit::Millisecond) => QHIP_TIME32_MS,
        DataType::Time64(TimeUnit::Microsecond) => QHIP_TIME64_US,
        DataType::Time64(TimeUnit::Nanosecond) => QHIP_TIME64_NS,
        // Timestamp(unit, None): MIN / MAX / comparison / sort keys (physical/expr/aggregate/mod.rs:108-111); with a timezone: CPU node
        DataType::Timestamp(TimeUnit::Second, None) => QHIP_TIMESTAMP_S,
        DataType::Timestamp(TimeUnit::Millisecond, None) => QHIP_TIMESTAMP_MS,
        DataType::Timestamp(TimeUnit::Microsecond, None) => QHIP_TIMESTAMP_US,
        DataType::Timestamp(TimeUnit::Nanosecond, None) => QHIP_TIMESTAMP_NS,
        DataType::Decimal128(p, s) => {
            return Ok(qhip_dtype { id: QHIP_DECIMAL128, precision: *p as i32, scale: *s as i32 });
        }
        other => return unsupported(format!("data type {other} is not accelerated")),
    };
    Ok(qhip_dtype { id, precision: 0, scale: 0 })
}

/// qhip_operator has the order of datatypes/operator.rs:4-20
pub fn operator_code(op: &Operator) -> i32 {
    match op {
        Operator::Eq => 0,
        Operator::NotEq => 1,
        Operator::Gt => 2,
        Operator::GtEq => 3,
        Operator::Lt => 4,
        Operator::LtEq => 5,
        Operator::And => 6,
        Operator::Or => 7,
        Operator::Add => 8,
        Operator::Sub => 9,
        Operator::Mul => 10,
        Operator::Div => 11,
        Operator::Mod => 12,
    }
}

/// Index of a column in a physical schema: qualified lookup through the `qurious.field_qualifiers` metadata first, then by
/// name — `DefaultQueryPlanner::physical_expr_column` (planner/mod.rs:359-383).
pub fn resolve_column(schema: &Schema, column: &Column) -> HipResult<usize> {
    if let (Some(rel), Some(qualifiers)) = (column.relation.as_ref(), schema.metadata().get(FIELD_QUALIFIERS_META_KEY)) {
        let rel_name = rel.to_qualified_name();
        let qualifiers: Vec<&str> = qualifiers.split('\u{1f}').collect();
        if qualifiers.len() == schema.fields().len() {
            if let Some((index, _)) = schema.fields().iter().enumerate().find(|(i, f)| f.name() == &column.name && qualifiers[*i] == rel_name) {
                return Ok(index);
            }
        }
    }
    Ok(schema.index_of(&column.name)?)
}

fn blank(kind: i32) -> qhip_expr {
    qhip_expr {
        kind,
        op: 0,
        column: -1,
        left: -1,
        right: -1,
        third: -1,
        dtype: qhip_dtype::default(),
        lit_is_null: 0,
        lit_lo: 0,
        lit_hi: 0,
        lit_f64: 0.0,
        lit_str: std::ptr::null(),
        lit_len: 0,
    }
}

/// A flat expression array under construction. String literals are owned here; the raw pointers inside `nodes` stay valid
/// as long as the `ExprArray` lives (a node keeps its arrays for its whole life: plans are immutable after construction).
#[derive(Debug, Default)]
pub struct ExprArray {
    pub nodes: Vec<qhip_expr>,
    strings: Vec<CString>,
}

// the raw pointers point into `strings`, which moves with the struct
unsafe impl Send for ExprArray {}
unsafe impl Sync for ExprArray {}

impl ExprArray {
    pub fn new() -> Self {
        Self::default()
    }
    pub fn as_ptr(&self) -> *const qhip_expr {
        if self.nodes.is_empty() {
            std::ptr::null()
        } else {
            self.nodes.as_ptr()
        }
    }
    pub fn len(&self) -> i32 {
        self.nodes.len() as i32
    }
    pub fn is_empty(&self) -> bool {
        self.nodes.is_empty()
    }
    fn push(&mut self, n: qhip_expr) -> i32 {
        self.nodes.push(n);
        (self.nodes.len() - 1) as i32
    }

    pub fn column(&mut self, index: usize) -> i32 {
        let mut n = blank(QHIP_EXPR_COLUMN);
        n.column = index as i32;
        self.push(n)
    }

    /// ScalarValue -> a literal node (physical/expr/literal.rs:20-22; datatypes/scalar.rs:85-107)
    pub fn literal(&mut self, v: &ScalarValue) -> HipResult<i32> {
        let mut n = blank(QHIP_EXPR_LITERAL);
        fn int(n: &mut qhip_expr, id: i32, v: Option<i64>) {
            n.dtype.id = id;
            match v {
                Some(x) => n.lit_lo = x as u64,
                None => n.lit_is_null = 1,
            }
        }
        match v {
            ScalarValue::Null => {
                n.dtype.id = QHIP_NULL;
                n.lit_is_null = 1;
            }
            ScalarValue::Boolean(b) => int(&mut n, QHIP_BOOL, b.map(|x| x as i64)),
            ScalarValue::Int8(x) => int(&mut n, QHIP_INT8, x.map(|x| x as i64)),
            ScalarValue::Int16(x) => int(&mut n, QHIP_INT16, x.map(|x| x as i64)),
            ScalarValue::Int32(x) => int(&mut n, QHIP_INT32, x.map(|x| x as i64)),
            ScalarValue::Int64(x) => int(&mut n, QHIP_INT64, *x),
            ScalarValue::UInt8(x) => int(&mut n, QHIP_UINT8, x.map(|x| x as i64)),
            ScalarValue::UInt16(x) => int(&mut n, QHIP_UINT16, x.map(|x| x as i64)),
            ScalarValue::UInt32(x) => int(&mut n, QHIP_UINT32, x.map(|x| x as i64)),
            ScalarValue::UInt64(x) => int(&mut n, QHIP_UINT64, x.map(|x| x as i64)),
            ScalarValue::Float32(x) => {
                n.dtype.id = QHIP_FLOAT32;
                match x {
                    Some(f) => n.lit_f64 = *f as f64,
                    None => n.lit_is_null = 1,
                }
            }
            ScalarValue::Float64(x) => {
                n.dtype.id = QHIP_FLOAT64;
                match x {
                    Some(f) => n.lit_f64 = *f,
                    None => n.lit_is_null = 1,
                }
            }
            ScalarValue::Decimal128(x, p, s) => {
                n.dtype = qhip_dtype { id: QHIP_DECIMAL128, precision: *p as i32, scale: *s as i32 };
                match x {
                    Some(d) => {
                        n.lit_lo = *d as u64;
                        n.lit_hi = (*d >> 64) as i64;
                    }
                    None => n.lit_is_null = 1,
                }
            }
            ScalarValue::Utf8(x) => {
                n.dtype.id = QHIP_UTF8;
                match x {
                    Some(text) => {
                        n.lit_len = text.len() as i64;
                        // (lit_len carries the length: interior NULs are legal, the terminator is only for CString)
                        let owned = CString::new(text.replace('\0', "")).map_err(|e| HipError::Failed(Error::InternalError(e.to_string())))?;
                        if owned.as_bytes().len() != text.len() {
                            return unsupported("Utf8 literal with an interior NUL byte");
                        }
                        n.lit_str = owned.as_ptr() as *const c_char;
                        self.strings.push(owned);
                    }
                    None => n.lit_is_null = 1,
                }
            }
            other => return unsupported(format!("literal {other:?} is not accelerated")),
        }
        Ok(self.push(n))
    }

    fn unary(&mut self, kind: i32, child: i32) -> i32 {
        let mut n = blank(kind);
        n.left = child;
        self.push(n)
    }

    /// Lower `expr` over `schema` (the operator's input schema, qualifier metadata included); returns the root index.
    pub fn lower(&mut self, expr: &LogicalExpr, schema: &SchemaRef) -> HipResult<i32> {
        match expr {
            LogicalExpr::Column(c) => Ok(self.column(resolve_column(schema, c)?)),
            LogicalExpr::Literal(v) => self.literal(v),
            LogicalExpr::Alias(a) => self.lower(&a.expr, schema),
            LogicalExpr::BinaryExpr(BinaryExpr { left, op, right }) => {
                let l = self.lower(left, schema)?;
                let r = self.lower(right, schema)?;
                let mut n = blank(QHIP_EXPR_BINARY);
                n.op = operator_code(op);
                n.left = l;
                n.right = r;
                Ok(self.push(n))
            }
            LogicalExpr::Cast(c) => {
                let child = self.lower(&c.expr, schema)?;
                let mut n = blank(QHIP_EXPR_CAST);
                n.left = child;
                n.dtype = dtype_of(&c.data_type)?;
                Ok(self.push(n))
            }
            LogicalExpr::IsNull(e) => {
                let child = self.lower(e, schema)?;
                Ok(self.unary(QHIP_EXPR_IS_NULL, child))
            }
            LogicalExpr::IsNotNull(e) => {
                let child = self.lower(e, schema)?;
                Ok(self.unary(QHIP_EXPR_IS_NOT_NULL, child))
            }
            LogicalExpr::Negative(e) => {
                let child = self.lower(e, schema)?;
                Ok(self.unary(QHIP_EXPR_NEGATIVE, child))
            }
            // an aggregate referenced above its Aggregate node is the column the node produced (planner/mod.rs:109)
            LogicalExpr::AggregateExpr(a) => {
                let as_column = a.as_column()?;
                self.lower(&as_column, schema)
            }
            // CASE [operand] WHEN c1 THEN v1 ... ELSE e END -> IF(c1, v1, IF(c2, v2, e)), folded from the ELSE branch up like
            // physical/expr/case.rs:36-46; a simple CASE compares the operand with every WHEN value (planner/mod.rs:116-131)
            LogicalExpr::Case(case) => {
                let mut acc = self.lower(&case.else_expr, schema)?;
                for (when, then) in case.when_then.iter().rev() {
                    let cond = match &case.operand {
                        Some(operand) => {
                            let l = self.lower(operand, schema)?;
                            let r = self.lower(when, schema)?;
                            let mut eq = blank(QHIP_EXPR_BINARY);
                            eq.op = operator_code(&Operator::Eq);
                            eq.left = l;
                            eq.right = r;
                            self.push(eq)
                        }
                        None => self.lower(when, schema)?,
                    };
                    let value = self.lower(then, schema)?;
                    let mut n = blank(QHIP_EXPR_IF);
                    n.left = cond;
                    n.right = value;
                    n.third = acc;
                    acc = self.push(n);
                }
                Ok(acc)
            }
            LogicalExpr::Like(like) => {
                // a literal pattern or a pattern COLUMN (matched per row, like.rs:28-43); a computed pattern stays with the CPU node
                if !matches!(like.pattern.as_ref(), LogicalExpr::Literal(ScalarValue::Utf8(Some(_))) | LogicalExpr::Column(_)) {
                    return unsupported("LIKE with a computed pattern");
                }
                let value = self.lower(&like.expr, schema)?;
                let pattern = self.lower(&like.pattern, schema)?;
                let mut n = blank(QHIP_EXPR_LIKE);
                n.left = value;
                n.right = pattern;
                n.op = like.negated as i32;
                Ok(self.push(n))
            }
            other => unsupported(format!("expression {other} is not accelerated")),
        }
    }
}

/// One aggregate of an `Aggregate` node: the argument's root in `exprs` + kind + return type, as
/// `DefaultQueryPlanner::physical_plan_aggregate` derives them (planner/mod.rs:190-224; logical/expr/aggregate.rs:66-90).
pub fn lower_aggregate(exprs: &mut ExprArray, e: &LogicalExpr, input_schema: &SchemaRef) -> HipResult<qhip_agg> {
    let LogicalExpr::AggregateExpr(AggregateExpr { op, expr }) = e else {
        return Err(HipError::Failed(Error::InternalError(format!("LogicalExpr should be AggregateExpr, but got {e:?}"))));
    };
    let root = exprs.lower(expr, input_schema)?;
    let return_type = dtype_of(&e.data_type(input_schema)?)?;
    let kind = match op {
        AggregateOperator::Sum => QHIP_AGG_SUM,
        AggregateOperator::Avg => QHIP_AGG_AVG,
        AggregateOperator::Count => QHIP_AGG_COUNT,
        AggregateOperator::Min => QHIP_AGG_MIN,
        AggregateOperator::Max => QHIP_AGG_MAX,
    };
    Ok(qhip_agg { kind, expr: root, return_type })
}

/// helper for nodes that keep a schema next to their lowered arrays
pub fn schema_ref(s: &Schema) -> SchemaRef {
    Arc::new(s.clone())
}
