//! `HipQueryPlanner: QueryPlanner` — the reference's plug-in seam (qurious/src/planner/mod.rs:30-34). It walks the optimized
//! `LogicalPlan` like `DefaultQueryPlanner` does (planner/mod.rs:40-100, 174-348) but builds HIP nodes for the hot path —
//! table scan + pushed-down filter, Filter, Aggregate, equi / non-equi / cross joins, Projection, Sort (+ top-N) and Limit —
//! and the reference's own CPU nodes for everything else or whenever a sub-plan holds something libqhip does not accelerate.
//!
//! Injection: `DataFrame::new(plan, Arc::new(HipQueryPlanner::new()?))` (dataframe/mod.rs:21) needs no patch;
//! `ExecuteSession` hard-wires `DefaultQueryPlanner` (execution/session.rs:35,66): a five-line `with_planner` there makes
//! `session.sql(..)` use this planner too.
use std::sync::Arc;

use arrow::datatypes::SchemaRef;
use qurious::common::join_type::JoinType;
use qurious::error::{Error, Result};
use qurious::logical::expr::{LogicalExpr, SortExpr};
use qurious::logical::plan::{Aggregate, CrossJoin, Filter, Join, Limit, LogicalPlan, Projection, Sort, TableScan};
use qurious::physical::expr::PhysicalExpr;
use qurious::physical::plan::{JoinSide, PhysicalPlan};
use qurious::planner::{DefaultQueryPlanner, QueryPlanner};

use crate::lower::{lower_aggregate, unsupported, ExprArray, HipError, HipResult};
use crate::plan::*;

#[derive(Debug)]
pub struct HipQueryPlanner {
    ctx: Arc<HipContext>,
    cpu: DefaultQueryPlanner,
}

impl HipQueryPlanner {
    /// One context on the current HIP device; errors when no gfx950 device is visible.
    pub fn new() -> Result<Self> {
        Ok(Self { ctx: HipContext::new(-1)?, cpu: DefaultQueryPlanner })
    }
    pub fn with_context(ctx: Arc<HipContext>) -> Self {
        Self { ctx, cpu: DefaultQueryPlanner }
    }

    /// A HIP node for `plan` when its operator (and expressions) are accelerated, else the reference's CPU node. Children
    /// are planned the same way, so a CPU node may sit between HIP nodes (its batches are uploaded again above it).
    fn plan(&self, plan: &LogicalPlan) -> Result<Child> {
        let hip = match plan {
            LogicalPlan::TableScan(t) => self.scan(t),
            LogicalPlan::Filter(f) => self.filter(f),
            LogicalPlan::Aggregate(a) => self.aggregate(a),
            LogicalPlan::Join(j) if !j.on.is_empty() => self.hash_join(j),
            LogicalPlan::Join(j) => self.nested_loop_join(j),
            LogicalPlan::CrossJoin(c) => self.cross_join(c),
            LogicalPlan::Projection(p) => self.projection(p),
            LogicalPlan::Sort(s) => self.sort(s, None),
            LogicalPlan::Limit(l) => self.limit(l),
            _ => unsupported("plan node without a HIP counterpart"),
        };
        match hip {
            Ok(node) => Ok(Child::Hip(node)),
            Err(HipError::Unsupported(_)) | Err(HipError::Retry) => Ok(Child::Cpu(self.cpu.create_physical_plan(plan)?)),
            Err(HipError::Failed(e)) => Err(e),
        }
    }

    fn child(&self, plan: &LogicalPlan) -> HipResult<Child> {
        Ok(self.plan(plan)?)
    }

    // ---- physical_plan_table_scan (planner/mod.rs:244-257)
    fn scan(&self, t: &TableScan) -> HipResult<Arc<dyn HipNode>> {
        let schema = t.schema();
        let mut exprs = ExprArray::new();
        let predicate = match &t.filter {
            Some(f) => exprs.lower(f, &schema)?,
            None => -1,
        };
        Ok(Arc::new(HipScan { ctx: self.ctx.clone(), schema, source: t.source.clone(), exprs, predicate }))
    }

    // ---- physical_plan_filter (planner/mod.rs:174-179)
    fn filter(&self, f: &Filter) -> HipResult<Arc<dyn HipNode>> {
        let input = self.child(&f.input)?;
        let mut exprs = ExprArray::new();
        let predicate = exprs.lower(&f.expr, &f.schema())?;
        Ok(Arc::new(HipFilter { ctx: self.ctx.clone(), input, exprs, predicate }))
    }

    /// the pushed-down filter of a scan that a parent may fuse: `TableScan { filter }`, or `Filter(TableScan)` before the
    /// PushdownFilter rule ran (optimizer/rule/pushdown_filter.rs:75-83)
    fn fusable_filter<'a>(plan: &'a LogicalPlan) -> Option<(&'a TableScan, &'a LogicalExpr)> {
        match plan {
            LogicalPlan::TableScan(t) => t.filter.as_ref().map(|f| (t, f)),
            LogicalPlan::Filter(f) => match f.input.as_ref() {
                LogicalPlan::TableScan(t) if t.filter.is_none() => Some((t, &f.expr)),
                _ => None,
            },
            _ => None,
        }
    }

    // ---- physical_plan_aggregate (planner/mod.rs:181-242): HashAggregate iff there are group expressions
    fn aggregate(&self, a: &Aggregate) -> HipResult<Arc<dyn HipNode>> {
        let input_schema = a.input.schema();
        let mut exprs = ExprArray::new();
        let groups = a.group_expr.iter().map(|e| exprs.lower(e, &input_schema)).collect::<HipResult<Vec<_>>>()?;
        let aggs = a.aggr_expr.iter().map(|e| lower_aggregate(&mut exprs, e, &input_schema)).collect::<HipResult<Vec<_>>>()?;
        // Scan(filter) below: one fused pass over the referenced columns, nothing materialised in between
        let (input, fused_predicate) = match Self::fusable_filter(&a.input) {
            Some((scan, predicate)) => {
                let root = exprs.lower(predicate, &scan.schema())?;
                let leaf = TableScan { filter: Some(predicate.clone()), ..scan.clone() };
                (Child::Hip(self.scan(&leaf)?), root)
            }
            None => (self.child(&a.input)?, -1),
        };
        Ok(Arc::new(HipAggregate { ctx: self.ctx.clone(), schema: a.schema.arrow_schema(), input, exprs, fused_predicate, groups, aggs }))
    }

    /// JoinFilter { expr, schema, column_indices } from the filter's columns (planner/mod.rs:268-295): the intermediate
    /// schema is [left columns used..., right columns used...]
    fn join_filter(filter: &Option<LogicalExpr>, left: &SchemaRef, right: &SchemaRef) -> HipResult<(ExprArray, i32, Vec<i32>, Vec<i32>)> {
        let mut exprs = ExprArray::new();
        let Some(filter) = filter else {
            return Ok((exprs, -1, vec![], vec![]));
        };
        let using = filter.using_columns();
        let mut fields = Vec::new();
        let (mut sides, mut cols) = (Vec::new(), Vec::new());
        for (schema, side) in [(left, JoinSide::Left), (right, JoinSide::Right)] {
            for c in using.iter() {
                if let Some((i, f)) = schema.fields().find(&c.name) {
                    fields.push(f.clone());
                    sides.push(if matches!(side, JoinSide::Left) { 0 } else { 1 });
                    cols.push(i as i32);
                }
            }
        }
        let filter_schema: SchemaRef = Arc::new(arrow::datatypes::Schema::new(fields));
        let root = exprs.lower(filter, &filter_schema)?;
        Ok((exprs, root, sides, cols))
    }

    // ---- physical_plan_join with equi-keys -> HashJoinExec (planner/mod.rs:265-315)
    fn hash_join(&self, j: &Join) -> HipResult<Arc<dyn HipNode>> {
        let (ls, rs) = (j.left.schema(), j.right.schema());
        qurious::planner::check_join_is_valid(&ls, &rs, &j.on)?;
        let (mut left_exprs, mut right_exprs) = (ExprArray::new(), ExprArray::new());
        let mut on_left = Vec::new();
        let mut on_right = Vec::new();
        for (l, r) in j.on.iter() {
            on_left.push(left_exprs.lower(l, &ls)?);
            on_right.push(right_exprs.lower(r, &rs)?);
        }
        // Inner joins take a Scan(filter) child as (unfiltered table, predicate): rows the filter rejects get no key
        let fuse = j.join_type == JoinType::Inner;
        let mut side = |plan: &LogicalPlan, exprs: &mut ExprArray| -> HipResult<(Child, i32)> {
            match Self::fusable_filter(plan) {
                Some((scan, predicate)) if fuse => {
                    let root = exprs.lower(predicate, &scan.schema())?;
                    let leaf = TableScan { filter: Some(predicate.clone()), ..scan.clone() };
                    Ok((Child::Hip(self.scan(&leaf)?), root))
                }
                _ => Ok((self.child(plan)?, -1)),
            }
        };
        let (left, left_scan_filter) = side(&j.left, &mut left_exprs)?;
        let (right, right_scan_filter) = side(&j.right, &mut right_exprs)?;
        let (filter, filter_root, filter_sides, filter_cols) = Self::join_filter(&j.filter, &ls, &rs)?;
        Ok(Arc::new(HipHashJoin {
            ctx: self.ctx.clone(),
            schema: j.schema.arrow_schema(),
            left,
            right,
            join_type: j.join_type,
            left_exprs,
            right_exprs,
            on_left,
            on_right,
            filter,
            filter_root,
            filter_sides,
            filter_cols,
            left_scan_filter,
            right_scan_filter,
        }))
    }

    // ---- no equi-keys -> NestedLoopJoinExec (planner/mod.rs:316-320)
    fn nested_loop_join(&self, j: &Join) -> HipResult<Arc<dyn HipNode>> {
        let (ls, rs) = (j.left.schema(), j.right.schema());
        let (filter, filter_root, filter_sides, filter_cols) = Self::join_filter(&j.filter, &ls, &rs)?;
        Ok(Arc::new(HipNestedLoopJoin {
            ctx: self.ctx.clone(),
            schema: j.schema.arrow_schema(),
            left: self.child(&j.left)?,
            right: self.child(&j.right)?,
            join_type: j.join_type,
            filter,
            filter_root,
            filter_sides,
            filter_cols,
            cross: false,
        }))
    }

    // ---- physical_plan_cross_join (planner/mod.rs:259-263)
    fn cross_join(&self, c: &CrossJoin) -> HipResult<Arc<dyn HipNode>> {
        Ok(Arc::new(HipNestedLoopJoin {
            ctx: self.ctx.clone(),
            schema: c.schema.arrow_schema(),
            left: self.child(&c.left)?,
            right: self.child(&c.right)?,
            join_type: JoinType::Inner,
            filter: ExprArray::new(),
            filter_root: -1,
            filter_sides: vec![],
            filter_cols: vec![],
            cross: true,
        }))
    }

    // ---- physical_plan_projection (planner/mod.rs:155-172)
    fn projection(&self, p: &Projection) -> HipResult<Arc<dyn HipNode>> {
        let input_schema = p.input.schema();
        let mut exprs = ExprArray::new();
        let roots = p.exprs.iter().map(|e| exprs.lower(e, &input_schema)).collect::<HipResult<Vec<_>>>()?;
        Ok(Arc::new(HipProjection { ctx: self.ctx.clone(), schema: p.schema.arrow_schema(), input: self.child(&p.input)?, exprs, roots }))
    }

    // ---- physical_plan_sort / physical_plan_sort_with_limit (planner/mod.rs:330-348): nulls_first = true for every key
    fn sort(&self, s: &Sort, limit: Option<usize>) -> HipResult<Arc<dyn HipNode>> {
        let input_schema = s.input.schema();
        let mut exprs = ExprArray::new();
        let mut keys = Vec::new();
        let mut descending = Vec::new();
        for SortExpr { expr, asc } in s.exprs.iter() {
            keys.push(exprs.lower(expr, &input_schema)?);
            descending.push(!*asc as i32);
        }
        let nulls_first = vec![1; keys.len()];
        Ok(Arc::new(HipSort {
            ctx: self.ctx.clone(),
            input: self.child(&s.input)?,
            exprs,
            keys,
            descending,
            nulls_first,
            limit: limit.map(|n| n as i64).unwrap_or(-1),
        }))
    }

    // ---- LogicalPlan::Limit with the top-N pushdown into the Sort below it (planner/mod.rs:67-83)
    fn limit(&self, l: &Limit) -> HipResult<Arc<dyn HipNode>> {
        let input = match (l.fetch, l.input.as_ref()) {
            (Some(fetch), LogicalPlan::Sort(sort)) => Child::Hip(self.sort(sort, Some(fetch.saturating_add(l.skip)))?),
            _ => self.child(&l.input)?,
        };
        Ok(Arc::new(HipLimit { ctx: self.ctx.clone(), input, skip: l.skip as i64, fetch: l.fetch.map(|n| n as i64).unwrap_or(-1) }))
    }
}

impl QueryPlanner for HipQueryPlanner {
    fn create_physical_plan(&self, plan: &LogicalPlan) -> Result<Arc<dyn PhysicalPlan>> {
        match self.plan(plan)? {
            Child::Hip(node) => Ok(upcast(node)),
            Child::Cpu(node) => Ok(node),
        }
    }

    /// stand-alone expressions (Values rows, DML filters, ...) stay the reference's own: the HIP nodes evaluate theirs in
    /// generated kernels and never go through `PhysicalExpr::evaluate`
    fn create_physical_expr(&self, input_schema: &SchemaRef, expr: &LogicalExpr) -> Result<Arc<dyn PhysicalExpr>> {
        self.cpu.create_physical_expr(input_schema, expr)
    }
}

/// keep `Error` in scope for the doc links above
#[allow(dead_code)]
fn _error_type(e: Error) -> Error {
    e
}
