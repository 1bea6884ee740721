"""Physical-planner choices of the reference for the hot path (planner/mod.rs:174-321), restated at the physical level:
which node a filter / aggregate / table scan / join becomes. The logical plan, the SQL front-end and the optimizer are out
of scope; a Rust `HipQueryPlanner` makes exactly these choices."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import pyarrow as pa

from .datatypes import JoinType
from .expr import AggregateExpr, PhysicalExpr
from .plan import (CrossJoin, Filter, HashAggregate, HashJoinExec, JoinFilter, Limit, MemoryTable, NestedLoopJoinExec, NoGroupingAggregate, PhysicalPlan,
                   PhysicalSortExpr, Scan, Sort, SortOptions)


class DefaultQueryPlanner:
    def physical_plan_filter(self, input: PhysicalPlan, predicate: PhysicalExpr) -> PhysicalPlan:
        """planner/mod.rs:174-179"""
        return Filter(input, predicate)

    def physical_plan_aggregate(self, schema: pa.Schema, input: PhysicalPlan, group_expr: Sequence[PhysicalExpr],
                                aggr_expr: Sequence[AggregateExpr]) -> PhysicalPlan:
        """planner/mod.rs:181-242: NoGroupingAggregate iff there is no GROUP BY expression, else HashAggregate"""
        if len(group_expr) == 0:
            return NoGroupingAggregate(schema, input, aggr_expr)
        return HashAggregate(schema, input, group_expr, aggr_expr)

    def physical_plan_table_scan(self, schema: pa.Schema, source: MemoryTable, filter: Optional[PhysicalExpr]) -> PhysicalPlan:
        """planner/mod.rs:244-257: projections are always None (no projection pushdown), the pushed-down filter rides along"""
        return Scan(schema, source, None, filter)

    def physical_plan_join(self, left: PhysicalPlan, right: PhysicalPlan, join_type: JoinType,
                           on: Sequence[Tuple[PhysicalExpr, PhysicalExpr]], filter: Optional[JoinFilter]) -> PhysicalPlan:
        """planner/mod.rs:265-321: HashJoinExec iff there are equi-join keys, NestedLoopJoinExec otherwise"""
        if len(on) == 0:
            return NestedLoopJoinExec.try_new(left, right, join_type, filter)
        return HashJoinExec.try_new(left, right, join_type, on, filter)

    def physical_plan_cross_join(self, left: PhysicalPlan, right: PhysicalPlan) -> PhysicalPlan:
        """planner/mod.rs:259-263"""
        return CrossJoin(left, right)

    def physical_plan_sort(self, input: PhysicalPlan, exprs: Sequence[Tuple[PhysicalExpr, bool]], limit: Optional[int] = None) -> PhysicalPlan:
        """planner/mod.rs:330-348: every ORDER BY expression (expr, asc) becomes SortOptions{descending: !asc,
        nulls_first: true}"""
        return Sort.new_with_limit([PhysicalSortExpr(e, SortOptions(descending=not asc, nulls_first=True)) for e, asc in exprs], input, limit)

    def physical_plan_limit(self, input: PhysicalPlan, fetch: Optional[int], skip: int,
                            sort_exprs: Optional[Sequence[Tuple[PhysicalExpr, bool]]] = None) -> PhysicalPlan:
        """planner/mod.rs:67-83: LIMIT directly over ORDER BY (pass the ORDER BY's expressions as `sort_exprs` and its
        input as `input`) sorts only the first skip + fetch rows (top-N), then applies the window; any other LIMIT is a
        plain Limit node."""
        if sort_exprs is not None and fetch is not None:
            return Limit(self.physical_plan_sort(input, sort_exprs, fetch + skip), fetch, skip)
        if sort_exprs is not None:
            return Limit(self.physical_plan_sort(input, sort_exprs, None), fetch, skip)
        return Limit(input, fetch, skip)
