"""PhysicalPlan mirrors — the host side of the drop-in boundary.

Same node names, constructor arguments and error behaviour as the reference's operators
(physical/plan/{scan,filter}.rs, aggregate/{hash,no_grouping}.rs, join/hash_join.rs and
datasource/memory.rs), so a parity test reads like the reference's own test. ``execute()`` returns
``list[pyarrow.RecordBatch]`` exactly like ``PhysicalPlan::execute() -> Result<Vec<RecordBatch>>``
(physical/plan/mod.rs:25-29). All compute happens in libqhip's HIP kernels; between two HIP nodes the
batches stay in HBM (``execute_device``) and only the root's result is downloaded.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import pyarrow as pa

from . import _ffi
from ._ffi import DeviceTable, get_context
from .datatypes import JoinSide, JoinType
from .expr import AggregateExpr, ExprArray, PhysicalExpr, int32_array

FIELD_QUALIFIERS_META_KEY = "qurious.field_qualifiers"   # common/table_schema.rs:18


class PhysicalPlan:
    """trait PhysicalPlan (physical/plan/mod.rs:25-29)."""

    def schema(self) -> pa.Schema:
        raise NotImplementedError

    def execute(self) -> List[pa.RecordBatch]:
        t = self.execute_device()
        batches = t.to_batches()
        sch = self.schema()
        # re-attach names/metadata the C side is agnostic of (SURVEY §8b "Data contract")
        return [_with_schema(b, sch) for b in batches]

    def execute_device(self) -> DeviceTable:
        raise NotImplementedError

    def children(self) -> Optional[List["PhysicalPlan"]]:
        return None


def _with_schema(batch: pa.RecordBatch, schema: Optional[pa.Schema]) -> pa.RecordBatch:
    if schema is None or len(schema) != batch.num_columns:
        return batch
    fields = []
    for k, f in enumerate(schema):
        got = batch.schema.field(k)
        if got.type != f.type:
            raise _ffi.InternalError(_ffi.QHIP_INVALID_ARGUMENT,
                                     f"column {k} ('{f.name}') has type {got.type} but the plan schema says {f.type}")
        fields.append(pa.field(f.name, f.type, nullable=True if batch.column(k).null_count else f.nullable, metadata=f.metadata))
    return pa.RecordBatch.from_arrays(batch.columns, schema=pa.schema(fields, metadata=schema.metadata))


class MemoryTable:
    """datasource/memory.rs:20-98 — also the universal fixture source of the reference's tests."""

    def __init__(self, schema: pa.Schema, data: Sequence[pa.RecordBatch], lazy_upload: bool = False):
        self._schema = schema
        self.data = list(data)
        self.lazy_upload = lazy_upload   # True: a column moves to HBM when a query first reads it (file-backed tables)
        self._device: Optional[DeviceTable] = None
        self._device_of: Optional[tuple] = None   # the batch objects the device copy was made from (held: ids cannot be reused)

    @staticmethod
    def try_new(schema: pa.Schema, data: Sequence[pa.RecordBatch]) -> "MemoryTable":
        return MemoryTable(schema, data)

    def schema(self) -> pa.Schema:
        return self._schema

    def device_table(self) -> DeviceTable:
        """Batches pinned in HBM (the reference keeps them in host memory behind an RwLock): uploaded once — and again when
        `data` no longer holds the very batches the copy was made from. MemoryTable::insert appends and ::delete rewrites /
        clears the batch list (memory.rs:104-137), so neither the table's identity nor its batch and row counts tell
        whether the data changed; the batch OBJECTS do (immutable, compared by identity, kept alive by the cache entry).
        The same rule as the Rust shim's HipContext::table_of."""
        now = tuple(self.data)
        same = self._device_of is not None and len(now) == len(self._device_of) and all(a is b for a, b in zip(now, self._device_of))
        if self._device is None or not same:
            self._device = DeviceTable.from_batches(get_context(), self._schema, self.data, lazy=self.lazy_upload)
            self._device_of = now
        return self._device

    def insert(self, batches: Sequence[pa.RecordBatch]) -> int:
        """MemoryTable::insert (memory.rs:104-111): appends; returns the reference's (quirky) 0 affected rows"""
        self.data.extend(batches)
        return 0

    def delete(self, predicate_mask_of=None) -> int:
        """MemoryTable::delete (memory.rs:113-137): no filter clears the table; a filter (here: batch -> boolean keep mask
        of the rows that STAY, the host's business) rewrites every batch"""
        before = sum(b.num_rows for b in self.data)
        if predicate_mask_of is None:
            self.data = []
        else:
            self.data = [b.filter(predicate_mask_of(b)) for b in self.data]
        return before - sum(b.num_rows for b in self.data)

    def scan(self, projection: Optional[List[str]], filters: Optional[PhysicalExpr]) -> List[pa.RecordBatch]:
        """TableProvider::scan (memory.rs:69-98)."""
        return Scan(self._schema, self, projection, filters).execute()


class Scan(PhysicalPlan):
    """physical/plan/scan.rs:12-47"""

    def __init__(self, schema: pa.Schema, datasource: MemoryTable, projections: Optional[List[str]] = None,
                 filter: Optional[PhysicalExpr] = None):
        self._schema, self.datasource, self.projections, self.filter = schema, datasource, projections, filter

    def schema(self) -> pa.Schema:
        return self._schema

    def _projection_indices(self) -> Optional[List[int]]:
        if self.projections is None:
            return None
        src = self.datasource.schema()
        idx = []
        for name in self.projections:
            k = src.get_field_index(name)
            if k < 0:
                raise _ffi.ArrowError(_ffi.QHIP_INVALID_ARGUMENT, f"Schema error: Unable to get field named \"{name}\"")
            idx.append(k)
        return idx

    def execute_device(self) -> DeviceTable:
        base = self.datasource.device_table()
        proj = self._projection_indices()
        if self.filter is None and proj is None:
            return base
        return _filter_device(base, self.filter, proj)

    def execute(self) -> List[pa.RecordBatch]:
        if self.filter is None and self.projections is None:
            return list(self.datasource.data)   # scan.rs:40-42 without filter: the stored batches themselves
        return super().execute()


def _feeding(ctx, node: "PhysicalPlan") -> DeviceTable:
    """Execute `node` as the input of an operator that reads a device-side row count (HashAggregate's input, a hash
    join's build side, an exchange): a hash join at the top of `node` may then skip the wait for its output size
    (qhip.h: qhip_ctx_allow_deferred_sizes). The consumer runs right after, inside `_retrying`.
    A subtree with exchange operators may defer too (round 4): a QHIP_RETRY on ONE rank re-executes `node`, but an exchange
    operator inside it hands back what it received in this execution (`exchange_cache`) instead of repeating its
    collectives, which the other ranks would not take part in — only local operators run again. (QHIP_EXCHANGE_NO_DEFER=1:
    the round-3 rule — such a subtree never defers.)"""
    if _subtree_exchanges(node) and os.environ.get("QHIP_EXCHANGE_NO_DEFER") == "1":
        with ctx.no_deferred_sizes():
            return node.execute_device()
    ctx.allow_deferred_sizes(+1)
    try:
        return node.execute_device()
    finally:
        ctx.allow_deferred_sizes(-1)


def _subtree_exchanges(node) -> bool:
    """True when `node` or a descendant is a multi-rank operator (exchange.py marks its classes `_exchanges = True`).
    Walks the operator attributes rather than `children()`: Sort / Limit / NoGroupingAggregate report their
    input's children (or none) there, like the reference's (sort.rs:83-85, limit.rs:59-61, no_grouping.rs:63-65)."""
    if getattr(node, "_exchanges", False):
        return True
    return any(_subtree_exchanges(child) for attr in ("input", "left", "right", "partial")
               for child in [getattr(node, attr, None)] if isinstance(child, PhysicalPlan))


def _retrying(ctx, run):
    """run() = execute the input(s), then the operator. QHIP_RETRY from the operator: a join of deferred size below had
    too little room; it has forgotten its hint, so the second run waits for the size.
    The outermost call brackets one EXECUTION of a plan: what the exchange operators below received during it stays in
    `exchange_cache(ctx)` until it ends, so that a re-run of their subtree repeats no collective."""
    depth = getattr(ctx, "_retry_depth", 0)
    ctx._retry_depth = depth + 1
    try:
        for attempt in range(3):
            try:
                return run()
            except _ffi.RetryInput:
                if attempt == 2:
                    raise
            except Exception:
                ctx.allow_deferred_sizes(0)   # reset: nothing of deferred size is left in flight after an error
                raise
    finally:
        ctx._retry_depth = depth
        if depth == 0:
            exchange_cache(ctx).clear()


def exchange_cache(ctx) -> dict:
    """{id(exchange operator): what it received} for the plan execution in progress (cleared when the outermost
    `_retrying` returns): the tables an exchange delivered are a function of the query's inputs, so an operator that is
    executed again inside the same execution — a local retry after QHIP_RETRY — reuses them."""
    cache = getattr(ctx, "_exchange_cache", None)
    if cache is None:
        cache = ctx._exchange_cache = {}
    return cache


# Instrumented executions (bench.py's per-kernel roofline records): while a list is installed here every device operator
# appends (operator label, qhip_exec_stats of the call). Reading the stats waits for the operator's events, so this is for
# separate, untimed passes only.
STATS_SINK: Optional[list] = None


def _record_stats(ctx, label: str):
    if STATS_SINK is not None:
        STATS_SINK.append((label, ctx.last_stats()))


def _filter_device(table: DeviceTable, predicate: Optional[PhysicalExpr], projection: Optional[List[int]]) -> DeviceTable:
    ctx = table.ctx
    ea = ExprArray()
    root = -1
    if predicate is not None and projection is not None:
        # memory.rs:79-93: the projection is applied first (column buffers are shared, nothing is copied), the filter
        # then sees — and indexes — the projected batch
        return _filter_device(_filter_device(table, None, projection), predicate, None)
    if predicate is not None:
        root = ea.lower(predicate)
    arr, n = ea.c_array()
    out = C.c_void_p()
    pj = int32_array(projection or [])
    ctx.check(ctx.lib.qhip_filter_execute(ctx.handle, table.handle, arr, n, root, pj if projection is not None else None,
                                          len(projection) if projection is not None else -1, C.byref(out)))
    return DeviceTable(ctx, out)


class Filter(PhysicalPlan):
    """physical/plan/filter.rs:12-48"""

    def __init__(self, input: PhysicalPlan, predicate: PhysicalExpr):
        self.input, self.predicate = input, predicate

    def schema(self) -> pa.Schema:
        return self.input.schema()

    def execute_device(self) -> DeviceTable:
        return _filter_device(self.input.execute_device(), self.predicate, None)

    def children(self):
        return [self.input]


def _lower_aggregate(predicate: Optional[PhysicalExpr], group_exprs: Sequence[PhysicalExpr],
                     aggregate_exprs: Sequence[AggregateExpr], names: Sequence[str]):
    """The POD description of an aggregation (expression array, roots, qhip_agg array, output names) for the C ABI."""
    ea = ExprArray()
    pred = ea.lower(predicate) if predicate is not None else -1
    groups = [ea.lower(g) for g in group_exprs]
    aggs = (_ffi.qhip_agg * max(1, len(aggregate_exprs)))()
    from .datatypes import to_qhip_dtype
    for k, a in enumerate(aggregate_exprs):
        aggs[k].kind = a.kind
        aggs[k].expr = ea.lower(a.expression())
        aggs[k].return_type = to_qhip_dtype(a._return_type())
    arr, n = ea.c_array()
    cnames = (C.c_char_p * max(1, len(names)))(*[s.encode() for s in names])
    return (arr, n, pred, int32_array(groups), len(groups), aggs, len(aggregate_exprs), cnames, ea)


def _aggregate_device(table: DeviceTable, predicate: Optional[PhysicalExpr], group_exprs: Sequence[PhysicalExpr],
                      aggregate_exprs: Sequence[AggregateExpr], names: Sequence[str], lowered=None) -> DeviceTable:
    ctx = table.ctx
    arr, n, pred, groups, n_groups, aggs, n_aggs, cnames, _keep = lowered or _lower_aggregate(predicate, group_exprs, aggregate_exprs, names)
    out = C.c_void_p()
    ctx.check(ctx.lib.qhip_hash_aggregate_execute(ctx.handle, table.handle, arr, n, pred, groups, n_groups, aggs, n_aggs, cnames, C.byref(out)))
    _record_stats(ctx, "aggregate")
    return DeviceTable(ctx, out)


class HashAggregate(PhysicalPlan):
    """physical/plan/aggregate/hash.rs:110-176. When the input is a Scan with a pushed-down filter
    (the shape PushdownFilter produces for Q1) the predicate is fused into the aggregation
    kernel: one pass over the referenced columns, nothing materialised in between."""

    def __init__(self, schema: pa.Schema, input: PhysicalPlan, group_exprs: Sequence[PhysicalExpr],
                 aggregate_exprs: Sequence[AggregateExpr]):
        self._schema, self.input = schema, input
        self.group_exprs, self.aggregate_exprs = list(group_exprs), list(aggregate_exprs)

    def schema(self) -> pa.Schema:
        return self._schema

    def _source(self) -> Tuple[DeviceTable, Optional[PhysicalExpr]]:
        node = self.input
        if isinstance(node, Scan) and node.projections is None:
            return node.datasource.device_table(), node.filter
        if isinstance(node, Filter) and isinstance(node.input, Scan) and node.input.filter is None and node.input.projections is None:
            return node.input.datasource.device_table(), node.predicate
        return _feeding(get_context(), node), None

    def execute_device(self) -> DeviceTable:
        return _retrying(get_context(), self._execute_once)

    def _execute_once(self) -> DeviceTable:
        table, pred = self._source()
        # plan nodes are immutable after construction (like the reference's): the lowered description is built once
        cached = getattr(self, "_lowered", None)
        if cached is None or cached[0] is not pred:
            names = [f.name for f in self._schema] if self._schema is not None else \
                [f"c{k}" for k in range(len(self.group_exprs) + len(self.aggregate_exprs))]
            cached = (pred, _lower_aggregate(pred, self.group_exprs, self.aggregate_exprs, names))
            self._lowered = cached
        return _aggregate_device(table, pred, self.group_exprs, self.aggregate_exprs, None, cached[1])

    def children(self):
        return [self.input]

    def __str__(self):
        return f"HashAggregateExec: groupExpr={[str(g) for g in self.group_exprs]}, aggrExpr={[str(a) for a in self.aggregate_exprs]}"


class NoGroupingAggregate(HashAggregate):
    """physical/plan/aggregate/no_grouping.rs:9-66"""

    def __init__(self, schema: pa.Schema, input: PhysicalPlan, aggr_expr: Sequence[AggregateExpr]):
        super().__init__(schema, input, [], aggr_expr)

    def children(self):
        return None   # no_grouping.rs:63-65


class Projection(PhysicalPlan):
    """physical/plan/projection.rs:10-51: one output column per expression, per input batch"""

    def __init__(self, schema: pa.Schema, input: PhysicalPlan, exprs: Sequence[PhysicalExpr]):
        self._schema, self.input, self.exprs = schema, input, list(exprs)

    def schema(self) -> pa.Schema:
        return self._schema

    def execute_device(self) -> DeviceTable:
        table = self.input.execute_device()
        ctx = table.ctx
        cached = getattr(self, "_lowered", None)
        if cached is None:
            ea = ExprArray()
            roots = [ea.lower(e) for e in self.exprs]
            arr, n = ea.c_array()
            names = [f.name for f in self._schema] if self._schema is not None else [f"c{k}" for k in range(len(roots))]
            cached = (arr, n, int32_array(roots), len(roots), (C.c_char_p * max(1, len(names)))(*[s.encode() for s in names]), ea)
            self._lowered = cached
        arr, n, roots, n_out, cnames, _keep = cached
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_projection_execute(ctx.handle, table.handle, arr, n, roots, n_out, cnames, C.byref(out)))
        return DeviceTable(ctx, out)

    def children(self):
        return [self.input]


class SortOptions:
    """arrow::compute::SortOptions { descending, nulls_first } (arrow-rs default: ascending, nulls first)"""

    def __init__(self, descending: bool = False, nulls_first: bool = True):
        self.descending, self.nulls_first = bool(descending), bool(nulls_first)


class PhysicalSortExpr:
    """physical/plan/sort.rs:12-21 (`PhyscialSortExpr` in the reference)"""

    def __init__(self, expr: PhysicalExpr, options: SortOptions):
        self.expr, self.options = expr, options


class Sort(PhysicalPlan):
    """physical/plan/sort.rs:23-86: lexsort of the concatenated input by the key expressions, ties in input order, one
    output batch; `limit` keeps the first rows only (Sort::new_with_limit, the planner's top-N pushdown)."""

    def __init__(self, exprs: Sequence[PhysicalSortExpr], input: PhysicalPlan, limit: Optional[int] = None):
        self.exprs, self.input, self.limit = list(exprs), input, limit

    @staticmethod
    def new_with_limit(exprs: Sequence[PhysicalSortExpr], input: PhysicalPlan, limit: Optional[int]) -> "Sort":
        return Sort(exprs, input, limit)

    def schema(self) -> pa.Schema:
        return self.input.schema()

    def execute_device(self) -> DeviceTable:
        table = self.input.execute_device()
        ctx = table.ctx
        cached = getattr(self, "_lowered", None)
        if cached is None:
            ea = ExprArray()
            roots = [ea.lower(e.expr) for e in self.exprs]
            arr, n = ea.c_array()
            cached = (arr, n, int32_array(roots), int32_array([int(e.options.descending) for e in self.exprs]),
                      int32_array([int(e.options.nulls_first) for e in self.exprs]), len(roots), ea)
            self._lowered = cached
        arr, n, roots, desc, nf, n_keys, _keep = cached
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_sort_execute(ctx.handle, table.handle, arr, n, roots, desc, nf, n_keys,
                                            -1 if self.limit is None else int(self.limit), C.byref(out)))
        return DeviceTable(ctx, out)

    def children(self):
        return self.input.children()   # sort.rs:83-85


class Limit(PhysicalPlan):
    """physical/plan/limit.rs:10-62: rows [skip, skip + fetch) of the input's batch list"""

    def __init__(self, input: PhysicalPlan, fetch: Optional[int], skip: int):
        self.input, self.fetch, self.skip = input, fetch, int(skip)

    def schema(self) -> pa.Schema:
        return self.input.schema()

    def execute_device(self) -> DeviceTable:
        table = self.input.execute_device()
        ctx = table.ctx
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_limit_execute(ctx.handle, table.handle, self.skip, -1 if self.fetch is None else int(self.fetch), C.byref(out)))
        return DeviceTable(ctx, out)

    def children(self):
        return self.input.children()   # limit.rs:59-61


ColumnIndex = Tuple[int, JoinSide]


class JoinFilter:
    """physical/plan/join/mod.rs JoinFilter { expr, schema, column_indices }"""

    def __init__(self, expr: PhysicalExpr, column_indices: Sequence[ColumnIndex], schema: pa.Schema):
        self.expr, self.column_indices, self.schema = expr, list(column_indices), schema


def build_join_schema(left: pa.Schema, right: pa.Schema, join_type: JoinType) -> Tuple[pa.Schema, List[ColumnIndex]]:
    """physical/plan/join/mod.rs:26-123 (field nullability by join type, merged qualifier metadata)."""
    sep = "\x1f"
    key = FIELD_QUALIFIERS_META_KEY.encode()
    lmeta = dict(left.metadata or {})
    if join_type in (JoinType.LeftSemi, JoinType.LeftAnti):
        fields = [f for f in left]
        return pa.schema(fields, metadata=lmeta or None), [(k, JoinSide.Left) for k in range(len(left))]
    ln, rn = {JoinType.Left: (False, True), JoinType.Right: (True, False), JoinType.Inner: (False, False),
              JoinType.Full: (True, True)}[join_type]
    fields, idx = [], []
    for k, f in enumerate(left):
        fields.append(f.with_nullable(True) if ln else f)
        idx.append((k, JoinSide.Left))
    for k, f in enumerate(right):
        fields.append(f.with_nullable(True) if rn else f)
        idx.append((k, JoinSide.Right))

    def parts(s: pa.Schema):
        q = (s.metadata or {}).get(key)
        q = q.decode() if q is not None else sep * max(0, len(s) - 1)
        p = q.split(sep)
        return p if len(p) == len(s) else [""] * len(s)

    meta = dict(lmeta)
    meta[key] = sep.join(parts(left) + parts(right)).encode()
    return pa.schema(fields, metadata=meta), idx


class NestedLoopJoinExec(PhysicalPlan):
    """physical/plan/join/nest_loop_join.rs:42-228 — the reference's join for ON clauses without equi-keys"""

    def __init__(self, left, right, join_type, filter, schema, column_indices):
        self.left, self.right, self.join_type, self.filter = left, right, JoinType(join_type), filter
        self._schema, self.column_indices = schema, column_indices

    @staticmethod
    def try_new(left: PhysicalPlan, right: PhysicalPlan, join_type: JoinType, filter: Optional["JoinFilter"] = None) -> "NestedLoopJoinExec":
        schema, ci = build_join_schema(left.schema(), right.schema(), JoinType(join_type))
        return NestedLoopJoinExec(left, right, join_type, filter, schema, ci)

    def schema(self) -> pa.Schema:
        return self._schema

    def children(self):
        return [self.left, self.right]

    def execute_device(self) -> DeviceTable:
        lt, rt = self.left.execute_device(), self.right.execute_device()
        ctx = lt.ctx
        cached = getattr(self, "_lowered", None)
        if cached is None:
            fe = ExprArray()
            froot, fsides, fcols = -1, [], []
            if self.filter is not None:
                froot = fe.lower(self.filter.expr)
                fsides = [int(s) for _, s in self.filter.column_indices]
                fcols = [int(c) for c, _ in self.filter.column_indices]
            fa, fn = fe.c_array()
            cached = (fa, fn, froot, int32_array(fsides), int32_array(fcols), len(fcols), fe)
            self._lowered = cached
        fa, fn, froot, fsides, fcols, n_fcols, _keep = cached
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_nested_loop_join_execute(ctx.handle, lt.handle, rt.handle, int(self.join_type), fa, fn, froot, fsides, fcols,
                                                        n_fcols, C.byref(out)))
        return DeviceTable(ctx, out)


class CrossJoin(PhysicalPlan):
    """physical/plan/join/cross_join.rs:56-170 — cartesian product, one output batch per (left row, right batch)"""

    def __init__(self, left: PhysicalPlan, right: PhysicalPlan):
        self.left, self.right = left, right
        ls, rs = left.schema(), right.schema()
        sep, key = "\x1f", FIELD_QUALIFIERS_META_KEY.encode()
        meta = dict(ls.metadata or {})

        def parts(schema):
            raw = (schema.metadata or {}).get(key)
            p = raw.decode().split(sep) if raw is not None else [""] * len(schema)
            return p if len(p) == len(schema) else [""] * len(schema)
        meta[key] = sep.join(parts(ls) + parts(rs)).encode()   # cross_join.rs:76-112: qualifiers of both sides, concatenated
        self._schema = pa.schema(list(ls) + list(rs), metadata=meta)

    def schema(self) -> pa.Schema:
        return self._schema

    def children(self):
        return [self.left, self.right]

    def execute_device(self) -> DeviceTable:
        lt, rt = self.left.execute_device(), self.right.execute_device()
        ctx = lt.ctx
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_cross_join_execute(ctx.handle, lt.handle, rt.handle, C.byref(out)))
        return DeviceTable(ctx, out)


class HashJoinExec(PhysicalPlan):
    """physical/plan/join/hash_join.rs:110-384 — build = left, probe = right."""

    def __init__(self, left, right, join_type, on, filter, schema, column_indices):
        self.left, self.right, self.join_type, self.on, self.filter = left, right, JoinType(join_type), list(on), filter
        self._schema, self.column_indices = schema, column_indices

    @staticmethod
    def try_new(left: PhysicalPlan, right: PhysicalPlan, join_type: JoinType,
                on: Sequence[Tuple[PhysicalExpr, PhysicalExpr]], filter: Optional[JoinFilter] = None) -> "HashJoinExec":
        if len(on) == 0:
            raise _ffi.InternalError(_ffi.QHIP_INVALID_ARGUMENT, "Internal error: On constraints in HashJoinExec should be non-empty")
        schema, ci = build_join_schema(left.schema(), right.schema(), JoinType(join_type))
        return HashJoinExec(left, right, join_type, on, filter, schema, ci)

    def schema(self) -> pa.Schema:
        return self._schema

    def children(self):
        return [self.left, self.right]

    @staticmethod
    def _side(node: PhysicalPlan, fuse: bool):
        """(device table, scan filter or None): an Inner join takes a Scan(filter) child as (unfiltered table, predicate)
        and fuses the predicate into its key evaluation instead of materialising the filtered batches."""
        if fuse and isinstance(node, Scan) and node.filter is not None and node.projections is None:
            return node.datasource.device_table(), node.filter
        return node.execute_device(), None

    def execute_device(self) -> DeviceTable:
        return _retrying(get_context(), self._execute_once)

    def _execute_once(self) -> DeviceTable:
        fuse = self.join_type == JoinType.Inner
        # the build side may arrive with a device-side row count when nothing executes between it and this join, i.e.
        # when the probe side is a table access (a Scan, fused or plain) — the reference's order, left first, is kept
        right_is_table = isinstance(self.right, Scan) and self.right.projections is None and (self.right.filter is None or fuse)
        if right_is_table and not isinstance(self.left, Scan):
            lt, lpred = _feeding(get_context(), self.left), None
        else:
            lt, lpred = self._side(self.left, fuse)
        rt, rpred = self._side(self.right, fuse)
        return self._join_tables(lt, rt, lpred, rpred)

    def _lower(self, lpred, rpred):
        le, re_, fe = ExprArray(), ExprArray(), ExprArray()
        on_l = [le.lower(l) for l, _ in self.on]
        on_r = [re_.lower(r) for _, r in self.on]
        lp = le.lower(lpred) if lpred is not None else -1
        rp = re_.lower(rpred) if rpred is not None else -1
        froot, fsides, fcols = -1, [], []
        if self.filter is not None:
            froot = fe.lower(self.filter.expr)
            fsides = [int(s) for _, s in self.filter.column_indices]
            fcols = [int(c) for c, _ in self.filter.column_indices]
        la, ln = le.c_array()
        ra, rn = re_.c_array()
        fa, fn = fe.c_array()
        return (la, ln, ra, rn, int32_array(on_l), int32_array(on_r), len(self.on), fa, fn, froot, int32_array(fsides), int32_array(fcols),
                len(fcols), lp, rp, (le, re_, fe))

    def _join_tables(self, lt: DeviceTable, rt: DeviceTable, lpred=None, rpred=None) -> DeviceTable:
        ctx = lt.ctx
        # plan nodes are immutable after construction (like the reference's): the lowered description is built once
        cached = getattr(self, "_lowered", None)
        if cached is None or cached[0] is not lpred or cached[1] is not rpred:
            cached = (lpred, rpred, self._lower(lpred, rpred))
            self._lowered = cached
        la, ln, ra, rn, on_l, on_r, n_on, fa, fn, froot, fsides, fcols, n_fcols, lp, rp, _keep = cached[2]
        out = C.c_void_p()
        ctx.check(ctx.lib.qhip_hash_join_execute(ctx.handle, lt.handle, rt.handle, int(self.join_type), la, ln, ra, rn, on_l, on_r, n_on,
                                                 fa, fn, froot, fsides, fcols, n_fcols, lp, rp, C.byref(out)))
        _record_stats(ctx, "hash_join")
        return DeviceTable(ctx, out)
