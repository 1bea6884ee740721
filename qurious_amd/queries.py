"""Physical plans of the benchmark configurations (BASELINE.json configs), built exactly as the reference's
planner would build them after PushdownFilter + TypeCoercion."""
from __future__ import annotations

import pyarrow as pa

from .datatypes import Operator, ScalarValue
from .expr import (AvgAggregateExpr, BinaryExpr, CastExpr, Column, CountAggregateExpr, Literal, SumAggregateExpr, avg_return_type)
from .plan import HashAggregate, MemoryTable, Scan
from .synth import DEC, LINEITEM_SCHEMA


def _date(s: str):
    # Date32 column vs Utf8 literal is coerced to `col < CAST(Utf8 AS Date32)` (utils/type_coercion.rs:50-55)
    return CastExpr(Literal(ScalarValue.Utf8(s)), pa.date32())


def q1_mini(table: MemoryTable) -> HashAggregate:
    """configs[0]/[1]: SELECT l_returnflag, SUM(l_quantity) FROM lineitem WHERE l_shipdate < '1998-09-01' GROUP BY l_returnflag"""
    pred = BinaryExpr(Column("l_shipdate", 0), Operator.Lt, _date("1998-09-01"))
    scan = Scan(LINEITEM_SCHEMA, table, None, pred)
    schema = pa.schema([pa.field("l_returnflag", pa.string()), pa.field("SUM(l_quantity)", DEC)])
    return HashAggregate(schema, scan, [Column("l_returnflag", 1)], [SumAggregateExpr(Column("l_quantity", 3), DEC)])


def q1_full(table: MemoryTable) -> HashAggregate:
    """configs[2]: TPC-H Q1 (tests/tpch/q1.slt:2-22) up to the HashAggregate output."""
    pred = BinaryExpr(Column("l_shipdate", 0), Operator.LtEq, _date("1998-09-02"))
    scan = Scan(LINEITEM_SCHEMA, table, None, pred)
    one = CastExpr(Literal(ScalarValue.Int64(1)), pa.decimal128(20, 0))   # utils/type_coercion.rs:145-164
    qty, price, disc, tax = Column("l_quantity", 3), Column("l_extendedprice", 4), Column("l_discount", 5), Column("l_tax", 6)
    disc_price = BinaryExpr(price, Operator.Mul, BinaryExpr(one, Operator.Sub, disc))          # Decimal128(38,4)
    charge = BinaryExpr(disc_price, Operator.Mul, BinaryExpr(one, Operator.Add, tax))          # Decimal128(38,6)
    t4, t6 = pa.decimal128(38, 4), pa.decimal128(38, 6)
    aggs = [
        SumAggregateExpr(qty, DEC), SumAggregateExpr(price, DEC), SumAggregateExpr(disc_price, t4), SumAggregateExpr(charge, t6),
        AvgAggregateExpr(qty, DEC, avg_return_type(DEC)), AvgAggregateExpr(price, DEC, avg_return_type(DEC)),
        AvgAggregateExpr(disc, DEC, avg_return_type(DEC)), CountAggregateExpr(Literal(ScalarValue.Int64(1))),
    ]
    names = ["l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "avg_qty", "avg_price",
             "avg_disc", "count_order"]
    types = [pa.string(), pa.string(), DEC, DEC, t4, t6] + [avg_return_type(DEC)] * 3 + [pa.int64()]
    schema = pa.schema([pa.field(n, t) for n, t in zip(names, types)])
    return HashAggregate(schema, scan, [Column("l_returnflag", 1), Column("l_linestatus", 2)], aggs)


def q1_partial(table: MemoryTable) -> HashAggregate:
    """Q1 over ONE RANK's slice of lineitem, as mergeable partials: the four SUMs of q1_full plus SUM(l_discount) and COUNT —
    AVG(x) of the whole table is SUM of the ranks' sums / SUM of their counts (avg.rs:91-116 applied after the merge)."""
    full = q1_full(table)
    aggs = full.aggregate_exprs[:4] + [SumAggregateExpr(Column("l_discount", 5), DEC), full.aggregate_exprs[7]]
    names = ["l_returnflag", "l_linestatus", "sum_qty", "sum_base_price", "sum_disc_price", "sum_charge", "sum_disc", "count_order"]
    types = [pa.string(), pa.string(), DEC, DEC, pa.decimal128(38, 4), pa.decimal128(38, 6), DEC, pa.int64()]
    return HashAggregate(pa.schema([pa.field(n, t) for n, t in zip(names, types)]), full.input, full.group_exprs, aggs)


def q3(customer, orders, lineitem, join_cls=None, agg_cls=None, join2_cls=None):
    """configs[3]: TPC-H Q3 (tests/tpch/q3.slt:2-24) up to the HashAggregate output, in the plan shape the reference's
    optimizer produces: filters pushed into the scans, build side = left child, no side swapping.
    customer / orders / lineitem are MemoryTables over synth.{CUSTOMER,ORDERS,LINEITEM_Q3}_SCHEMA."""
    from .datatypes import JoinType
    from .plan import HashJoinExec
    from .synth import CUSTOMER_SCHEMA, LINEITEM_Q3_SCHEMA, ORDERS_SCHEMA
    join_cls = join_cls or HashJoinExec
    join2_cls = join2_cls or join_cls      # (multi-GPU: the two joins may take different exchange strategies, exchange.py)
    day = _date("1995-03-15")
    c_scan = Scan(CUSTOMER_SCHEMA, customer, None, BinaryExpr(Column("c_mktsegment", 1), Operator.Eq, Literal(ScalarValue.Utf8("BUILDING"))))
    o_scan = Scan(ORDERS_SCHEMA, orders, None, BinaryExpr(Column("o_orderdate", 2), Operator.Lt, day))
    l_scan = Scan(LINEITEM_Q3_SCHEMA, lineitem, None, BinaryExpr(Column("l_shipdate", 1), Operator.Gt, day))
    j1 = join_cls.try_new(c_scan, o_scan, JoinType.Inner, [(Column("c_custkey", 0), Column("o_custkey", 1))], None)
    # j1 schema: c_custkey c_mktsegment | o_orderkey o_custkey o_orderdate o_shippriority
    j2 = join2_cls.try_new(j1, l_scan, JoinType.Inner, [(Column("o_orderkey", 2), Column("l_orderkey", 0))], None)
    # j2 schema: j1 (6) | l_orderkey l_shipdate l_extendedprice l_discount
    one = CastExpr(Literal(ScalarValue.Int64(1)), pa.decimal128(20, 0))
    revenue = BinaryExpr(Column("l_extendedprice", 8), Operator.Mul, BinaryExpr(one, Operator.Sub, Column("l_discount", 9)))
    t4 = pa.decimal128(38, 4)
    schema = pa.schema([pa.field("l_orderkey", pa.int64()), pa.field("o_orderdate", pa.date32()), pa.field("o_shippriority", pa.int64()),
                        pa.field("revenue", t4)])
    return (agg_cls or HashAggregate)(schema, j2, [Column("l_orderkey", 6), Column("o_orderdate", 4), Column("o_shippriority", 5)],
                                      [SumAggregateExpr(revenue, t4)])


def q3_top10(customer, orders, lineitem, join_cls=None):
    """Q3 through its ORDER BY revenue DESC, o_orderdate LIMIT 10 (q3.slt:20-24), lowered like the reference's planner does
    (planner/mod.rs:67-83): Limit(fetch 10) over Sort(top-N = 10) over the aggregate. (The reference also reorders the
    select list with a Projection in between — SURVEY §8f rank 2; the columns here stay in aggregate order:
    l_orderkey, o_orderdate, o_shippriority, revenue.)"""
    from .planner import DefaultQueryPlanner
    agg = q3(customer, orders, lineitem, join_cls)
    return DefaultQueryPlanner().physical_plan_limit(agg, 10, 0, sort_exprs=[(Column("revenue", 3), False), (Column("o_orderdate", 1), True)])


def q1_full_ordered(table: MemoryTable):
    """Q1 through its ORDER BY l_returnflag, l_linestatus (q1.slt:19-21)"""
    from .planner import DefaultQueryPlanner
    return DefaultQueryPlanner().physical_plan_sort(q1_full(table), [(Column("l_returnflag", 0), True), (Column("l_linestatus", 1), True)])

