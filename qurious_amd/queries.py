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
