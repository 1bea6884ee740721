"""Kernel catalog: the plan shapes of the benchmark configurations, compiled for gfx950 ahead of time.

``compile_catalog`` needs no GPU (hiprtc cross-compiles); it fills qurious_amd/_kcache so that the first
execution on a GPU box loads code objects instead of compiling. Plans outside the catalog are compiled by
libqhip on first use and cached the same way.
"""
from __future__ import annotations

import time

from . import planning, queries
from .plan import MemoryTable
from .synth import LINEITEM_SCHEMA


def catalog_sources():
    """(name, policy source) of every generated kernel the benchmark queries launch — one of each kernel body of
    csrc/device/qhip_device.hpp gets instantiated, so compiling the catalog is the compile check of the device templates."""
    import pyarrow as pa

    from . import expr as E
    from .datatypes import Operator, ScalarValue
    from .synth import CUSTOMER_SCHEMA, LINEITEM_Q3_SCHEMA, ORDERS_SCHEMA
    table = MemoryTable.try_new(LINEITEM_SCHEMA, [])
    out = []
    for name, plan in (("q1_mini", queries.q1_mini(table)), ("q1_full", queries.q1_full(table))):
        scan = plan.input
        out.append((name + " filter+aggregate", planning.aggregate_source(LINEITEM_SCHEMA, scan.filter, plan.group_exprs, plan.aggregate_exprs)))
    # ... and their narrow variants: what an execution over the synthetic lineitem compiles once it has the columns' |value|
    # statistics (quantity 13 bits, price 24, discount / tax 4): 32 / 64-bit multiplies, 64-bit lane accumulators
    import os
    os.environ["QHIP_PLAN_VALUE_BITS"] = "3:13,4:24,5:4,6:4"
    os.environ["QHIP_PLAN_UTF8_FIXED1"] = "1,2"   # l_returnflag / l_linestatus: one byte each, addressed by row number
    try:
        for name, plan in (("q1_mini", queries.q1_mini(table)), ("q1_full", queries.q1_full(table)), ("q1_partial", queries.q1_partial(table))):
            scan = plan.input
            out.append((name + " filter+aggregate, bounded values", planning.aggregate_source(LINEITEM_SCHEMA, scan.filter, plan.group_exprs, plan.aggregate_exprs)))
        # ... over the 16-byte Arrow layout: what the FIRST execution over a table runs (its decimal columns get their narrow
        # copies at the second big read, common.hpp DevColumn::big_reads)
        os.environ["QHIP_NARROW_DECIMALS"] = "0"
        for name, plan in (("q1_mini", queries.q1_mini(table)), ("q1_full", queries.q1_full(table)), ("q1_partial", queries.q1_partial(table))):
            scan = plan.input
            out.append((name + " filter+aggregate, bounded values, Arrow layout", planning.aggregate_source(LINEITEM_SCHEMA, scan.filter, plan.group_exprs, plan.aggregate_exprs)))
    finally:
        os.environ.pop("QHIP_PLAN_VALUE_BITS", None)
        os.environ.pop("QHIP_PLAN_UTF8_FIXED1", None)
        os.environ.pop("QHIP_NARROW_DECIMALS", None)
    # Q3: build-side key words (+ fused scan filter), fused probe kernels, the aggregate over the second join's output
    tabs = (MemoryTable.try_new(CUSTOMER_SCHEMA, []), MemoryTable.try_new(ORDERS_SCHEMA, []), MemoryTable.try_new(LINEITEM_Q3_SCHEMA, []))
    agg = queries.q3(*tabs)
    j2 = agg.input
    j1 = j2.left
    out.append(("q3 customer keys+filter", planning.keys_source(CUSTOMER_SCHEMA, [j1.on[0][0]])))
    out.append(("q3 customer build entries", planning.scatter_source(CUSTOMER_SCHEMA, [j1.on[0][0]], j1.left.filter)))
    out.append(("q3 join-1 output build entries", planning.scatter_source(j1.schema(), [j2.on[0][0]])))
    out.append(("q3 orders probe", planning.probe_source(ORDERS_SCHEMA, [j1.on[0][1]], j1.right.filter)))
    out.append(("q3 join-1 output keys", planning.keys_source(j1.schema(), [j2.on[0][0]])))
    out.append(("q3 lineitem probe", planning.probe_source(LINEITEM_Q3_SCHEMA, [j2.on[0][1]], j2.right.filter)))
    out.append(("q3 aggregate", planning.aggregate_source(j2.schema(), None, agg.group_exprs, agg.aggregate_exprs)))
    # small inputs (SF10: 0.3 M joined rows) run the one-row-per-thread variant of the same kernel (csrc/agg.cpp)
    import os
    saved = os.environ.get("QHIP_AGG_R")
    os.environ["QHIP_AGG_R"] = "1"
    try:
        out.append(("q3 aggregate, 1 row/thread", planning.aggregate_source(j2.schema(), None, agg.group_exprs, agg.aggregate_exprs)))
    finally:
        if saved is None:
            os.environ.pop("QHIP_AGG_R", None)
        else:
            os.environ["QHIP_AGG_R"] = saved
    # a repeated Q3 runs its joins with DEFERRED sizes (qhip.h: qhip_ctx_allow_deferred_sizes): join 2's build kernel and the
    # aggregate then read their input's row count on the device — separate instantiations of the same bodies
    # ... and read the fixed-width columns of their input (deferred gathers) through the joins' index vectors
    os.environ["QHIP_PLAN_INDIRECT"] = "1"
    os.environ["QHIP_AGG_R"] = "1"
    try:
        out.append(("q3 aggregate, 1 row/thread, indirect columns", planning.aggregate_source(j2.schema(), None, agg.group_exprs, agg.aggregate_exprs)))
        os.environ["QHIP_PLAN_DEV_ROWS"] = "1"
        out.append(("q3 aggregate, 1 row/thread, indirect columns, device-side row count", planning.aggregate_source(j2.schema(), None, agg.group_exprs, agg.aggregate_exprs)))
        # ... the columns of one source table as fields of its record copy (ColRange::rec_buf): price | discount of lineitem in
        # 8-byte records (both narrow), date | priority of orders in 16-byte ones
        js = j2.schema()
        os.environ["QHIP_PLAN_VALUE_BITS"] = f"{js.get_field_index('l_extendedprice')}:24,{js.get_field_index('l_discount')}:4"
        os.environ["QHIP_PLAN_RECORDS"] = (f"{js.get_field_index('l_extendedprice')}:8,{js.get_field_index('l_discount')}:8,"
                                           f"{js.get_field_index('o_orderdate')}:16,{js.get_field_index('o_shippriority')}:16")
        try:
            out.append(("q3 aggregate, indirect columns from record copies", planning.aggregate_source(j2.schema(), None, agg.group_exprs, agg.aggregate_exprs)))
        finally:
            os.environ.pop("QHIP_PLAN_VALUE_BITS", None)
            os.environ.pop("QHIP_PLAN_RECORDS", None)
        os.environ.pop("QHIP_PLAN_INDIRECT", None)   # (join 2's build side gathers its key: measured faster than the indirect read)
        out.append(("q3 join-1 output build entries, device-side row count", planning.scatter_source(j1.schema(), [j2.on[0][0]])))
    finally:
        os.environ.pop("QHIP_PLAN_INDIRECT", None)
        os.environ.pop("QHIP_PLAN_DEV_ROWS", None)
        os.environ.pop("QHIP_AGG_R", None)
        if saved is not None:
            os.environ["QHIP_AGG_R"] = saved
    # the dense (direct-address) join layout: what Q3's joins run since round 3 (integer keys of a small value range,
    # csrc/join.cpp): build + probe kernels of both joins, join 2's build also reading its row count on the device
    os.environ["QHIP_PLAN_DENSE"] = "1"
    try:
        out.append(("q3 customer dense build", planning.scatter_source(CUSTOMER_SCHEMA, [j1.on[0][0]], j1.left.filter)))
        out.append(("q3 join-1 output dense build", planning.scatter_source(j1.schema(), [j2.on[0][0]])))
        out.append(("q3 orders dense probe", planning.probe_source(ORDERS_SCHEMA, [j1.on[0][1]], j1.right.filter)))
        out.append(("q3 lineitem dense probe", planning.probe_source(LINEITEM_Q3_SCHEMA, [j2.on[0][1]], j2.right.filter)))
        # ... with the key column read as its 4-byte narrow copy (TPC-H's keys fit 32 bits: relops.cpp ensure_narrow_int_columns)
        for name, schema, join in (("q3 orders dense probe, narrow key", ORDERS_SCHEMA, j1), ("q3 lineitem dense probe, narrow key", LINEITEM_Q3_SCHEMA, j2)):
            os.environ["QHIP_PLAN_NARROW_INTS"] = str(join.on[0][1].index)
            try:
                out.append((name, planning.probe_source(schema, [join.on[0][1]], join.right.filter)))
            finally:
                os.environ.pop("QHIP_PLAN_NARROW_INTS", None)
        os.environ["QHIP_PLAN_DEV_ROWS"] = "1"
        out.append(("q3 join-1 output dense build, device-side row count", planning.scatter_source(j1.schema(), [j2.on[0][0]])))
        # ... which reads its key through join 1's index vector (a deferred gather, InputCol::indirect)
        os.environ["QHIP_PLAN_INDIRECT"] = "1"
        out.append(("q3 join-1 output dense build, indirect key, device-side row count", planning.scatter_source(j1.schema(), [j2.on[0][0]])))
        os.environ.pop("QHIP_PLAN_DEV_ROWS", None)
        out.append(("q3 join-1 output dense build, indirect key", planning.scatter_source(j1.schema(), [j2.on[0][0]])))
    finally:
        os.environ.pop("QHIP_PLAN_DENSE", None)
        os.environ.pop("QHIP_PLAN_DEV_ROWS", None)
        os.environ.pop("QHIP_PLAN_INDIRECT", None)
    # the exchange of a repartitioned Q3 at 2 / 4 / 8 ranks (qhip_partition_filtered, pass 1): scan filter + key -> part
    for world in (2, 4, 8):
        out.append((f"q3 customer partition x{world}", planning.partition_source(CUSTOMER_SCHEMA, [j1.on[0][0]], j1.left.filter, world)))
        out.append((f"q3 orders partition x{world}", planning.partition_source(ORDERS_SCHEMA, [j1.on[0][1]], j1.right.filter, world)))
        out.append((f"q3 lineitem partition x{world}", planning.partition_source(LINEITEM_Q3_SCHEMA, [j2.on[0][1]], j2.right.filter, world)))
    # ... with the Int64 key read as its 4-byte narrow copy (resident tables whose keys fit 32 bits: TPC-H's)
    for name, schema, join in (("q3 orders partition x8, narrow key", ORDERS_SCHEMA, j1), ("q3 lineitem partition x8, narrow key", LINEITEM_Q3_SCHEMA, j2)):
        os.environ["QHIP_PLAN_NARROW_INTS"] = str(join.on[0][1].index)
        try:
            out.append((name, planning.partition_source(schema, [join.on[0][1]], join.right.filter, 8)))
        finally:
            os.environ.pop("QHIP_PLAN_NARROW_INTS", None)
    # ... pass 2 for the column shapes of Q3's four exchanges (the ranking variant for <= 8 parts serves 2 / 4 / 8 ranks):
    # customer: c_custkey; orders: o_orderkey, o_custkey, o_orderdate, o_shippriority; lineitem: l_orderkey, price, discount;
    # join 1's output: the three columns the plan reads, through join 1's index vectors
    out.append(("partition scatter: customer (8)", planning.part_scatter_source([8], 8)))
    out.append(("partition scatter: orders (8, 8, 4, 8)", planning.part_scatter_source([8, 8, 4, 8], 8)))
    out.append(("partition scatter: lineitem (8, 16, 16)", planning.part_scatter_source([8, 16, 16], 8)))
    out.append(("partition scatter: join-1 output (8, 4, 8), indirect", planning.part_scatter_source([8, 4, 8], 8, [True, True, True])))
    top = queries.q3_top10(*tabs)
    out.append(("q3 order-by keys", planning.sort_keys_source(agg.schema(), [e.expr for e in top.input.exprs])))
    # a Filter node's mask kernel and a Projection with CASE / LIKE (Q12 / Q14 shapes)
    out.append(("filter mask", planning.filter_source(LINEITEM_SCHEMA, queries.q1_full(table).input.filter)))
    out.append(("filter mask, <", planning.filter_source(LINEITEM_SCHEMA, E.BinaryExpr(E.Column("l_shipdate", 0), Operator.Lt, queries._date("1992-01-27")))))
    like = E.Like(False, E.Column("l_returnflag", 1), E.Literal(ScalarValue.Utf8("A%")))
    case = E.CaseExpr([(like, E.Column("l_extendedprice", 4))], E.CastExpr(E.Literal(ScalarValue.Int64(0)), pa.decimal128(15, 2)))
    ratio = E.BinaryExpr(E.Column("l_extendedprice", 4), Operator.Div, E.Column("l_quantity", 3))
    out.append(("projection CASE/LIKE", planning.projection_source(LINEITEM_SCHEMA, [E.Column("l_shipdate", 0), case, ratio])))
    # a CASE that PRODUCES strings (the reference's type.slt:51 shape): the two-pass policy (qk_project + qk_project_copy)
    label = E.CaseExpr([(like, E.Literal(ScalarValue.Utf8("returned")))], E.Column("l_linestatus", 2))
    out.append(("projection CASE -> Utf8 (two passes)", planning.projection_source(LINEITEM_SCHEMA, [label])))
    return out


def compile_catalog(verbose: bool = False):
    for name, src in catalog_sources():
        t = time.time()
        planning.compile_to_cache(src)
        if verbose:
            print(f"[catalog] {name}: compiled for gfx950 in {time.time() - t:.2f}s")
