"""GPU parity + host round trips of joins of DEFERRED SIZE (include/qhip.h: qhip_ctx_allow_deferred_sizes).

A hash join that feeds a HashAggregate or another join's build side remembers how many pairs it produced; the next time
the same join runs it does not wait for its pair total: the row count stays on the device and the consumer's one
synchronisation checks it. These tests pin (a) the results, always against the oracle, (b) the number of host waits a
repeated Q3 costs (one: the aggregate's), (c) what happens when the remembered size does not hold — more pairs than the
room, duplicate build keys — and (d) every other reader of such a table (Filter, Projection, Limit, Sort, export)."""
import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator, queries, synth

from .helpers import col, lit_i64, rows_of

pytestmark = pytest.mark.gpu
I64 = pa.int64()


def _table(names, arrays, batch_rows=None):
    schema = pa.schema([pa.field(n, a.type, True) for n, a in zip(names, arrays)])
    n = len(arrays[0])
    step = batch_rows or max(n, 1)
    batches = [pa.RecordBatch.from_arrays([a.slice(o, step) for a in arrays], schema=schema) for o in range(0, max(n, 1), step)]
    return schema, q.MemoryTable.try_new(schema, batches)


def _join_agg(build, probe, group_col=3):
    """SELECT pv, COUNT(*), SUM(bv) FROM build JOIN probe ON bk = pk GROUP BY pv — build (bk, bv), probe (pk, pv)"""
    (bs, bt), (ps, pt) = build, probe
    j = q.HashJoinExec.try_new(q.Scan(bs, bt), q.Scan(ps, pt), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    schema = pa.schema([pa.field("pv", I64), pa.field("n", I64), pa.field("s", I64)])
    agg = q.HashAggregate(schema, j, [col("pv", group_col)], [q.CountAggregateExpr(q.Literal(q.ScalarValue.Int64(1))),
                                                              q.SumAggregateExpr(col("bv", 1), I64)])
    return agg, j


def _syncs(ctx, fn):
    before = ctx.sync_count()
    out = fn()
    return out, ctx.sync_count() - before


def test_repeated_q3_waits_for_the_device_once(ctx, oracle, monkeypatch):
    c, o, l = synth.q3_tables(0.05, orders_per_batch=4096)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    plan = queries.q3(*tabs)
    want = sorted(rows_of(oracle.execute(plan)))
    assert sorted(rows_of(plan.execute())) == want and len(want) > 200   # first run: the joins wait and remember
    plan.execute_device()                                                 # (statistics of the gathered columns settle)
    t, waits = _syncs(ctx, plan.execute_device)
    assert waits == 1, waits          # the aggregate's; both joins left their sizes on the device
    got = sorted(rows_of([_b for _b in t.to_batches()]))
    assert got == want
    # the same plan with deferral switched off: one wait per join more
    monkeypatch.setenv("QHIP_JOIN_NO_DEFER", "1")
    t2, waits2 = _syncs(ctx, plan.execute_device)
    assert waits2 == 3, waits2
    assert sorted(rows_of(t2.to_batches())) == want
    monkeypatch.delenv("QHIP_JOIN_NO_DEFER")
    # ... and the join alone (nobody above to check a deferred size) always waits
    j2 = plan.input
    tj, waits_j = _syncs(ctx, j2.execute_device)
    assert waits_j >= 1 and tj.num_rows == sum(b.num_rows for b in oracle.execute(j2))


def test_more_pairs_than_remembered_runs_the_input_again(ctx, oracle):
    """same join (expressions, row counts), different data: first few matches, then every probe row matches"""
    rng = np.random.default_rng(5)
    nb, npr = 6000, 80000
    build = _table(["bk", "bv"], [pa.array(np.arange(nb), I64), pa.array(rng.integers(0, 1000, nb), I64)])
    few = _table(["pk", "pv"], [pa.array(rng.integers(nb - 50, nb * 40, npr), I64), pa.array(rng.integers(0, 37, npr), I64)], 8192)
    many = _table(["pk", "pv"], [pa.array(rng.integers(0, nb, npr), I64), pa.array(rng.integers(0, 37, npr), I64)], 8192)
    for probe in (few, few, many, many, few):
        agg, j = _join_agg(build, probe)
        want = sorted(rows_of(oracle.execute(agg)))
        (t, waits) = _syncs(ctx, agg.execute_device)
        assert sorted(rows_of(t.to_batches())) == want
    # the last three runs: `many` after `few` had too little room (ran twice), `many` again and `few` after it fit
    agg, _ = _join_agg(build, many)
    agg.execute_device()
    _, waits = _syncs(ctx, agg.execute_device)
    assert waits == 1


def test_duplicate_build_keys_behind_a_remembered_size(ctx, oracle):
    rng = np.random.default_rng(6)
    nb, npr = 5000, 30000
    uniq = _table(["bk", "bv"], [pa.array(rng.permutation(nb), I64), pa.array(rng.integers(0, 9, nb), I64)])
    dup = _table(["bk", "bv"], [pa.array(rng.integers(0, nb // 2, nb), I64), pa.array(rng.integers(0, 9, nb), I64)])
    probe = _table(["pk", "pv"], [pa.array(rng.integers(0, nb, npr), I64), pa.array(rng.integers(0, 11, npr), I64)], 4096)
    for build in (uniq, uniq, dup, dup, uniq):
        agg, _ = _join_agg(build, probe)
        assert sorted(rows_of(agg.execute())) == sorted(rows_of(oracle.execute(agg)))


def test_other_readers_of_a_table_of_deferred_size(ctx, oracle):
    """Filter / Projection / Limit / Sort between the join and the aggregate, a join of deferred size as the build side of a
    join that is read directly: each makes the row count exact by itself"""
    rng = np.random.default_rng(7)
    nb, npr = 4096, 50000
    build = _table(["bk", "bv"], [pa.array(rng.permutation(nb), I64), pa.array(rng.integers(0, 100, nb), I64, mask=rng.random(nb) < 0.1)])
    probe = _table(["pk", "pv"], [pa.array(rng.integers(0, nb * 3, npr), I64), pa.array(rng.integers(0, 23, npr), I64)], 4096)
    (bs, bt), (ps, pt) = build, probe

    def join():
        return q.HashJoinExec.try_new(q.Scan(bs, bt), q.Scan(ps, pt), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)

    schema = pa.schema([pa.field("pv", I64), pa.field("n", I64), pa.field("s", I64)])
    aggs = [q.CountAggregateExpr(q.Literal(q.ScalarValue.Int64(1))), q.SumAggregateExpr(col("bv", 1), I64)]
    js = join().schema()
    between = [
        lambda j: q.Filter(j, q.BinaryExpr(col("pv", 3), Operator.Gt, lit_i64(4))),
        lambda j: q.Projection(js, j, [col("bk", 0), col("bv", 1), col("pk", 2), col("pv", 3)]),
        lambda j: q.Limit(j, 7000, 100),
        lambda j: q.Sort([q.PhysicalSortExpr(col("pv", 3), q.SortOptions()), q.PhysicalSortExpr(col("pk", 2), q.SortOptions())], j),
    ]
    for mk in between:
        for _ in range(3):
            plan = q.HashAggregate(schema, mk(join()), [col("pv", 3)], aggs)
            assert sorted(rows_of(plan.execute()), key=str) == sorted(rows_of(oracle.execute(plan)), key=str)
    # a join of deferred size as the build side of a second join whose result is exported (ordered, batch structure)
    small = _table(["k2", "w"], [pa.array(rng.integers(0, nb * 3, 9000), I64), pa.array(rng.integers(0, 5, 9000), I64)], 2048)
    for _ in range(3):
        j2 = q.HashJoinExec.try_new(join(), q.Scan(*small), JoinType.Inner, [(col("pk", 2), col("k2", 0))], None)
        got, want = j2.execute(), oracle.execute(j2)
        assert [b.num_rows for b in got] == [b.num_rows for b in want] and rows_of(got) == rows_of(want)
        agg2 = q.HashAggregate(pa.schema([pa.field("w", I64), pa.field("n", I64)]), j2, [col("w", 5)], [aggs[0]])
        assert sorted(rows_of(agg2.execute())) == sorted(rows_of(oracle.execute(agg2)))


def test_no_matches_after_many_and_back(ctx, oracle):
    rng = np.random.default_rng(8)
    nb, npr = 3000, 20000
    build = _table(["bk", "bv"], [pa.array(np.arange(nb), I64), pa.array(rng.integers(0, 9, nb), I64)])
    hit = _table(["pk", "pv"], [pa.array(rng.integers(0, nb, npr), I64), pa.array(rng.integers(0, 5, npr), I64)], 4096)
    miss = _table(["pk", "pv"], [pa.array(rng.integers(nb, nb * 2, npr), I64), pa.array(rng.integers(0, 5, npr), I64)], 4096)
    for probe in (hit, hit, miss, miss, hit, hit):
        agg, _ = _join_agg(build, probe)
        got, want = agg.execute(), oracle.execute(agg)
        assert [b.num_rows for b in got] == [b.num_rows for b in want]
        assert sorted(rows_of(got)) == sorted(rows_of(want))


@pytest.mark.parametrize("join_type", [JoinType.Left, JoinType.Full, JoinType.LeftSemi, JoinType.LeftAnti, JoinType.Right, JoinType.Inner])
def test_join_with_a_build_side_tail_over_a_join_of_deferred_size(ctx, oracle, join_type):
    """A join whose BUILD side is an Inner join of deferred size and whose probe side is a table scan takes that build side
    with its row count on the device (plan.py: `_feeding(self.left)` for every join type). Join types with a build-side tail
    (Left / Full / LeftSemi / LeftAnti) scan the visited bitmap over the build rows: the pad rows [count, capacity) of a
    deferred table are never inserted and never visited, so unless the build side is made exact first they surface as
    phantom unmatched rows from the second execution on (round-2 advisor finding, join.cpp). Every execution — the first
    (sizes waited for), the second and third (remembered) — must equal the oracle, batch structure included."""
    rng = np.random.default_rng(11)
    nb, npr, n3 = 5000, 40000, 6000
    build = _table(["bk", "bv"], [pa.array(rng.permutation(nb), I64), pa.array(rng.integers(0, 100, nb), I64)])
    # about a third of the probe rows match: the inner join's capacity (count + 1/8 + 1024) leaves > 1000 pad rows
    probe = _table(["pk", "pv"], [pa.array(rng.integers(0, nb * 3, npr), I64), pa.array(rng.integers(0, 23, npr), I64)], 4096)
    third = _table(["k3", "w"], [pa.array(rng.integers(0, nb * 4, n3), I64, mask=rng.random(n3) < 0.05), pa.array(rng.integers(0, 5, n3), I64)], 2048)
    (bs, bt), (ps, pt) = build, probe
    for execution in range(4):
        inner = q.HashJoinExec.try_new(q.Scan(bs, bt), q.Scan(ps, pt), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
        outer = q.HashJoinExec.try_new(inner, q.Scan(*third), join_type, [(col("pk", 2), col("k3", 0))], None)
        got, want = outer.execute(), oracle.execute(outer)
        assert [b.num_rows for b in got] == [b.num_rows for b in want], (execution, join_type)
        assert rows_of(got) == rows_of(want), (execution, join_type)
        # ... and under an aggregate (the outer join may itself defer when it is Inner)
        if join_type in (JoinType.LeftSemi, JoinType.LeftAnti):
            g, v = col("pv", 3), col("bv", 1)
        else:
            g, v = col("w", 5), col("bv", 1)
        agg = q.HashAggregate(pa.schema([pa.field("g", I64), pa.field("n", I64), pa.field("s", I64)]), outer, [g],
                              [q.CountAggregateExpr(q.Literal(q.ScalarValue.Int64(1))), q.SumAggregateExpr(v, I64)])
        assert sorted(rows_of(agg.execute()), key=str) == sorted(rows_of(oracle.execute(agg)), key=str), (execution, join_type)
