"""GPU parity of the DENSE (direct-address) hash-join layout (csrc/join.cpp, qh_join_dense_build_body /
qh_join_probe_dense_body): for ONE integer key column whose build-side values span a small range the join table is an
exact bitmap over the range plus row_of[key - min]. JoinHashMap's semantics (hash_join.rs:39-108, 177-216) must not change:
every test compares with the CPU oracle, batch structure and row order included, with the layout chosen automatically,
forced on (QHIP_JOIN_DENSE=2), staged in LDS (QHIP_JOIN_DENSE_LDS=2), forced off (QHIP_JOIN_DENSE=0) and built from a byte map
(QHIP_JOIN_DENSE_BYTEMAP=1)."""
import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, Operator
from qurious_amd import ScalarValue as S

from .helpers import col, rows_of, table_scan

pytestmark = pytest.mark.gpu
I64 = pa.int64()


def _same(got, want):
    assert [b.num_rows for b in got] == [b.num_rows for b in want]
    assert rows_of(got) == rows_of(want)


def _layout_of(ctx):
    return ctx.last_stats()["main_kernel_name"]


@pytest.fixture(params=["auto", "forced", "lds", "off", "bytemap"])
def dense(request, monkeypatch):
    if request.param == "forced":
        monkeypatch.setenv("QHIP_JOIN_DENSE", "2")
    elif request.param == "bytemap":   # (round 4: the bitmap packed from a byte map of plain stores; measured slower, off by default)
        monkeypatch.setenv("QHIP_JOIN_DENSE", "2")
        monkeypatch.setenv("QHIP_JOIN_DENSE_BYTEMAP", "1")
    elif request.param == "lds":
        monkeypatch.setenv("QHIP_JOIN_DENSE", "2")
        monkeypatch.setenv("QHIP_JOIN_DENSE_LDS", "2")
    elif request.param == "off":
        monkeypatch.setenv("QHIP_JOIN_DENSE", "0")
    return request.param


def _sides(rng, key_type, lo, hi, nb, npr, unique=True, null_p=0.05, probe_lo=None, probe_hi=None):
    """build (bk, bv) with keys in [lo, hi], probe (pk, pv) with keys in a wider range (misses on both ends)"""
    span = hi - lo + 1
    if unique:
        bk = lo + rng.permutation(span)[:nb] if span < 50_000_000 else lo + np.unique(rng.integers(0, span, nb * 2))[:nb]
        nb = len(bk)
    else:
        bk = lo + rng.integers(0, span, nb)
    np_t = {pa.int64(): np.int64, pa.int32(): np.int32, pa.uint8(): np.uint8, pa.date32(): np.int32, pa.date64(): np.int64}[key_type]
    plo = probe_lo if probe_lo is not None else lo - span // 4 - 3
    phi = probe_hi if probe_hi is not None else hi + span // 4 + 3
    info = np.iinfo(np_t)
    plo, phi = max(plo, info.min), min(phi, info.max)
    pk = rng.integers(plo, phi, npr, dtype=np.int64, endpoint=True)

    def arr(v, mask):
        a = pa.array(v.astype(np_t), mask=mask)
        return a.cast(key_type) if a.type != key_type else a
    ls = pa.schema([pa.field("bk", key_type), pa.field("bv", I64)])
    rs = pa.schema([pa.field("pk", key_type), pa.field("pv", I64)])
    lb = pa.RecordBatch.from_arrays([arr(np.asarray(bk), rng.random(nb) < null_p), pa.array(rng.integers(0, 1000, nb), I64)], schema=ls)
    rb = pa.RecordBatch.from_arrays([arr(pk, rng.random(npr) < null_p), pa.array(rng.integers(0, 50, npr), I64)], schema=rs)
    return (ls, lb), (rs, rb)


@pytest.mark.parametrize("key_type,lo,hi", [
    (pa.int64(), 1, 150_000),                               # TPC-H c_custkey shape
    (pa.int64(), -70_000, 70_000),                          # range across zero
    (pa.int64(), -(2 ** 63), -(2 ** 63) + 99_999),          # at the type's minimum
    (pa.int64(), 2 ** 63 - 100_000, 2 ** 63 - 1),           # at the type's maximum
    (pa.int64(), 2 ** 40, 2 ** 40 + 3_000_000),             # far from zero, sparse (1 key in ~60)
    (pa.int32(), -(2 ** 31), -(2 ** 31) + 50_000),
    (pa.int32(), 2 ** 31 - 60_000, 2 ** 31 - 1),
    (pa.date32(), 8000, 11_000),
    (pa.date64(), 86_400_000 * 9000, 86_400_000 * 9000 + 200_000),
    (pa.uint8(), 0, 255),
])
def test_dense_layout_every_key_type_and_range(ctx, oracle, dense, key_type, lo, hi):
    rng = np.random.default_rng(abs(lo) % 1000 + 17)
    nb = min(50_000, hi - lo + 1) if key_type != pa.uint8() else 200
    (ls, lb), (rs, rb) = _sides(rng, key_type, lo, hi, nb, 300_000)
    left = table_scan(ls, [lb])
    right = table_scan(rs, [rb.slice(a, 70_000) for a in range(0, 300_000, 70_000)])
    on = [(col("bk", 0), col("pk", 0))]
    for jt in JoinType:
        plan = q.HashJoinExec.try_new(left, right, jt, on, None)
        _same(plan.execute(), oracle.execute(plan))
        if dense in ("forced", "lds", "bytemap"):
            assert _layout_of(ctx).startswith("qk_join_probe_dense"), (jt, _layout_of(ctx))
        elif dense == "off":
            assert _layout_of(ctx) in ("qk_join_probe", "qk_join_probe_onetable")


def test_dense_layout_is_what_runs_for_tpch_shaped_keys_and_not_for_wide_ranges(ctx, oracle, monkeypatch):
    rng = np.random.default_rng(3)
    # keys 1 .. N, every row a key: range == rows
    (ls, lb), (rs, rb) = _sides(rng, I64, 1, 80_000, 80_000, 200_000, null_p=0)
    plan = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb]), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    _same(plan.execute(), oracle.execute(plan))
    st = ctx.last_stats()
    assert st["main_kernel_name"].startswith("qk_join_probe_dense") and st["table_capacity"] == 80_000
    # 3 000 keys scattered over 2^40 values: range >> 256 x rows -> the hashed layout, unless forced — and forcing is
    # refused beyond 2^30 keys
    wide = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 2 ** 40, 3000), I64), pa.array(np.arange(3000), I64)], schema=ls)
    probe = pa.RecordBatch.from_arrays([pa.array(np.concatenate([wide.column(0).to_numpy()[:500], rng.integers(0, 2 ** 40, 9500)]), I64),
                                        pa.array(np.arange(10_000), I64)], schema=rs)
    plan = q.HashJoinExec.try_new(table_scan(ls, [wide]), table_scan(rs, [probe]), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    for mode in ("1", "2"):
        monkeypatch.setenv("QHIP_JOIN_DENSE", mode)
        got = plan.execute()
        assert ctx.last_stats()["main_kernel_name"] in ("qk_join_probe", "qk_join_probe_onetable")
        _same(got, oracle.execute(plan))
    # a modest range with few rows: automatic says no (range > 256 x rows + 64 k), forced says yes
    few = pa.RecordBatch.from_arrays([pa.array(rng.permutation(4_000_000)[:100], I64), pa.array(np.arange(100), I64)], schema=ls)
    plan = q.HashJoinExec.try_new(table_scan(ls, [few]), table_scan(rs, [probe]), JoinType.Left, [(col("bk", 0), col("pk", 0))], None)
    monkeypatch.setenv("QHIP_JOIN_DENSE", "1")
    _same(plan.execute(), oracle.execute(plan))
    assert ctx.last_stats()["main_kernel_name"] in ("qk_join_probe", "qk_join_probe_onetable")
    monkeypatch.setenv("QHIP_JOIN_DENSE", "2")
    _same(plan.execute(), oracle.execute(plan))
    assert ctx.last_stats()["main_kernel_name"].startswith("qk_join_probe_dense")


def test_dense_layout_duplicate_build_keys_fall_back_to_the_chains_order(ctx, oracle, dense):
    """a second build row with an equal key: detected by the build kernel's returning bit-set, remembered, and the join
    runs again with the hashed / CSR layout, whose pairs come in the reference's ascending-chain order (hash_join.rs:164)"""
    rng = np.random.default_rng(21)
    (ls, lb), (rs, rb) = _sides(rng, I64, 100, 5000, 12_000, 60_000, unique=False)
    left, right = table_scan(ls, [lb.slice(0, 5000), lb.slice(5000, 7000)]), table_scan(rs, [rb.slice(0, 30_000), rb.slice(30_000, 30_000)])
    for jt in JoinType:
        for _ in range(2):   # (the second execution starts from the remembered duplicate flag)
            plan = q.HashJoinExec.try_new(left, right, jt, [(col("bk", 0), col("pk", 0))], None)
            _same(plan.execute(), oracle.execute(plan))
    # exactly ONE duplicated key among 40 000 unique ones, in the last rows
    keys = np.concatenate([rng.permutation(100_000)[:40_000], [7]])
    keys[-1] = keys[5]
    one = pa.RecordBatch.from_arrays([pa.array(keys, I64), pa.array(np.arange(len(keys)), I64)], schema=ls)
    plan = q.HashJoinExec.try_new(table_scan(ls, [one]), right, JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
    _same(plan.execute(), oracle.execute(plan))


def test_dense_layout_fused_scan_filters_residual_filter_and_expression_probe_key(ctx, oracle, dense):
    rng = np.random.default_rng(33)
    nb, npr = 30_000, 400_000
    ls = pa.schema([pa.field("bk", I64), pa.field("seg", pa.string()), pa.field("bv", I64)])
    rs = pa.schema([pa.field("pk", I64), pa.field("d", pa.date32()), pa.field("pv", I64)])
    segs = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]
    lb = pa.RecordBatch.from_arrays([pa.array(np.arange(1, nb + 1), I64), pa.array([segs[k] for k in rng.integers(0, 5, nb)]),
                                     pa.array(rng.integers(0, 100, nb), I64)], schema=ls)
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(-50, nb + 2000, npr), I64, mask=rng.random(npr) < 0.02),
                                     pa.array(rng.integers(9000, 9400, npr), pa.int32()).cast(pa.date32()),
                                     pa.array(rng.integers(0, 100, npr), I64)], schema=rs)
    lpred = q.BinaryExpr(col("seg", 1), Operator.Eq, q.Literal(S.Utf8("BUILDING")))
    rpred = q.BinaryExpr(col("d", 1), Operator.Lt, q.Literal(S.Date32(9204)))
    left = table_scan(ls, [lb], lpred)
    right = table_scan(rs, [rb.slice(a, 100_000) for a in range(0, npr, 100_000)], rpred)
    on = [(col("bk", 0), col("pk", 0))]
    plan = q.HashJoinExec.try_new(left, right, JoinType.Inner, on, None)
    _same(plan.execute(), oracle.execute(plan))
    # residual filter bv < pv, every join type (no fused filters there: separate Filter semantics)
    fschema = pa.schema([pa.field("bv", I64), pa.field("pv", I64)])
    jf = q.JoinFilter(q.BinaryExpr(col("bv", 0), Operator.Lt, col("pv", 1)), [(2, q.JoinSide.Left), (2, q.JoinSide.Right)], fschema)
    for jt in JoinType:
        plan = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb.slice(0, 50_000), rb.slice(50_000, 30_000)]), jt, on, jf)
        _same(plan.execute(), oracle.execute(plan))
    # the probe key may be any expression of the key's type: pk + 1 (the build key must be a plain column)
    plan = q.HashJoinExec.try_new(left, right, JoinType.Inner, [(col("bk", 0), q.BinaryExpr(col("pk", 0), Operator.Add, q.Literal(S.Int64(1))))], None)
    _same(plan.execute(), oracle.execute(plan))
    # ... and a computed BUILD key takes the hashed layout
    plan = q.HashJoinExec.try_new(table_scan(ls, [lb]), right, JoinType.Inner, [(q.BinaryExpr(col("bk", 0), Operator.Add, q.Literal(S.Int64(1))), col("pk", 0))], None)
    _same(plan.execute(), oracle.execute(plan))
    assert _layout_of(ctx) in ("qk_join_probe", "qk_join_probe_onetable")


def test_dense_layout_build_side_is_a_join_of_deferred_size_under_an_aggregate(ctx, oracle, dense):
    """Q3's shape: (A join B) as the build side of a join with C, aggregated; repeated executions run the first join with its
    size left on the device (qk_join_dense_build reads the row count there), key read from a gathered column whose value range
    is inherited from B's base column"""
    rng = np.random.default_rng(44)
    na, nbb, nc = 20_000, 200_000, 900_000
    a_s = pa.schema([pa.field("ak", I64), pa.field("av", I64)])
    b_s = pa.schema([pa.field("bk", I64), pa.field("b_ak", I64), pa.field("bd", I64)])
    c_s = pa.schema([pa.field("c_bk", I64), pa.field("cv", I64)])
    A = pa.RecordBatch.from_arrays([pa.array(np.arange(1, na + 1), I64), pa.array(rng.integers(0, 5, na), I64)], schema=a_s)
    Bk = (np.arange(nbb) // 8) * 32 + np.arange(nbb) % 8 + 1      # TPC-H's sparse order keys
    B = pa.RecordBatch.from_arrays([pa.array(Bk, I64), pa.array(rng.integers(1, na * 2, nbb), I64), pa.array(rng.integers(0, 7, nbb), I64)], schema=b_s)
    Ck = np.sort(Bk[rng.integers(0, nbb, nc)])                      # clustered by key like lineitem
    C = pa.RecordBatch.from_arrays([pa.array(Ck, I64), pa.array(rng.integers(0, 1000, nc), I64)], schema=c_s)
    apred = q.BinaryExpr(col("av", 1), Operator.Eq, q.Literal(S.Int64(2)))
    ta, tb, tc = table_scan(a_s, [A], apred), table_scan(b_s, [B.slice(k, 50_000) for k in range(0, nbb, 50_000)]), table_scan(c_s, [C.slice(k, 300_000) for k in range(0, nc, 300_000)])
    want = None
    for execution in range(4):
        j1 = q.HashJoinExec.try_new(ta, tb, JoinType.Inner, [(col("ak", 0), col("b_ak", 1))], None)
        j2 = q.HashJoinExec.try_new(j1, tc, JoinType.Inner, [(col("bk", 2), col("c_bk", 0))], None)
        agg = q.HashAggregate(pa.schema([pa.field("bd", I64), pa.field("n", I64), pa.field("s", I64)]), j2, [col("bd", 4)],
                              [q.CountAggregateExpr(q.Literal(S.Int64(1))), q.SumAggregateExpr(col("cv", 6), I64)])
        if want is None:
            want = sorted(rows_of(oracle.execute(agg)))
        assert sorted(rows_of(agg.execute())) == want, execution
    # from the second execution on the plan waits for the device once, dense or not
    before = ctx.sync_count()
    agg.execute_device()
    assert ctx.sync_count() - before == 1


def test_dense_layout_empty_and_degenerate_inputs(ctx, oracle, dense):
    ls = pa.schema([pa.field("bk", I64), pa.field("bv", I64)])
    rs = pa.schema([pa.field("pk", I64), pa.field("pv", I64)])

    def t(schema, ks, mask=None):
        ks = np.asarray(ks, dtype=np.int64)
        return table_scan(schema, [pa.RecordBatch.from_arrays([pa.array(ks, I64, mask=mask), pa.array(np.arange(len(ks)), I64)], schema=schema)])
    cases = [
        (t(ls, [5]), t(rs, [5, 5, 4, 6, 5])),                                             # one key: a range of one value
        (t(ls, [7, 8, 9], mask=np.array([True, True, True])), t(rs, [7, 8, 9])),          # every build key NULL
        (t(ls, [1, 2, 3]), t(rs, [1, 2, 3], mask=np.array([True, True, True]))),          # every probe key NULL
        (t(ls, [-1, 0, 1]), t(rs, [2 ** 63 - 1, -(2 ** 63), 0, -1, 1, 2, -2])),           # probe keys at the type's ends (idx wraps)
        (t(ls, np.arange(64)), t(rs, np.arange(-64, 128))),                               # exactly one bitmap word
        (t(ls, np.arange(33)), t(rs, [])),                                                # no probe rows
    ]
    for left, right in cases:
        for jt in JoinType:
            plan = q.HashJoinExec.try_new(left, right, jt, [(col("bk", 0), col("pk", 0))], None)
            _same(plan.execute(), oracle.execute(plan))
