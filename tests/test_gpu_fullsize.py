"""GPU parity at BASELINE.json's FULL sizes, where the C oracle would take minutes: the HIP path is checked against
independent vectorised numpy restatements (exact integer arithmetic on the raw Arrow buffers) and through
size-independent properties — partition linearity of the aggregates, a checksum of checksums over the join's groups,
sortedness + permutation + stability of the sort, idempotence of re-execution.

  configs[1]/[2]  100M synthetic lineitem rows in 2^20-row batches: Q1-mini and the full Q1 aggregate list
  configs[3]      TPC-H Q3 at SF10 (1.5M customers, 15M orders, ~60M lineitems)
  §8f rank 1      ORDER BY over the ~60M-row SF10 lineitem table
"""
import datetime
import gc
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import qurious_amd as q
from qurious_amd import queries, synth

from .helpers import col, rows_of
from .numpy_checks import U64, _char, _concat_np, _days, _dec_lo, _numpy_q1, _unscaled, check_stable_sorted_permutation, numpy_q3, order_row_of, sort_plan_with_rowid

pytestmark = pytest.mark.gpu

N_ROWS = 100_000_000
BATCH = 1 << 20
U64 = np.uint64


@pytest.fixture(scope="module")
def lineitem_100m():
    starts = list(range(0, N_ROWS, BATCH))
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        batches = list(ex.map(lambda s: synth.lineitem_batch(s, min(BATCH, N_ROWS - s)), starts))
    yield batches
    del batches
    gc.collect()


def test_q1_at_100m_rows_equals_exact_numpy_and_is_partition_linear(ctx, lineitem_100m):
    table = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, lineitem_100m)
    plan = queries.q1_full(table)
    got = {(r[0], r[1]): r[2:] for r in rows_of(plan.execute())}
    want = _numpy_q1(lineitem_100m, _days(1998, 9, 2))
    assert len(got) == len(want) == 4
    rows_seen = 0
    for g, (cnt, s_qty, s_price, s_dp, s_ch, s_disc) in want.items():
        key = (chr(g >> 8), chr(g & 255))
        sum_qty, sum_price, sum_dp, sum_ch, avg_qty, avg_price, avg_disc, count = got[key]
        assert count == cnt
        assert (_unscaled(sum_qty, 2), _unscaled(sum_price, 2), _unscaled(sum_dp, 4), _unscaled(sum_ch, 6)) == (s_qty, s_price, s_dp, s_ch)
        # DecimalAvgAccumulator (aggregate/avg.rs:105-106): (sum * 10^4) div count, truncating
        assert (_unscaled(avg_qty, 6), _unscaled(avg_price, 6), _unscaled(avg_disc, 6)) == (s_qty * 10**4 // cnt, s_price * 10**4 // cnt, s_disc * 10**4 // cnt)
        rows_seen += cnt
    assert 95_000_000 < rows_seen < N_ROWS            # the predicate keeps ~98% of the rows
    # idempotence: the operator is re-entrant (physical/plan/mod.rs:25-29), its arena and plan cache are reused
    assert {(r[0], r[1]): r[2:] for r in rows_of(plan.execute())} == got
    # partition linearity: aggregating two halves separately and merging gives the whole (SUM/COUNT add up)
    half = len(lineitem_100m) // 2
    parts = [rows_of(queries.q1_full(q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, lineitem_100m[a:b])).execute())
             for a, b in ((0, half), (half, len(lineitem_100m)))]
    for key, whole in got.items():
        pieces = [r[2:] for p in parts for r in p if (r[0], r[1]) == key]
        for k in (0, 1, 2, 3, 7):
            assert sum(p[k] for p in pieces) == whole[k]


def test_q1_mini_at_100m_rows_equals_exact_numpy(ctx, lineitem_100m):
    """configs[1], the headline workload: SUM(l_quantity) GROUP BY l_returnflag WHERE l_shipdate < 1998-09-01"""
    table = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, lineitem_100m)
    got = dict(rows_of(queries.q1_mini(table).execute()))
    cutoff = _days(1998, 9, 1)
    want = {}
    for b in lineitem_100m:
        ship = np.frombuffer(b.column(0).buffers()[1], dtype=np.int32)[:b.num_rows]
        keep = ship < cutoff
        sums = np.bincount(_char(b.column(1))[keep], weights=None, minlength=256)   # row counts per flag (sanity)
        qty = _dec_lo(b.column(3))
        flags = _char(b.column(1))
        for f in np.nonzero(sums)[0]:
            want[chr(f)] = want.get(chr(f), 0) + int(qty[keep & (flags == f)].sum())
    assert {k: _unscaled(v, 2) for k, v in got.items()} == want and sorted(want) == ["A", "N", "R"]


def test_q3_at_sf10_group_set_and_checksums_equal_numpy(ctx):
    c, o, l = synth.q3_tables(10.0)
    day = _days(1995, 3, 15)
    hit, want_total, want_mix, odate, okey, per_order = numpy_q3(c, o, l, day)   # independent restatement on the raw buffers
    want_groups = int(hit.sum())
    # ---- the HIP path
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    plan = queries.q3(*tabs)
    out = pa.Table.from_batches(plan.execute()).combine_chunks()
    assert out.num_rows == want_groups and want_groups > 100_000
    g_key = out.column(0).to_numpy()
    g_rev = _dec_lo(out.column(3).chunk(0))
    assert len(np.unique(g_key)) == want_groups and hit[order_row_of(g_key)].all()
    assert int(g_rev.sum()) == want_total
    assert int((g_rev.astype(U64) * g_key.astype(U64)).sum(dtype=U64)) == want_mix
    g_idx = order_row_of(g_key)
    assert (out.column(1).cast(pa.int32()).to_numpy() == odate[g_idx]).all() and (out.column(2).to_numpy() == 0).all()
    assert (g_rev == per_order[g_idx]).all()                                      # every group's revenue, exactly
    # idempotence + the device-resident top-10 agrees with a host-side ordering of the full result
    again = pa.Table.from_batches(plan.execute()).combine_chunks()
    assert int(_dec_lo(again.column(3).chunk(0)).sum()) == want_total and again.num_rows == want_groups
    # a repeated Q3 waits for the device ONCE (the aggregate's read-back): both joins leave their sizes on the device
    # (tests/test_gpu_deferred_sizes.py), the 113 k groups are assembled there too
    before = ctx.sync_count()
    t = plan.execute_device()
    assert ctx.sync_count() - before == 1
    third = pa.Table.from_batches(t.to_batches()).combine_chunks()
    assert int(_dec_lo(third.column(3).chunk(0)).sum()) == want_total and third.num_rows == want_groups
    top = rows_of(queries.q3_top10(*tabs).execute())
    order = np.lexsort((out.column(1).cast(pa.int32()).to_numpy(), -g_rev))[:10]
    assert [r[0] for r in top] == [int(g_key[i]) for i in order]
    assert [_unscaled(r[3], 4) for r in top] == [int(g_rev[i]) for i in order]


def test_sort_60m_rows_is_the_stable_sorted_permutation(ctx):
    """Sort (physical/plan/sort.rs:48-82) over SF10's lineitem: l_shipdate DESC, l_orderkey ASC with the implicit
    input-row tie-break — checked as sortedness + permutation + stability, without an oracle run"""
    _, _, l = synth.q3_tables(10.0)
    plan, batches = sort_plan_with_rowid(l)
    n = check_stable_sorted_permutation(plan.execute(), batches)
    assert n > 59_000_000


def test_q3_sf100_zipf_slice_of_one_rank_equals_exact_numpy(ctx, monkeypatch):
    """BASELINE configs[4] at its own workload, as far as one GPU goes: rank 0's 1/8 slice of TPC-H Q3 at SF100 with the join
    keys re-drawn from Zipf(1.1) (75 M lineitem rows, 18.75 M orders, 1.875 M customers; hot keys in both joins and in the
    GROUP BY; probe keys far outside the slice's build-key range) through the HIP path, against an exact numpy restatement
    on the raw buffers: group set, total revenue, key-weighted checksum mod 2^64, every group's date, the top-10, and the LDS
    hash-table occupancy figures the config asks to be reported."""
    c, o, l = synth.q3_tables_skewed(100.0, 1.1, 0, 8)
    day = _days(1995, 3, 15)
    # ---- exact restatement for a SLICE: keys may point outside the slice's customers / orders (no match there)
    import pyarrow.compute as pc
    from .numpy_checks import _concat_np
    n_c = sum(b.num_rows for b in c)
    building = np.concatenate([pc.equal(b.column(1), "BUILDING").to_numpy(zero_copy_only=False) for b in c])
    okey, ocust, odate = _concat_np(o, 0, np.int64), _concat_np(o, 1, np.int64), _concat_np(o, 2, np.int32)
    n_o = len(okey)
    k = np.arange(n_o, dtype=np.int64)
    assert (okey == (k // 8) * 32 + k % 8 + 1).all()
    in_c = ocust <= n_c
    order_ok = (odate < day) & in_c & building[np.minimum(ocust, n_c) - 1]
    lkey, lship = _concat_np(l, 0, np.int64), _concat_np(l, 1, np.int32)
    price = np.concatenate([_dec_lo(b.column(2)) for b in l])
    disc = np.concatenate([_dec_lo(b.column(3)) for b in l])
    oidx = order_row_of(lkey)
    in_o = (oidx < n_o) & ((lkey - 1) % 32 < 8)
    keep = (lship > day) & in_o & order_ok[np.minimum(oidx, n_o - 1)]
    revenue = price[keep] * (100 - disc[keep])
    per_order = np.zeros(n_o, dtype=np.int64)
    np.add.at(per_order, oidx[keep], revenue)
    hit = np.zeros(n_o, dtype=bool)
    hit[oidx[keep]] = True
    want_groups, want_total = int(hit.sum()), int(revenue.sum())
    want_mix = int((revenue.astype(U64) * lkey[keep].astype(U64)).sum(dtype=U64))
    assert want_groups > 100_000 and int(keep.sum()) > 5 * want_groups          # skew: many rows per hot group
    # ---- the HIP path (three executions: sizes waited for, then remembered)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    plan = queries.q3(*tabs)
    for execution in range(3):
        out = pa.Table.from_batches(plan.execute()).combine_chunks()
        assert out.num_rows == want_groups, execution
        g_key = out.column(0).to_numpy()
        g_rev = _dec_lo(out.column(3).chunk(0))
        g_idx = order_row_of(g_key)
        assert len(np.unique(g_key)) == want_groups and hit[g_idx].all()
        assert int(g_rev.sum()) == want_total
        assert int((g_rev.astype(U64) * g_key.astype(U64)).sum(dtype=U64)) == want_mix
        assert (g_rev == per_order[g_idx]).all()                                  # every group's revenue, exactly
        assert (out.column(1).cast(pa.int32()).to_numpy() == odate[g_idx]).all() and (out.column(2).to_numpy() == 0).all()
    top = rows_of(queries.q3_top10(*tabs).execute())
    order = np.lexsort((out.column(1).cast(pa.int32()).to_numpy(), -g_rev))[:10]
    assert [r[0] for r in top] == [int(g_key[i]) for i in order]
    assert [_unscaled(r[3], 4) for r in top] == [int(g_rev[i]) for i in order]
    # the occupancy figures configs[4] asks for (instrumented aggregate)
    monkeypatch.setenv("QHIP_AGG_STATS", "1")
    again = pa.Table.from_batches(plan.execute()).combine_chunks()
    st = ctx.last_stats()
    assert again.num_rows == want_groups and st["groups"] == want_groups
    assert st["lds_table_slots"] > 0 and 0.0 < st["lds_occupancy"] <= 1.0 and 0.0 < st["hbm_table_load"] <= 1.0
