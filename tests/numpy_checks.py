"""Vectorised numpy restatements of the benchmark queries on the raw Arrow buffers (exact integer arithmetic): the checkers
of tests/test_gpu_fullsize.py at BASELINE.json's full sizes, themselves pinned against the C oracle at small sizes by
tests/test_numpy_checks.py. Test infrastructure only."""
import datetime
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

U64 = np.uint64


def _days(y, m, d):
    return (datetime.date(y, m, d) - datetime.date(1970, 1, 1)).days


def _dec_lo(column: pa.Array) -> np.ndarray:
    """unscaled values of a Decimal128 column whose values fit 64 bits (asserted), straight from the Arrow buffer"""
    words = np.frombuffer(column.buffers()[1], dtype=np.int64).reshape(-1, 2)[column.offset:column.offset + len(column)]
    assert ((words[:, 1] == 0) | (words[:, 1] == -1)).all()
    return words[:, 0]


def _char(column: pa.Array) -> np.ndarray:
    """the single byte of every value of a Utf8 column whose values are all 1 byte long (asserted)"""
    n = len(column)
    off = np.frombuffer(column.buffers()[1], dtype=np.int32)[column.offset:column.offset + n + 1]
    assert off[-1] - off[0] == n
    return np.frombuffer(column.buffers()[2], dtype=np.uint8)[off[0]:off[0] + n]


def _unscaled(v, scale):
    return int(v.scaleb(scale))


def _numpy_q1(batches, cutoff_incl):
    """exact Q1 aggregate list per (returnflag, linestatus): python-int accumulators over per-batch int64 sums"""
    def one(b):
        ship = np.frombuffer(b.column(0).buffers()[1], dtype=np.int32)[:b.num_rows]
        keep = ship <= cutoff_incl
        gid = (_char(b.column(1)).astype(np.int32) << 8) | _char(b.column(2))
        qty, price, disc, tax = (_dec_lo(b.column(k)) for k in (3, 4, 5, 6))
        disc_price = price * (100 - disc)
        charge = disc_price * (100 + tax)
        out = {}
        for g in np.unique(gid[keep]):
            m = keep & (gid == g)
            out[int(g)] = [int(m.sum())] + [int(a[m].sum()) for a in (qty, price, disc_price, charge, disc)]
        return out
    total = {}
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for part in ex.map(one, batches):
            for g, vals in part.items():
                acc = total.setdefault(g, [0] * 6)
                for k, v in enumerate(vals):
                    acc[k] += v
    return total


def _concat_np(batches, column, dtype):
    return np.concatenate([np.frombuffer(b.column(column).buffers()[1], dtype=dtype)[:b.num_rows] for b in batches])


def numpy_q3(c, o, l, day):
    """Q3's group set from the batch lists of synth.q3_tables: (order-row mask of the groups, total revenue, sum of
    revenue * l_orderkey mod 2^64, o_orderdate per order row, o_orderkey per order row)"""
    ckey = _concat_np(c, 0, np.int64)
    assert (ckey == np.arange(ckey[0], ckey[0] + len(ckey))).all() and ckey[0] == 1
    building = np.concatenate([pc.equal(b.column(1), "BUILDING").to_numpy(zero_copy_only=False) for b in c])
    okey, ocust, odate = _concat_np(o, 0, np.int64), _concat_np(o, 1, np.int64), _concat_np(o, 2, np.int32)
    k = np.arange(len(okey), dtype=np.int64)
    assert (okey == (k // 8) * 32 + k % 8 + 1).all()                      # TPC-H's sparse order keys, ascending
    order_ok = (odate < day) & building[ocust - 1]
    lkey, lship = _concat_np(l, 0, np.int64), _concat_np(l, 1, np.int32)
    price = np.concatenate([_dec_lo(b.column(2)) for b in l])
    disc = np.concatenate([_dec_lo(b.column(3)) for b in l])
    oidx = order_row_of(lkey)
    assert (okey[oidx] == lkey).all()
    keep = (lship > day) & order_ok[oidx]
    revenue = price[keep] * (100 - disc[keep])                            # Decimal(38,4) unscaled
    hit = np.zeros(len(okey), dtype=bool)
    hit[oidx[keep]] = True
    # every group's revenue, at any size: bincount sums in float64, exact while a group's sum stays below 2^53 (an order has at
    # most 7 lines of < 2^40 each)
    assert len(revenue) == 0 or int(revenue.max()) < (1 << 49)
    per_order = np.bincount(oidx[keep], weights=revenue.astype(np.float64), minlength=len(okey)).astype(np.int64)
    mix = int((revenue.astype(U64) * lkey[keep].astype(U64)).sum(dtype=U64))
    return hit, int(revenue.sum()), mix, odate, okey, per_order


def order_row_of(orderkey):
    """inverse of TPC-H's sparse order-key mapping: row of the orders table that holds `orderkey`"""
    return ((orderkey - 1) // 32) * 8 + (orderkey - 1) % 32


def sort_plan_with_rowid(lineitem_batches):
    """Sort(l_shipdate DESC, l_orderkey ASC) over (l_orderkey, l_shipdate, rowid = input row number)"""
    import qurious_amd as q
    from qurious_amd import synth
    schema = pa.schema([synth.LINEITEM_Q3_SCHEMA.field(0), synth.LINEITEM_Q3_SCHEMA.field(1), pa.field("rowid", pa.int64(), False)])
    batches, first = [], 0
    for b in lineitem_batches:
        batches.append(pa.RecordBatch.from_arrays([b.column(0), b.column(1), pa.array(np.arange(first, first + b.num_rows, dtype=np.int64))], schema=schema))
        first += b.num_rows
    scan = q.Scan(schema, q.MemoryTable.try_new(schema, batches), None, None)
    plan = q.Sort([q.PhysicalSortExpr(q.Column("l_shipdate", 1), q.SortOptions(descending=True, nulls_first=True)),
                   q.PhysicalSortExpr(q.Column("l_orderkey", 0), q.SortOptions(descending=False, nulls_first=True))], scan)
    return plan, batches


def check_stable_sorted_permutation(out_batches, in_batches) -> int:
    """the output of sort_plan_with_rowid's plan is THE stable sorted permutation of its input; returns the row count"""
    out = pa.Table.from_batches(out_batches).combine_chunks()
    n = sum(b.num_rows for b in in_batches)
    assert out.num_rows == n
    key, ship, rowid = out.column(0).to_numpy(), out.column(1).cast(pa.int32()).to_numpy(), out.column(2).to_numpy()
    in_key, in_ship = _concat_np(in_batches, 0, np.int64), _concat_np(in_batches, 1, np.int32)
    # permutation: every input row exactly once, carrying its own keys
    assert (np.bincount(rowid, minlength=n) == 1).all()
    assert (in_key[rowid] == key).all() and (in_ship[rowid] == ship).all()
    # sortedness, with the implicit input-row tie-break (sort.rs:48-82: lexsort over the keys + the row index)
    comp = (np.int64(20000) - ship.astype(np.int64)) * (1 << 40) + key
    d = np.diff(comp)
    assert (d >= 0).all() and ((d > 0) | (np.diff(rowid) > 0)).all() and (d == 0).sum() > 100
    return n
