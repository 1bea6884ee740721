"""The numpy restatements that check the HIP path at full size (tests/test_gpu_fullsize.py) are themselves pinned here,
on CPU, against the C oracle at sizes it finishes in a second."""
import numpy as np

import qurious_amd as q
from qurious_amd import queries, synth

from .helpers import rows_of
from .numpy_checks import _days, _dec_lo, _numpy_q1, _unscaled, check_stable_sorted_permutation, numpy_q3, order_row_of, sort_plan_with_rowid


def test_numpy_q1_equals_oracle(oracle):
    batches = synth.lineitem(300_000, batch_rows=65_536)
    plan = queries.q1_full(q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, batches))
    want = {(r[0], r[1]): r[2:] for r in rows_of(oracle.execute(plan))}
    got = _numpy_q1(batches, _days(1998, 9, 2))
    assert len(got) == len(want) == 4
    for g, (cnt, s_qty, s_price, s_dp, s_ch, s_disc) in got.items():
        sum_qty, sum_price, sum_dp, sum_ch, avg_qty, avg_price, avg_disc, count = want[(chr(g >> 8), chr(g & 255))]
        assert (count, _unscaled(sum_qty, 2), _unscaled(sum_price, 2), _unscaled(sum_dp, 4), _unscaled(sum_ch, 6)) == (cnt, s_qty, s_price, s_dp, s_ch)
        assert (_unscaled(avg_qty, 6), _unscaled(avg_price, 6), _unscaled(avg_disc, 6)) == (s_qty * 10**4 // cnt, s_price * 10**4 // cnt, s_disc * 10**4 // cnt)


def test_numpy_q3_equals_oracle(oracle):
    c, o, l = synth.q3_tables(0.02)
    hit, total, mix, odate, okey, per_order = numpy_q3(c, o, l, _days(1995, 3, 15))
    plan = queries.q3(q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
                      q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    rows = rows_of(oracle.execute(plan))
    assert len(rows) == int(hit.sum()) > 100
    assert sum(_unscaled(r[3], 4) for r in rows) == total
    assert sum(_unscaled(r[3], 4) * r[0] for r in rows) % (1 << 64) == mix
    for key, date, prio, rev in rows:
        k = int(order_row_of(np.int64(key)))
        assert hit[k] and okey[k] == key and (date - __import__("datetime").date(1970, 1, 1)).days == odate[k] and prio == 0
        assert per_order[k] == _unscaled(rev, 4)


def test_sort_property_checker_accepts_the_oracle_and_rejects_an_unstable_order(oracle):
    import pyarrow as pa
    import pytest
    _, _, l = synth.q3_tables(0.02)
    plan, batches = sort_plan_with_rowid(l)
    out = oracle.execute(plan)
    assert check_stable_sorted_permutation(out, batches) > 100_000
    # swapping two tied neighbours keeps the keys sorted but breaks the tie-break: the checker must notice
    t = pa.Table.from_batches(out).combine_chunks()
    key, ship = t.column(0).to_numpy(), t.column(1).cast(pa.int32()).to_numpy()
    i = int(np.nonzero((key[1:] == key[:-1]) & (ship[1:] == ship[:-1]))[0][0])
    idx = np.arange(t.num_rows)
    idx[i], idx[i + 1] = i + 1, i
    with pytest.raises(AssertionError):
        check_stable_sorted_permutation(t.take(pa.array(idx)).to_batches(), batches)
