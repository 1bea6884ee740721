"""What bounds TPC-H Q1's fused kernel? (run on the GPU box)  python tools/q1_ceiling.py [rows]

The same lineitem table through aggregates of decreasing work over the SAME columns, kernel time from HIP events:
  q1          the query (filter, GROUP BY two flags, 8 aggregates)
  nogroup     the same aggregate list and filter without GROUP BY (lane-private accumulators: no cache passes, no LDS table)
  nofilter    ... and without the filter
  sums4       SUM of the four decimal columns, no filter, no GROUP BY (four 16-byte streams, minimal arithmetic)
  sum1        SUM(l_quantity) alone (one 16-byte stream)
Each line: kernel ms, bytes per row the kernel reads, TB/s.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa   # noqa: E402

import bench   # noqa: E402
import qurious_amd as q   # noqa: E402
from qurious_amd import queries   # noqa: E402
from qurious_amd import Column, Scan, SumAggregateExpr   # noqa: E402
from qurious_amd.synth import LINEITEM_SCHEMA   # noqa: E402

DEC = pa.decimal128(15, 2)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 59986052
    q.get_context()
    table = bench.lineitem_table(0, rows, 1 << 20)
    table.device_table()
    full = queries.q1_full(table)
    pred = full.input.filter
    qty, price, disc, tax = Column("l_quantity", 3), Column("l_extendedprice", 4), Column("l_discount", 5), Column("l_tax", 6)

    def nogroup(aggs, names, types, filt):
        schema = pa.schema([pa.field(n, t) for n, t in zip(names, types)])
        return q.NoGroupingAggregate(schema, Scan(LINEITEM_SCHEMA, table, None, filt), aggs)

    names = list(full.schema().names)[2:]
    types = [f.type for f in full.schema()][2:]
    plans = {
        "q1": full,
        "nogroup": nogroup(full.aggregate_exprs, names, types, pred),
        "nofilter": nogroup(full.aggregate_exprs, names, types, None),
        "sums4": nogroup([SumAggregateExpr(c, DEC) for c in (qty, price, disc, tax)], ["a", "b", "c", "d"], [DEC] * 4, None),
        "sum1": nogroup([SumAggregateExpr(qty, DEC)], ["a"], [DEC], None),
    }
    for name, plan in plans.items():
        for _ in range(4):
            plan.execute_device()
        st = bench.operator_stats(lambda: plan.execute_device(), passes=8)[-1]
        ms = st["main_kernel_ms"]
        bpr = st.get("bytes_per_row_read") or 0
        print(f"{name:9s} kernel {ms:7.4f} ms  {bpr:5.1f} B/row  {rows * bpr / ms / 1e9:6.3f} TB/s  ({st['main_kernel_name']})", flush=True)


if __name__ == "__main__":
    main()
