"""What bounds the join probe? (GPU) Q3's second join at SF10, timed with HIP events per operator:
  * the probe kernel under the environment's shape switches (QHIP_PROBE_TILES_PER_WAVE, QHIP_DENSE_PROBE_R, ...);
  * the STREAMING FLOOR of the same two probe columns (l_orderkey 8 B + l_shipdate 4 B) through the fused filter +
    aggregate kernel: SELECT COUNT(*), SUM(l_orderkey) FROM lineitem WHERE l_shipdate > DATE (no grouping: one pass, no
    table) — what a kernel that only streams these 12 B/row reaches on this box.
    python tools/probe_floor.py [sf]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
import qurious_amd as q
from qurious_amd import queries, synth
from qurious_amd.exchange import DeviceSource

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ctx = q.get_context()
ctx.set_timing(True)
c, o, l = synth.q3_tables(sf)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs)
j2 = plan.input
j1 = j2.left
rows_l = sum(b.num_rows for b in l)
rows_o = sum(b.num_rows for b in o)
best = None
for it in range(6):
    a = j1.execute_device(); ctx.synchronize(); s1 = ctx.last_stats()
    j2b = q.HashJoinExec.try_new(DeviceSource(j1.schema(), a), j2.right, j2.join_type, j2.on, None)
    b = j2b.execute_device(); ctx.synchronize(); s2 = ctx.last_stats()
    cur = (s1["main_kernel_ms"], s2["main_kernel_ms"], s1["build_ms"], s2["build_ms"])
    best = cur if best is None else tuple(min(x, y) for x, y in zip(best, cur))
print(f"probe kernels: J1 {best[0]*1e3:.1f} us ({rows_o * 12 / best[0] / 1e9:.2f} TB/s of 12 B/row)  J2 {best[1]*1e3:.1f} us ({rows_l * 12 / best[1] / 1e9:.2f} TB/s) | builds {best[2]*1e3:.1f} / {best[3]*1e3:.1f} us | {s2['main_kernel_name']}", flush=True)
# streaming floor of the same columns
scan = j2.right
agg = q.NoGroupingAggregate(pa.schema([pa.field("n", pa.int64()), pa.field("s", pa.int64())]), scan,
                            [q.CountAggregateExpr(q.Literal(q.ScalarValue.Int64(1))), q.SumAggregateExpr(q.Column("l_orderkey", 0), pa.int64())])
t = None
for it in range(6):
    agg.execute_device(); ctx.synchronize(); s = ctx.last_stats()
    t = s["main_kernel_ms"] if t is None else min(t, s["main_kernel_ms"])
print(f"streaming floor (filter + COUNT + SUM over the two probe columns, {s['main_kernel_name']}): {t*1e3:.1f} us = {rows_l * 12 / t / 1e9:.2f} TB/s", flush=True)
o_scan = j1.right
agg = q.NoGroupingAggregate(pa.schema([pa.field("n", pa.int64()), pa.field("s", pa.int64())]), o_scan,
                            [q.CountAggregateExpr(q.Literal(q.ScalarValue.Int64(1))), q.SumAggregateExpr(q.Column("o_custkey", 1), pa.int64())])
t = None
for it in range(6):
    agg.execute_device(); ctx.synchronize(); s = ctx.last_stats()
    t = s["main_kernel_ms"] if t is None else min(t, s["main_kernel_ms"])
print(f"streaming floor, orders (o_custkey + o_orderdate): {t*1e3:.1f} us = {rows_o * 12 / t / 1e9:.2f} TB/s", flush=True)
print(f"plain streaming read of this box: {ctx.measure_stream_read() / 1e3:.2f} TB/s")
