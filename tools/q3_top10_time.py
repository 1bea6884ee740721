"""Time of Q3 with and without its ORDER BY revenue DESC, o_orderdate LIMIT 10 tail (one GPU, SF10)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth
ctx = q.get_context()
c, o, l = synth.q3_tables(10.0)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
for name, plan in (("q3 (to the aggregate)", queries.q3(*tabs)), ("q3 + order by + limit 10", queries.q3_top10(*tabs))):
    for _ in range(3):
        plan.execute_device()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        out = plan.execute_device()
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 20
    t = time.perf_counter(); rows = plan.execute(); t2 = time.perf_counter() - t
    print(f"{name}: {dt * 1e3:.3f} ms per query on the device, {t2 * 1e3:.3f} ms including the download of {sum(b.num_rows for b in rows)} rows", flush=True)
