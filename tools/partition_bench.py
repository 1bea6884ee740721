import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, sys
import qurious_amd as q
from qurious_amd import synth, exchange, queries
from qurious_amd.expr import Column
ctx = q.get_context()
c, o, l = synth.q3_tables(10.0)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs)
j2 = plan.input; j1 = j2.left
l_scan = j2.right; o_scan = j1.right
for name, scan, key in (("orders", o_scan, Column("o_custkey", 1)), ("lineitem", l_scan, Column("l_orderkey", 0))):
    for it in range(3):
        ctx.synchronize(); t = time.perf_counter()
        ft = scan.execute_device()
        ctx.synchronize(); t1 = time.perf_counter()
        parts = exchange.partition_by_key(ft, [key], 8)
        # touch the parts (deferred?)
        ctx.synchronize(); t2 = time.perf_counter()
        print(name, "filter ms %.3f partition ms %.3f rows %d -> %s" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, ft.num_rows, [p.num_rows for p in parts][:3]), flush=True)
