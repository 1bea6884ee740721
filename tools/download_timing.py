import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth
ctx = q.get_context()
c, o, l = synth.q3_tables(10.0)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs)
for _ in range(3):
    out = plan.execute_device(); ctx.synchronize()
    t = time.perf_counter(); b = out.to_batches(); t1 = time.perf_counter() - t
    t = time.perf_counter(); b2 = plan._finish(b) if hasattr(plan, "_finish") else None; t2 = time.perf_counter() - t
    t = time.perf_counter(); r = plan.execute(); t3 = time.perf_counter() - t
    print(f"to_batches {t1*1e3:.3f} ms ({sum(x.nbytes for x in b)/1e6:.1f} MB); execute() total {t3*1e3:.3f} ms", flush=True)
