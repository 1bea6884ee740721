"""Per-operator row counts and device times of Q3 (uniform or Zipf-skewed keys) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q
from qurious_amd import queries, synth

skew = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
sf = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ctx = q.get_context()
ctx.set_timing(True)
c, o, l = synth.q3_tables_skewed(sf, skew) if skew > 0 else synth.q3_tables(sf)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs)
j2 = plan.input
j1 = j2.left
for it in range(3):
    t0 = time.perf_counter(); a = j1.execute_device(); ctx.synchronize(); t1 = time.perf_counter(); s1 = ctx.last_stats()
    src = q.exchange.DeviceSource(j1.schema(), a) if hasattr(q, "exchange") else None
    from qurious_amd.exchange import DeviceSource
    j2b = q.HashJoinExec.try_new(DeviceSource(j1.schema(), a), j2.right, j2.join_type, j2.on, None)
    t2 = time.perf_counter(); b = j2b.execute_device(); ctx.synchronize(); t3 = time.perf_counter(); s2 = ctx.last_stats()
    agg = q.HashAggregate(plan.schema(), DeviceSource(j2.schema(), b), plan.group_exprs, plan.aggregate_exprs)
    t4 = time.perf_counter(); g = agg.execute_device(); ctx.synchronize(); t5 = time.perf_counter(); s3 = ctx.last_stats()
    print(f"iter {it}: J1 {a.num_rows} rows {1e3*(t1-t0):.3f} ms (dev {s1['total_device_ms']:.3f}) | J2 {b.num_rows} rows {1e3*(t3-t2):.3f} ms (dev {s2['total_device_ms']:.3f}) | "
          f"agg {g.num_rows} groups {1e3*(t5-t4):.3f} ms (kernel {s3['main_kernel_ms']:.3f}, table {s3['table_capacity']}, retries {s3['retries']}, wg {s3['workgroups']})", flush=True)
