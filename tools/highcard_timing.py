"""GROUP BY with many groups on a big input: time of the aggregate for uniform and Zipf-distributed Int64 keys."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import qurious_amd as q
from qurious_amd import synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
groups = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
ctx = q.get_context()
ctx.set_timing(True)
schema = pa.schema([pa.field("k", pa.int64(), False), pa.field("v", pa.int64(), False)])
rng = np.random.default_rng(3)
for name, keys in (("uniform", rng.integers(0, groups, n)), ("zipf 1.1", synth.zipf_ranks(0, n, groups, 1.1, 7).astype(np.int64))):
    vals = rng.integers(0, 1000, n)
    step = 1 << 20
    batches = [pa.RecordBatch.from_arrays([pa.array(keys[o:o + step]), pa.array(vals[o:o + step])], schema=schema) for o in range(0, n, step)]
    scan = q.Scan(schema, q.MemoryTable.try_new(schema, batches), None, None)
    out_schema = pa.schema([pa.field("k", pa.int64()), pa.field("s", pa.int64()), pa.field("c", pa.int64())])
    plan = q.HashAggregate(out_schema, scan, [q.Column("k", 0)], [q.SumAggregateExpr(q.Column("v", 1), pa.int64()), q.CountAggregateExpr(q.Column("v", 1))])
    for _ in range(3):
        out = plan.execute_device()
    ctx.synchronize()
    ks = []
    t0 = time.perf_counter()
    for _ in range(5):
        out = plan.execute_device()
        ks.append(ctx.last_stats()["main_kernel_ms"])
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = ctx.last_stats()
    print(f"{name}: {n} rows -> {out.num_rows} groups: {dt * 1e3:.2f} ms per aggregate (kernel {sum(ks) / len(ks):.2f} ms, {n / dt / 1e9:.1f} G rows/s, wg {st['workgroups']}, "
          f"lds slots {st['lds_table_slots']})", flush=True)
