"""Sweep of the exchange's fused filter + partition (qhip_partition_filtered) over Q3's lineitem side at SF10: one table, many
kernel variants (environment switches read when a plan is lowered; the context forgets its plans between variants).

    python tools/partition_sweep.py [--sf 10] [--slice R/N] "VAR=1 VAR2=2" "..."
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qurious_amd as q  # noqa: E402
from qurious_amd import exchange, queries, synth  # noqa: E402


def main():
    args = sys.argv[1:]
    sf, rank, world = 10.0, 0, 1
    while args and args[0].startswith("--"):
        if args[0] == "--sf":
            sf = float(args[1])
        elif args[0] == "--slice":
            rank, world = (int(x) for x in args[1].split("/"))
        args = args[2:]
    variants = args or [""]
    ctx = q.get_context()
    c, o, l = synth.q3_tables(sf, rank, world)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    plan = queries.q3(*tabs)
    j2 = plan.input
    scan, key = j2.right, j2.on[0][1]
    need = exchange.referenced_columns(list(plan.group_exprs) + [a.expression() for a in plan.aggregate_exprs])
    nl = len(j2.left.schema())
    keep = [(nl + c) in need or c == key.index for c in range(len(scan.schema()))]
    dev = tabs[2].device_table()
    print(f"lineitem rows {dev.num_rows}, keep {keep}", flush=True)
    for v in variants:
        saved = {}
        for kv in v.split():
            k, val = kv.split("=")
            saved[k] = os.environ.get(k)
            os.environ[k] = val
        ctx.synchronize()
        ctx.forget_plans()
        n_parts = int(os.environ.get("PARTS", "8"))
        run = lambda: exchange.partition_filtered(dev, [key], n_parts, predicate=scan.filter, keep=keep)   # noqa: E731
        for _ in range(3):
            parts = run()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        ctx.set_timing(True)
        p1, p2 = [], []
        for _ in range(5):
            run()
            st = ctx.last_stats()
            p1.append(st["build_ms"]); p2.append(st["main_kernel_ms"])
        ctx.set_timing(False)
        print(f"{v or 'default':60s} wall {wall:.3f} ms  pass1 {statistics.median(p1):.3f}  pass2 {statistics.median(p2):.3f}  kept {sum(p.num_rows for p in parts)}", flush=True)
        for k, old in saved.items():
            if old is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old


if __name__ == "__main__":
    main()
