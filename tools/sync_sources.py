"""Where do the host waits of one REPEATED multi-rank Q3 come from? One-rank rehearsal (QHIP_EXCHANGE_FORCE=1, libqhip's RCCL
transport) with QHIP_SYNC_TRACE=1 around a single execution; prints the waits grouped by the libqhip entry point on the stack.
    MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 python tools/sync_sources.py [repartition|broadcast] [sf]
"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("QHIP_SYNC_CHILD") != "1":
    env = dict(os.environ, QHIP_SYNC_CHILD="1", QHIP_SYNC_TRACE="1", QHIP_EXCHANGE_FORCE="1", QHIP_COMM_FORCE_RCCL="1", QHIP_COMM_SELF_RCCL="1")
    env.setdefault("MASTER_ADDR", "127.0.0.1"); env.setdefault("MASTER_PORT", "29551")
    r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    err = r.stderr
    start = err.rfind("=== MEASURED QUERY ===")
    end = err.rfind("=== END ===")
    chunk = err[start:end]
    waits = chunk.split("[qhip] host wait #")[1:]
    by = collections.Counter()
    for w in waits:
        frames = re.findall(r"libqhip\.so\((\w+)", w)
        names = [f for f in frames if f.startswith("qhip_")] or frames[-1:]
        inner = [re.sub(r"^_ZN?\d*", "", f)[:48] for f in frames if not f.startswith("qhip_")][:3]
        by[(" <- ".join(names[:1]) or "?") + "   via " + " < ".join(inner)] += 1
    print(r.stdout[-400:])
    print(f"{len(waits)} host waits in one repeated query:")
    for k, v in by.most_common():
        print(f"  {v:3d}  {k}")
    if not waits:
        print(err[-3000:])
    sys.exit(r.returncode)
sys.path.insert(0, ROOT)
import qurious_amd as q
from qurious_amd import exchange, queries, synth
strategy = sys.argv[1] if len(sys.argv) > 1 else "repartition"
sf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
ctx = q.get_context()
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
c, o, l = synth.q3_tables(sf)
tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
plan = queries.q3(*tabs, join_cls=exchange.DistributedHashJoinExec) if strategy == "repartition" else \
    queries.q3(*tabs, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate)
exchange.prune_exchange_columns(plan)
for _ in range(3):
    plan.execute_device()
ctx.synchronize()
print("=== MEASURED QUERY ===", file=sys.stderr, flush=True)
n0 = ctx.sync_count()
plan.execute_device()
print("=== END ===", file=sys.stderr, flush=True)
print(f"{strategy}: {ctx.sync_count() - n0} waits counted by libqhip")
dist.destroy_process_group()
