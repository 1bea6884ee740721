#!/usr/bin/env python3
"""Rehearsal of the multi-GPU operators on ONE GPU: a one-rank RCCL process group with QHIP_EXCHANGE_FORCE=1 keeps every
exchange step in place (partition by key -> wire images -> transport rounds -> unpack + concat -> join / merge), so the
code a multi-GPU launch runs is executed — and checked against the plain single-process plan and the CPU oracle — on the
one-GPU box the tests get. (What it cannot show is the transport between two different GPUs; the world_size-2 gloo tests
in tests/test_distributed_cpu.py cover the protocol of the rounds.)

With --world N (> 1) it instead starts N processes that share the GPU, each holding its slice of the tables, and lets
them exchange over gloo (device buffers staged through the host): everything but RCCL's own transport runs across ranks.

    python tools/exchange_rehearsal.py [--sf 0.2] [--world 2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import qurious_amd as q  # noqa: E402
from qurious_amd import exchange, queries, synth  # noqa: E402


# host waits (stream synchronisations inside libqhip, the transport's included) of one REPEATED Q3 through the multi-rank
# operators, measured on the one-rank rehearsal; asserted as ceilings
# (round 4: one wait per exchange CALL — both sides of a repartitioned join in one — and joins of deferred size below the
# exchanges: a repartitioned Q3 waits 3 times, a broadcast one 5 times; VERDICT r03 item 2 asked for <= 5)
WAIT_CEILING = {"repartition": 5, "broadcast": 5}


def rows_of(batches):
    out = []
    for b in batches:
        out.extend(zip(*[c.to_pylist() for c in b.columns]))
    return sorted(out, key=repr)


def two_rank_worker(rank, world, port, sf, out_dir, skew=0.0):
    """`world` PROCESSES sharing the one GPU, each with its own libqhip context and its slice of the tables, exchanging over
    gloo (device tensors staged through the host): partition -> wire images -> transport between different ranks ->
    unpack + concat -> join / merge. The union of the ranks' results must equal the single-process plan over all rows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ctx = q.get_context()
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, o, l = synth.q3_tables_skewed(sf, skew, rank, world) if skew > 0 else synth.q3_tables(sf, rank, world)
        mine = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
                q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
        results = {}
        # "range" (round 4, DESIGN §7 "routing by key range"): join 1 broadcasts the customer keys and leaves the orders where they
        # are; join 2 leaves orders AND lineitem where they are and sends a build row only to the ranks whose lineitem keys can
        # reach it (RangeBroadcastHashJoinExec) — tables sliced in key order exchange the orders at the slice borders; the partial
        # groups of an order whose lineitems straddle a border are merged (DistributedHashAggregate)
        for name, plan in (("repartition", queries.q3(*mine, join_cls=exchange.DistributedHashJoinExec)),
                           ("broadcast", queries.q3(*mine, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate)),
                           ("range", queries.q3(*mine, join_cls=exchange.BroadcastHashJoinExec, join2_cls=exchange.RangeBroadcastHashJoinExec,
                                                agg_cls=exchange.DistributedHashAggregate))):
            exchange.prune_exchange_columns(plan)
            exchange.exchange_stats()
            if name == "range":
                os.environ["QHIP_EXCHANGE_RANGE"] = "1"
            try:
                local = rows_of(plan.execute_device().to_batches())
            finally:
                os.environ.pop("QHIP_EXCHANGE_RANGE", None)
            st = exchange.exchange_stats()
            gathered = [None] * world
            dist.all_gather_object(gathered, (local, st["bytes_sent"], st["heavy_keys"], st["probe_rows_received"]))
            results[name] = gathered
        if rank == 0:
            cc, oo, ll = synth.q3_tables_skewed(sf, skew) if skew > 0 else synth.q3_tables(sf)
            whole = queries.q3(q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, cc), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, oo),
                               q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, ll))
            want = rows_of(whole.execute())
            for name, gathered in results.items():
                got = sorted((r for part, *_ in gathered for r in part), key=repr)
                keys = [r[0] for r in got]
                assert got == want and len(keys) == len(set(keys)) and len(want) > 100, f"{name}: union of the ranks differs from the single-process plan"
                assert all(g[1] > 0 for g in gathered)            # every rank really sent rows to the others
                print(f"[rehearsal] {world} processes, {name}: {len(got)} groups in all ({[len(g[0]) for g in gathered]} per rank) equal to the "
                      f"single-process plan; bytes sent per rank {[g[1] for g in gathered]}")
                if name == "range" and skew == 0:
                    # uniform TPC-H-shaped tables, sliced in key order: far less than the repartitioned plan moves
                    assert sum(g[1] for g in gathered) * 4 < sum(g[1] for g in results["repartition"]), "range routing moved as much as hash routing"
                if skew > 0 and name == "repartition":
                    # Zipf keys: the repartitioned joins found heavy hitters, kept their probe rows local, and the ranks'
                    # probe sides stay balanced
                    recv = [g[3] for g in gathered]
                    assert all(g[2] > 0 for g in gathered), "no heavy keys found on skewed data"
                    assert max(recv) <= 1.3 * (sum(recv) / len(recv)), recv
                    print(f"[rehearsal] skew {skew}: heavy keys per rank {[g[2] for g in gathered]}, probe rows joined per rank {recv}")
            open(os.path.join(out_dir, "ok_two_rank"), "w").write(str(len(want)))
        ctx.synchronize()
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=0.2)
    ap.add_argument("--world", type=int, default=1, help="> 1: that many processes share the GPU and exchange over gloo")
    ap.add_argument("--skew", type=float, default=0.0, help="with --world > 1: join keys re-drawn from Zipf(s) (heavy-hitter handling)")
    args = ap.parse_args()
    if args.world > 1:
        import socket
        import tempfile
        import torch.multiprocessing as mp
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(two_rank_worker, args=(args.world, port, args.sf, d, args.skew), nprocs=args.world, join=True)
            assert os.path.exists(os.path.join(d, "ok_two_rank"))
        print("REHEARSAL OK")
        return
    os.environ["QHIP_EXCHANGE_FORCE"] = "1"
    # libqhip's own transport (qhip_exchange_tables / qhip_all_gather_table) over a REAL one-rank RCCL communicator whose own
    # part travels through ncclSend / ncclRecv / ncclAllGather too: every RCCL entry point the library binds is executed
    os.environ["QHIP_COMM_FORCE_RCCL"] = "1"
    os.environ["QHIP_COMM_SELF_RCCL"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    ctx = q.get_context()
    import torch
    import torch.distributed as dist
    from oracle import qoracle
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)                      # RCCL itself, beside libqhip's HIP runtime in one process
        assert t.tolist() == [1.0] * 4
        c, o, l = synth.q3_tables(args.sf)
        tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
                q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
        plain = queries.q3(*tabs)
        want = rows_of(plain.execute())
        assert want == rows_of(qoracle.execute(plain)) and len(want) > 100
        wire, waits = {}, {}
        for prune in (False, True):
            for name, make in (("repartition", lambda: queries.q3(*tabs, join_cls=exchange.DistributedHashJoinExec)),
                               ("broadcast", lambda: queries.q3(*tabs, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate))):
                for transport in ("rccl", "torch"):     # libqhip's communicator / torch.distributed point-to-point: bit for bit the same
                    os.environ["QHIP_TRANSPORT"] = transport
                    plan = make()
                    if prune:
                        exchange.prune_exchange_columns(plan)
                    for execution in range(4):          # (from the second execution on the heavy-key set is remembered, from the third the joins' sizes)
                        exchange.exchange_stats()
                        before = ctx.sync_count()
                        t_dev = plan.execute_device()
                        lib_waits = ctx.sync_count() - before
                        got = rows_of(t_dev.to_batches())
                        st = exchange.exchange_stats()
                        assert st["exchanges"] >= 3, st     # the exchange steps really ran
                        assert got == want, f"{name} over {transport}: distributed plan differs from the single-process plan"
                    if transport == "rccl":
                        assert st["rccl_version"] > 0, st   # a real RCCL communicator moved the images
                    wire[(name, prune, transport)] = st["bytes_packed"]
                    waits[(name, prune, transport)] = (lib_waits, st["transport_waits"], st["heavy_key_rounds"])
                    print(f"[rehearsal] {name}{' (pruned columns)' if prune else ''} over {transport}: {len(got)} groups equal to the plain plan and the "
                          f"oracle; {st['exchanges']} exchanges, {st['bytes_packed'] / 1e6:.1f} MB of wire images; host waits of a repeated query: "
                          f"{lib_waits} in libqhip (incl. the transport's {st['transport_waits'] if transport == 'rccl' else 0}) + "
                          f"{st['transport_waits'] if transport == 'torch' else 0} in torch; heavy-key rounds {st['heavy_key_rounds']}")
                os.environ.pop("QHIP_TRANSPORT", None)
        for name in ("repartition", "broadcast"):
            # (the one-call exchange moves the columns' runs themselves: no 16-byte section padding of a wire image)
            assert wire[(name, True, "rccl")] < wire[(name, False, "rccl")] and wire[(name, True, "torch")] - 4096 <= wire[(name, True, "rccl")] <= wire[(name, True, "torch")]
            # the host-wait budget of a REPEATED query through libqhip's transport (ceilings asserted so that a regression shows):
            # no heavy-key sampling any more, ONE wait per exchange inside the transport
            lib_waits, transport_waits, heavy_rounds = waits[(name, True, "rccl")]
            assert heavy_rounds == 0 and transport_waits <= 5 and lib_waits <= WAIT_CEILING[name], (name, waits)
        # a join output with NULLs and strings through the exchange (Full join: both sides padded)
        import numpy as np
        import pyarrow as pa
        rng = np.random.default_rng(5)
        ls = pa.schema([pa.field("lk", pa.int64()), pa.field("ls", pa.string()), pa.field("lb", pa.bool_())])
        rs = pa.schema([pa.field("rk", pa.int64()), pa.field("rd", pa.decimal128(15, 2))])
        n = 30_000
        lb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 4000, n), type=pa.int64(), mask=rng.random(n) < 0.05),
                                         pa.array(["s%d" % v for v in rng.integers(0, 50, n)], mask=rng.random(n) < 0.1),
                                         pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1)], schema=ls)
        import decimal
        rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 4000, n), type=pa.int64(), mask=rng.random(n) < 0.05),
                                         pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**6, 10**6, n)], type=pa.decimal128(15, 2))], schema=rs)
        L = q.Scan(ls, q.MemoryTable.try_new(ls, [lb.slice(0, 10_000), lb.slice(10_000)]), None, None)
        R = q.Scan(rs, q.MemoryTable.try_new(rs, [rb]), None, None)
        on = [(q.Column("lk", 0), q.Column("rk", 0))]
        for jt in (q.JoinType.Inner, q.JoinType.Full, q.JoinType.LeftAnti):
            plain = q.HashJoinExec.try_new(L, R, jt, on, None)
            d = exchange.DistributedHashJoinExec.try_new(L, R, jt, on, None)
            assert rows_of(d.execute_device().to_batches()) == rows_of(plain.execute()) == rows_of(qoracle.execute(plain)), jt
        print("[rehearsal] Inner / Full / LeftAnti joins with NULL keys, strings and booleans through the exchange: equal")
    finally:
        dist.destroy_process_group()
    print("REHEARSAL OK")


if __name__ == "__main__":
    main()
