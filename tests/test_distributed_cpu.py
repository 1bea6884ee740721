"""The N > 1 path on CPUs: world_size-2 `gloo` processes run the hash-join exchange (partition by key -> variable-size
all-to-all -> local join -> union) with the oracle as the local engine, and the union must equal the single-process join.
Covers qurious_amd.exchange.all_to_all_bytes (the same grouped send/recv code RCCL runs on GPUs) and the partitioning rule."""
import io
import os
import socket

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType

from .helpers import col, rows_of, table_scan

I64 = pa.int64()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tables(seed=21, nl=600, nr=2000, nkeys=97):
    rng = np.random.default_rng(seed)
    ls = pa.schema([pa.field("lk", I64), pa.field("lv", I64)])
    rs = pa.schema([pa.field("rk", I64), pa.field("rv", I64), pa.field("rs", pa.string())])
    lb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, nkeys, nl), type=I64, mask=rng.random(nl) < 0.05),
                                     pa.array(rng.integers(0, 10**6, nl), type=I64)], schema=ls)
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, nkeys, nr), type=I64, mask=rng.random(nr) < 0.05),
                                     pa.array(rng.integers(0, 10**6, nr), type=I64),
                                     pa.array(["s%d" % v for v in rng.integers(0, 9, nr)])], schema=rs)
    return (ls, lb), (rs, rb)


def _ipc(batch: pa.RecordBatch) -> bytes:
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    return sink.getvalue()


def _unipc(buf: bytes) -> pa.RecordBatch:
    t = pa.ipc.open_stream(buf).read_all().combine_chunks()
    return t.to_batches()[0] if t.num_rows else pa.RecordBatch.from_arrays([pa.array([], type=f.type) for f in t.schema], schema=t.schema)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd.exchange import all_to_all_bytes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (ls, lb), (rs, rb) = _tables()
        sides = []
        for schema, batch, key in ((ls, lb, "lk"), (rs, rb, "rk")):
            n = batch.num_rows
            mine = batch.slice(n * rank // world, n * (rank + 1) // world - n * rank // world)       # this rank's contiguous slice
            pid = qoracle.partition_ids([mine.column(key)], world)
            send = [torch.frombuffer(bytearray(_ipc(mine.filter(pa.array(pid == r)))), dtype=torch.uint8) for r in range(world)]
            # metadata words ride with the sizes in the first round (the wire images' layout words on the GPU path)
            got = all_to_all_bytes(send, meta=[[rank, r, int(send[r].numel()) * 3] for r in range(world)])
            assert got.meta == [[r, rank, int(got[r].numel()) * 3] for r in range(world)]
            parts = [_unipc(bytes(t.numpy().tobytes())) for t in got]
            sides.append((schema, pa.concat_batches(parts)))
        (ls2, lpart), (rs2, rpart) = sides
        # co-location: every key of this rank's inputs hashes to this rank
        for b, key in ((lpart, "lk"), (rpart, "rk")):
            assert (qoracle.partition_ids([b.column(key)], world) == rank).all()
        plan = q.HashJoinExec.try_new(table_scan(ls2, [lpart]), table_scan(rs2, [rpart]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
        local = rows_of(qoracle.execute(plan))
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            full = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
            want = sorted(rows_of(qoracle.execute(full)), key=repr)
            got_rows = sorted([r for part in gathered for r in part], key=repr)
            assert got_rows == want and len(want) > 1000
            open(os.path.join(out_dir, "ok"), "w").write(str(len(want)))
    finally:
        dist.destroy_process_group()


def _agg_over(join_plan, aggs_of):
    """GROUP BY lk over a join's output: SUM / COUNT / MIN / MAX of rv"""
    schema = pa.schema([pa.field("lk", I64), pa.field("s", I64), pa.field("c", I64), pa.field("mn", I64), pa.field("mx", I64)])
    return q.HashAggregate(schema, join_plan, [col("lk", 0)], aggs_of(col("rv", 3))), schema


def _aggs(arg):
    return [q.SumAggregateExpr(arg, I64), q.CountAggregateExpr(arg), q.MinAggregateExpr(arg, I64), q.MaxAggregateExpr(arg, I64)]


def _worker_broadcast(rank, world, port, out_dir):
    """broadcast strategy: the build side is all-gathered, the probe slice stays local, partial groups are repartitioned
    by group key and merged (qurious_amd.exchange.BroadcastHashJoinExec / DistributedHashAggregate, with the oracle as the
    local engine)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd.exchange import all_to_all_bytes, merge_aggregate_exprs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (ls, lb), (rs, rb) = _tables(seed=33, nl=300, nr=4000, nkeys=150)
        cut = lambda b: b.slice(b.num_rows * rank // world, b.num_rows * (rank + 1) // world - b.num_rows * rank // world)   # noqa: E731
        lmine, rmine = cut(lb), cut(rb)
        # all-gather of the build side = an all-to-all that sends the same bytes to everybody
        payload = torch.frombuffer(bytearray(_ipc(lmine)), dtype=torch.uint8)
        build = pa.concat_batches([_unipc(bytes(t.numpy().tobytes())) for t in all_to_all_bytes([payload] * world)])
        assert build.num_rows == lb.num_rows
        join = q.HashJoinExec.try_new(table_scan(ls, [build]), table_scan(rs, [rmine]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
        partial_plan, pschema = _agg_over(join, _aggs)
        partial = pa.concat_batches(qoracle.execute(partial_plan))
        pid = qoracle.partition_ids([partial.column(0)], world)
        send = [torch.frombuffer(bytearray(_ipc(partial.filter(pa.array(pid == r)))), dtype=torch.uint8) for r in range(world)]
        mine = pa.concat_batches([_unipc(bytes(t.numpy().tobytes())) for t in all_to_all_bytes(send)])
        merge = q.HashAggregate(pschema, table_scan(pschema, [mine]), [col("lk", 0)], merge_aggregate_exprs(partial_plan.aggregate_exprs, 1))
        local = rows_of(qoracle.execute(merge))
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            full_join = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb]), JoinType.Inner, [(col("lk", 0), col("rk", 0))], None)
            want = sorted(rows_of(qoracle.execute(_agg_over(full_join, _aggs)[0])))
            got = sorted(r for part in gathered for r in part)
            keys = [r[0] for r in got]
            assert len(keys) == len(set(keys))            # after the merge every group lives on exactly one rank
            assert got == want and len(want) > 50
            open(os.path.join(out_dir, "ok_broadcast"), "w").write(str(len(want)))
    finally:
        dist.destroy_process_group()


def test_broadcast_join_and_partial_aggregate_merge_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_broadcast, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok_broadcast").read()) > 50


def test_merge_rules_of_partial_aggregates():
    from qurious_amd.exchange import merge_aggregate_exprs
    merged = merge_aggregate_exprs(_aggs(col("rv", 3)), 2)
    assert [type(m).__name__ for m in merged] == ["SumAggregateExpr", "SumAggregateExpr", "MinAggregateExpr", "MaxAggregateExpr"]
    assert [m.expression().index for m in merged] == [2, 3, 4, 5]
    with pytest.raises(q.UnsupportedError, match="AVG"):
        merge_aggregate_exprs([q.AvgAggregateExpr(col("rv", 3), I64, pa.float64())], 1)


def test_exchange_join_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok").read()) > 1000


def test_exchange_join_world3_gloo(tmp_path):
    """an odd number of ranks: uneven slices, three-way grouped send / receive rounds"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert int(open(tmp_path / "ok").read()) > 1000


def test_partition_rule_is_deterministic_and_balanced():
    from oracle import qoracle
    keys = pa.array(np.arange(100000), type=I64)
    pid = qoracle.partition_ids([keys], 8)
    counts = np.bincount(pid, minlength=8)
    assert counts.min() > 11000 and counts.max() < 14000
    assert (qoracle.partition_ids([keys], 8) == pid).all()
    nulls = pa.array([None, None, 5], type=I64)
    p = qoracle.partition_ids([nulls], 4)
    assert p[0] == p[1]


def test_exchange_column_pruning_analysis_on_q3():
    """prune_exchange_columns: every distributed join learns which output columns the plan above it reads; its exchanges
    carry those, its keys and nothing else (Q3: 4 of customer+orders' 7 columns, 3 of lineitem's 4)"""
    from qurious_amd import exchange, queries, synth
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, []), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, []),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, []))
    plan = queries.q3(*tabs, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate)
    assert exchange.prune_exchange_columns(plan) is plan
    j2 = plan.partial.input
    j1 = j2.left
    names2 = [f.name for f in j2.schema()]
    assert sorted(names2[c] for c in j2._needed) == ["l_discount", "l_extendedprice", "l_orderkey", "o_orderdate", "o_shippriority"]
    l2, r2 = j2._needed_per_side()
    assert sorted(j1.schema().field(c).name for c in l2) == ["o_orderdate", "o_orderkey", "o_shippriority"]
    assert sorted(synth.LINEITEM_Q3_SCHEMA.field(c).name for c in r2) == ["l_discount", "l_extendedprice", "l_orderkey"]
    l1, r1 = j1._needed_per_side()
    assert [synth.CUSTOMER_SCHEMA.field(c).name for c in l1] == ["c_custkey"]           # c_mktsegment is only the scan filter's
    assert sorted(synth.ORDERS_SCHEMA.field(c).name for c in r1) == ["o_custkey", "o_orderdate", "o_orderkey", "o_shippriority"]
    wire = exchange._wire_schema(synth.CUSTOMER_SCHEMA, l1)
    assert [str(f.type) for f in wire] == ["int64", "null"]
    # an un-annotated plan keeps everything
    plain = queries.q3(*tabs, join_cls=exchange.DistributedHashJoinExec)
    assert plain.input._needed_per_side() == (None, None)


# ---------------------------------------------------------------- heavy hitters of the repartitioned join
def _skew_tables(s, n_probe=24000, domain=300, seed=5):
    from qurious_amd import synth
    rng = np.random.default_rng(seed)
    ls = pa.schema([pa.field("bk", I64), pa.field("bv", I64)])
    rs = pa.schema([pa.field("pk", I64), pa.field("pv", I64)])
    lb = pa.RecordBatch.from_arrays([pa.array(np.arange(1, domain + 1), type=I64), pa.array(rng.integers(0, 10**6, domain), type=I64)], schema=ls)
    pk = synth.zipf_ranks(0, n_probe, domain, s, 5)
    rb = pa.RecordBatch.from_arrays([pa.array(pk, type=I64, mask=rng.random(n_probe) < 0.01), pa.array(np.arange(n_probe), type=I64)], schema=rs)
    return (ls, lb), (rs, rb)


def _worker_heavy(rank, world, port, out_dir, s):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import collections
    import torch
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd import exchange
    from qurious_amd.exchange import all_to_all_bytes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (ls, lb), (rs, rb) = _skew_tables(s)
        mine = lambda b: b.slice(b.num_rows * rank // world, b.num_rows * (rank + 1) // world - b.num_rows * rank // world)   # noqa: E731
        build, probe = mine(lb), mine(rb)

        def shuffle(batch, key):
            pid = qoracle.partition_ids([batch.column(key)], world)
            send = [torch.frombuffer(bytearray(_ipc(batch.filter(pa.array(pid == r)))), dtype=torch.uint8) for r in range(world)]
            return pa.concat_batches([_unipc(bytes(t.numpy().tobytes())) for t in all_to_all_bytes(send)])

        # (a) blind repartitioning: how many probe rows land here
        blind = shuffle(probe, "pk").num_rows
        # (b) the operator's flow (exchange.DistributedHashJoinExec) with the oracle as the local engine: count a strided
        # sample, agree on the heavy keys, keep their probe rows, broadcast their build rows, repartition the rest
        stride = 8
        sample = probe.column("pk").to_pylist()[::stride]
        top = collections.Counter(k for k in sample if k is not None).most_common(exchange.HEAVY_CANDIDATES)
        keys = exchange.heavy_keys(top, len(sample))
        assert keys, "Zipf keys must produce heavy hitters"
        hb, lb_pred = exchange.heavy_split_predicates(col("bk", 0), I64, keys)
        hp, lp_pred = exchange.heavy_split_predicates(col("pk", 0), I64, keys)
        filt = lambda schema, batch, pred: pa.concat_batches(qoracle.execute(q.Filter(table_scan(schema, [batch]), pred)))   # noqa: E731
        heavy_probe, light_probe = filt(rs, probe, hp), filt(rs, probe, lp_pred)
        assert heavy_probe.num_rows + light_probe.num_rows == probe.num_rows      # the predicates split the table exactly (NULL keys are light)
        heavy_build_local, light_build = filt(ls, build, hb), filt(ls, build, lb_pred)
        send = [torch.frombuffer(bytearray(_ipc(heavy_build_local)), dtype=torch.uint8)] * world                               # all-gather
        heavy_build = pa.concat_batches([_unipc(bytes(t.numpy().tobytes())) for t in all_to_all_bytes(send)])
        lpart = pa.concat_batches([shuffle(light_build, "bk"), heavy_build])
        rpart = pa.concat_batches([shuffle(light_probe, "pk"), heavy_probe])
        plan = q.HashJoinExec.try_new(table_scan(ls, [lpart]), table_scan(rs, [rpart]), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
        local = rows_of(qoracle.execute(plan))
        gathered = [None] * world
        dist.all_gather_object(gathered, (local, blind, rpart.num_rows, keys))
        if rank == 0:
            full = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb]), JoinType.Inner, [(col("bk", 0), col("pk", 0))], None)
            want = sorted(rows_of(qoracle.execute(full)), key=repr)
            got_rows = sorted([r for part, _, _, _ in gathered for r in part], key=repr)
            assert got_rows == want and len(want) > 20000
            assert all(g[3] == keys for g in gathered)                            # every rank agreed on the same heavy keys
            blind_rows, handled = [g[1] for g in gathered], [g[2] for g in gathered]
            mean = rb.num_rows / world
            open(os.path.join(out_dir, "ok"), "w").write(f"{max(blind_rows) / mean:.3f} {max(handled) / mean:.3f} {max(blind_rows) / rb.num_rows:.3f}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,s", [(2, 1.5), (3, 1.1)])
def test_heavy_hitters_keep_the_repartitioned_join_balanced(tmp_path, world, s):
    """Zipf-skewed probe keys: blind hash partitioning gives one rank far more than its share (world 2, Zipf 1.5: > 60 % of
    the probe rows); with the heavy keys' probe rows kept local and their build rows broadcast every rank stays within 1.3x
    of the mean, every rank derives the same heavy keys, and the union of the ranks' joins is the single-process join."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_heavy, args=(world, port, str(tmp_path), s), nprocs=world, join=True)
    blind, handled, blind_share = (float(x) for x in open(os.path.join(str(tmp_path), "ok")).read().split())
    assert handled <= 1.3 and handled < blind
    if world == 2:
        assert blind_share > 0.6


def test_heavy_split_predicates_partition_every_row():
    """is-heavy / is-not-heavy must split a table exactly, NULL keys on the not-heavy side (they take the ordinary path)"""
    from oracle import qoracle
    from qurious_amd import exchange
    schema = pa.schema([pa.field("k", I64), pa.field("v", I64)])
    batch = pa.RecordBatch.from_arrays([pa.array([1, 2, None, 3, 2, 2, None, 7], type=I64), pa.array(range(8), type=I64)], schema=schema)
    heavy, light = exchange.heavy_split_predicates(col("k", 0), I64, [2, 7])
    run = lambda pred: rows_of(qoracle.execute(q.Filter(table_scan(schema, [batch]), pred)))   # noqa: E731
    assert [r[1] for r in run(heavy)] == [1, 4, 5, 7] and [r[1] for r in run(light)] == [0, 2, 3, 6]
    assert exchange.heavy_split_predicates(col("k", 0), pa.float64(), [1.5]) == (None, None)   # unsupported key type: no split


# ---------------------------------------------------------------- the operators themselves, end to end, on CPUs
# exchange.set_local_engine is the documented hook: the multi-rank operators ask a LocalEngine for everything rank-local
# (libqhip by default). Here the engine is Arrow batches + the oracle, so DistributedHashJoinExec / BroadcastHashJoinExec /
# DistributedHashAggregate run THEIR OWN code — which side moves, which columns travel, heavy-key split, exchange cache,
# merge of the partial groups — over gloo, and the union over ranks must equal the single-process oracle result.
class _StubContext:
    """what plan._retrying / exchange_cache need of a context"""

    def allow_deferred_sizes(self, n):
        pass


def _stub_engine():
    import collections
    import copy
    import torch
    from oracle import qoracle
    from qurious_amd import exchange

    def one_batch(batches, schema):
        batches = [b for b in batches if b.num_rows] or list(batches)[:1]
        if not batches:
            return pa.RecordBatch.from_arrays([pa.array([], type=f.type) for f in schema], schema=schema)
        return pa.Table.from_batches(batches).combine_chunks().to_batches()[0] if batches[0].num_rows else batches[0]

    class Engine(exchange.LocalEngine):
        fast_exchange = False          # (no libqhip communicator on a CPU)
        log = collections.Counter()
        _ctx = _StubContext()

        def context(self):
            return self._ctx

        def execute(self, node):
            if isinstance(node, (exchange.DistributedHashJoinExec, exchange.BroadcastHashJoinExec, exchange.DistributedHashAggregate)):
                return node.execute_device()                       # the operator under test: its own rank logic
            if isinstance(node, q.Scan):
                return one_batch(qoracle.execute(node), node.schema())
            clone = copy.copy(node)                                # children first (they may hold exchange operators) ...
            for name in ("input", "left", "right"):
                child = getattr(node, name, None)
                if isinstance(child, q.PhysicalPlan):
                    got = self.execute(child)
                    setattr(clone, name, table_scan(got.schema, [got]))
            return one_batch(qoracle.execute(clone), node.schema())   # ... then this node alone, by the oracle

        def probe_side(self, join, fuse):
            node = join.right
            if fuse and isinstance(node, q.Scan) and node.filter is not None and node.projections is None:
                return self.base_table(node), node.filter
            return self.execute(node), None

        def base_table(self, scan):
            return one_batch(scan.datasource.data, scan.datasource.schema())

        def num_rows(self, table):
            return table.num_rows

        def keep_columns(self, table, mask):
            if mask is None:
                return table
            self.log["columns_dropped"] += sum(1 for m in mask if not m)
            cols = [c if m else pa.nulls(table.num_rows, type=c.type) for c, m in zip(table.columns, mask)]
            return pa.RecordBatch.from_arrays(cols, schema=table.schema)

        def partition(self, table, keys, n_parts, range_bounds=None):
            if not table.num_rows:
                pid = np.zeros(0, dtype=np.int64)
            elif range_bounds is not None:
                self.log["range_partitions"] += 1
                pid = qoracle.partition_ids_by_range(table.column(keys[0].index), range_bounds)
            else:
                pid = qoracle.partition_ids([table.column(k.index) for k in keys], n_parts)
            return [table.filter(pa.array(pid == p)) for p in range(n_parts)]

        def key_share_in_range(self, table, schema, key, lo_excl, hi_incl):
            v = table.column(key.index).drop_null().to_numpy(zero_copy_only=False)[::8]
            if not len(v):
                return 1.0
            m = np.ones(len(v), dtype=bool)
            if lo_excl is not None:
                m &= v > lo_excl
            if hi_incl is not None:
                m &= v <= hi_incl
            return float(m.mean())

        def key_range(self, table, col):
            v = table.column(col).drop_null().to_numpy(zero_copy_only=False)
            return (int(v.min()), int(v.max())) if len(v) else (0, -1)

        def pack(self, table):
            self.log["packed"] += 1
            return [table.num_rows], torch.frombuffer(bytearray(_ipc(table)), dtype=torch.uint8)

        def unpack(self, schema, metas, images):
            parts = [_unipc(bytes(t.numpy().tobytes())) for t in images]
            assert [p.num_rows for p in parts] == [m[0] for m in metas]
            assert len(schema) == len(parts[0].schema)
            return one_batch(parts, parts[0].schema)

        def concat(self, tables):
            return one_batch(list(tables), tables[0].schema)

        def filter(self, table, predicate):
            return one_batch(qoracle.execute(q.Filter(table_scan(table.schema, [table]), predicate)), table.schema)

        def top_keys(self, table, schema, key, dtype):
            sample = table.column(key.index).to_pylist()[::8]
            top = collections.Counter(k for k in sample if k is not None).most_common(exchange.HEAVY_CANDIDATES)
            return top, len(sample)

        def join(self, op, lt, rt, lpred=None, rpred=None):
            self.log["joins"] += 1
            plain = q.HashJoinExec(table_scan(lt.schema, [lt], lpred), table_scan(rt.schema, [rt], rpred), op.join_type, op.on, op.filter,
                                   op._schema, op.column_indices)
            return one_batch(qoracle.execute(plain), op.schema())

        def aggregate(self, schema, table, keys, aggs):
            return one_batch(qoracle.execute(q.HashAggregate(schema, table_scan(table.schema, [table]), keys, aggs)), schema)

    return Engine()


def _worker_operators(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd import exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    engine = _stub_engine()
    previous = exchange.set_local_engine(engine)
    try:
        assert type(previous) is exchange.LocalEngine
        cut = lambda b: b.slice(b.num_rows * rank // world, b.num_rows * (rank + 1) // world - b.num_rows * rank // world)   # noqa: E731
        on = [(col("lk", 0), col("rk", 0))]
        pred = q.BinaryExpr(col("rv", 1), q.Operator.Lt, q.Literal(q.ScalarValue.Int64(700000)))
        results = {}

        # (1) repartitioned join, uniform keys, every join type whose rows the exchange co-locates
        (ls, lb), (rs, rb) = _tables(seed=7, nl=500, nr=3000, nkeys=120)
        for jt in (JoinType.Inner, JoinType.Left, JoinType.Right, JoinType.Full):
            plan = exchange.DistributedHashJoinExec.try_new(table_scan(ls, [cut(lb)]), table_scan(rs, [cut(rb)], pred), jt, on)
            results["repartition-" + jt.name] = rows_of([engine.execute(plan)])
        # (2) Zipf-skewed probe keys: the heavy keys' probe rows stay, their build rows are broadcast
        (hs, hb), (ps, pb) = _skew_tables(1.5, n_probe=12000, domain=200)
        skew = exchange.DistributedHashJoinExec.try_new(table_scan(hs, [cut(hb)]), table_scan(ps, [cut(pb)]), JoinType.Inner, [(col("bk", 0), col("pk", 0))])
        exchange.exchange_stats(reset=True)
        results["skew"] = rows_of([engine.execute(skew)])
        st = exchange.exchange_stats(reset=True)
        assert st["heavy_keys"] >= 1 and st["heavy_key_rounds"] == 1
        assert st["probe_rows_received"] <= 1.3 * pb.num_rows / world     # balance: what heavy-hitter handling is for
        results["skew-again"] = rows_of([engine.execute(skew)])          # the cached heavy-key set: no second sampling round
        assert exchange.exchange_stats(reset=True)["heavy_key_rounds"] == 0
        # (3) broadcast join under a distributed aggregate, with column pruning: the build side's value column is not read
        join = exchange.BroadcastHashJoinExec.try_new(table_scan(ls, [cut(lb)]), table_scan(rs, [cut(rb)], pred), JoinType.Inner, on)
        aschema = _agg_over(join, _aggs)[1]
        agg = exchange.DistributedHashAggregate(aschema, join, [col("lk", 0)], _aggs(col("rv", 3)))
        exchange.prune_exchange_columns(agg)
        before = engine.log["columns_dropped"]
        results["broadcast-agg"] = rows_of([engine.execute(agg)])
        assert engine.log["columns_dropped"] - before >= 1                  # lv never travelled
        # ... the same plan with the repartitioned join
        join2 = exchange.DistributedHashJoinExec.try_new(table_scan(ls, [cut(lb)]), table_scan(rs, [cut(rb)], pred), JoinType.Inner, on)
        agg2 = exchange.DistributedHashAggregate(aschema, join2, [col("lk", 0)], _aggs(col("rv", 3)))
        results["repartition-agg"] = rows_of([engine.execute(agg2)])

        # every local join went through the engine, every exchange through pack / all_to_all_bytes / unpack
        assert engine.log["joins"] == 8 and engine.log["packed"] >= 2 * 6 * world + 2 * world
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        if rank == 0:
            union = lambda name: sorted((r for g in gathered for r in g[name]), key=repr)   # noqa: E731
            for jt in (JoinType.Inner, JoinType.Left, JoinType.Right, JoinType.Full):
                full = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb], pred), jt, on)
                want = sorted(rows_of(qoracle.execute(full)), key=repr)
                assert union("repartition-" + jt.name) == want and len(want) > 1000, jt
            want = sorted(rows_of(qoracle.execute(q.HashJoinExec.try_new(table_scan(hs, [hb]), table_scan(ps, [pb]), JoinType.Inner,
                                                                         [(col("bk", 0), col("pk", 0))]))), key=repr)
            assert union("skew") == want and union("skew-again") == want and len(want) > 10000
            full_join = q.HashJoinExec.try_new(table_scan(ls, [lb]), table_scan(rs, [rb], pred), JoinType.Inner, on)
            want = sorted(rows_of(qoracle.execute(_agg_over(full_join, _aggs)[0])), key=repr)
            for name in ("broadcast-agg", "repartition-agg"):
                got = union(name)
                assert len({r[0] for r in got}) == len(got)         # every group on exactly one rank after the merge
                assert got == want and len(want) > 50, name
            open(os.path.join(out_dir, "ok_operators"), "w").write(str(len(want)))
    finally:
        exchange.set_local_engine(None)
        dist.destroy_process_group()


def test_exchange_operators_end_to_end_with_a_stub_local_engine_world2_gloo(tmp_path):
    """DistributedHashJoinExec (all four join types, heavy hitters, cached heavy keys), BroadcastHashJoinExec and
    DistributedHashAggregate (with exchange column pruning) executed BY THEIR OWN CODE on two gloo ranks through
    exchange.set_local_engine; the union of the ranks' results equals the single-process oracle result."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_operators, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert int(open(tmp_path / "ok_operators").read()) > 50


def _worker_range(rank, world, port, out_dir):
    """QHIP_EXCHANGE_RANGE=1: tables sliced in KEY order (TPC-H's orders / lineitem) are routed by key range — almost nothing
    moves; tables whose slices overlap in key space fall back to hash routing; either way the union equals the single-process join"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), QHIP_EXCHANGE_RANGE="1")
    import torch.distributed as dist
    from oracle import qoracle
    from qurious_amd import exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    engine = _stub_engine()
    exchange.set_local_engine(engine)
    try:
        rng = np.random.default_rng(3)
        n_orders, n_items = 4000, 15000
        os_ = pa.schema([pa.field("o_key", I64), pa.field("o_v", I64)])
        ls_ = pa.schema([pa.field("l_key", I64), pa.field("l_v", I64)])
        ob = pa.RecordBatch.from_arrays([pa.array(np.arange(n_orders) * 4 + 1, type=I64), pa.array(rng.integers(0, 100, n_orders), type=I64)], schema=os_)
        lk = np.sort(rng.integers(0, n_orders, n_items)) * 4 + 1
        lb = pa.RecordBatch.from_arrays([pa.array(lk, type=I64, mask=rng.random(n_items) < 0.01), pa.array(np.arange(n_items), type=I64)], schema=ls_)
        cut = lambda b: b.slice(b.num_rows * rank // world, b.num_rows * (rank + 1) // world - b.num_rows * rank // world)   # noqa: E731
        on = [(col("o_key", 0), col("l_key", 0))]
        results = {}
        # (1) both tables in key order, sliced by row ranges: the ranks' order-key ranges are disjoint -> range routing
        exchange.exchange_stats(reset=True)
        for jt in (JoinType.Inner, JoinType.Left, JoinType.Full):
            plan = exchange.DistributedHashJoinExec.try_new(table_scan(os_, [cut(ob)]), table_scan(ls_, [cut(lb)]), jt, on)
            results["sorted-" + jt.name] = rows_of([engine.execute(plan)])
        st = exchange.exchange_stats(reset=True)
        assert engine.log["range_partitions"] == 6 and st["range_rounds"] == 3
        moved_sorted = st["bytes_sent"]
        # (2) the same rows shuffled before slicing: every rank's key range covers everything -> hash routing, as before
        perm_o, perm_l = rng.permutation(n_orders), rng.permutation(n_items)
        ob2, lb2 = ob.take(pa.array(perm_o)), lb.take(pa.array(perm_l))
        plan = exchange.DistributedHashJoinExec.try_new(table_scan(os_, [cut(ob2)]), table_scan(ls_, [cut(lb2)]), JoinType.Inner, on)
        results["shuffled"] = rows_of([engine.execute(plan)])
        st = exchange.exchange_stats(reset=True)
        assert engine.log["range_partitions"] == 6 and st["range_rounds"] == 1       # asked, ranges overlap, hash
        moved_hashed = st["bytes_sent"]
        # (3) orders in key order but the probe keys drawn at random over ALL orders: the ranges are aligned, the rows are not
        lb3 = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, n_orders, n_items) * 4 + 1, type=I64), pa.array(np.arange(n_items), type=I64)], schema=ls_)
        plan = exchange.DistributedHashJoinExec.try_new(table_scan(os_, [cut(ob)]), table_scan(ls_, [cut(lb3)]), JoinType.Inner, on)
        results["random-probe"] = rows_of([engine.execute(plan)])
        assert engine.log["range_partitions"] == 6                                     # hash again
        # (4) RangeBroadcastHashJoinExec: the probe side never moves; a rank receives the build rows inside ITS probe key range —
        # key-ordered slices: the orders at the slice border only; shuffled slices: the ranges cover each other -> plain broadcast
        exchange.exchange_stats(reset=True)
        for jt in (JoinType.Inner, JoinType.Right):
            plan = exchange.RangeBroadcastHashJoinExec.try_new(table_scan(os_, [cut(ob)]), table_scan(ls_, [cut(lb)]), jt, on)
            results["rb-" + jt.name] = rows_of([engine.execute(plan)])
        st = exchange.exchange_stats(reset=True)
        assert st["range_rounds"] == 2 and st["build_rows_received"] <= 100, st      # (the orders around the slice border: the two tables are cut by row counts)
        plan = exchange.RangeBroadcastHashJoinExec.try_new(table_scan(os_, [cut(ob2)]), table_scan(ls_, [cut(lb2)]), JoinType.Inner, on)
        results["rb-shuffled"] = rows_of([engine.execute(plan)])
        st = exchange.exchange_stats(reset=True)
        assert st["build_rows_received"] == 0 and st["bytes_sent"] > 10_000          # the whole build side was all-gathered instead
        gathered = [None] * world
        dist.all_gather_object(gathered, (results, moved_sorted, moved_hashed))
        if rank == 0:
            union = lambda name: sorted((r for g in gathered for r in g[0][name]), key=repr)   # noqa: E731
            for jt in (JoinType.Inner, JoinType.Right):
                full = q.HashJoinExec.try_new(table_scan(os_, [ob]), table_scan(ls_, [lb]), jt, on)
                assert union("rb-" + jt.name) == sorted(rows_of(qoracle.execute(full)), key=repr), jt
            full = q.HashJoinExec.try_new(table_scan(os_, [ob]), table_scan(ls_, [lb]), JoinType.Inner, on)
            assert union("rb-shuffled") == sorted(rows_of(qoracle.execute(full)), key=repr)
            for jt in (JoinType.Inner, JoinType.Left, JoinType.Full):
                full = q.HashJoinExec.try_new(table_scan(os_, [ob]), table_scan(ls_, [lb]), jt, on)
                assert union("sorted-" + jt.name) == sorted(rows_of(qoracle.execute(full)), key=repr), jt
            full = q.HashJoinExec.try_new(table_scan(os_, [ob]), table_scan(ls_, [lb]), JoinType.Inner, on)
            assert union("shuffled") == sorted(rows_of(qoracle.execute(full)), key=repr)
            full3 = q.HashJoinExec.try_new(table_scan(os_, [ob]), table_scan(ls_, [lb3]), JoinType.Inner, on)   # (same seed on every rank: lb3 is the whole table)
            assert union("random-probe") == sorted(rows_of(qoracle.execute(full3)), key=repr)
            # three range-routed joins moved less than a tenth of what ONE hash-routed join moves (only rows at the slice borders)
            assert sum(g[1] for g in gathered) * 10 < sum(g[2] for g in gathered) * 3, [(g[1], g[2]) for g in gathered]
            open(os.path.join(out_dir, "ok_range"), "w").write("ok")
    finally:
        exchange.set_local_engine(None)
        dist.destroy_process_group()


def test_exchange_by_key_range_world2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_range, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok_range").read() == "ok"
