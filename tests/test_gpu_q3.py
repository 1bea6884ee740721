"""GPU parity: TPC-H Q3 pipeline (two hash joins + aggregate, everything device-resident between operators) and the
single-GPU pieces of the multi-GPU exchange (partition by key, raw buffers, concat) vs the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pyarrow as pa
import pytest

import qurious_amd as q
from qurious_amd import JoinType, exchange, queries, synth

from .helpers import col, rows_of, table_scan

pytestmark = pytest.mark.gpu
I64 = pa.int64()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _q3_tables(sf):
    c, o, l = synth.q3_tables(sf, orders_per_batch=4096)
    return (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))


def test_q3_sf001_matches_oracle(ctx, oracle, join_layout):
    plan = queries.q3(*_q3_tables(0.01))
    got = sorted(rows_of(plan.execute()))
    want = sorted(rows_of(oracle.execute(plan)))
    assert got == want and len(got) > 50
    assert plan.execute()[0].schema.field(3).type == pa.decimal128(38, 4)
    # the reference's own join order inside one process is preserved: compare the second join's output ordered
    j2 = plan.input
    got_j = j2.execute()
    want_j = oracle.execute(j2)
    assert [b.num_rows for b in got_j] == [b.num_rows for b in want_j]
    assert rows_of(got_j) == rows_of(want_j)


def test_q3_order_by_revenue_limit_10(ctx, oracle, golden):
    """Q3 to its last operator: top-10 by revenue DESC, o_orderdate (q3.slt:20-24), bit-exact and in the same order as the
    oracle; the reference's SF0.01 golden pins the output type's scale (4 digits) and the ordering property."""
    plan = queries.q3_top10(*_q3_tables(0.01))
    got = plan.execute()
    want = oracle.execute(plan)
    assert [b.num_rows for b in got] == [b.num_rows for b in want] == [10]
    assert rows_of(got) == rows_of(want)
    rev = [r[3] for r in rows_of(got)]
    assert rev == sorted(rev, reverse=True)
    ref = golden["tpch_q3_sf001"]["rows"]
    assert len(ref) == 10 and [r[1] for r in ref] == sorted((r[1] for r in ref), key=float, reverse=True)
    assert all(len(r[1].split(".")[1]) == 4 for r in ref) and got[0].schema.field(3).type.scale == 4


def test_q3_skewed_keys_match_oracle_and_report_table_occupancy(ctx, oracle, monkeypatch):
    """configs[4] shape: o_custkey / l_orderkey re-drawn from Zipf(1.1) — hot probe keys in both joins, hot groups in
    the aggregate (the wave-resident hot-key cache and the LDS table both see them). Bit-exact vs the oracle; the
    instrumented run reports the LDS hash-table occupancy BASELINE.json asks for."""
    c, o, l = synth.q3_tables_skewed(0.02, 1.1, orders_per_batch=4096)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    plan = queries.q3(*tabs)
    got = sorted(rows_of(plan.execute()))
    want = sorted(rows_of(oracle.execute(plan)))
    assert got == want and len(got) > 50
    top = queries.q3_top10(*tabs)
    assert rows_of(top.execute()) == rows_of(oracle.execute(top))
    monkeypatch.setenv("QHIP_AGG_STATS", "1")
    assert sorted(rows_of(plan.execute())) == want
    st = ctx.last_stats()
    assert st["groups"] == len(want) and st["lds_table_slots"] > 0 and 0.0 < st["lds_occupancy"] <= 1.0 and 0.0 < st["hbm_table_load"] <= 1.0
    monkeypatch.delenv("QHIP_AGG_STATS")
    plan.execute()
    assert ctx.last_stats()["lds_occupancy"] == -1.0


def test_broadcast_strategy_pieces_on_one_gpu(ctx, oracle):
    """What the broadcast strategy does on every rank, checked on one GPU: (1) with a single process the distributed nodes are
    the plain operators; (2) partial aggregates of two input halves, concatenated and merged with merge_aggregate_exprs on
    the device, equal the aggregate of the whole input — for SUM / COUNT / MIN / MAX over Q3's join output."""
    tabs = _q3_tables(0.01)
    plain = queries.q3(*tabs)
    dist_plan = queries.q3(*tabs, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate)
    assert sorted(rows_of(dist_plan.execute())) == sorted(rows_of(plain.execute()))
    j2 = plain.input
    joined = j2.execute_device()
    n = joined.num_rows
    assert n > 100
    dec = pa.decimal128(38, 4)
    schema = pa.schema([pa.field("l_orderkey", I64), pa.field("rev", dec), pa.field("cnt", I64), pa.field("lo", pa.decimal128(15, 2)), pa.field("hi", pa.decimal128(15, 2))])
    aggs = [plain.aggregate_exprs[0], q.CountAggregateExpr(col("l_discount", 9)), q.MinAggregateExpr(col("l_extendedprice", 8), pa.decimal128(15, 2)),
            q.MaxAggregateExpr(col("l_extendedprice", 8), pa.decimal128(15, 2))]
    keys = [col("l_orderkey", 6)]
    src = exchange.DeviceSource(j2.schema(), joined)
    whole = q.HashAggregate(schema, src, keys, aggs)
    # two disjoint parts of the join output that both hold rows of the same groups: by the parity of l_shipdate's day number
    # (a Limit with OFFSET over this many-batch input would not do: the reference's Limit drops rows of every batch behind the
    # offset, limit.rs:39-44)
    from qurious_amd import Operator
    parity = q.BinaryExpr(q.CastExpr(col("l_shipdate", 7), pa.int32()), Operator.Mod, q.Literal(q.ScalarValue.Int32(2)))
    part = lambda k: q.Filter(src, q.BinaryExpr(parity, Operator.Eq, q.Literal(q.ScalarValue.Int32(k))))   # noqa: E731
    halves = [q.HashAggregate(schema, part(0), keys, aggs).execute_device(), q.HashAggregate(schema, part(1), keys, aggs).execute_device()]
    assert halves[0].num_rows > 10 and halves[1].num_rows > 10
    both = exchange.concat_tables(halves)
    merged = q.HashAggregate(schema, exchange.DeviceSource(schema, both), [col("l_orderkey", 0)], exchange.merge_aggregate_exprs(aggs, 1))
    assert sorted(rows_of(merged.execute())) == sorted(rows_of(whole.execute())) == sorted(rows_of(oracle.execute(q.HashAggregate(schema, j2, keys, aggs))))


def test_q1_order_by_flags(ctx, oracle):
    li = synth.lineitem(200_000, batch_rows=65_536)
    plan = queries.q1_full_ordered(q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, li))
    got, want = plan.execute(), oracle.execute(plan)
    assert rows_of(got) == rows_of(want) and [r[:2] for r in rows_of(got)] == [("A", "F"), ("N", "F"), ("N", "O"), ("R", "F")]


def test_partition_by_key_matches_mirror_and_roundtrips(ctx, oracle):
    rng = np.random.default_rng(31)
    n = 50_000
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("d", pa.decimal128(15, 2)), pa.field("b", pa.bool_())])
    import decimal
    batch = pa.RecordBatch.from_arrays([
        pa.array(rng.integers(0, 5000, n), type=I64, mask=rng.random(n) < 0.03),
        pa.array(["v%d" % v for v in rng.integers(0, 40, n)], type=pa.string(), mask=rng.random(n) < 0.03),
        pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**8, 10**8, n)], type=pa.decimal128(15, 2)),
        pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=rng.random(n) < 0.03)], schema=schema)
    scan = table_scan(schema, [batch.slice(0, 20_000), batch.slice(20_000, 30_000)])
    dev = scan.execute_device()
    parts = exchange.partition_by_key(dev, [col("k", 0)], 4)
    pid = oracle.partition_ids([batch.column("k")], 4)
    for p, part in enumerate(parts):
        got = rows_of(part.to_batches())
        want = rows_of([batch.filter(pa.array(pid == p))])
        assert got == want            # rows keep their relative order inside a part
    # Utf8 partition key (packed into 4 words on both sides of the exchange)
    sparts = exchange.partition_by_key(dev, [col("s", 1)], 3)
    spid = oracle.partition_ids([batch.column("s")], 3)
    for p, part in enumerate(sparts):
        assert rows_of(part.to_batches()) == rows_of([batch.filter(pa.array(spid == p))])
    back = exchange.concat_tables(parts)
    assert sorted(rows_of(back.to_batches()), key=repr) == sorted(rows_of([batch]), key=repr)
    # raw buffers -> qhip_table_from_device -> same rows (what the all-to-all receiver does)
    import torch
    part = parts[1]
    bufs, meta = [], []
    for c in range(len(schema)):
        tens = []
        for ptr, nb in exchange._column_buffers(part, c):
            tens.append(torch.as_tensor(exchange._DevMem(ptr, nb), device="cuda").clone() if nb else torch.empty(0, dtype=torch.uint8, device="cuda"))
        bufs.append(tens)
        meta.append((part.num_rows if tens[1].numel() else 0, tens[2].numel()))
    rebuilt = exchange._table_from_buffers(ctx, schema, part.num_rows, meta, bufs)
    assert rows_of(rebuilt.to_batches()) == rows_of(part.to_batches())


def test_wire_images_rebuild_the_concatenation(ctx, oracle):
    """qhip_table_pack -> qhip_table_unpack_concat (what one exchange does with the parts it sends and receives): NULLs at
    bit positions that are not byte-aligned, strings, booleans, decimals, empty parts and a part without any NULL"""
    import decimal
    rng = np.random.default_rng(77)
    n = 10_007
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("d", pa.decimal128(15, 2)), pa.field("b", pa.bool_()),
                        pa.field("i", pa.int32()), pa.field("z", pa.null())])
    batch = pa.RecordBatch.from_arrays([
        pa.array(rng.integers(0, 5, n), type=I64),
        pa.array(["v%d" % v if v % 7 else "" for v in rng.integers(0, 400, n)], type=pa.string(), mask=rng.random(n) < 0.2),
        pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**8, 10**8, n)], type=pa.decimal128(15, 2), mask=rng.random(n) < 0.01),
        pa.array(rng.random(n) < 0.5, type=pa.bool_(), mask=rng.random(n) < 0.3),
        pa.array(rng.integers(-100, 100, n), type=pa.int32()),
        pa.nulls(n)], schema=schema)
    dev = table_scan(schema, [batch]).execute_device()
    parts = exchange.partition_by_key(dev, [col("k", 0)], 16)        # 5 distinct keys -> most of the 16 parts are empty
    assert sum(1 for p in parts if p.num_rows == 0) >= 11
    packed = [exchange.pack_table(p) for p in parts]
    for order in (list(range(16)), list(reversed(range(16)))):
        back = exchange.unpack_concat(ctx, schema, [packed[k][0] for k in order], [packed[k][1] for k in order])
        want = [r for k in order for r in rows_of(parts[k].to_batches())]
        assert rows_of(back.to_batches()) == want and back.num_rows == n
    # the device-side concat of tables is the same assembly
    assert rows_of(exchange.concat_tables(parts).to_batches()) == [r for p in parts for r in rows_of(p.to_batches())]
    # an image is rejected when its metadata does not describe it
    bad = [list(packed[0][0])] + [m for m, _ in packed[1:]]
    bad[0][1] += 16
    with pytest.raises(q.InternalError, match="metadata"):
        exchange.unpack_concat(ctx, schema, bad, [img for _, img in packed])


def test_multi_rank_code_path_rehearsal_on_one_gpu():
    """tools/exchange_rehearsal.py: a one-rank RCCL group with the exchange steps forced on — Q3 through both multi-GPU
    strategies equals the plain plan and the oracle; then bench.py's N > 1 branch (process group, barrier, collectives)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_rehearsal.py"), "--sf", "0.1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REHEARSAL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    # two PROCESSES sharing the GPU, each with its slice of the tables, exchanging over gloo (device buffers staged through
    # the host): real cross-rank partition / transport / unpack / join / merge; the union equals the single-process plan
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_rehearsal.py"), "--sf", "0.1", "--world", "2"],
                       env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REHEARSAL OK" in r.stdout and r.stdout.count("equal to the single-process plan") == 3, r.stdout[-2000:] + r.stderr[-4000:]
    # ... and with Zipf(1.1) join keys: the repartitioned joins detect the heavy keys on a sample, keep their probe rows local and
    # broadcast their build rows (exchange.DistributedHashJoinExec); the ranks' probe sides stay within 1.3x of the mean
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_rehearsal.py"), "--sf", "0.1", "--world", "2", "--skew", "1.1"],
                       env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REHEARSAL OK" in r.stdout and "heavy keys per rank" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    env.update(QHIP_BENCH_FORCE_DIST="1", QHIP_EXCHANGE_FORCE="1", MASTER_PORT="29549", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               QHIP_BENCH_ZIPF_SF="0.2")   # (the configs[4] record of the N > 1 line at a size that fits a test)
    # (1) exactly the command the driver's scaling run issues per rank (default workload: the metric's step = Q1 at SF10 +
    # Q3 at SF10), with the multi-rank branch forced on: Q1 partial groups merged through an all-gather, Q3 through both
    # exchange strategies; (2) single-configuration modes at small sizes
    for extra in ([], ["--workload", "q1_full", "--rows", "3000000"], ["--workload", "q3", "--sf", "0.2"],
                  ["--workload", "q3", "--sf", "0.2", "--strategy", "repartition"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"] + extra,
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-4000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["value"] > 0 and line["n_gpus"] == 1 and line["scaling"] == "strong"
        assert line["distributed"]["world_size"] == 1 and line["cpu_baseline"] is None   # (the CPU baseline is an N = 1 item)
        if not extra:
            rec = line["records"]
            assert set(rec) >= {"q1_sf10", "q3_sf10", "q3_sf10_repartition"} and "configs[2]" in line["config"]["workload"]
            assert rec["q1_sf10"]["rows"] == 59986052 and rec["q1_sf10"]["groups"] == 4
            assert rec["q3_sf10"]["groups"] == rec["q3_sf10_repartition"]["groups"] > 100000
            assert rec["q3_sf10"]["exchange"]["exchanges_per_query"] > 0 and line["exchange"]["bytes_sent_per_query"] >= 0
            # (`frac` is on the bytes the kernel reads from the resident layout — SURVEY §8d's rule for narrowed layouts)
            assert line["roofline"]["kernel"] in ("qk_filter_agg", "qk_filter_agg_cons") and 0 < line["roofline"]["frac"] < 1
            assert line["roofline"]["arrow_layout_bytes_per_launch"] >= line["roofline"]["bytes_read_per_launch"]
            # configs[4] in the N > 1 line: Zipf(1.1) keys, both strategies, LDS-table occupancy, exchange rate against xGMI, host waits
            z = rec["q3_sf100_zipf"]
            assert "configs[4]" in z["workload"] and z["groups"] > 100 and z["other_strategy"]["groups"] == z["groups"]
            assert 0 < z["aggregate_table"]["lds_occupancy"] <= 1 and z["aggregate_table"]["groups"] > 0
            assert z["exchange"]["xgmi_peak_GBps"] == 7 * 153.0 and z["exchange"]["transport"] == "rccl"
            assert 0 < z["exchange"]["host_waits_per_query"] < 40 and z["exchange"]["heavy_key_rounds_per_query"] <= 1
            # a repeated distributed Q3 waits for the device at most 5 times, whatever the strategy (round 3:
            # 16-17): one wait per exchange CALL (both sides of a repartitioned join in one), joins of deferred size below them
            assert rec["q3_sf10"]["exchange"]["transport"] == "rccl"
            assert 0 < rec["q3_sf10"]["host_waits_per_query"] <= 5 and 0 < rec["q3_sf10_repartition"]["host_waits_per_query"] <= 5, \
                (rec["q3_sf10"]["host_waits_per_query"], rec["q3_sf10_repartition"]["host_waits_per_query"])
        if "q3" in extra:
            assert line["exchange"]["exchanges_per_query"] > 0


def test_repartitioned_join_equals_plain_join(ctx, oracle):
    """join(concat(partition(L)), concat(partition(R))) has the same rows as join(L, R)"""
    rng = np.random.default_rng(32)
    ls = pa.schema([pa.field("lk", I64), pa.field("lv", I64)])
    rs = pa.schema([pa.field("rk", I64), pa.field("rv", I64)])
    lb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 300, 4000), type=I64), pa.array(rng.integers(0, 10**6, 4000), type=I64)], schema=ls)
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 300, 9000), type=I64), pa.array(rng.integers(0, 10**6, 9000), type=I64)], schema=rs)
    L, R = table_scan(ls, [lb]), table_scan(rs, [rb])
    on = [(col("lk", 0), col("rk", 0))]
    plain = q.HashJoinExec.try_new(L, R, JoinType.Inner, on, None)
    lt = exchange.concat_tables(exchange.partition_by_key(L.execute_device(), [col("lk", 0)], 8))
    rt = exchange.concat_tables(exchange.partition_by_key(R.execute_device(), [col("rk", 0)], 8))
    re = q.HashJoinExec.try_new(exchange.DeviceSource(ls, lt), exchange.DeviceSource(rs, rt), JoinType.Inner, on, None)
    assert sorted(rows_of(re.execute())) == sorted(rows_of(plain.execute())) == sorted(rows_of(oracle.execute(plain)))
    # world == 1: the distributed node degenerates to the local join
    d = exchange.DistributedHashJoinExec.try_new(L, R, JoinType.Inner, on, None)
    assert rows_of(d.execute()) == rows_of(plain.execute())


def _rebatch(batches, rows):
    out = []
    for b in batches:
        out.extend(b.slice(o, min(rows, b.num_rows - o)) for o in range(0, b.num_rows, rows))
    return out


def test_q3_over_1024_row_batches_like_the_reference_csv_loader(ctx, oracle):
    """TPC-H tables loaded by COPY arrive as 1024-row batches (datasource/file/csv.rs:63-66): thousands of
    small batches per table go up through the coalesced upload, the join emits one batch per non-empty probe batch
    (hash_join.rs:363-372) and the many-batch download slices one transfer — same rows, same batch structure as the oracle"""
    c, o, l = synth.q3_tables(0.25, orders_per_batch=50_000)
    c, o, l = _rebatch(c, 1024), _rebatch(o, 1024), _rebatch(l, 1024)
    assert len(l) > 1400 and all(b.num_rows <= 1024 for b in l)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    plan = queries.q3(*tabs)
    assert sorted(rows_of(plan.execute())) == sorted(rows_of(oracle.execute(plan)))
    j2 = plan.input
    got, want = j2.execute(), oracle.execute(j2)
    assert [b.num_rows for b in got] == [b.num_rows for b in want] and len(got) > 500
    assert rows_of(got) == rows_of(want)
    # a Filter keeps one (possibly empty) batch per input batch; Utf8 + NULLs through the coalesced upload
    rng = np.random.default_rng(12)
    n = 300_000
    schema = pa.schema([pa.field("k", I64), pa.field("s", pa.string()), pa.field("b", pa.bool_())])
    big = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1000, n), type=I64, mask=rng.random(n) < 0.1),
                                      pa.array(["w%d" % v for v in rng.integers(0, 5000, n)], mask=rng.random(n) < 0.1),
                                      pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1)], schema=schema)
    small = _rebatch([big], 1000)
    f = q.Filter(table_scan(schema, small), q.BinaryExpr(col("k", 0), q.Operator.Lt, q.Literal(q.ScalarValue.Int64(300))))
    gf, wf = f.execute(), oracle.execute(f)
    assert [b.num_rows for b in gf] == [b.num_rows for b in wf] and len(gf) == 300
    assert rows_of(gf) == rows_of(wf)
    assert rows_of(table_scan(schema, small).execute()) == rows_of([big])


def test_compiled_host_runs_the_metrics_step(ctx):
    """tools/bench_host — the metric's step through the C++ mirror of the reference's operator API (include/qhip_plan.hpp),
    no Python between the plan nodes and the C ABI — produces the groups the Python mirror produces on the same synthetic tables."""
    import json
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "bench_host"])   # (g++ over the C ABI: no GPU needed to build)
    exe = os.path.join(ROOT, "tools", "bench_host")
    r = subprocess.run([exe, "3", "3", "0.5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    from qurious_amd import queries, synth
    c, o, l = synth.q3_tables(0.5, 0, 1)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o), q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    want = queries.q3(*tabs).execute_device().num_rows
    assert line["q1_groups"] == 4 and line["q3_groups"] == want > 1000 and line["ms_per_step"] > 0
    assert line["q3_lineitem_rows"] == sum(b.num_rows for b in l)
    # ... and not only as many groups: EVERY row and column of both results, folded into an order-independent checksum the same
    # way on both sides (tools/bench_host.cpp result_checksum)
    li = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, synth.lineitem(line["q1_rows"], 1 << 20))
    for name, plan in (("q1", queries.q1_full(li)), ("q3", queries.q3(*tabs))):
        rows, checksum = _result_checksum(plan.execute())
        assert rows == line[f"{name}_result_rows"] and str(checksum) == line[f"{name}_result_checksum"], name


def _result_checksum(batches):
    """tools/bench_host.cpp result_checksum in numpy: sum over the rows of mix64-chained column values + the row count"""
    U = np.uint64

    def mix64(x):
        x = x ^ (x >> U(33)); x = x * U(0xff51afd7ed558ccd); x = x ^ (x >> U(33)); x = x * U(0xc4ceb9fe1a85ec53); return x ^ (x >> U(33))
    total, rows = U(0), 0
    with np.errstate(over="ignore"):
        for b in batches:
            n = b.num_rows
            rows += n
            h = np.zeros(n, dtype=U)
            for c in b.columns:
                valid = np.asarray(c.is_valid().to_numpy(zero_copy_only=False), dtype=bool)
                t = c.type
                if pa.types.is_string(t):
                    x = h.copy()
                    vals = [v.encode() if v is not None else b"" for v in c.to_pylist()]
                    for i, v in enumerate(vals):
                        xi = x[i]
                        for k in range(0, len(v), 8):
                            xi = mix64(xi ^ U(int.from_bytes(v[k:k + 8], "little")))
                        x[i] = mix64(xi ^ U(len(v)))
                else:
                    width = 16 if pa.types.is_decimal128(t) else t.bit_width // 8
                    raw = np.frombuffer(c.buffers()[1], dtype=np.uint8)[c.offset * width:(c.offset + n) * width].reshape(n, width)
                    w0 = np.zeros((n, 8), dtype=np.uint8)
                    w0[:, :min(width, 8)] = raw[:, :min(width, 8)]
                    x = mix64(h ^ w0.view(U).reshape(n))
                    if width == 16:
                        x = mix64(x ^ np.ascontiguousarray(raw[:, 8:]).view(U).reshape(n))
                x = np.where(valid, x, mix64(h ^ U(0x9E3779B97F4A7C15)))
                h = x
            total = total + mix64(h).sum(dtype=U)
    return rows, int(total)
