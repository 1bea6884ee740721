#!/usr/bin/env python3
"""bench.py — headline benchmark of the qurious-hip backend (contract: see the task prompt / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
BASELINE.json configs[1] — SELECT l_returnflag, SUM(l_quantity) FROM lineitem WHERE l_shipdate < '1998-09-01'
GROUP BY l_returnflag over 100M synthetic rows (2^20-row Arrow batches) — executed by the fused filter +
hash-aggregate HIP kernel through the C ABI. With N > 1 every rank owns its own 100M-row shard (weak scaling,
rows are independent; partial aggregates are merged once on the host for verification — no data-path collective).

Prints ONE JSON line on rank 0: value = rows/s of the whole job, plus
  roofline     — algorithmic bytes (25 B/row) / mean device time of the dominant kernel (HIP events on
                 the library's stream) vs the 8 TB/s HBM peak; `traffic` = PMC-measured HBM bytes per launch when a
                 profile summary for this round exists under profiles/ (else null)
                 `stream_read_GBps` = a plain streaming-read kernel timed in the same run (the achievable ceiling)
  cpu_baseline — the CPU oracle's faithful-cost restatement of the reference executor (oracle/qoracle.c, 1 thread)
                 timed on a bounded sample of the same workload on this box's host cores (single-GPU runs only).
  extra        — (N = 1, default workload) the other single-GPU configurations: Q1's aggregate list and Q3 at SF10.

Other workloads: --workload q1_full | q3 [--sf S] [--skew 1.1] ; q3 with N > 1 shards the tables over the ranks and joins
them with --strategy broadcast (all-gather the small build sides, default) or repartition (all-to-all both sides by key).
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import qurious_amd as q  # noqa: E402  (before torch: binds the system ROCm runtime first)
from qurious_amd import queries, synth  # noqa: E402

ALGO_BYTES_PER_ROW = {"q1_mini": 25, "q1_full": 78}   # Date32 4 + Utf8 (4+1) [x2 for Q1] + Decimal128 16 [x4]
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def gen_table(first_row: int, n_rows: int, batch_rows: int) -> q.MemoryTable:
    starts = list(range(0, n_rows, batch_rows))
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        batches = list(ex.map(lambda s: synth.lineitem_batch(first_row + s, min(batch_rows, n_rows - s)), starts))
    return q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, batches)


def result_key(batches):
    rows = []
    for b in batches:
        rows.extend(zip(*[c.to_pylist() for c in b.columns]))
    return sorted(rows)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


USE_DIST = False   # set by main(): a torch.distributed process group (RCCL) is up

Q3_BYTES = {"customer": 21, "orders": 28, "lineitem": 44}   # SURVEY §8(d) algorithmic bytes per row


def bench_q3(args, ctx, rank, world, barrier, dist, torch):
    """configs[3]: TPC-H Q3 (customer |><| orders |><| lineitem -> GROUP BY) at --sf, tables sliced over the ranks, both
    joins repartitioned by key with the RCCL exchange (qurious_amd/exchange.py) when world > 1."""
    from qurious_amd import exchange
    t0 = time.time()
    c, o, l = synth.q3_tables_skewed(args.sf, args.skew, rank, world) if args.skew > 0 else synth.q3_tables(args.sf, rank, world)
    tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
            q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
    rows = [sum(b.num_rows for b in t.data) for t in tabs]
    log(f"generated SF{args.sf} slice {rows} rows in {time.time() - t0:.1f}s")
    for t in tabs:
        t.device_table()
    if USE_DIST and args.strategy == "broadcast":
        # the small build sides are all-gathered, the big probe sides stay where they are, partial groups are merged
        plan = queries.q3(*tabs, join_cls=exchange.BroadcastHashJoinExec, agg_cls=exchange.DistributedHashAggregate)
    else:
        plan = queries.q3(*tabs, join_cls=exchange.DistributedHashJoinExec if USE_DIST else None)
    if USE_DIST:
        exchange.prune_exchange_columns(plan)   # the exchanges move only the columns the plan above them reads
    for _ in range(args.warmup):
        out = plan.execute_device()
        log(f"warmup step: {out.num_rows} groups")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = plan.execute_device()
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if USE_DIST:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor(rows + [out.num_rows], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        rows_all = tot.tolist()
    else:
        rows_all = rows + [out.num_rows]
    table_stats = None
    if args.skew > 0 or os.environ.get("QHIP_AGG_STATS"):
        # BASELINE configs[4]: LDS hash-table occupancy of the final aggregate (one extra, untimed, instrumented execution)
        os.environ["QHIP_AGG_STATS"] = "1"
        plan.execute_device()
        st = ctx.last_stats()
        os.environ.pop("QHIP_AGG_STATS", None)
        table_stats = {"lds_table_slots_per_workgroup": st["lds_table_slots"], "lds_occupancy": st["lds_occupancy"],
                       "lds_spilled": bool(st["lds_spilled"]), "hbm_table_slots": st["table_capacity"], "hbm_table_load": st["hbm_table_load"],
                       "groups": st["groups"], "workgroups": st["workgroups"]}
    xgmi = exchange.exchange_stats() if USE_DIST else None
    if rank != 0:
        dist.destroy_process_group()
        return
    algo_bytes = rows_all[0] * Q3_BYTES["customer"] + rows_all[1] * Q3_BYTES["orders"] + rows_all[2] * Q3_BYTES["lineitem"]
    ms = elapsed / args.steps * 1e3
    achieved = algo_bytes / (ms * 1e-3) / 1e9 / world
    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import qoracle
        sf_small = min(args.sf, 0.05)
        cs, os_, ls = synth.q3_tables(sf_small)
        small = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, cs), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, os_),
                 q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, ls))
        splan = queries.q3(*small)
        t0 = time.perf_counter()
        want = qoracle.execute(splan)
        cdt = time.perf_counter() - t0
        assert result_key(want) == result_key(splan.execute()), "HIP Q3 result differs from the CPU oracle"
        nl = sum(b.num_rows for b in ls)
        cpu_baseline = {"value": nl / cdt, "unit": "rows/s", "cores": 1, "kind": "port", "seconds": cdt,
                        "sample": f"Q3 at SF{sf_small} ({nl} lineitem rows) through oracle/qoracle.py + qoracle.c, 1 of {os.cpu_count()} host cores"}
    line = {
        "metric": "rows/s on TPC-H Q1 scan+agg and Q3 hash-join, SF10, 1/2/4/8 MI355X",
        "value": rows_all[2] * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i128", "data": "synthetic",
        "config": {"workload": f"configs[3] q3: TPC-H Q3 SF{args.sf} customer|><|orders|><|lineitem + GROUP BY, HBM-resident, "
                               f"lineitem rows/s", "rows": {"customer": rows_all[0], "orders": rows_all[1], "lineitem": rows_all[2]},
                   "groups": rows_all[3], "parallelism": (f"broadcast build sides, local probes, merged partial groups x{world}" if USE_DIST and args.strategy == "broadcast"
                                   else f"hash-partitioned joins x{world}")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "q3 pipeline (all kernels of one query, per GPU)", "kernel_ms": ms,
                     "algorithmic_bytes": algo_bytes},
        "cpu_baseline": cpu_baseline, "device": ctx.device_name(),
    }
    if args.skew > 0:
        line["config"]["workload"] += f", join keys re-drawn from Zipf(s={args.skew}) (configs[4] shape)"
    if table_stats:
        line["aggregate_table"] = table_stats
    if xgmi:
        line["exchange"] = xgmi   # bytes this rank sent over xGMI and the time spent in the exchanges, per query
    print(json.dumps(line))
    if USE_DIST:
        dist.destroy_process_group()


def cpu_all_cores(workload, batches, threads):
    """The CPU oracle over `threads` disjoint batch ranges at once (ctypes releases the GIL; the C code keeps no shared
    state): rows/s of the wall time. Labelled context in the bench line — the reference executor is single-threaded."""
    from oracle import qoracle
    parts = [batches[i::threads] for i in range(threads)]
    plans = [getattr(queries, workload)(q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, p)) for p in parts if p]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(plans)) as ex:
        res = list(ex.map(qoracle.scan_filter_aggregate_timed, plans))
    dt = time.perf_counter() - t0
    rows = sum(r[2] for r in res)
    return {"value": rows / dt, "unit": "rows/s", "threads": len(plans), "seconds": dt, "note": "oracle on row ranges in parallel; not the reference"}


def extra_single_gpu(args, ctx, table):
    """configs[2] (TPC-H Q1 aggregate list over the resident lineitem rows) and configs[3] (Q3 at SF10) on this GPU."""
    out = {}
    try:
        plan = queries.q1_full(table)
        for _ in range(3):
            plan.execute_device()
        ctx.synchronize()
        ks = []
        t0 = time.perf_counter()
        for _ in range(10):
            plan.execute_device()
            ks.append(ctx.last_stats()["main_kernel_ms"])
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 10
        km = sum(ks) / len(ks)
        out["q1_full"] = {"workload": f"configs[2] TPC-H Q1 (2 keys, 8 aggregates) over {args.rows} resident lineitem rows",
                          "rows_per_s": args.rows / dt, "ms_per_step": dt * 1e3, "kernel_ms": km,
                          "kernel_bytes_per_row": min(ALGO_BYTES_PER_ROW["q1_full"], ctx.last_stats().get("bytes_per_row_read") or 78),
                          "roofline_GBps": args.rows * min(ALGO_BYTES_PER_ROW["q1_full"], ctx.last_stats().get("bytes_per_row_read") or 78) / (km * 1e-3) / 1e9,
                          "roofline_frac": args.rows * min(ALGO_BYTES_PER_ROW["q1_full"], ctx.last_stats().get("bytes_per_row_read") or 78) / (km * 1e-3) / 1e9 / HBM_PEAK_GBS}
        log(f"extra q1_full: kernel {km:.3f} ms")
        c, o, l = synth.q3_tables(10.0)
        tabs = (q.MemoryTable.try_new(synth.CUSTOMER_SCHEMA, c), q.MemoryTable.try_new(synth.ORDERS_SCHEMA, o),
                q.MemoryTable.try_new(synth.LINEITEM_Q3_SCHEMA, l))
        rows = [sum(b.num_rows for b in t.data) for t in tabs]
        p3 = queries.q3(*tabs)
        for _ in range(3):
            p3.execute_device()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            res = p3.execute_device()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 10
        algo = rows[0] * Q3_BYTES["customer"] + rows[1] * Q3_BYTES["orders"] + rows[2] * Q3_BYTES["lineitem"]
        out["q3_sf10"] = {"workload": "configs[3] TPC-H Q3 SF10 (two hash joins + GROUP BY) on one GPU", "rows": rows, "groups": res.num_rows,
                          "lineitem_rows_per_s": rows[2] / dt, "ms_per_query": dt * 1e3, "algorithmic_GBps": algo / dt / 1e9}
        log(f"extra q3 sf10: {dt * 1e3:.2f} ms/query")
    except Exception as e:   # the headline line must not depend on the extras
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (configs[1]: 100M)")
    ap.add_argument("--batch-rows", type=int, default=1 << 20)
    ap.add_argument("--workload", default="q1_mini", choices=["q1_mini", "q1_full", "q3"])
    ap.add_argument("--sf", type=float, default=10.0, help="TPC-H scale factor of the q3 workload (whole job, sliced over the ranks)")
    ap.add_argument("--strategy", default="broadcast", choices=["broadcast", "repartition"],
                    help="q3 on several GPUs: all-gather the small build sides (default) or repartition both sides of every join by key")
    ap.add_argument("--skew", type=float, default=0.0, help="q3: re-draw the join keys from Zipf(s) (configs[4] uses 1.1); 0 = uniform")
    ap.add_argument("--cpu-sample-rows", type=int, default=64 << 20, help="rows of the workload timed through the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra single-GPU Q1 / Q3 measurements")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("QHIP_DEVICE", str(local_rank))
    ctx = q.get_context()   # raises without a gfx950 device: there is no CPU fallback
    log(f"context on {ctx.device_name()}")

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    # QHIP_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL process group, collectives, exchange operators) even
    # with ONE rank — the rehearsal of the N > 1 launch that fits a one-GPU box (tests/test_gpu_q3.py runs it)
    global USE_DIST
    USE_DIST = world > 1 or os.environ.get("QHIP_BENCH_FORCE_DIST") == "1"
    if USE_DIST:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)

    def barrier():
        if USE_DIST:
            dist.barrier()
        torch.cuda.synchronize()

    log("torch imported")
    if args.workload == "q3":
        return bench_q3(args, ctx, rank, world, barrier, dist, torch)
    # ---- synthetic input, resident in HBM before the timed region
    t0 = time.time()
    table = gen_table(rank * args.rows, args.rows, args.batch_rows)
    t_gen = time.time() - t0
    log(f"generated {args.rows} rows in {t_gen:.1f}s")
    t0 = time.time()
    dev = table.device_table()
    t_upload = time.time() - t0
    log(f"uploaded in {t_upload:.1f}s")
    resident = sum(dev.column_bytes(c) for c in range(dev.num_columns))
    plan = getattr(queries, args.workload)(table)

    for _ in range(args.warmup):
        plan.execute_device()
        log(f"warmup step: kernel {ctx.last_stats()['main_kernel_ms']:.3f} ms")
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.execute_device()
        kernel_ms.append(ctx.last_stats()["main_kernel_ms"])
    ctx.synchronize()   # the result table of the last step is ordered on libqhip's stream
    barrier()
    elapsed = time.perf_counter() - t0
    if USE_DIST:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    stats = ctx.last_stats()
    log(f"timed {args.steps} steps in {elapsed:.3f}s")

    # ---- verification of the whole-job result (outside the timed region)
    result = plan.execute()
    partial = result_key(result)
    if USE_DIST:
        gathered = [None] * world
        dist.all_gather_object(gathered, partial)
    else:
        gathered = [partial]

    if rank != 0:
        dist.destroy_process_group()
        return

    merged = {}
    for part in gathered:
        for row in part:
            nk = len(plan.group_exprs)
            k = row[:nk]
            if k not in merged:
                merged[k] = list(row[nk:])
            elif args.workload == "q1_mini":
                merged[k] = [a + b for a, b in zip(merged[k], row[nk:])]

    total_rows = args.rows * world
    value = total_rows * args.steps / elapsed
    mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    # SURVEY §8d's algorithmic bytes per row for the Arrow layout as uploaded — unless the kernel reads FEWER bytes (it
    # skips the offsets of a Utf8 column whose every value is 1 byte long): the roofline is computed from what is read
    bpr = min(ALGO_BYTES_PER_ROW[args.workload], stats.get("bytes_per_row_read") or ALGO_BYTES_PER_ROW[args.workload])
    achieved = args.rows * bpr / (mean_kernel_ms * 1e-3) / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                traffic = json.load(f).get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:   # the CPU baseline is a single-GPU-run item (rank 0 at N = 1 only)
        from oracle import qoracle
        n = min(args.cpu_sample_rows, args.rows)
        sample = q.MemoryTable.try_new(synth.LINEITEM_SCHEMA, table.data[: max(1, n // args.batch_rows)])
        cplan = getattr(queries, args.workload)(sample)
        log(f"cpu baseline: oracle over {sum(b.num_rows for b in sample.data)} rows ...")
        cres, cdt, crows = qoracle.scan_filter_aggregate_timed(cplan)
        log(f"cpu baseline: {crows / cdt / 1e6:.2f} Mrows/s ({cdt:.1f}s)")
        # same rows through the HIP path must agree bit-exactly with the oracle
        assert result_key([cres]) == result_key(cplan.execute()), "HIP result differs from the CPU oracle on the baseline sample"
        all_cores = None
        try:   # context only: the same restatement on row ranges in parallel — NOT the reference, whose executor has no parallelism
            all_cores = cpu_all_cores(args.workload, sample.data, min(16, os.cpu_count() or 1))
            log(f"cpu baseline, {all_cores['threads']} threads (not the reference): {all_cores['value'] / 1e6:.1f} Mrows/s")
        except Exception as e:
            log(f"all-cores CPU figure not measured: {e}")
        cpu_baseline = {"value": crows / cdt, "unit": "rows/s", "cores": 1, "kind": "port", "all_cores_context": all_cores,
                        "sample": f"{crows} rows ({len(sample.data)} batches of {args.batch_rows}) of the same workload through "
                                  f"oracle/qoracle.c qo_scan_filter_aggregate, 1 of {os.cpu_count()} host cores "
                                  "(the reference executor is single-threaded)",
                        "seconds": cdt}

    stream_ceiling = None
    if rank == 0:
        try:
            stream_ceiling = ctx.measure_stream_read(4 << 30, 5)
        except Exception as e:   # a measurement aid only: never fail the bench line over it
            log(f"stream-read ceiling not measured: {e}")
    line = {
        "metric": "rows/s on TPC-H Q1 scan+agg and Q3 hash-join, SF10, 1/2/4/8 MI355X",
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i128",
        "data": "synthetic",
        "config": {"workload": f"configs[1] {args.workload}: filter+GROUP BY, {args.rows} synthetic lineitem rows per GPU in "
                               f"{args.batch_rows}-row Arrow batches, HBM-resident",
                   "rows_per_gpu": args.rows, "batch_rows": args.batch_rows, "groups": len(merged),
                   "resident_bytes_per_gpu": resident, "parallelism": f"replicated-shards x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": stats["main_kernel_name"], "kernel_ms": mean_kernel_ms,
                     "algorithmic_bytes_per_row": ALGO_BYTES_PER_ROW[args.workload], "kernel_bytes_per_row": bpr,
                     # the achievable ceiling measured with a plain 16 B/lane streaming-read kernel, same run
                     "stream_read_GBps": stream_ceiling,
                     "frac_of_stream_read": (achieved / stream_ceiling) if stream_ceiling else None},
        "cpu_baseline": cpu_baseline,
        "device": ctx.device_name(),
        "setup_s": {"generate": t_gen, "upload_h2d": t_upload, "h2d_GBps": resident / t_upload / 1e9},
    }
    if world == 1 and args.workload == "q1_mini" and not args.no_extra:
        # the other two single-GPU configurations of BASELINE.json, reported beside the headline (not part of `value`)
        line["extra"] = extra_single_gpu(args, ctx, table)
    print(json.dumps(line))
    if USE_DIST:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
