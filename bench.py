#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ipcr primer scan.

Workload (BASELINE.json configs[1], "C2"): one primer pair as `ipcr` scans it with the default
--self (3 pairs / 12 orientation slots / 4 distinct patterns), --mismatches 2,
--terminal-window 5, --max-length 2000, --hit-cap 10000, over a synthetic 3.0 Gb genome
(24 records x 125 Mb of the reference's benchDNA LCG, 1000 planted 180-bp amplicons:
exact / 1 / 2 mismatches outside the 3' window).  One STEP = one pass of the hot path over the
whole resident genome: the specialised sweep (block test, exact count, verification and hand-over
in one kernel) -> hit records on the host -> reference-order match lists -> amplicon join ->
products; the product count of EVERY pass is checked, all planted amplicons of the last one.
Inputs (the 2-bit + invalid-bit tiles) are resident in HBM when the timed region starts.
Passes are pipelined over three scratches (pipelined_passes); --no-pipeline runs them one by one.

N > 1 (torchrun, one process per GPU): weak scaling -- every rank scans its own 3 Gb genome
(seed + rank) with the same panel and joins its own records; the hit records of every pass are
exchanged by one all-gather over RCCL (straight out of the device hit buffer, two in flight), and
rank 0 joins the whole job from the gathered records of the last pass as a check.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_BASE = 0.375         # lo + hi + inv planes: the encoded tile the filter reads once
RECORDS = 24
RECORD_LEN = 125_000_000
N_PLANTS = 1000
PRODUCT_LEN = 180


def build_genome(torch, engine, workloads, revcomp_fn, genome_idx: int, records: int, record_len: int,
                 keep_host_record0: bool):
    """Synthetic genome of SURVEY.md 8(d): LCG stream seed 0x5eed1234+g cut into records, amplicons
    planted as makeEngineBenchFixture does (performance_benchmark_test.go:47-62)."""
    from ipcr_amd import workloads as W
    g = engine.Genome(records * record_len, records)
    pair = W.bench_pair(0)
    fwd = pair.Forward
    rc_rev = revcomp_fn(pair.Reverse)
    buf = torch.empty(record_len, dtype=torch.uint8, device="cuda")
    plants = []  # (record, start, n_mismatches)
    per_rec = (N_PLANTS + records - 1) // records
    stride = max((record_len - 4096) // (per_rec + 1), 400)
    host0 = None
    for r in range(records):
        engine.lcg_fill_device(buf.data_ptr(), record_len, 0x5eed1234 + genome_idx, r * record_len)
        for t in range(per_rec):
            gidx = t * records + r
            if gidx >= N_PLANTS:
                break
            start = 2048 + t * stride
            if start + PRODUCT_LEN + 64 > record_len:
                break
            nm = gidx % 3
            site = list(fwd)
            if nm >= 1:
                site[10] = W.different_base(site[10])
            if nm >= 2:
                site[3] = W.different_base(site[3])
            buf[start:start + 20] = torch.tensor(list("".join(site).encode()), dtype=torch.uint8)
            buf[start + PRODUCT_LEN - 20:start + PRODUCT_LEN] = torch.tensor(list(rc_rev), dtype=torch.uint8)
            plants.append((r, start, nm))
        torch.cuda.synchronize()
        if keep_host_record0 and r == 0:
            host0 = buf.cpu().numpy().copy()
        g.add_record_device("chr%d" % (r + 1), buf.data_ptr(), record_len)
    del buf
    return g, plants, host0


def pipelined_passes(k, scratches, begin, end, chain=None, start_exchange=None, finish_exchange=None, pipeline=True,
                     recv_slots=2):
    """k passes of the hot path over `scratches` in rotation, pipelined the way the reference's worker pool +
    collector are (internal/pipeline/pipeline.go:60-161): while the host waits for and joins pass i, the sweep of pass
    i+1 is already queued (chain(cur, prev): behind pass i's on the device); with several GPUs the all-gather of pass
    i's hit records runs under the following passes.  Invariants (tests/test_host_logic.py checks them with fakes):
      * a scratch is begun again only after its previous pass has been ended AND the exchange that reads its device
        hit buffer has finished;
      * at most `recv_slots` exchanges are in flight (the exchanger has that many receive buffers);
      * when the function returns every pass has been ended and every exchange finished.
    Returns what end() returned for the last pass."""
    n, works, ns = None, {}, len(scratches)   # works: pass -> exchange in flight (it reads that pass's scratch)
    if k <= 0:
        return n

    def do_begin(j):
        if finish_exchange is not None:
            for jj in [q for q in works if q <= j - ns]:
                finish_exchange(works.pop(jj))
        if j > 0 and pipeline and chain is not None:
            chain(scratches[j % ns], scratches[(j - 1) % ns])
        begin(scratches[j % ns])

    do_begin(0)
    for i in range(k):
        cur = scratches[i % ns]
        if i + 1 < k and pipeline:
            do_begin(i + 1)
        n = end(i, cur)
        if start_exchange is not None:
            for jj in [q for q in works if q <= i - recv_slots]:
                finish_exchange(works.pop(jj))
            works[i] = start_exchange(cur)
        if i + 1 < k and not pipeline:
            do_begin(i + 1)
    for jj in sorted(works):
        finish_exchange(works.pop(jj))
    return n


def launch_plan(gpus: int, env, argv, python: str = sys.executable, script: str = os.path.abspath(__file__)):
    """How `bench.py --gpus N` gets its N ranks (one process per GPU, internal/pipeline/pipeline.go:60-125 is the
    reference's unit of parallelism: independent records over a pool of workers).
      * RANK in the environment: this process IS a rank (torchrun or the driver started it) -> None = run here;
        WORLD_SIZE must then equal --gpus when --gpus > 1 was given.
      * no RANK and N > 1: this process only launches -- it returns the command of
        `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, which main() starts as a CHILD
        process (never exec) before anything has touched the GPU, and whose exit code it returns.
      * N <= 1: None."""
    if "RANK" in env:
        world = int(env.get("WORLD_SIZE", "1"))
        if gpus > 1 and world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: start one rank per GPU "
                             f"(python bench.py --gpus {gpus} does it itself)")
        return None
    if gpus <= 1:
        return None
    port = env.get("MASTER_PORT") or str(29500 + (os.getpid() % 400))
    return [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", port, script] + list(argv)


def main() -> None:
    # HIP maps streams onto a few hardware queues (4 by default), in creation order.  This job has a dozen (three
    # scratches, the genome, torch, RCCL): when the sweep lane shares a queue with the collective's stream, every
    # all-gather is serialised between two sweeps (+25 us per step measured).  One queue per stream instead.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # one step is ~0.2 ms: the default region (about half a second) is long enough for the clocks to settle; the
    # first ~50 steps after idle run 15-25 % slower
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--records", type=int, default=RECORDS)
    ap.add_argument("--record-len", type=int, default=RECORD_LEN)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo to rehearse)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every pass before the next one is enqueued")
    args = ap.parse_args()

    cmd = launch_plan(args.gpus, os.environ, sys.argv[1:])
    if cmd is not None:   # launcher only: no torch import, no HIP call in this process
        import subprocess
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ)))

    import numpy as np
    import torch
    from ipcr_amd import _lib, dist, engine, workloads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the ipcr_amd scan path has no CPU fallback")
    if os.environ.get("IPCR_BENCH_ONE_DEVICE"):  # rehearsal: several ranks share GPU 0 (gloo collectives)
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local, backend = dist.init_process_group(args.backend)
    torch.cuda.set_device(local)
    _lib.check(_lib.lib().ipcr_set_device(local))
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where collective payloads live
    import torch.distributed as tdist
    multi = world > 1 or bool(os.environ.get("IPCR_EXCHANGE_SELFTEST"))  # selftest: one rank runs the RCCL exchange too

    def revcomp(s: str) -> bytes:
        from ipcr_amd import primer
        return primer.RevComp(s)

    cfg = engine.Config(MaxMM=2, TerminalWindow=5, MinLen=0, MaxLen=2000, HitCap=10000, SeedLen=12)
    pairs = workloads.c2_pairs()
    eng = engine.New(cfg)
    cp = eng.CompilePanel(pairs)
    sc = eng.NewSimulationScratch(cp)

    want_cpu = (not args.no_cpu_baseline) and rank == 0 and world == 1
    genome, plants, host0 = build_genome(torch, engine, workloads, revcomp, rank, args.records, args.record_len, want_cpu)
    nrec = genome.num_records
    lens = [genome.record_len(r) for r in range(nrec)]
    flags = [genome.record_flags(r) for r in range(nrec)]
    all_lens, all_flags = dist.allgather_record_meta(lens, flags, device=cdev) if multi else (lens, flags)
    host_sc = engine.SimulationScratch(cp, host_only=True) if multi else None
    xchg = dist.HitExchanger(device=cdev) if multi else None
    rec_off = 0
    if multi:  # one synchronous exchange: sizes the buffers on every rank, yields this rank's record offset
        eng.ScanGenomeHits(genome, cp, sc)
        _, _, offs = xchg.allgather(dist.hits_from_scratch(sc), nrec)
        rec_off = offs[rank]
        xchg.agree_on_device_path(sc)   # zero-copy view of the device hit buffer on every rank, or the host copy on all

    # three scratches in rotation: one being swept, one being joined on the host, one whose hit buffer the
    # all-gather of the pass before may still be reading (several GPUs); two would do on one GPU
    scs = [sc, eng.NewSimulationScratch(cp), eng.NewSimulationScratch(cp)]
    expect_products = None
    for s_ in scs:  # untimed set-up: kernel specialisation (hiprtc) and buffer sizing happen here
        n_ = eng.ScanGenomeCount(genome, cp, s_)
        assert expect_products in (None, n_)
        expect_products = n_

    def run_steps(k):
        """k passes of the hot path (see pipelined_passes); returns (filter ms per pass, products of the last pass)"""
        fms = []

        def end(i, cur):
            n = eng.ScanGenomeEndCount(genome, cp, cur)          # this rank's partition of the join: its records
            if n != expect_products:                             # every pass is checked, not only the last one
                raise SystemExit(f"pass {i}: {n} products, the set-up scan found {expect_products}")
            fms.append(cur.stats().filter_ms)
            return n

        n = pipelined_passes(
            k, scs,
            begin=lambda cur: eng.ScanGenomeBegin(genome, cp, cur),
            end=end,
            chain=None if (args.no_pipeline or os.environ.get("IPCR_BENCH_NO_CHAIN")) else (lambda cur, prev: cur.chain_after(prev)),
            start_exchange=(lambda cur: xchg.start_scratch(cur, nrec)) if multi else None,   # all-gatherv of hit records, async
            finish_exchange=xchg.finish if multi else None,
            pipeline=not args.no_pipeline)
        return fms, n

    # part of the untimed set-up: the first ~50 sweeps after idle run 15-25 % slower (clock ramp);
    # whatever --warmup the caller picks, the device has done at least 300 passes before the timed region
    run_steps(max(0, 300 - max(args.warmup, 0)))
    run_steps(max(args.warmup, 0))

    if multi:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    filter_ms, nprod = run_steps(args.steps)
    torch.cuda.synchronize()
    if multi:
        tdist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        tdist.all_reduce(tmax, op=tdist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ptot = torch.tensor([nprod], dtype=torch.int64, device=cdev)
        tdist.all_reduce(ptot, op=tdist.ReduceOp.SUM)
        nprod = int(ptot.item())
    last = scs[(args.steps - 1) % len(scs)]   # scratch holding the last pass

    # ---- correctness outside the timed region: every planted amplicon must come back exactly ----
    if not multi:
        prods = last.products(genome.ids)
    else:  # rank 0 joins the WHOLE job from the gathered hits and checks its own genome's plants
        allhits, _, _ = xchg.gathered()                     # what the last step's all-gatherv left on every rank
        prods = eng.JoinHits(cp, host_sc, allhits, all_lens, all_flags) if rank == 0 else []
        if rank == 0:
            assert len(prods) == nprod, f"whole-job join on rank 0 found {len(prods)} products, partitioned join {nprod}"
    if rank == 0:
        found = {(p.Record, p.Start): p for p in prods if p.ExperimentID == "bench_000" and p.Type == "forward" and p.Length == PRODUCT_LEN}
        for (r, start, nm) in plants:           # rank 0's own genome occupies records [0, nrec)
            p = found.get((r, start))
            assert p is not None, f"planted amplicon missing: record {r} start {start}"
            want_idx = () if nm == 0 else ((10,) if nm == 1 else (3, 10))
            assert p.FwdMM == nm and p.FwdMismatchIdx == want_idx and p.RevMM == 0, (p, nm)

    total_bases = genome.total_bases * world
    ms_per_step = elapsed * 1e3 / args.steps
    value = total_bases * args.steps / elapsed / 1e9
    fms_avg = sum(filter_ms) / len(filter_ms)
    alg_bytes = genome.total_bases * BYTES_PER_BASE
    achieved = alg_bytes / (fms_avg * 1e-3) / 1e9
    traffic = None
    import glob
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_filter_pmc.json")))   # latest round's PMC passes
    tfile = pmc_files[-1] if pmc_files else ""
    if tfile and os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "genome Gbases scanned/sec (k=2)",
        "value": round(value, 2),
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": "C2: 1 primer pair (+self pairs: 12 orientation slots, 4 distinct patterns), k=2, "
                        "3'-window=5, hit-cap 10000, max-length 2000, synthetic %.2f Gb genome per GPU "
                        "(%d x %d b LCG records, %d planted amplicons)" % (genome.total_bases / 1e9, nrec, args.record_len, len(plants)),
            "input": "2-bit + invalid-bit tiles resident in HBM (0.375 B/base); pack kernel timed separately",
            "products_per_step": int(nprod),
            "hits_per_step_rank0": int(last.stats().hits),
            "filter_candidates_rank0": int(last.stats().candidates),
            "filter_kernel": "panel-specialised (hiprtc)" if last.stats().kernel_kind == 1 else "table-driven",
            "pipelining": "off" if args.no_pipeline else
                          "pass i+1's sweep is queued behind pass i's on one in-order stream (two scratches) while the host "
                          "waits for, sorts and joins pass i; each sweep verifies its own survivors and its last wave "
                          "publishes counters + hits to pinned memory (no verify kernel, no copy operation)",
            "pack_ms_per_genome": round(genome.pack_ms, 3),
            "gbases_per_s_incl_pack": round(genome.total_bases * world / ((genome.pack_ms + ms_per_step) * 1e-3) / 1e9, 1),
            "step_breakdown_ms_rank0": {k: round(getattr(last.stats(), k), 4) for k in
                                        ("filter_ms", "verify_ms", "enqueue_ms", "wait_ms", "sort_ms", "join_ms", "total_ms")},
            "parallelism": ("1 genome per GPU, one all-gatherv of hit records per step (%s, %s), join partitioned by record"
                            % (backend, "out of the device hit buffer" if xchg.device_path else "host copy of the hits"))
                           if multi else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "ipcr_filter",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "frac_of_copy_peak": round(achieved / 6290.0, 4),  # 6.29 TB/s: what a float4 copy reaches (MI355X_MICROARCH.md)
            "avg_launch_ms": round(fms_avg, 4),
            "gbases_per_s_kernel": round(genome.total_bases / (fms_avg * 1e-3) / 1e9, 1),
        },
    }

    if want_cpu:
        out["cpu_baseline"] = cpu_baseline(host0, args.cpu_seconds, prods)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        tdist.barrier()
        tdist.destroy_process_group()


def _join_count(eng, cp, host_sc, hits, lens, flags) -> int:
    """join without materialising Python objects per product"""
    import ctypes as C
    from ipcr_amd import _lib
    n, nrec = len(hits), len(lens)
    lens_c = (C.c_uint64 * max(nrec, 1))(*lens)
    flags_c = (C.c_uint8 * max(nrec, 1))(*flags)
    ptr = C.c_void_p(hits.ctypes.data) if n else None
    _lib.check(_lib.lib().ipcr_join_hits(cp._h, host_sc._h, ptr, n, lens_c, flags_c, nrec, None, None))
    return host_sc.num_products()


def cpu_baseline(host0, budget_s: float, gpu_products):
    """Reference algorithm restated in C (oracle/: approximate-seed Aho-Corasick scan + verify +
    join, one worker per rolling chunk like internal/pipeline/pipeline.go:60-125) timed on this
    box's host cores over a bounded sample: record 0 of the same genome, repeated to fill the
    budget.  Checker/baseline only -- never on the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ipcr_oracle as O
    from ipcr_amd import workloads
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    pairs = [O.Pair(p.ID, p.Forward, p.Reverse, p.MinProduct, p.MaxProduct) for p in workloads.c2_pairs()]
    panel = O.Panel(O.Config(max_mm=2, terminal_window=5, min_len=0, max_len=2000, hit_cap=10000, seed_len=12), pairs)
    n = int(host0.shape[0])
    ptr = host0.ctypes.data
    passes, t0, nprod = 0, time.perf_counter(), 0
    while True:
        nprod = panel.baseline_scan_mt(ptr, n, 4_000_000, 2000, cores)
        passes += 1
        el = time.perf_counter() - t0
        if el >= budget_s or passes >= 4096:
            break
    gpu_rec0 = len([p for p in gpu_products if p.Record == 0])
    assert nprod == gpu_rec0, f"CPU baseline found {nprod} products in record 0, GPU path {gpu_rec0}"
    return {
        "value": round(n * passes / el / 1e9, 4),
        "unit": "Gbases/s",
        "cores": cores,
        "kind": "port",
        "sample": "record 0 (%d bases) of the same genome, %d passes, chunk 4 Mb / overlap 2000, %d threads; "
                  "C restatement of the reference's seeded AC scan + verify + join (not the Go binary)" % (n, passes, cores),
        "products_in_sample": int(nprod),
    }


if __name__ == "__main__":
    main()
